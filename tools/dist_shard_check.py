"""Run under torch.distributed.run with N ranks (gloo or nccl): every rank receives the broadcast integer constants
(ivit_amd.dist.build_engine_broadcast), runs the HIP ENGINE on ITS shard of one seeded batch, the shards' logits are gathered
on rank 0 and compared with (i) the unsharded forward of the same engine and (ii) the reference's golden logits of the images
that open the batch.  Prints `SHARD_CHECK_OK ...` on rank 0, exits non-zero on any difference.  With one GPU all ranks share
device 0 (IVIT_DIST_BACKEND=gloo): the partitioning, the broadcast and the gather are what is tested, not the scaling.
Used by tests/test_gpu_parity.py::test_engine_forward_sharded_over_two_ranks."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ivit_amd as iv  # noqa: E402
from ivit_amd import dist as ivdist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("IVIT_DIST_BACKEND", "nccl")
    local = int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    dist.init_process_group(backend, rank=rank, world_size=world)
    name, total = sys.argv[1], int(sys.argv[2])
    g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", name))
    cfg = iv.CONFIGS[str(g["cfg_name"])]
    scales = {k[len("scale/"):]: np.float32(g[k]) for k in g.files if k.startswith("scale/")}
    weights = iv.make_vit_weights(cfg, int(g["seed"])) if rank == 0 else None       # only rank 0 owns the weights
    eng = ivdist.build_engine_broadcast(cfg, weights, scales, device, rank, world)
    gb = int(g["batch"])
    images = np.concatenate([iv.make_images_int8(cfg, gb, int(g["images_seed"])), iv.make_images_int8(cfg, total - gb, seed=77)])
    lo, hi = ivdist.shard_range(total, rank, world)
    mine = eng.forward(torch.from_numpy(np.ascontiguousarray(images[lo:hi])).to(device), copy=True)
    pad = max(b - a for a, b in (ivdist.shard_range(total, r, world) for r in range(world)))
    on = device if backend == "nccl" else "cpu"
    buf = torch.zeros(pad, cfg.num_classes, dtype=torch.int32, device=on)
    buf[:hi - lo] = mine.to(on)
    gathered = [torch.zeros_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf)
    ok = True
    if rank == 0:
        parts = []
        for r in range(world):
            a, b = ivdist.shard_range(total, r, world)
            parts.append(gathered[r][:b - a].cpu().numpy())
        got = np.concatenate(parts)
        full = eng.forward(torch.from_numpy(np.ascontiguousarray(images)).to(device), copy=True).cpu().numpy()
        ok = bool(np.array_equal(got, full)) and bool(np.array_equal(got[:gb], g["logits_int"]))
        print(f"SHARD_CHECK_{'OK' if ok else 'FAIL'} world {world} shards {[ivdist.shard_range(total, r, world) for r in range(world)]} "
              f"backend {backend}", flush=True)
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=on)
    dist.broadcast(flag, src=0)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if int(flag.item()) else 1)


if __name__ == "__main__":
    main()
