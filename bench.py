#!/usr/bin/env python
"""bench.py — images/s of the frozen integer DeiT-S forward (batch 256 per GPU, 224x224
synthetic int8) on N MI355X, one process per GPU, weights broadcast once over RCCL.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement): whole-job images/s,
`roofline` for the dominant kernel class (int8 MFMA GEMMs, HIP-event timed on the
launch stream) and `cpu_baseline` (the CPU oracle port timed on the host cores).
A "step" = one forward of the hot path over one resident batch.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_PEAK_TOPS = 5033.0   # 256 CU x 4 SIMD x 2048 OP/clk x 2.4 GHz (dense; = 2x bf16 peak)
HBM_PEAK_GBS = 8000.0


def model_ops_per_image(cfg):
    """algorithmic int8 OPs (2 x MAC) per image, SURVEY.md §8(d)."""
    T, D, H, dh, Hd = cfg.num_tokens, cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden_dim
    Kp = cfg.in_chans * cfg.patch_size ** 2
    lin = cfg.num_patches * Kp * D + cfg.depth * T * (3 * D * D + D * D + 2 * D * Hd) + D * cfg.num_classes
    bmm = cfg.depth * 2 * H * T * T * dh
    return 2 * lin, 2 * bmm


class EventTimer:
    """Brackets every C-ABI call with HIP events on the launch stream (torch.cuda.Event
    records on torch's current stream = the stream the handle launches on)."""

    def __init__(self, handle, torch):
        self.h, self.torch = handle, torch
        self.records = []
        self._orig = handle.call

    def __enter__(self):
        def call(name, *args):
            a = self.torch.cuda.Event(enable_timing=True)
            b = self.torch.cuda.Event(enable_timing=True)
            a.record()
            self._orig(name, *args)
            b.record()
            self.records.append((name, args, a, b))
        self.h.call = call
        return self

    def __exit__(self, *exc):
        self.h.call = self._orig

    def summary(self):
        self.torch.cuda.synchronize()
        out = {}
        for name, args, a, b in self.records:
            d = out.setdefault(name, [0.0, 0])
            d[0] += a.elapsed_time(b)
            d[1] += 1
        return out


def pmc_traffic():
    """HBM bytes per GEMM launch from the committed rocprofv3 PMC passes of this command
    (profiles/pmc_traffic.json: FETCH_SIZE x2-corrected + WRITE_SIZE); None if absent."""
    try:
        return round(json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["avg_bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(cfg, weights, scales, target_seconds=15.0):
    """CPU oracle (port of the reference algorithm, OpenMP over the host cores) on a
    bounded sample of the same workload."""
    from oracle import oracle as orc
    import ivit_amd as iv
    o = orc.OracleViT(cfg, weights, scales)
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    imgs = iv.make_images_int8(cfg, 2, seed=1)
    t = time.time()
    o.forward(imgs)
    per_img = (time.time() - t) / 2
    n = int(max(2, min(64, target_seconds / max(per_img, 1e-3))))
    imgs = iv.make_images_int8(cfg, n, seed=1)
    t = time.time()
    o.forward(imgs)
    dt = time.time() - t
    return {"value": round(n / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} images of the same {cfg.name} int8 forward, oracle/ivit_oracle.c (OpenMP, {cores} threads), {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="deit_small")
    ap.add_argument("--batch", type=int, default=256, help="images per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("IVIT_STREAMS", "4")),
                    help="batch slices on the runner's internal HIP streams (VALU/MFMA overlap)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("IVIT_GRAPH", "1")), help="replay a captured hipGraph")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import ivit_amd as iv
    from ivit_amd.engine import ViTEngine, pack_constants
    from ivit_amd import dist as ivdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; IVIT_DIST_BACKEND=gloo only for single-GPU plumbing tests
        dist.init_process_group(os.environ.get("IVIT_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    cfg = iv.CONFIGS[args.model]
    gname = {"deit_small": "deit_small_b4.npz", "deit_tiny": "deit_tiny_b1.npz"}[args.model]
    g = np.load(os.path.join(ROOT, "tests", "golden", gname))
    scales = {k[len("scale/"):]: np.float32(g[k]) for k in g.files if k.startswith("scale/")}
    weights = None
    if rank == 0:
        weights = iv.make_vit_weights(cfg, int(g["seed"]))
    # rank 0 freezes; the packed integer constants travel once over RCCL (xGMI)
    eng = ivdist.build_engine_broadcast(cfg, weights, scales, device, rank, world)

    # per-GPU batch is fixed (weak scaling); every rank owns different images
    imgs = torch.from_numpy(iv.make_images_int8(cfg, args.batch, seed=1 + rank)).to(device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.graph:
        step = eng.capture(imgs, args.streams)
    else:
        step = lambda: eng.forward(imgs, nslices=args.streams)
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt

    # parity guard inside the bench: first 4 images of rank 0 are the golden batch
    ok = None
    if rank == 0 and args.model == "deit_small" and args.batch >= 4:
        logits = step()[:4].cpu().numpy()
        ok = bool(np.array_equal(logits, g["logits_int"]))

    # per-kernel HIP-event timing (separate instrumented steps, same stream)
    per = {}
    if rank == 0 and args.profile_steps > 0:
        with EventTimer(eng.h, torch) as et:
            for _ in range(args.profile_steps):
                eng.forward_ops(imgs)      # same kernels, one C-ABI call per operator
            per = et.summary()
    if rank == 0:
        lin_ops, bmm_ops = model_ops_per_image(cfg)
        gemm_names = ["ivit_linear_i8_requant", "ivit_linear_i8_qkv", "ivit_linear_i8_requant_residual",
                      "ivit_linear_i8"]
        g_ms = sum(per.get(n, [0, 0])[0] for n in gemm_names) / max(args.profile_steps, 1)
        g_n = sum(per.get(n, [0, 0])[1] for n in gemm_names) / max(args.profile_steps, 1)
        achieved = (lin_ops * args.batch / (g_ms * 1e-3) / 1e12) if g_ms > 0 else None
        roofline = {
            "kernel": "gemm_glds_kernel (QuantLinear GEMMs with fused requant epilogues: patch-embed, qkv, proj, fc1, fc2) + head",
            "bound": "mfma", "achieved": None if achieved is None else round(achieved, 1),
            "peak": INT8_PEAK_TOPS, "unit": "TOP/s",
            "frac": None if achieved is None else round(achieved / INT8_PEAK_TOPS, 4),
            "traffic": pmc_traffic(),
            "launches_per_step": g_n, "ms_per_step_in_kernel": round(g_ms, 4),
            "avg_launch_ms": round(g_ms / g_n, 5) if g_n else None,
            "algorithmic_ops_per_step": lin_ops * args.batch,
            "mfma_ubench_ceiling_tops_random_operands": 3400.0,   # tools/ubench/mfma_peak.hip, profiles/README.md
        }
        breakdown = {n: {"ms_per_step": round(v[0] / args.profile_steps, 4), "launches": v[1] // args.profile_steps}
                     for n, v in sorted(per.items(), key=lambda kv: -kv[1][0])}
        out = {
            "metric": "images/sec + int8-MFMA-roofline% for DeiT-S bs256@224 on 1/2/4/8 MI355X",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": f"{cfg.name} int8 forward, batch {args.batch}/GPU, {cfg.img_size}x{cfg.img_size}x3 synthetic int8 "
                                   f"(BASELINE.json configs[1]); weights seeded synthetic, activation scales from the reference calibration",
                       "global_batch": args.batch * world, "parallelism": f"dp{world} (batch-sharded, weights RCCL-broadcast once)",
                       "streams_per_gpu": args.streams, "hipgraph": bool(args.graph)},
            "model_int8_tops": round((lin_ops + bmm_ops) * args.batch * world / (ms_per_step * 1e-3) / 1e12, 1),
            "model_roofline_frac": round((lin_ops + bmm_ops) * args.batch / (ms_per_step * 1e-3) / 1e12 / INT8_PEAK_TOPS, 4),
            "bit_exact_vs_reference_golden": ok,
            "roofline": roofline,
            "kernel_breakdown_ms": breakdown,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg, weights, scales)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
