"""world_size-2 gloo test of the multi-GPU plumbing (runs on CPU): rank 0 freezes and packs
the integer constants, every rank receives identical bytes; image shards are disjoint,
contiguous and cover the batch."""
import hashlib
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_golden, golden_scales
import ivit_amd as iv
from ivit_amd import dist as ivdist
from ivit_amd.engine import pack_constants


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load_golden("micro_vit_b2.npz")
    cfg = iv.CONFIGS[str(g["cfg_name"])]
    consts = f32 = None
    if rank == 0:
        consts, f32 = iv.freeze.freeze_vit(cfg, iv.make_vit_weights(cfg, int(g["seed"])), golden_scales(g))
    blob, table, f32r = ivdist.broadcast_constants(consts, f32, rank, world, "cpu")
    digest = hashlib.sha256(blob.numpy().tobytes()).hexdigest()
    lo, hi = ivdist.shard_range(37, rank, world)
    q.put((rank, digest, sorted(table)[:3], len(table), f32r["ln.s"], lo, hi))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_and_sharding_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, d0, k0, n0, s0, lo0, hi0), (r1, d1, k1, n1, s1, lo1, hi1) = res
    assert d0 == d1 and k0 == k1 and n0 == n1 and s0 == s1
    assert (lo0, hi0, lo1, hi1) == (0, 19, 19, 37)
    # the broadcast bytes are exactly rank 0's packed constants
    g = load_golden("micro_vit_b2.npz")
    cfg = iv.CONFIGS[str(g["cfg_name"])]
    consts, _ = iv.freeze.freeze_vit(cfg, iv.make_vit_weights(cfg, int(g["seed"])), golden_scales(g))
    blob, _ = pack_constants(consts)
    assert hashlib.sha256(blob.tobytes()).hexdigest() == d0


def _swin_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ivit_amd.swin_engine import freeze_swin, pack_swin_constants
    g = load_golden("micro_swin_b2.npz")
    cfg = iv.SWIN_CONFIGS[str(g["cfg_name"])]
    packed = None
    if rank == 0:
        packed = pack_swin_constants(freeze_swin(cfg, iv.make_swin_weights(cfg, int(g["seed"])), golden_scales(g)))
    blob, table, host = ivdist.broadcast_packed(packed, rank, world, "cpu")
    q.put((rank, hashlib.sha256(blob.numpy().tobytes()).hexdigest(), len(table), sorted(host)[:4], host["dy_pool"]))
    dist.barrier()
    dist.destroy_process_group()


def test_swin_constants_broadcast_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_swin_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1:] == res[1][1:]
    assert res[0][4][0] == "dy" and res[0][2] > 50


def test_shard_range_covers_batch():
    for total in (1, 7, 256, 513):
        for world in (1, 2, 3, 8):
            parts = [ivdist.shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def _shard_forward_worker(rank, world, port, q):
    """every rank: receive the broadcast constants' source (scales + seeded weights are rank-0 state), run the CPU
    oracle on ITS shard of the batch, gather the logits."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    g = load_golden("micro_vit_b2.npz")
    cfg = iv.CONFIGS[str(g["cfg_name"])]
    # rank 0 owns weights and scales; the others get them through the collective (as the constants blob travels)
    payload = [None]
    if rank == 0:
        payload = [(iv.make_vit_weights(cfg, int(g["seed"])), golden_scales(g))]
    dist.broadcast_object_list(payload, src=0)
    weights, scales = payload[0]
    total = 5                                   # ragged on purpose: shards of 3 and 2
    images = iv.make_images_int8(cfg, total, seed=11)
    lo, hi = ivdist.shard_range(total, rank, world)
    logits, _ = orc.OracleViT(cfg, weights, scales).forward(images[lo:hi])
    mine = torch.zeros(3, cfg.num_classes, dtype=torch.int32)       # padded to the largest shard
    mine[:hi - lo] = torch.from_numpy(np.asarray(logits, dtype=np.int32))
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    if rank == 0:
        parts = []
        for r in range(world):
            a, b = ivdist.shard_range(total, r, world)
            parts.append(gathered[r][:b - a].numpy())
        full, _ = orc.OracleViT(cfg, weights, scales).forward(images)
        q.put(bool(np.array_equal(np.concatenate(parts), np.asarray(full, dtype=np.int32))))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_forward_equals_unsharded_world2():
    """SURVEY §4 multi-GPU row: two shards run independently (no per-step collective), the gathered logits are the
    single-process result bit for bit."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard_forward_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    ok = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_bench_refuses_missing_devices():
    """`python bench.py --gpus 2` without a launcher starts the ranks itself and fails loudly when the node has fewer
    devices — it must never report a 1-GPU number as n_gpus: 2."""
    import subprocess
    import sys
    from conftest import ROOT
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "--gpus 2" in (r.stderr + r.stdout)
