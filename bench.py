#!/usr/bin/env python
"""bench.py — images/s of the frozen integer ViT / Swin forward on N MI355X, one process per GPU,
integer constants broadcast once over RCCL (xGMI), images sharded by rank, no per-step collective.

    python bench.py                                   # DeiT-S b256, 1 GPU (BASELINE.json configs[1])
    python bench.py --gpus 8 --steps 20 --warmup 3    # launches 8 ranks itself (torch.distributed.run)
    python bench.py --model {deit_tiny,deit_small,deit_base,swin_tiny,vit_base_384}
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W        # what the driver runs for N > 1

Prints ONE JSON line on rank 0 (contract in the task statement).  A "step" = one forward of the hot path
over one resident per-GPU batch.  The timed region (EXACTLY K steps between barriers, max over ranks) is repeated
until at least `--min-seconds` (default 3 s) of timed GPU work has accumulated and at least `--reps` times, so that
the clocks are in steady state; `value` is the MEDIAN repetition (all repetitions are listed).  `roofline`: the int8 MFMA GEMM class and the HBM-bound operators, each timed with HIP events
on the launch stream while the same forward is issued ONE C-ABI CALL PER OPERATOR on a single stream
(`timed_on`), which is not the sliced / hipGraph path that produced `value`.  `cpu_baseline` (N = 1 only): the
PyTorch-CPU counterpart of the reference's fake-quant path and the OpenMP C port of the integer algorithm on the host
cores, plus the DeiT-T batch-1 latency of BASELINE.json configs[0].
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

INT8_PEAK_TOPS = 5033.0   # 256 CU x 4 SIMD x 2048 OP/clk x 2.4 GHz (dense; = 2x bf16 peak)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 achievable)

# model -> (family, golden fixture, images per GPU of the BASELINE.json config, which config)
WORKLOADS = {
    "deit_tiny": ("vit", "deit_tiny_b1.npz", 1, "configs[0]: DeiT-T b1 plumbing baseline"),
    "deit_small": ("vit", "deit_small_b4.npz", 256, "configs[1]: DeiT-S b256 on one GPU"),
    "deit_base": ("vit", "deit_base_b2.npz", 64, "configs[2]: DeiT-B b512 over 8 GPUs = 64 per GPU"),
    "swin_tiny": ("swin", "swin_tiny_b1.npz", 256, "configs[3]: Swin-T b256 on one GPU"),
    "vit_base_384": ("vit", "vit_base_384_b1.npz", 128, "configs[4]: ViT-B@384 b1024 over 8 GPUs = 128 per GPU"),
}


def vit_ops_per_image(cfg):
    """algorithmic int8 OPs (2 x MAC) per image, SURVEY.md §8(d): (linear, bmm)."""
    T, D, H, dh, Hd = cfg.num_tokens, cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden_dim
    Kp = cfg.in_chans * cfg.patch_size ** 2
    lin = cfg.num_patches * Kp * D + cfg.depth * T * (3 * D * D + D * D + 2 * D * Hd) + D * cfg.num_classes
    bmm = cfg.depth * 2 * H * T * T * dh
    return 2 * lin, 2 * bmm


def swin_ops_per_image(cfg):
    """Swin: linear layers + patch merging + windowed attention (49-token windows)."""
    lin = cfg.grid ** 2 * cfg.in_chans * cfg.patch_size ** 2 * cfg.embed_dim
    bmm = 0
    res, C = cfg.grid, cfg.embed_dim
    for li, depth in enumerate(cfg.depths):
        L, heads = res * res, cfg.num_heads[li]
        lin += depth * L * (3 * C * C + C * C + 2 * C * cfg.mlp_ratio * C)
        bmm += depth * 2 * (L // cfg.window_size ** 2) * heads * (cfg.window_size ** 2) ** 2 * (C // heads)
        if li < len(cfg.depths) - 1:
            lin += (L // 4) * 4 * C * 2 * C
            res, C = res // 2, C * 2
    lin += C * cfg.num_classes
    return 2 * lin, 2 * bmm


class EventTimer:
    """Brackets every C-ABI call with HIP events on the launch stream (torch.cuda.Event records on torch's
    current stream = the stream the handle launches on)."""

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
            self.records.append((name, a, b))
        self.h.call = call
        return self

    def __exit__(self, *exc):
        self.h.call = self._orig

    def summary(self):
        self.torch.cuda.synchronize()
        out = {}
        for name, a, b in self.records:
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


def measure_traffic_live(model, timeout_s=150):
    """HBM bytes per GEMM-class launch measured BY THIS RUN: two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE — the two do not fit one
    pass on gfx950) of a short single-stream eager forward of this very script, as subprocesses after the timed region.  Collected and
    corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes: separate --pmc passes, KiB units, FETCH_SIZE doubled (the counter
    tallies 128-byte requests at 64 B for wide coalesced reads).  Returns the dict of profiles/pmc_traffic.json, or None on any
    failure (no rocprofv3, a refused counter, a timeout): the caller then prints the committed file and says so."""
    import csv
    import glob
    import re
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    tmp = tempfile.mkdtemp(prefix="ivit_pmc_", dir="/tmp")
    cmd_tail = [sys.executable, os.path.abspath(__file__), "--model", model, "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                "--profile-steps", "0", "--streams", "1", "--graph", "0", "--reps", "1", "--min-seconds", "0", "--box-probe", "0",
                "--measure-traffic", "0"]
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            r = subprocess.run([rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--"] + cmd_tail,
                               cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                               timeout=timeout_s)
            fs = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not fs:
                return None
            for row in csv.DictReader(open(fs[0])):
                if row["Counter_Name"] != counter:
                    continue
                name = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")[:70]
                if name.startswith("gemm_") or name.startswith("mlp384"):
                    per.setdefault(name, {}).setdefault(counter, []).append(float(row["Counter_Value"]))
        out, tot_b, tot_n = {}, 0.0, 0
        for k, v in per.items():
            if "FETCH_SIZE" not in v or "WRITE_SIZE" not in v:
                return None
            n = len(v["FETCH_SIZE"])
            fb = sum(v["FETCH_SIZE"]) / n * 1024 * 2
            wb = sum(v["WRITE_SIZE"]) / len(v["WRITE_SIZE"]) * 1024
            out[k] = {"launches": n, "fetch_MB_x2": round(fb / 1e6, 2), "write_MB": round(wb / 1e6, 2)}
            tot_b += (fb + wb) * n
            tot_n += n
        if not tot_n:
            return None
        return {"per_kernel": out, "avg_bytes_per_launch": tot_b / tot_n}
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pmc_traffic_per_kernel(batch, T, D, Hd, per=None):
    """The same per kernel of the GEMM class, next to the algorithmic bytes of that launch (DeiT-S shapes): a class
    average hides one kernel's wasted re-reads behind another's clean stream."""
    try:
        per = per if per is not None else json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["per_kernel"]
    except Exception:
        return None
    M = batch * T
    alg = {"mlp384": M * D + 2 * 2 * M * D + 2 * D * Hd,                        # mlp384_kernel / mlp384rs_kernel: x, identity in, out, both weight matrices
           "gemm_as_kernel<5": M * D + 3 * M * D + 3 * D * D,                   # qkv: x, q / k / v^T, weights
           "gemm_glds_kernel<3": M * D + 2 * 2 * M * D + D * D,                 # proj: ctx, identity in, out, weights
           "gemm_ws_qkv_kernel<true, true, 0": 2 * M * D + 3 * M * D + 3 * D * D,     # norm1 + qkv: the 16-bit rows, q / k / v, weights
           "gemm_ws_qkv_kernel<false, true, 0": 2 * M * D + 3 * M * D + 3 * D * D,
           "gemm_ws_qkv_kernel<true, false, 0": M * D + 3 * M * D + 3 * D * D,        # qkv alone on the same kernel
           "gemm_ws_qkv_kernel<false, false, 0": M * D + 3 * M * D + 3 * D * D,
           "gemm_ws_qkv_kernel<true, false, 1": M * D + 2 * 2 * M * D + D * D,        # proj + residual on the same kernel
           "gemm_ws_qkv_kernel<false, false, 1": M * D + 2 * 2 * M * D + D * D}
    out = {}
    for k, v in per.items():
        a = next((b for pfx, b in alg.items() if k.startswith(pfx)), None)
        hbm = (v["fetch_MB_x2"] + v["write_MB"]) * 1e6
        out[k] = {"hbm_bytes": round(hbm), "algorithmic_bytes": a, "ratio": None if not a else round(hbm / a, 3)}
    return out


def box_probe(device_index, target_ms=50.0):
    """Three numbers that identify the box, measured in THIS process right after the timed region (tools/ubench/box_probe.hip,
    built by __graft_entry__.build()): register-only int8 MFMA loops on random operands (~50 ms each) and a 256 MB device copy.
    Boxes of one pool differ by several percent in sustained clock under MFMA load; without this a faster box and a faster
    kernel are indistinguishable in a single bench line.  None if the probe library is absent and cannot be built."""
    import ctypes
    so = os.path.join(ROOT, "tools", "ubench", "libbox_probe.so")
    src = os.path.join(ROOT, "tools", "ubench", "box_probe.hip")
    try:
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-fPIC", "-shared",
                                   src, "-o", so], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        lib = ctypes.CDLL(so)
        lib.box_probe.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.POINTER(ctypes.c_double)]
        out = (ctypes.c_double * 4)()
        rc = lib.box_probe(device_index, target_ms, out)
        if rc != 0:
            return {"error": f"box_probe rc {rc}"}
        return {"mfma_tops_random": {"32x32x32_i8": round(out[0], 1), "16x16x64_i8": round(out[1], 1)},
                "copy_gbs": round(out[2], 1), "cus": int(out[3]),
                "what": "tools/ubench/box_probe.hip run in this process after the timed region: register-only MFMA loops on random "
                        "int8 operands (4 waves per SIMD), 256 MB device copy (read + write bytes), ~50 ms each"}
    except Exception as e:                    # the probe must never take the bench line down
        return {"error": str(e)}


def _timed_forward(fwd, make_images, target_seconds, chunk=8, max_images=256):
    """images/s of `fwd` on a time-bounded sample: chunks of `chunk` images until ~target_seconds have passed (never more
    than `max_images`, never less than one chunk); the first chunk is a warm-up and is not counted unless it is the only one"""
    imgs = make_images(chunk)
    t0 = time.time()
    fwd(imgs)
    first = time.time() - t0
    n, dt = 0, 0.0
    while dt < target_seconds and n < max_images and first < 4 * target_seconds:
        t = time.time()
        fwd(imgs)
        dt += time.time() - t
        n += chunk
    return (n, dt) if n else (chunk, first)


def cpu_baseline(family, cfg, weights, scales, golden_dir, target_seconds=12.0):
    """The CPU side of the comparison, on the GPU box's host cores, on a bounded sample of the same workload:
    `value` = the PyTorch-CPU counterpart of the reference's fake-quant path (oracle/torch_ref.py: the reference's op
    sequence on torch CPU operators, pinned to the reference's logits) where one exists (ViT family), next to the OpenMP C
    port of the integer algorithm (oracle/ivit_oracle.c) and the DeiT-T batch-1 latency of BASELINE.json configs[0]."""
    import torch
    from oracle import oracle as orc
    import ivit_amd as iv
    cores = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(cores))
    mk = lambda n: iv.make_images_int8(cfg, n, seed=1)
    o = orc.OracleViT(cfg, weights, scales) if family == "vit" else orc.OracleSwin(cfg, weights, scales)
    n_c, dt_c = _timed_forward(o.forward, mk, target_seconds if family != "vit" else target_seconds / 2, chunk=2, max_images=64)
    c_port = {"value": round(n_c / dt_c, 3), "unit": "images/s", "cores": cores, "kind": "port",
              "sample": f"{n_c} images of the same {cfg.name} int8 forward, oracle/ivit_oracle.c (OpenMP, {cores} threads), {dt_c:.1f} s"}
    if family != "vit":
        return c_port
    from oracle.torch_ref import TorchRefViT
    # torch's CPU operators on these tensor sizes are fastest at ~16 threads on the 256-thread host (measured: 4 DeiT-S images
    # take 0.16 s at 16 threads, 0.30 s at 32, 0.70 s at 64): `cores` reports the threads actually used
    threads = min(cores, 16)
    torch.set_num_threads(threads)
    tr = TorchRefViT(cfg, weights, scales)
    n_t, dt_t = _timed_forward(tr.forward, mk, target_seconds)
    out = {"value": round(n_t / dt_t, 3), "unit": "images/s", "cores": threads, "kind": "port",
           "sample": f"{n_t} images of the same {cfg.name} forward through oracle/torch_ref.py — the reference's fake-quant op sequence "
                     f"on torch {torch.__version__} CPU operators ({torch.get_num_threads()} threads), {dt_t:.1f} s",
           "c_port": c_port}
    try:      # BASELINE.json configs[0]: DeiT-T, batch 1 (the reference's own CPU-runnable case), median of 5
        gt = np.load(os.path.join(golden_dir, "deit_tiny_b1.npz"))
        cfg_t = iv.CONFIGS["deit_tiny"]
        tr_t = TorchRefViT(cfg_t, iv.make_vit_weights(cfg_t, int(gt["seed"])),
                           {k[len("scale/"):]: np.float32(gt[k]) for k in gt.files if k.startswith("scale/")})
        img1 = iv.make_images_int8(cfg_t, 1, int(gt["images_seed"]))
        lat = []
        for _ in range(6):
            t = time.time()
            lg, sc = tr_t.forward(img1)
            lat.append(time.time() - t)
        exact = bool(np.array_equal(torch.round(lg / sc).numpy().astype(np.int64), gt["logits_int"]))
        out["deit_tiny_b1"] = {"ms_per_image_median": round(sorted(lat[1:])[2] * 1e3, 2), "bit_exact_vs_reference_golden": exact,
                               "what": "BASELINE.json configs[0]: DeiT-T int8 forward, batch 1, torch-CPU counterpart"}
    except Exception as e:          # a missing fixture must not take the bench line down
        out["deit_tiny_b1"] = {"error": str(e)}
    return out


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start N ranks (one per GPU) and relay rank 0's line."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but this node exposes {have} HIP device(s)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reps", type=int, default=3, help="minimum repetitions of the K-step timed region (value = median)")
    ap.add_argument("--min-seconds", type=float, default=3.0,
                    help="keep repeating the timed region until this much timed GPU work has accumulated (steady clocks; long enough "
                         "for a 5-second utilisation sampler to see the GPU busy)")
    ap.add_argument("--model", default="deit_small", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=0, help="images per GPU (default: the BASELINE.json config's share)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=2)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("IVIT_STREAMS", "0")),
                    help="batch slices on the runner's internal HIP streams (0 = per-model default)")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("IVIT_GRAPH", "1")), help="replay a captured hipGraph")
    ap.add_argument("--measure-traffic", type=int, default=1,
                    help="N = 1: measure roofline.traffic with two rocprofv3 PMC subprocess passes after the timed region (~25 s); 0 / failure: the committed profiles/pmc_traffic.json")
    ap.add_argument("--box-probe", type=int, default=1, help="0: skip the MFMA / copy probe (profiling runs: keeps its kernels out of the trace)")
    ap.add_argument("--auto-mode", type=int, default=1,
                    help="unless --streams / --graph are given: try (default slices + hipGraph) and (one stream, eager) untimed, time the faster")
    args = ap.parse_args()

    world_env = os.environ.get("WORLD_SIZE")
    if world_env is None and args.gpus > 1:
        self_launch(args)

    import torch
    import torch.distributed as dist
    import ivit_amd as iv
    from ivit_amd import dist as ivdist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback for the product path)")
    if torch.cuda.device_count() < world and os.environ.get("IVIT_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"bench.py: {world} ranks but only {torch.cuda.device_count()} HIP device(s)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # "nccl" is RCCL on ROCm; IVIT_DIST_BACKEND=gloo only for single-GPU plumbing tests
        dist.init_process_group(os.environ.get("IVIT_DIST_BACKEND", "nccl"), rank=rank, world_size=world)

    family, gname, cfg_batch, which = WORKLOADS[args.model]
    batch = args.batch or cfg_batch
    streams = args.streams or {"deit_tiny": 1, "deit_small": 2, "deit_base": 2, "swin_tiny": 2, "vit_base_384": 2}[args.model]   # Swin-T: 2 since round 5 (the role-split Mlp of stage 2 needs two units per CU: 51.0 k vs 49.5 k img/s with 4)
    streams = max(1, min(streams, batch))
    cfg = iv.CONFIGS[args.model] if family == "vit" else iv.SWIN_CONFIGS[args.model]
    g = np.load(os.path.join(ROOT, "tests", "golden", gname))
    scales = {k[len("scale/"):]: np.float32(g[k]) for k in g.files if k.startswith("scale/")}
    weights = None
    if rank == 0:
        weights = (iv.make_vit_weights if family == "vit" else iv.make_swin_weights)(cfg, int(g["seed"]))
    # rank 0 freezes; the packed integer constants travel once over RCCL (xGMI)
    if family == "vit":
        eng = ivdist.build_engine_broadcast(cfg, weights, scales, device, rank, world)
    else:
        eng = ivdist.build_swin_engine_broadcast(cfg, weights, scales, device, rank, world)

    # per-GPU batch is fixed (weak scaling); every rank owns different images.  Rank 0's batch starts with the
    # golden images so the run carries its own parity guard.
    gb = int(g["batch"])
    host_imgs = iv.make_images_int8(cfg, batch, seed=101 + rank)
    if rank == 0:
        host_imgs = np.concatenate([iv.make_images_int8(cfg, gb, int(g["images_seed"])), host_imgs])[:batch]
    imgs = torch.from_numpy(np.ascontiguousarray(host_imgs)).to(device)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def all_ranks_exit_unless(ok_here, msg):
        """ok_here is rank 0's verdict (other ranks pass True): every rank learns it and leaves together."""
        on = device if os.environ.get("IVIT_DIST_BACKEND", "nccl") == "nccl" else "cpu"
        flag = torch.tensor([1 if ok_here else 0], dtype=torch.int32, device=on)
        if world > 1:
            dist.broadcast(flag, src=0)
        if int(flag.item()) == 0:
            if world > 1:
                dist.destroy_process_group()
            raise SystemExit(msg)

    def make_step(nstreams, graph):
        return eng.capture(imgs, nstreams) if graph else (lambda: eng.forward(imgs, nslices=nstreams))

    # Launch mode.  Two ways of issuing the SAME forward: `streams` batch slices on the runner's internal streams replayed as
    # one hipGraph, or the whole batch eagerly on one stream (shorter kernel boundaries, full-batch kernel geometry).  Which one
    # is faster differs by box (+-1.5 %, profiles/README.md round 6), so unless --streams / --graph pin it, both are tried for a
    # few untimed steps and the faster one is the mode of the warm-up and of the timed region; the choice is reported.
    mode_trials = None
    pinned = any(a.split("=")[0] in ("--streams", "--graph") for a in sys.argv[1:]) or "IVIT_STREAMS" in os.environ or "IVIT_GRAPH" in os.environ
    if args.auto_mode and not pinned:        # (batch 1 too: DeiT-T b1 is 0.69 ms eager against 0.83 as a graph replay)
        mode_trials = []
        for ns, gr in ((streams, 1), (1, 0)):
            st = make_step(ns, gr)
            for _ in range(3):
                st()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                st()
            torch.cuda.synchronize()
            mode_trials.append({"streams": ns, "hipgraph": bool(gr), "ms_per_step": round((time.perf_counter() - t0) / 10 * 1e3, 4)})
        best = min(mode_trials, key=lambda m: m["ms_per_step"])
        streams, args.graph = best["streams"], int(best["hipgraph"])
    step = make_step(streams, args.graph)
    # every image of the timed mode (slices on internal streams, hipGraph) against ONE unsliced forward on the current stream,
    # five times, before anything is timed: a race between slices shows up in some image of some run, never reliably in the
    # golden prefix checked below
    ok_all = None
    if rank == 0:
        ref_all = eng.forward(imgs, nslices=1).clone()
        ok_all = all(bool(torch.equal(step(), ref_all)) for _ in range(5))
        del ref_all
    # a wrong image is not a benchmark result: no JSON line, non-zero exit — on EVERY rank (rank 0 alone leaving would
    # strand the others in the next barrier until the RCCL timeout)
    all_ranks_exit_unless(ok_all is not False, "bench.py: the sliced / graph forward differs from the unsliced forward in at least one image; nothing timed")
    for _ in range(args.warmup):
        step()
    rep_dt = []
    while len(rep_dt) < max(1, args.reps) or (sum(rep_dt) < args.min_seconds and len(rep_dt) < 200):
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
        rep_dt.append(dt)
    dt = sorted(rep_dt)[len(rep_dt) // 2]
    ms_per_step = dt / args.steps * 1e3
    value = batch * world * args.steps / dt

    # parity guard inside the bench: the first images of rank 0 are the golden batch
    ok = None
    if rank == 0 and batch >= gb:
        want = g["logits_int"]
        if os.environ.get("IVIT_BENCH_SELFTEST_WRONG_GOLDEN"):      # tests/test_gpu_parity.py only: every rank must leave, non-zero
            want = want + 1
        ok = bool(np.array_equal(step()[:gb].cpu().numpy(), want))
    all_ranks_exit_unless(ok is not False, "bench.py: logits of the golden prefix differ from the reference's; nothing reported")

    box = box_probe(local_rank) if (rank == 0 and args.box_probe) else None
    live = measure_traffic_live(args.model) if (rank == 0 and world == 1 and args.measure_traffic) else None
    # per-operator HIP-event timing (separate instrumented steps, one stream, one C-ABI call per operator)
    per = {}
    if rank == 0 and args.profile_steps > 0:
        eng.forward_ops(imgs)
        with EventTimer(eng.h, torch) as et:
            for _ in range(args.profile_steps):
                eng.forward_ops(imgs)
            per = et.summary()
        # the same operators with both LayerNorms as launches of their own (one more instrumented step): what the fused launches cost
        # WITHOUT the LayerNorm phase inside them, so that the MFMA fraction of the GEMM work itself stays comparable across rounds
        per_unfused = {}
        if family == "vit" and getattr(eng, "fuse_ln_mlp", False):
            eng.fuse_ln_mlp = eng.fuse_ln_qkv = False
            try:
                eng.forward_ops(imgs)
                with EventTimer(eng.h, torch) as et2:
                    eng.forward_ops(imgs)
                    per_unfused = et2.summary()
            finally:
                eng.fuse_ln_mlp = eng.fuse_ln_qkv = True
    if rank == 0:
        ps = max(args.profile_steps, 1)
        lin_ops, bmm_ops = (vit_ops_per_image if family == "vit" else swin_ops_per_image)(cfg)
        # (round 6: the qkv layer of a D = 384 block carries norm1 in its prologue — ivit_layernorm_linear_i8_qkv_planned; the whole
        # launch, LayerNorm included, is inside the time the class's OPs are divided by)
        is_gemm = lambda n: (n.startswith("ivit_linear_i8") or n.startswith("ivit_mlp_fused") or n == "ivit_patch_embed" or
                             n in ("ivit_layernorm_linear_i8_qkv_planned", "ivit_layernorm_mlp_fused_planned"))
        ln_fused = int("ivit_layernorm_linear_i8_qkv_planned" in per) + int("ivit_layernorm_mlp_fused_planned" in per)
        g_ms = sum(v[0] for n, v in per.items() if is_gemm(n)) / ps
        g_n = sum(v[1] for n, v in per.items() if is_gemm(n)) / ps
        achieved = (lin_ops * batch / (g_ms * 1e-3) / 1e12) if g_ms > 0 else None

        def hbm(name, nbytes):
            """achieved HBM GB/s of one HBM-bound operator class: algorithmic bytes / HIP-event time"""
            if name not in per or per[name][0] <= 0:
                return None
            ms = per[name][0] / ps
            gbs = nbytes / (ms * 1e-3) / 1e9
            return {"ms_per_step": round(ms, 4), "launches": per[name][1] // ps, "algorithmic_bytes_per_step": int(nbytes),
                    "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)}
        hbm_ops = {}
        attention = None
        if family == "vit":
            T, D, H, dh, Hd = cfg.num_tokens, cfg.embed_dim, cfg.num_heads, cfg.head_dim, cfg.hidden_dim
            M = batch * T
            # LayerNorm + requant int16 -> int8 3 B/elem; ShiftGELU + requant int8 -> int8 2 B/elem;
            # fused attention reads q, k, v and writes ctx: 4 B per (token, channel)
            # (norm1 of a block whose qkv launch computes it is not a LayerNorm launch any more)
            hbm_ops["layernorm_requant"] = hbm("ivit_layernorm_requant", ((2 - ln_fused) * cfg.depth * M + batch) * D * 3)
            # ShiftGELU is a launch of its own only where the Mlp is not fused (D != 384)
            hbm_ops["shiftgelu_requant"] = hbm("ivit_shiftgelu_requant_lut", cfg.depth * M * Hd * 2)
            # fused attention is bound by NEITHER roofline (VALU / LDS chains per score): both fractions are printed, outside
            # the HBM-bound list
            # (layers differ in the Shiftmax form their scale admits: row tables, two-level tables, arithmetic — one class here)
            att_names = [n for n in per if n.startswith("ivit_attention_fused")]
            if att_names:
                per["attention_fused (all forms)"] = [sum(per[n][0] for n in att_names), sum(per[n][1] for n in att_names)]
            attention = hbm("attention_fused (all forms)", cfg.depth * M * D * 4)
            per.pop("attention_fused (all forms)", None)
            if attention is not None:
                attention["forms"] = {n: per[n][1] // ps for n in att_names}
            if attention is not None:
                a_tops = bmm_ops * batch / (attention["ms_per_step"] * 1e-3) / 1e12
                attention = {"ms_per_step": attention["ms_per_step"], "launches": attention["launches"], "forms": attention["forms"],
                             "hbm": {k: attention[k] for k in ("algorithmic_bytes_per_step", "achieved", "peak", "unit", "frac")},
                             "mfma": {"algorithmic_ops_per_step": bmm_ops * batch, "achieved": round(a_tops, 1), "peak": INT8_PEAK_TOPS,
                                      "unit": "TOP/s", "frac": round(a_tops / INT8_PEAK_TOPS, 4)},
                             "bound": "neither (VALU issue + LDS gathers of the Shiftmax between the two MFMA phases)"}
        else:
            # Swin: per stage L tokens of C channels; LayerNorm twice per block (+ PatchMerging's over 4C), windowed
            # attention reads q, k, v and writes ctx, ShiftGELU only in the stages whose Mlp is not fused (C != 96, 384)
            ln_b = att_b = gelu_b = tok_b = 0
            res, C = cfg.grid, cfg.embed_dim
            for li, depth in enumerate(cfg.depths):
                Mi = batch * res * res
                (ln_b, tok_b) = (ln_b, tok_b + depth * 2 * Mi * C * 3) if li == 0 else (ln_b + depth * 2 * Mi * C * 3, tok_b)
                att_b += depth * Mi * C * 4
                if C not in (96, 384):
                    gelu_b += depth * Mi * cfg.mlp_ratio * C * 2
                if li < len(cfg.depths) - 1:
                    ln_b += (Mi // 4) * 4 * C * 3
                    res, C = res // 2, C * 2
            ln_b += batch * res * res * C * 3
            hbm_ops["layernorm_requant"] = hbm("ivit_layernorm_requant", ln_b)
            hbm_ops["layernorm_tokenorder_requant"] = hbm("ivit_layernorm_tokenorder_requant", tok_b)
            hbm_ops["shiftgelu_requant"] = hbm("ivit_shiftgelu_requant_lut", gelu_b)
            hbm_ops["window_attention_fused"] = hbm("ivit_window_attention_fused", att_b)
        hbm_ops = {k: v for k, v in hbm_ops.items() if v is not None}
        # the single kernel with the most GPU time among the GEMM class, from the same HIP-event timings: its own OPs / its own
        # average launch time (the class average above hides a slow kernel behind a fast one)
        dominant = None
        if family == "vit" and per:
            T, D, Hd = cfg.num_tokens, cfg.embed_dim, cfg.hidden_dim
            M = batch * T
            fused = "ivit_mlp_fused_planned" in per
            ops_of = {"ivit_mlp_fused_planned": ("fc1 + ShiftGELU + fc2 + residual in one launch (mlp384rs_kernel / mlp384_kernel)", 4.0 * M * D * Hd),
                      "ivit_layernorm_mlp_fused_planned": ("norm2 + fc1 + ShiftGELU + fc2 + residual in one launch (mlp384rs_kernel<.., LNH>: the LayerNorm's "
                                                           "time is inside, its operations are not counted)", 4.0 * M * D * Hd),
                      "ivit_linear_i8_qkv_planned": ("qkv QuantLinear (gemm_as_kernel<5> / gemm_ws_qkv_kernel)", 6.0 * M * D * D),
                      "ivit_layernorm_linear_i8_qkv_planned": ("norm1 + qkv QuantLinear in one launch (gemm_ws_qkv_kernel<.., LN>: the LayerNorm's "
                                                               "time is inside, its operations are not counted)", 6.0 * M * D * D),
                      "ivit_linear_i8_requant_planned": ("fc1 QuantLinear, 8-bit epilogue", 2.0 * M * D * Hd),
                      "ivit_linear_i8_requant_residual_planned": (("proj" if fused else "proj and fc2 (average)") + " QuantLinear + residual QuantAct",
                                                                  2.0 * M * D * D if fused else (M * D * D + M * D * Hd))}
            cand = [(v[0], n) for n, v in per.items() if n in ops_of]
            if cand:
                _, n = max(cand)
                us = per[n][0] / per[n][1] * 1e3
                tops = ops_of[n][1] / (us * 1e-6) / 1e12
                dominant = {"name": n, "what": ops_of[n][0], "ops_per_launch": int(ops_of[n][1]), "us_per_launch": round(us, 2),
                            "launches_per_step": per[n][1] // ps, "share_of_instrumented_step": round(per[n][0] / sum(v[0] for v in per.values()), 4),
                            "achieved": round(tops, 1), "peak": INT8_PEAK_TOPS, "unit": "TOP/s", "frac": round(tops / INT8_PEAK_TOPS, 4)}
                # the same launch without the LayerNorm in its head (norm1 / norm2 as launches of their own, one extra instrumented step)
                plain = {"ivit_layernorm_mlp_fused_planned": "ivit_mlp_fused_planned", "ivit_layernorm_linear_i8_qkv_planned": "ivit_linear_i8_qkv_planned"}.get(n)
                if plain and plain in per_unfused and per_unfused[plain][1]:
                    us0 = per_unfused[plain][0] / per_unfused[plain][1] * 1e3
                    tops0 = ops_of[n][1] / (us0 * 1e-6) / 1e12
                    dominant["without_layernorm_head"] = {"name": plain, "us_per_launch": round(us0, 2), "achieved": round(tops0, 1),
                                                          "frac": round(tops0 / INT8_PEAK_TOPS, 4),
                                                          "what": "the same GEMM work with the LayerNorm as a launch of its own (one extra instrumented step, not the timed path)"}
        roofline = {
            "kernel": "QuantLinear GEMM class: gemm_ws_qkv_kernel (D = 384: norm1 + qkv in one launch, proj + residual), gemm_as_kernel / gemm_ps_kernel / "
                      "gemm_glds_kernel (patch-embed, qkv, proj, head; fused requant epilogues) and mlp384rs_kernel / mlp384_kernel (fc1 + ShiftGELU + fc2 + "
                      "residual QuantAct in one launch where D = 384, norm2 in its head); the ShiftGELU table pass, both fused LayerNorms and the patch gather are "
                      "inside the time the OPs are divided by",
            "bound": "mfma", "achieved": None if achieved is None else round(achieved, 1),
            "peak": INT8_PEAK_TOPS, "unit": "TOP/s",
            "frac": None if achieved is None else round(achieved / INT8_PEAK_TOPS, 4),
            "traffic": (round(live["avg_bytes_per_launch"]) if live else (pmc_traffic() if args.model == "deit_small" else None)),
            "traffic_per_kernel": (pmc_traffic_per_kernel(batch, cfg.num_tokens, cfg.embed_dim, cfg.hidden_dim, live["per_kernel"]) if live and family == "vit"
                                   else (live["per_kernel"] if live
                                         else (pmc_traffic_per_kernel(batch, cfg.num_tokens, cfg.embed_dim, cfg.hidden_dim) if args.model == "deit_small" else None))),
            "traffic_source": ("measured by THIS run: two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE x2-corrected, WRITE_SIZE) of a 3-forward "
                               "single-stream eager run of this script, as subprocesses after the timed region; mean per GEMM-class launch") if live
                              else (("profiles/pmc_traffic.json — a COMMITTED rocprofv3 PMC run (tools/prof.sh: bench.py --streams 1 --graph 0, "
                                     "separate FETCH_SIZE / WRITE_SIZE passes, FETCH x2-corrected); the live measurement was off or failed")
                                    if args.model == "deit_small" else None),
            "dominant_kernel": dominant,
            "launches_per_step": g_n, "ms_per_step_in_kernel": round(g_ms, 4),
            "avg_launch_ms": round(g_ms / g_n, 5) if g_n else None,
            "algorithmic_ops_per_step": lin_ops * batch,
            # every int8 MAC of the forward (QuantLinear + attention matmuls) over the TIMED step — the product path, whatever it fuses:
            # the one fraction that stays comparable when a LayerNorm moves into a GEMM launch (the class fraction above then drops,
            # because that launch's time now includes the LayerNorm)
            "whole_model": {"algorithmic_ops_per_step": (lin_ops + bmm_ops) * batch,
                            "achieved": round((lin_ops + bmm_ops) * batch / (ms_per_step * 1e-3) / 1e12, 1), "peak": INT8_PEAK_TOPS,
                            "unit": "TOP/s per GPU", "frac": round((lin_ops + bmm_ops) * batch / (ms_per_step * 1e-3) / 1e12 / INT8_PEAK_TOPS, 4)},
            # register-only MFMA loops on random int8 operands, measured by THIS run on THIS box (`box` below): the chip clocks
            # down to ~1.7-2.0 GHz under int8 MFMA load, so the nominal 5033 is not reachable by any kernel
            "mfma_ubench_ceiling_tops_random_operands": (box or {}).get("mfma_tops_random"),
            "timed_on": "single stream, one C-ABI call per operator (engine.forward_ops), HIP events on the launch stream — "
                        "not the sliced / hipGraph path that produced `value`",
            "hbm_bound_operators": hbm_ops,
            "attention_fused": attention,
        }
        breakdown = {n: {"ms_per_step": round(v[0] / ps, 4), "launches": v[1] // ps}
                     for n, v in sorted(per.items(), key=lambda kv: -kv[1][0])}
        out = {
            "metric": "images/sec + int8-MFMA-roofline% for DeiT-S bs256@224 on 1/2/4/8 MI355X",
            "value": round(value, 1), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "int8", "data": "synthetic",
            "config": {"workload": f"{cfg.name} int8 forward, batch {batch}/GPU, {cfg.img_size}x{cfg.img_size}x3 synthetic int8 "
                                   f"(BASELINE.json {which}); weights seeded synthetic, activation scales from the reference calibration",
                       "global_batch": batch * world, "parallelism": f"dp{world} (batch-sharded, weights RCCL-broadcast once)",
                       "streams_per_gpu": streams, "hipgraph": bool(args.graph),
                       "launch_mode_trials": mode_trials},
            "repetitions_ms_per_step": [round(d / args.steps * 1e3, 4) for d in rep_dt],
            "value_is": "median repetition",
            "model_int8_tops": round((lin_ops + bmm_ops) * batch * world / (ms_per_step * 1e-3) / 1e12, 1),
            "model_roofline_frac": round((lin_ops + bmm_ops) * batch / (ms_per_step * 1e-3) / 1e12 / INT8_PEAK_TOPS, 4),
            "bit_exact_vs_reference_golden": ok,
            "all_images_equal_unsliced_forward": ok_all,
            "roofline": roofline,
            "box": box,
            "kernel_breakdown_ms": breakdown,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(family, cfg, weights, scales, os.path.join(ROOT, "tests", "golden"))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
