"""Multi-GPU: one process per GPU, images sharded by rank, NO per-step collective.
The only collective is one broadcast of the packed integer constants (int8 weights,
int32 biases, dyadic tables, int16 position embedding) from rank 0 — RCCL over xGMI
on GPUs (`nccl` backend), gloo on CPU for the tests.
"""
import numpy as np
import torch
import torch.distributed as dist

from .engine import ViTEngine, pack_constants
from .freeze import freeze_vit


def broadcast_constants(consts, f32, rank, world, device):
    """rank 0 passes (consts, f32); other ranks pass (None, None).
    Returns (blob tensor on `device`, table, f32) on every rank."""
    if world == 1:
        blob, table = pack_constants(consts)
        return torch.from_numpy(blob).to(device), table, f32
    meta = [None]
    blob_t = None
    if rank == 0:
        blob, table = pack_constants(consts)
        meta = [(table, {k: float(np.float32(v)) for k, v in f32.items()}, int(blob.size))]
        blob_t = torch.from_numpy(blob).to(device)
    dist.broadcast_object_list(meta, src=0)
    table, f32, nbytes = meta[0]
    if rank != 0:
        blob_t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(blob_t, src=0)   # RCCL: one large message, ring over xGMI links
    return blob_t, table, f32


def shard_range(total, rank, world):
    """contiguous, balanced split of `total` images over `world` ranks."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def build_engine_broadcast(cfg, weights, scales, device, rank, world):
    consts = f32 = None
    if rank == 0:
        consts, f32 = freeze_vit(cfg, weights, scales)
    blob, table, f32 = broadcast_constants(consts, f32, rank, world, device)
    return ViTEngine(cfg, None, f32, device=device, blob=blob, table=table)


def broadcast_packed(packed, rank, world, device):
    """(blob, table, host) of any engine from rank 0 to every rank: the table / host scalars as one small
    object, the byte blob as ONE tensor broadcast (RCCL ring over xGMI on GPUs)."""
    if world == 1:
        blob, table, host = packed
        return torch.from_numpy(blob).to(device), table, host
    meta, blob_t = [None], None
    if rank == 0:
        blob, table, host = packed
        meta = [(table, host, int(blob.size))]
        blob_t = torch.from_numpy(blob).to(device)
    dist.broadcast_object_list(meta, src=0)
    table, host, nbytes = meta[0]
    if rank != 0:
        blob_t = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dist.broadcast(blob_t, src=0)
    return blob_t, table, host


def build_swin_engine_broadcast(cfg, weights, scales, device, rank, world):
    from .swin_engine import SwinEngine, freeze_swin, pack_swin_constants
    packed = pack_swin_constants(freeze_swin(cfg, weights, scales)) if rank == 0 else None
    return SwinEngine(cfg, None, None, device=device, packed=broadcast_packed(packed, rank, world, device))
