"""Sharding of the detect path over the GPUs of one node (one process per GPU).

The path shards without any data-path exchange: every (frame, scale) pair is
independent given that frame's integral images.  So
  * a batch with at least as many frames as ranks is split by whole frames
    (contiguous blocks, so each rank integrates only its own frames);
  * fewer frames than ranks (e.g. one 4096x4096 frame): every rank takes all frames
    but only a subset of the scales, balanced by estimated cost (window count x a
    per-window weight for the LDS-tile and the global-gather scales, each kind dealt
    longest-first so that every rank keeps both of its chains busy);
and the only collective is the final all-gather of the detection rectangles
(torch.distributed: backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU
tests).  Payloads are a few KB, so this is latency-, not bandwidth-bound.
"""
from __future__ import annotations

import numpy as np

# every field of a result row travels: `weight` is the neighbour count of a grouped rectangle (min_neighbors != 0), an
# integer-valued float (0 for raw candidates), so the gather can stay one int32 tensor
RECT_FIELDS = ("x", "y", "w", "h", "frame", "scale_idx", "weight")


def shard_frames(n_frames: int, rank: int, world: int) -> range:
    """Contiguous block of frame indices owned by `rank`."""
    base, extra = divmod(n_frames, world)
    lo = rank * base + min(rank, extra)
    return range(lo, lo + base + (1 if rank < extra else 0))


def shard_scales(window_counts: list[int], rank: int, world: int, window_sides: list[int] | None = None,
                 base_side: int = 0) -> list[int]:
    """vj_shard_scales' rule (csrc/vj_plan.cpp), integer arithmetic throughout so that every host computes the same shares:
    a scale costs windows x a per-window weight — scales whose window is at most 72 px run on LDS tiles (weight 16/16 at s = 1,
    rising with the staged footprint), larger ones are global gathers (about six times as much per window) — and the two
    kinds, which run as two overlapping chains on a device, are dealt separately, longest first: tiles to the rank with the
    least tile cost, gathers to the rank with the least (gather cost + half its tile cost).
    Ties go to the lower rank.  Without `window_sides` (the scaled window's larger side per scale; `base_side` = the
    cascade's own) every scale weighs the same: plain LPT on the window counts."""
    n = len(window_counts)
    gather = [0] * n
    cost = [16 * int(w) for w in window_counts]
    if window_sides is not None:
        for k in range(n):
            side = int(window_sides[k])
            q = max(16, side * 16 // max(int(base_side), 1))
            gather[k] = 1 if side > 72 else 0
            weight = 96 + 13 * (max(q, 61) - 61) // 16 if gather[k] else 16 + 10 * (q - 16) // 16
            cost[k] = int(window_counts[k]) * weight
    order = sorted(range(n), key=lambda k: (-cost[k], k))
    tile, gath = [0] * world, [0] * world
    est = lambda t, g: g + t // 2
    mine = []
    for cls in (0, 1):
        for k in order:
            if gather[k] != cls:
                continue
            if cls == 0:
                r = min(range(world), key=lambda i: (tile[i], i))
                tile[r] += cost[k]
            else:
                r = min(range(world), key=lambda i: (est(tile[i], gath[i] + cost[k]), i))
                gath[r] += cost[k]
            if r == rank:
                mine.append(k)
    return sorted(mine)


def plan(n_frames: int, window_counts: list[int], rank: int, world: int, window_sides: list[int] | None = None, base_side: int = 0):
    """-> (frame indices, scale indices or None for all scales) for this rank."""
    if world == 1 or n_frames >= world:
        return list(shard_frames(n_frames, rank, world)), None
    return list(range(n_frames)), shard_scales(window_counts, rank, world, window_sides, base_side)


def allgather_rects(rects: np.ndarray, device=None, group=None) -> np.ndarray:
    """All-gather variable-length detection lists; every rank returns the same array,
    sorted by (frame, scale_idx, y, x).  `rects` carries GLOBAL frame indices.

    Two collectives: counts, then one padded int32 [max_count, 7] tensor per rank."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    if len(rects) and not np.array_equal(rects["weight"], np.rint(rects["weight"])):
        raise ValueError("rectangle weights are neighbour counts: integers")
    mine = np.stack([rects[f].astype(np.int32) for f in RECT_FIELDS], axis=1) if len(rects) else \
        np.zeros((0, len(RECT_FIELDS)), np.int32)
    n = torch.tensor([len(mine)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    cap = max(max(counts), 1)
    pad = torch.zeros((cap, len(RECT_FIELDS)), dtype=torch.int32, device=dev)
    if len(mine):
        pad[:len(mine)] = torch.from_numpy(mine).to(dev)
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    parts = [b[:c].cpu().numpy() for b, c in zip(bufs, counts)]
    allr = np.concatenate(parts) if parts else mine
    out = np.zeros(len(allr), rects.dtype)
    for i, f in enumerate(RECT_FIELDS):
        out[f] = allr[:, i]
    return out[np.lexsort((out["x"], out["y"], out["scale_idx"], out["frame"]))]
