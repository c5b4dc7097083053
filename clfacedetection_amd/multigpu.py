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
tests): ONE all-gather of fixed-capacity `[count | rects x cap]` blocks per step (RectGather).
Payloads are a few KB, so this is latency-, not bandwidth-bound.
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


class RectGather:
    """The one collective of the N > 1 path: every rank contributes a fixed-capacity block
    `[count | rects x cap]` (int32 rows of RECT_FIELDS; row 0 carries the count) and ONE all-gather moves all of them.  The
    capacity persists across steps; when any rank's count exceeds it every rank sees that in the gathered headers, all of
    them double the capacity the same way and repeat the step's collective — so a steady workload costs one collective and
    one device-to-host copy per step (`n_collectives`, `last_ms`).  `device`: where the tensors live (the GPU for backend
    "nccl" = RCCL, the CPU for gloo)."""

    def __init__(self, device=None, group=None, cap: int = 1024):
        import torch
        self.device = device if device is not None else torch.device("cpu")
        self.group = group
        self.cap = max(1, int(cap))
        self.n_collectives = 0
        self.last_ms = 0.0
        self.total_ms = 0.0
        self._send = self._recv = None

    def _buffers(self, world):
        import torch
        shape = (self.cap + 1, len(RECT_FIELDS))
        if self._send is None or tuple(self._send.shape) != shape:
            self._send = torch.zeros(shape, dtype=torch.int32, device=self.device)
            self._recv = torch.zeros((world * shape[0], shape[1]), dtype=torch.int32, device=self.device)   # (ranks concatenated along dim 0: the form gloo accepts too)
        return self._send, self._recv

    def __call__(self, rects: np.ndarray) -> np.ndarray:
        import time
        import torch
        import torch.distributed as dist
        t0 = time.perf_counter()
        world = dist.get_world_size(self.group)
        if len(rects) and not np.array_equal(rects["weight"], np.rint(rects["weight"])):
            raise ValueError("rectangle weights are neighbour counts: integers")
        mine = np.stack([rects[f].astype(np.int32) for f in RECT_FIELDS], axis=1) if len(rects) else \
            np.zeros((0, len(RECT_FIELDS)), np.int32)
        while True:
            send, recv = self._buffers(world)
            block = np.zeros((self.cap + 1, len(RECT_FIELDS)), np.int32)
            block[0, 0] = len(mine)
            k = min(len(mine), self.cap)
            block[1:1 + k] = mine[:k]
            send.copy_(torch.from_numpy(block))          # (one host-to-device copy; CPU tensors: a memcpy)
            dist.all_gather_into_tensor(recv, send, group=self.group)
            self.n_collectives += 1
            got = recv.cpu().numpy().reshape(world, self.cap + 1, len(RECT_FIELDS))   # (one device-to-host copy = the step's only synchronisation)
            counts = got[:, 0, 0].astype(np.int64)
            if int(counts.max(initial=0)) <= self.cap:
                break
            while self.cap < int(counts.max()):           # every rank sees the same counts: the same new capacity everywhere
                self.cap *= 2
            self._send = None
        parts = [got[r, 1:1 + int(counts[r])] for r in range(world)]
        allr = np.concatenate(parts) if parts else mine
        out = np.zeros(len(allr), rects.dtype)
        for i, f in enumerate(RECT_FIELDS):
            out[f] = allr[:, i]
        out = out[np.lexsort((out["x"], out["y"], out["scale_idx"], out["frame"]))]
        self.last_ms = (time.perf_counter() - t0) * 1e3
        self.total_ms += self.last_ms
        return out


_GATHERERS = {}


def allgather_rects(rects: np.ndarray, device=None, group=None) -> np.ndarray:
    """All-gather variable-length detection lists; every rank returns the same array, sorted by (frame, scale_idx, y, x).
    `rects` carries GLOBAL frame indices.  One collective per call (RectGather; the gatherer and its capacity are kept per
    (device, group))."""
    key = (str(device), id(group))
    g = _GATHERERS.get(key)
    if g is None:
        g = _GATHERERS[key] = RectGather(device, group)
    return g(rects)
