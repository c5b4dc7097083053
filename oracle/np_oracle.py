"""numpy twin of oracle/vj_oracle.c (TEST INFRASTRUCTURE ONLY).

A second, vectorised restatement of rows a1-a7 and a9 of SURVEY.md §8 — written
independently of the C file (whole-array float32 arithmetic, windows as vectors, masks for
the tree walks) so that the two can check each other, which matters most for the tree
cascades whose parity no reference artefact pins.  Same citations as vj_oracle.c.
Pure numpy: use it at small sizes only.
"""
from __future__ import annotations

import numpy as np

F = np.float32


def integral(gray: np.ndarray):
    """cvIntegral layout (clif.cpp:280-285): (h+1, w+1), zero first row/column; sum wraps mod 2^32."""
    g = gray.astype(np.uint64)
    h, w = g.shape
    s = np.zeros((h + 1, w + 1), np.uint64)
    q = np.zeros((h + 1, w + 1), np.uint64)
    s[1:, 1:] = np.cumsum(np.cumsum(g, 0), 1)
    q[1:, 1:] = np.cumsum(np.cumsum(g * g, 0), 1)
    return (s & np.uint64(0xFFFFFFFF)).astype(np.uint32), q


def _round_half_away(v):
    v = np.asarray(v, np.float64)
    return np.where(v >= 0, np.floor(v + 0.5), np.ceil(v - 0.5))


def scales(c, W, H, min_size=(0, 0), max_size=(0, 0), scale_factor=1.1):
    """clod.cpp:1198-1204 + setupScale (clod.cpp:371-415)."""
    sf = F(scale_factor)
    out = []
    s = F(1)
    k = 0
    while F(s * F(c.win_w)) < F(W - 10) and F(s * F(c.win_h)) < F(H - 10) and k < 4096:
        d = {"scale_idx": k, "scale": s, "accepted": False}
        d["step"] = F(max(2.0, float(s)))
        sw = int(_round_half_away(F(F(c.win_w) * s)))
        sh = int(_round_half_away(F(F(c.win_h) * s)))
        ok = not (sw < min_size[0] or sh < min_size[1])
        ok = ok and not (max_size[0] != 0 and sw > max_size[0]) and not (max_size[1] != 0 and sh > max_size[1])
        ok = ok and not (sw > W or sh > H)
        if ok:
            ex = int(_round_half_away(s))
            ew = int(_round_half_away(F(F(c.win_w - 2) * s)))
            eh = int(_round_half_away(F(F(c.win_h - 2) * s)))
            nx = int(np.rint(np.float64(F(F(W - sw) / d["step"]))))   # lrint: half-even
            ny = int(np.rint(np.float64(F(F(H - sh) / d["step"]))))
            d.update(accepted=True, win_w=sw, win_h=sh, equ_x=ex, equ_y=ex, equ_w=ew, equ_h=eh, area=ew * eh,
                     nx=max(nx, 0), ny=max(ny, 0))
        out.append(d)
        s = F(s * sf)
        k += 1
    return out


def detect(c, gray: np.ndarray, min_size=(0, 0), max_size=(0, 0), scale_factor=1.1, signed_mean=False):
    """Raw detections [(scale_idx, x, y, w, h)] sorted by (scale_idx, y, x) and per-stage entered counts."""
    H, W = gray.shape
    ii, qq = integral(gray)
    stride = W + 1
    pad = 3 * stride
    iif = np.concatenate([ii.reshape(-1), np.zeros(pad, np.uint32)])       # +slack rows, like the C oracle
    qqf = np.concatenate([qq.reshape(-1), np.zeros(pad, np.uint64)])
    n_stages = c.n_stages
    entered = [0] * n_stages
    dets = []
    rects = c.node_rect.reshape(-1, 3, 4)
    wts = c.node_weight.reshape(-1, 3)
    for sc in scales(c, W, H, min_size, max_size, scale_factor):
        if not sc["accepted"] or sc["nx"] <= 0 or sc["ny"] <= 0:
            continue
        s, step, area = sc["scale"], sc["step"], F(sc["area"])
        # precomputeWindows (clod.cpp:495-527)
        xs = np.rint((np.arange(sc["nx"], dtype=F) * step).astype(np.float64)).astype(np.int64)
        ys = np.rint((np.arange(sc["ny"], dtype=F) * step).astype(np.float64)).astype(np.int64)
        X, Y = np.meshgrid(xs, ys)
        X, Y = X.reshape(-1), Y.reshape(-1)
        off = Y * stride + X
        # computeVariance (clod.cpp:418-446)
        a = off + sc["equ_y"] * stride + sc["equ_x"]
        b, cc, d = a + sc["equ_w"], a + sc["equ_h"] * stride, a + sc["equ_h"] * stride + sc["equ_w"]
        S = (iif[a] - iif[b] - iif[cc] + iif[d]).astype(np.uint32)
        Q = (qqf[a] - qqf[b] - qqf[cc] + qqf[d]).astype(np.uint64)
        mean = (S.astype(np.int32).astype(F) if signed_mean else S.astype(F)) / area
        var = Q.astype(F) / area - mean * mean
        with np.errstate(invalid="ignore"):
            var = np.where(var >= 0, np.sqrt(np.maximum(var, F(0))), F(1)).astype(F)

        # precomputeKernelCascade (clod.cpp:529-578) per node
        def node_tables():
            r = _round_half_away((rects.astype(F) * s)).astype(np.int64)          # x y w h, scaled
            present = wts != 0
            w_scaled = (wts / area).astype(F)
            rw, rh = r[..., 2].astype(F), r[..., 3].astype(F)
            sum_area = np.zeros(len(wts), F)
            for q_ in (1, 2):
                term = ((w_scaled[:, q_] * rw[:, q_]).astype(F) * rh[:, q_]).astype(F)
                sum_area = np.where(present[:, q_], (sum_area + term).astype(F), sum_area)
            first = (r[:, 0, 2] * r[:, 0, 3]).astype(F)
            w_scaled[:, 0] = (-sum_area / first).astype(F)
            lt = r[..., 1] * stride + r[..., 0]
            return lt, r[..., 2], r[..., 3] * stride, w_scaled, present
        lt, dw, dh, wsc, present = node_tables()

        def node_sum(n, idx):
            o = off[idx]
            tot = None
            for q_ in range(3):
                if q_ == 2 and not present[n, 2]:
                    break
                p0 = o + lt[n, q_]
                v = (iif[p0] - iif[p0 + dw[n, q_]] - iif[p0 + dh[n, q_]] + iif[p0 + dh[n, q_] + dw[n, q_]]).astype(np.uint32)
                t = (v.astype(F) * wsc[n, q_]).astype(F)
                tot = t if tot is None else (tot + t).astype(F)
            return tot

        def stage_sum(stage, idx):
            ssum = np.zeros(len(idx), F)
            t0 = int(c.stage_first_tree[stage])
            for t in range(t0, t0 + int(c.stage_n_trees[stage])):
                n0, nn, a0 = int(c.tree_first_node[t]), int(c.tree_n_nodes[t]), int(c.tree_first_alpha[t])
                if nn == 1:   # clod.cl:81: alpha[rect_sum >= norm_threshold]
                    rs = node_sum(n0, idx)
                    thr = (c.node_threshold[n0] * var[idx]).astype(F)
                    ssum = (ssum + np.where(rs >= thr, c.alpha[a0 + 1], c.alpha[a0]).astype(F)).astype(F)
                else:         # tempcv.cpp:771-792
                    cur = np.zeros(len(idx), np.int64)
                    val = np.zeros(len(idx), F)
                    done = np.zeros(len(idx), bool)
                    for k in range(nn):
                        m = (~done) & (cur == k)
                        if not m.any():
                            continue
                        sub = idx[m]
                        rs = node_sum(n0 + k, sub)
                        thr = (c.node_threshold[n0 + k] * var[sub]).astype(F)
                        nxt = np.where(rs < thr, int(c.node_left[n0 + k]), int(c.node_right[n0 + k]))
                        leaf = nxt <= 0
                        v = val[m]
                        v[leaf] = c.alpha[a0 - nxt[leaf]]
                        val[m] = v
                        dn = done[m]
                        dn[leaf] = True
                        done[m] = dn
                        cu = cur[m]
                        cu[~leaf] = nxt[~leaf]
                        cur[m] = cu
                    ssum = (ssum + val).astype(F)
            return ssum

        # stage walk (tempcv.cpp:834-861) on index sets: target stage per window
        target = np.zeros(len(off), np.int64)
        alive = np.ones(len(off), bool)
        accepted = np.zeros(len(off), bool)
        order = _topo_order(c)
        for stg in order:
            idx = np.nonzero(alive & (target == stg))[0]
            if len(idx) == 0:
                continue
            entered[stg] += len(idx)
            passed = stage_sum(stg, idx) >= c.stage_threshold[stg]
            on_pass = int(c.stage_child[stg])
            ptr = stg
            while ptr != -1 and c.stage_next[ptr] == -1:
                ptr = int(c.stage_parent[ptr])
            on_fail = -2 if ptr == -1 else int(c.stage_next[ptr])
            p_idx, f_idx = idx[passed], idx[~passed]
            if on_pass == -1:
                accepted[p_idx] = True
                alive[p_idx] = False
            else:
                target[p_idx] = on_pass
            if on_fail == -2:
                alive[f_idx] = False
            else:
                target[f_idx] = on_fail
        for i in np.nonzero(accepted)[0]:
            dets.append((sc["scale_idx"], int(X[i]), int(Y[i]), sc["win_w"], sc["win_h"]))
    dets.sort(key=lambda d: (d[0], d[2], d[1]))
    return dets, entered


def _topo_order(c):
    n = c.n_stages
    on_pass = [int(v) for v in c.stage_child]
    on_fail = []
    for s in range(n):
        ptr = s
        while ptr != -1 and c.stage_next[ptr] == -1:
            ptr = int(c.stage_parent[ptr])
        on_fail.append(-2 if ptr == -1 else int(c.stage_next[ptr]))
    seen, post = set(), []

    def visit(s):
        if s < 0 or s in seen:
            return
        seen.add(s)
        visit(on_pass[s])
        visit(on_fail[s])
        post.append(s)
    import sys
    sys.setrecursionlimit(10000)
    visit(0)
    return post[::-1]
