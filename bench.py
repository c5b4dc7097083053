#!/usr/bin/env python3
"""Headline benchmark of the MI355X-native Viola–Jones detect path.

Metric (BASELINE.json): candidate windows/sec (+ Mpix/s) on 1080p frames with
haarcascade_frontalface_alt, 1/2/4/8 GPUs.  A "step" is one pass of the whole hot path
(integral + squared-integral kernels, all cascade passes, detection read-back and — for
N > 1 — the all-gather of detection rectangles) over one batch of synthetic frames that
is already resident in HBM.  Per-GPU work is fixed (weak scaling): every rank owns a
batch of --frames 1080p frames; there is no data-path collective besides that gather.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` is whole-job windows/s over the K timed steps
(max over ranks of the wall time, barrier + synchronize on both sides).  `roofline`
is for the dominant kernel, the first cascade pass, from HIP events recorded inside the
library on the stream the kernels run on.  `cpu_baseline` is the CPU oracle
(oracle/vj_oracle.c, a single-threaded restatement of the reference's clod path) timed
on a bounded sample of the same frames, and doubles as a parity check of that sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64, help="1080p frames per GPU per step")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cascade", default="frontalface_alt")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames of the batch timed on the CPU oracle (0 = skip)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the N > 1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--pass-split", default=None)
    ap.add_argument("--blocks-per-cu", type=int, default=None)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run",
                  file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        return 3
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    coll_dev = dev if args.backend == "nccl" else torch.device("cpu")   # where the collectives' tensors live

    from clfacedetection_amd import (VJ_FLAG_COUNTERS, Cascade, DeviceFrames, Environment, default_params, multigpu,
                                     synth)

    env = Environment(local_rank)
    if args.pass_split is not None:
        env.configure("pass_split", args.pass_split)
    if args.blocks_per_cu is not None:
        env.configure("blocks_per_cu", args.blocks_per_cu)
    casc = Cascade.load(args.cascade)
    H, W, B = args.height, args.width, args.frames

    # synthetic frames (seeds 1.., kinds cycling noise / smooth / blocks), per rank
    frames_h = synth.batch(B, H, W, seed0=1 + rank * B)
    frames_d = torch.from_numpy(frames_h).to(dev)
    torch.cuda.synchronize()
    dframes = DeviceFrames.from_torch(frames_d)
    windows_per_frame = casc.count_windows(W, H)

    def step(params):
        r = env.detect(casc, dframes, params)
        rects = r.rects
        if world > 1:
            rects = rects.copy()
            rects["frame"] += rank * B          # global frame index
            rects = multigpu.allgather_rects(rects, device=coll_dev)
        return r, rects

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # one counted run: algorithmic bytes + the detections used for the parity sample
    counted, _ = step(default_params(flags=VJ_FLAG_COUNTERS))
    p = default_params()
    for _ in range(args.warmup):
        step(p)
    barrier()
    t0 = time.perf_counter()
    integral_ms = cascade_ms = 0.0
    pass_ms = None
    launch_ms = None
    launches = None
    n_det_total = 0
    for _ in range(args.steps):
        r, rects = step(p)
        integral_ms += r.integral_ms
        cascade_ms += r.cascade_ms
        pm = [x[2] for x in r.passes]
        pass_ms = pm if pass_ms is None else [a + b for a, b in zip(pass_ms, pm)]
        lm = [l["ms"] for l in r.launches]
        launch_ms = lm if launch_ms is None else [a + b for a, b in zip(launch_ms, lm)]
        launches = r.launches
        n_det_total = len(rects)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    total_windows = windows_per_frame * B * world * args.steps
    value = total_windows / elapsed
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)

    out = None
    if rank == 0:
        K = max(args.steps, 1)
        # ---- roofline of the dominant kernel.  Launches are grouped by kernel (the names
        # rocprofv3 --kernel-trace --stats reports); the group with the most time is the
        # dominant kernel.  Algorithmic bytes: 48 B per window for the
        # variance gathers + 16 B per evaluated rectangle, from counted runs restricted to
        # the scales each launch covers; time = HIP events recorded inside the library on
        # the stream the kernels run on, averaged over the timed steps.
        nodes, trees, stages = casc.nodes, casc.trees, casc.stages
        rects_per_stage = []
        for st in stages:
            tr = trees[st["first_tree"]:st["first_tree"] + st["n_trees"]]
            rects_per_stage.append(int(sum(int(nodes["n_rects"][t["first_node"]:t["first_node"] + t["n_nodes"]].sum())
                                           for t in tr)))
        kname = {"tile": "vj::cascade_tile_pass<false, false, true>", "block": "vj::cascade_tile_pass<false, false, false>",
                 "grid": "vj::cascade_pass<true, false, *, false, false>",
                 "queue": "vj::cascade_pass<false, false, *, false, false>"}
        groups = {}
        for l, ms in zip(launches, launch_ms):
            g = groups.setdefault(l["kind"], {"ms": 0.0, "n": 0, "launches": []})
            g["ms"] += ms / K
            g["n"] += 1
            g["launches"].append(l)
        dom_kind = max(groups, key=lambda k: groups[k]["ms"])
        dom = groups[dom_kind]
        # algorithmic bytes of the dominant kernel's launches from the counted run's PER-LAUNCH counters
        alg = 0
        for l in counted.launches:
            if l["kind"] != dom_kind:
                continue
            if l["kind"] != "queue":
                alg += 48 * l["stage_entered"][0]
            alg += 16 * sum(n * rects_per_stage[st] for st, n in enumerate(l["stage_entered"]))
        # the same figure for every kernel group (the chains overlap, so each kernel's own time is the time it was
        # resident, not a share of the step)
        per_kernel = {}
        for kind, grp in groups.items():
            b = 0
            for l in counted.launches:
                if l["kind"] != kind:
                    continue
                if kind != "queue":
                    b += 48 * l["stage_entered"][0]
                b += 16 * sum(n * rects_per_stage[st] for st, n in enumerate(l["stage_entered"]))
            per_kernel[kind] = {"kernel": kname[kind], "launches_per_step": grp["n"], "ms_per_step": round(grp["ms"], 3),
                                "algorithmic_GB_per_step": round(b / 1e9, 2),
                                "achieved_GBps": round(b / (grp["ms"] * 1e-3) / 1e9, 1) if grp["ms"] > 0 else None}
        achieved = alg / (dom["ms"] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_dominant.json")   # from rocprofv3 --pmc passes (tools/pmc_traffic.py)
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("kernel_kind") == dom_kind:
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": kname[dom_kind], "launches_per_step": dom["n"],
                    "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic,
                    "algorithmic_bytes_per_launch": int(alg // dom["n"]),
                    "avg_launch_ms": round(dom["ms"] / dom["n"], 4),
                    "note": "gathers are served from LDS tiles / L2, not HBM: see DESIGN.md for the LDS and "
                            "texture-address ceilings that actually bound these kernels"}

        # ---- CPU baseline: the oracle, 1 thread, on the first --cpu-frames frames (also a parity check)
        cpu = None
        cpu_mt = None
        cpu_cv = None
        parity = None
        if world == 1 and args.cpu_frames > 0:
            from oracle.oracle import Oracle, load_vjc
            from clfacedetection_amd.api import DATA_DIR
            o = Oracle()
            a = load_vjc(os.path.join(DATA_DIR, f"haarcascade_{args.cascade}.vjc"))
            n_cpu = min(args.cpu_frames, B)
            t1 = time.perf_counter()
            parity = True
            for f in range(n_cpu):
                ro, _ = o.detect(a, frames_h[f])
                mine = counted.rects[counted.rects["frame"] == f]
                same = len(ro) == len(mine) and all(np.array_equal(ro[k], mine[k])
                                                    for k in ("scale_idx", "x", "y", "w", "h"))
                parity = parity and bool(same)
            cpu_s = time.perf_counter() - t1
            cpu = {"value": round(windows_per_frame * n_cpu / cpu_s, 1), "unit": "windows/s", "cores": 1,
                   "kind": "port", "sample": f"first {n_cpu} of the {B} frames, {cpu_s:.1f} s, oracle/vj_oracle.c "
                   f"(gcc -O2 -ffp-contract=off, integral + all scales + all stages)"}
            # the same oracle on every host core the box gives us (one frame per thread; ctypes drops the GIL)
            from concurrent.futures import ThreadPoolExecutor
            n_thr = max(1, min(len(os.sched_getaffinity(0)), B))
            t2 = time.perf_counter()
            with ThreadPoolExecutor(n_thr) as ex:
                list(ex.map(lambda f: o.detect(a, frames_h[f]), range(n_thr)))
            mt_s = time.perf_counter() - t2
            cpu_mt = {"value": round(windows_per_frame * n_thr / mt_s, 1), "unit": "windows/s", "cores": n_thr,
                      "kind": "port", "sample": f"{n_thr} frames, one per thread, {mt_s:.1f} s"}
            # north_star's other CPU leg: the OpenCV-style scale-cascade path as tempcv.cpp keeps it (f64 sums,
            # ystep = max(2, factor), stage-0 skip).  It visits fewer windows than clod by design, so its honest
            # unit is frames/s; compare with this line's "frames_per_s".
            t3 = time.perf_counter()
            vis = 0
            for f in range(n_cpu):
                _, st_cv = o.detect_opencvlike(a, frames_h[f])
                vis += st_cv["windows"]
            cv_s = time.perf_counter() - t3
            t4 = time.perf_counter()
            with ThreadPoolExecutor(n_thr) as ex:
                list(ex.map(lambda f: o.detect_opencvlike(a, frames_h[f]), range(n_thr)))
            cvmt_s = time.perf_counter() - t4
            cpu_cv = {"value": round(n_cpu / cv_s, 3), "unit": "frames/s", "cores": 1, "kind": "port",
                      "all_cores": {"value": round(n_thr / cvmt_s, 3), "unit": "frames/s", "cores": n_thr},
                      "windows_visited_per_frame": vis // n_cpu,
                      "sample": f"first {n_cpu} frames, {cv_s:.1f} s, oc_detect_opencvlike (restates tempcv.cpp "
                                f"cvHaarDetectObjects; unpinned, timing only)"}
        # the OpenCV arithmetic profile on the same frames (vj_detect_opencv: f64 sums, skip rule; one global-gather
        # kernel, not tuned): what a cvHaarDetectObjects user gets, next to cpu_baseline_opencvlike
        cv_profile = None
        if world == 1:
            env.detect_opencv(casc, dframes)
            t5 = time.perf_counter()
            rcv = env.detect_opencv(casc, dframes, flags=VJ_FLAG_COUNTERS)
            cv_s = time.perf_counter() - t5
            cv_profile = {"frames_per_s": round(B / cv_s, 1), "ms_per_step": round(cv_s * 1e3, 2),
                          "windows_visited_per_frame": rcv.windows // B, "detections": len(rcv.rects), "dtype": "f64"}
        out = {
            "metric": "candidate windows/sec, 1080p, haarcascade_frontalface_alt",
            "value": round(value, 1), "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32 gathers + f32 stage sums", "data": "synthetic",
            "config": {"workload": f"{B}x{W}x{H} 8-bit frames per GPU per step (noise/smooth/blocks mix), "
                                   f"haarcascade_{args.cascade}, scaleFactor 1.1f, raw candidates, frames resident in HBM",
                       "frames_per_gpu": B, "windows_per_frame": windows_per_frame,
                       "pass_split": [x[0] for x in counted.passes], "device": env.device_name},
            "mpix_per_s": round(W * H * B * world * args.steps / elapsed / 1e6, 1),
            "frames_per_s": round(B * world * args.steps / elapsed, 1),
            "detections_last_step": int(n_det_total),
            "kernel_ms_per_step": {"integral": round(integral_ms / K, 4), "cascade": round(cascade_ms / K, 4),
                                   "cascade_passes": [round(x / K, 4) for x in pass_ms],
                                   "launches": [{"kind": l["kind"], "lds_class": l["lds_class"],
                                                 "stages": [l["stage_begin"], l["stage_end"]], "n_scales": len(l["scales"]),
                                                 "ms": round(ms / K, 4)} for l, ms in zip(launches, launch_ms)]},
            # the one genuinely HBM-bound kernel group: 1 B read + 12 B written per pixel
            "integral_roofline": {"bound": "hbm", "achieved": round(13 * W * H * B / (integral_ms / K * 1e-3) / 1e9, 1),
                                  "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                  "frac": round(13 * W * H * B / (integral_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)},
            "stump_evals_per_window": round(counted.stump_evals / max(counted.windows, 1), 3),
            "cascade_algorithmic_GBps": round(counted.gather_bytes / (cascade_ms / K * 1e-3) / 1e9, 2),
            "roofline": roofline, "kernels": per_kernel, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_mt,
            "cpu_baseline_opencvlike": cpu_cv, "opencv_profile": cv_profile, "parity_sample_ok": parity,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out), flush=True)
        if out["parity_sample_ok"] is False:
            print("bench.py: PARITY FAILURE on the CPU sample", file=sys.stderr)
            return 4
    return 0


if __name__ == "__main__":
    sys.exit(main())
