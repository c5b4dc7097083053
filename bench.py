#!/usr/bin/env python3
"""Headline benchmark of the MI355X-native Viola–Jones detect path.

Metric (BASELINE.json): candidate windows/sec (+ Mpix/s) on 1080p frames with
haarcascade_frontalface_alt, 1/2/4/8 GPUs — BASELINE config 3.  A "step" is one pass of the whole hot path
(integral + squared-integral kernels, all cascade passes, detection read-back and — for
N > 1 — the all-gather of detection rectangles) over one batch of synthetic frames that
is already resident in HBM.  BASELINE config 3 is a FIXED batch of 64 frames sharded over the GPUs, so the headline
line is strong scaling: --frames is the whole job's batch and rank r takes its vj_shard_frames block of whole
frames (it integrates only its own frames; the only collective is the final gather of rectangles).  For N > 1 the
same run also times the weak-scaling variant (--frames per GPU) into `weak_scaling` and the same fixed job under the OTHER
split into `other_sharding` (--shard levels: BASELINE config 3 as worded — the pyramid levels sharded, every rank
integrating every frame; --shard frames, the default and the better split, shards whole frames); `per_rank_ms_per_step`,
`allgather_ms_per_step` and `collectives_per_step` say what every rank did and what the one collective cost.
`--scaling weak` makes the weak variant the headline instead.  At N = 1 all of them are the same run.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` is whole-job windows/s over the K timed steps
(max over ranks of the wall time, barrier + synchronize on both sides).

`roofline` is for the dominant kernel — the kernel group with the most time per step — against the ceiling that
binds THAT kernel: the LDS-tile kernel gathers from LDS (`bound: "lds"`, peak = the guide's
aggregate ds_read_b32 rate), the global-gather passes from the L2 (`"l2"`), the integral kernels stream HBM
(`"hbm"`).  `achieved` = algorithmic bytes per launch (48 B per window + 16 B per evaluated rectangle,
from a counted run's PER-LAUNCH counters) over the launch's HIP-event time, measured live on the stream the kernel
runs on.  `kernels` carries the same figures for every group.  `traffic` (HBM bytes per launch from rocprofv3 PMC
passes) is only reported when profiles/pmc_dominant.json was collected from the kernel sources of this very build.

`cpu_baseline` is the CPU oracle (oracle/vj_oracle.c, a single-threaded restatement of the reference's clod path)
timed on a bounded sample of the same frames; the sample doubles as a parity check of the TIMED (uncounted) step.

`extra` (N = 1 only; --extras "" turns it off) measures the other BASELINE configs in the same run:
  config2  one 1080p frame, host buffer in -> rectangles out: p50 / p90 latency over >= 50 calls
  config3_host_frames  config 3 with the frames in (page-locked) HOST memory: double-buffered vj_stream, H2D included
  config4  one 4096x4096 frame, frontalface_alt_tree
  config5  256 x 720p, frontalface_alt2 -> haarcascade_eye inside the grouped faces (and: inside every raw candidate), hand-off on the device
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec; L2 ~34.5 TB/s aggregate; LDS ~75 TB/s aggregate for ds_read_b32 gathers
# (128 B/clk/CU x 256 CUs at ~2.4 GHz; b64 / b128 reads reach ~150 TB/s but a dword gather cannot use them)
PEAK_GBPS = {"hbm": 8000.0, "l2": 34500.0, "lds": 75000.0}
KERNEL_OF = {"tile": ("vj::cascade_tile_pass<false, false, true>", "lds"),
             "block": ("vj::cascade_tile_pass<false, false, false>", "l2"),
             "grid": ("vj::cascade_pass<true, false, *, false, false>", "l2"),
             "queue": ("vj::cascade_pass<false, false, *, false, false>", "l2")}


def kernel_source_hash() -> str:
    h = hashlib.sha256()
    for f in ("vj_kernels.hip", "vj_device.hpp", "vj_devutil.hpp"):
        h.update(open(os.path.join(ROOT, "clfacedetection_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def rows_of(rects, frame=None):
    r = rects if frame is None else rects[rects["frame"] == frame]
    return [tuple(int(x[k]) for k in ("scale_idx", "x", "y", "w", "h")) for x in r]


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=64,
                    help="1080p frames per step: of the whole job (strong scaling) or per GPU (weak)")
    ap.add_argument("--shard", choices=("frames", "levels"), default="frames",
                    help="what the headline shards for N > 1: whole frames (vj_shard_frames; the better split for a batch) or the "
                         "pyramid levels (vj_shard_scales: BASELINE config 3 as worded — every rank integrates every frame and runs "
                         "its share of the scales); the other split of the same fixed job is timed in the same run (`other_sharding`)")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong: --frames frames in total, sharded by whole frames (BASELINE config 3 as stated); "
                         "weak: --frames frames per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--cascade", default="frontalface_alt")
    ap.add_argument("--cpu-frames", type=int, default=8, help="frames of the batch timed on the CPU oracle (0 = skip)")
    ap.add_argument("--extras", "--config", default="1,2,3h,4,5", help="other BASELINE configs measured into `extra` "
                    "(1 = the reference's 640 x 480 case, 2, 3h = config 3 from host frames, 4, 5; \"\" = none)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "rehearse the N > 1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--no-pipeline", action="store_true", help="time blocking vj_detect calls instead of a two-deep vj_stream")
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
    H, W = args.height, args.width
    windows_per_frame = casc.count_windows(W, H)
    kinds = ("noise", "smooth", "blocks")
    p = default_params()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gather = multigpu.RectGather(device=coll_dev) if world > 1 else None      # one collective per step, capacity kept across steps

    def measure(scaling, shard="frames"):
        """One variant: this rank's frames on the device, a counted run, warm-up, args.steps timed steps."""
        p = default_params()
        if shard == "levels":      # every rank: all frames of the job, its share of the pyramid levels (strong scaling only)
            first, B, job_frames = 0, args.frames, args.frames
            p = default_params(scales=casc.shard_scales(W, H, rank, world)) if world > 1 else p
        elif scaling == "strong":
            mine = multigpu.shard_frames(args.frames, rank, world)      # == vj_shard_frames: contiguous blocks of whole frames
            first, B, job_frames = mine.start, len(mine), args.frames
        else:
            first, B, job_frames = rank * args.frames, args.frames, args.frames * world
        # global frame g has seed 1 + g and kind g % 3 whatever the number of ranks: the job's result does not depend on N
        frames_h = np.empty((max(B, 1), H, W), np.uint8)
        for i in range(max(B, 1)):
            frames_h[i] = synth.frame(kinds[(first + i) % 3], 1 + first + i, H, W)
        frames_d = torch.from_numpy(frames_h).to(dev)
        torch.cuda.synchronize()
        dframes = DeviceFrames.from_torch(frames_d[:B]) if B > 0 else None

        def finish(r):
            rects = r.rects
            if world > 1:
                rects = rects.copy()
                rects["frame"] += first             # global frame index
                rects = gather(rects)
            return r, rects

        def step(params):
            return finish(env.detect(casc, dframes if B > 0 else [], params))

        # one counted run: algorithmic bytes per launch (the counted kernel variants are slower and never timed)
        pc = default_params(flags=VJ_FLAG_COUNTERS)
        pc.scale_mask[0], pc.scale_mask[1] = p.scale_mask[0], p.scale_mask[1]
        counted, _ = step(pc)
        # The timed steps go through a vj_stream (two batches in flight): step k's read-back, decode and sort on the host
        # overlap step k+1's kernels.  Every step still does everything — integral images, all cascade passes, read-back,
        # sorted rectangles (and the all-gather for N > 1) — and all of it completes inside the timed bracket; the frames
        # are device-resident, so the stream uses them in place (no copy).  --no-pipeline times blocking calls instead.
        stream = None if (args.no_pipeline or B == 0) else env.stream(casc, W, H, B, p)

        def run_steps(n):
            """n whole steps; yields (result, gathered rects) of each."""
            if stream is None:
                for _ in range(n):
                    yield step(p)
                return
            if n == 0:
                return
            stream.submit(dframes)
            for _ in range(n - 1):
                stream.submit(dframes)
                yield finish(stream.collect())
            yield finish(stream.collect())

        for _ in run_steps(args.warmup):
            pass
        barrier()
        if gather is not None:
            gather.total_ms, g0 = 0.0, gather.n_collectives
        t0 = time.perf_counter()
        m = {"integral_ms": 0.0, "cascade_ms": 0.0, "pass_ms": None, "launch_ms": None, "launches": None,
             "n_det_total": 0, "timed_rects": None}
        for r, rects in run_steps(args.steps):
            m["integral_ms"] += r.integral_ms
            m["cascade_ms"] += r.cascade_ms
            pm = [x[2] for x in r.passes]
            m["pass_ms"] = pm if m["pass_ms"] is None else [a + b for a, b in zip(m["pass_ms"], pm)]
            lm = [l["ms"] for l in r.launches]
            m["launch_ms"] = lm if m["launch_ms"] is None else [a + b for a, b in zip(m["launch_ms"], lm)]
            m["launches"] = r.launches
            m["n_det_total"] = len(rects)
            m["timed_rects"] = r.rects
        own = time.perf_counter() - t0        # this rank's own steps (before it waits for the others)
        barrier()
        elapsed = time.perf_counter() - t0
        per_rank = [own]
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            mine_t = torch.tensor([own], dtype=torch.float64, device=coll_dev)
            alls = [torch.zeros_like(mine_t) for _ in range(world)]
            dist.all_gather(alls, mine_t)
            per_rank = [float(x.item()) for x in alls]
        if stream is not None:
            stream.close()
        K = max(args.steps, 1)
        m.update(shard=shard, allgather_ms=(round(gather.total_ms / K, 4) if gather is not None else None),
                 collectives_per_step=(round((gather.n_collectives - g0) / K, 3) if gather is not None else None))
        m.update(scaling=scaling, first=first, B=B, job_frames=job_frames, frames_h=frames_h, dframes=dframes, counted=counted,
                 elapsed=elapsed, ms_per_step=1e3 * elapsed / K, per_rank_ms=[round(1e3 * x / K, 4) for x in per_rank],
                 value=windows_per_frame * job_frames * args.steps / elapsed, keep=frames_d)
        return m

    m = measure(args.scaling if args.shard == "frames" else "strong", args.shard)
    other = other_shard = None
    if world > 1:      # the other variants in the same run, so that every figure comes from the same box and build
        brief = lambda o: {"scaling": o["scaling"], "shard": o["shard"], "value": round(o["value"], 1), "unit": "windows/s",
                           "ms_per_step": round(o["ms_per_step"], 4), "frames_per_step_whole_job": o["job_frames"],
                           "per_rank_ms_per_step": o["per_rank_ms"], "allgather_ms_per_step": o["allgather_ms"],
                           "collectives_per_step": o["collectives_per_step"]}
        if args.shard == "frames":
            o = measure("weak" if args.scaling == "strong" else "strong")
            other = brief(o)
            del o
        o = measure("strong", "levels" if args.shard == "frames" else "frames")      # the same fixed job under the other split
        other_shard = brief(o)
        del o
    elapsed, ms_per_step, value = m["elapsed"], m["ms_per_step"], m["value"]
    B, frames_h, dframes, counted = m["B"], m["frames_h"], m["dframes"], m["counted"]
    integral_ms, cascade_ms, pass_ms, launch_ms, launches = m["integral_ms"], m["cascade_ms"], m["pass_ms"], m["launch_ms"], m["launches"]
    n_det_total, timed_rects = m["n_det_total"], m["timed_rects"]

    out = None
    if rank == 0:
        K = max(args.steps, 1)
        # ---- per-kernel rooflines.  Launches are grouped by kernel (the names rocprofv3 --kernel-trace --stats
        # reports); algorithmic bytes come from the counted run's PER-LAUNCH counters, time from the
        # HIP events the library records around every launch on the stream it runs on, averaged over the timed steps.
        nodes, trees, stages = casc.nodes, casc.trees, casc.stages
        rects_per_stage = []
        for st in stages:
            tr = trees[st["first_tree"]:st["first_tree"] + st["n_trees"]]
            rects_per_stage.append(int(sum(int(nodes["n_rects"][t["first_node"]:t["first_node"] + t["n_nodes"]].sum())
                                           for t in tr)))
        groups = {}
        for l, ms in zip(launches, launch_ms):
            g = groups.setdefault(l["kind"], {"ms": 0.0, "n": 0})
            g["ms"] += ms / K
            g["n"] += 1
        per_kernel = {}
        for kind, grp in groups.items():
            b = 0
            for l in counted.launches:
                if l["kind"] != kind:
                    continue
                if kind != "queue":
                    b += 48 * l["stage_entered"][0]
                b += 16 * sum(n * rects_per_stage[st] for st, n in enumerate(l["stage_entered"]))
            name, bound = KERNEL_OF[kind]
            ach = b / (grp["ms"] * 1e-3) / 1e9 if grp["ms"] > 0 else 0.0
            per_kernel[kind] = {"kernel": name, "bound": bound, "launches_per_step": grp["n"], "ms_per_step": round(grp["ms"], 3),
                                "avg_launch_ms": round(grp["ms"] / grp["n"], 4),
                                "algorithmic_bytes_per_launch": int(b // grp["n"]),
                                "achieved": round(ach, 1), "peak": PEAK_GBPS[bound], "unit": "GB/s",
                                "frac": round(ach / PEAK_GBPS[bound], 4)}
            if bound == "l2":
                # what binds a global gather is the texture-address unit, one per CU, which takes every lane of an uncoalesced
                # wave-load in turn: the figure that compares passes is CU-cycles per lane-gather (4 bytes each; profiles/r04_notes.md
                # has the counters).  Round 3 printed a "ceiling" measured at ONE lane stride here, which a pass that spans strides
                # 3.8 ... 50 can beat: removed.
                w_in = sum(l["stage_entered"][0] for l in counted.launches if l["kind"] == kind) if kind != "queue" else 0
                lane_gathers = (b - 48 * w_in) / 4 + 8 * w_in      # four corners per evaluated rectangle + the variance's 4 + 4 corners
                per_kernel[kind]["texture_address"] = {
                    "lane_gathers_per_step": int(lane_gathers),
                    "cu_ns_per_lane_gather": round(grp["ms"] * 1e6 * 256 / max(lane_gathers, 1), 4),
                    "note": "launch time x 256 CUs / lane-gathers (rectangle corners + the variance's eight corners); "
                            "x clock in GHz = texture-address cycles per lane"}
        int_ach = 13 * W * H * B / (integral_ms / K * 1e-3) / 1e9
        per_kernel["integral"] = {"kernel": "vj::band_colsum + vj::band_scan + vj::band_rows", "bound": "hbm", "launches_per_step": 3,
                                  "ms_per_step": round(integral_ms / K, 4), "algorithmic_bytes_per_launch": 13 * W * H * B,
                                  "achieved": round(int_ach, 1), "peak": PEAK_GBPS["hbm"], "unit": "GB/s",
                                  "frac": round(int_ach / PEAK_GBPS["hbm"], 4),
                                  "note": "1 B read + 12 B written per pixel over the three launches together"}
        dom_kind = max(groups, key=lambda k: groups[k]["ms"])
        dom = per_kernel[dom_kind]
        # measured HBM traffic and SQ counters of the dominant kernel: only from a profile of THIS build's kernels
        traffic, pmc_note, pmc_extra = None, "no rocprofv3 PMC profile of this build's kernels is committed", {}
        pmc = os.path.join(ROOT, "profiles", "pmc_dominant.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("kernel_kind") == dom_kind and j.get("kernel_source_hash") == kernel_source_hash():
                    traffic = j.get("hbm_bytes_per_launch")
                    pmc_note = f"profiles/pmc_dominant.json (from {j.get('source')}, kernel sources {j.get('kernel_source_hash')})"
                    pmc_extra = {k: j[k] for k in ("lds_bank_conflict_share", "valu_busy", "lds_busy", "wave_wait_share") if k in j}
                else:
                    pmc_note = "profiles/pmc_dominant.json was collected from other kernel sources: not reported"
            except Exception:
                pass
        roofline = {"bound": dom["bound"], "kernel": dom["kernel"], "launches_per_step": dom["launches_per_step"],
                    "achieved": dom["achieved"], "peak": dom["peak"], "unit": "GB/s", "frac": dom["frac"],
                    "traffic": traffic, "traffic_source": pmc_note,
                    "algorithmic_bytes_per_launch": dom["algorithmic_bytes_per_launch"], "avg_launch_ms": dom["avg_launch_ms"],
                    "hbm_frac_of_traffic": round(traffic / (dom["avg_launch_ms"] * 1e-3) / 1e9 / PEAK_GBPS["hbm"], 4) if traffic else None,
                    **pmc_extra,
                    "note": "the dominant kernel gathers from image tiles staged in LDS: its ceiling is the LDS gather rate "
                            "(ds_read_b32), not HBM; measured HBM traffic is `traffic`"}

        # ---- CPU baseline: the oracle, 1 thread, on the first --cpu-frames frames; the same sample checks the
        # rectangles of the TIMED step (uncounted kernel variants) and of the counted run
        cpu = cpu_mt = cpu_cv = None
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
                want = rows_of(ro)
                parity = parity and rows_of(timed_rects, f) == want and rows_of(counted.rects, f) == want
            cpu_s = time.perf_counter() - t1
            parity = parity and bool(np.array_equal(timed_rects, counted.rects))   # every frame: timed == counted variants
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
            # north_star's other CPU leg: the OpenCV-style scale-cascade path as tempcv.cpp keeps it.  It visits fewer
            # windows than clod by design, so its honest unit is frames/s; compare with this line's "frames_per_s".
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
        # the OpenCV arithmetic profile on the same frames (vj_detect_opencv): what a cvHaarDetectObjects user gets
        cv_profile = None
        extra = {}
        if world == 1:
            rcv = env.detect_opencv(casc, dframes, flags=VJ_FLAG_COUNTERS)   # (the counted kernel variant is slower: not timed)
            env.detect_opencv(casc, dframes)
            cv_t = []
            for _ in range(3):       # (the GPU idled through the CPU baselines above: one call alone reads the clock ramp)
                t5 = time.perf_counter()
                env.detect_opencv(casc, dframes)
                cv_t.append(time.perf_counter() - t5)
            cv_s = sorted(cv_t)[1]
            cv_profile = {"frames_per_s": round(B / cv_s, 1), "ms_per_step": round(cv_s * 1e3, 2),
                          "windows_visited_per_frame": rcv.windows // B, "detections": len(rcv.rects), "dtype": "f64"}
            extra = run_extras(env, casc, frames_h, [x.strip() for x in args.extras.split(",") if x.strip()], torch)
        out = {
            "metric": "candidate windows/sec, 1080p, haarcascade_frontalface_alt",
            "value": round(value, 1), "unit": "windows/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": m["scaling"], "vs_baseline": None,
            "dtype": "u32 gathers + f32 stage sums", "data": "synthetic",
            "config": {"workload": f"BASELINE config 3: {m['job_frames']}x{W}x{H} 8-bit frames per step in the whole job "
                                   f"(noise/smooth/blocks mix; {'sharded by whole frames over' if m['scaling'] == 'strong' else 'per-GPU batches of ' + str(B) + ' on'} "
                                   f"{world} GPU{'s' if world > 1 else ''}), haarcascade_{args.cascade}, scaleFactor 1.1f, raw candidates, "
                                   f"frames resident in HBM",
                       "frames_per_step_whole_job": m["job_frames"], "frames_on_rank_0": B, "windows_per_frame": windows_per_frame,
                       "parallelism": ((f"whole frames sharded over {world} ranks (vj_shard_frames)" if m["shard"] == "frames" else
                                        f"pyramid levels sharded over {world} ranks (vj_shard_scales; every rank integrates every frame)") +
                                       ", no data-path collective, ONE all-gather of fixed-capacity rectangle blocks per step") if world > 1 else "single GPU",
                       "pass_split": [x[0] for x in counted.passes], "device": env.device_name,
                       "pipeline": "blocking vj_detect calls" if args.no_pipeline else "vj_stream, two batches in flight"},
            "mpix_per_s": round(W * H * m["job_frames"] * args.steps / elapsed / 1e6, 1),
            "frames_per_s": round(m["job_frames"] * args.steps / elapsed, 1),
            "per_rank_ms_per_step": m["per_rank_ms"],
            "torch_world_size": (dist.get_world_size() if world > 1 else 1),      # (torch's number; the C++ host reads ncclCommCount: include/vj_rccl.h)
            "collective_backend": (args.backend if world > 1 else None),
            "allgather_ms_per_step": m["allgather_ms"], "collectives_per_step": m["collectives_per_step"],
            ("weak_scaling" if m["scaling"] == "strong" else "strong_scaling"): other,
            "other_sharding": other_shard,
            "detections_last_step": int(n_det_total),
            "kernel_ms_per_step": {"integral": round(integral_ms / K, 4), "cascade": round(cascade_ms / K, 4),
                                   "cascade_passes": [round(x / K, 4) for x in pass_ms],
                                   "launches": [{"kind": l["kind"], "lds_class": l["lds_class"],
                                                 "stages": [l["stage_begin"], l["stage_end"]], "n_scales": len(l["scales"]),
                                                 "ms": round(ms / K, 4)} for l, ms in zip(launches, launch_ms)]},
            "stump_evals_per_window": round(counted.stump_evals / max(counted.windows, 1), 3),
            "stump_evals_per_s": round(counted.stump_evals / (ms_per_step * 1e-3), 1),
            "roofline": roofline, "kernels": per_kernel, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_mt,
            "cpu_baseline_opencvlike": cpu_cv, "opencv_profile": cv_profile, "parity_sample_ok": parity, "extra": extra,
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


def three_states(env, call, torch, timed=8) -> dict:
    """A workload whose chain balance the library finds by feedback, in the three states a caller can meet it: `static` — the
    feedback switched off, the batch-size defaults —, `first_call` — feedback on, nothing known: the one-shot caller's time,
    plan building included —, and `settled` — after the search, whose length is reported (`feedback_calls`, measured by the
    library; `calls_until_settled` counts every call made meanwhile)."""
    pct = lambda v, q: float(np.percentile(np.asarray(v), q))

    def time_calls(k):
        lat, r = [], None
        for _ in range(k):
            torch.cuda.synchronize()
            t = time.perf_counter()
            r = call()
            lat.append((time.perf_counter() - t) * 1e3)
        return lat, r
    env.configure("auto_balance", "0")
    time_calls(3)
    static, r = time_calls(timed)
    static_split = r.tile_split
    env.configure("auto_balance", "reset")       # forget every workload, feedback on again
    env.configure("auto_balance", "1")
    first, r = time_calls(1)
    n_calls = 1
    while r.balance_state == 1 and n_calls < 64:
        r = call()
        n_calls += 1
    settled, r = time_calls(timed)
    return {"static_ms_p50": round(pct(static, 50), 3), "static_tile_split": round(static_split, 3),
            "first_call_ms": round(first[0], 3), "settled_ms_p50": round(pct(settled, 50), 3),
            "settled_tile_split": round(r.tile_split, 3), "feedback_calls": int(r.balance_calls), "calls_until_settled": n_calls,
            "balance_state": int(r.balance_state), "detections": len(r.rects)}


def run_extras(env, casc_alt, frames_h, which, torch) -> dict:
    """The other BASELINE configs, bounded to a few seconds each.  Synthetic frames; every figure is wall-clock around
    whole library calls (host in -> rectangles out), so launch overhead, read-back and the host-side sort are included."""
    from clfacedetection_amd import Cascade, DeviceFrames, default_params, synth
    extra = {}
    pct = lambda v, q: float(np.percentile(np.asarray(v), q))
    if "1" in which:
        # config 1 (BASELINE configs[0]): the reference's own CPU-runnable case — one 640 x 480 frame, frontalface_default, scaleFactor 1.1,
        # minNeighbors 3 — as the device answers it: host frame in, grouped rectangles out (clod arithmetic; the OpenCV profile beside it)
        c1 = Cascade.load("frontalface_default")
        f1 = synth.batch(8, 480, 640, seed0=101, kinds=("faces", "noise", "smooth", "blocks"))
        p1 = default_params(min_neighbors=3)
        for _ in range(5):
            env.detect(c1, f1[0], p1)
            env.detect_opencv(c1, f1[0], min_neighbors=3)
        lat, lat_cv, n_out = [], [], 0
        for i in range(60):
            t = time.perf_counter()
            r = env.detect(c1, f1[i % 8], p1)
            lat.append((time.perf_counter() - t) * 1e3)
            n_out += len(r.rects)
            t = time.perf_counter()
            env.detect_opencv(c1, f1[i % 8], min_neighbors=3)
            lat_cv.append((time.perf_counter() - t) * 1e3)
        extra["config1"] = {"workload": "1x640x480, frontalface_default, minNeighbors 3, host frame in -> grouped rects out, 60 calls",
                            "latency_ms_p50": round(pct(lat, 50), 3), "latency_ms_p90": round(pct(lat, 90), 3),
                            "opencv_profile_latency_ms_p50": round(pct(lat_cv, 50), 3), "grouped_rects_in_60_calls": n_out}
    if "2" in which:
        # config 2: one 1080p frame per call, the reference's per-frame pattern (main.cpp:159-184): host buffer in, rects out
        f = frames_h[0]
        ws = casc_alt.count_windows(f.shape[1], f.shape[0])
        for _ in range(5):
            env.detect(casc_alt, f)
        lat, kern, integ = [], [], []
        for i in range(60):
            img = frames_h[i % len(frames_h)]
            t = time.perf_counter()
            r = env.detect(casc_alt, img)
            lat.append((time.perf_counter() - t) * 1e3)
            kern.append(r.total_ms)
            integ.append(r.integral_ms)
        extra["config2"] = {"workload": "1x1920x1080, frontalface_alt, host frame in -> rects out, 60 calls",
                            "latency_ms_p50": round(pct(lat, 50), 3), "latency_ms_p90": round(pct(lat, 90), 3),
                            "kernels_ms_p50": round(pct(kern, 50), 3), "integral_ms_p50": round(pct(integ, 50), 4),
                            "windows_per_s_at_p50": round(ws / (pct(lat, 50) * 1e-3), 1)}
    if "3h" in which:
        # config 3 with the frames in page-locked HOST memory: vj_stream uploads batch k+1 while batch k computes
        B, H, W = frames_h.shape
        st = env.stream(casc_alt, W, H, B)
        bufs = [env.host_alloc((B, H, W)) for _ in range(2)]
        try:
            for b in bufs:
                b[...] = frames_h
            st.submit(bufs[0])        # warm-up: one whole batch through the stream, collected before the clock starts
            st.collect()
            n = 12
            torch.cuda.synchronize()
            t = time.perf_counter()
            # n batches, every one of them SUBMITTED AND COLLECTED inside the bracket (round 3 collected n - 1 of the n it timed
            # and divided by n: 8/7 too fast).  The first upload of the bracket has nothing to hide behind: 1/n of the H2D time
            # is in the figure, as it is for any caller who starts a stream.
            st.submit(bufs[0])
            for k in range(1, n):
                st.submit(bufs[k % 2])
                st.collect()
            st.collect()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / n
            extra["config3_host_frames"] = {"workload": f"{B}x{W}x{H} frames in page-locked host memory, double-buffered vj_stream "
                                                        f"(H2D of batch k+1 overlaps the kernels of batch k), {n} batches submitted and collected inside the bracket",
                                            "ms_per_step": round(dt * 1e3, 3),
                                            "windows_per_s": round(casc_alt.count_windows(W, H) * B / dt, 1)}
        finally:
            st.close()
            for b in bufs:
                env.host_free(b)
    if "4" in which:
        c = Cascade.load("frontalface_alt_tree")
        f = synth.batch(1, 4096, 4096, seed0=4001, kinds=("blocks",))
        d = torch.from_numpy(f).cuda()
        df = DeviceFrames.from_torch(d)
        ws = c.count_windows(4096, 4096)
        three = three_states(env, lambda: env.detect(c, df), torch, timed=8)
        extra["config4"] = {"workload": "1x4096x4096 (blocks), frontalface_alt_tree, frame resident in HBM, 8 calls per state",
                            "ms_p50": three["settled_ms_p50"], "windows_per_frame": ws,
                            "windows_per_s": round(ws / (three["settled_ms_p50"] * 1e-3), 1), "detections": three["detections"], **three}
        del d
    if "5" in which:
        face, eye = Cascade.load("frontalface_alt2"), Cascade.load("eye")
        n = 256
        # a quarter of the frames carry drawn faces (clusters of candidates that grouping keeps), the rest is the config-3 mix
        f = synth.batch(n, 720, 1280, seed0=5001, kinds=("faces", "noise", "smooth", "blocks"))
        d = torch.from_numpy(f).cuda()
        df = DeviceFrames.from_torch(d)
        ws = face.count_windows(1280, 720)
        for tag, p1, what in (("config5", default_params(min_neighbors=3), "the faces (candidates grouped on the device, minNeighbors 3)"),
                              ("config5_raw_candidates", default_params(), "every raw face candidate")):
            last = {}

            def call():
                last["r"] = env.detect_chain(face, eye, df, p1)
                return last["r"][0]
            three = three_states(env, call, torch, timed=3)
            r1, r2 = last["r"]
            ms = three["settled_ms_p50"]
            extra[tag] = {"workload": "256x1280x720 (faces/noise/smooth/blocks), frontalface_alt2 -> haarcascade_eye inside " + what +
                                      ", regions handed over on the device (vj_detect_chain), frames resident in HBM",
                          "ms_per_step": ms, "frames_per_s": round(n / (ms * 1e-3), 1),
                          "face_windows_per_s": round(ws * n / (ms * 1e-3), 1),
                          "face_regions": len(r1.rects), "eye_candidates": len(r2.rects),
                          "grouping_and_second_cascade_ms": round(r2.cascade_ms, 3), **three}
        del d
    return extra


if __name__ == "__main__":
    sys.exit(main())
