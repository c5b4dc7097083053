"""The other BASELINE configs as small fixed workloads, for timing by hand and for rocprofv3 (program after "--"):
     python tools/configs.py config2 [calls]      one 1080p frame per call, frontalface_alt, host frame in -> rectangles out
     python tools/configs.py config4 [calls]      one 4096 x 4096 frame, frontalface_alt_tree (stage tree), frame resident in HBM
     python tools/configs.py config5 [calls]      256 x 720p, frontalface_alt2 -> eye inside the grouped faces, on the device
     python tools/configs.py config5raw [calls]   ... inside every raw candidate
     python tools/configs.py cv [calls]           64 x 1080p through the OpenCV arithmetic profile (vj_detect_opencv)
     python tools/configs.py cvtree [calls]       ... with frontalface_alt_tree (stage tree: prefix on tiles, chain sweeps)
   key=value arguments go to vj_env_configure first.  Prints wall ms per call (median) and the per-launch HIP-event times."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, DeviceFrames, Environment, default_params, synth

args = [a for a in sys.argv[1:] if "=" not in a]
what = args[0] if args else "config2"
calls = int(args[1]) if len(args) > 1 else 20
env = Environment(0)
for kv in sys.argv[1:]:
    if "=" in kv:
        env.configure(*kv.split("=", 1))
dev = lambda a: DeviceFrames.from_torch(torch.from_numpy(a).cuda())
if what == "config2":
    c, frames = Cascade.load("frontalface_alt"), synth.batch(8, 1080, 1920, seed0=1, kinds=("noise", "smooth", "blocks", "faces"))
    run = lambda i: env.detect(c, frames[i % len(frames)])
elif what == "config4":
    c, df = Cascade.load("frontalface_alt_tree"), dev(synth.batch(1, 4096, 4096, seed0=4001, kinds=("blocks",)))
    run = lambda i: env.detect(c, df)
elif what in ("config5", "config5raw"):
    face, eye = Cascade.load("frontalface_alt2"), Cascade.load("eye")
    df = dev(synth.batch(256, 720, 1280, seed0=5001, kinds=("faces", "noise", "smooth", "blocks")))
    p1 = default_params(min_neighbors=3 if what == "config5" else 0)
    run = lambda i: env.detect_chain(face, eye, df, p1)[0]
elif what == "cv":
    c, df = Cascade.load("frontalface_alt"), dev(synth.batch(64, 1080, 1920, seed0=1))
    run = lambda i: env.detect_opencv(c, df)
elif what == "cvtree":
    c, df = Cascade.load("frontalface_alt_tree"), dev(synth.batch(64, 1080, 1920, seed0=1))
    run = lambda i: env.detect_opencv(c, df)
else:
    sys.exit(__doc__)
for i in range(30 if what in ("config4", "config5", "config5raw") else 3):      # (the first calls of these workloads settle the chain balance)
    run(i)
torch.cuda.synchronize()
wall, last = [], None
for i in range(calls):
    t = time.perf_counter()
    last = run(i)
    wall.append((time.perf_counter() - t) * 1e3)
print(f"{what}: wall ms/call p50 {np.percentile(wall, 50):.3f} p90 {np.percentile(wall, 90):.3f} | kernels {last.total_ms:.3f} ms "
      f"(integral {last.integral_ms:.3f}) | launches " + " ".join(f"{l['kind']}{l['lds_class']}[{l['stage_begin']},{l['stage_end']}):{l['ms']:.3f}" for l in (last.launches or [])),
      flush=True)
