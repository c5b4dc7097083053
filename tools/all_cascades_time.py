"""Every shipped cascade on the same batch (16 x 1080p, the bench textures): clod profile (raw candidates; tilted rectangles read as the
reference reads them) and OpenCV profile — ms per call, windows, node evaluations per second.  Looks for cascades that fall on a slow path.
    python tools/all_cascades_time.py [frames]      (on the GPU box)"""
import glob, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clfacedetection_amd import Cascade, DeviceFrames, Environment, VJ_FLAG_COUNTERS, VJ_FLAG_TILTED_AS_UPRIGHT, default_params, synth
from clfacedetection_amd.api import DATA_DIR
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
env = Environment(0)
t = torch.from_numpy(synth.batch(B, 1080, 1920, seed0=1)).cuda(); torch.cuda.synchronize()
df = DeviceFrames.from_torch(t)
names = sorted(os.path.basename(p)[len("haarcascade_"):-4] for p in glob.glob(os.path.join(DATA_DIR, "haarcascade_*.vjc")))
for name in names:
    c = Cascade.load(name)
    i = c.info
    p = default_params(flags=VJ_FLAG_TILTED_AS_UPRIGHT)
    rc = env.detect(c, df, default_params(flags=VJ_FLAG_TILTED_AS_UPRIGHT | VJ_FLAG_COUNTERS))
    for _ in range(14): env.detect(c, df, p)          # (the chain balance settles)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); r = env.detect(c, df, p); ts.append(time.perf_counter() - t0)
    ms = sorted(ts)[2] * 1e3
    env.detect_opencv(c, df)
    tc = []
    for _ in range(3):
        t0 = time.perf_counter(); ro = env.detect_opencv(c, df); tc.append(time.perf_counter() - t0)
    print(f"{name:22s} {i.win_w}x{i.win_h} stages {i.n_stages} nodes {i.n_nodes} tilted {i.n_tilted} | clod {ms:7.2f} ms, {rc.windows / B / 1e6:.2f} M windows per frame, "
          f"{rc.stump_evals / rc.windows:.1f} nodes per window, {rc.stump_evals / ms / 1e6:.0f} G nodes/s, {len(r.rects)} candidates | OpenCV profile {sorted(tc)[1] * 1e3:7.2f} ms, {len(ro.rects)} detections", flush=True)
