import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from clfacedetection_amd import Cascade, Environment, synth
env = Environment(0)
for kv in sys.argv[1:]:
    env.configure(*kv.split("=", 1))
e2 = Cascade.load("eye"); fa = Cascade.load("frontalface_alt")
for c, n, h, w in ((e2, 2048, 100, 100), (fa, 512, 200, 200), (fa, 64, 240, 320), (fa, 16, 480, 640)):
    many = synth.batch(n, h, w, seed0=3, kinds=("noise", "faces") if min(h, w) >= 130 else ("noise",))
    env.detect(c, many)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); r = env.detect(c, many); ts.append((time.perf_counter() - t0) * 1e3)
    lst = [np.ascontiguousarray(f[:, :w - 1]) for f in many[:min(n, 512)]]      # separate buffers, a width off the 4-byte grid
    env.detect(c, lst)
    t0 = time.perf_counter(); env.detect(c, lst); t_list = (time.perf_counter() - t0) * 1e3
    print(f"   list of {len(lst)} separate {w - 1}x{h} frames: {t_list:.1f} ms wall", flush=True)
    print(f"{n} x {w}x{h}: {min(ts):.1f} ms wall, kernels {r.total_ms:.1f} ms, {len(r.rects)} rects | " + " ".join(f"{x['kind']}{x['lds_class']}:{x['ms']:.2f}" for x in r.launches), flush=True)
