"""Per-call wall time for the shapes a caller may hand over: widths off the 4-byte grid, BGR / BGRA frames, lists of separate
host arrays, strided views, device-resident frames."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, synth
env = Environment(0); c = Cascade.load("frontalface_alt")
def t(tag, fn, reps=12):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"{tag}: p50 {np.percentile(ts, 50):.2f} ms", flush=True)
g = synth.frame("noise", 1, 1080, 1920)
t("gray 1920x1080, contiguous", lambda: env.detect(c, g))
for w in (1919, 1921, 1922):
    gw = synth.frame("noise", 1, 1080, w)
    t(f"gray {w}x1080, contiguous", lambda: env.detect(c, gw))
big = synth.frame("noise", 2, 1200, 2100)
view = big[60:1140, 90:2010]
t("gray 1920x1080 view into 2100x1200 (row stride 2100)", lambda: env.detect(c, view))
bgr = np.repeat(g[:, :, None], 3, 2).copy()
t("BGR 1920x1080", lambda: env.detect(c, bgr, color=True))
bgra = np.repeat(g[:, :, None], 4, 2).copy()
t("BGRA 1920x1080", lambda: env.detect(c, bgra, color=True))
bgr2 = np.repeat(synth.frame("noise", 1, 1080, 1921)[:, :, None], 3, 2).copy()
t("BGR 1921x1080", lambda: env.detect(c, bgr2, color=True))
lst = [synth.frame("noise", k, 1080, 1920) for k in range(8)]
t("8 separate gray frames", lambda: env.detect(c, lst))
arr = np.stack(lst)
t("8 gray frames, one array", lambda: env.detect(c, arr))
d = DeviceFrames.from_torch(torch.from_numpy(arr).cuda())
t("8 gray frames, device-resident", lambda: env.detect(c, d))
t("OpenCV profile gray 1921x1080", lambda: env.detect_opencv(c, synth.frame("noise", 1, 1080, 1921)))
s_, q_ = env.integral(g)
t("vj_integral 1920x1080 (host in, host out)", lambda: env.integral(g))
t("vj_integral 1921x1080", lambda: env.integral(synth.frame("noise", 1, 1080, 1921)))
