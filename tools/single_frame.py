"""Per-launch times of a single 1080p frame (BASELINE config 2) for a few tunable settings."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
env = Environment(0); c = Cascade.load("frontalface_alt")
frames = synth.batch(6, 1080, 1920, seed0=1)
def run(tag):
    for f in frames[:2]: env.detect(c, f)
    lat = []; tot = []; lm = None
    for i in range(30):
        t0 = time.perf_counter(); r = env.detect(c, frames[i % 6]); lat.append((time.perf_counter() - t0) * 1e3)
        tot.append(r.total_ms)
        l = [x["ms"] for x in r.launches]
        lm = l if lm is None else [a + b for a, b in zip(lm, l)]
    print(f"{tag}: p50 {np.percentile(lat,50):.3f} ms kernels {np.percentile(tot,50):.3f} integral {r.integral_ms:.3f} | " +
          " ".join(f"{x['kind']}{x['lds_class']}:{y/30:.3f}" for x, y in zip(r.launches, lm)), flush=True)
run("default")
def cfg(**kw):
    for k, v in kw.items(): env.configure(k, v)
base = dict(tile_split="0.5", pass_cut_nodes="150", tile_accept_windows="768", tile_min_windows="768", tile_max_dwords_per_window="600")
for name, kw in (("split0", dict(tile_split="0")),
                 ("split0 cut60", dict(tile_split="0", pass_cut_nodes="60")),
                 ("split0 accept256", dict(tile_split="0", tile_accept_windows="256", tile_min_windows="256")),
                 ("split0 accept256 dw2000", dict(tile_split="0", tile_accept_windows="256", tile_min_windows="256", tile_max_dwords_per_window="2000")),
                 ("split0 accept128 dw4000", dict(tile_split="0", tile_accept_windows="128", tile_min_windows="128", tile_max_dwords_per_window="4000")),
                 ("split0 accept128 dw4000 cut60", dict(tile_split="0", tile_accept_windows="128", tile_min_windows="128", tile_max_dwords_per_window="4000", pass_cut_nodes="60")),
                 ("split0 accept64 dw8000 cut60", dict(tile_split="0", tile_accept_windows="64", tile_min_windows="64", tile_max_dwords_per_window="8000", pass_cut_nodes="60")),
                 ):
    cfg(**base); cfg(**kw)
    run(name)
