"""Which entry point, if any, keeps device memory: free memory before / after many repetitions of one operation each.
     python tools/leak_probe.py      (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, Environment, default_params, synth
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
free_at_start = torch.cuda.mem_get_info()[0] / 2**20
env = Environment(0)
c = Cascade.load("frontalface_alt"); eye = Cascade.load("eye"); tree = Cascade.load("frontalface_alt_tree")
img = synth.frame("noise", 1, 300, 400)
rng = np.random.default_rng(1)
def free():
    torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0] / 2**20
def probe(name, fn, n):
    fn(0); fn(1)
    a = free()
    for i in range(n): fn(i)
    b = free()
    print(f"{name}: {n} repetitions, free {a:.0f} -> {b:.0f} MiB ({(a - b) * 1024 / n:+.1f} KiB per repetition)", flush=True)
def env_cycle(i):
    e = Environment(0); e.detect(c, img); e.close()
def env_cycle_empty(i):
    e = Environment(0); e.close()
def stream_cycle(i):
    s = env.stream(c, 400, 300, 3); s.submit([img, img]); s.collect(); s.close()
def detect_sizes(i):
    h, w = 100 + (i * 37) % 500, 120 + (i * 53) % 700
    env.detect(c, synth.frame("noise", 1, h, w))
def opencv_sizes(i):
    h, w = 100 + (i * 37) % 500, 120 + (i * 53) % 700
    env.detect_opencv(tree if i % 2 else c, synth.frame("noise", 1, h, w))
def chain_sizes(i):
    h, w = 150 + (i * 37) % 400, 160 + (i * 53) % 600
    env.detect_chain(c, eye, synth.frame("blocks", 2, h, w), default_params())
def same_call(i):
    env.detect(c, img)
if len(sys.argv) > 1:
    env.configure("plan_cache_max", sys.argv[1])
probe("same vj_detect call", same_call, 2000)
probe("vj_detect, a new frame size every call", detect_sizes, 1500)
probe("vj_detect_opencv, a new frame size every call", opencv_sizes, 1000)
probe("vj_detect_chain, a new frame size every call", chain_sizes, 600)
probe("vj_stream create / submit / collect / destroy", stream_cycle, 600)
probe("environment create / destroy (no call)", env_cycle_empty, 300)
probe("environment create / detect / destroy", env_cycle, 300)
probe("vj_detect_opencv, a new frame size every call (again)", opencv_sizes, 1000)
probe("vj_detect_opencv, a new frame size every call (third time)", opencv_sizes, 1000)
before_close = free()
env.close()
print(f"free memory before the first environment {free_at_start:.0f} MiB, before closing it {before_close:.0f} MiB, after closing it {free():.0f} MiB")
