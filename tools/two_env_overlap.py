"""How much a second batch in flight with its OWN integral images, queues and streams would gain: two environments on one GPU, each
running the bench batch (64 x 1080p, frontalface_alt) from its own thread, against one environment alone.
    python tools/two_env_overlap.py [frames] [calls]      (on the GPU box)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
c = Cascade.load("frontalface_alt")
t = torch.from_numpy(synth.batch(B, 1080, 1920, seed0=1)).cuda(); torch.cuda.synchronize()
envs = [Environment(0), Environment(0)]
dfs = [DeviceFrames.from_torch(t), DeviceFrames.from_torch(t)]
for e, d in zip(envs, dfs):
    for _ in range(45):          # the chain balance settles
        e.detect(c, d)
def run(k, n):
    for _ in range(n):
        envs[k].detect(c, dfs[k])
t0 = time.perf_counter(); run(0, N); t1 = time.perf_counter()
print(f"one environment: {(t1 - t0) / N * 1e3:.2f} ms per batch of {B}")
th = [threading.Thread(target=run, args=(k, N)) for k in range(2)]
t0 = time.perf_counter()
for x in th: x.start()
for x in th: x.join()
t1 = time.perf_counter()
print(f"two environments, one thread each: {(t1 - t0) / (2 * N) * 1e3:.2f} ms per batch of {B} ({2 * N} batches in {(t1 - t0) * 1e3:.1f} ms)")
