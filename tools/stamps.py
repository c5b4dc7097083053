"""Diagnostic build only (VJ_STAMPS=1): per-phase s_memtime shares of the tile kernel (wave 0 of each workgroup)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from clfacedetection_amd import Cascade, Environment, DeviceFrames, default_params, synth
env = Environment(0); c = Cascade.load("frontalface_alt")
B = int(os.environ.get("B", "16"))
t = torch.from_numpy(synth.batch(B, 1080, 1920, seed0=1)).cuda(); torch.cuda.synchronize()
r = env.detect(c, DeviceFrames.from_torch(t), default_params())   # counters flag OFF: slots 40.. hold stamps
import ctypes
# stage_entered slots are only returned for n_stages; read raw via a counted call is not possible -> library prints? use env var hook
