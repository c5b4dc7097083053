import sys, time
sys.path.insert(0, '.')
import numpy as np
from clfacedetection_amd import Cascade, Environment, default_params, VJ_FLAG_COUNTERS
from clfacedetection_amd import synth
from oracle.oracle import Oracle, load_vjc
o = Oracle()
env = Environment(0)
print(env.device_name)
# integral
for (h,w) in [(7,5),(480,640),(1080,1920),(211,317)]:
    g = synth.frame('noise', 3, h, w)
    s,q = env.integral(g); s2,q2 = o.integral(g)
    print('integral', h, w, np.array_equal(s,s2), np.array_equal(q,q2))
for name,kind,h,w in [('frontalface_alt','noise',480,640),('frontalface_default','noise',480,640),('frontalface_alt','smooth',480,640),('eye','blocks',300,400),('frontalface_alt','noise',1080,1920)]:
    c = Cascade.load(name); a = load_vjc(f'clfacedetection_amd/data/haarcascade_{name}.vjc')
    g = synth.frame(kind, 12345, h, w)
    p = default_params(flags=VJ_FLAG_COUNTERS)
    t=time.time(); r = env.detect(c, g, p); t1=time.time()-t
    ro, st = o.detect(a, g)
    same = len(r.rects)==len(ro) and all(np.array_equal(r.rects[k], ro[k]) for k in ['x','y','w','h','scale_idx'])
    print(name, kind, h, w, 'dets', len(r.rects), len(ro), 'same', same, 'stages', r.stage_entered==st['stage_entered'], 'evals', r.stump_evals==st['stump_evals'], 'bytes', r.gather_bytes==st['gather_bytes'], 'ms', round(r.integral_ms,3), round(r.cascade_ms,3), 'wall', round(t1,3))
    if not same:
        print(r.rects[:5], ro[:5]); print(r.stage_entered, st['stage_entered'])
