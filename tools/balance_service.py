"""What the chain-balance feedback costs a service whose batch sizes vary: `calls` vj_detect calls of
8 ... 96 frames of 1280 x 720 (sizes drawn at random, two cascades), once with the balance keyed on the exact frame count
(round 3: "balance_exact" = 1) and once keyed on the batch-size class (8-15, 16-31, 32-63, >= 64).  Printed per mode: the
share of calls that ran a split other than the best known (candidates of a search), how many workloads finished their
search, mean kernel ms per frame over the whole run and over its last third, and a second environment that IMPORTS the
table the first one exported (its calls must all be on the best split from the first one on).

    python tools/balance_service.py [calls]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from clfacedetection_amd import Cascade, DeviceFrames, Environment, default_params, synth

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 600
H, W = 720, 1280
pool = torch.from_numpy(synth.batch(96, H, W, seed0=1)).cuda()
torch.cuda.synchronize()
cascs = [Cascade.load("frontalface_alt"), Cascade.load("frontalface_default")]


def run(env, tag, seed=7):
    rng = np.random.default_rng(seed)
    ms_pf, states = [], []
    ref = {}
    for i in range(calls):
        n = int(rng.integers(8, 97))
        c = cascs[int(rng.integers(0, 2))]
        r = env.detect(c, DeviceFrames.from_torch(pool[:n]))
        ms_pf.append(r.cascade_ms / n)
        states.append(r.balance_state)
        k = (id(c), n)
        if k in ref:
            assert np.array_equal(ref[k], r.rects), "results changed with the balance"
        else:
            ref[k] = r.rects
    path = os.path.join(tempfile.gettempdir(), f"vj_balance_{tag}.txt")
    env.configure("balance_export", path)
    rows = [l.split() for l in open(path) if l.startswith("vjbal1")]
    total = sum(int(r[16]) for r in rows)
    cand = sum(int(r[17]) for r in rows)
    done = sum(1 for r in rows if int(r[15]) == 3)
    third = len(ms_pf) // 3
    print(f"{tag}: {calls} calls, {len(rows)} workloads in the table, {done} finished their search; calls on a candidate split "
          f"{cand} of {total} = {100.0 * cand / max(total, 1):.1f} %; calls made while their workload was still searching "
          f"{100.0 * sum(1 for s in states if s == 1) / len(states):.1f} %; kernel ms per frame: whole run {np.mean(ms_pf):.4f}, last third "
          f"{np.mean(ms_pf[-third:]):.4f}", flush=True)
    return path


for tag, exact in (("exact frame count (round 3)", "1"), ("batch-size class", "0")):
    env = Environment(0)
    env.configure("balance_exact", exact)
    path = run(env, "exact" if exact == "1" else "class")
    env.close()
env2 = Environment(0)
env2.configure("balance_import", path)
run(env2, "class_imported")
env2.close()
