"""Turns the rocprofv3 outputs of tools/profile_bench.sh into the committed summaries:
  profiles/rNN_kernel_stats.csv     (copy of the --kernel-trace --stats summary)
  profiles/rNN_pmc_traffic.json     per kernel: launches, FETCH_SIZE, WRITE_SIZE, HBM bytes per launch, SQ / LDS counters
  profiles/pmc_dominant.json        what bench.py reads for roofline.traffic — stamped with the hash of the kernel sources
                                    it was collected from, so that bench.py can refuse a stale one
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950,
FETCH_SIZE reports half of the bytes of a wide coalesced read stream (MI355X_MICROARCH.md §HBM);
for other access shapes the factor is uncalibrated, so the figure is an upper estimate."""
import collections, csv, glob, json, os, shutil, sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_source_hash

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r04"
os.makedirs("profiles", exist_ok=True)

def per_kernel(d, counters):
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(f"{src}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] in counters:
                a = agg[r["Kernel_Name"]][r["Counter_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return agg

for f in glob.glob(f"{src}/kt/*/*kernel_stats.csv"):
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
for cfg in ("config2", "config4", "config5", "cv", "cvtree"):      # the other configs' kernel-trace summaries (tools/configs.py under rocprofv3)
    for f in glob.glob(f"{src}/kt_{cfg}/*/*kernel_stats.csv"):
        shutil.copy(f, f"profiles/{tag}_{cfg}_kernel_stats.csv")
fetch, write = per_kernel("fetch", {"FETCH_SIZE"}), per_kernel("write", {"WRITE_SIZE"})
SQ = ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES",
      "SQ_WAIT_ANY", "SQ_INSTS_VALU")
sq = per_kernel("sq", set(SQ))
out = {}
for k in sorted(set(fetch) | set(write) | set(sq)):
    fn, fv = fetch[k]["FETCH_SIZE"]
    wn, wv = write[k]["WRITE_SIZE"]
    fk, wk = fv / max(fn, 1), wv / max(wn, 1)
    e = {"launches_profiled": max(fn, wn, 1), "FETCH_SIZE_KiB_per_launch": round(fk, 1), "WRITE_SIZE_KiB_per_launch": round(wk, 1),
         "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
    if k in sq:
        c = {n: sq[k][n][1] / max(sq[k][n][0], 1) for n in SQ}
        e["sq_per_launch"] = {n: round(v) for n, v in c.items()}
        if c["SQ_LDS_IDX_ACTIVE"] > 0:
            e["lds_bank_conflict_share"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 3)
        if c["SQ_WAVE_CYCLES"] > 0:
            e["wave_wait_share"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 3)
        if c["SQ_BUSY_CYCLES"] > 0:   # SQ_BUSY_CYCLES sums the busy cycles of the shader engines; the ratios are relative measures only
            e["valu_active_per_busy"] = round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"], 3)
            e["lds_idx_active_per_busy"] = round(c["SQ_LDS_IDX_ACTIVE"] / c["SQ_BUSY_CYCLES"], 3)
    out[k] = e
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
# dominant kernel = most total time in the kernel stats
stats = [r for r in csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv"))
         if "cv_profile_pass" not in r["Name"] and "cv_tile_pass" not in r["Name"]]      # the OpenCV profile is a second path, measured in the same bench run
stats.sort(key=lambda r: -float(r["TotalDurationNs"]))
dom = stats[0]["Name"]
kind = "tile" if "cascade_tile_pass" in dom else ("grid" if "cascade_pass<true" in dom else "queue")
key = next((k for k in out if k.startswith(dom.split("(")[0])), None)
d = {"kernel": dom, "kernel_kind": kind, "avg_ns": float(stats[0]["AverageNs"]),
     "hbm_bytes_per_launch": out[key]["hbm_bytes_per_launch"] if key else None,
     "source": f"profiles/{tag}_pmc_traffic.json", "kernel_source_hash": kernel_source_hash()}
if key:
    for n in ("lds_bank_conflict_share", "wave_wait_share"):
        if n in out[key]:
            d[n] = out[key][n]
json.dump(d, open("profiles/pmc_dominant.json", "w"), indent=1)
for r in stats[:8]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
print(json.load(open("profiles/pmc_dominant.json")))
