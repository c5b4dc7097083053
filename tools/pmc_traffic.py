"""Turns the rocprofv3 outputs of tools/profile_bench.sh into the committed summaries:
  profiles/rNN_kernel_stats.csv     (copy of the --kernel-trace --stats summary)
  profiles/rNN_pmc_traffic.json     per kernel: launches, FETCH_SIZE, WRITE_SIZE, HBM bytes per launch
  profiles/pmc_dominant.json        what bench.py reads for roofline.traffic
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE/WRITE_SIZE are in KiB and, on gfx950,
FETCH_SIZE reports half of the bytes of a wide coalesced read stream (MI355X_MICROARCH.md §HBM);
for other access shapes the factor is uncalibrated, so the figure is an upper estimate."""
import collections, csv, glob, json, os, shutil, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r01"
os.makedirs("profiles", exist_ok=True)

def per_kernel(d, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(f"{src}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = agg[r["Kernel_Name"]]
                a[0] += 1
                a[1] += float(r["Counter_Value"])
    return agg

for f in glob.glob(f"{src}/kt/*/*kernel_stats.csv"):
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
fetch, write = per_kernel("fetch", "FETCH_SIZE"), per_kernel("write", "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write)):
    n = max(fetch[k][0], write[k][0], 1)
    fk, wk = fetch[k][1] / max(fetch[k][0], 1), write[k][1] / max(write[k][0], 1)
    out[k] = {"launches_profiled": n, "FETCH_SIZE_KiB_per_launch": round(fk, 1), "WRITE_SIZE_KiB_per_launch": round(wk, 1),
              "hbm_bytes_per_launch": int((2 * fk + wk) * 1024)}
json.dump(out, open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
# dominant kernel = most total time in the kernel stats
stats = list(csv.DictReader(open(f"profiles/{tag}_kernel_stats.csv")))
stats.sort(key=lambda r: -float(r["TotalDurationNs"]))
dom = stats[0]["Name"]
kind = "tile" if "cascade_tile_pass" in dom else ("grid" if "cascade_pass<true" in dom else "queue")
key = next((k for k in out if k.startswith(dom.split("(")[0])), None)
json.dump({"kernel": dom, "kernel_kind": kind, "avg_ns": float(stats[0]["AverageNs"]),
           "hbm_bytes_per_launch": out[key]["hbm_bytes_per_launch"] if key else None,
           "source": f"profiles/{tag}_pmc_traffic.json"}, open("profiles/pmc_dominant.json", "w"), indent=1)
for r in stats[:8]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
print(json.load(open("profiles/pmc_dominant.json")))
