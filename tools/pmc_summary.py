"""rocprofv3 output directories -> per-kernel summary (avg duration from the kernel trace, median counter values from the PMC
passes) as JSON.  usage: pmc_summary.py <out.json> <label>=<dir> [<label>=<dir> ...]
Each <dir> holds the csv files of ONE rocprofv3 run (--kernel-trace --stats, or --pmc <COUNTER> --kernel-trace)."""
import csv, glob, json, os, statistics, sys

out, runs = sys.argv[1], dict(a.split("=", 1) for a in sys.argv[2:])
res = {}
for label, d in runs.items():
    entry = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        dur = {}
        for r in csv.DictReader(open(f)):
            dur.setdefault(r["Kernel_Name"], []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for k, v in dur.items():
            entry.setdefault(k, {})["calls"] = len(v)
            entry[k]["avg_us"] = sum(v) / len(v) / 1e3
            entry[k]["median_us"] = statistics.median(v) / 1e3
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        vals = {}
        for r in csv.DictReader(open(f)):
            vals.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (k, c), v in vals.items():
            entry.setdefault(k, {})[c + "_median"] = statistics.median(v)
            entry[k][c + "_sum"] = sum(v)
            entry[k][c + "_dispatches"] = len(v)
    res[label] = {k: v for k, v in entry.items() if v.get("calls", 0) >= 3 or any(x.endswith("_median") for x in v)}
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
print("wrote", out, {k: len(v) for k, v in res.items()})
