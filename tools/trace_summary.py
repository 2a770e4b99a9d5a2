"""Timeline summary of a rocprofv3 kernel trace (csv) of tools/trace_update.py: over the LAST update phase (the span of the last 40 adamw
launches) -- wall span per optimiser step, time with no kernel running, time with at least one GEMM kernel running, busy time by kernel.
usage: trace_summary.py <dir with *_kernel_trace.csv> [steps]"""
import csv, glob, os, sys, collections
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
opt = [i for i, r in enumerate(rows) if "adamw" in r[2] or "sgd_kernel" in r[2] or "optimizer_step" in r[2]]
first, last = opt[-steps - 1], opt[-1]           # from the end of the optimiser launch before the phase's second step ... keep whole steps
t0, t1 = rows[first][1], rows[last][1]
sel = [r for r in rows if r[0] >= t0 and r[1] <= t1]
span = (t1 - t0) / 1e3
def union(iv):
    iv = sorted(iv); tot = 0; cs, ce = None, None
    for s, e in iv:
        if cs is None: cs, ce = s, e
        elif s <= ce: ce = max(ce, e)
        else: tot += ce - cs; cs, ce = s, e
    if cs is not None: tot += ce - cs
    return tot / 1e3
is_gemm = lambda n: "gemm" in n
busy = union([(s, e) for s, e, n in sel])
gemm = union([(s, e) for s, e, n in sel if is_gemm(n)])
by = collections.defaultdict(lambda: [0, 0.0])
for s, e, n in sel:
    k = n.split("(")[0].split("<")[0].replace("void ", "").split("::")[-1]
    by[k][0] += 1; by[k][1] += (e - s) / 1e3
print(f"{f}\n{steps} optimiser steps: span {span / steps:.1f} us per step; some kernel running {busy / steps:.1f} us ({100 * busy / span:.1f} %), "
      f"a GEMM kernel running {gemm / steps:.1f} us ({100 * gemm / span:.1f} %); sum of kernel durations {sum(v[1] for v in by.values()) / steps:.1f} us per step")
for k, (n, us) in sorted(by.items(), key=lambda kv: -kv[1][1])[:18]:
    print(f"  {k:40s} x{n / steps:6.1f} per step {us / steps:9.1f} us per step  avg {us / n:8.1f} us")
