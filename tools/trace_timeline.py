"""One optimiser step of a rocprofv3 kernel trace (csv) of tools/trace_update.py as a text timeline: start (us from the step's first
launch), duration, queue, kernel -- and the gaps in which no GEMM kernel runs.  usage: trace_timeline.py <dir> [step index from the end]"""
import csv, glob, os, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
rows.sort()
opt = [i for i, r in enumerate(rows) if "adamw" in r[2] or "optimizer_step" in r[2]]
a, b = opt[-back - 1], opt[-back]
t0 = rows[a][1]
sel = [r for r in rows[a + 1:b + 1]]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("addhip_dma::", "").split("(")[0][:70]
last_gemm_end = t0
for s, e, n, q in sel:
    g = "gemm" in n
    gap = ""
    if g:
        if s > last_gemm_end: gap = f"   <-- no GEMM for {(s - last_gemm_end) / 1e3:.1f} us"
        last_gemm_end = max(last_gemm_end, e)
    print(f"{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:7.1f}  q{q:>3} {short(n)}{gap}")
print(f"step span {(rows[b][1] - t0) / 1e3:.1f} us")
