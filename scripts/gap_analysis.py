"""Gaps between consecutive kernels of the timed region of bench.py from a rocprofv3 kernel trace (CSV): how much of a step's idle time is the dispatch gap between queued
launches, how much the bubble behind the host's one synchronisation per MPGP step.   usage: python scripts/gap_analysis.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed region: the longest run of launches whose pattern repeats the chain -- take everything between the first and last k_dc_final of the second half of the trace
names = [r["Kernel_Name"] for r in rows]
idx = [i for i, n in enumerate(names) if "k_dc_final" in n]
lo, hi = idx[len(idx) // 3], idx[-1]
gaps = collections.defaultdict(list)
for i in range(lo + 1, hi + 1):
    g = (int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"])) / 1e3
    prev, cur = names[i - 1].split("(")[0].split("<")[0].replace("void ", ""), names[i].split("(")[0].split("<")[0].replace("void ", "")
    gaps[(prev, cur)].append(g)
tot = sum(sum(v) for v in gaps.values())
n = sum(len(v) for v in gaps.values())
print("launches %d, idle between kernels %.1f us in total, %.2f us per launch" % (n, tot, tot / n))
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-22s -> %-22s  n %5d  mean %6.2f us  total %8.1f us" % (k[0][:22], k[1][:22], len(v), sum(v) / len(v), sum(v)))
