"""Per-step kernel time from two rocprofv3 --stats runs of the same command with different step counts:
python scripts/stats_diff.py <dir_small> <dir_large> <step_difference>"""
import csv
import glob
import sys


def load(d):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}


a, b, n = load(sys.argv[1]), load(sys.argv[2]), float(sys.argv[3])
rows = []
for k, (c1, t1) in b.items():
    c0, t0 = a.get(k, (0, 0.0))
    if c1 - c0 > 0:
        rows.append((t1 - t0, c1 - c0, k))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("GPU busy per step: %.3f ms, launches per step: %.1f" % (tot / n / 1e6, sum(r[1] for r in rows) / n))
for t, c, k in rows[:40]:
    print("%-100s calls/step %7.1f  avg_us %7.2f  us/step %8.1f  %5.1f%%" % (k[:100], c / n, t / c / 1e3, t / n / 1e3, 100 * t / tot))
