"""Per-step kernel table from a rocprofv3 --stats kernel_stats.csv of the timed region of bench.py and the bench line of the same run.
usage: python scripts/per_step.py kernel_stats.csv bench.json"""
import csv
import json
import sys

d = json.load(open(sys.argv[2]))
steps = d["steps"]
rows = [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(sys.argv[1]))]
tot = sum(t for _, _, t in rows)
print("# %d timed steps, %.3f ms/step wall under the profiler, step mix %s" % (steps, d["ms_per_step"], json.dumps(d["config"]["steps_by_type"])))
print("GPU busy per step: %.3f ms, launches per step: %.1f" % (tot / steps / 1e6, sum(c for _, c, _ in rows) / steps))
for n, c, t in sorted(rows, key=lambda r: -r[2]):
    print("%-100s calls/step %7.2f  avg_us %8.2f  us/step %8.1f  %5.1f%%" % (n[:100], c / steps, t / c / 1e3, t / steps / 1e3, 100 * t / tot))
