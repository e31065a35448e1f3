"""Reduce rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes to per-launch HBM traffic per kernel.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports half the bytes of a wide coalesced stream
-> doubled; WRITE_SIZE is exact.  Both counters are in KiB."""
import csv, glob, json, os, sys, collections
root = sys.argv[1]
out = {}
for C in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "pmc_" + C, "**", "*counter_collection.csv"), recursive=True)
    vals = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != C:
                continue
            vals[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    out[C] = {}
    for k, v in vals.items():
        # launches of a halted speculative chain return at once (no traffic): not counted
        real = [x for x in v if x >= 0.25 * max(v)] if max(v) > 0 else v
        out[C][k] = {"launches": len(real), "avg_KiB": sum(real) / len(real)}
res = {}
for k in out["FETCH_SIZE"]:
    f = out["FETCH_SIZE"][k]["avg_KiB"] * 1024 * 2.0  # gfx950: x2
    w = out["WRITE_SIZE"].get(k, {"avg_KiB": 0})["avg_KiB"] * 1024
    res[k] = {"launches": out["FETCH_SIZE"][k]["launches"], "fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes_per_launch": f + w}
if len(sys.argv) > 2:  # provenance: the commit measured and the command (bench.py quotes it as roofline.traffic_source)
    res["_meta"] = {"git": sys.argv[2], "command": sys.argv[3] if len(sys.argv) > 3 else ""}
json.dump(res, open(os.path.join(root, "pmc_traffic.json"), "w"), indent=1)
for k, v in sorted(((k, v) for k, v in res.items() if k != "_meta"), key=lambda kv: -kv[1]["hbm_bytes_per_launch"])[:8]:
    print("%-60s launches=%5d  fetch(x2)=%8.1f MB  write=%8.1f MB  total=%8.1f MB" % (k[:60], v["launches"], v["fetch_bytes_corrected"] / 1e6, v["write_bytes"] / 1e6, v["hbm_bytes_per_launch"] / 1e6))
