#!/bin/bash
# Diagnostic: per-chunk phase timings (shader-clock stamps) of the orbit GEMM's workgroups, from a -DFXO_TRACE build of fshared.hip made on the GPU box only.
set -o pipefail
rm -f permon_amd/csrc/fshared.o
make -C permon_amd/csrc -j8 -s all EXTRA=-DFXO_TRACE > gpurun_out/trace_build.log 2>&1 || { tail -5 gpurun_out/trace_build.log; exit 1; }
env ${TRACE_ENV} python bench.py --steps 216 --warmup 8 --no-c2 --no-iterative --no-cpu-baseline --no-dual-spmv --details gpurun_out/trace_details.json > gpurun_out/trace.line 2> gpurun_out/trace.err
grep -A48 "FXO_TRACE workgroup" gpurun_out/trace.err | head -120 > gpurun_out/fxo_trace.txt
python - <<P
import re,collections
rows=[]
for ln in open("gpurun_out/trace.err"):
    m=re.match(r"\s+chunk\s+(\d+):\s+(\d+) \|\s+(\d+) \|\s+(\d+) \|\s+(\d+) \|\s+(\d+) \|\s+(\d+)",ln)
    if m: rows.append([int(v) for v in m.groups()])
import numpy as np
a=np.array(rows)
if len(a):
    a=a[a[:,0]>1]
    print("chunks traced %d; mean cycles: loads issued %.0f | products %.0f | wait vmcnt %.0f | LDS store %.0f | barrier %.0f | total %.0f" % ((len(a),)+tuple(a[:,1:].mean(axis=0))))
    print("median: ", np.median(a[:,1:],axis=0))
P
