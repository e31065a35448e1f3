"""The FETI dual SpMV block of bench.py alone (MatMult_BlockDiag on 8 DISTINCT K_i of 43^3 elements: CSR kernel, then the 3x3-block kernel with a device copy per block) --
the program the PMC passes of scripts/gpu_final_r04.sh profile (FETCH_SIZE / WRITE_SIZE of exactly these launches).  usage: python scripts/dual_spmv_only.py [nel]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import permon_amd as pa  # noqa: E402

nel = int(sys.argv[1]) if len(sys.argv) > 1 else 43
ctx = pa.Context(0)
f = pa.CubeFeti((2, 2, 2), nel, contact=True)
print(json.dumps(bench.dual_spmv_hbm(ctx, f)))
ctx.close()
