"""PMH_CONTACT_TIMING=1: the set-up stages of pmh_feti_contact_solve at the configs[2] size (2 x 2 x 2 cubes of 43^3 nodes), twice (the first call pays the code-object loads)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PMH_CONTACT_TIMING"] = "1"
import numpy as np
import permon_amd as pa

nel = int(sys.argv[1]) if len(sys.argv) > 1 else 42
ctx = pa.Context(0)
t0 = time.time()
f = pa.CubeFeti((2, 2, 2), nel, contact=True)
print("generator %.2f s, N = %d" % (time.time() - t0, f.N), flush=True)
for rep in range(2):
    t0 = time.time()
    u, lam, st = pa.FETIContactSolve(ctx, f, explicit=True, explicit_storage="class_orbit", explicit_symmetry=True)
    print("call %d: wall %.3f s, setup %.3f s (explicit %.3f s, %d solves), solve %.4f s, outer %d inner %d" % (rep, time.time() - t0, st.setup_seconds, st.explicit_seconds, st.explicit_solves, st.solve_seconds,
          st.smalxe.iteration, st.smalxe.inner_iter_accu), flush=True)
