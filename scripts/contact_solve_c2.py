"""The whole configs[2] contact TFETI solve in ONE library call (pmh_feti_contact_solve: hierarchy, explicit operators, SMALXE + MPGP, rigid-body
recovery), set-up included: explicit class-shared operator assembled by symmetry vs assembled row by row vs the inner-Krylov K^+.
usage: python scripts/contact_solve_c2.py [nel=43] [all]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import permon_amd as pa  # noqa: E402

nel = int(sys.argv[1]) if len(sys.argv) > 1 else 43
ctx = pa.Context(0)
f = pa.CubeFeti((2, 2, 2), nel, contact=True)
cases = [("explicit class_orbit (representatives' rows, GEMM apply)", dict(explicit=True, explicit_storage="class_orbit", explicit_symmetry=True)),
         ("explicit class_sym, set-up by symmetry", dict(explicit=True, explicit_storage="class_sym", explicit_symmetry=True)),
         ("inner-Krylov K^+ (fp16 V-cycle PC)", dict(explicit=False))]
if len(sys.argv) > 2:
    cases.append(("explicit class_sym, one solve per row", dict(explicit=True, explicit_storage="class_sym", explicit_symmetry=False)))
for name, kw in cases:
    t = time.perf_counter()
    u, lam, st = pa.FETIContactSolve(ctx, f, **kw)
    wall = time.perf_counter() - t
    s = st.smalxe
    print(json.dumps({"case": name, "nel": nel, "N": f.N, "n_lambda": f.n_lambda, "wall_s_incl_upload": round(wall, 3), "setup_s": round(st.setup_seconds, 3), "explicit_assembly_s": round(st.explicit_seconds, 3),
                      "explicit_solves": st.explicit_solves, "symmetries": st.explicit_symmetries, "solve_s": round(st.solve_seconds, 3), "outer": s.iteration, "inner": s.inner_iter_accu, "hessian_mults": s.inner.nmv,
                      "cg": s.inner.ncg, "expansion": s.inner.nexp, "active_contact_rows": st.n_active, "norm_Glambda_minus_e": st.norm_Glambda_minus_e}), flush=True)
ctx.close()
