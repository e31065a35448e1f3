#!/usr/bin/env python3
"""Transcribes the reference's golden test outputs for the QPS hot path into JSON fixtures.

Run in the build container only (needs /root/reference); the JSON files it writes are committed and
are what the tests read.  The fixtures are DATA (expected iteration counts, KKT residuals, monitor
traces) taken from src/tutorials/output/*.out and src/tutorials/feti/output/*.out together with the
test arguments from the /*TEST*/ blocks of the tutorials.
"""
import json
import os
import re

REF = "/root/reference/src/tutorials"
OUT = os.path.dirname(os.path.abspath(__file__))

RE_SOLVE = re.compile(r"last QPSSolve (CONVERGED|DIVERGED) due to (\w+), KSPReason=(-?\d+), required (\d+) iterations")
RE_NUM = re.compile(r"number of (Hessian multiplications|CG steps|expansion steps|proportioning steps) (\d+)")
RE_KKT = re.compile(r"r = (.*?)\s*= (\S+)\s+rO?/\|\|b\|\| = (\S+)")
RE_MON = re.compile(r"\s*(\d+) MPGP \[(.)\] \|\|gp\|\|=(\S+),\s+\|\|gf\|\|=(\S+),\s+\|\|gc\|\|=(\S+),\s+alpha=(\S+)")
RE_INNER = re.compile(r"Total number of inner iterations (\d+)")

KEYS = {"Hessian multiplications": "nmv", "CG steps": "ncg", "expansion steps": "nexp", "proportioning steps": "nprop"}


def parse(path):
    d = {"source": os.path.relpath(path, "/root/reference"), "solves": [], "kkt": [], "trace": []}
    d["text"] = [ln.rstrip("\n") for ln in open(path)]  # the expected output itself (what the reference's harness diffs against)
    for line in open(path):
        m = RE_SOLVE.search(line)
        if m:
            d["solves"].append({"converged": m.group(1) == "CONVERGED", "reason_name": m.group(2), "reason": int(m.group(3)), "iterations": int(m.group(4))})
            continue
        m = RE_NUM.search(line)
        if m:
            d["solves"][-1][KEYS[m.group(1)]] = int(m.group(2))
            continue
        m = RE_INNER.search(line)
        if m:
            d["solves"][-1]["inner_iterations"] = int(m.group(1))
            continue
        m = RE_KKT.match(line)
        if m:
            d["kkt"].append({"name": m.group(1).strip(), "r": m.group(2), "r_rel": m.group(3)})
            continue
        m = RE_MON.match(line)
        if m:
            d["trace"].append({"it": int(m.group(1)), "step": m.group(2), "gp": m.group(3), "gf": m.group(4), "gc": m.group(5), "alpha": m.group(6)})
    if not d["trace"]:
        del d["trace"]
    return d


CASES = {
    # name: (file, generator args / solver options from the /*TEST*/ block)
    "ex1_1": ("output/ex1_1.out", {"problem": "ex1", "n": 100, "opts": {}}),
    "ex1_opt": ("output/ex1_opt.out", {"problem": "ex1", "n": 100, "opts": {"exptype": "gf", "explengthtype": "opt"}}),
    "ex1_optapprox": ("output/ex1_optapprox.out", {"problem": "ex1", "n": 100, "opts": {"exptype": "g", "explengthtype": "optapprox"}}),
    "ex1_bb": ("output/ex1_bb.out", {"problem": "ex1", "n": 100, "opts": {"exptype": "gfgr", "explengthtype": "bb"}}),
    "ex1_projcg": ("output/ex1_projcg.out", {"problem": "ex1", "n": 100, "opts": {"exptype": "projcg"}}),
    "ex2_1_infinite-false": ("output/ex2_1_infinite-false.out", {"problem": "ex2", "n": 100, "infinite": False, "opts": {}}),
    "ex2_1_infinite-true": ("output/ex2_1_infinite-true.out", {"problem": "ex2", "n": 100, "infinite": True, "opts": {}}),
    "ex3_1": ("output/ex3_1.out", {"problem": "ex3", "n": 100, "opts": {}}),
    "ex3_nullspace": ("output/ex3_nullspace.out", {"problem": "ex3", "n": 100, "empty_nullsp": True, "opts": {}}),
    "jbearing2_4": ("output/jbearing2_4.out", {"problem": "jbearing2", "mx": 8, "my": 12, "opts": {"rtol": 1e-6, "atol": 1e-8}}),
    "jbearing2_5": ("output/jbearing2_5.out", {"problem": "jbearing2", "mx": 10, "my": 16, "opts": {"rtol": 1e-6, "atol": 1e-8}}),
    "jbearing2_6": ("output/jbearing2_6.out", {"problem": "jbearing2", "mx": 30, "my": 30, "opts": {"rtol": 1e-6, "atol": 1e-8}}),
    "feti_ex1_1": ("feti/output/ex1_1.out", {"problem": "feti_ex1"}),
    "feti_ex1_2": ("feti/output/ex1_2.out", {"problem": "feti_ex1"}),
    "feti_ex1_smalxe_orth_gs": ("feti/output/ex1_smalxe_orth_dual_qp_E_orth_type-gs.out", {"problem": "feti_ex1"}),
    "feti_ex1_smalxe_orth_implicit": ("feti/output/ex1_smalxe_orth_dual_qp_E_orth_type-implicit.out", {"problem": "feti_ex1"}),
    "feti_ex71_1_full": ("feti/output/ex71_1_feti_gluing_type-full.out", {"problem": "feti_ex71"}),
    "feti_ex71_1_nonred": ("feti/output/ex71_1_feti_gluing_type-nonred.out", {"problem": "feti_ex71"}),
    "feti_ex71_1_orth": ("feti/output/ex71_1_feti_gluing_type-orth.out", {"problem": "feti_ex71"}),
    "feti_ex71_2_lumped": ("feti/output/ex71_2_dual_pc_dual_type-lumped.out", {"problem": "feti_ex71"}),
    "feti_ex71_2_none": ("feti/output/ex71_2_dual_pc_dual_type-none.out", {"problem": "feti_ex71"}),
}

if __name__ == "__main__":
    allc = {}
    for name, (f, args) in CASES.items():
        d = parse(os.path.join(REF, f))
        d["args"] = args
        allc[name] = d
    with open(os.path.join(OUT, "reference_goldens.json"), "w") as fh:
        json.dump(allc, fh, indent=1, sort_keys=True)
    print("wrote", len(allc), "cases")
