#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the COMPILED REFERENCE.

Runs only where /root/reference exists (this container): it executes
oracle/_ref/ref_driver, which oracle/Makefile builds from the unmodified
reference sources, and records for every case

  * the input as data: the CSC arrays of the reference's own example matrices
    (SLIP_LU/ExampleMats, data files its demos/tests hold) or the spec of the
    deterministic generator (slip_matgen.h),
  * the column order q the reference's SLIP_LU_analyze (COLAMD/AMD/none) chose,
  * the reference's result: full L/U/rho/pinv arrays for small cases, a SHA-256
    over the canonical arrays for large ones, plus the algorithmic counters of
    SURVEY.md 8(d) and the reference's own wall time.

Nothing of the reference's source travels; fixtures are inputs and outputs.
Usage:  python tests/golden/make_golden.py [case ...]
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import slabfile  # noqa: E402

REF = "/root/reference/SLIP_LU/ExampleMats"
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver")

# name: (input spec, pivot, order, tol, K, cap, keep_full_factor)
CASES = {
    # the reference's own example matrices, default options (COLAMD, TOL_SMALLEST, tol 1)
    "test_mat":        (f"trip:{REF}/test_mat.txt", 3, 1, 1.0, 0, 0, True),
    "10teams":         (f"trip:{REF}/10teams_mat.txt", 3, 1, 1.0, 0, 0, False),
    "prob159":         (f"trip:{REF}/prob159_mat.txt", 3, 1, 1.0, 0, 0, False),
    "NSR8K":           (f"trip:{REF}/NSR8K_mat.txt", 3, 1, 1.0, 0, 0, False),
    "NSR8K_w600":      (f"trip:{REF}/NSR8K_mat.txt", 3, 1, 1.0, 600, 0, False),
    # every pivoting scheme and ordering on small inputs
    "test_mat_p0":     (f"trip:{REF}/test_mat.txt", 0, 1, 1.0, 0, 0, True),
    "test_mat_p1":     (f"trip:{REF}/test_mat.txt", 1, 1, 1.0, 0, 0, True),
    "test_mat_p2":     (f"trip:{REF}/test_mat.txt", 2, 1, 1.0, 0, 0, True),
    "test_mat_p4":     (f"trip:{REF}/test_mat.txt", 4, 1, 1.0, 0, 0, True),
    "test_mat_p5":     (f"trip:{REF}/test_mat.txt", 5, 1, 1.0, 0, 0, True),
    "test_mat_noord":  (f"trip:{REF}/test_mat.txt", 3, 0, 1.0, 0, 0, True),
    "test_mat_amd":    (f"trip:{REF}/test_mat.txt", 3, 2, 1.0, 0, 0, True),
    "test_mat_tol01":  (f"trip:{REF}/test_mat.txt", 3, 1, 0.1, 0, 0, True),
    "test_mat_p4tol":  (f"trip:{REF}/test_mat.txt", 4, 1, 0.3, 0, 0, True),
    "10teams_p0":      (f"trip:{REF}/10teams_mat.txt", 0, 1, 1.0, 0, 0, False),
    "10teams_p5":      (f"trip:{REF}/10teams_mat.txt", 5, 1, 1.0, 0, 0, False),
    "10teams_tol":     (f"trip:{REF}/10teams_mat.txt", 3, 1, 0.01, 0, 0, False),
    "prob159_p1":      (f"trip:{REF}/prob159_mat.txt", 1, 1, 1.0, 0, 0, False),
    "prob159_p4tol":   (f"trip:{REF}/prob159_mat.txt", 4, 1, 0.5, 0, 0, False),
    "prob159_w200c8":  (f"trip:{REF}/prob159_mat.txt", 3, 1, 1.0, 0, 8, False),
    # LP bases shipped with the reference (complete runs spanning the limb classes)
    "rail4284":        (f"trip:{REF}/BasisLIB_ALL/RHS/rail4284.mat", 3, 1, 1.0, 0, 0, False),
    "fome12":          (f"trip:{REF}/BasisLIB_ALL/RHS/fome12.mat", 3, 1, 1.0, 0, 0, False),
    "rl5934":          (f"trip:{REF}/BasisLIB_ALL/RHS/rl5934.mat", 3, 1, 1.0, 0, 0, False),
    "d18512":          (f"trip:{REF}/BasisLIB_ALL/RHS/d18512.mat", 3, 1, 1.0, 0, 0, False),
    "model6":          (f"trip:{REF}/BasisLIB_ALL/RHS/model6.mat", 3, 1, 1.0, 0, 0, False),
    "de080285":        (f"trip:{REF}/BasisLIB_ALL/RHS/de080285.mat", 3, 1, 1.0, 0, 0, False),
    # synthetic matrices of the benchmark shapes (generator spec: n,density,bits,seed)
    "gen_n40":         ("gen:40,0.15,8,5", 3, 1, 1.0, 0, 0, True),
    "gen_n40_pm1":     ("gen:40,0.2,1,6", 3, 1, 1.0, 0, 0, True),
    "gen_n300":        ("gen:300,0.01,16,2", 3, 1, 1.0, 0, 0, False),
    "gen_n2000_pm1":   ("gen:2000,0.002,1,7", 3, 1, 1.0, 0, 0, False),
    "gen_n5000_c8":    ("gen:5000,0.0008,16,3", 3, 1, 1.0, 0, 8, False),
    "gen_n20000_c16":  ("gen:20000,0.001,16,1", 3, 1, 1.0, 0, 16, False),
    "C3_n50k_c32":     ("gen:50000,0.001,16,1", 3, 1, 1.0, 0, 32, False),
    "C4_n100k_c64":    ("gen:100000,0.001,16,1", 3, 1, 1.0, 0, 64, False),
    "C5_n200k_c64":    ("gen:200000,0.0005,16,1", 3, 1, 1.0, 0, 64, False),
}


def run_case(name):
    spec, pivot, order, tol, K, cap, full = CASES[name]
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "o.slab")
        subprocess.check_call([DRIVER, "window", spec, out, str(K), str(cap), str(pivot), str(order), repr(tol)])
        d = slabfile.load(out)
        entry = dict(name=name, input=spec if spec.startswith("gen:") else "slab", pivot=pivot, order=order,
                     tol=tol, kmax=K, cap=cap, n=int(d["n"][0]), K=int(d["K"][0]),
                     status=int(d["counters"][7]),
                     lnz=int(len(d.get("Li", []))), unz=int(len(d.get("Ui", []))),
                     counters={k: int(v) for k, v in zip(
                         ("N_upd", "B_read", "B_write", "N_src", "L_streamed", "maxlimbs", "K_done"), d["counters"][:7])},
                     ref_seconds=float(d["timing"][0]), ref_sym_seconds=float(d["timing"][1]),
                     digest=slabfile.factor_digest(d) if int(d["K"][0]) > 0 else None, full=full)
        fix = {"q": d["q"], "pinv": d["pinv"]}
        if not spec.startswith("gen:"):
            subprocess.check_call([DRIVER, "order", spec, os.path.join(td, "a.slab"), str(order)],
                                  stderr=subprocess.DEVNULL)
            a = slabfile.load(os.path.join(td, "a.slab"))
            assert np.array_equal(a["q"], d["q"])
            for k in ("Ap", "Ai", "Alen", "Alimbs"):
                fix[k] = a[k]
        if full:
            for k in slabfile.FACTOR_KEYS:
                fix[k] = d[k]
        else:
            fix["rholen"] = d["rholen"]
        slabfile.save(os.path.join(HERE, name + ".slab.gz"), fix)
        return entry


# rational solutions of SLIP_LU_solve for the deterministic right-hand side of ref_driver's `solve` mode
SOLVE_CASES = {
    "solve_test_mat": f"trip:{REF}/test_mat.txt",
    "solve_10teams":  f"trip:{REF}/10teams_mat.txt",
    "solve_gen_n40":  "gen:40,0.15,8,5",
}


def run_solve_cases():
    idx = []
    with tempfile.TemporaryDirectory() as td:
        for name, spec in SOLVE_CASES.items():
            out = os.path.join(td, "s.slab")
            subprocess.check_call([DRIVER, "solve", spec, out], stderr=subprocess.DEVNULL)
            d = slabfile.load(out)
            fix = {k: d[k] for k in ("q", "pinv", "xnumlen", "xnumlimbs", "xdenlen", "xdenlimbs")}
            if not spec.startswith("gen:"):
                subprocess.check_call([DRIVER, "order", spec, os.path.join(td, "a.slab")], stderr=subprocess.DEVNULL)
                a = slabfile.load(os.path.join(td, "a.slab"))
                for k in ("Ap", "Ai", "Alen", "Alimbs"):
                    fix[k] = a[k]
            slabfile.save(os.path.join(HERE, name + ".slab.gz"), fix)
            idx.append(dict(name=name, input=spec if spec.startswith("gen:") else "slab", n=int(d["n"][0])))
    json.dump(idx, open(os.path.join(HERE, "solve_index.json"), "w"), indent=1)


def main():
    if sys.argv[1:] == ["solve"]:
        run_solve_cases()
        return
    names = sys.argv[1:] or list(CASES)
    idx_path = os.path.join(HERE, "index.json")
    index = {}
    if os.path.exists(idx_path):
        index = {e["name"]: e for e in json.load(open(idx_path))}
    for nm in names:
        index[nm] = run_case(nm)
        print(nm, index[nm]["K"], index[nm]["lnz"] + index[nm]["unz"] - index[nm]["K"], index[nm]["digest"])
    json.dump([index[k] for k in CASES if k in index], open(idx_path, "w"), indent=1)


if __name__ == "__main__":
    main()
