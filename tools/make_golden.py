#!/usr/bin/env python3
"""Generate the committed golden fixtures under tests/golden/ (run here, on the CPU).

1. baseline_probes.json -- the reference outputs recorded in BASELINE.md section 2 (objective values to 17
   digits, f / grad callback counts measured on the compiled reference during the survey), with the exact
   inputs that produced them.  These are DATA transcribed from BASELINE.md, not reference source.
2. gpu_parity_vectors.npz -- for fixed seeded inputs, the oracle's results in the kernels' summation order
   (FLO_SUM_TREE with the geometry recorded alongside): what `pytest -m gpu` must reproduce BIT FOR BIT without
   the oracle being involved at run time.  The oracle itself is pinned by (1).
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O

GOLD = os.path.join(ROOT, "tests", "golden")
GEO = {10: (64, 2), 96: (64, 2), 256: (64, 4), 512: (64, 8), 1024: (128, 8)}  # fl_reduction_geometry
GEO_CG_1024 = (64, 16)  # fl_reduction_geometry_for(FL_SOLVER_CG, 1024): the fused SD / CG kernels, 512 < n <= 1024


def probes():
    return {
        "source": "BASELINE.md section 2 (oracle probe numbers measured on the compiled reference, survey container)",
        "note": "nf in the probes includes one extra f call made by the probe driver after the solve to print f",
        "cases": [
            {"name": "lbfgs_quartic_n10", "solver": "LBFGS", "objective": "quartic", "n": 10, "x0": "0.1*i", "f": 1.3759308360776471e-21, "norm_x": 8.50e-06},
            {"name": "bfgs0_quartic_n10", "solver": "BFGS", "exact_step": 0, "objective": "quartic", "n": 10, "x0": "0.1*i", "f": 1.0336918948999602e-21, "norm_x": 9.16e-06},
            {"name": "cgdy_quartic_n10", "solver": "CG", "objective": "quartic", "n": 10, "x0": "0.1*i", "f": 1.0566146259484697e-22, "norm_x": 5.00e-06},
            {"name": "bfgs0_rosen_n10", "solver": "BFGS", "exact_step": 0, "objective": "rosenbrock", "n": 10, "x0": "(-1.2,1,...)", "f": 0.0, "nf_plus_1": 588, "ng": 446},
            {"name": "cgdy_rosen_n10", "solver": "CG", "objective": "rosenbrock", "n": 10, "x0": "(-1.2,1,...)", "f": 1.0269890190168409},
            {"name": "lbfgs_rosen_n256_maxit1000", "solver": "LBFGS", "objective": "rosenbrock", "n": 256, "x0": "(-1.2,1,...)", "f": 4.2659329580565036e+01, "nf_plus_1": 7273, "ng": 6706},
            {"name": "lbfgs_rosen_n256_maxit3000", "solver": "LBFGS", "objective": "rosenbrock", "n": 256, "x0": "(-1.2,1,...)", "maxit": 3000, "f_approx": 1.38e-28, "nf_plus_1": 9504, "ng": 8860},
            {"name": "lbfgs_rosen_n256_near", "solver": "LBFGS", "objective": "rosenbrock", "n": 256, "x0": "1+0.1*sin(i)", "f_approx": 5.27e-28, "nf_plus_1": 1030, "ng": 924},
            {"name": "cgdy_quad_n1024_k1000", "solver": "CG", "objective": "diagquad", "n": 1024, "kappa": 1000, "b": "sin(i)", "x0": "0", "f": -2.0426388375377749, "nf_plus_1": 1967, "ng": 1465},
            {"name": "lbfgs_quad_n1024_k1000", "solver": "LBFGS", "objective": "diagquad", "n": 1024, "kappa": 1000, "b": "sin(i)", "x0": "0", "f": -2.0426388375378717, "nf_plus_1": 5384, "ng": 3386},
            {"name": "cgdy_quad_n1024_k10", "solver": "CG", "objective": "diagquad", "n": 1024, "kappa": 10, "b": "sin(i)", "x0": "0", "f": -65.685441991456955, "nf_plus_1": 1478, "ng": 668},
            {"name": "cgpr_quad_n1024_k10", "solver": "CG", "method": "PR", "objective": "diagquad", "n": 1024, "kappa": 10, "b": "sin(i)", "x0": "0", "f": -65.685441991457367, "nf_plus_1": 1375, "ng": 498},
            {"name": "lbfgs_quad_n1024_k10", "solver": "LBFGS", "objective": "diagquad", "n": 1024, "kappa": 10, "b": "sin(i)", "x0": "0", "f": -65.685441991457296, "nf_plus_1": 729, "ng": 214},
            {"name": "auglag_lbfgs_n512_m8", "solver": "AugmentedLagrangian+LBFGS", "objective": "diagquad", "n": 512, "m": 8, "kappa": 10, "b": "sin(i)", "x0": "0.1+0.05*cos(i)", "precision": 1e-10, "f": -23.331108193268726, "cnorm": 1.9e-11, "nf_plus_1": 143674, "ng": 3183},
        ],
    }


def vectors():
    out = {}
    rng = np.random.default_rng(20261003)

    def put(name, r, extra=None):
        for k in ("x", "f", "gg", "iters", "status", "nf", "ng"):
            if k in r:
                out[f"{name}/{k}"] = r[k]
        for k, v in (extra or {}).items():
            out[f"{name}/{k}"] = v

    # quartic dim 10 (reference test shape) -- every solver
    x0 = np.vstack([0.1 * np.arange(1, 11), rng.random((3, 10))])
    T, E = GEO[10]
    for nm, solver, o, ffd, form in (("lbfgs", O.LBFGS, O.defaults(), 0, 0), ("cgdy", O.CG, O.defaults(c2=0.45), 0, 0),
                                     ("cgpr", O.CG, O.defaults(c2=0.45, method=1), 0, 0),
                                     ("sd", O.SD, O.defaults(maxit=200), 0, 0),
                                     ("bfgs", O.BFGS, O.defaults(exact_step=0), 0, 1),
                                     ("lbfgs_ffd", O.LBFGS, O.defaults(), 1, 0),
                                     ("lbfgs_wolfe", O.LBFGS, O.defaults(strong=0), 0, 0)):
        put(f"quartic10_{nm}", O.solve_batch(solver, O.QUARTIC, x0, opts=o, use_ffd=bool(ffd), bfgs_form=form,
                                             sum_mode=O.TREE, threads=T, ept=E), {"x0": x0})
    # config 2 shape: Rosenbrock n=256, x0 = 1 + 0.1 u
    x0 = 1.0 + 0.1 * rng.uniform(-1, 1, (6, 256))
    T, E = GEO[256]
    put("rosen256_lbfgs", O.solve_batch(O.LBFGS, O.ROSENBROCK, x0, opts=O.defaults(precision=1e-10, maxit=3000),
                                        sum_mode=O.TREE, threads=T, ept=E), {"x0": x0})
    # config 3 / headline shape: diagonal quadratics n=1024
    n = 1024
    kappa = np.exp(rng.uniform(np.log(10), np.log(1000), 4))
    d = 1.0 + (kappa[:, None] - 1.0) * (np.arange(n) / (n - 1))[None, :]
    b = rng.uniform(-1, 1, (4, n))
    for nm, solver, o in (("lbfgs", O.LBFGS, O.defaults(precision=1e-6)), ("cgdy", O.CG, O.defaults(c2=0.45, precision=1e-6))):
        T, E = GEO_CG_1024 if solver == O.CG else GEO[1024]
        put(f"quad1024_{nm}", O.solve_batch(solver, O.DIAGQUAD, np.zeros((4, n)), d=d, b=b, opts=o, sum_mode=O.TREE,
                                            threads=T, ept=E), {"d": d, "b": b})
    # config 5 shape: augmented Lagrangian n=512, M=8
    n, m = 512, 8
    i = np.arange(1, n + 1).astype(float)
    d = (1 + 9 * (i - 1) / (n - 1))[None, :].repeat(2, 0)
    b = np.sin(i)[None, :].repeat(2, 0)
    x0 = (0.1 + 0.05 * np.cos(i))[None, :] + 0.01 * rng.standard_normal((2, n))
    T, E = GEO[512]
    r = O.auglag_batch(O.LBFGS, O.DIAGQUAD, x0, m, d=d, b=b, opts=O.defaults(precision=1e-10), sum_mode=O.TREE, threads=T, ept=E)
    put("auglag512", {"x": r["x"], "iters": r["iters"], "nf": r["nf"], "ng": r["ng"]},
        {"x0": x0, "d": d, "b": b, "lam": r["lam"], "outer": r["outer"], "cnorm2": r["cnorm2"]})
    out["geometry"] = np.array([[n_, t, e] for n_, (t, e) in GEO.items()])
    return out


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    json.dump(probes(), open(os.path.join(GOLD, "baseline_probes.json"), "w"), indent=1)
    np.savez_compressed(os.path.join(GOLD, "gpu_parity_vectors.npz"), **vectors())
    print("wrote", os.listdir(GOLD))
