#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle.

The reference itself cannot be built or imported here (it needs ROOT), and it
ships no golden vectors, so these fixtures freeze the ORACLE's outputs (parity
unpinned, see oracle/oracle_core.h): the CPU suite checks that the oracle still
reproduces them, the GPU suite checks the HIP path against them without the
oracle in the loop.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SEED = 20240607


def frozen_chains(dim, kind, nchains, steps, x0):
    out = {"dim": dim, "kind": kind, "steps": steps, "seed": SEED, "x0": x0}
    acc, logl, sigma, xs = [], [], [], []
    for ch in range(nchains):
        c = O.Chain(dim, kind=kind, seed=SEED, chain_id=ch)
        c.set_covariance_frozen(1)
        assert c.start(x0[:, ch] if x0.ndim == 2 else x0)
        r = c.run(steps)
        acc.append(r["accepted"]); logl.append(r["logl_accepted"]); sigma.append(r["sigma"]); xs.append(c.accepted)
    out.update(accepted=np.array(acc), logl=np.array(logl), sigma=np.array(sigma), x=np.array(xs).T)
    return out


def pooled(dim, kind, nchains, window, nwin):
    e = O.Ensemble(nchains, dim, kind=kind, seed=SEED, mode=O.MODE_POOLED)
    assert e.start(np.zeros(dim))
    moments = None
    for _ in range(nwin):
        e.step(window)
        moments = e.reduce_moments()
        e.apply_moments(moments)
    e.step(4)
    return {"dim": dim, "kind": kind, "nchains": nchains, "window": window, "nwin": nwin, "seed": SEED,
            "x": e.x, "logl": e.lane("logl"), "sigma": e.lane("sigma"), "naccept": e.lane("naccept"),
            "last_moments": moments, "covariance": e.covariance, "center": e.center,
            "decomposition": e.decomposition}


def frozen_ensemble(dim, kind, nchains, steps, exact, x0):
    """Large dimensions: end state of a FROZEN ensemble in the reference or the fused order."""
    prm = [100.0] if kind == O.LIKE_ROSENBROCK else None
    e = O.Ensemble(nchains, dim, kind=kind, params=prm, seed=SEED, mode=O.MODE_FROZEN, exact=exact)
    assert e.start(x0)
    e.step(steps)
    return {"dim": dim, "kind": kind, "nchains": nchains, "steps": steps, "exact": int(exact), "seed": SEED, "x0": x0,
            "x": e.x, "logl": e.lane("logl"), "sigma": e.lane("sigma"), "naccept": e.lane("naccept"),
            "step_rms": e.lane("step_rms")}


def spd(dim, seed):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((dim, dim)) / np.sqrt(dim)
    return a @ a.T + np.eye(dim)


def hmc_chains(dim, nchains, steps, leap, eps, alpha, fused):
    """TSimpleHMC with the quadratic-form likelihood, fixed step and leapfrog count."""
    err = spd(dim, dim)
    q, m, pot, acc = [], [], [], []
    for ch in range(nchains):
        h = O.Hmc(dim, kind=O.LIKE_QUADFORM, params=err, seed=SEED, chain_id=ch, potential_from_gradient=True,
                  fused_gradient=fused)
        h.set_alpha(alpha)
        h.start(np.ones(dim))
        h.set_mean_epsilon(-eps)
        h.set_leapfrog(leap)
        h.run(steps)
        q.append(h.accepted); m.append(h.momentum); pot.append(h.scalars["accepted_potential"])
        acc.append(h.scalars["current_acceptance"])
    return {"dim": dim, "nchains": nchains, "steps": steps, "leapfrog": leap, "epsilon": eps, "alpha": alpha,
            "fused": int(fused), "seed": SEED, "error": err, "q": np.array(q).T, "momentum": np.array(m).T,
            "potential": np.array(pot), "acceptance": np.array(acc)}


def main():
    O.build()
    np.savez(os.path.join(HERE, "frozen_iso_d5.npz"), **frozen_chains(5, O.LIKE_ISO, 4, 300, np.zeros(5)))
    rng = np.random.default_rng(1)
    np.savez(os.path.join(HERE, "frozen_rosenbrock_d6.npz"),
             **frozen_chains(6, O.LIKE_ROSENBROCK, 4, 300, rng.uniform(0.5, 1.5, (6, 4))))
    np.savez(os.path.join(HERE, "frozen_iso_d50.npz"), **frozen_chains(50, O.LIKE_ISO, 3, 60, np.zeros(50)))
    np.savez(os.path.join(HERE, "pooled_iso_d5.npz"), **pooled(5, O.LIKE_ISO, 70, 16, 3))
    np.savez(os.path.join(HERE, "pooled_iso_d20.npz"), **pooled(20, O.LIKE_ISO, 130, 8, 3))
    for exact in (True, False):
        tag = "reference" if exact else "fused"
        np.savez(os.path.join(HERE, f"frozen_iso_d200_{tag}.npz"),
                 **frozen_ensemble(200, O.LIKE_ISO, 40, 12, exact, np.zeros(200)))
        np.savez(os.path.join(HERE, f"frozen_rosenbrock_d100_{tag}.npz"),
                 **frozen_ensemble(100, O.LIKE_ROSENBROCK, 40, 12, exact, rng.uniform(0.5, 1.5, (100, 40))))
    np.savez(os.path.join(HERE, "hmc_quadform_d60_reference.npz"), **hmc_chains(60, 6, 6, 8, 0.05, 0.3, False))
    np.savez(os.path.join(HERE, "hmc_quadform_d60_fused.npz"), **hmc_chains(60, 6, 6, 8, 0.05, 0.3, True))


if __name__ == "__main__":
    main()
