#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the CPU oracle.

The reference itself cannot be built or imported here (it needs ROOT), and it
ships no golden vectors, so these fixtures freeze the ORACLE's outputs (parity
unpinned, see oracle/oracle_core.h): the CPU suite checks that the oracle still
reproduces them, the GPU suite checks the HIP path against them without the
oracle in the loop.  Run from the repo root:  python tests/golden/make_golden.py   (everything)
or  python tests/golden/make_golden.py --round2 / --round3   (only the fixtures of that round that are not there yet);
--all rewrites every fixture (needed whenever include/smcmc_detmath.h changes the draws: round 3 did, Philox4x32-7 and the
table-driven normal transform).  Since round 3 the fixtures are FROZEN (MANIFEST.json): the script refuses to run
without --i-am-changing-the-arithmetic.
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


def tdummy_error(dim):
    """TDummyLogLikelihood::Init() (TDummyLogLikelihood.H:44-142)."""
    return O.dummy_error_matrix(dim)[1]


def vaat(dim, kind, nchains, steps, exact, params=None):
    """TSimpleMCMC<L, TProposeVAATStep>: SimpleVAAT.C's call sequence (Start, UpdateProposal, Step...)."""
    rng = np.random.default_rng(dim + 17)
    x0 = rng.uniform(-1.0, 1.0, size=(dim, nchains))
    v = O.Vaat(nchains, dim, kind=kind, params=params, seed=SEED, exact=exact)
    v.set_step_rms_window(50)
    assert v.start(x0)
    v.update_proposal()
    v.step(steps)
    qlen = int(v.lane("queue_len")[0])
    out = {"dim": dim, "kind": kind, "nchains": nchains, "steps": steps, "exact": int(exact), "seed": SEED, "x0": x0,
           "x": v.x, "queue_len": qlen, "queue": v.per_dim("queue")[:qlen]}
    if params is not None:
        out["params"] = np.asarray(params)
    for name in ("logl", "logl_proposed", "step_rms", "proposed_value", "trials", "successes", "naccept", "last_index"):
        out[name] = v.lane(name)
    for name in ("sigma", "acceptance", "acceptance_trials"):
        out[name] = v.per_dim(name)
    return out


def stress(dim, kind, nchains, mode, exact, window, nwin):
    """The stress likelihoods (asymmetric, horrific, constrained) in the adaptive ensemble."""
    rng = np.random.default_rng(kind * 31 + dim)
    if kind == O.LIKE_ASYM:
        x0 = rng.normal(0.3, 0.2, size=(dim, nchains))
    elif kind == O.LIKE_HORRIFIC:
        x0 = rng.uniform(-0.9, 0.9, size=(dim, nchains))
    else:
        x0 = 76.0 + rng.normal(0.0, 1.0, size=(dim, nchains))
    prm = O.like_params(kind, dim)
    e = O.Ensemble(nchains, dim, kind=kind, params=prm if prm.size else None, seed=SEED, mode=mode, exact=exact)
    assert e.start(x0)
    for _ in range(nwin):
        e.step(window)
        if mode == O.MODE_POOLED:
            e.sync()
    e.step(3)
    return {"dim": dim, "kind": kind, "nchains": nchains, "mode": mode, "exact": int(exact), "window": window,
            "nwin": nwin, "seed": SEED, "x0": x0, "x": e.x, "logl": e.lane("logl"), "sigma": e.lane("sigma"),
            "naccept": e.lane("naccept"), "logl_proposed": e.lane("logl_proposed"), "covariance": e.covariance,
            "decomposition": e.decomposition}


def hmc_ensemble(dim, kind, nchains, schedule, params=None, fix_leapfrog=None):
    """The HMC engine's pooled semantics: `schedule` = [(steps, gradient type), ...] after Start at 0.7."""
    e = O.HmcEnsemble(nchains, dim, kind=kind, params=params, seed=SEED, group=64, sync_every=1,
                      potential_from_gradient=True)
    e.start(np.full(dim, 0.7))
    if fix_leapfrog is not None:
        e.set_leapfrog(fix_leapfrog)
    for steps, gtype in schedule:
        e.set_gradient_type(gtype)
        e.step(steps)
    q, m = e.state()
    out = {"dim": dim, "kind": kind, "nchains": nchains, "seed": SEED, "group": 64, "schedule": np.array(schedule),
           "fix_leapfrog": -1 if fix_leapfrog is None else fix_leapfrog, "q": q, "momentum": m,
           "potential": e.lane("accepted_potential"), "mean_epsilon": e.lane("mean_epsilon"),
           "leapfrog": e.lane("leapfrog_steps"), "reversal_len": e.lane("reversal_len"),
           "acceptance": e.lane("current_acceptance"), "average": e.average, "covariance": e.covariance,
           "shared": np.array([e.shared[k] for k in O.HMC_SHARED])}
    if params is not None:
        out["params"] = np.asarray(params)
    return out


def adaptive_chains(dim, kind, nchains, x0, window=120, forced=(20, 20, 20), run=250, tail=600):
    """The reference's own mode: every chain adapts its own covariance every step (oracle.Chain, NOT frozen) and runs
    UpdateProposal on its own schedule -- shortened through the acceptance window and SetNextUpdate so that a few
    hundred steps cross several updates.  The engine's SMCMC_MODE_PER_CHAIN replays it."""
    prm = O.like_params(kind, dim)
    out = {"dim": dim, "kind": kind, "seed": SEED, "x0": x0, "window": window, "forced": np.array(forced), "run": run,
           "tail": tail}
    acc, xs, centre, cov, dec, sc = [], [], [], [], [], []
    names = ["accepted_logl", "sigma", "acceptance", "acceptance_trials", "rigidity", "step_rms", "central_trials",
             "cov_trials", "sigma_trace", "trials", "successes", "next_update", "update_count", "total_steps"]
    for ch in range(nchains):
        c = O.Chain(dim, kind=kind, params=prm if prm.size else None, seed=SEED, chain_id=ch)
        c.set_acceptance_window(window)
        assert c.start(x0[:, ch] if x0.ndim == 2 else x0)
        bits = []
        for nxt in forced:
            c.set_next_update(nxt)
            bits.append(c.run(run)["accepted"])
        bits.append(c.run(tail)["accepted"])
        c.update_proposal()
        bits.append(c.run(16)["accepted"])
        acc.append(np.concatenate(bits)); xs.append(c.accepted); centre.append(c.center); cov.append(c.covariance)
        dec.append(c.decomposition)
        s = c.scalars
        sc.append([s[k] for k in names])
    out.update(accepted=np.array(acc), x=np.array(xs).T, centre=np.array(centre), covariance=np.array(cov),
               decomposition=np.array(dec), scalars=np.array(sc), scalar_names=np.array(names))
    return out


def round3():
    rng = np.random.default_rng(3)
    save_new("perchain_iso_d20.npz", lambda: adaptive_chains(20, O.LIKE_ISO, 70, np.zeros(20)))
    save_new("perchain_rosenbrock_d6.npz", lambda: adaptive_chains(6, O.LIKE_ROSENBROCK, 66, rng.uniform(0.5, 1.5, (6, 66))))
    save_new("perchain_iso_d50.npz", lambda: adaptive_chains(50, O.LIKE_ISO, 4, np.zeros(50), window=60, forced=(40, 30), run=500,
                                                              tail=300))


def save_new(name, make):
    """Round-2 fixtures are written only when absent (np.savez archives are not byte-stable; the old ones stay put)."""
    path = os.path.join(HERE, name)
    if not os.path.exists(path) or "--all" in sys.argv:
        np.savez(path, **make())
        print("wrote", name)


def round2():
    save_new("vaat_iso_d7.npz", lambda: vaat(7, O.LIKE_ISO, 70, 300, True))
    save_new("vaat_rosenbrock_d31_fused.npz", lambda: vaat(31, O.LIKE_ROSENBROCK, 64, 200, False, [100.0]))
    save_new("vaat_quadform_d100.npz", lambda: vaat(100, O.LIKE_QUADFORM, 64, 150, True, tdummy_error(100)))
    save_new("stress_asym_d20_pooled.npz", lambda: stress(20, O.LIKE_ASYM, 128, O.MODE_POOLED, True, 20, 3))
    save_new("stress_constrained_d25_pooled_fused.npz", lambda: stress(25, O.LIKE_CONSTRAINED, 128, O.MODE_POOLED, False, 20, 3))
    save_new("stress_horrific_d75_frozen.npz", lambda: stress(75, O.LIKE_HORRIFIC, 64, O.MODE_FROZEN, True, 15, 2))
    save_new("hmc_adaptive_iso_d20.npz", lambda: hmc_ensemble(20, O.LIKE_ISO, 70, [(30, 0)]))
    save_new("hmc_adaptive_quadform_d100.npz",
             lambda: hmc_ensemble(100, O.LIKE_QUADFORM, 64, [(10, 0)], params=np.linalg.inv(spd(100, 3)), fix_leapfrog=6))
    save_new("hmc_gradient_types_d12.npz", lambda: hmc_ensemble(12, O.LIKE_ROSENBROCK, 64, [(3, 5), (3, 0), (3, 2), (2, 3)], params=[100.0]))


def frozen_definition():
    """The one golden set whose DEFINITION does not move with the engine: the draws are Philox4x32
    with the paper's ten rounds and the textbook Box-Muller pair through the <= 1 ulp functions of include/smcmc_detmath.h
    (-DSMCMC_PHILOX_ROUNDS=10 -DSMCMC_NORMAL_TEXTBOOK, honoured by the oracle and by the kernels of
    lib/libsmcmc_amd_frozen_definition.so).  Written once (round 4); `python tests/golden/make_golden.py
    --frozen-definition` refuses to overwrite."""
    global O
    plain = O
    O = plain.frozen_definition()
    try:
        for name, make in (("frozen_definition_iso_d5.npz", lambda: frozen_chains(5, O.LIKE_ISO, 6, 300, np.zeros(5))),
                           ("frozen_definition_pooled_iso_d12.npz", lambda: pooled(12, O.LIKE_ISO, 200, 24, 3))):
            path = os.path.join(HERE, name)
            if os.path.exists(path):
                print(name, "exists: left alone")
                continue
            np.savez(path, **make())
            print("wrote", name)
    finally:
        O = plain


def main():
    O.build()
    if "--round2" in sys.argv:
        round2()
        return
    if "--round3" in sys.argv:
        round3()
        return
    round2()
    round3()
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


if __name__ == "__main__" and "--frozen-definition" in sys.argv:
    frozen_definition()
    raise SystemExit(0)

if __name__ == "__main__":
    # The fixtures are frozen (MANIFEST.json, tests/test_golden.py::test_the_fixtures_are_the_frozen_ones): a kernel
    # change has to reproduce them.  Regenerating is a decision, not a build step.
    if "--i-am-changing-the-arithmetic" not in sys.argv:
        raise SystemExit("tests/golden/*.npz are frozen since round 3 (MANIFEST.json).  To regenerate them because the engine's "
                         "DEFINITION changed (draws, summation order), pass --i-am-changing-the-arithmetic, update MANIFEST.json "
                         "and say so in DESIGN.md.")
    sys.argv.remove("--i-am-changing-the-arithmetic")
    main()
