#!/usr/bin/env python3
"""The north-star's acceptance criterion: "posterior mean/covariance within 1 % of the CPU reference after 10^6 steps".

Config 2's workload (README-form TDummyLogLikelihood D = 50, 65 536 chains, pooled covariance, sync every 256 steps):
10^6 steps of EVERY chain on the device (6.6e10 chain-steps, ~25 s), the posterior mean and covariance of everything
the ensemble visited taken by the device reducers (the pooled moment sums, PosteriorMoments), against
  (a) the closed form (mean 0, covariance I: SURVEY.md section 8c), in units of sigma, and
  (b) the CPU reference chain: oracle.Chain (the restatement of TSimpleMCMC<L, TProposeAdaptiveStep>, covariance NOT
      frozen) run for 10^6 steps on the host, whose own Monte-Carlo error (one chain, ESS of a few thousand) is what
      bounds that comparison -- it is printed next to the differences.
Not part of the test suite (a minute); tests/test_gpu_acceptance.py asserts the same on a shorter run.
usage: python tools/acceptance_criterion.py [--steps 1000000] [--chains 65536] [--json out.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def device_posterior(pkg, dim, chains, steps, window=256, burn=20):
    e = pkg.Engine(dim, chains, seed=20240607)
    assert e.Start(np.zeros(dim))
    for _ in range(burn):
        e.Step(window)
        e.sync()
    acc = pkg.PosteriorMoments(dim)
    t0 = time.perf_counter()
    for _ in range(steps // window):
        e.Step(window)
        e.reduce_moments()
        acc.add(e)
        e.apply_moments()
    dt = time.perf_counter() - t0
    accept = float(e.lane("naccept").sum() / (e.get_param("TOTAL_STEPS") * chains))
    e.close()
    return acc, dt, accept


def reference_chain_posterior(dim, steps, burn=50000):
    from oracle import oracle as O
    O.build()
    c = O.Chain(dim)
    assert c.start(np.zeros(dim))
    c.run_quiet(burn)
    t0 = time.perf_counter()
    s1, s2, nacc = c.run_moments(steps)
    dt = time.perf_counter() - t0
    mean = s1 / steps
    return mean, s2 / steps - np.outer(mean, mean), nacc / steps, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dim", type=int, default=50)
    ap.add_argument("--chains", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=1000000)
    ap.add_argument("--json")
    a = ap.parse_args()
    from smcmc_amd_loader import load_package
    pkg = load_package()
    acc, dt, accept = device_posterior(pkg, a.dim, a.chains, a.steps)
    mean, cov = acc.mean, acc.covariance
    eye = np.eye(a.dim)
    out = {"dim": a.dim, "chains": a.chains, "steps_per_chain": (a.steps // 256) * 256, "device_seconds": dt,
           "chain_steps_per_s": acc.n / dt, "accept_rate": accept,
           "max_abs_mean_in_sigma": float(np.max(np.abs(mean))),
           "max_abs_cov_minus_identity": float(np.max(np.abs(cov - eye)))}
    rmean, rcov, racc, rdt = reference_chain_posterior(a.dim, a.steps)
    # the reference chain's own error: its integrated autocorrelation time is ~ 3 D / acceptance steps for this target
    out.update({"reference_chain_steps": a.steps, "reference_seconds": rdt, "reference_accept_rate": racc,
                "reference_max_abs_mean": float(np.max(np.abs(rmean))),
                "reference_max_abs_cov_minus_identity": float(np.max(np.abs(rcov - eye))),
                "max_abs_mean_device_minus_reference": float(np.max(np.abs(mean - rmean))),
                "max_abs_cov_device_minus_reference": float(np.max(np.abs(cov - rcov)))})
    out["within_1_percent_of_closed_form"] = bool(out["max_abs_mean_in_sigma"] < 0.01 and out["max_abs_cov_minus_identity"] < 0.01)
    print(json.dumps(out, indent=1))
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
