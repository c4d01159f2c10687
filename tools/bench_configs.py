#!/usr/bin/env python3
"""Timing of the non-headline BASELINE configs (3, 4 per-GPU share, 5) on one GPU.
Not the bench.py contract: a helper whose numbers go to profiles/*_notes.md."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from smcmc_amd_loader import load_package  # noqa: E402

pkg = load_package()


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


out = {}
# config 3: Rosenbrock D=200, 16 384 chains (frozen covariance)
e = pkg.Engine(200, 16384, likelihood=pkg.LIKE_ROSENBROCK, likelihood_params=[100.0], mode=pkg.MODE_FROZEN)
rng = np.random.default_rng(0)
e.Start(rng.uniform(0.5, 1.5, (200, 16384)))
dt = timed(lambda: e.Step(32), 3)
out["config3_rosenbrock_d200_16384_frozen"] = {"chain_steps_per_s": 16384 * 32 / dt, "ms_per_step": dt / 32 * 1e3}
# config 4, one GPU's share: iso D=500, 32 768 chains (frozen covariance)
e = pkg.Engine(500, 32768, mode=pkg.MODE_FROZEN)
e.Start(np.zeros(500))
dt = timed(lambda: e.Step(16), 3)
out["config4_iso_d500_32768_frozen"] = {"chain_steps_per_s": 32768 * 16 / dt, "ms_per_step": dt / 16 * 1e3}
# the same two in the fused order: the proposal on the FP64 matrix pipe
e = pkg.Engine(200, 16384, likelihood=pkg.LIKE_ROSENBROCK, likelihood_params=[100.0], mode=pkg.MODE_FROZEN, exact=False)
e.Start(rng.uniform(0.5, 1.5, (200, 16384)))
dt = timed(lambda: e.Step(64), 3)
out["config3_rosenbrock_d200_16384_frozen_fused_matrix_pipe"] = {"chain_steps_per_s": 16384 * 64 / dt, "ms_per_step": dt / 64 * 1e3}
e = pkg.Engine(500, 32768, mode=pkg.MODE_FROZEN, exact=False)
e.Start(np.zeros(500))
dt = timed(lambda: e.Step(32), 3)
out["config4_iso_d500_32768_frozen_fused_matrix_pipe"] = {
    "chain_steps_per_s": 32768 * 32 / dt, "ms_per_step": dt / 32 * 1e3,
    "proposal_TFLOPs": 32768 * 32 / dt * 500 * 501 / 1e12}
# config 4 with the header-form TDummyLogLikelihood (quadratic form, Error from Init()): fused order only at D > 63
# TDummyLogLikelihood::Init() (TDummyLogLikelihood.H:44-142): identity covariance except the (0, D-1) pair
cov = np.eye(500)
cov[0, 499] = cov[499, 0] = 0.999999
err = np.linalg.inv(cov)
e = pkg.Engine(500, 32768, likelihood=pkg.LIKE_QUADFORM, likelihood_params=err, mode=pkg.MODE_FROZEN, exact=False)
e.Start(np.zeros(500))
dt = timed(lambda: e.Step(32), 3)
out["config4_tdummy_quadform_d500_32768_frozen_fused_matrix_pipe"] = {
    "chain_steps_per_s": 32768 * 32 / dt, "ms_per_step": dt / 32 * 1e3,
    "matrix_TFLOPs": 32768 * 32 / dt * (500 * 501 + 2 * 500 * 500) / 1e12}
# configs 3 and 4 with the pooled covariance: moment fold every 16th step, sync every 256 steps (BASELINE config 4)
for name, dim, n, kind, prm in (("config3_rosenbrock_d200_16384_pooled", 200, 16384, pkg.LIKE_ROSENBROCK, [100.0]),
                               ("config4_iso_d500_32768_pooled", 500, 32768, pkg.LIKE_ISO_GAUSS, None)):
    for exact in (True, False):
        e = pkg.Engine(dim, n, likelihood=kind, likelihood_params=prm, mode=pkg.MODE_POOLED, exact=exact)
        e.set_param("MOMENT_STRIDE", 16)
        e.Start(rng.uniform(0.5, 1.5, (dim, n)) if kind == pkg.LIKE_ROSENBROCK else np.zeros(dim))

        def window():
            e.Step(256)
            e.sync()
        dt = timed(window, 1)
        out[name + ("" if exact else "_fused_matrix_pipe")] = {
            "chain_steps_per_s": n * 256 / dt, "ms_per_step": dt / 256 * 1e3,
            "accept": float(e.lane("naccept").sum() / (e.get_param("TOTAL_STEPS") * n))}
# config 5: HMC, header TDummy D=500 (Error from Init()), 8 192 chains x 20 leapfrog steps
for tag, exact, nstep in (("reference_order", True, 2), ("fused_order_matrix_pipe", False, 20)):
    h = pkg.HmcEngine(500, 8192, likelihood=pkg.LIKE_QUADFORM, likelihood_params=err, exact=exact)
    h.Start(np.ones(500)); h.SetMeanEpsilon(-0.0005); h.SetLeapFrog(20)
    dt = timed(lambda: h.Step(nstep), 2)
    out["config5_hmc_quadform_d500_8192_L20_" + tag] = {
        "trajectories_per_s": 8192 * nstep / dt, "ms_per_step": dt / nstep * 1e3,
        "gradient_TFLOPs": 8192 * nstep / dt * 21 * 2 * 500 * 500 / 1e12,
        "accept": float(h.lane("naccept").mean() / h.lane("trials").mean())}
print(json.dumps(out, indent=1))
