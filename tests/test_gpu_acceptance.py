"""The north-star's acceptance criterion on the device (tools/acceptance_criterion.py is the 10^6-step form): config 2's
workload, posterior mean and covariance of everything the ensemble visits from the device reducers, within 1 % (in
units of sigma) of the closed form (iso-Gaussian: mean 0, covariance I -- SURVEY.md section 8c) and, inside its own
Monte-Carlo error, of the CPU reference chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config2_posterior_within_one_percent(gpu, oracle):
    dim, chains, window = 50, 65536, 256
    e = gpu.Engine(dim, chains, seed=20240607)
    assert e.Start(np.zeros(dim))
    for _ in range(12):                                        # adaptation from the start at the origin
        e.Step(window); e.sync()
    acc = gpu.PosteriorMoments(dim)
    for _ in range(40):                                        # 10 240 steps of all 65 536 chains: 6.7e8 points
        e.Step(window); e.reduce_moments(); acc.add(e); e.apply_moments()
    assert acc.n == chains * window * 40
    mean, cov = acc.mean, acc.covariance
    assert np.max(np.abs(mean)) < 0.01, np.max(np.abs(mean))                       # 1 % of sigma = 1
    assert np.max(np.abs(cov - np.eye(dim))) < 0.01, np.max(np.abs(cov - np.eye(dim)))
    assert abs(e.lane("acceptance").mean() - 0.234) < 0.03     # the chains' running acceptance sits on the target
    # the CPU reference chain (covariance adapting, TSimpleMCMC.H:1780-1820), 3e5 steps: one chain's estimate carries a
    # Monte-Carlo error of a few percent (ESS ~ steps / (3 D / 0.234)), the comparison is made at that level
    c = oracle.Chain(dim)
    assert c.start(np.zeros(dim))
    c.run_quiet(30000)
    steps = 300000
    s1, s2, _ = c.run_moments(steps)
    rmean = s1 / steps
    rcov = s2 / steps - np.outer(rmean, rmean)
    ess = steps / (3.0 * dim / 0.234)
    assert np.max(np.abs(mean - rmean)) < 5.0 / np.sqrt(ess)
    assert np.max(np.abs(np.diag(cov) - np.diag(rcov))) < 5.0 * np.sqrt(2.0 / ess)
