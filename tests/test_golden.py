"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py
from the CPU oracle).  CPU: the oracle still reproduces them.  GPU: the HIP path
reproduces them with no oracle in the loop."""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FROZEN = ["frozen_iso_d5.npz", "frozen_rosenbrock_d6.npz", "frozen_iso_d50.npz"]
POOLED = ["pooled_iso_d5.npz", "pooled_iso_d20.npz"]
LARGE = ["frozen_iso_d200_reference.npz", "frozen_iso_d200_fused.npz", "frozen_rosenbrock_d100_reference.npz",
         "frozen_rosenbrock_d100_fused.npz"]
HMC = ["hmc_quadform_d60_reference.npz", "hmc_quadform_d60_fused.npz"]


def _load(name):
    return {k: v for k, v in np.load(os.path.join(GOLDEN, name)).items()}


FROZEN_DEFINITION = {"frozen": "frozen_definition_iso_d5.npz", "pooled": "frozen_definition_pooled_iso_d12.npz"}


@pytest.mark.parametrize("name", FROZEN)
def test_oracle_reproduces_frozen_golden(oracle, name):
    _oracle_frozen(oracle, name)


def _oracle_frozen(oracle, name):
    g = _load(name)
    dim, kind, steps = int(g["dim"]), int(g["kind"]), int(g["steps"])
    for ch in range(g["accepted"].shape[0]):
        c = oracle.Chain(dim, kind=kind, seed=int(g["seed"]), chain_id=ch)
        c.set_covariance_frozen(1)
        x0 = g["x0"][:, ch] if g["x0"].ndim == 2 else g["x0"]
        assert c.start(x0)
        r = c.run(steps)
        assert np.array_equal(r["accepted"], g["accepted"][ch])
        assert np.array_equal(r["logl_accepted"], g["logl"][ch])
        assert np.array_equal(r["sigma"], g["sigma"][ch])
        assert np.array_equal(c.accepted, g["x"][:, ch])


@pytest.mark.parametrize("name", POOLED)
def test_oracle_reproduces_pooled_golden(oracle, name):
    _oracle_pooled(oracle, name)


def _oracle_pooled(oracle, name):
    g = _load(name)
    e = oracle.Ensemble(int(g["nchains"]), int(g["dim"]), kind=int(g["kind"]), seed=int(g["seed"]),
                        mode=oracle.MODE_POOLED)
    assert e.start(np.zeros(int(g["dim"])))
    m = None
    for _ in range(int(g["nwin"])):
        e.step(int(g["window"]))
        m = e.reduce_moments()
        e.apply_moments(m)
    e.step(4)
    assert np.array_equal(e.x, g["x"]) and np.array_equal(m, g["last_moments"])
    assert np.array_equal(e.decomposition, g["decomposition"]) and np.array_equal(e.covariance, g["covariance"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", FROZEN)
def test_hip_reproduces_frozen_golden(gpu, name):
    _hip_frozen(gpu, name)


def _hip_frozen(gpu, name, library=None):
    g = _load(name)
    dim, kind, steps = int(g["dim"]), int(g["kind"]), int(g["steps"])
    nch = g["accepted"].shape[0]
    prm = [100.0] if kind == 2 else None
    e = gpu.Engine(dim, nch, likelihood=kind, likelihood_params=prm, seed=int(g["seed"]), mode=gpu.MODE_FROZEN, library=library)
    assert e.Start(g["x0"])
    acc, logl, sigma = [], [], []
    for _ in range(steps):
        e.Step(1)
        acc.append(e.lane("last_accept").copy()); logl.append(e.lane("logl").copy()); sigma.append(e.lane("sigma").copy())
    assert np.array_equal(np.array(acc).T.astype(np.uint8), g["accepted"])   # identical accept/reject sequence
    assert np.array_equal(np.array(logl).T, g["logl"])                        # log-likelihoods: 0 ulp
    assert np.array_equal(np.array(sigma).T, g["sigma"])
    assert np.array_equal(e.GetAccepted(), g["x"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", POOLED)
def test_hip_reproduces_pooled_golden(gpu, name):
    _hip_pooled(gpu, name)


def _hip_pooled(gpu, name, library=None):
    g = _load(name)
    dim = int(g["dim"])
    e = gpu.Engine(dim, int(g["nchains"]), likelihood=int(g["kind"]), seed=int(g["seed"]), mode=gpu.MODE_POOLED, library=library)
    assert e.Start(np.zeros(dim))
    m = None
    for _ in range(int(g["nwin"])):
        e.Step(int(g["window"]))
        e.reduce_moments()
        m = e.read_moments()
        e.apply_moments()
    e.Step(4)
    assert np.array_equal(m, g["last_moments"])
    assert np.array_equal(e.GetAccepted(), g["x"])
    assert np.array_equal(e.lane("logl"), g["logl"]) and np.array_equal(e.lane("sigma"), g["sigma"])
    assert np.array_equal(e.lane("naccept"), g["naccept"])
    assert np.array_equal(e.decomposition, g["decomposition"]) and np.array_equal(e.covariance, g["covariance"])
    assert np.array_equal(e.GetEstimatedCenter(), g["center"])


# ---------------------------------------------------------------- the frozen-definition set
# One golden set whose definition does not move with the engine: Philox4x32 with the paper's ten rounds and the textbook
# normal pair (include/smcmc_detmath.h behind -DSMCMC_PHILOX_ROUNDS=10 -DSMCMC_NORMAL_TEXTBOOK), honoured by a second build
# of the oracle and by lib/libsmcmc_amd_frozen_definition.so (the D <= 15 README-form step kernels recompiled).  Whatever
# the production draws become, these two files and these tests stay.
def test_frozen_definition_oracle_reproduces_its_golden(oracle):
    F = oracle.frozen_definition()
    assert F.philox_draw_rounds() == 10 and oracle.philox_draw_rounds() == 7
    _oracle_frozen(F, FROZEN_DEFINITION["frozen"])
    _oracle_pooled(F, FROZEN_DEFINITION["pooled"])
    # ... and it is another definition than the production one: the same chain under the two builds
    g = _load(FROZEN_DEFINITION["frozen"])
    c = oracle.Chain(int(g["dim"]), kind=int(g["kind"]), seed=int(g["seed"]), chain_id=0)
    c.set_covariance_frozen(1)
    assert c.start(g["x0"])
    c.run(int(g["steps"]))
    assert not np.array_equal(c.accepted, g["x"][:, 0])


def test_frozen_definition_normal_pair_is_the_textbook_formula(oracle):
    """r = sqrt(-2 ln u1), theta = 2 pi u2 with u = (w + 1/2) 2^-32: the frozen build's pair against numpy's libm to a few
    ulp, and against the production transform (another function of the same two words, equal to ~1e-15)."""
    F = oracle.frozen_definition()
    rng = np.random.default_rng(5)
    w0 = rng.integers(0, 2 ** 32, 4000, dtype=np.uint64).astype(np.uint32)
    w1 = rng.integers(0, 2 ** 32, 4000, dtype=np.uint64).astype(np.uint32)
    n0, n1 = F.det_normal_pair(w0, w1)
    u1 = (w0.astype(np.float64) + 0.5) * 2.0 ** -32
    th = 2.0 * np.pi * (w1.astype(np.float64) + 0.5) * 2.0 ** -32
    r = np.sqrt(-2.0 * np.log(u1))
    assert np.max(np.abs(n0 - r * np.cos(th))) < 2e-15 * 7 and np.max(np.abs(n1 - r * np.sin(th))) < 2e-15 * 7
    p0, p1 = oracle.det_normal_pair(w0, w1)
    assert np.max(np.abs(n0 - p0)) < 1e-14 and np.max(np.abs(n1 - p1)) < 1e-14
    assert not (np.array_equal(n0, p0) and np.array_equal(n1, p1))


@pytest.mark.gpu
def test_hip_frozen_definition_build_reproduces_its_golden(gpu):
    lib = gpu.FROZEN_DEFINITION_LIB_PATH
    assert os.path.exists(lib), "lib/libsmcmc_amd_frozen_definition.so is missing: __graft_entry__.build() makes it"
    _hip_frozen(gpu, FROZEN_DEFINITION["frozen"], library=lib)
    _hip_pooled(gpu, FROZEN_DEFINITION["pooled"], library=lib)


# ---------------------------------------------------------------- large dimensions and HMC
@pytest.mark.parametrize("name", LARGE)
def test_oracle_reproduces_large_dim_golden(oracle, name):
    g = _load(name)
    dim, kind, n = int(g["dim"]), int(g["kind"]), int(g["nchains"])
    prm = [100.0] if kind == 2 else None
    e = oracle.Ensemble(n, dim, kind=kind, params=prm, seed=int(g["seed"]), mode=oracle.MODE_FROZEN,
                        exact=bool(g["exact"]))
    assert e.start(g["x0"])
    e.step(int(g["steps"]))
    assert np.array_equal(e.x, g["x"]) and np.array_equal(e.lane("logl"), g["logl"])
    assert np.array_equal(e.lane("sigma"), g["sigma"]) and np.array_equal(e.lane("naccept"), g["naccept"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", LARGE)
def test_hip_reproduces_large_dim_golden(gpu, name):
    """D > 63: the panel kernel (reference order) and the matrix-pipe kernel (fused order)."""
    g = _load(name)
    dim, kind, n = int(g["dim"]), int(g["kind"]), int(g["nchains"])
    prm = [100.0] if kind == 2 else None
    e = gpu.Engine(dim, n, likelihood=kind, likelihood_params=prm, seed=int(g["seed"]), mode=gpu.MODE_FROZEN,
                   exact=bool(g["exact"]))
    assert e.Start(g["x0"])
    e.Step(int(g["steps"]))
    assert np.array_equal(e.GetAccepted(), g["x"]) and np.array_equal(e.lane("logl"), g["logl"])
    assert np.array_equal(e.lane("sigma"), g["sigma"]) and np.array_equal(e.lane("naccept"), g["naccept"])
    assert np.array_equal(e.lane("step_rms"), g["step_rms"])


def _hmc_oracle_state(oracle, g):
    dim, n = int(g["dim"]), int(g["nchains"])
    q, m, pot = [], [], []
    for ch in range(n):
        h = oracle.Hmc(dim, kind=1, params=g["error"], seed=int(g["seed"]), chain_id=ch, potential_from_gradient=True,
                       fused_gradient=bool(g["fused"]))
        h.set_alpha(float(g["alpha"])); h.start(np.ones(dim))
        h.set_mean_epsilon(-float(g["epsilon"])); h.set_leapfrog(int(g["leapfrog"]))
        h.run(int(g["steps"]))
        q.append(h.accepted); m.append(h.momentum); pot.append(h.scalars["accepted_potential"])
    return np.array(q).T, np.array(m).T, np.array(pot)


@pytest.mark.parametrize("name", HMC)
def test_oracle_reproduces_hmc_golden(oracle, name):
    g = _load(name)
    q, m, pot = _hmc_oracle_state(oracle, g)
    assert np.array_equal(q, g["q"]) and np.array_equal(m, g["momentum"]) and np.array_equal(pot, g["potential"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", HMC)
def test_hip_reproduces_hmc_golden(gpu, name):
    g = _load(name)
    dim, n = int(g["dim"]), int(g["nchains"])
    e = gpu.HmcEngine(dim, n, likelihood=1, likelihood_params=g["error"], seed=int(g["seed"]),
                      exact=not bool(g["fused"]))
    e.SetAlpha(float(g["alpha"]))
    e.Start(np.ones(dim))
    e.SetMeanEpsilon(-float(g["epsilon"]))
    e.SetLeapFrog(int(g["leapfrog"]))
    e.Step(int(g["steps"]))
    q, m, logl = e.state()
    assert np.array_equal(q, g["q"]) and np.array_equal(m, g["momentum"]) and np.array_equal(-logl, g["potential"])
    assert np.array_equal(e.lane("acceptance"), g["acceptance"])


# ---------------------------------------------------------------- round 2: VAAT, stress likelihoods, adaptive HMC
VAAT = ["vaat_iso_d7.npz", "vaat_rosenbrock_d31_fused.npz", "vaat_quadform_d100.npz"]
STRESS = ["stress_asym_d20_pooled.npz", "stress_constrained_d25_pooled_fused.npz", "stress_horrific_d75_frozen.npz"]
HMC2 = ["hmc_adaptive_iso_d20.npz", "hmc_adaptive_quadform_d100.npz", "hmc_gradient_types_d12.npz"]


def _vaat_check(g, x, lane, per_dim, qlen):
    assert np.array_equal(x, g["x"])
    for name in ("logl", "logl_proposed", "step_rms", "proposed_value", "trials", "successes", "naccept", "last_index"):
        assert np.array_equal(lane(name), g[name]), name
    for name in ("sigma", "acceptance", "acceptance_trials"):
        assert np.array_equal(per_dim(name), g[name]), name
    assert qlen == int(g["queue_len"]) and np.array_equal(per_dim("queue")[:qlen], g["queue"])


@pytest.mark.parametrize("name", VAAT)
def test_oracle_reproduces_vaat_golden(oracle, name):
    g = _load(name)
    v = oracle.Vaat(int(g["nchains"]), int(g["dim"]), kind=int(g["kind"]), params=g.get("params"), seed=int(g["seed"]),
                    exact=bool(g["exact"]))
    v.set_step_rms_window(50)
    assert v.start(g["x0"])
    v.update_proposal()
    v.step(int(g["steps"]))
    _vaat_check(g, v.x, v.lane, v.per_dim, int(v.lane("queue_len")[0]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", VAAT)
def test_hip_reproduces_vaat_golden(gpu, name):
    g = _load(name)
    e = gpu.VaatEngine(int(g["dim"]), int(g["nchains"]), likelihood=int(g["kind"]), likelihood_params=g.get("params"),
                       seed=int(g["seed"]), exact=bool(g["exact"]))
    e.SetStepRMSWindow(50)
    assert e.Start(g["x0"])
    e.UpdateProposal()
    e.Step(int(g["steps"]))
    _vaat_check(g, e.GetAccepted(), e.lane, e.per_dim, e.queue_length)


def _stress_run(g, make, sync_name):
    e = make()
    assert e.Start(g["x0"]) if hasattr(e, "Start") else e.start(g["x0"])
    step = e.Step if hasattr(e, "Step") else e.step
    for _ in range(int(g["nwin"])):
        step(int(g["window"]))
        if int(g["mode"]) == 1:
            e.sync()
    step(3)
    return e


def _stress_check(g, x, lane, cov, dec):
    assert np.array_equal(x, g["x"])
    for name in ("logl", "sigma", "naccept", "logl_proposed"):
        assert np.array_equal(lane(name), g[name]), name
    assert np.array_equal(cov, g["covariance"]) and np.array_equal(dec, g["decomposition"])


@pytest.mark.parametrize("name", STRESS)
def test_oracle_reproduces_stress_golden(oracle, name):
    g = _load(name)
    dim, kind = int(g["dim"]), int(g["kind"])
    prm = oracle.like_params(kind, dim)
    e = _stress_run(g, lambda: oracle.Ensemble(int(g["nchains"]), dim, kind=kind, params=prm if prm.size else None,
                                               seed=int(g["seed"]), mode=int(g["mode"]), exact=bool(g["exact"])), "sync")
    _stress_check(g, e.x, e.lane, e.covariance, e.decomposition)


@pytest.mark.gpu
@pytest.mark.parametrize("name", STRESS)
def test_hip_reproduces_stress_golden(gpu, name):
    g = _load(name)
    dim, kind = int(g["dim"]), int(g["kind"])
    # the parameters of the likelihoods as the reference's headers fix them (constrained: example4's Init())
    prm = None
    if kind == 6:
        expected = np.full(dim, 76.0); prior = np.full(dim, 76.0 * 0.08)
        expected[-1], prior[-1] = 80.0, 2.0
        prm = np.concatenate([[1902.0, 16.0], expected, prior])
    e = _stress_run(g, lambda: gpu.Engine(dim, int(g["nchains"]), likelihood=kind, likelihood_params=prm, seed=int(g["seed"]),
                                          mode=int(g["mode"]), exact=bool(g["exact"])), "sync")
    _stress_check(g, e.GetAccepted(), e.lane, e.covariance, e.decomposition)


def _hmc2_check(g, q, m, potential, lane_eps, lane_l, lane_rev, lane_acc, average, covariance, shared):
    assert np.array_equal(q, g["q"]) and np.array_equal(m, g["momentum"]) and np.array_equal(potential, g["potential"])
    assert np.array_equal(lane_eps, g["mean_epsilon"]) and np.array_equal(lane_l, g["leapfrog"].astype(lane_l.dtype))
    assert np.array_equal(lane_rev, g["reversal_len"]) and np.array_equal(lane_acc, g["acceptance"])
    assert np.array_equal(average, g["average"]) and np.array_equal(covariance, g["covariance"])
    assert np.array_equal(shared, g["shared"])


@pytest.mark.parametrize("name", HMC2)
def test_oracle_reproduces_adaptive_hmc_golden(oracle, name):
    g = _load(name)
    dim = int(g["dim"])
    e = oracle.HmcEnsemble(int(g["nchains"]), dim, kind=int(g["kind"]), params=g.get("params"), seed=int(g["seed"]),
                           group=int(g["group"]), sync_every=1, potential_from_gradient=True)
    e.start(np.full(dim, 0.7))
    if int(g["fix_leapfrog"]) >= 0:
        e.set_leapfrog(int(g["fix_leapfrog"]))
    for steps, gtype in g["schedule"]:
        e.set_gradient_type(int(gtype))
        e.step(int(steps))
    q, m = e.state()
    _hmc2_check(g, q, m, e.lane("accepted_potential"), e.lane("mean_epsilon"), e.lane("leapfrog_steps"), e.lane("reversal_len"),
                e.lane("current_acceptance"), e.average, e.covariance, np.array([e.shared[k] for k in oracle.HMC_SHARED]))


@pytest.mark.gpu
@pytest.mark.parametrize("name", HMC2)
def test_hip_reproduces_adaptive_hmc_golden(gpu, name):
    g = _load(name)
    dim = int(g["dim"])
    e = gpu.HmcEngine(dim, int(g["nchains"]), likelihood=int(g["kind"]), likelihood_params=g.get("params"), seed=int(g["seed"]))
    assert e.moment_group == int(g["group"])      # the fixture's moment groups are the engine's for this ensemble size
    e.Start(np.full(dim, 0.7))
    if int(g["fix_leapfrog"]) >= 0:
        e.SetLeapFrog(int(g["fix_leapfrog"]))
    for steps, gtype in g["schedule"]:
        e.Step(int(steps), gradient_type=int(gtype))
    q, m, logl = e.state()
    t = e.tuning
    shared = np.array([t[k] for k in ("trace", "orbit", "updates", "cov_trials", "average_trials", "steps_remaining",
                                      "steps_since_update", "max_scale", "min_scale", "est_trace")])
    _hmc2_check(g, q, m, -logl, e.lane("mean_epsilon"), e.lane("leapfrog"), e.lane("reversal_len"), e.lane("acceptance"),
                e.average, e.covariance, shared)


# ---- round 3: the reference's own mode, every chain adapting its own covariance (SMCMC_MODE_PER_CHAIN) ----
PERCHAIN = ["perchain_iso_d20.npz", "perchain_rosenbrock_d6.npz", "perchain_iso_d50.npz"]


def _perchain_schedule(g):
    """(next_update or None, steps) segments of tests/golden/make_golden.py::adaptive_chains"""
    seg = [(int(nxt), int(g["run"])) for nxt in g["forced"]]
    return seg + [(None, int(g["tail"])), ("update", 16)]


@pytest.mark.parametrize("name", PERCHAIN)
def test_oracle_reproduces_perchain_golden(oracle, name):
    g = _load(name)
    dim, kind = int(g["dim"]), int(g["kind"])
    names = [str(s) for s in g["scalar_names"]]
    for ch in (0, g["accepted"].shape[0] - 1):
        c = oracle.Chain(dim, kind=kind, seed=int(g["seed"]), chain_id=ch)
        c.set_acceptance_window(int(g["window"]))
        assert c.start(g["x0"][:, ch] if g["x0"].ndim == 2 else g["x0"])
        bits = []
        for nxt, steps in _perchain_schedule(g):
            if nxt == "update":
                c.update_proposal()
            elif nxt is not None:
                c.set_next_update(nxt)
            bits.append(c.run(steps)["accepted"])
        assert np.array_equal(np.concatenate(bits), g["accepted"][ch])
        assert np.array_equal(c.accepted, g["x"][:, ch]) and np.array_equal(c.covariance, g["covariance"][ch])
        assert np.array_equal(c.decomposition, g["decomposition"][ch]) and np.array_equal(c.center, g["centre"][ch])
        sc = c.scalars
        assert [sc[k] for k in names] == list(g["scalars"][ch])
        assert sc["update_count"] >= 4                       # Start's, forced ones inside the runs, the explicit one


@pytest.mark.gpu
@pytest.mark.parametrize("name", PERCHAIN)
def test_hip_reproduces_perchain_golden(gpu, name):
    """No oracle in the loop: the engine's per-chain mode against the committed end states and accept sequences."""
    g = _load(name)
    dim, kind, nch = int(g["dim"]), int(g["kind"]), g["accepted"].shape[0]
    prm = [100.0] if kind == 2 else None
    e = gpu.Engine(dim, nch, likelihood=kind, likelihood_params=prm, seed=int(g["seed"]), mode=gpu.MODE_PER_CHAIN)
    e.SetAcceptanceWindow(int(g["window"]))
    assert e.Start(g["x0"])
    bits = []
    for nxt, steps in _perchain_schedule(g):
        if nxt == "update":
            e.UpdateProposal()
        elif nxt is not None:
            e.SetNextUpdate(nxt)
        # the accept bits of every step: the first segment one step per launch, the rest through the naccept counter
        if not bits:
            seg = []
            for _ in range(steps):
                e.Step(1)
                seg.append(e.lane("last_accept").copy())
            bits.append(np.array(seg).T)
        else:
            before = e.lane("naccept").copy()
            e.Step(steps)
            bits.append(e.lane("naccept") - before)
    first = int(g["run"])
    assert np.array_equal(bits[0].astype(np.uint8), g["accepted"][:, :first])
    off = first
    for (nxt, steps), got in zip(_perchain_schedule(g)[1:], bits[1:]):
        assert np.array_equal(got, g["accepted"][:, off:off + steps].sum(axis=1))
        off += steps
    assert np.array_equal(e.GetAccepted(), g["x"])
    names = [str(s) for s in g["scalar_names"]]
    lane_of = {"accepted_logl": "logl", "central_trials": "center_trials", "cov_trials": "covariance_trials",
               "total_steps": "chain_steps"}
    for k, name_k in enumerate(names):
        assert np.array_equal(e.lane(lane_of.get(name_k, name_k)).astype(np.float64), g["scalars"][:, k]), name_k
    for ch in range(nch):
        centre, cov, dec = e.chain_proposal(ch)
        assert np.array_equal(centre, g["centre"][ch]) and np.array_equal(cov, g["covariance"][ch])
        assert np.array_equal(dec, g["decomposition"][ch])


def test_the_fixtures_are_the_frozen_ones():
    """tests/golden/*.npz are FROZEN: tests/golden/MANIFEST.json holds the hash of every fixture as of round 3, and a
    round that changes a kernel has to reproduce these files, not regenerate them (round 4 rebuilt the large-dimension
    moment fold and the per-chain mode against them).  A fixture may only change together with the manifest and a line
    in DESIGN.md saying which arithmetic moved and why."""
    import hashlib
    import json
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    man = json.load(open(os.path.join(here, "MANIFEST.json")))["sha256"]
    found = sorted(f for f in os.listdir(here) if f.endswith(".npz"))
    assert found == sorted(man), "fixtures added or removed without the manifest"
    for name in found:
        assert hashlib.sha256(open(os.path.join(here, name), "rb").read()).hexdigest() == man[name], name
