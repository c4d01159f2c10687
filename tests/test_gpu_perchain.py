"""SMCMC_MODE_PER_CHAIN: the reference's own mode on the device -- every chain keeps its own centre, covariance and
decomposition, runs UpdateState every step (TSimpleMCMC.H:1721-1831, the covariance loop of :1795-1820 included) and
UpdateProposal (:1009-1390) when its own --fNextUpdate < 1 on an accepted step (:1824-1826).  Checked lane by lane,
bit for bit, against oracle.Chain with the covariance NOT frozen: the restatement of the reference's single chain."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F64 = {"logl": "accepted_logl", "logl_proposed": "proposed_logl", "sigma": "sigma", "acceptance": "acceptance",
       "acceptance_trials": "acceptance_trials", "rigidity": "rigidity", "step_rms": "step_rms",
       "center_trials": "central_trials", "covariance_trials": "cov_trials", "sigma_trace": "sigma_trace"}
I32 = {"trials": "trials", "successes": "successes", "next_update": "next_update", "step_rms_trials": "step_rms_trials",
       "chain_steps": "total_steps", "update_count": "update_count", "last_update_path": "last_update_path"}


KERNEL = {"wave": -1}


@pytest.fixture(autouse=True, params=[0, 1], ids=["chain-per-lane", "chain-per-wavefront"])
def which_kernel(request):
    """Every test of the module runs on both kernels of the mode (SMCMC_P_PERCHAIN_WAVE): perchain_step_kernel and
    perchain_wave_kernel share the images and must give the same bits -- oracle.Chain's."""
    KERNEL["wave"] = request.param
    yield request.param


def _params(oracle, kind, dim):
    prm = oracle.like_params(kind, dim)
    return prm if prm.size else None


def _make(gpu, oracle, dim, n, kind=0, which=None, seed=20240607, offset=0, x0=None, setup=None):
    """An engine of n chains and oracle chains for the lanes in `which` (default: all)."""
    prm = _params(oracle, kind, dim)
    e = gpu.Engine(dim, n, likelihood=kind, likelihood_params=prm, seed=seed, chain_offset=offset,
                   mode=gpu.MODE_PER_CHAIN)
    e.set_param("PERCHAIN_WAVE", KERNEL["wave"])
    assert e.get_param("PERCHAIN_WAVE") == KERNEL["wave"]       # every built-in likelihood runs on either kernel
    which = list(range(n)) if which is None else list(which)
    chains = {c: oracle.Chain(dim, kind=kind, params=prm, seed=seed, chain_id=offset + c) for c in which}
    if setup:
        setup(e)
        for c in chains.values():
            setup(c)
    x0 = np.zeros(dim) if x0 is None else np.asarray(x0, dtype=np.float64)
    assert e.Start(x0)
    for c, ch in chains.items():
        assert ch.start(x0 if x0.ndim == 1 else x0[:, c])
    return e, chains


def _both(e, chains, name_e, name_c, *args):
    getattr(e, name_e)(*args)
    for ch in chains.values():
        getattr(ch, name_c)(*args)


def _step(e, chains, n, metropolis=0):
    e.Step(n, metropolis)
    for ch in chains.values():
        if metropolis == 0:
            ch.run_quiet(n)
        else:
            for _ in range(n):
                ch.step(False, metropolis)


def _same(e, chains, tag):
    x = e.GetAccepted()
    lanes = {k: e.lane(k) for k in list(F64) + list(I32)}
    for c, ch in chains.items():
        sc = ch.scalars
        assert np.array_equal(x[:, c], ch.accepted), f"{tag}: chain {c}: accepted point differs"
        for k, ok in F64.items():
            assert lanes[k][c] == sc[ok], f"{tag}: chain {c}: {k} = {lanes[k][c]!r}, reference chain {sc[ok]!r}"
        for k, ok in I32.items():
            assert lanes[k][c] == int(sc[ok]), f"{tag}: chain {c}: {k} = {lanes[k][c]}, reference chain {int(sc[ok])}"
        centre, cov, dec = e.chain_proposal(c)
        assert np.array_equal(centre, ch.center), f"{tag}: chain {c}: centre differs"
        assert np.array_equal(cov, ch.covariance), f"{tag}: chain {c}: covariance differs (max |d| = {np.max(np.abs(cov - ch.covariance))})"
        assert np.array_equal(dec, ch.decomposition), f"{tag}: chain {c}: decomposition differs"
        assert np.array_equal(e.chain(c)["proposed"], ch.proposed), f"{tag}: chain {c}: proposed point differs"


@pytest.mark.parametrize("kind,dim", [(0, 5), (0, 20), (1, 12), (2, 6), (2, 31), (4, 20), (5, 30), (6, 25), (0, 63)])
def test_every_lane_is_the_reference_chain(gpu, oracle, kind, dim):
    """Default settings, covariance adapting every step; the chains' own schedules fire UpdateProposal inside the
    launches (forced early through SetNextUpdate, then at the reference's own W + D^2 pace)."""
    n = 70                                                       # two wavefronts, the second one ragged
    rng = np.random.default_rng(100 * kind + dim)
    x0 = {2: rng.uniform(0.5, 1.5, size=(dim, n)), 5: rng.uniform(-0.5, 0.5, size=(dim, n)),
          6: 76.0 + rng.normal(0.0, 1.0, size=(dim, n))}.get(kind, np.zeros(dim))
    which = range(n) if dim <= 20 else (0, 1, 31, 63, 64, 69)
    e, chains = _make(gpu, oracle, dim, n, kind, which=which, x0=x0,
                      setup=lambda o: (o.SetAcceptanceWindow if hasattr(o, "SetAcceptanceWindow") else o.set_acceptance_window)(150))
    _same(e, chains, "start")
    _step(e, chains, 1)
    _same(e, chains, "first step")
    updates0 = e.lane("update_count").copy()
    for k in range(4):
        _both(e, chains, "SetNextUpdate", "set_next_update", 12)   # every chain updates on its 12th accepted step from here
        _step(e, chains, 300 if kind not in (2, 5) else 600)
        _same(e, chains, f"forced schedule {k}")
    assert np.median(e.lane("update_count") - updates0) >= 3, "the launches must cross UpdateProposal events"
    _step(e, chains, 1500)
    _same(e, chains, "free running")
    _both(e, chains, "UpdateProposal", "update_proposal")
    _same(e, chains, "explicit UpdateProposal")
    _step(e, chains, 64)
    _same(e, chains, "after the explicit update")


def test_headline_dimension_through_two_updates(gpu, oracle):
    """D = 50: the covariance stream is 1 275 elements per chain; two UpdateProposal events at the reference's own
    pace need ~10^4 steps, so the schedule is shortened through the acceptance window."""
    dim, n = 50, 130
    which = (0, 17, 63, 64, 128, 129)
    e, chains = _make(gpu, oracle, dim, n, 0, which=which,
                      setup=lambda o: (o.SetAcceptanceWindow if hasattr(o, "SetAcceptanceWindow") else o.set_acceptance_window)(60))
    _both(e, chains, "SetNextUpdate", "set_next_update", 40)
    _step(e, chains, 700)
    _same(e, chains, "first update")
    assert np.all(e.lane("update_count") >= 2)                   # Start's and the forced one
    _both(e, chains, "SetNextUpdate", "set_next_update", 30)
    _step(e, chains, 500)
    _same(e, chains, "second update")
    assert np.all(e.lane("update_count") >= 3)
    assert np.all(e.lane("decomp_full") == 0) and np.all(e.lane("last_update_path") == 0)


def test_simplemcmc_schedule(gpu, oracle):
    """SimpleMCMC.C:163-256 with BURNIN_CHAIN: a burn-in of `steps` steps, ResetProposal, four cycles with an explicit
    UpdateProposal, then the frozen-step-size settings and cycles x steps with UpdateProposal() per cycle."""
    dim, n, steps, cycles = 5, 64, 400, 3
    e, chains = _make(gpu, oracle, dim, n, 0, which=(0, 1, 40, 63))
    awin = max(100, min(1000, int(0.1 * steps)))
    _both(e, chains, "SetAcceptanceWindow", "set_acceptance_window", awin)            # :171-172
    _both(e, chains, "SetCovarianceWindow", "set_covariance_window", steps)
    _step(e, chains, steps)                                                           # :176-181
    _both(e, chains, "ResetProposal", "reset_proposal")                               # :182
    _same(e, chains, "after ResetProposal")
    _both(e, chains, "SetAcceptanceWindow", "set_acceptance_window", awin)            # :185-187
    _both(e, chains, "SetCovarianceWindow", "set_covariance_window", 2 * steps)
    _both(e, chains, "SetCovarianceUpdateDeweighting", "set_covariance_deweight", 0.5)
    for cycle in range(4):                                                            # :189-200
        _step(e, chains, steps)
        _both(e, chains, "UpdateProposal", "update_proposal")
        _same(e, chains, f"burn-in cycle {cycle}")
    _both(e, chains, "SetAcceptanceWindow", "set_acceptance_window", 1000)            # :204-208
    _both(e, chains, "SetAcceptanceRigidity", "set_acceptance_rigidity", 2.0)
    _both(e, chains, "SetCovarianceWindow", "set_covariance_window", cycles * steps)
    _both(e, chains, "SetCovarianceUpdateDeweighting", "set_covariance_deweight", 0.20)
    _both(e, chains, "SetNextUpdate", "set_next_update", 1E+9)
    for cycle in range(cycles):                                                       # :212-256
        _step(e, chains, steps)
        _both(e, chains, "UpdateProposal", "update_proposal")
        _both(e, chains, "SetAcceptanceRigidity", "set_acceptance_rigidity", 2.0)
        _both(e, chains, "SetCovarianceUpdateDeweighting", "set_covariance_deweight", 0.0)
        _both(e, chains, "SetNextUpdate", "set_next_update", 10 * steps)
        _same(e, chains, f"cycle {cycle}")
    assert e.get_param("TOTAL_STEPS") == steps * (5 + cycles)


def test_sharding_and_single_steps(gpu, oracle):
    """An engine holding chains [a, b) is that slice of the whole ensemble; one launch of k steps is k launches of one."""
    dim, n = 8, 128
    whole, chains = _make(gpu, oracle, dim, n, 0, which=(64, 127))
    part, _ = _make(gpu, oracle, dim, 64, 0, which=(), offset=64)
    for obj in (whole, part):
        obj.SetNextUpdate(10)
    for ch in chains.values():
        ch.set_next_update(10)
    whole.Step(120)
    for _ in range(120):
        part.Step(1)
    for ch in chains.values():
        ch.run_quiet(120)
    _same(whole, chains, "whole")
    assert np.array_equal(part.GetAccepted(), whole.GetAccepted()[:, 64:])
    for c in (0, 63):
        for a, b in zip(part.chain_proposal(c), whole.chain_proposal(64 + c)):
            assert np.array_equal(a, b)
    assert np.array_equal(part.lane("sigma"), whole.lane("sigma")[64:])


def test_frozen_covariance_and_metropolis_modes(gpu, oracle):
    """SetCovarianceFrozen(true) (TSimpleMCMC.H:937): the covariance loop is skipped, the centre keeps running, the
    chain's own UpdateProposal still fires; metropolis = 1 / 2 (:361-369) and ForceStep (:811-817)."""
    dim, n = 10, 64
    e, chains = _make(gpu, oracle, dim, n, 2, which=(0, 5, 63), x0=np.full(dim, 0.9))
    e.SetCovarianceFrozen(True)
    for ch in chains.values():
        ch.set_covariance_frozen(1)
    _both(e, chains, "SetNextUpdate", "set_next_update", 15)
    _step(e, chains, 250)
    _same(e, chains, "frozen covariance")
    e.SetCovarianceFrozen(False)
    for ch in chains.values():
        ch.set_covariance_frozen(0)
    _step(e, chains, 100)
    _same(e, chains, "thawed")
    _step(e, chains, 30, metropolis=1)
    _same(e, chains, "metropolis = 1")
    _step(e, chains, 5, metropolis=2)
    _same(e, chains, "metropolis = 2")
    p = np.linspace(0.8, 1.1, dim)
    e.ForceStep(p)
    for ch in chains.values():
        ch.force_step(p)
    _step(e, chains, 3)
    _same(e, chains, "forced step")


def test_the_fallback_ladder_inside_a_launch(gpu, oracle):
    """A covariance whose Cholesky pivot fails when the chain's own schedule fires: the chain stops inside the launch,
    the host runs the ladder of TSimpleMCMC.H:1134-1389 for it (here: conditioning, rung 1) and the chain catches up;
    the other chains of the wavefront never notice."""
    dim, n = 6, 64
    e, chains = _make(gpu, oracle, dim, n, 0)
    _step(e, chains, 50)
    bad = np.eye(dim)
    bad[0, 1] = bad[1, 0] = 1.0 + 1e-3                           # correlation > 1: no Cholesky factor
    e.SetCovariance(bad)
    for ch in chains.values():
        ch.set_covariance(bad)
    _both(e, chains, "SetCovarianceWindow", "set_covariance_window", 10 ** 6)   # the running average barely moves it
    _both(e, chains, "SetCovarianceTrials", "set_covariance_trials", 1e6)
    _both(e, chains, "SetNextUpdate", "set_next_update", 3)
    _step(e, chains, 60)
    paths = e.lane("last_update_path")
    assert np.all(paths >= 1), "the ladder must have run in every chain"
    _same(e, chains, "after the ladder")
    if np.any(paths == 2):                                       # the eigen rung leaves a full decomposition (:1252-1321)
        assert np.array_equal(e.lane("decomp_full") == 1, paths == 2)
    _step(e, chains, 200)
    _same(e, chains, "and on")


def test_the_fallback_ladder_for_thousands_of_chains_at_once(gpu, oracle):
    """Every chain of a 5 000-chain ensemble stops for the host's ladder in the same launches: the flagged chains go
    through the staging records in batches (4 096 + 904 here; pc_host_ladder), and the sampled chains -- first and last
    of each batch -- are still the reference chains.  (Chain by chain with strided copies this took minutes.)"""
    import time
    dim, n = 6, 5000
    e, chains = _make(gpu, oracle, dim, n, 0, which=(0, 63, 4095, 4096, 4999))
    _step(e, chains, 50)
    bad = np.eye(dim)
    bad[0, 1] = bad[1, 0] = 1.0 + 1e-3
    e.SetCovariance(bad)
    for ch in chains.values():
        ch.set_covariance(bad)
    _both(e, chains, "SetCovarianceWindow", "set_covariance_window", 10 ** 6)
    _both(e, chains, "SetCovarianceTrials", "set_covariance_trials", 1e6)
    _both(e, chains, "SetNextUpdate", "set_next_update", 3)
    t0 = time.perf_counter()
    e.Step(60)
    e.sync()
    assert time.perf_counter() - t0 < 30.0
    for ch in chains.values():
        ch.run_quiet(60)
    assert np.all(e.lane("last_update_path") >= 1), "the ladder must have run in every chain"
    _same(e, chains, "after the ladder")
    _step(e, chains, 100)
    _same(e, chains, "and on")


def test_the_fallback_ladder_on_the_last_step_of_a_launch(gpu, oracle):
    """The same ladder driven one step per launch (what TSimpleMCMC_amd.H::Step() does for a single chain): every ladder
    event then lands on the LAST step of its launch, the chain stops with its step counter already at the launch's
    target, and the relaunch has to finish that step -- proposal, likelihood, accept test -- before anything else."""
    dim, n = 6, 64
    e, chains = _make(gpu, oracle, dim, n, 0, which=(0, 1, 17, 63))
    _step(e, chains, 50)
    bad = np.eye(dim)
    bad[0, 1] = bad[1, 0] = 1.0 + 1e-3
    e.SetCovariance(bad)
    for ch in chains.values():
        ch.set_covariance(bad)
    _both(e, chains, "SetCovarianceWindow", "set_covariance_window", 10 ** 6)
    _both(e, chains, "SetCovarianceTrials", "set_covariance_trials", 1e6)
    _both(e, chains, "SetNextUpdate", "set_next_update", 3)
    for i in range(60):
        _step(e, chains, 1)
        if i % 10 == 9:
            _same(e, chains, f"one step per launch, step {i + 1}")
    assert np.all(e.lane("last_update_path") >= 1), "the ladder must have run in every chain"
    assert np.all(e.lane("chain_steps") == 110)
    _step(e, chains, 40)
    _same(e, chains, "and on")


def test_restore_continues_a_chain(gpu, oracle):
    """SaveStep(true) / Restore (TSimpleMCMC.H:282-352, 1501-1612) per chain: chain 3's saved state restored into every
    chain of a new engine = oracle.Chain.restore."""
    dim, n = 7, 64
    e, chains = _make(gpu, oracle, dim, n, 0, which=(3,))
    _both(e, chains, "SetNextUpdate", "set_next_update", 20)
    _step(e, chains, 300)
    st = e.saved_state(3)
    ref = chains[3].saved_state()
    for k in st:
        assert np.array_equal(np.asarray(st[k]), np.asarray(ref[k])), k
    e2, chains2 = _make(gpu, oracle, dim, n, 0, which=(0, 3, 63))
    e2.Restore(st)
    for ch in chains2.values():
        ch.restore(ref)
    _same(e2, chains2, "restored")
    _step(e2, chains2, 150)
    _same(e2, chains2, "continued")


def test_what_the_mode_refuses(gpu):
    e = gpu.Engine(100, 64)
    with pytest.raises(gpu.SmcmcError):
        e._check(e._lib.smcmc_set_mode(e._h, gpu.MODE_PER_CHAIN))      # dim > 63
    e = gpu.Engine(8, 64, mode=gpu.MODE_PER_CHAIN)
    e.SetUniform(2, -1.0, 1.0)
    with pytest.raises(gpu.SmcmcError) as err:
        e.Start(np.zeros(8))
    assert err.value.status == 5                                       # SMCMC_ERR_UNSUPPORTED, never a fallback
    e = gpu.Engine(8, 64, mode=gpu.MODE_PER_CHAIN, exact=False)
    with pytest.raises(gpu.SmcmcError):
        e.Start(np.zeros(8))


def test_config1_a_million_adaptive_steps(gpu, oracle):
    """BASELINE config 1: D = 5, 10^6 Step() calls of a chain with SimpleMCMC.C's defaults -- the covariance adapting
    every step, UpdateProposal on the chain's own schedule (~230 of them) -- on the device next to the CPU restatement:
    after a million steps the lanes are still bit for bit the reference chains of their chain ids, and what the chains
    estimate (fCentralPoint, fCurrentCov: running averages over the covariance window of 245 steps) is the posterior's
    mean 0 and covariance I, averaged over the 64 chains."""
    dim, n, steps = 5, 64, 1000000
    e, chains = _make(gpu, oracle, dim, n, 0, which=(0, 63))
    _step(e, chains, steps)
    _same(e, chains, "10^6 steps")
    assert np.all(e.lane("update_count") > 100)
    assert np.all(np.abs(e.lane("naccept") / steps - 0.234) < 0.03)
    centre = np.array([e.chain_proposal(c)[0] for c in range(n)])
    cov = np.array([e.chain_proposal(c)[1] for c in range(n)])
    assert np.max(np.abs(centre.mean(axis=0))) < 0.2            # 64 windows of ~10 independent points each
    assert np.max(np.abs(cov.mean(axis=0) - np.eye(dim))) < 0.25


def test_the_full_ensemble_sampled_lanes(gpu, oracle):
    """BASELINE config 2 at its full size in this mode: 65 536 chains (1 024 wavefront tiles of covariance and
    decomposition, 2.1 GB), a few hundred steps through an UpdateProposal of every chain; lanes from the first, a middle
    and the last tile against their reference chains.  Chains are independent here, so sampled lanes are the whole check."""
    dim, n = 50, 65536
    which = (0, 63, 64, 32767, 32768, 65471, 65472, 65535)
    e, chains = _make(gpu, oracle, dim, n, 0, which=which,
                      setup=lambda o: (o.SetAcceptanceWindow if hasattr(o, "SetAcceptanceWindow") else o.set_acceptance_window)(60))
    _both(e, chains, "SetNextUpdate", "set_next_update", 40)
    _step(e, chains, 400)
    _same(e, chains, "full ensemble")
    assert np.all(e.lane("update_count") >= 2)
    assert np.all(e.lane("chain_steps") == 400)
    e.close()


def test_step_recorded_is_the_chain_step_by_step(gpu, oracle, which_kernel):
    """smcmc_step_recorded: one launch of many steps leaves, for one chain, what TSimpleMCMC::Step() shows after EVERY
    step -- fAccepted, fProposed, both likelihoods, StepRMS, the accept flag, the Adaptive* scalars and the covariance
    trace SaveStep writes -- equal to oracle.Chain stepped one call at a time, UpdateProposal events included."""
    dim, n, nsteps, c = 9, 5, 400, 3
    e, chains = _make(gpu, oracle, dim, n, 0, which=(c,))
    _both(e, chains, "SetNextUpdate", "set_next_update", 7)
    if not which_kernel:
        with pytest.raises(gpu.SmcmcError) as err:
            e.StepRecorded(4, chain=c)
        assert err.value.status == 5                                   # SMCMC_ERR_UNSUPPORTED: the record is the wave kernel's
        return
    rec = e.StepRecorded(nsteps, chain=c)
    ch = chains[c]
    names = {"logl": "accepted_logl", "logl_proposed": "proposed_logl", "step_rms": "step_rms", "trials": "trials",
             "successes": "successes", "next_update": "next_update", "acceptance": "acceptance",
             "acceptance_trials": "acceptance_trials", "sigma": "sigma", "center_trials": "central_trials",
             "covariance_trials": "cov_trials", "total_steps": "total_steps"}
    updates = 0
    for s in range(nsteps):
        moved = ch.step(False, 0)
        sc = ch.scalars
        assert np.array_equal(rec["accepted"][s], ch.accepted), f"step {s}: accepted"
        assert np.array_equal(rec["proposed"][s], ch.proposed), f"step {s}: proposed"
        assert bool(rec["last_accept"][s]) == bool(moved), f"step {s}: accept flag"
        for k, ok in names.items():
            assert rec[k][s] == sc[ok], f"step {s}: {k} = {rec[k][s]!r}, reference chain {sc[ok]!r}"
        assert rec["covariance_trace"][s] == np.add.accumulate(np.diag(ch.covariance))[-1], f"step {s}: trace"
        updates = int(sc["update_count"])
    assert updates >= 2
    _same(e, chains, "after the recorded launch")
