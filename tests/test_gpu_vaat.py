"""TProposeVAATStep on the device: N independent TSimpleMCMC<L, TProposeVAATStep> chains through the
C ABI against oracle/vaat_oracle.c, bit for bit -- positions, likelihoods, the per-dimension widths / acceptances /
trial counts, the index queue, the step counters."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(gpu, oracle, dim, nchains, kind, exact=True, params=None, offset=0):
    prm = oracle.like_params(kind, dim, params)
    prm = prm if prm.size else None
    e = gpu.VaatEngine(dim, nchains, likelihood=kind, likelihood_params=prm, seed=11, chain_offset=offset, exact=exact)
    o = oracle.Vaat(nchains, dim, kind=kind, params=prm, seed=11, chain_offset=offset, exact=exact)
    return e, o


def _same(e, o, tag):
    assert np.array_equal(e.GetAccepted(), o.x), f"{tag}: accepted points differ"
    for name in ("logl", "logl_proposed", "step_rms", "proposed_value"):
        a, b = e.lane(name), o.lane(name)
        assert np.array_equal(a, b), f"{tag}: {name} differs (max |d| = {np.max(np.abs(a - b))})"
    for name in ("trials", "successes", "naccept", "last_accept", "last_index", "step_rms_trials"):
        assert np.array_equal(e.lane(name), o.lane(name)), f"{tag}: {name} differs"
    for name in ("sigma", "acceptance", "acceptance_trials"):
        assert np.array_equal(e.per_dim(name), o.per_dim(name)), f"{tag}: per-dimension {name} differs"
    qlen = e.queue_length
    assert np.all(o.lane("queue_len") == qlen), f"{tag}: queue length"
    assert np.array_equal(e.per_dim("queue")[:qlen], o.per_dim("queue")[:qlen]), f"{tag}: index queue differs"


def _start(kind, dim, n, rng):
    if kind == 2:
        return rng.uniform(0.5, 1.5, size=(dim, n))
    if kind == 5:
        return rng.uniform(-0.9, 0.9, size=(dim, n))
    if kind == 6:
        return 76.0 + rng.normal(0.0, 1.0, size=(dim, n))
    return rng.uniform(-1.0, 1.0, size=(dim, n))                     # SimpleVAAT.C:41


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("kind,dim", [(0, 5), (0, 63), (1, 12), (1, 50), (2, 6), (2, 31), (4, 20), (5, 30), (6, 25)])
def test_vaat_small_dimensions(gpu, oracle, kind, dim, exact):
    n = 130
    rng = np.random.default_rng(kind * 100 + dim)
    e, o = _pair(gpu, oracle, dim, n, kind, exact)
    x0 = _start(kind, dim, n, rng)
    assert e.Start(x0) and o.start(x0)
    assert e.GetAcceptanceWindow() == 100 and o.acceptance_window == 100      # InitializeState :211
    _same(e, o, "start")
    e.UpdateProposal(); o.update_proposal()                                   # SimpleVAAT.C:44
    _same(e, o, "after the explicit UpdateProposal")
    done = 0
    for chunk in (1, 2, dim - 1, 3 * dim + 1, 900):                           # launches cut inside and across queue refills
        e.Step(chunk); o.step(chunk)
        done += chunk
        _same(e, o, f"after {done} steps")
    assert e.lane("naccept").sum() > 0
    assert (e.per_dim("acceptance_trials") > 10).all()                        # the widths are adapting (:245)
    assert not np.all(e.per_dim("sigma") == 2.34)


@pytest.mark.parametrize("exact", [True, False])
@pytest.mark.parametrize("kind,dim", [(0, 64), (1, 100), (2, 200), (4, 100), (5, 75), (0, 500), (6, 100)])
def test_vaat_large_dimensions(gpu, oracle, kind, dim, exact):
    """dim = 100 with the header-form TDummyLogLikelihood is SimpleVAAT.C's own configuration."""
    n = 70
    rng = np.random.default_rng(kind * 1000 + dim)
    e, o = _pair(gpu, oracle, dim, n, kind, exact)
    x0 = _start(kind, dim, n, rng)
    assert e.Start(x0) and o.start(x0)
    e.UpdateProposal(); o.update_proposal()
    for chunk in (1, dim, dim + 7):
        e.Step(chunk); o.step(chunk)
        _same(e, o, f"chunk {chunk}")
    assert e.lane("naccept").sum() > 0


def test_vaat_without_the_explicit_update_proposal(gpu, oracle):
    """Without SimpleVAAT.C:44 the first Step() shuffles (with that step's draws)."""
    e, o = _pair(gpu, oracle, 9, 64, 0)
    assert e.Start(np.zeros(9)) and o.start(np.zeros(9))
    e.Step(25); o.step(25)
    _same(e, o, "25 steps")
    e.UpdateProposal(); o.update_proposal()                                   # queue not empty: nothing happens (:178)
    _same(e, o, "no-op UpdateProposal")


@pytest.mark.parametrize("dim", [9, 100])
def test_vaat_launches_of_whole_queues(gpu, oracle, dim):
    """SimpleVAAT.C's own call sequence: Start, UpdateProposal (a full queue), then launches whose step count is a
    multiple of the dimension.  The host's copy of the queue length has to come out 0, not dim, or the next launch
    replays the old permutation."""
    n = 70
    e, o = _pair(gpu, oracle, dim, n, 0)
    rng = np.random.default_rng(dim)
    x0 = rng.uniform(-1.0, 1.0, size=(dim, n))
    assert e.Start(x0) and o.start(x0)
    e.UpdateProposal(); o.update_proposal()
    assert e.queue_length == dim
    e.Step(2 * dim); o.step(2 * dim)
    assert e.queue_length == 0
    _same(e, o, "UpdateProposal; Step(2 dim)")
    e.Step(5); o.step(5)
    _same(e, o, "... Step(5)")
    e2, o2 = _pair(gpu, oracle, dim, n, 0)
    assert e2.Start(x0) and o2.start(x0)
    for k in range(4):                                                        # Step(dim) repeated, no explicit UpdateProposal
        e2.Step(dim); o2.step(dim)
        assert e2.queue_length == 0
        _same(e2, o2, f"Step(dim) number {k + 1}")
    e2.UpdateProposal(); o2.update_proposal()                                 # empty queue: refilled now (:178)
    assert e2.queue_length == dim
    e2.Step(dim); o2.step(dim)
    _same(e2, o2, "UpdateProposal on an empty queue, Step(dim)")


def test_vaat_second_start_keeps_the_proposal_state(gpu, oracle):
    """A second Start() moves the chain and recomputes its likelihood; InitializeState returns at once
    (TProposeVAATStep.H:197), so fLastValue, fStepRMS, the widths and the counters stay as they were."""
    dim, n = 8, 64
    e, o = _pair(gpu, oracle, dim, n, 0)
    assert e.Start(np.zeros(dim)) and o.start(np.zeros(dim))
    e.Step(150); o.step(150)
    last_value, rms = e.lane("last_value").copy(), e.lane("step_rms").copy()
    assert e.Start(np.full(dim, 0.3)) and o.start(np.full(dim, 0.3))
    assert np.array_equal(e.lane("last_value"), last_value) and np.array_equal(e.lane("step_rms"), rms)
    _same(e, o, "second Start")
    e.Step(60); o.step(60)
    _same(e, o, "steps after the second Start")


def test_vaat_proposal_settings(gpu, oracle):
    """SetUniform / SetGaussian (:101-133), the acceptance window and rigidity (:137-148), the StepRMS window."""
    dim, n = 7, 96
    e, o = _pair(gpu, oracle, dim, n, 0)
    for obj, uni, gau, win, rig, rms in ((e, e.SetUniform, e.SetGaussian, e.SetAcceptanceWindow, e.SetAcceptanceRigidity,
                                          e.SetStepRMSWindow),
                                         (o, o.set_uniform, o.set_gaussian, o.set_acceptance_window,
                                          o.set_acceptance_rigidity, o.set_step_rms_window)):
        uni(1, -0.5, 0.5)                                                     # SimpleVAAT.C:32 (commented there)
        gau(3, 0.25)
        win(37.9)                                                             # before Start: overwritten by InitializeState
        rms(50)
    assert e.Start(np.full(dim, 0.1)) and o.start(np.full(dim, 0.1))
    assert e.GetAcceptanceWindow() == 100
    e.Step(300); o.step(300)
    _same(e, o, "defaults restored by Start")
    e.SetAcceptanceWindow(37.9); o.set_acceptance_window(37.9)                # an int member: 37
    e.SetAcceptanceRigidity(0.7); o.set_acceptance_rigidity(0.7)
    assert e.GetAcceptanceWindow() == 37 and e.GetAcceptanceRigidity() == 0.7
    e.Step(300); o.step(300)
    _same(e, o, "window 37, rigidity 0.7")
    x = e.GetAccepted()
    assert np.all(np.abs(x[1]) <= 0.5)                                        # the uniform dimension stays in its range
    e.SetAcceptanceRigidity(-1.0); o.set_acceptance_rigidity(-1.0)            # fixed widths (:144-146)
    before = e.per_dim("sigma").copy()
    e.Step(100); o.step(100)
    _same(e, o, "negative rigidity")
    assert np.array_equal(before, e.per_dim("sigma"))
    with pytest.raises(gpu.SmcmcError):
        e.SetUniform(dim, 0.0, 1.0)                                           # out of range (:102-107)


def test_vaat_chain_offset_and_sharding(gpu, oracle):
    """Chains are independent: an engine holding chains [a, b) is that slice of the whole ensemble."""
    dim, n = 10, 192
    whole, o = _pair(gpu, oracle, dim, n, 2)
    rng = np.random.default_rng(3)
    x0 = rng.uniform(0.5, 1.5, size=(dim, n))
    assert whole.Start(x0) and o.start(x0)
    part, _ = _pair(gpu, oracle, dim, 64, 2, offset=128)
    assert part.Start(np.ascontiguousarray(x0[:, 128:]))
    whole.Step(150); part.Step(150); o.step(150)
    _same(whole, o, "whole")
    assert np.array_equal(part.GetAccepted(), whole.GetAccepted()[:, 128:])
    assert np.array_equal(part.per_dim("sigma"), whole.per_dim("sigma")[:, 128:])


def test_vaat_saves_inside_a_launch(gpu, oracle):
    import torch
    dim, n, steps, stride = 6, 100, 40, 4
    e, o = _pair(gpu, oracle, dim, n, 0)
    assert e.Start(np.zeros(dim)) and o.start(np.zeros(dim))
    np_ = e.nchains_padded
    sx = torch.zeros((steps // stride, dim, np_), dtype=torch.float64, device="cuda")
    sl = torch.zeros((steps // stride, np_), dtype=torch.float64, device="cuda")
    e.step_save(steps, stride, sx.data_ptr(), sl.data_ptr())
    torch.cuda.synchronize()
    got_x, got_l = sx.cpu().numpy(), sl.cpu().numpy()
    for slot in range(steps // stride):
        o.step(stride)
        assert np.array_equal(got_x[slot][:, :n], o.x), slot
        assert np.array_equal(got_l[slot][:n], o.lane("logl")), slot
    _same(e, o, "end")


def test_vaat_posterior_known_answer(gpu):
    """The sampler samples: a correlated Gaussian (precision = example4's form) recovered by 4096 chains."""
    dim, n = 8, 4096
    s = np.array([6.08] * (dim - 1) + [2.0])
    prec = np.diag(1.0 / s ** 2) + np.ones((dim, dim)) / 16.0 ** 2
    cov = np.linalg.inv(prec)
    e = gpu.VaatEngine(dim, n, likelihood=1, likelihood_params=prec, seed=5, exact=False)
    assert e.Start(np.zeros(dim))
    e.Step(6000)                                                              # widths adapt, chains forget the start
    draws = []
    for _ in range(10):
        e.Step(400)
        draws.append(e.GetAccepted().copy())
    x = np.concatenate(draws, axis=1)
    sd = np.sqrt(np.diag(cov))
    assert np.all(np.abs(x.mean(axis=1)) < 5.0 * sd / np.sqrt(n))
    got = np.cov(x)
    assert np.max(np.abs(got - cov) / np.sqrt(np.outer(np.diag(cov), np.diag(cov)))) < 0.06
    acc = e.per_dim("acceptance")
    assert 0.3 < acc.mean() < 0.7        # drifting to the 44 % target (:30) at pow(., 1/500) per visit (:249-252)


def test_vaat_unsupported_and_invalid(gpu):
    with pytest.raises(gpu.SmcmcError) as err:
        gpu.VaatEngine(64, 64, likelihood=3)                                   # USER: not in the plain library
    assert err.value.status == 5
    with pytest.raises(gpu.SmcmcError) as err:
        gpu.VaatEngine(600, 64)
    assert err.value.status == 5
    e = gpu.VaatEngine(4, 64)
    with pytest.raises(gpu.SmcmcError):
        e.Step(1)                                                             # "Must initialize starting point"
    assert e.Start(np.array([np.nan, 0, 0, 0])) is False                      # TSimpleMCMC.H:265-268
