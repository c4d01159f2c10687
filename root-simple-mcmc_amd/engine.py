"""Host-side mirror of the reference's operator interface for the Step() path.

`Engine` is the many-chain counterpart of sMCMC::TSimpleMCMC<L, TProposeAdaptiveStep>
(reference TSimpleMCMC.H:185-590): the same verbs (Start, Step, SaveStep-style
read back, GetProposeStep()-style setters/getters named as in TSimpleMCMC.H:732-1003)
on top of the C ABI of include/smcmc.h.  All compute happens in the HIP library.
"""
import ctypes as C

import numpy as np

from . import _capi
from ._capi import (LIKE_ASYM, LIKE_CONSTRAINED, LIKE_HORRIFIC, LIKE_ISO_GAUSS, LIKE_QUADFORM, LIKE_ROSENBROCK, LIKE_USER, MODE_FROZEN, MODE_PER_CHAIN, MODE_POOLED, P,  # noqa: F401
                    SmcmcError)

_dp = C.POINTER(C.c_double)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ptr(a):
    return a.ctypes.data_as(_dp)


class Engine:
    """N chains of dimension D advancing in lock step on one GPU."""

    def __init__(self, dim, nchains=1, likelihood=LIKE_ISO_GAUSS, likelihood_params=None, seed=20240607,
                 chain_offset=0, device=0, mode=MODE_POOLED, exact=True, stream=None, library=None):
        # library: path of a build that carries a user likelihood (LIKE_USER), see build.py --user-likelihood
        self._lib = _capi.load(library)
        self.dim, self.nchains = int(dim), int(nchains)
        h = C.c_void_p()
        st = self._lib.smcmc_create(self.dim, self.nchains, likelihood, seed, chain_offset, device, C.byref(h))
        self._h = h
        if st != _capi.OK:
            msg = self._lib.smcmc_last_error(h).decode() if h else self._lib.smcmc_status_string(st).decode()
            if h:
                self._lib.smcmc_destroy(h)
            self._h = None
            raise SmcmcError(st, msg)
        self._check(self._lib.smcmc_set_mode(self._h, mode))
        self.mode = mode
        self.set_param("EXACT_ARITHMETIC", 1.0 if exact else 0.0)
        if likelihood_params is not None:
            prm = _f64(likelihood_params).ravel()
            self._check(self._lib.smcmc_set_likelihood_params(self._h, _ptr(prm), prm.size))
        if stream is not None:
            self.set_stream(stream)

    # -- plumbing ---------------------------------------------------------
    def _check(self, st):
        if st != _capi.OK:
            raise SmcmcError(st, self._lib.smcmc_last_error(self._h).decode()
                             or self._lib.smcmc_status_string(st).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.smcmc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        """stream: a raw hipStream_t value (e.g. torch.cuda.current_stream().cuda_stream)."""
        self._check(self._lib.smcmc_set_stream(self._h, C.c_void_p(int(stream))))

    @property
    def nchains_padded(self):
        return self._lib.smcmc_nchains_padded(self._h)

    @property
    def dim_padded(self):
        return self._lib.smcmc_dim_padded(self._h)

    # -- GetProposeStep() surface (TSimpleMCMC.H:732-1003) ------------------
    def set_param(self, name, value):
        self._check(self._lib.smcmc_set_param(self._h, P[name], float(value)))

    def get_param(self, name):
        out = C.c_double(0)
        self._check(self._lib.smcmc_get_param(self._h, P[name], C.byref(out)))
        return out.value

    def SetGaussian(self, dim, sigma):
        self._check(self._lib.smcmc_set_gaussian(self._h, dim, sigma))

    def SetUniform(self, dim, minimum, maximum):
        self._check(self._lib.smcmc_set_uniform(self._h, dim, minimum, maximum))

    def SetScanDimension(self, dim):
        self._check(self._lib.smcmc_set_scan_dimension(self._h, dim))

    def SetCorrelation(self, dim1, dim2, correlation):
        self._check(self._lib.smcmc_set_correlation(self._h, dim1, dim2, correlation))

    def ResetCorrelations(self):
        self._check(self._lib.smcmc_reset_correlations(self._h))

    def SetCovarianceWindow(self, w): self.set_param("COVARIANCE_WINDOW", w)
    def GetCovarianceWindow(self): return self.get_param("COVARIANCE_WINDOW")
    def SetCovarianceUpdateDeweighting(self, d): self.set_param("COVARIANCE_DEWEIGHT", d)
    def SetAcceptanceWindow(self, w): self.set_param("ACCEPTANCE_WINDOW", w)
    def GetAcceptanceWindow(self): return self.get_param("ACCEPTANCE_WINDOW")
    def SetAcceptanceUpdateDeweighting(self, d): self.set_param("ACCEPTANCE_DEWEIGHT", d)
    def SetAcceptanceRigidity(self, r): self.set_param("ACCEPTANCE_RIGIDITY", r)
    def GetAcceptanceRigidity(self): return self.get_param("ACCEPTANCE_RIGIDITY")
    def SetTargetAcceptance(self, a): self.set_param("TARGET_ACCEPTANCE", a)
    def GetTargetAcceptance(self): return self.get_param("TARGET_ACCEPTANCE")
    def SetSigma(self, s): self.set_param("SIGMA", s)
    def GetSigma(self, chain=None):
        """fSigma (TSimpleMCMC.H:770): of chain 0 by default, of `chain` otherwise."""
        return self.get_param("SIGMA") if chain is None else float(self.lane("sigma")[chain])
    def SetMaximumCorrelation(self, c): self.set_param("MAXIMUM_CORRELATION", c)
    def SetStepRMSWindow(self, n): self.set_param("STEP_RMS_WINDOW", n)
    def SetNextUpdate(self, n): self.set_param("NEXT_UPDATE", n)
    def GetNextUpdate(self): return self.get_param("NEXT_UPDATE")
    def GetCovarianceTrials(self): return self.get_param("COVARIANCE_TRIALS")
    def SetCovarianceTrials(self, v): self.set_param("COVARIANCE_TRIALS", v)
    def GetEstimatedCenterTrials(self): return self.get_param("CENTER_TRIALS")
    def SetEstimatedCenterTrials(self, v): self.set_param("CENTER_TRIALS", v)
    def GetCovarianceTrace(self): return self.get_param("COVARIANCE_TRACE")

    def UpdateProposal(self):
        self._check(self._lib.smcmc_update_proposal(self._h))

    def ResetProposal(self):
        self._check(self._lib.smcmc_reset_proposal(self._h))

    def ForceStep(self, point):
        point = _f64(point)
        self._check(self._lib.smcmc_force_step(self._h, _ptr(point), int(point.ndim == 1)))

    def GetEstimatedCenter(self):
        out = np.zeros(self.dim)
        self._check(self._lib.smcmc_get_center(self._h, _ptr(out)))
        return out

    def SetEstimatedCenter(self, v):
        v = _f64(v)
        if v.shape != (self.dim,):
            return False
        self._check(self._lib.smcmc_set_center(self._h, _ptr(v)))
        return True

    # -- TSimpleMCMC surface (TSimpleMCMC.H:246-532) ------------------------
    def Start(self, start):
        """start: [dim] (every chain) or [dim][nchains].  False = bad start (:265-268)."""
        start = _f64(start)
        broadcast = int(start.ndim == 1)
        if not broadcast and start.shape != (self.dim, self.nchains):
            raise ValueError("start must be [dim] or [dim][nchains]")
        st = self._lib.smcmc_start(self._h, _ptr(start), broadcast)
        if st == _capi.ERR_BAD_START:
            return False
        self._check(st)
        return True

    def SetCovarianceFrozen(self, frozen=True):
        """SetCovarianceFrozen (TSimpleMCMC.H:937) inside MODE_PER_CHAIN (MODE_FROZEN is the shared-decomposition form)."""
        self.set_param("COVARIANCE_FROZEN", 1.0 if frozen else 0.0)

    def chain(self, c=0):
        """One chain's members (smcmc_read_chain): dict with accepted, proposed (when kept) and every lane by name."""
        x, lf = np.zeros(self.dim), np.zeros(len(_capi.LANE_F64))
        li = np.zeros(len(_capi.LANE_I32), np.int32)
        keep = self.get_param("KEEP_PROPOSED") != 0.0
        prop = np.zeros(self.dim) if keep else None
        self._check(self._lib.smcmc_read_chain(self._h, int(c), _ptr(x), _ptr(prop) if keep else None, _ptr(lf),
                                               li.ctypes.data_as(C.POINTER(C.c_int32))))
        out = dict(accepted=x, proposed=prop)
        out.update({k: float(lf[i]) for k, i in _capi.LANE_F64.items()})
        out.update({k: int(li[i]) for k, i in _capi.LANE_I32.items()})
        return out

    def chain_proposal(self, c=0):
        """(fCentralPoint, fCurrentCov, fDecomposition) of chain c (MODE_PER_CHAIN: its own; else the shared ones)."""
        centre, cov, dec = np.zeros(self.dim), np.zeros((self.dim, self.dim)), np.zeros((self.dim, self.dim))
        self._check(self._lib.smcmc_read_chain_proposal(self._h, int(c), _ptr(centre), _ptr(cov), _ptr(dec)))
        return centre, cov, dec

    def saved_state(self, chain=0):
        """What SaveStep(true) writes for one chain (branches of TSimpleMCMC.H:208-215, 1616-1626):
        its point and scalar state plus the centre and covariance (the chain's own in MODE_PER_CHAIN, else shared)."""
        if self.mode == MODE_PER_CHAIN:
            ch = self.chain(chain)
            centre, cov, _ = self.chain_proposal(chain)
            return dict(accepted=ch["accepted"], log_likelihood=ch["logl"], total_steps=ch["chain_steps"],
                        step_rms=ch["step_rms"], trials=ch["trials"], successes=ch["successes"],
                        next_update=ch["next_update"], acceptance=ch["acceptance"],
                        acceptance_trials=ch["acceptance_trials"], sigma=ch["sigma"], central_point=centre,
                        central_point_trials=ch["center_trials"],
                        covariance=np.array([cov[i, j] for i in range(self.dim) for j in range(i + 1)]),
                        covariance_trials=ch["covariance_trials"])
        cov = self.covariance
        return dict(accepted=self.GetAccepted()[:, chain].copy(),
                    log_likelihood=float(self.GetAcceptedLogLikelihood()[chain]),
                    total_steps=int(self.get_param("TOTAL_STEPS")), step_rms=float(self.lane("step_rms")[chain]),
                    trials=int(self.lane("trials")[chain]), successes=int(self.lane("successes")[chain]),
                    next_update=int(self.lane("next_update")[chain]),
                    acceptance=float(self.lane("acceptance")[chain]),
                    acceptance_trials=float(self.lane("acceptance_trials")[chain]),
                    sigma=float(self.lane("sigma")[chain]), central_point=self.GetEstimatedCenter(),
                    central_point_trials=self.get_param("CENTER_TRIALS"),
                    covariance=np.array([cov[i, j] for i in range(self.dim) for j in range(i + 1)]),
                    covariance_trials=self.get_param("COVARIANCE_TRIALS"))

    def Restore(self, state, accepted=None):
        """Restore(tree) (TSimpleMCMC.H:282-352) from a saved_state() dict; `accepted` ([dim] or
        [dim][nchains]) overrides state["accepted"] as the point(s) to continue from."""
        x = _f64(state["accepted"] if accepted is None else accepted)
        broadcast = int(x.ndim == 1)
        if not broadcast and x.shape != (self.dim, self.nchains):
            raise ValueError("accepted must be [dim] or [dim][nchains]")
        centre, cov = _f64(state["central_point"]), _f64(state["covariance"])
        if centre.size != self.dim or cov.size != self.dim * (self.dim + 1) // 2:
            raise ValueError("saved centre / packed covariance have the wrong size")
        st = _capi.SavedState(state["log_likelihood"], state["total_steps"], state["step_rms"], state["trials"],
                              state["successes"], state["next_update"], state["acceptance"],
                              state["acceptance_trials"], state["sigma"], _ptr(centre),
                              state["central_point_trials"], _ptr(cov), state["covariance_trials"])
        self._check(self._lib.smcmc_restore(self._h, _ptr(x), broadcast, C.byref(st)))

    def Step(self, nsteps=1, metropolis=0):
        """nsteps x Step(save=false, metropolis) of every chain, one launch."""
        self._check(self._lib.smcmc_step(self._h, int(nsteps), int(metropolis)))

    def snapshot(self): self._check(self._lib.smcmc_snapshot(self._h))
    def rollback(self): self._check(self._lib.smcmc_rollback(self._h))

    def StepRecorded(self, nsteps, chain=0, metropolis=0):
        """nsteps x Step(false) in one launch with the per-step record of one chain (smcmc_step_recorded): a dict of
        arrays over the steps -- "accepted" / "proposed" [step][dim] and the scalars of smcmc_record_field."""
        stride = self._lib.smcmc_record_stride(self._h)
        rec = np.zeros((int(nsteps), stride))
        self._check(self._lib.smcmc_step_recorded(self._h, int(nsteps), int(metropolis), int(chain), _ptr(rec)))
        out = {"accepted": rec[:, :self.dim].copy(), "proposed": rec[:, self.dim:2 * self.dim].copy(),
               "covariance_diagonal": rec[:, 2 * self.dim:3 * self.dim].copy()}
        for k, name in enumerate(_capi.RECORD_FIELDS):
            out[name] = rec[:, 3 * self.dim + k].copy()
        # GetCovarianceTrace (TSimpleMCMC.H:961-967): the diagonal added up in index order
        out["covariance_trace"] = np.add.accumulate(out["covariance_diagonal"], axis=1)[:, -1]
        return out

    def StepSave(self, nsteps, save_x_ptr, save_logl_ptr, stride=1, metropolis=0):
        """As Step, also writing the accepted points into device buffers (raw pointers)."""
        self._check(self._lib.smcmc_step_save(self._h, int(nsteps), int(metropolis), int(stride),
                                              C.c_void_p(int(save_x_ptr)), C.c_void_p(int(save_logl_ptr))))

    def AutocorrelationSums(self, trace_ptr, nslots, centre=None, stream=0):
        """Lagged-product sums of a trace StepSave wrote ([slot][dim_padded][nchains_padded] on the device), pooled
        over slots and chains about `centre` (default: the origin, as the macro has it): the inputs of the reference's
        autocorrelation (MakeAutocorrelation.C:108-148).  Returns an Autocorrelation."""
        c = _f64(np.zeros(self.dim) if centre is None else centre)
        s = np.zeros(self.dim)
        lagged = np.zeros((_capi.AUTOCORR_LAGS, self.dim))
        self._check(self._lib.smcmc_autocorrelation_sums(C.c_void_p(int(trace_ptr)), int(nslots), self.dim,
                                                         self.dim_padded, self.nchains, self.nchains_padded, _ptr(c),
                                                         _ptr(s), _ptr(lagged), C.c_void_p(int(stream))))
        return Autocorrelation(s, lagged, int(nslots), self.nchains)

    def GetAccepted(self):
        x = np.zeros((self.dim, self.nchains))
        self._check(self._lib.smcmc_read_state(self._h, _ptr(x), None))
        return x

    def KeepProposed(self, on=True):
        """Leave the proposal of every launch's last step on the device for GetProposed()."""
        self.set_param("KEEP_PROPOSED", 1.0 if on else 0.0)

    def GetProposed(self):
        """fProposed (TSimpleMCMC.H:514) of every chain, [dim][nchains]; needs KeepProposed()."""
        x = np.zeros((self.dim, self.nchains))
        self._check(self._lib.smcmc_read_proposed(self._h, _ptr(x)))
        return x

    def GetAcceptedLogLikelihood(self):
        return self.lane("logl")

    def GetProposedLogLikelihood(self):
        return self.lane("logl_proposed")

    def GetStepRMS(self):
        return self.lane("step_rms")

    def lane(self, name):
        if name in _capi.LANE_F64:
            out = np.zeros(self.nchains)
            self._check(self._lib.smcmc_read_lane_f64(self._h, _capi.LANE_F64[name], _ptr(out)))
            return out
        out = np.zeros(self.nchains, np.int32)
        self._check(self._lib.smcmc_read_lane_i32(self._h, _capi.LANE_I32[name],
                                                  out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    # -- pooled adaptation ---------------------------------------------------
    @property
    def moments_size(self):
        return self._lib.smcmc_moments_size(self._h)

    def reduce_moments(self):
        self._check(self._lib.smcmc_reduce_moments(self._h))

    def export_moments(self, dst_device_ptr):
        self._check(self._lib.smcmc_export_moments(self._h, C.c_void_p(int(dst_device_ptr))))

    def import_moments(self, src_device_ptr):
        self._check(self._lib.smcmc_import_moments(self._h, C.c_void_p(int(src_device_ptr))))

    def apply_moments(self):
        self._check(self._lib.smcmc_apply_moments(self._h))

    # native RCCL communicator (include/smcmc.h): rank 0 makes the id, the caller hands it to the other ranks
    @staticmethod
    def comm_unique_id(library=None):
        buf = C.create_string_buffer(_capi.COMM_ID_BYTES)
        st = _capi.load(library).smcmc_comm_unique_id(buf)
        if st != _capi.OK:
            raise SmcmcError(st, "smcmc_comm_unique_id failed (is librccl.so loadable?)")
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        buf = C.create_string_buffer(bytes(unique_id), _capi.COMM_ID_BYTES)
        self._check(self._lib.smcmc_comm_init(self._h, buf, int(rank), int(nranks)))

    def comm_ranks(self):
        """Ranks of the attached RCCL communicator as RCCL reports them (ncclCommCount); 0 without one."""
        return self._lib.smcmc_comm_ranks(self._h)

    def comm_destroy(self):
        self._check(self._lib.smcmc_comm_destroy(self._h))

    def allreduce_moments(self):
        self._check(self._lib.smcmc_allreduce_moments(self._h))

    def sync(self):
        self._check(self._lib.smcmc_sync(self._h))

    def read_moments(self):
        out = np.zeros(self.moments_size)
        self._check(self._lib.smcmc_read_moments(self._h, _ptr(out)))
        return out

    @property
    def covariance(self):
        out = np.zeros((self.dim, self.dim))
        self._check(self._lib.smcmc_get_covariance(self._h, _ptr(out)))
        return out

    def SetCovariance(self, cov):
        """Overwrite fCurrentCov ([dim][dim]); takes effect at the next UpdateProposal (the hook the reference's
        IMPOSE_RANDOM_CORRELATIONS experiment uses through SetCorrelation, SimpleMCMC.C:107-115)."""
        cov = _f64(cov)
        if cov.shape != (self.dim, self.dim):
            raise ValueError("covariance must be [dim][dim]")
        self._check(self._lib.smcmc_set_covariance(self._h, _ptr(cov)))

    @property
    def decomposition(self):
        out = np.zeros((self.dim, self.dim))
        self._check(self._lib.smcmc_get_decomposition(self._h, _ptr(out)))
        return out

    def state_device_ptr(self):
        x, l = C.c_void_p(), C.c_void_p()
        self._check(self._lib.smcmc_state_device_ptr(self._h, C.byref(x), C.byref(l)))
        return x.value, l.value


class Autocorrelation:
    """a(lag) = (E[x_t x_(t-lag)] - mean^2) / var per dimension (MakeAutocorrelation.C:127-139) from the pooled sums
    of Engine.AutocorrelationSums; sums of several ranks add (`+`)."""

    def __init__(self, total, lagged, nslots, nchains):
        self.sum, self.lagged, self.nslots, self.nchains = np.array(total), np.array(lagged), nslots, nchains

    def __add__(self, other):
        if self.nslots != other.nslots:
            raise ValueError("traces of different lengths do not pool")
        return Autocorrelation(self.sum + other.sum, self.lagged + other.lagged, self.nslots, self.nchains + other.nchains)

    def rho(self):
        """[lag][dim]"""
        nlag = min(self.lagged.shape[0], self.nslots)
        n = (self.nslots - np.arange(nlag))[:, None] * float(self.nchains)
        mean = self.sum / n[0]
        var = self.lagged[0] / n[0] - mean * mean
        return (self.lagged[:nlag] / n - mean * mean) / var

    def tau(self):
        """Integrated autocorrelation time per dimension in slots: 1 + 2 sum rho, the sum cut where a pair of
        consecutive lags turns negative (initial positive sequence)."""
        rho = self.rho()
        out = np.ones(rho.shape[1])
        for d in range(rho.shape[1]):
            for k in range(1, rho.shape[0] - 1, 2):
                pair = rho[k, d] + rho[k + 1, d]
                if pair < 0:
                    break
                out[d] += 2.0 * pair
        return out


class PosteriorMoments:
    """Posterior mean / covariance of everything the ensemble visits (the reducers of
    MakeCovariance.C:63-89), fed by the pooled moment sums the device already forms:

        e.Step(window); e.reduce_moments(); acc.add(e); e.apply_moments()

    Each window's packed vector M holds n, sum(x - c0) and sum (x - c0)(x - c0)^T about the centre c0 the
    window ran with; they are re-centred on zero here and added up on the host (a few KB per window)."""

    def __init__(self, dim):
        self.dim = dim
        self.n = 0.0
        self.s1 = np.zeros(dim)
        self.s2 = np.zeros((dim, dim))

    def add(self, engine):
        D = self.dim
        m = engine.read_moments()
        c0 = engine.GetEstimatedCenter()
        tri = np.zeros((D + 1, D + 1))
        tri[np.tril_indices(D + 1)] = m
        n, s1 = tri[D, D], tri[D, :D]
        s2 = tri[:D, :D] + np.tril(tri[:D, :D], -1).T
        self.n += n
        self.s1 += s1 + n * c0
        self.s2 += s2 + np.outer(c0, s1) + np.outer(s1, c0) + n * np.outer(c0, c0)

    @property
    def mean(self):
        return self.s1 / self.n

    @property
    def covariance(self):
        mu = self.mean
        return self.s2 / self.n - np.outer(mu, mu)


def selftest_detmath(kind, x, y=None, device=0):
    lib = _capi.load()
    x = _f64(x)
    out = np.empty_like(x)
    yy = _f64(y) if y is not None else None
    st = lib.smcmc_selftest_detmath(device, kind, x.size, _ptr(x), _ptr(yy) if yy is not None else None, _ptr(out))
    if st != _capi.OK:
        raise SmcmcError(st, lib.smcmc_status_string(st).decode())
    return out


def selftest_mfma(a, b, device=0):
    lib = _capi.load()
    a, b = _f64(a), _f64(b)
    K = a.shape[1]
    c = np.zeros((16, 16))
    st = lib.smcmc_selftest_mfma(device, K, _ptr(a), _ptr(b), _ptr(c))
    if st != _capi.OK:
        raise SmcmcError(st, lib.smcmc_status_string(st).decode())
    return c


def selftest_mfma_strip(a, b, device=0):
    lib = _capi.load()
    a, b = _f64(a), _f64(b)
    c = np.zeros((4, 16))
    st = lib.smcmc_selftest_mfma_strip(device, a.shape[1], _ptr(a), _ptr(b), _ptr(c))
    if st != _capi.OK:
        raise SmcmcError(st, lib.smcmc_status_string(st).decode())
    return c


class HmcEngine:
    """N sMCMC::TSimpleHMC chains (reference TSimpleHMC.H:119-973) with the analytic gradient of a device
    likelihood.  SetMeanEpsilon(negative) + SetLeapFrog(n) fix the step: independent chains, many steps per launch.
    Otherwise every chain retunes its own step length and leapfrog count (:302-345) and the covariance-driven
    retuning (:665-858) is pooled over the ensemble every SetSyncInterval steps."""

    def __init__(self, dim, nchains=1, likelihood=LIKE_ISO_GAUSS, likelihood_params=None, seed=20240607,
                 chain_offset=0, device=0, stream=None, exact=True, library=None):
        # library: path of a build that carries a user likelihood (LIKE_USER as an HMC target through gradient type 2 / 3 / 5)
        self._lib = _capi.load(library)
        self.dim, self.nchains = int(dim), int(nchains)
        h = C.c_void_p()
        st = self._lib.smcmc_hmc_create(self.dim, self.nchains, likelihood, seed, chain_offset, device, C.byref(h))
        self._h = h
        if st != _capi.OK:
            msg = self._lib.smcmc_hmc_last_error(h).decode() if h else self._lib.smcmc_status_string(st).decode()
            if h:
                self._lib.smcmc_hmc_destroy(h)
            self._h = None
            raise SmcmcError(st, msg)
        if likelihood_params is not None:
            prm = _f64(likelihood_params).ravel()
            self._check(self._lib.smcmc_hmc_set_likelihood_params(self._h, _ptr(prm), prm.size))
        if stream is not None:
            self._check(self._lib.smcmc_hmc_set_stream(self._h, C.c_void_p(int(stream))))
        if not exact:
            self._check(self._lib.smcmc_hmc_set_exact_arithmetic(self._h, 0))

    def _check(self, st):
        if st != _capi.OK:
            raise SmcmcError(st, self._lib.smcmc_hmc_last_error(self._h).decode()
                             or self._lib.smcmc_status_string(st).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.smcmc_hmc_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def SetAlpha(self, a): self._check(self._lib.smcmc_hmc_set_alpha(self._h, float(a)))
    def SetMeanEpsilon(self, e): self._check(self._lib.smcmc_hmc_set_mean_epsilon(self._h, float(e)))
    def SetLeapFrog(self, n): self._check(self._lib.smcmc_hmc_set_leapfrog(self._h, int(n)))

    def GetMeanEpsilon(self):
        out = C.c_double(0)
        self._check(self._lib.smcmc_hmc_get_mean_epsilon(self._h, C.byref(out)))
        return out.value

    def GetLeapFrog(self):
        """fLeapFrogSteps of chain 0, signed as the reference keeps it (negative = fixed by SetLeapFrog)."""
        out = C.c_int(0)
        self._check(self._lib.smcmc_hmc_get_leapfrog(self._h, C.byref(out)))
        return out.value

    def SetSyncInterval(self, steps): self._check(self._lib.smcmc_hmc_set_sync_interval(self._h, int(steps)))
    def TrackCovariance(self, on=True): self._check(self._lib.smcmc_hmc_set_track_covariance(self._h, int(on)))

    def SetGradientType(self, gradient_type):
        """Step(save, gradientType) of TSimpleHMC.H:279: 0 / 1 / 4 the likelihood's gradient, 2 covariant, 3 finite
        differences, 5 zero (PotentialGradient, :467-532)."""
        self._check(self._lib.smcmc_hmc_set_gradient_type(self._h, int(gradient_type)))

    def GetGradientType(self): return int(self._lib.smcmc_hmc_get_gradient_type(self._h))
    def sync(self): self._check(self._lib.smcmc_hmc_sync(self._h))

    # the pooled update in pieces (a sharded ensemble adds the ranks' moment vectors between export and import)
    @property
    def moments_size(self): return self._lib.smcmc_hmc_moments_size(self._h)
    def reduce_moments(self): self._check(self._lib.smcmc_hmc_reduce_moments(self._h))
    def export_moments(self, dst_device_ptr): self._check(self._lib.smcmc_hmc_export_moments(self._h, C.c_void_p(int(dst_device_ptr))))
    def import_moments(self, src_device_ptr): self._check(self._lib.smcmc_hmc_import_moments(self._h, C.c_void_p(int(src_device_ptr))))
    def apply_moments(self): self._check(self._lib.smcmc_hmc_apply_moments(self._h))

    @property
    def moment_group(self):
        return self._lib.smcmc_hmc_moment_group(self._h)

    @property
    def tuning(self):
        out = np.zeros(len(_capi.HMC_TUNING))
        self._check(self._lib.smcmc_hmc_get_tuning(self._h, _ptr(out)))
        return dict(zip(_capi.HMC_TUNING, out))

    @property
    def average(self):
        out = np.zeros(self.dim)
        self._check(self._lib.smcmc_hmc_get_average_point(self._h, _ptr(out)))
        return out

    @property
    def covariance(self):
        out = np.zeros((self.dim, self.dim))
        self._check(self._lib.smcmc_hmc_get_covariance(self._h, _ptr(out)))
        return out

    def Start(self, start):
        start = _f64(start)
        broadcast = int(start.ndim == 1)
        if not broadcast and start.shape != (self.dim, self.nchains):
            raise ValueError("start must be [dim] or [dim][nchains]")
        self._check(self._lib.smcmc_hmc_start(self._h, _ptr(start), broadcast))

    def Step(self, nsteps=1, gradient_type=None):
        if gradient_type is not None:
            self.SetGradientType(gradient_type)
        self._check(self._lib.smcmc_hmc_step(self._h, int(nsteps)))

    def state(self):
        q = np.zeros((self.dim, self.nchains))
        m = np.zeros((self.dim, self.nchains))
        logl = np.zeros(self.nchains)
        self._check(self._lib.smcmc_hmc_read_state(self._h, _ptr(q), _ptr(m), _ptr(logl)))
        return q, m, logl

    @property
    def nchains_padded(self): return self._lib.smcmc_hmc_nchains_padded(self._h)

    def copy_positions(self, dst_device_ptr):
        """fAccepted of every chain into a device buffer [dim][nchains_padded] (one slot of a trace), on the engine's stream."""
        self._check(self._lib.smcmc_hmc_copy_positions(self._h, C.c_void_p(int(dst_device_ptr))))

    def AutocorrelationSums(self, trace_ptr, nslots, centre=None, stream=0):
        """As Engine.AutocorrelationSums, over a trace of copy_positions slots ([slot][dim][nchains_padded])."""
        c = _f64(np.zeros(self.dim) if centre is None else centre)
        s = np.zeros(self.dim)
        lagged = np.zeros((_capi.AUTOCORR_LAGS, self.dim))
        self._check(self._lib.smcmc_autocorrelation_sums(C.c_void_p(int(trace_ptr)), int(nslots), self.dim, self.dim,
                                                         self.nchains, self.nchains_padded, _ptr(c), _ptr(s), _ptr(lagged),
                                                         C.c_void_p(int(stream))))
        return Autocorrelation(s, lagged, int(nslots), self.nchains)

    def lane(self, name):
        if name in _capi.HMC_LANE_F64:
            out = np.zeros(self.nchains)
            self._check(self._lib.smcmc_hmc_read_lane_f64(self._h, _capi.HMC_LANE_F64[name], _ptr(out)))
            return out
        out = np.zeros(self.nchains, np.int32)
        self._check(self._lib.smcmc_hmc_read_lane_i32(self._h, _capi.HMC_LANE_I32[name],
                                                      out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out


class VaatEngine:
    """N independent sMCMC::TSimpleMCMC<L, sMCMC::TProposeVAATStep> chains (reference TProposeVAATStep.H:22-307 driven
    by TSimpleMCMC::Step): one coordinate per step from a shuffled queue of the dimensions, a proposal width per
    dimension adapted to a 44 % acceptance.  Chain c is the reference chain on the random stream (seed, chain_offset + c)."""

    def __init__(self, dim, nchains=1, likelihood=LIKE_ISO_GAUSS, likelihood_params=None, seed=20240607,
                 chain_offset=0, device=0, stream=None, exact=True, library=None):
        # library: path of a build that carries a user likelihood (LIKE_USER), see build.py --user-likelihood
        self._lib = _capi.load(library)
        self.dim, self.nchains = int(dim), int(nchains)
        h = C.c_void_p()
        st = self._lib.smcmc_vaat_create(self.dim, self.nchains, likelihood, seed, chain_offset, device, C.byref(h))
        self._h = h
        if st != _capi.OK:
            msg = self._lib.smcmc_vaat_last_error(h).decode() if h else self._lib.smcmc_status_string(st).decode()
            if h:
                self._lib.smcmc_vaat_destroy(h)
            self._h = None
            raise SmcmcError(st, msg)
        if likelihood_params is not None:
            prm = _f64(likelihood_params).ravel()
            self._check(self._lib.smcmc_vaat_set_likelihood_params(self._h, _ptr(prm), prm.size))
        if stream is not None:
            self._check(self._lib.smcmc_vaat_set_stream(self._h, C.c_void_p(int(stream))))
        if not exact:
            self._check(self._lib.smcmc_vaat_set_exact_arithmetic(self._h, 0))

    def _check(self, st):
        if st != _capi.OK:
            raise SmcmcError(st, self._lib.smcmc_vaat_last_error(self._h).decode()
                             or self._lib.smcmc_status_string(st).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.smcmc_vaat_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- TProposeVAATStep's interface (GetProposeStep() of the reference) ----
    def SetUniform(self, dim, minimum, maximum):
        self._check(self._lib.smcmc_vaat_set_uniform(self._h, int(dim), float(minimum), float(maximum)))

    def SetGaussian(self, dim, sigma): self._check(self._lib.smcmc_vaat_set_gaussian(self._h, int(dim), float(sigma)))
    def SetAcceptanceWindow(self, a): self._check(self._lib.smcmc_vaat_set_acceptance_window(self._h, float(a)))
    def SetAcceptanceRigidity(self, r): self._check(self._lib.smcmc_vaat_set_acceptance_rigidity(self._h, float(r)))
    def SetStepRMSWindow(self, n): self._check(self._lib.smcmc_vaat_set_step_rms_window(self._h, int(n)))
    def UpdateProposal(self): self._check(self._lib.smcmc_vaat_update_proposal(self._h))

    def GetAcceptanceWindow(self):
        out = C.c_double(0)
        self._check(self._lib.smcmc_vaat_get_acceptance_window(self._h, C.byref(out)))
        return out.value

    def GetAcceptanceRigidity(self):
        out = C.c_double(0)
        self._check(self._lib.smcmc_vaat_get_acceptance_rigidity(self._h, C.byref(out)))
        return out.value

    def GetSuccesses(self, chain=0): return int(self.lane("successes")[chain])      # :151
    def GetTrials(self, chain=0): return int(self.lane("trials")[chain])            # :154
    def GetAcceptance(self, chain=0):                                               # :157-165: the mean over the dimensions
        a = self.per_dim("acceptance")[:, chain]
        return float(np.add.reduce(a) / a.size) if a.size else 0.0

    def GetSigma(self, chain=0):                                                    # :168-175
        s = self.per_dim("sigma")[:, chain]
        return float(np.add.reduce(s) / s.size) if s.size else 0.0

    # ---- TSimpleMCMC's interface ----
    def Start(self, start):
        start = _f64(start)
        broadcast = int(start.ndim == 1)
        if not broadcast and start.shape != (self.dim, self.nchains):
            raise ValueError("start must be [dim] or [dim][nchains]")
        st = self._lib.smcmc_vaat_start(self._h, _ptr(start), broadcast)
        if st == _capi.ERR_BAD_START:
            return False
        self._check(st)
        return True

    def Step(self, nsteps=1): self._check(self._lib.smcmc_vaat_step(self._h, int(nsteps)))

    def step_save(self, nsteps, stride, save_x_ptr, save_logl_ptr=None):
        self._check(self._lib.smcmc_vaat_step_save(self._h, int(nsteps), int(stride), C.c_void_p(int(save_x_ptr)),
                                                   C.c_void_p(int(save_logl_ptr)) if save_logl_ptr else None))

    @property
    def total_steps(self): return self._lib.smcmc_vaat_total_steps(self._h)

    @property
    def queue_length(self): return self._lib.smcmc_vaat_queue_length(self._h)

    @property
    def nchains_padded(self): return self._lib.smcmc_vaat_nchains_padded(self._h)

    def GetAccepted(self):
        x = np.zeros((self.dim, self.nchains))
        self._check(self._lib.smcmc_vaat_read_state(self._h, _ptr(x), None))
        return x

    def lane(self, name):
        if name in _capi.VAAT_LANE_F64:
            out = np.zeros(self.nchains)
            self._check(self._lib.smcmc_vaat_read_lane_f64(self._h, _capi.VAAT_LANE_F64[name], _ptr(out)))
            return out
        out = np.zeros(self.nchains, np.int32)
        self._check(self._lib.smcmc_vaat_read_lane_i32(self._h, _capi.VAAT_LANE_I32[name],
                                                       out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def per_dim(self, name):
        """fSigma / fAcceptance / fAcceptanceTrials / fNextIndex as [dim][chain]."""
        if name in _capi.VAAT_DIM_F64:
            out = np.zeros((self.dim, self.nchains))
            self._check(self._lib.smcmc_vaat_read_dim_f64(self._h, _capi.VAAT_DIM_F64[name], _ptr(out)))
            return out
        out = np.zeros((self.dim, self.nchains), np.int32)
        self._check(self._lib.smcmc_vaat_read_dim_i32(self._h, _capi.VAAT_DIM_I32[name],
                                                      out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def state_device_ptr(self):
        x, logl = C.c_void_p(), C.c_void_p()
        self._check(self._lib.smcmc_vaat_state_device_ptr(self._h, C.byref(x), C.byref(logl)))
        return x.value, logl.value
