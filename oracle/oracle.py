"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package must never do so.  See oracle/oracle_core.h for
what the oracle restates (reference file:line per function) and its parity
status (PARITY UNPINNED: the reference ships no golden vectors).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libsmcmc_oracle.so")

LIKE_ISO, LIKE_QUADFORM, LIKE_ROSENBROCK, LIKE_ASYM, LIKE_HORRIFIC, LIKE_CONSTRAINED = 0, 1, 2, 4, 5, 6
MODE_FROZEN, MODE_POOLED = 0, 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_bp = C.POINTER(C.c_uint8)


def build(force=False):
    """Compile the oracle with the committed Makefile (gcc, a second or two)."""
    srcs = [os.path.join(_HERE, f) for f in
            ("smcmc_oracle.c", "ensemble_oracle.c", "hmc_oracle.c", "vaat_oracle.c", "oracle_core.h", "oracle_linalg.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "smcmc_detmath.h"))
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


_FROZEN_LIB_PATH = os.path.join(_HERE, "_build", "libsmcmc_oracle_frozen_definition.so")


def build_frozen_definition(force=False):
    """The oracle compiled with -DSMCMC_PHILOX_ROUNDS=10 -DSMCMC_NORMAL_TEXTBOOK (Makefile target `frozen`)."""
    srcs = [os.path.join(_HERE, f) for f in
            ("smcmc_oracle.c", "ensemble_oracle.c", "hmc_oracle.c", "vaat_oracle.c", "oracle_core.h", "oracle_linalg.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "smcmc_detmath.h"))
    stale = force or not os.path.exists(_FROZEN_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_FROZEN_LIB_PATH) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "frozen"] + (["-B"] if force else []))
    return _FROZEN_LIB_PATH


def frozen_definition():
    """This module once more, bound to the frozen-definition build of the oracle: the same classes (Chain, Ensemble, ...)
    drawing with ten Philox rounds and the textbook normal pair."""
    import importlib.util
    import sys
    name = __name__ + "_frozen_definition"
    if name in sys.modules:
        return sys.modules[name]
    build_frozen_definition()
    spec = importlib.util.spec_from_file_location(name, os.path.abspath(__file__))
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    m._LIB_PATH = _FROZEN_LIB_PATH
    m.build = build_frozen_definition
    return m


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    L.oracle_chain_create.restype = C.c_void_p
    L.oracle_chain_create.argtypes = [C.c_int, C.c_int, _dp, C.c_int, C.c_uint64, C.c_uint32]
    L.oracle_chain_destroy.argtypes = [C.c_void_p]
    for name, args in [
        ("set_gaussian", [C.c_int, C.c_double]), ("set_uniform", [C.c_int, C.c_double, C.c_double]),
        ("set_correlation", [C.c_int, C.c_int, C.c_double]), ("set_covariance_frozen", [C.c_int]),
        ("set_covariance_window", [C.c_int]), ("set_covariance_deweight", [C.c_double]),
        ("set_acceptance_window", [C.c_double]), ("set_acceptance_deweight", [C.c_double]),
        ("set_acceptance_rigidity", [C.c_double]), ("set_target_acceptance", [C.c_double]),
        ("set_next_update", [C.c_double]), ("set_sigma", [C.c_double]),
        ("set_step_rms_window", [C.c_int]), ("set_scan_dimension", [C.c_int]),
        ("set_covariance_trials", [C.c_double]), ("set_center_trials", [C.c_double]), ("set_covariance", [_dp]),
        ("force_step", [_dp]), ("update_proposal", []), ("reset_proposal", []),
    ]:
        f = getattr(L, "oracle_chain_" + name)
        f.restype = None
        f.argtypes = [C.c_void_p] + args
    L.oracle_chain_start_api.restype = C.c_int
    L.oracle_chain_start_api.argtypes = [C.c_void_p, _dp]
    L.oracle_chain_restore_api.restype = None
    L.oracle_chain_restore_api.argtypes = [C.c_void_p, _dp, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int,
                                           C.c_double, C.c_double, C.c_double, _dp, C.c_double, _dp, C.c_double]
    L.oracle_chain_step_api.restype = C.c_int
    L.oracle_chain_step_api.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.oracle_chain_run.restype = None
    L.oracle_chain_run.argtypes = [C.c_void_p, C.c_int, C.c_int, _bp, _dp, _dp, _dp, _dp, _ip, _ip, _dp]
    L.oracle_chain_run_moments.restype = None
    L.oracle_chain_run_moments.argtypes = [C.c_void_p, C.c_int, _dp, _dp, C.POINTER(C.c_int)]
    for name in ("accepted", "proposed", "center", "covariance", "decomposition", "scalars"):
        f = getattr(L, "oracle_chain_get_" + name)
        f.restype = None
        f.argtypes = [C.c_void_p, _dp]

    L.oracle_log_v.argtypes = [C.c_int, _dp, _dp]
    L.oracle_exp_v.argtypes = [C.c_int, _dp, _dp]
    L.oracle_pow_small_v.argtypes = [C.c_int, _dp, _dp, _dp]
    L.oracle_sincos2pi_v.argtypes = [C.c_int, _dp, _dp, _dp]
    L.oracle_sincos2pi_u32_v.argtypes = [C.c_int, _dp, _dp, _dp]
    L.oracle_u01_v.argtypes = [C.c_int, _dp, _dp]
    L.oracle_normal_pair_v.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
    L.oracle_normal_pair_halfcircle_v.argtypes = [C.c_int, _dp, _dp, _dp, _dp]
    L.oracle_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
    L.oracle_philox_rounds.argtypes = [C.c_uint32] * 6 + [C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    L.oracle_philox_draw_rounds.restype = C.c_int
    L.oracle_draw_block.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32)]
    L.oracle_step_draws.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_int, _dp, _dp]
    L.oracle_cholesky.restype = C.c_int
    L.oracle_cholesky.argtypes = [C.c_int, _dp, _dp]
    L.oracle_eigen.argtypes = [C.c_int, _dp, _dp, _dp]
    L.oracle_loglike.restype = C.c_double
    L.oracle_loglike.argtypes = [C.c_int, C.c_int, _dp, _dp]
    L.oracle_dummy_error_matrix.restype = C.c_int
    L.oracle_dummy_error_matrix.argtypes = [C.c_int, _dp, _dp]

    L.oracle_ensemble_create.restype = C.c_void_p
    L.oracle_ensemble_create.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_uint64, C.c_uint32,
                                         C.c_int, C.c_int]
    L.oracle_ensemble_destroy.argtypes = [C.c_void_p]
    for name, args in [
        ("set_gaussian", [C.c_int, C.c_double]), ("set_uniform", [C.c_int, C.c_double, C.c_double]),
        ("set_correlation", [C.c_int, C.c_int, C.c_double]), ("set_covariance_window", [C.c_double]),
        ("set_covariance_deweight", [C.c_double]), ("set_acceptance_window", [C.c_double]),
        ("set_acceptance_deweight", [C.c_double]), ("set_acceptance_rigidity", [C.c_double]),
        ("set_target_acceptance", [C.c_double]), ("set_step_rms_window", [C.c_int]),
        ("set_sigma", [C.c_double]), ("set_moment_grouping", [C.c_int, C.c_int]), ("set_quadform_rowwise", [C.c_int]),
        ("step", [C.c_int, C.c_int]), ("reduce_moments", [_dp]), ("set_covariance", [_dp]),
        ("update_proposal", []), ("reset_proposal", []),
        ("apply_moments", [_dp]), ("sync", []), ("get_x", [_dp]), ("get_lane_f64", [C.c_int, _dp]),
        ("get_lane_i32", [C.c_int, _ip]), ("get_last_accept", [_bp]), ("get_center", [_dp]),
        ("get_covariance", [_dp]), ("get_decomposition", [_dp]), ("get_shared", [_dp]),
    ]:
        f = getattr(L, "oracle_ensemble_" + name)
        f.restype = None
        f.argtypes = [C.c_void_p] + args
    L.oracle_ensemble_start.restype = C.c_int
    L.oracle_ensemble_start.argtypes = [C.c_void_p, _dp, C.c_int]

    L.oracle_vaat_create.restype = C.c_void_p
    L.oracle_vaat_create.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_uint64, C.c_uint32, C.c_int]
    L.oracle_vaat_destroy.argtypes = [C.c_void_p]
    for name, args in [
        ("set_gaussian", [C.c_int, C.c_double]), ("set_uniform", [C.c_int, C.c_double, C.c_double]),
        ("set_acceptance_window", [C.c_double]), ("set_acceptance_rigidity", [C.c_double]),
        ("set_step_rms_window", [C.c_int]), ("update_proposal", []), ("step", [C.c_int]), ("get_x", [_dp]),
        ("get_lane_f64", [C.c_int, _dp]), ("get_lane_i32", [C.c_int, _ip]), ("get_dim_f64", [C.c_int, _dp]),
        ("get_dim_i32", [C.c_int, _ip]),
    ]:
        f = getattr(L, "oracle_vaat_" + name)
        f.restype = None
        f.argtypes = [C.c_void_p] + args
    L.oracle_vaat_start.restype = C.c_int
    L.oracle_vaat_start.argtypes = [C.c_void_p, _dp, C.c_int]
    L.oracle_vaat_get_acceptance_window.restype = C.c_int
    L.oracle_vaat_get_acceptance_window.argtypes = [C.c_void_p]

    L.oracle_hmc_create.restype = C.c_void_p
    L.oracle_hmc_create.argtypes = [C.c_int, C.c_int, _dp, C.c_int, C.c_uint64, C.c_uint32]
    L.oracle_hmc_destroy.argtypes = [C.c_void_p]
    L.oracle_hmc_set_alpha.argtypes = [C.c_void_p, C.c_double]
    L.oracle_hmc_set_mean_epsilon.argtypes = [C.c_void_p, C.c_double]
    L.oracle_hmc_set_leapfrog.argtypes = [C.c_void_p, C.c_int]
    L.oracle_hmc_set_potential_from_gradient.argtypes = [C.c_void_p, C.c_int]
    L.oracle_hmc_set_fused_gradient.argtypes = [C.c_void_p, C.c_int]
    L.oracle_hmc_start.argtypes = [C.c_void_p, _dp]
    L.oracle_hmc_step.restype = C.c_int
    L.oracle_hmc_step.argtypes = [C.c_void_p]
    L.oracle_hmc_run.argtypes = [C.c_void_p, C.c_int]
    for name in ("accepted", "momentum", "central", "scalars"):
        f = getattr(L, "oracle_hmc_get_" + name)
        f.restype = None
        f.argtypes = [C.c_void_p, _dp]
    L.oracle_hmc_gradient.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp]
    L.oracle_hmc_set_gradient_type.argtypes = [C.c_void_p, C.c_int]
    for name in ("average", "covariance"):
        f = getattr(L, "oracle_hmc_get_" + name)
        f.restype = None
        f.argtypes = [C.c_void_p, _dp]
    L.oracle_hmc_ensemble_create.restype = C.c_void_p
    L.oracle_hmc_ensemble_create.argtypes = [C.c_int, C.c_int, C.c_int, _dp, C.c_int, C.c_uint64, C.c_uint32, C.c_int,
                                             C.c_int]
    L.oracle_hmc_ensemble_destroy.argtypes = [C.c_void_p]
    L.oracle_hmc_ensemble_chain.restype = C.c_void_p
    L.oracle_hmc_ensemble_chain.argtypes = [C.c_void_p, C.c_int]
    L.oracle_hmc_ensemble_configure.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int]
    L.oracle_hmc_ensemble_set_mean_epsilon.argtypes = [C.c_void_p, C.c_double]
    L.oracle_hmc_ensemble_set_leapfrog.argtypes = [C.c_void_p, C.c_int]
    L.oracle_hmc_ensemble_start.argtypes = [C.c_void_p, _dp, C.c_int]
    L.oracle_hmc_ensemble_sync.argtypes = [C.c_void_p]
    L.oracle_hmc_ensemble_step.argtypes = [C.c_void_p, C.c_int]
    L.oracle_hmc_ensemble_get_state.argtypes = [C.c_void_p, _dp, _dp]
    L.oracle_hmc_ensemble_get_lane.argtypes = [C.c_void_p, C.c_int, _dp]
    for name in ("average", "covariance", "shared"):
        f = getattr(L, "oracle_hmc_ensemble_get_" + name)
        f.restype = None
        f.argtypes = [C.c_void_p, _dp]


def _p(a):
    return a.ctypes.data_as(_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dummy_error_matrix(dim):
    """TDummyLogLikelihood::Init() for a general D: (Covariance, Error)."""
    cov = np.zeros((dim, dim))
    err = np.zeros((dim, dim))
    ok = lib().oracle_dummy_error_matrix(dim, _p(cov), _p(err))
    assert ok
    return cov, err


def like_params(kind, dim, params=None):
    if params is not None:
        return _f64(params).ravel()
    if kind == LIKE_QUADFORM:
        return dummy_error_matrix(dim)[1].ravel()
    if kind == LIKE_ROSENBROCK:
        return np.array([100.0])
    if kind == LIKE_ASYM:
        return np.array([-1.0, 100.0])           # TAsymLogLikelihood.H:17-18
    if kind == LIKE_CONSTRAINED:
        return constrained_params(dim)
    return np.zeros(0)


def constrained_params(dim=25):
    """example4/TConstrainedLikelihood.H:55-110 (Init): the sum 1902 +- 16, 24 values 76 +- 8 % and one 80 +- 2; for another
    dimension the same pattern (the last value is the tight one).  Layout {SummedValues, SummedConstraint, Expected[D],
    Prior[D]}."""
    expected = np.full(dim, 76.0)
    prior = np.full(dim, 76.0 * 0.08)
    expected[dim - 1], prior[dim - 1] = 80.0, 2.0
    total = 1902.0 if dim == 25 else float(expected.sum()) - 2.0
    return np.concatenate([[total, 16.0], expected, prior])


SCALAR_NAMES = ["accepted_logl", "proposed_logl", "sigma", "acceptance", "acceptance_trials",
                "acceptance_window", "rigidity", "target", "sigma_trace", "cov_trials",
                "central_trials", "cov_window", "trials", "successes", "next_update", "total_steps",
                "like_count", "step_rms", "update_count", "last_update_path", "failed",
                "step_rms_trials"]


class Chain:
    """One reference chain: sMCMC::TSimpleMCMC<L, TProposeAdaptiveStep>."""

    def __init__(self, dim, kind=LIKE_ISO, params=None, seed=20240607, chain_id=0):
        self.dim = dim
        prm = like_params(kind, dim, params)
        self._h = lib().oracle_chain_create(dim, kind, _p(prm) if prm.size else None, prm.size, seed, chain_id)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_chain_destroy(self._h)
            self._h = None

    def __getattr__(self, name):
        if name.startswith("set_") or name in ("update_proposal", "reset_proposal"):
            f = getattr(lib(), "oracle_chain_" + name)
            return lambda *a: f(self._h, *a)
        raise AttributeError(name)

    def force_step(self, p):
        lib().oracle_chain_force_step(self._h, _p(_f64(p)))

    def set_covariance(self, cov):
        cov = _f64(cov)
        assert cov.shape == (self.dim, self.dim)
        lib().oracle_chain_set_covariance(self._h, _p(cov))

    def start(self, x0):
        return bool(lib().oracle_chain_start_api(self._h, _p(_f64(x0))))

    def step(self, save=False, metropolis=0):
        return bool(lib().oracle_chain_step_api(self._h, int(save), metropolis))

    def saved_state(self):
        """What SaveStep(true) writes for this chain (TSimpleMCMC.H:208-215, 1616-1673)."""
        sc = self.scalars
        cov = self.covariance
        return dict(accepted=self.accepted, log_likelihood=sc["accepted_logl"], total_steps=int(sc["total_steps"]),
                    step_rms=sc["step_rms"], trials=int(sc["trials"]), successes=int(sc["successes"]),
                    next_update=int(sc["next_update"]), acceptance=sc["acceptance"],
                    acceptance_trials=sc["acceptance_trials"], sigma=sc["sigma"], central_point=self.center,
                    central_point_trials=sc["central_trials"],
                    covariance=np.array([cov[i, j] for i in range(self.dim) for j in range(i + 1)]),
                    covariance_trials=sc["cov_trials"])

    def restore(self, st):
        """Restore(tree) + RestoreState (TSimpleMCMC.H:282-352, 1501-1612) from a saved_state() dict."""
        lib().oracle_chain_restore_api(self._h, _p(_f64(st["accepted"])), st["log_likelihood"], st["total_steps"],
                                       st["step_rms"], st["trials"], st["successes"], st["next_update"],
                                       st["acceptance"], st["acceptance_trials"], st["sigma"],
                                       _p(_f64(st["central_point"])), st["central_point_trials"],
                                       _p(_f64(st["covariance"])), st["covariance_trials"])

    def run(self, nsteps, metropolis=0):
        out = dict(accepted=np.zeros(nsteps, np.uint8), logl_proposed=np.zeros(nsteps),
                   logl_accepted=np.zeros(nsteps), sigma=np.zeros(nsteps), acceptance=np.zeros(nsteps),
                   trials=np.zeros(nsteps, np.int32), successes=np.zeros(nsteps, np.int32),
                   step_rms=np.zeros(nsteps))
        lib().oracle_chain_run(self._h, nsteps, metropolis, out["accepted"].ctypes.data_as(_bp),
                               _p(out["logl_proposed"]), _p(out["logl_accepted"]), _p(out["sigma"]),
                               _p(out["acceptance"]), out["trials"].ctypes.data_as(_ip),
                               out["successes"].ctypes.data_as(_ip), _p(out["step_rms"]))
        return out

    def run_quiet(self, nsteps):
        lib().oracle_chain_run(self._h, nsteps, 0, None, None, None, None, None, None, None, None)

    def run_moments(self, nsteps):
        s = np.zeros(self.dim)
        ss = np.zeros((self.dim, self.dim))
        na = C.c_int(0)
        lib().oracle_chain_run_moments(self._h, nsteps, _p(s), _p(ss), C.byref(na))
        return s, ss, na.value

    def _vec(self, name, n):
        out = np.zeros(n)
        getattr(lib(), "oracle_chain_get_" + name)(self._h, _p(out))
        return out

    accepted = property(lambda self: self._vec("accepted", self.dim))
    proposed = property(lambda self: self._vec("proposed", self.dim))
    center = property(lambda self: self._vec("center", self.dim))
    covariance = property(lambda self: self._vec("covariance", self.dim * self.dim).reshape(self.dim, self.dim))
    decomposition = property(lambda self: self._vec("decomposition", self.dim * self.dim).reshape(self.dim, self.dim))

    @property
    def scalars(self):
        return dict(zip(SCALAR_NAMES, self._vec("scalars", len(SCALAR_NAMES))))


LANE_F64 = {"logl": 0, "sigma": 1, "acceptance": 2, "acceptance_trials": 3, "rigidity": 4,
            "step_rms": 5, "logl_proposed": 6}
LANE_I32 = {"trials": 0, "successes": 1, "next_update": 2, "naccept": 3, "step_rms_trials": 4}
SHARED_NAMES = ["sigma_trace", "cov_trials", "central_trials", "cov_window", "acceptance_window",
                "target", "total_steps", "update_count", "last_update_path", "failed",
                "last_sigma_scale", "decomp_full"]


class Ensemble:
    """The many-chain engine semantics, restated on the CPU."""

    def __init__(self, nchains, dim, kind=LIKE_ISO, params=None, seed=20240607, chain_offset=0,
                 mode=MODE_POOLED, exact=True):
        self.nchains, self.dim = nchains, dim
        prm = like_params(kind, dim, params)
        self._h = lib().oracle_ensemble_create(nchains, dim, kind, _p(prm) if prm.size else None, prm.size,
                                               seed, chain_offset, mode, int(exact))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_ensemble_destroy(self._h)
            self._h = None

    def set_covariance(self, cov):
        cov = _f64(cov)
        assert cov.shape == (self.dim, self.dim)
        lib().oracle_ensemble_set_covariance(self._h, _p(cov))

    def __getattr__(self, name):
        if name.startswith("set_") or name in ("sync", "update_proposal", "reset_proposal"):
            f = getattr(lib(), "oracle_ensemble_" + name)
            return lambda *a: f(self._h, *a)
        raise AttributeError(name)

    def start(self, x0):
        x0 = _f64(x0)
        broadcast = int(x0.ndim == 1)
        if not broadcast:
            assert x0.shape == (self.dim, self.nchains)
        return bool(lib().oracle_ensemble_start(self._h, _p(x0), broadcast))

    def step(self, nsteps=1, metropolis=0):
        lib().oracle_ensemble_step(self._h, nsteps, metropolis)

    @property
    def npacked(self):
        return (self.dim + 1) * (self.dim + 2) // 2

    def reduce_moments(self):
        m = np.zeros(self.npacked)
        lib().oracle_ensemble_reduce_moments(self._h, _p(m))
        return m

    def apply_moments(self, m):
        lib().oracle_ensemble_apply_moments(self._h, _p(_f64(m)))

    @property
    def x(self):
        out = np.zeros((self.dim, self.nchains))
        lib().oracle_ensemble_get_x(self._h, _p(out))
        return out

    def lane(self, name):
        if name in LANE_F64:
            out = np.zeros(self.nchains)
            lib().oracle_ensemble_get_lane_f64(self._h, LANE_F64[name], _p(out))
            return out
        if name == "last_accept":
            out = np.zeros(self.nchains, np.uint8)
            lib().oracle_ensemble_get_last_accept(self._h, out.ctypes.data_as(_bp))
            return out
        out = np.zeros(self.nchains, np.int32)
        lib().oracle_ensemble_get_lane_i32(self._h, LANE_I32[name], out.ctypes.data_as(_ip))
        return out

    def _vec(self, name, n):
        out = np.zeros(n)
        getattr(lib(), "oracle_ensemble_get_" + name)(self._h, _p(out))
        return out

    center = property(lambda self: self._vec("center", self.dim))
    covariance = property(lambda self: self._vec("covariance", self.dim ** 2).reshape(self.dim, self.dim))
    decomposition = property(lambda self: self._vec("decomposition", self.dim ** 2).reshape(self.dim, self.dim))

    @property
    def shared(self):
        return dict(zip(SHARED_NAMES, self._vec("shared", len(SHARED_NAMES))))


# ---- vectorised detmath (tests/test_detmath.py) ----
def det_log(x):
    x = _f64(x); out = np.empty_like(x); lib().oracle_log_v(x.size, _p(x), _p(out)); return out


def det_exp(x):
    x = _f64(x); out = np.empty_like(x); lib().oracle_exp_v(x.size, _p(x), _p(out)); return out


def det_pow_small(x, y):
    x = _f64(x); y = _f64(y); out = np.empty_like(x)
    lib().oracle_pow_small_v(x.size, _p(x), _p(y), _p(out)); return out


def det_sincos2pi(u):
    u = _f64(u); s = np.empty_like(u); c = np.empty_like(u)
    lib().oracle_sincos2pi_v(u.size, _p(u), _p(s), _p(c)); return s, c


def det_sincos2pi_u32(w):
    w = _f64(w); s = np.empty_like(w); c = np.empty_like(w)
    lib().oracle_sincos2pi_u32_v(w.size, _p(w), _p(s), _p(c)); return s, c


def det_u01(w):
    w = _f64(w); out = np.empty_like(w); lib().oracle_u01_v(w.size, _p(w), _p(out)); return out


def det_normal_pair(w0, w1):
    w0 = _f64(w0); w1 = _f64(w1); a = np.empty_like(w0); b = np.empty_like(w0)
    lib().oracle_normal_pair_v(w0.size, _p(w0), _p(w1), _p(a), _p(b)); return a, b


def det_normal_pair_halfcircle(w0, w1):
    """SMCMC_NORMAL_PAIR_BODY_HALFCIRCLE (the form the step kernels run) with the table the kernels build."""
    w0 = _f64(w0); w1 = _f64(w1); a = np.empty_like(w0); b = np.empty_like(w0)
    lib().oracle_normal_pair_halfcircle_v(w0.size, _p(w0), _p(w1), _p(a), _p(b)); return a, b


def philox(ctr, key):
    out = (C.c_uint32 * 4)()
    lib().oracle_philox(*[int(v) for v in ctr], *[int(v) for v in key], out)
    return [int(v) for v in out]


def philox_rounds(ctr, key, first, rounds):
    out = (C.c_uint32 * 4)()
    lib().oracle_philox_rounds(*[int(v) for v in ctr], *[int(v) for v in key], int(first), int(rounds), out)
    return [int(v) for v in out]


def philox_draw_rounds():
    return lib().oracle_philox_draw_rounds()


def draw_block(seed, chain, step, block, stream=0):
    out = (C.c_uint32 * 4)()
    lib().oracle_draw_block(seed, chain, step, block, stream, out)
    return [int(v) for v in out]


def step_draws(seed, chain, step, dim):
    n = np.zeros(dim); u = C.c_double(0)
    lib().oracle_step_draws(seed, chain, step, dim, _p(n), C.byref(u))
    return n, u.value


def cholesky(A):
    A = _f64(A); U = np.zeros_like(A)
    ok = lib().oracle_cholesky(A.shape[0], _p(A), _p(U))
    return bool(ok), U


def eigen(A):
    A = _f64(A); n = A.shape[0]; vec = np.zeros_like(A); val = np.zeros(n)
    lib().oracle_eigen(n, _p(A), _p(vec), _p(val))
    return val, vec


def loglike(kind, p, params=None):
    p = _f64(p); prm = like_params(kind, p.size, params)
    return lib().oracle_loglike(kind, p.size, _p(p), _p(prm) if prm.size else None)


HMC_SCALARS = ["accepted_potential", "proposed_potential", "current_acceptance", "mean_epsilon",
               "leapfrog_steps", "step_count", "potential_count", "gradient_count", "last_accept",
               "central_potential", "reversal_len", "trace", "orbit", "updates", "cov_trials"]
HMC_SHARED = ["trace", "orbit", "updates", "cov_trials", "average_trials", "steps_remaining", "steps_since_update",
              "max_scale", "min_scale", "est_trace"]


class Hmc:
    """One reference HMC chain: sMCMC::TSimpleHMC<L, analytic gradient>."""

    def __init__(self, dim, kind=LIKE_ISO, params=None, seed=20240607, chain_id=0, potential_from_gradient=False,
                 fused_gradient=False):
        self.dim = dim
        prm = like_params(kind, dim, params)
        self._h = lib().oracle_hmc_create(dim, kind, _p(prm) if prm.size else None, prm.size, seed, chain_id)
        lib().oracle_hmc_set_potential_from_gradient(self._h, int(potential_from_gradient))
        lib().oracle_hmc_set_fused_gradient(self._h, int(fused_gradient))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_hmc_destroy(self._h)
            self._h = None

    def set_gradient_type(self, t): lib().oracle_hmc_set_gradient_type(self._h, int(t))
    def set_alpha(self, a): lib().oracle_hmc_set_alpha(self._h, a)
    def set_mean_epsilon(self, e): lib().oracle_hmc_set_mean_epsilon(self._h, e)
    def set_leapfrog(self, n): lib().oracle_hmc_set_leapfrog(self._h, n)
    def start(self, x0): lib().oracle_hmc_start(self._h, _p(_f64(x0)))
    def step(self): return lib().oracle_hmc_step(self._h)
    def run(self, n): lib().oracle_hmc_run(self._h, n)

    def _vec(self, name, n):
        out = np.zeros(n)
        getattr(lib(), "oracle_hmc_get_" + name)(self._h, _p(out))
        return out

    accepted = property(lambda self: self._vec("accepted", self.dim))
    momentum = property(lambda self: self._vec("momentum", self.dim))
    central = property(lambda self: self._vec("central", self.dim))
    average = property(lambda self: self._vec("average", self.dim))
    covariance = property(lambda self: self._vec("covariance", self.dim ** 2).reshape(self.dim, self.dim))

    @property
    def scalars(self):
        return dict(zip(HMC_SCALARS, self._vec("scalars", len(HMC_SCALARS))))


class HmcEnsemble:
    """The many-chain HMC engine's semantics on the CPU (oracle/hmc_oracle.c): per-chain reference chains whose
    covariance-derived tuning (UpdateCovariance / UpdateErrorMatrix) is pooled over moment groups of `group` chains
    every `sync_every` steps.  One chain, group 1, sync_every 1 is the reference chain."""

    def __init__(self, nchains, dim, kind=LIKE_ISO, params=None, seed=20240607, chain_offset=0, group=64, sync_every=1,
                 alpha=0.0, potential_from_gradient=False, fused_gradient=False, gradient_type=0):
        self.nchains, self.dim = nchains, dim
        prm = like_params(kind, dim, params)
        self._h = lib().oracle_hmc_ensemble_create(nchains, dim, kind, _p(prm) if prm.size else None, prm.size, seed,
                                                   chain_offset, group, sync_every)
        self._config = [alpha, int(potential_from_gradient), int(fused_gradient), int(gradient_type)]
        lib().oracle_hmc_ensemble_configure(self._h, *self._config)

    def set_alpha(self, a):
        self._config[0] = float(a)
        lib().oracle_hmc_ensemble_configure(self._h, *self._config)

    def set_gradient_type(self, t):
        """Step(save, gradientType) for the steps that follow (TSimpleHMC.H:279, 467-532)."""
        self._config[3] = int(t)
        lib().oracle_hmc_ensemble_configure(self._h, *self._config)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_hmc_ensemble_destroy(self._h)
            self._h = None

    def set_mean_epsilon(self, e): lib().oracle_hmc_ensemble_set_mean_epsilon(self._h, e)
    def set_leapfrog(self, n): lib().oracle_hmc_ensemble_set_leapfrog(self._h, n)

    def start(self, x0):
        x0 = _f64(x0)
        lib().oracle_hmc_ensemble_start(self._h, _p(x0), int(x0.ndim == 1))

    def step(self, n=1): lib().oracle_hmc_ensemble_step(self._h, n)
    def sync(self): lib().oracle_hmc_ensemble_sync(self._h)

    def state(self):
        q = np.zeros((self.dim, self.nchains)); m = np.zeros((self.dim, self.nchains))
        lib().oracle_hmc_ensemble_get_state(self._h, _p(q), _p(m))
        return q, m

    def lane(self, name):
        out = np.zeros(self.nchains)
        lib().oracle_hmc_ensemble_get_lane(self._h, HMC_SCALARS.index(name), _p(out))
        return out

    def _vec(self, name, n):
        out = np.zeros(n)
        getattr(lib(), "oracle_hmc_ensemble_get_" + name)(self._h, _p(out))
        return out

    average = property(lambda self: self._vec("average", self.dim))
    covariance = property(lambda self: self._vec("covariance", self.dim ** 2).reshape(self.dim, self.dim))

    @property
    def shared(self):
        return dict(zip(HMC_SHARED, self._vec("shared", len(HMC_SHARED))))


class Vaat:
    """N independent sMCMC::TSimpleMCMC<L, sMCMC::TProposeVAATStep> chains (oracle/vaat_oracle.c): chain c is the
    reference chain on the random stream (seed, chain_offset + c)."""

    LANE_F64 = {"logl": 0, "logl_proposed": 1, "step_rms": 2, "proposed_value": 3}
    LANE_I32 = {"trials": 0, "successes": 1, "last_index": 2, "queue_len": 3, "naccept": 4, "last_accept": 5,
                "step_rms_trials": 6}
    DIM_F64 = {"sigma": 0, "acceptance": 1}
    DIM_I32 = {"acceptance_trials": 0, "queue": 1}

    def __init__(self, nchains, dim, kind=LIKE_ISO, params=None, seed=20240607, chain_offset=0, exact=True):
        self.nchains, self.dim = nchains, dim
        prm = like_params(kind, dim, params)
        self._h = lib().oracle_vaat_create(nchains, dim, kind, _p(prm) if prm.size else None, prm.size, seed,
                                           chain_offset, int(exact))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_vaat_destroy(self._h)
            self._h = None

    def __getattr__(self, name):
        if name.startswith("set_") or name == "update_proposal":
            f = getattr(lib(), "oracle_vaat_" + name)
            return lambda *a: f(self._h, *a)
        raise AttributeError(name)

    def start(self, x0):
        x0 = _f64(x0)
        broadcast = int(x0.ndim == 1)
        if not broadcast:
            assert x0.shape == (self.dim, self.nchains)
        return bool(lib().oracle_vaat_start(self._h, _p(x0), broadcast))

    def step(self, nsteps=1): lib().oracle_vaat_step(self._h, nsteps)

    @property
    def x(self):
        out = np.zeros((self.dim, self.nchains))
        lib().oracle_vaat_get_x(self._h, _p(out))
        return out

    @property
    def acceptance_window(self): return lib().oracle_vaat_get_acceptance_window(self._h)

    def lane(self, name):
        if name in self.LANE_F64:
            out = np.zeros(self.nchains)
            lib().oracle_vaat_get_lane_f64(self._h, self.LANE_F64[name], _p(out))
            return out
        out = np.zeros(self.nchains, np.int32)
        lib().oracle_vaat_get_lane_i32(self._h, self.LANE_I32[name], out.ctypes.data_as(_ip))
        return out

    def per_dim(self, name):
        """fSigma / fAcceptance / fAcceptanceTrials / fNextIndex as [dim][chain]."""
        if name in self.DIM_F64:
            out = np.zeros((self.dim, self.nchains))
            lib().oracle_vaat_get_dim_f64(self._h, self.DIM_F64[name], _p(out))
            return out
        out = np.zeros((self.dim, self.nchains), np.int32)
        lib().oracle_vaat_get_dim_i32(self._h, self.DIM_I32[name], out.ctypes.data_as(_ip))
        return out


def hmc_gradient(kind, p, params=None):
    p = _f64(p); prm = like_params(kind, p.size, params); g = np.zeros_like(p)
    lib().oracle_hmc_gradient(kind, p.size, _p(p), _p(prm) if prm.size else None, _p(g))
    return g


# ---- autocorrelation of a saved trace (MakeAutocorrelation.C:108-148) ------------------------------
def autocorrelation_sums(x, centre=None, nlags=64):
    """x[slot][dim][chain].  The sums the macro's profile histograms hold (:112-122), pooled over chains and taken
    about `centre`: sum[d] = sum y, lagged[k][d] = sum_{t>=k} y_t y_(t-k)."""
    x = np.asarray(x, dtype=np.float64)
    y = x - (0.0 if centre is None else np.asarray(centre, dtype=np.float64)[None, :, None])
    n = y.shape[0]
    lagged = np.zeros((nlags, y.shape[1]))
    for k in range(min(nlags, n)):
        lagged[k] = (y[k:] * y[:n - k]).sum(axis=(0, 2))
    return y.sum(axis=(0, 2)), lagged


def autocorrelation_reference(series, maxlag):
    """MakeAutocorrelation.C:106-139 for one dimension of one chain, loop for loop: the ring buffer fill, then
    a(lag) = (v / e - mean^2) / err^2 with mean and err the profile histogram's mean and spread (option "s")."""
    series = np.asarray(series, dtype=np.float64)
    prod, count = np.zeros(maxlag), np.zeros(maxlag)
    for t, val in enumerate(series):
        fills = t + 1
        for lag in range(1, maxlag):
            if fills <= lag:
                break
            prod[lag] += series[t - lag] * val
            count[lag] += 1.0
    mean = series.mean()
    err2 = (series * series).mean() - mean * mean
    a = np.full(maxlag, np.nan)
    a[1:] = (prod[1:] / count[1:] - mean * mean) / err2
    return a
