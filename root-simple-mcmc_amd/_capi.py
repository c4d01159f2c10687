"""ctypes binding of the C ABI declared in include/smcmc.h.

This is the reference-side stub a Python caller needs; the C++ counterpart is
include/TSimpleMCMC_amd.H.  Loading fails loudly when the in-tree library is
missing: there is no CPU fallback behind this package.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libsmcmc_amd.so")

OK, ERR_INVALID, ERR_LOGIC, ERR_RUNTIME, ERR_BAD_START, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE = range(8)

LIKE_ISO_GAUSS, LIKE_QUADFORM, LIKE_ROSENBROCK, LIKE_USER = 0, 1, 2, 3
LIKE_ASYM, LIKE_HORRIFIC, LIKE_CONSTRAINED = 4, 5, 6   # the reference's stress targets (smcmc.h)
MODE_FROZEN, MODE_POOLED, MODE_PER_CHAIN = 0, 1, 2

PARAMS = ["COVARIANCE_WINDOW", "COVARIANCE_DEWEIGHT", "ACCEPTANCE_WINDOW", "ACCEPTANCE_DEWEIGHT",
          "ACCEPTANCE_RIGIDITY", "TARGET_ACCEPTANCE", "SIGMA", "MAXIMUM_CORRELATION", "STEP_RMS_WINDOW",
          "NEXT_UPDATE", "COVARIANCE_TRIALS", "CENTER_TRIALS", "COVARIANCE_TRACE", "TOTAL_STEPS",
          "SIGMA_TRACE", "UPDATE_COUNT", "LAST_UPDATE_PATH", "EXACT_ARITHMETIC", "MOMENT_STRIDE", "MOMENT_GROUP", "KEEP_PROPOSED",
          "DEVICE_UPDATE", "OVERLAP_UPDATE", "COVARIANCE_FROZEN", "DENSE_QUADFORM", "PERCHAIN_WAVE"]
P = {name: i for i, name in enumerate(PARAMS)}
RECORD_FIELDS = ["logl", "logl_proposed", "step_rms", "last_accept", "trials", "successes", "next_update", "acceptance",
                 "acceptance_trials", "sigma", "center_trials", "covariance_trials", "covariance_trace", "total_steps",
                 "update_status"]   # smcmc_record_field
LANE_F64 = {name: i for i, name in enumerate(
    ["logl", "sigma", "acceptance", "acceptance_trials", "rigidity", "last_value", "last_x0", "step_rms",
     "logl_proposed", "center_trials", "covariance_trials", "sigma_trace"])}
LANE_I32 = {name: i for i, name in enumerate(
    ["trials", "successes", "next_update", "naccept", "step_rms_trials", "last_accept", "update_status", "decomp_full",
     "chain_steps", "update_count", "last_update_path"])}
# the HMC engine's aliases (SMCMC_HMC_LANE_* of include/smcmc.h)
HMC_LANE_F64 = dict(LANE_F64, mean_epsilon=LANE_F64["sigma"], reversal_len=LANE_F64["rigidity"])
HMC_LANE_I32 = dict(LANE_I32, leapfrog=LANE_I32["next_update"], contributes=LANE_I32["successes"])
# the variable-at-a-time engine (SMCMC_VAAT_* of include/smcmc.h)
VAAT_LANE_I32 = dict(LANE_I32, last_index=LANE_I32["next_update"])
VAAT_LANE_F64 = dict(LANE_F64, proposed_value=LANE_F64["last_x0"])
VAAT_DIM_F64 = {"sigma": 0, "acceptance": 1}
VAAT_DIM_I32 = {"acceptance_trials": 2, "queue": 3}
HMC_TUNING = ["trace", "orbit", "updates", "cov_trials", "average_trials", "steps_remaining", "steps_since_update",
              "max_scale", "min_scale", "est_trace"]

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_H = C.c_void_p

class SavedState(C.Structure):
    """smcmc_saved_state: one entry of the reference's output tree."""
    _fields_ = [("log_likelihood", C.c_double), ("total_steps", C.c_int32), ("step_rms", C.c_double),
                ("trials", C.c_int32), ("successes", C.c_int32), ("next_update", C.c_int32),
                ("acceptance", C.c_double), ("acceptance_trials", C.c_double), ("sigma", C.c_double),
                ("central_point", _dp), ("central_point_trials", C.c_double),
                ("covariance", _dp), ("covariance_trials", C.c_double)]


# every symbol include/smcmc.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "smcmc_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(_H)]),
    "smcmc_destroy": (C.c_int, [_H]),
    "smcmc_last_error": (C.c_char_p, [_H]),
    "smcmc_status_string": (C.c_char_p, [C.c_int]),
    "smcmc_version": (C.c_int, []),
    "smcmc_max_register_dim": (C.c_int, []),
    "smcmc_max_dim": (C.c_int, []),
    "smcmc_set_stream": (C.c_int, [_H, C.c_void_p]),
    "smcmc_set_likelihood_params": (C.c_int, [_H, _dp, C.c_int]),
    "smcmc_set_mode": (C.c_int, [_H, C.c_int]),
    "smcmc_set_gaussian": (C.c_int, [_H, C.c_int, C.c_double]),
    "smcmc_set_uniform": (C.c_int, [_H, C.c_int, C.c_double, C.c_double]),
    "smcmc_set_scan_dimension": (C.c_int, [_H, C.c_int]),
    "smcmc_set_correlation": (C.c_int, [_H, C.c_int, C.c_int, C.c_double]),
    "smcmc_reset_correlations": (C.c_int, [_H]),
    "smcmc_set_param": (C.c_int, [_H, C.c_int, C.c_double]),
    "smcmc_get_param": (C.c_int, [_H, C.c_int, _dp]),
    "smcmc_start": (C.c_int, [_H, _dp, C.c_int]),
    "smcmc_restore": (C.c_int, [_H, _dp, C.c_int, C.POINTER(SavedState)]),
    "smcmc_step": (C.c_int, [_H, C.c_int, C.c_int]),
    "smcmc_step_save": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "smcmc_snapshot": (C.c_int, [_H]),
    "smcmc_rollback": (C.c_int, [_H]),
    "smcmc_record_stride": (C.c_int, [_H]),
    "smcmc_step_recorded": (C.c_int, [_H, C.c_int, C.c_int, C.c_int, _dp]),
    "smcmc_force_step": (C.c_int, [_H, _dp, C.c_int]),
    "smcmc_reduce_moments": (C.c_int, [_H]),
    "smcmc_moments_size": (C.c_int, [_H]),
    "smcmc_export_moments": (C.c_int, [_H, C.c_void_p]),
    "smcmc_import_moments": (C.c_int, [_H, C.c_void_p]),
    "smcmc_apply_moments": (C.c_int, [_H]),
    "smcmc_sync": (C.c_int, [_H]),
    "smcmc_comm_unique_id": (C.c_int, [C.c_void_p]),
    "smcmc_comm_init": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_int]),
    "smcmc_comm_destroy": (C.c_int, [_H]),
    "smcmc_comm_ranks": (C.c_int, [_H]),
    "smcmc_allreduce_moments": (C.c_int, [_H]),
    "smcmc_update_proposal": (C.c_int, [_H]),
    "smcmc_reset_proposal": (C.c_int, [_H]),
    "smcmc_nchains_padded": (C.c_int, [_H]),
    "smcmc_dim_padded": (C.c_int, [_H]),
    "smcmc_read_state": (C.c_int, [_H, _dp, _dp]),
    "smcmc_read_proposed": (C.c_int, [_H, _dp]),
    "smcmc_read_lane_f64": (C.c_int, [_H, C.c_int, _dp]),
    "smcmc_read_lane_i32": (C.c_int, [_H, C.c_int, _ip]),
    "smcmc_read_moments": (C.c_int, [_H, _dp]),
    "smcmc_get_center": (C.c_int, [_H, _dp]),
    "smcmc_set_center": (C.c_int, [_H, _dp]),
    "smcmc_get_covariance": (C.c_int, [_H, _dp]),
    "smcmc_set_covariance": (C.c_int, [_H, _dp]),
    "smcmc_get_decomposition": (C.c_int, [_H, _dp]),
    "smcmc_state_device_ptr": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "smcmc_read_chain": (C.c_int, [_H, C.c_int, _dp, _dp, _dp, _ip]),
    "smcmc_read_chain_proposal": (C.c_int, [_H, C.c_int, _dp, _dp, _dp]),
    "smcmc_hmc_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(_H)]),
    "smcmc_hmc_destroy": (C.c_int, [_H]),
    "smcmc_hmc_last_error": (C.c_char_p, [_H]),
    "smcmc_hmc_set_stream": (C.c_int, [_H, C.c_void_p]),
    "smcmc_hmc_set_likelihood_params": (C.c_int, [_H, _dp, C.c_int]),
    "smcmc_hmc_set_exact_arithmetic": (C.c_int, [_H, C.c_int]),
    "smcmc_hmc_set_alpha": (C.c_int, [_H, C.c_double]),
    "smcmc_hmc_set_mean_epsilon": (C.c_int, [_H, C.c_double]),
    "smcmc_hmc_get_mean_epsilon": (C.c_int, [_H, _dp]),
    "smcmc_hmc_set_leapfrog": (C.c_int, [_H, C.c_int]),
    "smcmc_hmc_get_leapfrog": (C.c_int, [_H, C.POINTER(C.c_int)]),
    "smcmc_hmc_set_sync_interval": (C.c_int, [_H, C.c_int]),
    "smcmc_hmc_set_track_covariance": (C.c_int, [_H, C.c_int]),
    "smcmc_hmc_set_gradient_type": (C.c_int, [_H, C.c_int]),
    "smcmc_hmc_get_gradient_type": (C.c_int, [_H]),
    "smcmc_hmc_moment_group": (C.c_int, [_H]),
    "smcmc_hmc_sync": (C.c_int, [_H]),
    "smcmc_hmc_moments_size": (C.c_int, [_H]),
    "smcmc_hmc_reduce_moments": (C.c_int, [_H]),
    "smcmc_hmc_export_moments": (C.c_int, [_H, C.c_void_p]),
    "smcmc_hmc_import_moments": (C.c_int, [_H, C.c_void_p]),
    "smcmc_hmc_apply_moments": (C.c_int, [_H]),
    "smcmc_hmc_get_tuning": (C.c_int, [_H, _dp]),
    "smcmc_hmc_get_average_point": (C.c_int, [_H, _dp]),
    "smcmc_hmc_get_covariance": (C.c_int, [_H, _dp]),
    "smcmc_hmc_start": (C.c_int, [_H, _dp, C.c_int]),
    "smcmc_hmc_step": (C.c_int, [_H, C.c_int]),
    "smcmc_hmc_read_state": (C.c_int, [_H, _dp, _dp, _dp]),
    "smcmc_hmc_copy_positions": (C.c_int, [_H, C.c_void_p]),
    "smcmc_hmc_nchains_padded": (C.c_int, [_H]),
    "smcmc_hmc_read_lane_f64": (C.c_int, [_H, C.c_int, _dp]),
    "smcmc_hmc_read_lane_i32": (C.c_int, [_H, C.c_int, _ip]),
    "smcmc_vaat_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(_H)]),
    "smcmc_vaat_destroy": (C.c_int, [_H]),
    "smcmc_vaat_last_error": (C.c_char_p, [_H]),
    "smcmc_vaat_set_stream": (C.c_int, [_H, C.c_void_p]),
    "smcmc_vaat_set_likelihood_params": (C.c_int, [_H, _dp, C.c_int]),
    "smcmc_vaat_set_exact_arithmetic": (C.c_int, [_H, C.c_int]),
    "smcmc_vaat_set_uniform": (C.c_int, [_H, C.c_int, C.c_double, C.c_double]),
    "smcmc_vaat_set_gaussian": (C.c_int, [_H, C.c_int, C.c_double]),
    "smcmc_vaat_set_acceptance_window": (C.c_int, [_H, C.c_double]),
    "smcmc_vaat_get_acceptance_window": (C.c_int, [_H, _dp]),
    "smcmc_vaat_set_acceptance_rigidity": (C.c_int, [_H, C.c_double]),
    "smcmc_vaat_get_acceptance_rigidity": (C.c_int, [_H, _dp]),
    "smcmc_vaat_set_step_rms_window": (C.c_int, [_H, C.c_int]),
    "smcmc_vaat_start": (C.c_int, [_H, _dp, C.c_int]),
    "smcmc_vaat_update_proposal": (C.c_int, [_H]),
    "smcmc_vaat_step": (C.c_int, [_H, C.c_int]),
    "smcmc_vaat_step_save": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "smcmc_vaat_total_steps": (C.c_int, [_H]),
    "smcmc_vaat_queue_length": (C.c_int, [_H]),
    "smcmc_vaat_nchains_padded": (C.c_int, [_H]),
    "smcmc_vaat_read_state": (C.c_int, [_H, _dp, _dp]),
    "smcmc_vaat_read_lane_f64": (C.c_int, [_H, C.c_int, _dp]),
    "smcmc_vaat_read_lane_i32": (C.c_int, [_H, C.c_int, _ip]),
    "smcmc_vaat_read_dim_f64": (C.c_int, [_H, C.c_int, _dp]),
    "smcmc_vaat_read_dim_i32": (C.c_int, [_H, C.c_int, _ip]),
    "smcmc_vaat_state_device_ptr": (C.c_int, [_H, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "smcmc_selftest_detmath": (C.c_int, [C.c_int, C.c_int, C.c_int, _dp, _dp, _dp]),
    "smcmc_selftest_mfma": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp]),
    "smcmc_selftest_mfma_strip": (C.c_int, [C.c_int, C.c_int, _dp, _dp, _dp]),
    "smcmc_autocorrelation_sums": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp, _dp,
                                             C.c_void_p]),
}
AUTOCORR_LAGS = 64
COMM_ID_BYTES = 128

_lib = None
_libs = {}


class SmcmcError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"smcmc status {status}: {message}")
        self.status = status


def _bind(path):
    # One HIP runtime per process: PyTorch-ROCm carries its own libamdhip64 and the library is linked against /opt/rocm's
    # (same SONAME).  Whichever is loaded first serves both -- but if this library came first and torch then initialised
    # its own copy, the process would hold two runtimes and the second sees no device ("no HIP device").  So torch, when
    # it is installed, is imported before the library is opened.
    try:
        import torch  # noqa: F401
    except Exception:   # not installed, or unusable: the library stands on /opt/rocm's runtime alone
        pass
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


def load(path=None):
    """Load the in-tree HIP library (or, with `path`, a build of it that carries a user likelihood:
    `build.py --user-likelihood`).  Raises if it has not been built."""
    global _lib
    if path is not None:
        path = os.path.abspath(path)
        if path not in _libs:
            if not os.path.exists(path):
                raise ImportError(f"{path} is missing: build it with `python root-simple-mcmc_amd/build.py "
                                  "--user-likelihood <header>`.  There is no CPU fallback.")
            _libs[path] = _bind(path)
        return _libs[path]
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python root-simple-mcmc_amd/build.py` "
                "(or __graft_entry__.build()).  There is no CPU fallback.")
        _lib = _bind(LIB_PATH)
    return _lib
