/* smcmc.h -- C ABI of the MI355X many-chain adaptive Metropolis engine.
 *
 * Drop-in boundary for ONE path of ClarkMcGrew/root-simple-mcmc: the
 * sMCMC::TSimpleMCMC<L, sMCMC::TProposeAdaptiveStep>::Step() loop.  The reference
 * exposes no FFI; its boundary is the C++ template surface of TSimpleMCMC.H:185-590
 * and the TProposeAdaptiveStep public methods (TSimpleMCMC.H:732-1003).  The
 * host-side mirror of that surface (include/TSimpleMCMC_amd.H and the Python
 * package root-simple-mcmc_amd/) is written on top of exactly these entry
 * points; each one names the reference member it stands behind.
 *
 * Conventions: every function returns an smcmc_status (0 = ok); no exception
 * crosses the ABI; vectors over chains are laid out [dim][chain] (chain index
 * fastest), host buffers are caller-owned, one engine per (device, stream),
 * calls on one engine are serialised by the caller.  Kernel launches are
 * asynchronous on the engine's stream; every smcmc_read_* / smcmc_get_* /
 * smcmc_apply_moments call synchronises that stream first.
 */
#ifndef SMCMC_H_SEEN
#define SMCMC_H_SEEN

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smcmc_engine smcmc_engine;

typedef enum {
    SMCMC_OK = 0,
    SMCMC_ERR_INVALID = 1,      /* std::invalid_argument in the reference (TSimpleMCMC.H:373,813,1153) */
    SMCMC_ERR_LOGIC = 2,        /* std::logic_error (TSimpleMCMC.H:666,673,1575) */
    SMCMC_ERR_RUNTIME = 3,      /* std::runtime_error (TSimpleMCMC.H:1027,1385,1479) */
    SMCMC_ERR_BAD_START = 4,    /* Start() returned false (TSimpleMCMC.H:265-268) */
    SMCMC_ERR_UNSUPPORTED = 5,  /* configuration the HIP path does not cover; never a CPU fallback */
    SMCMC_ERR_HIP = 6,          /* a HIP runtime call failed (smcmc_last_error has the text) */
    SMCMC_ERR_NO_DEVICE = 7
} smcmc_status;

/* Likelihood functors with a device implementation (the `UserLikelihood`
 * template argument of TSimpleMCMC.H:185). */
typedef enum {
    SMCMC_LIKE_ISO_GAUSS = 0,   /* README.md:57-66 / TSimpleMCMC.H:111-120: logL = sum -0.5 p_i^2 */
    SMCMC_LIKE_QUADFORM = 1,    /* TDummyLogLikelihood.H:21-31; params = Error matrix, dim*dim row-major */
    SMCMC_LIKE_ROSENBROCK = 2,  /* THardLogLikelihood.H:57-67; params = {ROSEN_B}, default 100 */
    /* A likelihood compiled in from user source (the `double operator()(const Vector&)` of the
     * reference's UserLikelihood concept, TSimpleMCMC.H:53-57, as a device function): a library built
     * with `python root-simple-mcmc_amd/build.py --user-likelihood my_likelihood.hip.h` carries it
     * (INTEGRATION.md); other builds answer SMCMC_ERR_UNSUPPORTED.  params = whatever the function reads (at most
     * dim_padded^2 doubles).  dim <= 63 with the register form (smcmc_user_loglike<DP>); a header that also defines
     * smcmc_user_loglike_at and SMCMC_USER_LIKELIHOOD_ANY_DIM is served up to dim = 512, above 63 in reference-order
     * arithmetic (SMCMC_P_EXACT_ARITHMETIC = 1).  Metropolis (smcmc_*) and variable-at-a-time (smcmc_vaat_*) engines. */
    SMCMC_LIKE_USER = 3,
    /* The reference's stress targets (Metropolis engine only; they have no gradient, TSimpleHMC.H:85-89):
     * TAsymLogLikelihood.H:20-31, params = {positiveSlope, negativeSlope} (default -1, 100);
     * THorrificLogLikelihood.H:26-38 (no parameters);
     * example4/TConstrainedLikelihood.H:26-46, params = {SummedValues, SummedConstraint, ExpectedValues[dim],
     * PriorConstraints[dim]}.
     * For dim > 63 all three run in reference-order arithmetic only. */
    SMCMC_LIKE_ASYM = 4,
    SMCMC_LIKE_HORRIFIC = 5,
    SMCMC_LIKE_CONSTRAINED = 6
} smcmc_likelihood;

/* How the proposal covariance adapts over the ensemble. */
typedef enum {
    SMCMC_MODE_FROZEN = 0,  /* SetCovarianceFrozen(true) (TSimpleMCMC.H:937): every chain is an
                               independent reference chain on the shared, fixed decomposition */
    SMCMC_MODE_POOLED = 1,  /* running centre/covariance (TSimpleMCMC.H:1780-1820) pooled over all
                               chains through batch moments; UpdateProposal at every smcmc_sync */
    SMCMC_MODE_PER_CHAIN = 2 /* the reference's own mode, per chain: every chain keeps its own centre, covariance and
                               decomposition ([k][chain] columns in device memory), runs UpdateState every step
                               (:1721-1831, the covariance loop of :1795-1820 included) and UpdateProposal when its own
                               --fNextUpdate < 1 on an accepted step (:1824-1826).  Every chain is bit for bit the
                               reference chain on its random stream.  dim <= 63, reference-order arithmetic, Gaussian
                               proposals in every dimension; memory 12 dim (dim + 1) + 32 dim bytes per chain.
                               The shared getters (centre, covariance, decomposition, trials, trace) answer for
                               chain 0, the setters act on every chain; smcmc_read_chain_proposal reads any chain. */
} smcmc_mode;

/* Scalar knobs / observables of TProposeAdaptiveStep and TSimpleMCMC. */
typedef enum {
    SMCMC_P_COVARIANCE_WINDOW = 0,      /* Set/GetCovarianceWindow        TSimpleMCMC.H:914-915 */
    SMCMC_P_COVARIANCE_DEWEIGHT = 1,    /* SetCovarianceUpdateDeweighting :927 */
    SMCMC_P_ACCEPTANCE_WINDOW = 2,      /* Set/GetAcceptanceWindow        :982-983 */
    SMCMC_P_ACCEPTANCE_DEWEIGHT = 3,    /* SetAcceptanceUpdateDeweighting :987 */
    SMCMC_P_ACCEPTANCE_RIGIDITY = 4,    /* Set/GetAcceptanceRigidity      :1002-1003 (set: all chains) */
    SMCMC_P_TARGET_ACCEPTANCE = 5,      /* Set/GetTargetAcceptance        :977-978 */
    SMCMC_P_SIGMA = 6,                  /* Get/SetSigma                   :770,775 (set: all chains; get: chain 0) */
    SMCMC_P_MAXIMUM_CORRELATION = 7,    /* SetMaximumCorrelation          :909 */
    SMCMC_P_STEP_RMS_WINDOW = 8,        /* SetStepRMSWindow               :511 */
    SMCMC_P_NEXT_UPDATE = 9,            /* Set/GetNextUpdate              :992-993 (FROZEN: all chains) */
    SMCMC_P_COVARIANCE_TRIALS = 10,     /* Get/SetCovarianceTrials        :942,947 */
    SMCMC_P_CENTER_TRIALS = 11,         /* Get/SetEstimatedCenterTrials   :741,747 */
    SMCMC_P_COVARIANCE_TRACE = 12,      /* GetCovarianceTrace             :961 (read only) */
    SMCMC_P_TOTAL_STEPS = 13,           /* fTotalSteps                    :554 (read only) */
    SMCMC_P_SIGMA_TRACE = 14,           /* fSigmaTrace                    :1960 (read only) */
    SMCMC_P_UPDATE_COUNT = 15,          /* number of UpdateProposal calls (diagnostic, read only) */
    SMCMC_P_LAST_UPDATE_PATH = 16,      /* 0 Cholesky 1 conditioned 2 eigen 3 emergency 4 reset (read only) */
    SMCMC_P_EXACT_ARITHMETIC = 17,      /* 1: reference operation order (default); 0: fused multiply-add */
    SMCMC_P_MOMENT_STRIDE = 18,         /* POOLED, dim > 63: fold the current point into the pooled moments every n-th step (default 1) */
    SMCMC_P_MOMENT_GROUP = 19,          /* chains per moment group (read only: 64, or the slice size of the dim > 63 path) */
    SMCMC_P_KEEP_PROPOSED = 20,         /* 1: every launch leaves the proposal of its last step on the device for
                                           smcmc_read_proposed (fProposed / GetProposed, TSimpleMCMC.H:514, 576); default 0 */
    SMCMC_P_DEVICE_UPDATE = 21,         /* 1 (default): the pooled update (running centre / covariance, sigma rescale, Cholesky, operand
                                         * layouts) runs on the device with no host synchronisation; only a failed decomposition brings
                                         * the host's fallback ladder (TSimpleMCMC.H:1134-1389) in.  0: all of it on the host.  Same bits. */
    SMCMC_P_OVERLAP_UPDATE = 22,        /* 1: the next launch does not wait for the status of the update before it;
                                         * identical results unless that update falls back to the ladder, which then takes effect one
                                         * window late.  0 (default): parity mode */
    SMCMC_P_COVARIANCE_FROZEN = 23,     /* Set/GetCovarianceFrozen :937-938 inside SMCMC_MODE_PER_CHAIN: the covariance loop is
                                         * skipped, the centre still runs (SMCMC_MODE_FROZEN is the shared-decomposition form) */
    SMCMC_P_DENSE_QUADFORM = 24,        /* SMCMC_LIKE_QUADFORM: 1 = always the dense D^2-term sum.  0 (default): when the Error matrix is
                                         * sparse (at most a quarter of its entries non-zero, full diagonal, all finite) the serial sums
                                         * of the reference-order kernels and of SMCMC_MODE_PER_CHAIN walk its non-zero entries only --
                                         * bit for bit the dense sum (a skipped term is +-0), dense again for a non-finite point.
                                         * Reads back 1 whenever the dense sum is what runs. */
    SMCMC_P_PERCHAIN_WAVE = 25,         /* SMCMC_MODE_PER_CHAIN: which kernel steps the chains.  1: one chain per WAVEFRONT (the chain's
                                         * covariance in registers, its decomposition in LDS for a whole launch: the kernel for few
                                         * chains, down to the single chain of SimpleMCMC.C); 0: one chain per lane (its O(D^2) state
                                         * streamed through HBM every step); -1 (default): per wavefront.  Same images, same
                                         * bits either way.  Every built-in likelihood runs on both; SMCMC_LIKE_USER (a library
                                         * built with a user likelihood) on the per-wavefront kernel only, whatever is set
                                         * here.  Reads back what runs. */
    SMCMC_P_COUNT_
} smcmc_param;

/* Per-chain double fields (smcmc_read_lane_f64). */
typedef enum {
    SMCMC_LANE_LOGL = 0,            /* fAcceptedLogLikelihood  TSimpleMCMC.H:568 */
    SMCMC_LANE_SIGMA = 1,           /* fSigma                  :1955 */
    SMCMC_LANE_ACCEPTANCE = 2,      /* fAcceptance             :1934 */
    SMCMC_LANE_ACCEPTANCE_TRIALS = 3,
    SMCMC_LANE_RIGIDITY = 4,
    SMCMC_LANE_LAST_VALUE = 5,      /* fLastValue              :1838 */
    SMCMC_LANE_LAST_X0 = 6,         /* fLastPoint[0]           :1835 */
    SMCMC_LANE_STEP_RMS = 7,        /* fStepRMS                :580 */
    SMCMC_LANE_LOGL_PROPOSED = 8,   /* fProposedLogLikelihood  :589 */
    /* SMCMC_MODE_PER_CHAIN only */
    SMCMC_LANE_CENTER_TRIALS = 9,   /* fCentralPointTrials     :1849 */
    SMCMC_LANE_COVARIANCE_TRIALS = 10, /* fCovarianceTrials    :1867 */
    SMCMC_LANE_SIGMA_TRACE = 11,    /* fSigmaTrace             :1960 */
    SMCMC_LANE_F64_COUNT_
} smcmc_lane_f64;

/* Per-chain int32 fields (smcmc_read_lane_i32). */
typedef enum {
    SMCMC_LANE_TRIALS = 0,          /* fTrials      :1922 */
    SMCMC_LANE_SUCCESSES = 1,       /* fSuccesses   :1926 */
    SMCMC_LANE_NEXT_UPDATE = 2,     /* fNextUpdate  :1930 */
    SMCMC_LANE_NACCEPT = 3,         /* sum of Step() return values */
    SMCMC_LANE_STEP_RMS_TRIALS = 4, /* fStepRMSTrials :583 */
    SMCMC_LANE_LAST_ACCEPT = 5,     /* Step() return value of the latest step */
    /* SMCMC_MODE_PER_CHAIN only */
    SMCMC_LANE_UPDATE_STATUS = 6,   /* 0 between calls (inside a launch: a chain waiting for the host's fallback ladder) */
    SMCMC_LANE_DECOMP_FULL = 7,     /* 1: fDecomposition is a full matrix (the eigen rung of the ladder, :1252-1321) */
    SMCMC_LANE_CHAIN_STEPS = 8,     /* fTotalSteps :554 of the chain */
    SMCMC_LANE_UPDATE_COUNT = 9,    /* UpdateProposal calls of the chain (diagnostic) */
    SMCMC_LANE_LAST_UPDATE_PATH = 10, /* rung of the latest UpdateProposal: 0 Cholesky 1 conditioned 2 eigen 3 emergency 4 reset */
    SMCMC_LANE_I32_COUNT_
} smcmc_lane_i32;

/* ---- lifetime ---------------------------------------------------------- */
/* TSimpleMCMC constructor (TSimpleMCMC.H:203) + SetDim (:786).  chain_offset is the
 * global id of local chain 0 (the Philox stream of chain c is keyed on
 * chain_offset + c), so an ensemble sharded over ranks draws the same numbers as
 * the unsharded one.  Fails with SMCMC_ERR_NO_DEVICE when no GPU is visible. */
int smcmc_create(int dim, int nchains, int likelihood, uint64_t seed, uint32_t chain_offset,
                 int device, smcmc_engine** out);
int smcmc_destroy(smcmc_engine* h);
const char* smcmc_last_error(const smcmc_engine* h);
const char* smcmc_status_string(int status);
/* Build facts, callable without a GPU: library version, compiled kernel families. */
int smcmc_version(void);
int smcmc_max_register_dim(void);   /* largest dim of the register-resident kernels (63) */
int smcmc_max_dim(void);            /* largest dim any kernel covers (512) */
/* hipStream_t to launch on (NULL = default stream). */
int smcmc_set_stream(smcmc_engine* h, void* hip_stream);

/* ---- configuration (before smcmc_start) -------------------------------- */
int smcmc_set_likelihood_params(smcmc_engine* h, const double* params, int count);   /* GetLogLikelihood() :239 */
int smcmc_set_mode(smcmc_engine* h, int mode);
int smcmc_set_gaussian(smcmc_engine* h, int dim, double sigma);                      /* SetGaussian :855 */
int smcmc_set_uniform(smcmc_engine* h, int dim, double minimum, double maximum);     /* SetUniform :833 */
/* SetScanDimension (TSimpleMCMC.H:820-830): while set, a step redraws only that dimension (about the
 * estimated centre, or uniformly) and leaves the proposal state alone; out of range = off. */
int smcmc_set_scan_dimension(smcmc_engine* h, int dim);
int smcmc_set_correlation(smcmc_engine* h, int dim1, int dim2, double correlation);  /* SetCorrelation :883 */
int smcmc_reset_correlations(smcmc_engine* h);                                       /* ResetCorrelations :874 */
int smcmc_set_param(smcmc_engine* h, int which, double value);
int smcmc_get_param(smcmc_engine* h, int which, double* value);

/* One entry of the reference's output tree (branches of TSimpleMCMC.H:208-215 and 1616-1626),
 * as SaveStep(true) writes it and Restore / RestoreState read it back (:282-352, :1501-1612). */
typedef struct smcmc_saved_state {
    double log_likelihood;              /* LogLikelihood */
    int32_t total_steps;                /* TotalSteps */
    double step_rms;                    /* StepRMS */
    int32_t trials;                     /* AdaptiveTrials */
    int32_t successes;                  /* AdaptiveSuccesses */
    int32_t next_update;                /* AdaptiveNextUpdate */
    double acceptance;                  /* AdaptiveAcceptance */
    double acceptance_trials;           /* AdaptiveAcceptanceTrials */
    double sigma;                       /* AdaptiveSigma */
    const double* central_point;        /* AdaptiveCentralPoint [dim] */
    double central_point_trials;        /* AdaptiveCentralPointTrials */
    const double* covariance;           /* AdaptiveCovariance: lower triangle, row major, dim (dim + 1) / 2 */
    double covariance_trials;           /* AdaptiveCovarianceTrials */
} smcmc_saved_state;
/* Restore(tree) (TSimpleMCMC.H:282-352, randomize = false) on a started engine: every chain
 * continues from `accepted` ([dim] broadcast or [dim][nchains]) with the saved state; the
 * likelihood is recomputed on the device and replaces the saved one when they differ by more
 * than 1E-4; the proposal is updated once (RestoreState, :1612). */
int smcmc_restore(smcmc_engine* h, const double* accepted, int broadcast, const smcmc_saved_state* state);

/* ---- the chain ---------------------------------------------------------- */
/* Start (TSimpleMCMC.H:246-276) + InitializeState (:1679-1714).  x0 is [dim] when
 * broadcast != 0, else [dim][nchains].  SMCMC_ERR_BAD_START when any chain's start
 * has a non-finite or < -0.999999E+10 log-likelihood. */
int smcmc_start(smcmc_engine* h, const double* x0, int broadcast);
/* nsteps x Step(save=false, metropolis) for every chain (TSimpleMCMC.H:370-496):
 * one kernel launch, state stays in registers between the steps. */
int smcmc_step(smcmc_engine* h, int nsteps, int metropolis);
/* Same, and after every `stride`-th step writes the accepted point and its
 * log-likelihood (the `Accepted` / `LogLikelihood` branches, TSimpleMCMC.H:208-210)
 * into caller-owned DEVICE buffers save_x[slot][dim_padded][nchains_padded] (rows
 * >= dim are zero), save_logl[slot][nchains_padded]; slots = nsteps / stride. */
/* nsteps x Step(false, metropolis) in one launch with a per-step record of one chain on the host: after every step
 * what TSimpleMCMC::Step() leaves in its members and SaveStep() reads -- records[step * smcmc_record_stride()]:
 * [0, dim) fAccepted, [dim, 2 dim) fProposed, [2 dim, 3 dim) the diagonal of the chain's covariance (GetCovarianceTrace,
 * TSimpleMCMC.H:961-967, is its sum in index order: the reader adds it up; SMCMC_REC_COVARIANCE_TRACE itself is 0), then
 * the scalars of smcmc_record_field.  SMCMC_MODE_PER_CHAIN with the
 * one-chain-per-wavefront kernel (SMCMC_P_PERCHAIN_WAVE); SMCMC_ERR_UNSUPPORTED otherwise.  include/TSimpleMCMC_amd.H
 * runs Step() ahead with it (TSimpleMCMC.H:370-496 one call at a time is one launch and three read-backs per step). */
typedef enum {
    SMCMC_REC_LOGL = 0, SMCMC_REC_LOGL_PROPOSED, SMCMC_REC_STEP_RMS, SMCMC_REC_LAST_ACCEPT, SMCMC_REC_TRIALS,
    SMCMC_REC_SUCCESSES, SMCMC_REC_NEXT_UPDATE, SMCMC_REC_ACCEPTANCE, SMCMC_REC_ACCEPTANCE_TRIALS, SMCMC_REC_SIGMA,
    SMCMC_REC_CENTER_TRIALS, SMCMC_REC_COVARIANCE_TRIALS, SMCMC_REC_COVARIANCE_TRACE, SMCMC_REC_TOTAL_STEPS,
    SMCMC_REC_UPDATE_STATUS, SMCMC_REC_COUNT_
} smcmc_record_field;
/* smcmc_snapshot remembers the state of the whole ensemble on the device (points, per-chain scalars, every chain's
 * centre / covariance / decomposition, the step count); smcmc_rollback returns to it, any number of times.  Draws are
 * keyed on (chain, step), so stepping again from a snapshot repeats the same steps: TSimpleMCMC_amd.H's Step() runs
 * ahead of its caller with smcmc_step_recorded and takes the steps it ran too far back this way when a setter,
 * UpdateProposal() or SaveStep(true) needs the state AT the caller's step.  A started SMCMC_MODE_PER_CHAIN ensemble;
 * settings changed in between (smcmc_set_param ...) are not part of the snapshot. */
int smcmc_snapshot(smcmc_engine* h);
int smcmc_rollback(smcmc_engine* h);
int smcmc_record_stride(const smcmc_engine* h);      /* 3 dim + SMCMC_REC_COUNT_ */
int smcmc_step_recorded(smcmc_engine* h, int nsteps, int metropolis, int chain, double* records);
int smcmc_step_save(smcmc_engine* h, int nsteps, int metropolis, int stride,
                    double* save_x_device, double* save_logl_device);
/* ForceStep (TSimpleMCMC.H:811-817): the next step of every chain proposes
 * exactly `point` ([dim] broadcast or [dim][nchains]) without updating the
 * proposal state. */
int smcmc_force_step(smcmc_engine* h, const double* point, int broadcast);

/* ---- adaptation --------------------------------------------------------- */
/* Pooled covariance exchange, POOLED mode.  reduce: sum the per-wavefront moment
 * accumulators into the packed device vector M[(dim+1)(dim+2)/2] (row i <= dim,
 * col j <= i, row `dim` holding sum(y) and the point count) and clear them.
 * export/import copy M to/from a caller-owned DEVICE buffer so the caller can
 * all-reduce it over ranks (RCCL).  apply: running centre/covariance update fed
 * with the batch (TSimpleMCMC.H:1780-1820) and UpdateProposal (:1009-1390).
 * smcmc_sync = reduce + apply, the single-GPU form. */
int smcmc_reduce_moments(smcmc_engine* h);
int smcmc_moments_size(const smcmc_engine* h);
int smcmc_export_moments(smcmc_engine* h, double* dst_device);
int smcmc_import_moments(smcmc_engine* h, const double* src_device);
int smcmc_apply_moments(smcmc_engine* h);
int smcmc_sync(smcmc_engine* h);
/* Native exchange for callers without torch.distributed (the C++ mirror): one RCCL communicator per engine,
 * one rank per GPU.  Rank 0 makes the id (SMCMC_COMM_ID_BYTES bytes, an ncclUniqueId) and hands it to the
 * other ranks by its own means (file, socket, MPI); every rank then calls smcmc_comm_init, which blocks until
 * all nranks have arrived.  With a communicator attached smcmc_sync is reduce + ncclAllReduce(sum, f64) of M
 * on the engine's stream + apply (the one collective of the path), and
 * smcmc_allreduce_moments is the middle step on its own.  RCCL is loaded at the first call (dlopen), so the
 * library has no link-time dependency on it; SMCMC_ERR_UNSUPPORTED when it cannot be loaded. */
#define SMCMC_COMM_ID_BYTES 128
int smcmc_comm_unique_id(void* id_out);
int smcmc_comm_init(smcmc_engine* h, const void* id, int rank, int nranks);
int smcmc_comm_destroy(smcmc_engine* h);
int smcmc_comm_ranks(smcmc_engine* h);         /* ncclCommCount of the attached communicator; 0 without one, < 0 on error */
int smcmc_allreduce_moments(smcmc_engine* h);
int smcmc_update_proposal(smcmc_engine* h);   /* UpdateProposal() :1009 on the shared proposal */
int smcmc_reset_proposal(smcmc_engine* h);    /* ResetProposal()  :1396 */

/* ---- read back ---------------------------------------------------------- */
int smcmc_nchains_padded(const smcmc_engine* h);          /* nchains rounded up to 64 */
int smcmc_dim_padded(const smcmc_engine* h);              /* rows of the device state: the kernel family's register-array size */
int smcmc_read_state(smcmc_engine* h, double* x, double* logl);   /* GetAccepted :502, x[dim][nchains] */
/* GetProposed :514, x[dim][nchains]: the point the latest step proposed (after Start / Restore: the start point).
 * Needs SMCMC_P_KEEP_PROPOSED = 1 (SMCMC_ERR_LOGIC otherwise). */
int smcmc_read_proposed(smcmc_engine* h, double* x);
int smcmc_read_lane_f64(smcmc_engine* h, int field, double* out);
int smcmc_read_lane_i32(smcmc_engine* h, int field, int32_t* out);
int smcmc_read_moments(smcmc_engine* h, double* out);     /* host copy of M */
int smcmc_get_center(smcmc_engine* h, double* out);       /* GetEstimatedCenter :732 */
int smcmc_set_center(smcmc_engine* h, const double* in);  /* SetEstimatedCenter :733 */
int smcmc_get_covariance(smcmc_engine* h, double* out);   /* fCurrentCov, dim*dim */
int smcmc_set_covariance(smcmc_engine* h, const double* in);
int smcmc_get_decomposition(smcmc_engine* h, double* out);/* fDecomposition, dim*dim */
/* One chain's members, as the host mirror of the reference class keeps them for chain 0 after every Step(): any
 * pointer may be NULL.  x[dim] = fAccepted; proposed[dim] = fProposed (needs SMCMC_P_KEEP_PROPOSED outside
 * SMCMC_MODE_PER_CHAIN, which always keeps it); lanes_f64[SMCMC_LANE_F64_COUNT_], lanes_i32[SMCMC_LANE_I32_COUNT_]. */
int smcmc_read_chain(smcmc_engine* h, int chain, double* x, double* proposed, double* lanes_f64, int32_t* lanes_i32);
/* SMCMC_MODE_PER_CHAIN: the adaptive state of one chain -- fCentralPoint [dim], fCurrentCov [dim*dim],
 * fDecomposition [dim*dim] (TSimpleMCMC.H:1841-1893); any pointer may be NULL.  In the other modes the shared
 * proposal's (the same for every chain). */
int smcmc_read_chain_proposal(smcmc_engine* h, int chain, double* centre, double* covariance, double* decomposition);
/* device pointers for zero-copy consumers (x: [dim_padded][nchains_padded], logl: [nchains_padded]) */
int smcmc_state_device_ptr(smcmc_engine* h, double** x, double** logl);

/* ---- Hamiltonian Monte Carlo (sMCMC::TSimpleHMC, reference TSimpleHMC.H:119-973) ------ */
/* Every chain is a TSimpleHMC chain with the analytic gradient of the device likelihood.
 * SetMeanEpsilon(negative value) keeps |epsilon| fixed (every update of fMeanEpsilon is
 * guarded by fMeanEpsilon > 0, TSimpleHMC.H:304-343, 833-846) and SetLeapFrog(n) fixes the
 * leapfrog count (:190, 302): with both the chains are independent and a launch runs many
 * steps.  Otherwise (the reference's default) the chains retune themselves, see
 * smcmc_hmc_set_sync_interval.
 * Per-chain columns reuse smcmc_lane_f64 / smcmc_lane_i32: LOGL = -fAcceptedPotential,
 * LOGL_PROPOSED = -fProposedPotential, ACCEPTANCE = fCurrentAcceptance, TRIALS = fStepCount,
 * NACCEPT, LAST_ACCEPT, and the SMCMC_HMC_LANE_* aliases below. */
#define SMCMC_HMC_LANE_MEAN_EPSILON SMCMC_LANE_SIGMA          /* f64: fMeanEpsilon  (TSimpleHMC.H:888) */
#define SMCMC_HMC_LANE_REVERSAL_LEN SMCMC_LANE_RIGIDITY       /* f64: fReversalLen  (:894) */
#define SMCMC_HMC_LANE_LEAPFROG SMCMC_LANE_NEXT_UPDATE        /* i32: fLeapFrogSteps (:891) */
#define SMCMC_HMC_LANE_CONTRIBUTES SMCMC_LANE_SUCCESSES       /* i32: the latest step fed UpdateCovariance (:336) */
typedef struct smcmc_hmc smcmc_hmc;
int smcmc_hmc_create(int dim, int nchains, int likelihood, uint64_t seed, uint32_t chain_offset, int device,
                     smcmc_hmc** out);                                   /* TSimpleHMC ctor :130 */
int smcmc_hmc_destroy(smcmc_hmc* h);
const char* smcmc_hmc_last_error(const smcmc_hmc* h);
int smcmc_hmc_set_stream(smcmc_hmc* h, void* hip_stream);
int smcmc_hmc_set_likelihood_params(smcmc_hmc* h, const double* params, int count);   /* GetLogLikelihood :157 */
int smcmc_hmc_set_alpha(smcmc_hmc* h, double alpha);                     /* SetAlpha :175 */
/* 1 (default): the reference's operation order throughout.  0: the gradient of the quadratic-form
 * likelihood (TDummyLogLikelihood.H:34-42) sums its terms in the same order with one fused multiply-add
 * each, on the FP64 matrix pipe (dim <= 512); nothing else changes.  Before smcmc_hmc_start. */
int smcmc_hmc_set_exact_arithmetic(smcmc_hmc* h, int exact);
int smcmc_hmc_set_mean_epsilon(smcmc_hmc* h, double epsilon);            /* SetMeanEpsilon :181 (after Start, which resets it to 0.05) */
int smcmc_hmc_get_mean_epsilon(smcmc_hmc* h, double* epsilon);           /* GetMeanEpsilon :184 */
int smcmc_hmc_set_leapfrog(smcmc_hmc* h, int steps);                     /* SetLeapFrog :190 */
int smcmc_hmc_get_leapfrog(smcmc_hmc* h, int* steps);                   /* fLeapFrogSteps of chain 0, signed as :190 keeps it */
/* The covariance-driven tuning (UpdateCovariance :665-695, UpdateErrorMatrix :703-858) is pooled over the ensemble:
 * every chain folds the point it stood on into moment groups of smcmc_hmc_moment_group() chains each step, and every
 * `steps` steps (default 1) the pooled running covariance is brought up to date and UpdateErrorMatrix runs once; when
 * it goes through, every chain takes the new step length and leapfrog count (:833-847).  One chain and an interval of
 * one step is the reference chain.  Runs whenever the step length or the leapfrog count is not fixed;
 * smcmc_hmc_set_track_covariance(h, 1) keeps it running for a fixed step too (the Trace / Orbit outputs). */
int smcmc_hmc_set_sync_interval(smcmc_hmc* h, int steps);
/* Step(save, gradientType), TSimpleHMC.H:279 / PotentialGradient :467-532.  0, 1, 4: the likelihood's own gradient;
 * 2: CovariantGradient (:447-454) from the pooled running covariance (tracked from then on); 3: FiniteDifferenceGradient
 * (:417-444); 5: zero.  Types 2, 3, 5 need reference-order arithmetic (SMCMC_ERR_UNSUPPORTED otherwise). */
int smcmc_hmc_set_gradient_type(smcmc_hmc* h, int type);
int smcmc_hmc_get_gradient_type(const smcmc_hmc* h);
int smcmc_hmc_set_track_covariance(smcmc_hmc* h, int on);
int smcmc_hmc_moment_group(const smcmc_hmc* h);
int smcmc_hmc_sync(smcmc_hmc* h);                                        /* the pooled update now (end of a run) */
/* The same in pieces, so that an ensemble sharded over engines / ranks pools its covariance as the Metropolis engine's does:
 * reduce (the moment groups of the steps since the last update summed into the packed device vector M[(dim+1)(dim+2)/2]:
 * sum x x^T by rows j <= i, then sum x and the number of points), export / import (copy M to / from a caller-owned DEVICE
 * buffer: the caller adds the ranks' vectors, e.g. one RCCL all-reduce), apply (UpdateCovariance fed with the batch, then
 * UpdateErrorMatrix, the same on every rank).  Set the sync interval beyond the steps of a window and call these at its
 * end; smcmc_hmc_sync = reduce + apply. */
/* (one reduction at a time: a second reduce -- explicit, or the sync a step triggers when the interval runs out --
 * before the apply of the first is SMCMC_ERR_LOGIC; a sharded run sets the interval beyond its window and syncs itself) */
int smcmc_hmc_moments_size(const smcmc_hmc* h);
int smcmc_hmc_reduce_moments(smcmc_hmc* h);
int smcmc_hmc_export_moments(smcmc_hmc* h, double* dst_device);
int smcmc_hmc_import_moments(smcmc_hmc* h, const double* src_device);
int smcmc_hmc_apply_moments(smcmc_hmc* h);
/* out[10]: fCurrentCovarianceTrace, fEstimatedOrbitLength, updates that went through, fCovarianceTrials,
 * fAveragePointTrials, fStepsRemaining, fStepsSinceUpdate, max scale, min scale, fEstimatedCovarianceTrace */
int smcmc_hmc_get_tuning(smcmc_hmc* h, double* out);
int smcmc_hmc_get_average_point(smcmc_hmc* h, double* out);              /* fAveragePoint [dim] */
int smcmc_hmc_get_covariance(smcmc_hmc* h, double* out);                 /* GetEstimatedCovariance :197, dim*dim */
int smcmc_hmc_start(smcmc_hmc* h, const double* x0, int broadcast);      /* Start :210-269 */
int smcmc_hmc_step(smcmc_hmc* h, int nsteps);                            /* nsteps x Step(false) :279-401 */
int smcmc_hmc_read_state(smcmc_hmc* h, double* q, double* momentum, double* logl);   /* fAccepted, fAcceptedMomentum */
/* fAccepted of every chain into a DEVICE buffer [dim][smcmc_hmc_nchains_padded] (one slot of a trace for
 * smcmc_autocorrelation_sums), on the engine's stream: what SimpleHMC.C:51-66 fills its tree with, without the trip to the host. */
int smcmc_hmc_copy_positions(smcmc_hmc* h, double* dst_device);
int smcmc_hmc_nchains_padded(const smcmc_hmc* h);
int smcmc_hmc_read_lane_f64(smcmc_hmc* h, int field, double* out);
int smcmc_hmc_read_lane_i32(smcmc_hmc* h, int field, int32_t* out);

/* ---- variable-at-a-time chains: TSimpleMCMC<L, TProposeVAATStep> ---------
 * N independent chains of sMCMC::TSimpleMCMC<L, sMCMC::TProposeVAATStep> (TProposeVAATStep.H:22-307, the proposal
 * SimpleVAAT.C drives): one coordinate per step from a shuffled queue of the dimensions (:52-78, 177-195), a proposal
 * width per dimension adapted to a 44 % acceptance (:219-255).  Nothing is shared between chains; chain c is the
 * reference chain on the random stream (seed, chain_offset + c).  Every likelihood id of smcmc_likelihood (USER in a
 * library built with one); dim <= 512.  The reference's quirks are kept: Start resets the acceptance window to 100
 * the first time (:211), SetGaussian's sigma is the width of Gaus() unsquared (:69-78), RestoreState / AttachState /
 * SaveState do nothing (:33-36). */
typedef struct smcmc_vaat smcmc_vaat;
int smcmc_vaat_create(int dim, int nchains, int likelihood, uint64_t seed, uint32_t chain_offset, int device,
                      smcmc_vaat** out);
int smcmc_vaat_destroy(smcmc_vaat* h);
const char* smcmc_vaat_last_error(const smcmc_vaat* h);
int smcmc_vaat_set_stream(smcmc_vaat* h, void* hip_stream);
int smcmc_vaat_set_likelihood_params(smcmc_vaat* h, const double* params, int count);
int smcmc_vaat_set_exact_arithmetic(smcmc_vaat* h, int exact);           /* 0: fused multiply-add order; before Start */
int smcmc_vaat_set_uniform(smcmc_vaat* h, int dim, double minimum, double maximum);   /* SetUniform  :101 */
int smcmc_vaat_set_gaussian(smcmc_vaat* h, int dim, double sigma);                    /* SetGaussian :123 */
int smcmc_vaat_set_acceptance_window(smcmc_vaat* h, double a);           /* SetAcceptanceWindow :137 (an int member) */
int smcmc_vaat_get_acceptance_window(const smcmc_vaat* h, double* a);
int smcmc_vaat_set_acceptance_rigidity(smcmc_vaat* h, double r);         /* SetAcceptanceRigidity :148 */
int smcmc_vaat_get_acceptance_rigidity(const smcmc_vaat* h, double* r);
int smcmc_vaat_set_step_rms_window(smcmc_vaat* h, int window);           /* TSimpleMCMC::SetStepRMSWindow */
int smcmc_vaat_start(smcmc_vaat* h, const double* x0, int broadcast);    /* TSimpleMCMC::Start + InitializeState :198 */
int smcmc_vaat_update_proposal(smcmc_vaat* h);                           /* UpdateProposal :177 (SimpleVAAT.C:44) */
int smcmc_vaat_step(smcmc_vaat* h, int nsteps);                          /* nsteps x TSimpleMCMC::Step(false) */
/* as smcmc_step_save: the accepted point [slot][dim][npad] (and logL [slot][npad]) after every stride-th step */
int smcmc_vaat_step_save(smcmc_vaat* h, int nsteps, int stride, double* save_x_device, double* save_logl_device);
int smcmc_vaat_total_steps(const smcmc_vaat* h);
int smcmc_vaat_queue_length(const smcmc_vaat* h);                        /* entries left in fNextIndex */
int smcmc_vaat_nchains_padded(const smcmc_vaat* h);
int smcmc_vaat_read_state(smcmc_vaat* h, double* x, double* logl);       /* GetAccepted, x[dim][nchains] */
/* lanes: SMCMC_LANE_LOGL / LAST_VALUE / STEP_RMS / LOGL_PROPOSED; TRIALS / SUCCESSES / NACCEPT / STEP_RMS_TRIALS /
 * LAST_ACCEPT and SMCMC_VAAT_LANE_LAST_INDEX (fLastIndex) */
#define SMCMC_VAAT_LANE_LAST_INDEX SMCMC_LANE_NEXT_UPDATE
#define SMCMC_VAAT_LANE_PROPOSED_VALUE SMCMC_LANE_LAST_X0   /* f64: fProposed[fLastIndex] of the latest step */
int smcmc_vaat_read_lane_f64(smcmc_vaat* h, int field, double* out);
int smcmc_vaat_read_lane_i32(smcmc_vaat* h, int field, int32_t* out);
/* per-dimension state as [dim][nchains] */
#define SMCMC_VAAT_DIM_SIGMA 0              /* f64 fSigma            :296 */
#define SMCMC_VAAT_DIM_ACCEPTANCE 1         /* f64 fAcceptance       :287 */
#define SMCMC_VAAT_DIM_ACCEPTANCE_TRIALS 2  /* i32 fAcceptanceTrials :290 */
#define SMCMC_VAAT_DIM_QUEUE 3              /* i32 fNextIndex        :266 (slots >= queue length are stale) */
int smcmc_vaat_read_dim_f64(smcmc_vaat* h, int field, double* out);
int smcmc_vaat_read_dim_i32(smcmc_vaat* h, int field, int32_t* out);
int smcmc_vaat_state_device_ptr(smcmc_vaat* h, double** x, double** logl);

/* ---- self test (no engine needed) --------------------------------------- */
/* Runs every function of include/smcmc_detmath.h on the device for n inputs so
 * tests can compare device and host bit for bit.  kind: 0 log, 1 exp,
 * 2 sin(2 pi x), 3 cos(2 pi x), 4 pow_small(x, y), 5 sqrt, 6 x / y, 7 / 8 the two normals of
 * the Box-Muller pair made from the 32-bit words x, y, 9 sqrt_mid. */
int smcmc_selftest_detmath(int device, int kind, int n, const double* x, const double* y, double* out);
/* One v_mfma_f64_16x16x4_f64 chain: c[16][16] = sum_k a[16][k] b[k][16] over K
 * (multiple of 4), the accumulation order the pooled moments rely on. */
int smcmc_selftest_mfma(int device, int K, const double* a, const double* b, double* c);
/* The same for v_mfma_f64_4x4x4_4b_f64 as the moment fold uses it: c[4][16] = sum_k a[4][k] b[k][16]. */
int smcmc_selftest_mfma_strip(int device, int K, const double* a, const double* b, double* c);

/* ---- posterior reducer: autocorrelation of a saved trace ----------------------
 * MakeAutocorrelation.C:108-148 defines a(lag) = (E[x_t x_(t-lag)] - mean^2) / var per dimension.
 * This takes its sums on the device from a trace smcmc_step_save wrote
 * (trace_device[slot][dim_stride][nchains_padded]), pooled over slots and chains, about the
 * reference point `centre` ([dim] host; NULL = 0 is the macro's definition, another point changes a(lag)
 * only through the edges of the lagged sums, O(lag / nslots), and conditions E[xx] - mean^2 better):
 *   sum[d]       = sum_{t, c} y            y = x[t][d][c] - centre[d]
 *   lagged[k][d] = sum_{t >= k, c} y_t y_(t-k)     k = 0 .. SMCMC_AUTOCORR_LAGS - 1
 * (host outputs; the number of terms is (nslots - k) * nchains).  Raw sums so that ranks can add
 * theirs.  Fixed summation order: the same bits on every run.  `stream` is a hipStream_t or NULL. */
#define SMCMC_AUTOCORR_LAGS 64
int smcmc_autocorrelation_sums(const double* trace_device, int nslots, int dim, int dim_stride, int nchains,
                               int nchains_padded, const double* centre, double* sum, double* lagged, void* stream);

#ifdef __cplusplus
}
#endif
#endif
