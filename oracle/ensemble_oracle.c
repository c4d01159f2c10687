/* ensemble_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the
 * many-chain engine semantics that the HIP path implements (DESIGN.md,
 * "Ensemble semantics").  The reference has no many-chain mode (its only
 * parallelism is continue-chain.sh launching processes), so this part restates
 * the build's own definition; what it inherits from the reference, verbatim and
 * per chain, is:
 *   - the scalar half of TProposeAdaptiveStep::UpdateState (TSimpleMCMC.H:1723-1776)
 *   - the proposal draw x' = x + sigma U^T r (TSimpleMCMC.H:709-724)
 *   - TSimpleMCMC::Step's StepRMS window, likelihood call and Metropolis test
 *     (TSimpleMCMC.H:391-406, 410, 432-463, 484-491)
 *   - UpdateProposal / ResetProposal / InitializeState on the shared proposal
 *     (TSimpleMCMC.H:1009-1390, 1396-1494, 1679-1714)
 * so that in FROZEN mode every chain is bit for bit an oracle_chain with
 * SetCovarianceFrozen(true) (tests/test_oracle_ensemble.py).  PARITY UNPINNED,
 * as oracle_core.h explains.
 *
 * Pooled moments (mode POOLED): chains are taken in groups of 64 (local chain
 * index / 64).  With y = x - c0 (c0 = shared centre at window start) and
 * y[D] = 1, each group keeps acc[i][j] (0 <= j <= i <= D) and folds one chain
 * after another, one step after another, with a fused multiply-add:
 *     acc[i][j] = fma(y_k[i], y_k[j], acc[i][j]),  k = 0..63 ascending.
 * Group sums are added in group order within chunks of 32 groups, chunk sums in
 * chunk order.  This is the order the wave-level v_mfma_f64_16x16x4_f64
 * accumulation and the two-level reduction kernels produce on the GPU.
 */
#include "oracle_core.h"

enum { ENS_MODE_FROZEN = 0, ENS_MODE_POOLED = 1 };

typedef struct {
    int nchains, dim;
    int like_kind;
    double* like_params;
    uint64_t seed;
    uint32_t chain_offset;
    int mode;
    int exact;                 /* 1: reference operation order, 0: fused multiply-add order */
    int quadform_rowwise;      /* fused order at D > 63: log L = -sum_i 0.5 p_i (sum_j fma(E(i,j), p_j, .)) */
    oracle_proposal prop;      /* shared proposal: U, cov, centre, windows, target ... */
    int total_steps;
    /* per chain, [d][chain] for vectors */
    double* x;
    double* logl;
    double* sigma;
    double* acceptance;
    double* acceptance_trials;
    double* rigidity;
    int32_t* trials;
    int32_t* successes;
    int32_t* next_update;
    double* last_value;
    double* last_x0;
    double* step_rms;
    int32_t* step_rms_trials;
    int32_t* naccept;          /* Step() return values summed */
    uint8_t* last_accept;      /* Step() return of the latest step */
    double* last_logl_proposed;
    int step_rms_window;
    /* moments */
    int ngroups;
    int moment_group;          /* chains per moment group: 64, or the slice size of the dim > 63 path */
    int moment_stride;         /* fold the point seen at the start of step t when (t-1) % stride == 0 */
    double* acc;               /* [group][packed (D+1)(D+2)/2] */
    double* c0;                /* centre the moments are taken about */
    double last_sigma_scale;   /* sigma rescale factor of the latest pooled update (diagnostic) */
} oracle_ensemble;

static int ens_npacked(int dim) { return (dim + 1) * (dim + 2) / 2; }

oracle_ensemble* oracle_ensemble_create(int nchains, int dim, int like_kind, const double* like_params,
                                        int n_like_params, uint64_t seed, uint32_t chain_offset,
                                        int mode, int exact) {
    oracle_ensemble* e = (oracle_ensemble*)calloc(1, sizeof(oracle_ensemble));
    size_t n = (size_t)nchains, d = (size_t)dim;
    e->nchains = nchains; e->dim = dim; e->like_kind = like_kind;
    if (n_like_params > 0) {
        e->like_params = (double*)malloc(sizeof(double) * (size_t)n_like_params);
        memcpy(e->like_params, like_params, sizeof(double) * (size_t)n_like_params);
    }
    e->seed = seed; e->chain_offset = chain_offset; e->mode = mode; e->exact = exact;
    oracle_proposal_init(&e->prop);
    oracle_proposal_set_dim(&e->prop, dim);
    e->prop.cov_frozen = (mode == ENS_MODE_FROZEN);
    e->x = (double*)calloc(n * d, sizeof(double));
    e->logl = (double*)calloc(n, sizeof(double));
    e->sigma = (double*)calloc(n, sizeof(double));
    e->acceptance = (double*)calloc(n, sizeof(double));
    e->acceptance_trials = (double*)calloc(n, sizeof(double));
    e->rigidity = (double*)calloc(n, sizeof(double));
    e->trials = (int32_t*)calloc(n, sizeof(int32_t));
    e->successes = (int32_t*)calloc(n, sizeof(int32_t));
    e->next_update = (int32_t*)calloc(n, sizeof(int32_t));
    e->last_value = (double*)calloc(n, sizeof(double));
    e->last_x0 = (double*)calloc(n, sizeof(double));
    e->step_rms = (double*)calloc(n, sizeof(double));
    e->step_rms_trials = (int32_t*)calloc(n, sizeof(int32_t));
    e->naccept = (int32_t*)calloc(n, sizeof(int32_t));
    e->last_accept = (uint8_t*)calloc(n, sizeof(uint8_t));
    e->last_logl_proposed = (double*)calloc(n, sizeof(double));
    e->step_rms_window = 1000;
    e->moment_group = 64;
    e->moment_stride = 1;
    e->ngroups = (nchains + 63) / 64;
    e->acc = (double*)calloc((size_t)e->ngroups * (size_t)ens_npacked(dim), sizeof(double));
    e->c0 = (double*)calloc(d, sizeof(double));
    e->last_sigma_scale = 1.0;
    return e;
}

void oracle_ensemble_destroy(oracle_ensemble* e) {
    if (!e) return;
    oracle_proposal_free(&e->prop);
    free(e->like_params); free(e->x); free(e->logl); free(e->sigma); free(e->acceptance);
    free(e->acceptance_trials); free(e->rigidity); free(e->trials); free(e->successes);
    free(e->next_update); free(e->last_value); free(e->last_x0); free(e->step_rms);
    free(e->step_rms_trials); free(e->naccept); free(e->last_accept); free(e->last_logl_proposed);
    free(e->acc); free(e->c0); free(e);
}

/* moment grouping of the large-dimension HIP path (before start) */
void oracle_ensemble_set_moment_grouping(oracle_ensemble* e, int group_chains, int stride) {
    e->moment_group = group_chains;
    e->moment_stride = stride;
    e->ngroups = (e->nchains + group_chains - 1) / group_chains;
    free(e->acc);
    e->acc = (double*)calloc((size_t)e->ngroups * (size_t)ens_npacked(e->dim), sizeof(double));
}

/* shared-proposal setters (before start) */
void oracle_ensemble_set_quadform_rowwise(oracle_ensemble* e, int f) { e->quadform_rowwise = f; }
void oracle_ensemble_set_gaussian(oracle_ensemble* e, int d, double sigma) { e->prop.ptype[d] = 0; e->prop.pparam1[d] = sigma * sigma; }
void oracle_ensemble_set_uniform(oracle_ensemble* e, int d, double lo, double hi) { e->prop.ptype[d] = 1; e->prop.pparam1[d] = lo; e->prop.pparam2[d] = hi; }
void oracle_ensemble_set_correlation(oracle_ensemble* e, int d1, int d2, double c) { oracle_proposal_set_correlation(&e->prop, d1, d2, c); }
void oracle_ensemble_set_covariance_window(oracle_ensemble* e, double w) { e->prop.cov_window = w; }
void oracle_ensemble_set_covariance_deweight(oracle_ensemble* e, double d) { e->prop.cov_deweight = d; }
void oracle_ensemble_set_acceptance_window(oracle_ensemble* e, double w) { e->prop.acceptance_window = w; }
void oracle_ensemble_set_acceptance_deweight(oracle_ensemble* e, double d) { e->prop.acceptance_deweight = d; }
void oracle_ensemble_set_acceptance_rigidity(oracle_ensemble* e, double r) {
    e->prop.rigidity = r;
    for (int c = 0; c < e->nchains; ++c) e->rigidity[c] = r;
}
void oracle_ensemble_set_target_acceptance(oracle_ensemble* e, double a) { e->prop.target = a; }
void oracle_ensemble_set_step_rms_window(oracle_ensemble* e, int n) { e->step_rms_window = n; }
void oracle_ensemble_set_sigma(oracle_ensemble* e, double s) {
    e->prop.sigma = s;
    for (int c = 0; c < e->nchains; ++c) e->sigma[c] = s;
}

static double ens_like(const oracle_ensemble* e, const double* p) {
    return oracle_like_order(e->like_kind, e->dim, p, e->like_params, e->exact, e->quadform_rowwise);
}

/* x0: [d][chain] when broadcast == 0, [d] when broadcast != 0.  Returns 0 if
 * any chain's start is rejected (TSimpleMCMC.H:265-268). */
int oracle_ensemble_start(oracle_ensemble* e, const double* x0, int broadcast) {
    const int N = e->nchains, D = e->dim;
    double* p = (double*)malloc(sizeof(double) * (size_t)D);
    int ok = 1;
    for (int c = 0; c < N; ++c) {
        for (int d = 0; d < D; ++d) {
            p[d] = broadcast ? x0[d] : x0[(size_t)d * (size_t)N + (size_t)c];
            e->x[(size_t)d * (size_t)N + (size_t)c] = p[d];
        }
        e->logl[c] = ens_like(e, p);
        if (!isfinite(e->logl[c]) || e->logl[c] < -0.999999E+10) ok = 0;
    }
    if (!ok) { free(p); return 0; }
    /* shared InitializeState on chain 0's start (centre := chain 0's start) */
    for (int d = 0; d < D; ++d) p[d] = e->x[(size_t)d * (size_t)N];
    double user_sigma = e->prop.sigma;
    oracle_proposal_initialize(&e->prop, D, p, e->logl[0]);
    (void)user_sigma;
    for (int c = 0; c < N; ++c) {
        e->sigma[c] = e->prop.sigma;
        e->acceptance[c] = e->prop.acceptance;
        e->acceptance_trials[c] = e->prop.acceptance_trials;
        e->rigidity[c] = e->prop.rigidity;
        e->trials[c] = 0; e->successes[c] = 0;
        e->next_update[c] = e->prop.next_update;
        e->last_value[c] = e->logl[c];
        e->last_logl_proposed[c] = e->logl[c];   /* fProposedLogLikelihood = L(start), TSimpleMCMC.H:258 */
        e->last_x0[c] = e->x[c];
        e->step_rms[c] = 0.0; e->step_rms_trials[c] = 0; e->naccept[c] = 0;
    }
    memcpy(e->c0, e->prop.central, sizeof(double) * (size_t)D);
    memset(e->acc, 0, sizeof(double) * (size_t)e->ngroups * (size_t)ens_npacked(D));
    free(p);
    return 1;
}

/* one ensemble step: every chain makes one TSimpleMCMC::Step() */
static void ens_step_once(oracle_ensemble* e, int metropolis) {
    const int N = e->nchains, D = e->dim;
    const oracle_proposal* P = &e->prop;
    const int npk = ens_npacked(D);
    double* x = (double*)malloc(sizeof(double) * (size_t)D);
    double* xp = (double*)malloc(sizeof(double) * (size_t)D);
    double* y = (double*)malloc(sizeof(double) * (size_t)(D + 1));
    ++e->total_steps;
    const double max_up = (double)D * (double)D;
    for (int c = 0; c < N; ++c) {
        for (int d = 0; d < D; ++d) x[d] = e->x[(size_t)d * (size_t)N + (size_t)c];
        /* --- UpdateState, scalar half (TSimpleMCMC.H:1723-1776) --- */
        int moved = (e->logl[c] != e->last_value[c] || x[0] != e->last_x0[c]);
        int accepted = oracle_update_scalars(&e->trials[c], &e->successes[c], &e->acceptance[c],
                                             &e->acceptance_trials[c], P->acceptance_window,
                                             &e->rigidity[c], P->target, &e->sigma[c], moved);
        /* --- pooled moments of the current point --- */
        if (e->mode == ENS_MODE_POOLED && ((e->total_steps - 1) % e->moment_stride) != 0) {
            /* not a fold step */
        } else if (e->mode == ENS_MODE_POOLED) {
            double* acc = e->acc + (size_t)(c / e->moment_group) * (size_t)npk;
            for (int d = 0; d < D; ++d) y[d] = x[d] - e->c0[d];
            y[D] = 1.0;
            for (int i = 0; i <= D; ++i)
                for (int j = 0; j <= i; ++j) {
                    int idx = i * (i + 1) / 2 + j;
                    acc[idx] = SMCMC_FMA(y[i], y[j], acc[idx]);
                }
        } else if (accepted && (--e->next_update[c]) < 1) {
            /* per-chain UpdateProposal with a frozen covariance (TSimpleMCMC.H:1824-1826,
             * 1042-1086): the trace is unchanged so sigma*sqrt(1) == sigma and the
             * decomposition is the same; what changes is the schedule and the
             * acceptance de-weighting. */
            double up = 0.5 * e->successes[c];
            e->next_update[c] = (int)(P->acceptance_window + max_up - max_up / (up + 1.0));
            if (P->acceptance_deweight > 0.0) {
                double w = 1.0 - fmin(P->acceptance_deweight, 1.0);
                e->acceptance_trials[c] = fmax(1.0, w * e->acceptance_trials[c]);
                e->acceptance_trials[c] = fmin(e->acceptance_trials[c], w * P->acceptance_window);
            }
        }
        e->last_value[c] = e->logl[c];
        e->last_x0[c] = x[0];

        /* --- proposal draw (TSimpleMCMC.H:709-724) --- */
        oracle_stream st; memset(&st, 0, sizeof(st));
        st.seed = e->seed; st.chain = e->chain_offset + (uint32_t)c;
        oracle_stream_set_step(&st, (uint64_t)(uint32_t)e->total_steps);
        memcpy(xp, x, sizeof(double) * (size_t)D);
        for (int i = 0; i < D; ++i) {
            if (P->ptype[i] == 1) {
                double u = oracle_stream_uniform(&st, (uint32_t)i);
                xp[i] = P->pparam1[i] + (P->pparam2[i] - P->pparam1[i]) * u;
                continue;
            }
            double r = oracle_stream_normal(&st, i);
            double s = e->sigma[c] * r;
            int j0 = P->decomp_full ? 0 : i;   /* entries below the diagonal of U are zero */
            for (int j = j0; j < D; ++j) {
                if (P->ptype[j] == 1) continue;
                if (e->exact) xp[j] += s * P->decomp[i * D + j];
                else xp[j] = SMCMC_FMA(s, P->decomp[i * D + j], xp[j]);
            }
        }
        /* --- StepRMS window (TSimpleMCMC.H:391-406) --- */
        if (e->step_rms_window > 0) {
            double sqr = 0.0;
            for (int i = 0; i < D; ++i) {
                double t = xp[i] - x[i];
                if (e->exact) sqr += t * t; else sqr = SMCMC_FMA(t, t, sqr);
            }
            double ms = e->step_rms[c] * e->step_rms[c];
            ms *= e->step_rms_trials[c];
            ms += sqr;
            ms /= e->step_rms_trials[c] + 1.0;
            e->step_rms_trials[c] = (e->step_rms_window < e->step_rms_trials[c] + 1)
                                        ? e->step_rms_window : e->step_rms_trials[c] + 1;
            e->step_rms[c] = sqrt(ms);
        }
        /* --- likelihood + Metropolis test (TSimpleMCMC.H:410-491) --- */
        double lp = ens_like(e, xp);
        e->last_logl_proposed[c] = lp;
        int take = 0;
        if (metropolis == 2) {
            take = 1;
        } else if (!isfinite(lp) || lp < -0.999999E+30) {
            take = 0;
        } else {
            double delta = lp - e->logl[c];
            take = 1;
            if (delta < 0.0) {
                if (metropolis == 1) take = 0;
                else {
                    double trial = smcmc_log(oracle_stream_uniform(&st, smcmc_accept_word((uint32_t)D)));
                    if (delta < trial) take = 0;
                }
            }
        }
        e->last_accept[c] = (uint8_t)take;
        if (take) {
            e->logl[c] = lp;
            for (int d = 0; d < D; ++d) e->x[(size_t)d * (size_t)N + (size_t)c] = xp[d];
            e->naccept[c]++;
        }
    }
    free(x); free(xp); free(y);
}

void oracle_ensemble_step(oracle_ensemble* e, int nsteps, int metropolis) {
    for (int s = 0; s < nsteps; ++s) ens_step_once(e, metropolis);
}

/* Sum the group accumulators into moments[(D+1)(D+2)/2] and clear them.  Two
 * ordered levels, as the HIP reduction does it: groups in ascending order within
 * chunks of 32 groups, then the chunk sums in ascending order. */
#define ENS_REDUCE_CHUNK 32
void oracle_ensemble_reduce_moments(oracle_ensemble* e, double* moments) {
    const int npk = ens_npacked(e->dim);
    for (int k = 0; k < npk; ++k) {
        double total = 0.0;
        for (int g0 = 0; g0 < e->ngroups; g0 += ENS_REDUCE_CHUNK) {
            double s = 0.0;
            for (int g = g0; g < e->ngroups && g < g0 + ENS_REDUCE_CHUNK; ++g)
                s += e->acc[(size_t)g * (size_t)npk + (size_t)k];
            total += s;
        }
        moments[k] = total;
    }
    memset(e->acc, 0, sizeof(double) * (size_t)e->ngroups * (size_t)npk);
}

static void ens_adjust_lanes(oracle_ensemble* e, double scale);
static void ens_after_update(oracle_ensemble* e);

/* Pooled update from (all-reduced) moments about c0: the reference running
 * averages (TSimpleMCMC.H:1780-1820) fed with a batch of n points, followed by
 * UpdateProposal (TSimpleMCMC.H:1009-1390) on the shared proposal. */
void oracle_ensemble_apply_moments(oracle_ensemble* e, const double* M) {
    oracle_proposal* P = &e->prop;
    const int D = e->dim;
    const double n = M[D * (D + 1) / 2 + D];
    if (!(n > 0.0)) return;
    const double* S1 = M + D * (D + 1) / 2;      /* row D: sum of y_j */
    double* delta = (double*)malloc(sizeof(double) * (size_t)D);
    for (int d = 0; d < D; ++d) {
        delta[d] = S1[d] / (P->central_trials + n);
        P->central_change[d] = delta[d];
        P->central[d] = e->c0[d] + delta[d];
    }
    P->central_trials = fmin(P->cov_window, P->central_trials + n);
    if (e->mode == ENS_MODE_POOLED) {
        for (int i = 0; i < D; ++i) {
            for (int j = 0; j <= i; ++j) {
                double b = M[i * (i + 1) / 2 + j];
                b -= S1[i] * delta[j];
                b -= delta[i] * S1[j];
                b += (n * delta[i]) * delta[j];
                double v = P->cov[i * D + j];
                v *= P->cov_trials;
                v += b;
                v /= P->cov_trials + n;
                if (i == j) P->cov[i * D + j] = v;
                else P->cov[i * D + j] = P->cov[j * D + i] = v;
            }
        }
        P->cov_trials = fmin(P->cov_window, P->cov_trials + n);
    }
    free(delta);
    int succ = 0;
    for (int c = 0; c < e->nchains; ++c) succ += e->successes[c];
    P->successes = succ;
    oracle_proposal_update(P, 0);
    e->last_sigma_scale = P->last_sigma_scale;    /* = sqrt(old_trace / new_trace) */
    ens_after_update(e);
}

/* smcmc_set_covariance: overwrite fCurrentCov of the shared proposal (takes effect at the next update) */
void oracle_ensemble_set_covariance(oracle_ensemble* e, const double* cov) {
    memcpy(e->prop.cov, cov, sizeof(double) * (size_t)e->dim * (size_t)e->dim);
}

/* what an UpdateProposal on the shared proposal does to every chain's scalars: sigma rescale
 * (TSimpleMCMC.H:1042-1043) and acceptance de-weighting (:1081-1086) */
static void ens_adjust_lanes(oracle_ensemble* e, double scale) {
    const oracle_proposal* P = &e->prop;
    for (int c = 0; c < e->nchains; ++c) {
        e->sigma[c] = e->sigma[c] * scale;
        if (P->acceptance_deweight > 0.0) {
            double w = 1.0 - fmin(P->acceptance_deweight, 1.0);
            e->acceptance_trials[c] = fmax(1.0, w * e->acceptance_trials[c]);
            e->acceptance_trials[c] = fmin(e->acceptance_trials[c], w * P->acceptance_window);
        }
    }
}

static void ens_reset_lanes(oracle_ensemble* e) {
    const oracle_proposal* P = &e->prop;
    const int D = e->dim;
    for (int c = 0; c < e->nchains; ++c) {
        e->trials[c] = 0; e->successes[c] = 0;
        e->next_update[c] = P->next_update;
        e->acceptance[c] = P->acceptance;
        e->acceptance_trials[c] = P->acceptance_trials;
        if (e->sigma[c] < 0.01 * sqrt(1.0 / D)) e->sigma[c] = sqrt(1.0 / D);
    }
    memset(e->acc, 0, sizeof(double) * (size_t)e->ngroups * (size_t)ens_npacked(D));
}

/* The consequences of an UpdateProposal on the shared proposal for the chains (the engine's update_shared):
 * sigma rescale and de-weighting per chain; when the ladder ended in ResetProposal (TSimpleMCMC.H:1389) the
 * chains are reset with it, the shared centre restarting from chain 0's current point. */
static void ens_after_update(oracle_ensemble* e) {
    oracle_proposal* P = &e->prop;
    const int N = e->nchains, D = e->dim;
    ens_adjust_lanes(e, e->last_sigma_scale);
    if (P->last_update_path == 4) {
        for (int d = 0; d < D; ++d) P->last_point[d] = P->central[d] = e->x[(size_t)d * (size_t)N];
        ens_reset_lanes(e);
    }
    memcpy(e->c0, P->central, sizeof(double) * (size_t)D);
}

/* smcmc_update_proposal: the user's UpdateProposal() (TSimpleMCMC.H:1009) on the shared proposal.  FROZEN: every
 * chain reschedules its own next update from its own successes (:1050-1052). */
void oracle_ensemble_update_proposal(oracle_ensemble* e) {
    oracle_proposal* P = &e->prop;
    oracle_proposal_update(P, 0);
    e->last_sigma_scale = P->last_sigma_scale;
    if (P->failed) return;
    ens_after_update(e);
    if (e->mode == ENS_MODE_FROZEN && P->last_update_path != 4) {
        const double max_up = (double)e->dim * (double)e->dim;
        for (int c = 0; c < e->nchains; ++c)
            e->next_update[c] = (int)(P->acceptance_window + max_up - max_up / (0.5 * e->successes[c] + 1.0));
    }
}

/* smcmc_reset_proposal: ResetProposal() (TSimpleMCMC.H:1396-1494) on the shared proposal, whose fLastPoint is
 * chain 0's current point; every chain's counters, acceptance history and (floored) sigma follow :1405-1410, 1481-1482. */
void oracle_ensemble_reset_proposal(oracle_ensemble* e) {
    oracle_proposal* P = &e->prop;
    const int N = e->nchains, D = e->dim;
    for (int d = 0; d < D; ++d) P->last_point[d] = e->x[(size_t)d * (size_t)N];
    oracle_proposal_reset(P);
    if (P->failed) return;
    ens_reset_lanes(e);
    memcpy(e->c0, P->central, sizeof(double) * (size_t)D);
}

void oracle_ensemble_sync(oracle_ensemble* e) {
    double* M = (double*)malloc(sizeof(double) * (size_t)ens_npacked(e->dim));
    oracle_ensemble_reduce_moments(e, M);
    oracle_ensemble_apply_moments(e, M);
    free(M);
}

/* getters */
void oracle_ensemble_get_x(const oracle_ensemble* e, double* out) { memcpy(out, e->x, sizeof(double) * (size_t)e->nchains * (size_t)e->dim); }
void oracle_ensemble_get_lane_f64(const oracle_ensemble* e, int field, double* out) {
    const double* src = NULL;
    switch (field) {
        case 0: src = e->logl; break;
        case 1: src = e->sigma; break;
        case 2: src = e->acceptance; break;
        case 3: src = e->acceptance_trials; break;
        case 4: src = e->rigidity; break;
        case 5: src = e->step_rms; break;
        case 6: src = e->last_logl_proposed; break;
        default: return;
    }
    memcpy(out, src, sizeof(double) * (size_t)e->nchains);
}
void oracle_ensemble_get_lane_i32(const oracle_ensemble* e, int field, int32_t* out) {
    const int32_t* src = NULL;
    switch (field) {
        case 0: src = e->trials; break;
        case 1: src = e->successes; break;
        case 2: src = e->next_update; break;
        case 3: src = e->naccept; break;
        case 4: src = e->step_rms_trials; break;
        default: return;
    }
    memcpy(out, src, sizeof(int32_t) * (size_t)e->nchains);
}
void oracle_ensemble_get_last_accept(const oracle_ensemble* e, uint8_t* out) { memcpy(out, e->last_accept, (size_t)e->nchains); }
void oracle_ensemble_get_center(const oracle_ensemble* e, double* out) { memcpy(out, e->prop.central, sizeof(double) * (size_t)e->dim); }
void oracle_ensemble_get_covariance(const oracle_ensemble* e, double* out) { memcpy(out, e->prop.cov, sizeof(double) * (size_t)e->dim * (size_t)e->dim); }
void oracle_ensemble_get_decomposition(const oracle_ensemble* e, double* out) { memcpy(out, e->prop.decomp, sizeof(double) * (size_t)e->dim * (size_t)e->dim); }
/* 0 sigma_trace 1 cov_trials 2 central_trials 3 cov_window 4 acceptance_window 5 target
 * 6 total_steps 7 update_count 8 last_update_path 9 failed 10 last_sigma_scale 11 decomp_full */
void oracle_ensemble_get_shared(const oracle_ensemble* e, double* out) {
    const oracle_proposal* p = &e->prop;
    out[0] = p->sigma_trace; out[1] = p->cov_trials; out[2] = p->central_trials; out[3] = p->cov_window;
    out[4] = p->acceptance_window; out[5] = p->target; out[6] = e->total_steps; out[7] = p->update_count;
    out[8] = p->last_update_path; out[9] = p->failed; out[10] = e->last_sigma_scale; out[11] = p->decomp_full;
}
