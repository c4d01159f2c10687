/* vaat_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of
 * sMCMC::TSimpleMCMC<L, sMCMC::TProposeVAATStep> (TProposeVAATStep.H:22-307 driving TSimpleMCMC.H:246-276, 370-496) for
 * N independent chains.  TProposeVAATStep keeps nothing that chains could share, so chain c of the ensemble IS the
 * reference chain whose random stream is (seed, chain_offset + c); gRandom is replaced by the counter-based stream of
 * include/smcmc_detmath.h (SMCMC_STREAM_VAAT: words 0,1 the step's Gaus, word 2 its Uniform(a,b), word 3 the
 * Metropolis Uniform(), words 4+i the i-th Uniform() of a shuffle made during the step).  PARITY UNPINNED, as
 * oracle_core.h explains: ROOT is absent, the reference cannot be built here, and it ships no expected output for
 * this proposal (SimpleVAAT.C writes a tree and prints acceptances).
 *
 * What is restated, per chain:
 *   operator()            TProposeVAATStep.H:40-80
 *   SetUniform/SetGaussian                   :101-133
 *   UpdateProposal (index queue + shuffle)   :177-195
 *   InitializeState                          :198-214
 *   UpdateState (per-index acceptance, sigma):219-255
 *   TSimpleMCMC::Start / Step around it      TSimpleMCMC.H:246-276, 370-496
 * The reference's quirks are kept: fAcceptanceWindow is an int that InitializeState resets to 100 (:211, 277);
 * SetGaussian's sigma is used as the Gaus() width unsquared (:69-78, 131); a dimension whose acceptance reaches exactly
 * zero drops to the 1E-4 floor (pow(0, y) = 0, :246-253).
 */
#include "oracle_core.h"

typedef struct {
    int nchains, dim;
    int like_kind;
    double* like_params;
    uint64_t seed;
    uint32_t chain_offset;
    int exact;                 /* 1: reference operation order, 0: fused multiply-add order */
    int total_steps;           /* fTotalSteps (the same for every chain) */
    /* settings, shared by the chains */
    int* ptype;                /* fProposalType[].type */
    double* pparam1;
    double* pparam2;
    int acceptance_window;     /* fAcceptanceWindow (int, :277) */
    double rigidity;           /* fAcceptanceRigidity */
    double target;             /* fTargetAcceptance = 0.44 (:30) */
    int step_rms_window;
    int initialized;           /* fStateInitialized */
    /* per chain; vectors are [d][chain] */
    double* x;
    double* logl;
    double* last_value;
    double* last_logl_proposed;
    double* proposed_value;    /* fProposed[fLastIndex] of the latest step */
    double* sigma;             /* fSigma[d] */
    double* acceptance;        /* fAcceptance[d] */
    int32_t* acc_trials;       /* fAcceptanceTrials[d] */
    int32_t* queue;            /* fNextIndex, [slot][chain] */
    int32_t* queue_len;
    int32_t* last_index;
    int32_t* trials;
    int32_t* successes;
    int32_t* naccept;
    uint8_t* last_accept;
    double* step_rms;
    int32_t* step_rms_trials;
} oracle_vaat;

oracle_vaat* oracle_vaat_create(int nchains, int dim, int like_kind, const double* like_params, int n_like_params,
                                uint64_t seed, uint32_t chain_offset, int exact) {
    oracle_vaat* e = (oracle_vaat*)calloc(1, sizeof(oracle_vaat));
    e->nchains = nchains; e->dim = dim; e->like_kind = like_kind; e->seed = seed; e->chain_offset = chain_offset;
    e->exact = exact;
    if (n_like_params > 0) {
        e->like_params = (double*)malloc(sizeof(double) * (size_t)n_like_params);
        memcpy(e->like_params, like_params, sizeof(double) * (size_t)n_like_params);
    }
    const size_t N = (size_t)nchains, D = (size_t)dim;
    e->ptype = (int*)calloc(D, sizeof(int));
    e->pparam1 = (double*)calloc(D, sizeof(double));
    e->pparam2 = (double*)calloc(D, sizeof(double));
    e->acceptance_window = -1;                                       /* :26 */
    e->rigidity = 2.0;                                               /* :27 */
    e->target = 0.44;                                                /* :30 */
    e->x = (double*)calloc(N * D, sizeof(double));
    e->logl = (double*)calloc(N, sizeof(double));
    e->last_value = (double*)calloc(N, sizeof(double));
    e->last_logl_proposed = (double*)calloc(N, sizeof(double));
    e->proposed_value = (double*)calloc(N, sizeof(double));
    e->sigma = (double*)calloc(N * D, sizeof(double));
    e->acceptance = (double*)calloc(N * D, sizeof(double));
    e->acc_trials = (int32_t*)calloc(N * D, sizeof(int32_t));
    e->queue = (int32_t*)calloc(N * D, sizeof(int32_t));
    e->queue_len = (int32_t*)calloc(N, sizeof(int32_t));
    e->last_index = (int32_t*)calloc(N, sizeof(int32_t));
    e->trials = (int32_t*)calloc(N, sizeof(int32_t));
    e->successes = (int32_t*)calloc(N, sizeof(int32_t));
    e->naccept = (int32_t*)calloc(N, sizeof(int32_t));
    e->last_accept = (uint8_t*)calloc(N, 1);
    e->step_rms = (double*)calloc(N, sizeof(double));
    e->step_rms_trials = (int32_t*)calloc(N, sizeof(int32_t));
    for (size_t k = 0; k < N * D; ++k) e->sigma[k] = 2.34;           /* SetDim :98 */
    for (size_t c = 0; c < N; ++c) e->last_index[c] = -1;            /* :27 */
    return e;
}

void oracle_vaat_destroy(oracle_vaat* e) {
    if (!e) return;
    free(e->like_params); free(e->ptype); free(e->pparam1); free(e->pparam2); free(e->x); free(e->logl);
    free(e->last_value); free(e->last_logl_proposed); free(e->proposed_value); free(e->sigma); free(e->acceptance); free(e->acc_trials);
    free(e->queue); free(e->queue_len); free(e->last_index); free(e->trials); free(e->successes); free(e->naccept);
    free(e->last_accept); free(e->step_rms); free(e->step_rms_trials);
    free(e);
}

void oracle_vaat_set_uniform(oracle_vaat* e, int d, double lo, double hi) {             /* :101-117 */
    if (d < 0 || d >= e->dim) return;
    e->ptype[d] = 1; e->pparam1[d] = lo; e->pparam2[d] = hi;
}
void oracle_vaat_set_gaussian(oracle_vaat* e, int d, double sigma) {                    /* :123-133 */
    if (d < 0 || d >= e->dim) return;
    e->ptype[d] = 0; e->pparam1[d] = sigma;
}
void oracle_vaat_set_acceptance_window(oracle_vaat* e, double a) { e->acceptance_window = (int)a; }   /* :137, int member */
void oracle_vaat_set_acceptance_rigidity(oracle_vaat* e, double r) { e->rigidity = r; }               /* :148 */
void oracle_vaat_set_step_rms_window(oracle_vaat* e, int n) { e->step_rms_window = n; }

static double vaat_like(const oracle_vaat* e, const double* p) {
    return oracle_like_order(e->like_kind, e->dim, p, e->like_params, e->exact, 0);
}

static oracle_stream vaat_stream(const oracle_vaat* e, int c, uint64_t step) {
    oracle_stream st; memset(&st, 0, sizeof(st));
    st.seed = e->seed; st.chain = e->chain_offset + (uint32_t)c; st.stream_id = SMCMC_STREAM_VAAT;
    oracle_stream_set_step(&st, step);
    return st;
}

/* UpdateProposal :177-195 for chain c, with the Uniform() draws of step `step` */
static void vaat_update_proposal(oracle_vaat* e, int c, uint64_t step) {
    const int N = e->nchains, D = e->dim;
    if (e->queue_len[c] != 0) return;                                 /* :178 */
    oracle_stream st = vaat_stream(e, c, step);
    e->last_index[c] = -1;                                            /* :183 */
    for (int i = 0; i < D; ++i) e->queue[(size_t)i * N + c] = i;      /* :184-186 */
    e->queue_len[c] = D;
    for (int i = 0; i < D; ++i) {                                     /* :190-193 */
        size_t s = (size_t)((double)D * oracle_stream_uniform(&st, 4u + (uint32_t)i));
        int32_t t = e->queue[(size_t)i * N + c];
        e->queue[(size_t)i * N + c] = e->queue[s * N + c];
        e->queue[s * N + c] = t;
    }
}

/* Start TSimpleMCMC.H:246-276 with InitializeState TProposeVAATStep.H:198-214.  x0: [d][chain], or [d] broadcast.
 * Returns 0 if a start point is rejected (TSimpleMCMC.H:265-268). */
int oracle_vaat_start(oracle_vaat* e, const double* x0, int broadcast) {
    const int N = e->nchains, D = e->dim;
    double* p = (double*)malloc(sizeof(double) * (size_t)D);
    int ok = 1;
    for (int c = 0; c < N; ++c) {
        for (int d = 0; d < D; ++d) {
            p[d] = broadcast ? x0[d] : x0[(size_t)d * N + c];
            e->x[(size_t)d * N + c] = p[d];
        }
        e->logl[c] = vaat_like(e, p);
        if (!isfinite(e->logl[c]) || e->logl[c] < -0.999999E+10) ok = 0;
        if (!e->initialized) e->last_value[c] = e->logl[c];           /* :207, behind the early return of :197 */
        e->last_logl_proposed[c] = e->logl[c];
    }
    free(p);
    if (!ok) return 0;
    if (!e->initialized) {
        e->initialized = 1;
        e->acceptance_window = 100;                                   /* :211 */
    }
    return 1;
}

/* the explicit GetProposeStep().UpdateProposal() of SimpleVAAT.C:44 (draws keyed on the current step count) */
void oracle_vaat_update_proposal(oracle_vaat* e) {
    for (int c = 0; c < e->nchains; ++c) vaat_update_proposal(e, c, (uint64_t)(uint32_t)e->total_steps);
}

static void vaat_step_once(oracle_vaat* e) {
    const int N = e->nchains, D = e->dim;
    double* x = (double*)malloc(sizeof(double) * (size_t)D);
    double* xp = (double*)malloc(sizeof(double) * (size_t)D);
    ++e->total_steps;                                                 /* TSimpleMCMC.H:376 */
    const uint64_t step = (uint64_t)(uint32_t)e->total_steps;
    for (int c = 0; c < N; ++c) {
        for (int d = 0; d < D; ++d) x[d] = e->x[(size_t)d * N + c];
        const double value = e->logl[c];
        /* ---- UpdateState :219-255 ---- */
        ++e->trials[c];
        const int accepted = (value != e->last_value[c]);             /* :225-226 */
        if (accepted) ++e->successes[c];
        e->last_value[c] = value;
        const int li = e->last_index[c];
        if (li >= 0) {                                                /* :235 */
            int32_t* at = &e->acc_trials[(size_t)li * N + c];
            double* acc = &e->acceptance[(size_t)li * N + c];
            double* sg = &e->sigma[(size_t)li * N + c];
            ++(*at);                                                  /* :238 */
            const int m = (e->acceptance_window < *at) ? e->acceptance_window : *at;
            *acc *= 1.0 * m;                                          /* :239-240 */
            if (accepted) *acc += 1.0;
            *acc /= 1.0 + 1.0 * m;                                    /* :242-243 */
            if (*at > 0.1 * e->acceptance_window && e->rigidity > 0 && e->rigidity < 100.0) {   /* :245-247 */
                double v = *sg;
                const double ratio = *acc / e->target;
                const double expo = fmin(1.0 / 500.0, 1.0 / (e->rigidity * e->acceptance_window));
                v *= (ratio > 0.0) ? smcmc_pow_small(ratio, expo) : 0.0;   /* pow(0, y > 0) = 0 */
                *sg = fmax(v, 1.0E-4);                                /* :253 */
            }
        }
        /* ---- operator() :52-78 ---- */
        memcpy(xp, x, sizeof(double) * (size_t)D);
        vaat_update_proposal(e, c, step);                             /* :55 */
        const int idx = e->queue[(size_t)(e->queue_len[c] - 1) * N + c];   /* :58-59 */
        --e->queue_len[c];
        e->last_index[c] = idx;
        oracle_stream st = vaat_stream(e, c, step);
        if (e->ptype[idx] == 1) {                                     /* :60-66 */
            const double u = oracle_stream_uniform(&st, 2u);
            xp[idx] = e->pparam1[idx] + (e->pparam2[idx] - e->pparam1[idx]) * u;
        } else {
            double width = 1.0;                                       /* "expectedVariance", handed to Gaus() as its sigma */
            if (e->ptype[idx] == 0 && e->pparam1[idx] > 0) width = e->pparam1[idx];
            const double g = 0.0 + width * oracle_stream_normal(&st, 0);   /* gRandom->Gaus(0.0, width) */
            const double sg = e->sigma[(size_t)idx * N + c];
            if (e->exact) xp[idx] = x[idx] + sg * g;                  /* :76-77 */
            else xp[idx] = SMCMC_FMA(sg, g, x[idx]);
        }
        /* ---- TSimpleMCMC::Step around it (TSimpleMCMC.H:391-491) ---- */
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
            e->step_rms_trials[c] = (e->step_rms_window < e->step_rms_trials[c] + 1) ? e->step_rms_window
                                                                                      : e->step_rms_trials[c] + 1;
            e->step_rms[c] = sqrt(ms);
        }
        e->proposed_value[c] = xp[idx];
        const double lp = vaat_like(e, xp);                           /* :410 */
        e->last_logl_proposed[c] = lp;
        int take = 1;
        if (!isfinite(lp) || lp < -0.999999E+30) {                    /* :432-436 */
            take = 0;
        } else {
            const double delta = lp - value;                          /* :441 */
            if (delta < 0.0) {
                const double trial = smcmc_log(oracle_stream_uniform(&st, 3u));   /* :455 */
                if (delta < trial) take = 0;
            }
        }
        e->last_accept[c] = (uint8_t)take;
        if (take) {                                                   /* :484-487 */
            e->logl[c] = lp;
            e->x[(size_t)idx * N + c] = xp[idx];
            e->naccept[c]++;
        }
    }
    free(x); free(xp);
}

void oracle_vaat_step(oracle_vaat* e, int nsteps) {
    for (int s = 0; s < nsteps; ++s) vaat_step_once(e);
}

void oracle_vaat_get_x(const oracle_vaat* e, double* out) { memcpy(out, e->x, sizeof(double) * (size_t)e->nchains * (size_t)e->dim); }
/* field: 0 logl, 1 logl_proposed, 2 step_rms, 3 proposed_value */
void oracle_vaat_get_lane_f64(const oracle_vaat* e, int field, double* out) {
    const double* src = field == 0 ? e->logl : field == 1 ? e->last_logl_proposed : field == 2 ? e->step_rms : e->proposed_value;
    memcpy(out, src, sizeof(double) * (size_t)e->nchains);
}
/* field: 0 trials, 1 successes, 2 last_index, 3 queue_len, 4 naccept, 5 last_accept, 6 step_rms_trials */
void oracle_vaat_get_lane_i32(const oracle_vaat* e, int field, int32_t* out) {
    for (int c = 0; c < e->nchains; ++c) {
        switch (field) {
            case 0: out[c] = e->trials[c]; break;
            case 1: out[c] = e->successes[c]; break;
            case 2: out[c] = e->last_index[c]; break;
            case 3: out[c] = e->queue_len[c]; break;
            case 4: out[c] = e->naccept[c]; break;
            case 5: out[c] = e->last_accept[c]; break;
            default: out[c] = e->step_rms_trials[c]; break;
        }
    }
}
/* [d][chain]: field 0 fSigma, 1 fAcceptance */
void oracle_vaat_get_dim_f64(const oracle_vaat* e, int field, double* out) {
    memcpy(out, field == 0 ? e->sigma : e->acceptance, sizeof(double) * (size_t)e->nchains * (size_t)e->dim);
}
/* [d][chain]: field 0 fAcceptanceTrials, 1 fNextIndex (slots >= queue_len are stale) */
void oracle_vaat_get_dim_i32(const oracle_vaat* e, int field, int32_t* out) {
    memcpy(out, field == 0 ? e->acc_trials : e->queue, sizeof(int32_t) * (size_t)e->nchains * (size_t)e->dim);
}
int oracle_vaat_get_acceptance_window(const oracle_vaat* e) { return e->acceptance_window; }
