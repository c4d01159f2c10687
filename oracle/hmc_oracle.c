/* hmc_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of sMCMC::TSimpleHMC
 * (/root/reference/TSimpleHMC.H:119-973) for one chain, with the analytic
 * gradients of the shipped likelihoods, and of the many-chain HMC engine the HIP
 * path implements.  PARITY UNPINNED, as for the rest of oracle/ (no golden vectors
 * exist, ROOT is absent).
 *
 * Ensemble semantics (the HIP engine's): every chain keeps its own position,
 * momentum, fMeanEpsilon, fLeapFrogSteps, fReversalLen, acceptance (reference :302-323, 342-344,
 * 367, 386 verbatim per chain).  What the reference derives from the chain's running covariance
 * (UpdateCovariance :665-695, UpdateErrorMatrix :703-858) is POOLED over the ensemble exactly as
 * the Metropolis engine pools its covariance: every step each chain whose leapfrog ended with a
 * finite potential (:336) folds its accepted point x (as it stood before the accept decision,
 * :338) into its moment group's accumulator acc[i][j] = fma(y_i, y_j, acc[i][j]), y = (x, 1),
 * chains of the group in ascending order; a sync sums the groups in order, feeds the running
 * averages of :671-693 with the batch (n points: v = (v T + sum) / (T + n)) and runs
 * UpdateErrorMatrix once; when it fires, every chain takes the new step length and leapfrog
 * count by the formulas of :833-847.  One chain, one group, a sync per step IS the reference chain.
 *
 * Draw slots of one HMC step (stream SMCMC_STREAM_HMC of include/smcmc_detmath.h):
 *   momentum normal i (TSimpleHMC.H:568)  : Box-Muller pair i/2, words (2p, 2p+1)
 *   epsilon uniform (TSimpleHMC.H:297)    : word 2*ceil(D/2)
 *   accept uniform (TSimpleHMC.H:347)     : word 2*ceil(D/2) + 1
 */
#include "oracle_core.h"

/* ---- gradients of log L (the `bool operator()(Vector& g, const Vector& p)` of
 * TSimpleHMC.H:85-89); returns 1 like the reference functors ------------- */

/* the README-form iso-Gaussian has no gradient in the reference; d/dp (-p^2/2) = -p */
static int hmc_grad_iso(int n, double* g, const double* p) {
    for (int i = 0; i < n; ++i) g[i] = -p[i];
    return 1;
}
/* TDummyLogLikelihood.H:34-42 */
static int hmc_grad_quadform(int n, double* g, const double* p, const double* E) {
    for (int i = 0; i < n; ++i) {
        g[i] = 0.0;
        for (int j = 0; j < n; ++j) g[i] -= E[i * n + j] * p[j];
    }
    return 1;
}
/* the same sum, same order, one fused multiply-add per term: the engine's fused order
 * (smcmc_hmc_set_exact_arithmetic(h, 0)), what a chain of v_mfma_f64_16x16x4_f64 computes */
static int hmc_grad_quadform_fused(int n, double* g, const double* p, const double* E) {
    for (int i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < n; ++j) s = SMCMC_FMA(E[i * n + j], p[j], s);
        g[i] = -s;
    }
    return 1;
}
/* THardLogLikelihood.H:70-91 */
static int hmc_grad_rosenbrock(int n, double* g, const double* p, double rb) {
    g[0] = -2.0 * (1.0 - p[0]) - 4.0 * rb * p[0] * (p[1] - p[0] * p[0]);
    for (int i = 1; i < n - 1; ++i) {
        g[i] = 2.0 * rb * (p[i] - p[i - 1] * p[i - 1]);
        g[i] += -2.0 * (1.0 - p[i]);
        g[i] += -4.0 * rb * p[i] * (p[i + 1] - p[i] * p[i]);
    }
    int i = n - 1;
    g[i] = +2.0 * rb * (p[i] - p[i - 1] * p[i - 1]);
    for (int k = 0; k < n; ++k) g[k] = -g[k];
    return 1;
}

typedef struct {
    int dim, like_kind;
    double* like_params;
    uint64_t seed;
    uint32_t chain;
    /* TSimpleHMC members (TSimpleHMC.H:864-970) */
    int step_count, potential_count, gradient_count;
    int leapfrog_steps;            /* fLeapFrogSteps; SetLeapFrog(n) stores -n (:190) */
    double alpha, mean_epsilon, reversal_len, target_acceptance, current_acceptance;
    double* accepted; double* accepted_momentum; double* proposed; double* proposed_momentum;
    double accepted_potential, proposed_potential;
    double* central; double central_potential;
    int last_accept;
    int potential_from_gradient;   /* 1: the HIP engine's association of the quadratic-form potential */
    int fused_gradient;            /* 1: quadratic-form gradient with fused multiply-adds */
    int gradient_type;             /* Step(save, gradientType): 0 user gradient, 2 covariant, 3 finite differences, 5 zero */
    struct hmc_shared_s* shared;   /* the covariance-derived state (own, or the ensemble's) */
    int owns_shared;
    double* pre_step;              /* fAccepted as UpdateCovariance sees it (:338) */
    int contributes;               /* okLeap && isfinite(fProposedPotential) of the latest step (:336) */
} oracle_hmc;

/* What TSimpleHMC derives from the running covariance (members :927-970). */
typedef struct hmc_shared_s {
    int dim;
    double cov_window;             /* fCovarianceWindow = 1000000 (:134) */
    double* average;               /* fAveragePoint */
    double average_trials;
    double* exxt;                  /* fEXXT */
    double* cov;                   /* fEstimatedCovariance */
    double cov_trials;
    double* error;                 /* fEstimatedError */
    double est_trace, cur_trace, orbit_length;
    int steps_remaining, steps_since_update;
    int step_count;                /* fStepCount of the ensemble */
    int leapfrog_zero;             /* SetLeapFrog(0): UpdateErrorMatrix returns at once (:704) */
    int update_count;              /* how often UpdateErrorMatrix went through */
    int fired;                     /* the latest sync went through */
    double max_scale, min_scale;   /* of the latest update */
} hmc_shared;

static hmc_shared* hmc_shared_create(int dim) {
    hmc_shared* s = (hmc_shared*)calloc(1, sizeof(hmc_shared));
    size_t n = (size_t)dim;
    s->dim = dim;
    s->cov_window = 1000000;
    s->average = (double*)calloc(n, sizeof(double));
    s->exxt = (double*)calloc(n * n, sizeof(double));
    s->cov = (double*)calloc(n * n, sizeof(double));
    s->error = (double*)calloc(n * n, sizeof(double));
    return s;
}
static void hmc_shared_destroy(hmc_shared* s) {
    if (!s) return;
    free(s->average); free(s->exxt); free(s->cov); free(s->error); free(s);
}
/* the covariance part of Start :236-266 */
static void hmc_shared_start(hmc_shared* s, const double* start) {
    const int n = s->dim;
    memcpy(s->average, start, sizeof(double) * (size_t)n);
    s->average_trials = 0.0;
    s->cov_trials = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            s->cov[i * n + j] = (i == j) ? 1.0 : 0.0;
            s->exxt[i * n + j] = 0.0;
        }
    oracle_invert(n, s->cov, s->error);
    s->est_trace = n;
    s->cur_trace = 0.0;
    s->orbit_length = 0.0;
    s->steps_remaining = 0;
    s->steps_since_update = 0;
    s->step_count = 0;
    s->update_count = 0;
    s->fired = 0;
}

/* UpdateCovariance :665-695 fed with a batch: M is the packed moment vector (row i <= D, column j <= i,
 * row D = {sum x_j, n}) of n accepted points; `steps` ensemble steps went into it. */
static void hmc_shared_absorb(hmc_shared* s, const double* M, int steps) {
    const int D = s->dim;
    const double* S1 = M + (size_t)D * (D + 1) / 2;
    const double n = S1[D];
    s->steps_since_update += steps;                                          /* :667-668 */
    s->steps_remaining -= steps;
    for (int i = 0; i < D; ++i) {                                            /* :671-677 */
        double v = s->average[i];
        v *= s->average_trials;
        v += S1[i];
        v /= s->average_trials + n;
        s->average[i] = v;
    }
    s->average_trials = fmin(s->cov_window, s->average_trials + n);         /* :678-679 */
    for (int i = 0; i < D; ++i) {                                            /* :681-691 */
        for (int j = 0; j < i + 1; ++j) {
            double v = s->exxt[i * D + j];
            v *= s->cov_trials;
            v += M[(size_t)i * (i + 1) / 2 + j];
            v /= s->cov_trials + n;
            s->exxt[i * D + j] = s->exxt[j * D + i] = v;
            s->cov[i * D + j] = s->cov[j * D + i] = s->exxt[i * D + j] - s->average[i] * s->average[j];
        }
    }
    s->cov_trials = fmin(s->cov_window, s->cov_trials + n);                 /* :692-693 */
}

/* UpdateErrorMatrix :703-858 without the central-point bookkeeping of :733-744 (outputs only; the
 * engine leaves it to its host mirror).  Returns 1 when the update went through. */
static int hmc_shared_update_error_matrix(hmc_shared* s) {
    const int D = s->dim;
    s->fired = 0;
    if (s->leapfrog_zero) return 0;                                          /* :704 */
    if (s->cov_trials < 2 * D) return 0;                                     /* :705 */
    s->cur_trace = 0.0;                                                      /* :708-711 */
    for (int i = 0; i < D; ++i) s->cur_trace += fabs(s->cov[i * D + i]);
    double change = fabs(s->cur_trace - s->est_trace);
    int do_it = 0;                                                           /* :715-719 */
    if (s->steps_remaining < 0) do_it = 1;
    if (s->steps_since_update > 2.0 * D && change > 0.01 * s->est_trace) do_it = 1;
    if (!do_it) return 0;
    s->steps_remaining = 2 * D + s->step_count;                              /* :760 */
    s->steps_since_update = 0;
    double* eig = (double*)malloc(sizeof(double) * (size_t)D);
    double max_scale = 0.0, min_scale = 1E+20;                               /* :764-765 */
    for (;;) {                                                               /* :766-809 */
        oracle_sym_eigenvalues(D, s->cov, eig);
        int positive = 1;
        for (int i = 0; i < D; ++i) {
            double e = eig[i];
            if (max_scale < fabs(e)) max_scale = fabs(e);
            if (min_scale > fabs(e)) min_scale = fabs(e);
            if (e < 0) positive = 0;
        }
        if (positive) break;
        for (int i = 0; i < D; ++i) {
            double r = s->est_trace * 1E-6;
            r /= D;
            r = fabs(r);
            if (s->cov[i * D + i] < r) s->cov[i * D + i] = r;
            for (int j = i + 1; j < D; ++j) {
                s->cov[i * D + j] = 0.0;
                s->cov[j * D + i] = s->cov[i * D + j];
            }
        }
    }
    free(eig);
    s->cur_trace = 0.0;                                                      /* :815-819 */
    for (int i = 0; i < D; ++i) s->cur_trace += fabs(s->cov[i * D + i]);
    s->est_trace = s->cur_trace;
    max_scale = sqrt(max_scale);                                             /* :822-827 */
    if (max_scale < 0.1) max_scale = 0.1;
    min_scale = sqrt(min_scale);
    if (min_scale < 0.01) min_scale = 0.01;
    s->orbit_length = 2.0 * 3.14 * max_scale;                                /* :830 */
    s->max_scale = max_scale;
    s->min_scale = min_scale;
    oracle_invert(D, s->cov, s->error);                                      /* :849-850 */
    s->update_count++;
    s->fired = 1;
    return 1;
}

/* What a chain takes from an update that went through (:833-847) */
static void hmc_chain_retune(double* mean_epsilon, int* leapfrog_steps, const hmc_shared* s) {
    if (*mean_epsilon > 0) {                                                 /* :835-839 */
        *mean_epsilon = 0.2 * s->max_scale;
        if (*mean_epsilon > 0.5 * s->min_scale) *mean_epsilon = 0.5 * s->min_scale;
        if (*mean_epsilon < 0.05 * s->max_scale) *mean_epsilon = 0.05 * s->max_scale;
    }
    if (*leapfrog_steps > 0) {                                               /* :841-848 */
        double target = 0.4 * s->orbit_length;
        *leapfrog_steps = (int)(target / fabs(*mean_epsilon));
        *leapfrog_steps = 2 * (*leapfrog_steps / 2 + 1);
        if (*leapfrog_steps > 3 * s->dim) *leapfrog_steps = 3 * s->dim;
        if (*mean_epsilon > 0) *mean_epsilon = target / *leapfrog_steps;
    }
}

static double hmc_potential(oracle_hmc* h, const double* p) {               /* :411-414 */
    ++h->potential_count;
    if (h->potential_from_gradient && h->like_kind == ORACLE_LIKE_QUADFORM) {
        /* the HIP engine's association for the quadratic form: -log L = 1/2 q^T (Error q),
         * folded over i from the gradient's rows (the reference's single D^2-term running
         * sum, TDummyLogLikelihood.H:24-28, is serial per chain) */
        const int n = h->dim;
        double* g = (double*)malloc(sizeof(double) * (size_t)n);
        if (h->fused_gradient) hmc_grad_quadform_fused(n, g, p, h->like_params);
        else hmc_grad_quadform(n, g, p, h->like_params);     /* g = -Error q */
        double usum = 0.0;
        for (int i = 0; i < n; ++i) usum += 0.5 * p[i] * (-g[i]);
        free(g);
        return usum;
    }
    return -oracle_like(h->like_kind, h->dim, p, h->like_params);
}

static double hmc_potential(oracle_hmc* h, const double* p);

/* PotentialGradient (:467-532).  Type 0 / 1 / 4 with a user gradient: grad = -gradLogL (:478-491);
 * 2: CovariantGradient (:447-454); 3: FiniteDifferenceGradient (:417-444); 5: zero (:524-528). */
static void hmc_potential_gradient(oracle_hmc* h, double* grad, const double* p) {
    ++h->gradient_count;
    const int n = h->dim;
    if (h->gradient_type == 2) {
        const hmc_shared* s = h->shared;
        for (int i = 0; i < n; ++i) {
            grad[i] = 0.0;
            for (int j = 0; j < n; ++j) grad[i] += s->error[i * n + j] * (p[j] - s->average[j]);
        }
        return;
    }
    if (h->gradient_type == 3) {
        double* work = (double*)malloc(sizeof(double) * (size_t)n);
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < n; ++j) work[j] = p[j];
            double du = 0.01;
            work[i] -= du;
            double u1 = hmc_potential(h, work);
            work[i] += 2.0 * du;
            double u2 = hmc_potential(h, work);
            grad[i] = 0.5 * (u2 - u1) / du;
        }
        free(work);
        return;
    }
    if (h->gradient_type == 5) {
        for (int i = 0; i < n; ++i) grad[i] = 0.0;
        return;
    }
    switch (h->like_kind) {
        case ORACLE_LIKE_ISO: hmc_grad_iso(h->dim, grad, p); break;
        case ORACLE_LIKE_QUADFORM:
            if (h->fused_gradient) hmc_grad_quadform_fused(h->dim, grad, p, h->like_params);
            else hmc_grad_quadform(h->dim, grad, p, h->like_params);
            break;
        default: hmc_grad_rosenbrock(h->dim, grad, p, h->like_params ? h->like_params[0] : 100.0); break;
    }
    for (int i = 0; i < h->dim; ++i) grad[i] = -grad[i];
}

static double hmc_kinetic(int n, const double* m) {                         /* :535-542 */
    double ke = 0.0;
    for (int i = 0; i < n; ++i) {
        double p = m[i];
        ke += p * p / 2.0;
    }
    return ke;
}

oracle_hmc* oracle_hmc_create(int dim, int like_kind, const double* like_params, int n_like_params,
                              uint64_t seed, uint32_t chain_id) {
    oracle_hmc* h = (oracle_hmc*)calloc(1, sizeof(oracle_hmc));
    h->dim = dim; h->like_kind = like_kind; h->seed = seed; h->chain = chain_id;
    if (n_like_params > 0) {
        h->like_params = (double*)malloc(sizeof(double) * (size_t)n_like_params);
        memcpy(h->like_params, like_params, sizeof(double) * (size_t)n_like_params);
    }
    h->leapfrog_steps = 10;                                                  /* :133 */
    h->alpha = 0.0;
    size_t n = (size_t)dim;
    h->accepted = (double*)calloc(n, sizeof(double));
    h->accepted_momentum = (double*)calloc(n, sizeof(double));
    h->proposed = (double*)calloc(n, sizeof(double));
    h->proposed_momentum = (double*)calloc(n, sizeof(double));
    h->central = (double*)calloc(n, sizeof(double));
    h->pre_step = (double*)calloc(n, sizeof(double));
    h->shared = hmc_shared_create(dim);
    h->owns_shared = 1;
    return h;
}

void oracle_hmc_destroy(oracle_hmc* h) {
    if (!h) return;
    free(h->like_params); free(h->accepted); free(h->accepted_momentum); free(h->proposed);
    free(h->proposed_momentum); free(h->central); free(h->pre_step);
    if (h->owns_shared) hmc_shared_destroy(h->shared);
    free(h);
}

void oracle_hmc_set_alpha(oracle_hmc* h, double a) { h->alpha = a; }                 /* :175 */
void oracle_hmc_set_mean_epsilon(oracle_hmc* h, double e) { h->mean_epsilon = e; }   /* :181 */
void oracle_hmc_set_leapfrog(oracle_hmc* h, int n) { h->leapfrog_steps = -n; }       /* :190 */
void oracle_hmc_set_potential_from_gradient(oracle_hmc* h, int f) { h->potential_from_gradient = f; }
void oracle_hmc_set_fused_gradient(oracle_hmc* h, int f) { h->fused_gradient = f; }
void oracle_hmc_set_gradient_type(oracle_hmc* h, int t) { h->gradient_type = t; }

/* Start :210-269 */
static void hmc_chain_start(oracle_hmc* h, const double* start) {
    const size_t n = (size_t)h->dim;
    h->step_count = 0;
    memcpy(h->accepted, start, sizeof(double) * n);
    h->accepted_potential = hmc_potential(h, start);                         /* SetPosition :202-205 */
    memcpy(h->proposed, h->accepted, sizeof(double) * n);
    h->proposed_potential = h->accepted_potential;
    h->mean_epsilon = 0.05;                                                  /* :229 */
    h->reversal_len = 0.0;
    h->target_acceptance = 0.65;                                             /* :234 */
    h->current_acceptance = h->target_acceptance;
    memcpy(h->central, h->accepted, sizeof(double) * n);
    h->central_potential = h->accepted_potential;
}
void oracle_hmc_start(oracle_hmc* h, const double* start) {
    hmc_chain_start(h, start);
    hmc_shared_start(h->shared, start);
}

/* LeapFrog :582-651; returns leapStatus */
static int hmc_leapfrog(oracle_hmc* h, double* qNew, double* pNew, const double* position, double epsilon,
                        int steps) {
    const int n = h->dim;
    memcpy(qNew, position, sizeof(double) * (size_t)n);
    double* momentum = (double*)malloc(sizeof(double) * (size_t)n);
    double* grad = (double*)malloc(sizeof(double) * (size_t)n);
    memcpy(momentum, pNew, sizeof(double) * (size_t)n);
    int leap_status = 1;
    if (steps < 1) {                                                          /* :598-611 */
        for (int j = 0; j < n; ++j) qNew[j] = qNew[j] + epsilon * (momentum[j] + pNew[j]) / 2.0;
        free(momentum); free(grad);
        return leap_status;
    }
    hmc_potential_gradient(h, grad, qNew);                                    /* :615 */
    for (int j = 0; j < n; ++j) pNew[j] = pNew[j] - epsilon * grad[j] / 2.0;  /* :618-620 */
    for (int i = 0; i < steps - 1; ++i) {                                     /* :623-639 */
        for (int j = 0; j < n; ++j) qNew[j] = qNew[j] + epsilon * pNew[j];
        hmc_potential_gradient(h, grad, qNew);
        for (int j = 0; j < n; ++j) pNew[j] = pNew[j] - epsilon * grad[j];
        double inner = 0.0;
        for (int j = 0; j < n; ++j) inner += pNew[j] * momentum[j];
        if (inner >= 0.0) continue;
        leap_status = 2;
    }
    for (int j = 0; j < n; ++j) qNew[j] = qNew[j] + epsilon * pNew[j];         /* :641-643 */
    hmc_potential_gradient(h, grad, qNew);
    for (int j = 0; j < n; ++j) pNew[j] = pNew[j] - epsilon * grad[j] / 2.0;  /* :645-648 */
    free(momentum); free(grad);
    return leap_status;
}

/* Step :279-401 of one chain, up to but excluding UpdateCovariance / UpdateErrorMatrix (:337-341): the point
 * they are fed (pre_step) and whether they run (contributes) are left for the caller, which pools them. */
static int hmc_chain_step(oracle_hmc* h) {
    const int n = h->dim;
    ++h->step_count;
    memcpy(h->pre_step, h->accepted, sizeof(double) * (size_t)n);
    oracle_stream st; memset(&st, 0, sizeof(st));
    st.seed = h->seed; st.chain = h->chain;
    st.step = (uint64_t)(uint32_t)h->step_count;
    /* oracle_stream_word keys on SMCMC_STREAM_STEP; HMC uses its own stream */
    #define HMC_WORD(w) (smcmc_draw_block(h->seed, h->chain, (uint64_t)(uint32_t)h->step_count, (uint32_t)(w) >> 2, SMCMC_STREAM_HMC).v[(w) & 3u])
    /* ProposeMomentum :554-570 */
    if (h->alpha >= 1.0) {
        h->alpha = fmax(1.0, h->alpha);
        for (int i = 0; i < n; ++i) h->proposed_momentum[i] = h->accepted_momentum[i] / h->alpha;
    } else {
        if (h->alpha < 0.0) h->alpha = 0.0;
        for (int i = 0; i < n; ++i) {
            uint32_t p = (uint32_t)i >> 1;
            double n0, n1;
            smcmc_normal_pair(HMC_WORD(2u * p), HMC_WORD(2u * p + 1u), &n0, &n1);
            double r = (i & 1) ? n1 : n0;
            h->proposed_momentum[i] = h->alpha * h->accepted_momentum[i] + sqrt(1.0 - h->alpha * h->alpha) * r;
        }
    }
    double initial_kinetic = hmc_kinetic(n, h->proposed_momentum);            /* :292 */
    const uint32_t ew = smcmc_accept_word((uint32_t)n);
    double lo = 0.9 * fabs(h->mean_epsilon), hi = 1.1 * fabs(h->mean_epsilon);
    double epsilon = lo + (hi - lo) * smcmc_u01(HMC_WORD(ew));                /* :297-298 */
    int ok_leap = hmc_leapfrog(h, h->proposed, h->proposed_momentum, h->accepted, epsilon,
                               abs(h->leapfrog_steps));                       /* :299-300 */
    if (h->leapfrog_steps > 0) {                                              /* :302-323 */
        if (ok_leap != 2) {
            if (h->mean_epsilon > 0 && h->reversal_len > h->mean_epsilon) {
                double target = h->reversal_len / 8.0;
                double delta_eps = target - h->mean_epsilon;
                if (delta_eps > 0.0) h->mean_epsilon += 0.1 * delta_eps;
            }
            if (h->leapfrog_steps < 50) h->leapfrog_steps += 1;
        } else {
            if (h->reversal_len < h->mean_epsilon) h->reversal_len = fabs(h->leapfrog_steps * epsilon);
            else {
                h->reversal_len = 0.95 * h->reversal_len;
                h->reversal_len += 0.05 * fabs(h->leapfrog_steps * epsilon);
            }
            if (h->leapfrog_steps > 3) h->leapfrog_steps -= 1;
            if (h->mean_epsilon > 0) h->mean_epsilon *= 0.99;
        }
    }
    double proposed_kinetic = hmc_kinetic(n, h->proposed_momentum);           /* :326 */
    h->proposed_potential = hmc_potential(h, h->proposed);                    /* :327 */
    double proposed_h = h->proposed_potential + proposed_kinetic;             /* :333 */
    double accepted_h = h->accepted_potential + initial_kinetic;              /* :334 */
    h->contributes = (ok_leap && isfinite(h->proposed_potential)) ? 1 : 0;    /* :336 */
    if (!h->contributes) {                                                    /* :342-344 */
        if (h->mean_epsilon > 0) h->mean_epsilon = 0.3 * h->mean_epsilon;
    }
    double delta = proposed_h - accepted_h;                                   /* :346 */
    double trial = -smcmc_log(smcmc_u01(HMC_WORD(ew + 1u)));                  /* :347 */
    if (delta > trial || !isfinite(delta)) {                                  /* :348-368 */
        for (int i = 0; i < n; ++i) h->accepted_momentum[i] = -h->accepted_momentum[i];
        h->current_acceptance = (h->current_acceptance * 4999.0) / 5000.0;
        h->last_accept = 0;
    } else {                                                                  /* :369-387 */
        for (int i = 0; i < n; ++i) {
            h->accepted[i] = h->proposed[i];
            h->accepted_momentum[i] = h->proposed_momentum[i];
        }
        h->accepted_potential = h->proposed_potential;
        h->current_acceptance = (h->current_acceptance * 4999.0 + 1.0) / 5000.0;
        h->last_accept = 1;
    }
    if (h->accepted_potential < h->central_potential) {                       /* :393-395 */
        memcpy(h->central, h->accepted, sizeof(double) * (size_t)n);
        h->central_potential = h->accepted_potential;
    }
    #undef HMC_WORD
    return 1;                                                                 /* :399 (always true) */
}

/* packed moments of one point: M[i (i + 1) / 2 + j] = fma(y_i, y_j, M[...]), y = (x, 1) */
static void hmc_fold_point(int D, const double* x, double* M) {
    for (int i = 0; i <= D; ++i) {
        const double yi = (i < D) ? x[i] : 1.0;
        for (int j = 0; j <= i; ++j) {
            const double yj = (j < D) ? x[j] : 1.0;
            const size_t k = (size_t)i * (i + 1) / 2 + j;
            M[k] = SMCMC_FMA(yi, yj, M[k]);
        }
    }
}

/* The reference chain: a step, then UpdateCovariance + UpdateErrorMatrix (:337-341) on its own point. */
int oracle_hmc_step(oracle_hmc* h) {
    const int D = h->dim;
    hmc_chain_step(h);
    hmc_shared* s = h->shared;
    s->step_count = h->step_count;
    s->leapfrog_zero = (h->leapfrog_steps == 0);
    if (h->contributes) {
        double* M = (double*)calloc((size_t)(D + 1) * (D + 2) / 2, sizeof(double));
        hmc_fold_point(D, h->pre_step, M);
        hmc_shared_absorb(s, M, 1);
        free(M);
        if (hmc_shared_update_error_matrix(s)) hmc_chain_retune(&h->mean_epsilon, &h->leapfrog_steps, s);
    }
    return 1;
}

void oracle_hmc_run(oracle_hmc* h, int nsteps) { for (int s = 0; s < nsteps; ++s) oracle_hmc_step(h); }
void oracle_hmc_get_accepted(const oracle_hmc* h, double* out) { memcpy(out, h->accepted, sizeof(double) * (size_t)h->dim); }
void oracle_hmc_get_momentum(const oracle_hmc* h, double* out) { memcpy(out, h->accepted_momentum, sizeof(double) * (size_t)h->dim); }
void oracle_hmc_get_central(const oracle_hmc* h, double* out) { memcpy(out, h->central, sizeof(double) * (size_t)h->dim); }
/* 0 accepted_potential 1 proposed_potential 2 current_acceptance 3 mean_epsilon 4 leapfrog_steps
 * 5 step_count 6 potential_count 7 gradient_count 8 last_accept 9 central_potential 10 reversal_len
 * 11 trace (fCurrentCovarianceTrace) 12 orbit (fEstimatedOrbitLength) 13 covariance updates 14 cov_trials */
void oracle_hmc_get_scalars(const oracle_hmc* h, double* out) {
    out[0] = h->accepted_potential; out[1] = h->proposed_potential; out[2] = h->current_acceptance;
    out[3] = h->mean_epsilon; out[4] = h->leapfrog_steps; out[5] = h->step_count; out[6] = h->potential_count;
    out[7] = h->gradient_count; out[8] = h->last_accept; out[9] = h->central_potential; out[10] = h->reversal_len;
    out[11] = h->shared->cur_trace; out[12] = h->shared->orbit_length; out[13] = h->shared->update_count;
    out[14] = h->shared->cov_trials;
}
void oracle_hmc_get_average(const oracle_hmc* h, double* out) { memcpy(out, h->shared->average, sizeof(double) * (size_t)h->dim); }
void oracle_hmc_get_covariance(const oracle_hmc* h, double* out) { memcpy(out, h->shared->cov, sizeof(double) * (size_t)h->dim * (size_t)h->dim); }

/* ---- the many-chain engine ------------------------------------------------------------------------ */
typedef struct {
    int nchains, dim, group, sync_every;
    oracle_hmc** chain;
    hmc_shared* shared;
    int ngroups;
    double* acc;          /* [group][packed] */
    int steps_in_window;
} oracle_hmc_ensemble;

static size_t hmce_npacked(int D) { return (size_t)(D + 1) * (D + 2) / 2; }

/* group: chains per moment group (the engine's SMCMC_HMC moment group); sync_every: steps per pooled update */
oracle_hmc_ensemble* oracle_hmc_ensemble_create(int nchains, int dim, int like_kind, const double* like_params,
                                                int n_like_params, uint64_t seed, uint32_t chain_offset, int group,
                                                int sync_every) {
    oracle_hmc_ensemble* e = (oracle_hmc_ensemble*)calloc(1, sizeof(oracle_hmc_ensemble));
    e->nchains = nchains; e->dim = dim; e->group = group; e->sync_every = sync_every;
    e->shared = hmc_shared_create(dim);
    e->chain = (oracle_hmc**)calloc((size_t)nchains, sizeof(oracle_hmc*));
    for (int c = 0; c < nchains; ++c) {
        e->chain[c] = oracle_hmc_create(dim, like_kind, like_params, n_like_params, seed, chain_offset + (uint32_t)c);
        hmc_shared_destroy(e->chain[c]->shared);
        e->chain[c]->shared = e->shared;
        e->chain[c]->owns_shared = 0;
    }
    e->ngroups = (nchains + group - 1) / group;
    e->acc = (double*)calloc((size_t)e->ngroups * hmce_npacked(dim), sizeof(double));
    return e;
}
void oracle_hmc_ensemble_destroy(oracle_hmc_ensemble* e) {
    if (!e) return;
    for (int c = 0; c < e->nchains; ++c) oracle_hmc_destroy(e->chain[c]);
    hmc_shared_destroy(e->shared);
    free(e->chain); free(e->acc); free(e);
}
oracle_hmc* oracle_hmc_ensemble_chain(oracle_hmc_ensemble* e, int c) { return e->chain[c]; }
void oracle_hmc_ensemble_configure(oracle_hmc_ensemble* e, double alpha, int potential_from_gradient, int fused_gradient,
                                   int gradient_type) {
    for (int c = 0; c < e->nchains; ++c) {
        e->chain[c]->alpha = alpha;
        e->chain[c]->potential_from_gradient = potential_from_gradient;
        e->chain[c]->fused_gradient = fused_gradient;
        e->chain[c]->gradient_type = gradient_type;
    }
}
void oracle_hmc_ensemble_set_mean_epsilon(oracle_hmc_ensemble* e, double eps) { for (int c = 0; c < e->nchains; ++c) e->chain[c]->mean_epsilon = eps; }
void oracle_hmc_ensemble_set_leapfrog(oracle_hmc_ensemble* e, int n) { for (int c = 0; c < e->nchains; ++c) e->chain[c]->leapfrog_steps = -n; }
/* x0: [dim] (broadcast) or [dim][nchains]; the shared covariance state starts from chain 0's point */
void oracle_hmc_ensemble_start(oracle_hmc_ensemble* e, const double* x0, int broadcast) {
    const int D = e->dim, N = e->nchains;
    double* p = (double*)malloc(sizeof(double) * (size_t)D);
    for (int c = N - 1; c >= 0; --c) {
        for (int d = 0; d < D; ++d) p[d] = broadcast ? x0[d] : x0[(size_t)d * N + c];
        hmc_chain_start(e->chain[c], p);
    }
    hmc_shared_start(e->shared, p);     /* p is chain 0's start */
    memset(e->acc, 0, sizeof(double) * (size_t)e->ngroups * hmce_npacked(D));
    e->steps_in_window = 0;
    free(p);
}
/* the pooled update: groups summed in ascending order, running averages fed with the batch, UpdateErrorMatrix once,
 * every chain retuned when it went through */
void oracle_hmc_ensemble_sync(oracle_hmc_ensemble* e) {
    const int D = e->dim;
    const size_t npk = hmce_npacked(D);
    hmc_shared* s = e->shared;
    double* M = (double*)calloc(npk, sizeof(double));
    /* the engine's reduction order (fold_reduce_kernel, as oracle_ensemble_reduce_moments): groups in ascending order
     * within chunks of 32, then the chunk sums in ascending order */
    for (size_t k = 0; k < npk; ++k) {
        double total = 0.0;
        for (int g0 = 0; g0 < e->ngroups; g0 += 32) {
            double t = 0.0;
            for (int g = g0; g < e->ngroups && g < g0 + 32; ++g) t += e->acc[(size_t)g * npk + k];
            total += t;
        }
        M[k] = total;
    }
    memset(e->acc, 0, sizeof(double) * (size_t)e->ngroups * npk);
    const int steps = e->steps_in_window;
    e->steps_in_window = 0;
    s->fired = 0;
    if (M[npk - 1] > 0.0) {
        s->step_count = e->chain[0]->step_count;
        s->leapfrog_zero = (e->chain[0]->leapfrog_steps == 0);
        hmc_shared_absorb(s, M, steps);
        if (hmc_shared_update_error_matrix(s))
            for (int c = 0; c < e->nchains; ++c) hmc_chain_retune(&e->chain[c]->mean_epsilon, &e->chain[c]->leapfrog_steps, s);
    }
    free(M);
}
void oracle_hmc_ensemble_step(oracle_hmc_ensemble* e, int nsteps) {
    const int D = e->dim;
    const size_t npk = hmce_npacked(D);
    for (int st = 0; st < nsteps; ++st) {
        for (int c = 0; c < e->nchains; ++c) {
            oracle_hmc* h = e->chain[c];
            hmc_chain_step(h);
            if (h->contributes) hmc_fold_point(D, h->pre_step, e->acc + (size_t)(c / e->group) * npk);
        }
        if (++e->steps_in_window >= e->sync_every) oracle_hmc_ensemble_sync(e);
    }
}
/* [dim][chain] */
void oracle_hmc_ensemble_get_state(const oracle_hmc_ensemble* e, double* q, double* momentum) {
    const int D = e->dim, N = e->nchains;
    for (int c = 0; c < N; ++c)
        for (int d = 0; d < D; ++d) {
            if (q) q[(size_t)d * N + c] = e->chain[c]->accepted[d];
            if (momentum) momentum[(size_t)d * N + c] = e->chain[c]->accepted_momentum[d];
        }
}
/* field: the index of oracle_hmc_get_scalars */
void oracle_hmc_ensemble_get_lane(const oracle_hmc_ensemble* e, int field, double* out) {
    double sc[16];
    for (int c = 0; c < e->nchains; ++c) {
        oracle_hmc_get_scalars(e->chain[c], sc);
        out[c] = sc[field];
    }
}
void oracle_hmc_ensemble_get_average(const oracle_hmc_ensemble* e, double* out) { memcpy(out, e->shared->average, sizeof(double) * (size_t)e->dim); }
void oracle_hmc_ensemble_get_covariance(const oracle_hmc_ensemble* e, double* out) { memcpy(out, e->shared->cov, sizeof(double) * (size_t)e->dim * (size_t)e->dim); }
/* 0 trace 1 orbit 2 updates 3 cov_trials 4 average_trials 5 steps_remaining 6 steps_since_update 7 max_scale 8 min_scale 9 est_trace */
void oracle_hmc_ensemble_get_shared(const oracle_hmc_ensemble* e, double* out) {
    const hmc_shared* s = e->shared;
    out[0] = s->cur_trace; out[1] = s->orbit_length; out[2] = s->update_count; out[3] = s->cov_trials;
    out[4] = s->average_trials; out[5] = s->steps_remaining; out[6] = s->steps_since_update; out[7] = s->max_scale;
    out[8] = s->min_scale; out[9] = s->est_trace;
}
void oracle_hmc_gradient(int kind, int dim, const double* p, const double* params, double* g) {
    switch (kind) {
        case ORACLE_LIKE_ISO: hmc_grad_iso(dim, g, p); break;
        case ORACLE_LIKE_QUADFORM: hmc_grad_quadform(dim, g, p, params); break;
        default: hmc_grad_rosenbrock(dim, g, p, params ? params[0] : 100.0); break;
    }
}
