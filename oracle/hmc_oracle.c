/* hmc_oracle.c -- TEST INFRASTRUCTURE ONLY.  CPU restatement of sMCMC::TSimpleHMC
 * (/root/reference/TSimpleHMC.H:119-973) for one chain, with the analytic
 * gradients of the shipped likelihoods, and of the many-chain HMC engine the HIP
 * path implements (every chain an independent reference chain: HMC has no shared
 * state once epsilon and the leapfrog count are fixed).  PARITY UNPINNED, as for
 * the rest of oracle/ (no golden vectors exist, ROOT is absent).
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
} oracle_hmc;

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

/* PotentialGradient type 0 with a user gradient (:467-492): grad = -gradLogL */
static void hmc_potential_gradient(oracle_hmc* h, double* grad, const double* p) {
    ++h->gradient_count;
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
    return h;
}

void oracle_hmc_destroy(oracle_hmc* h) {
    if (!h) return;
    free(h->like_params); free(h->accepted); free(h->accepted_momentum); free(h->proposed);
    free(h->proposed_momentum); free(h->central); free(h);
}

void oracle_hmc_set_alpha(oracle_hmc* h, double a) { h->alpha = a; }                 /* :175 */
void oracle_hmc_set_mean_epsilon(oracle_hmc* h, double e) { h->mean_epsilon = e; }   /* :181 */
void oracle_hmc_set_leapfrog(oracle_hmc* h, int n) { h->leapfrog_steps = -n; }       /* :190 */
void oracle_hmc_set_potential_from_gradient(oracle_hmc* h, int f) { h->potential_from_gradient = f; }
void oracle_hmc_set_fused_gradient(oracle_hmc* h, int f) { h->fused_gradient = f; }

/* Start :210-269 (the covariance bookkeeping is not restated: with a negative mean
 * epsilon and SetLeapFrog the chain never reads it) */
void oracle_hmc_start(oracle_hmc* h, const double* start) {
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

/* Step :279-401 (UpdateCovariance / UpdateErrorMatrix, :337-341, only feed the epsilon
 * and leapfrog auto-tuning, which a negative mean epsilon + SetLeapFrog switch off) */
int oracle_hmc_step(oracle_hmc* h) {
    const int n = h->dim;
    ++h->step_count;
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
    if (!(ok_leap && isfinite(h->proposed_potential))) {                      /* :336-344 */
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

void oracle_hmc_run(oracle_hmc* h, int nsteps) { for (int s = 0; s < nsteps; ++s) oracle_hmc_step(h); }
void oracle_hmc_get_accepted(const oracle_hmc* h, double* out) { memcpy(out, h->accepted, sizeof(double) * (size_t)h->dim); }
void oracle_hmc_get_momentum(const oracle_hmc* h, double* out) { memcpy(out, h->accepted_momentum, sizeof(double) * (size_t)h->dim); }
void oracle_hmc_get_central(const oracle_hmc* h, double* out) { memcpy(out, h->central, sizeof(double) * (size_t)h->dim); }
/* 0 accepted_potential 1 proposed_potential 2 current_acceptance 3 mean_epsilon 4 leapfrog_steps
 * 5 step_count 6 potential_count 7 gradient_count 8 last_accept 9 central_potential */
void oracle_hmc_get_scalars(const oracle_hmc* h, double* out) {
    out[0] = h->accepted_potential; out[1] = h->proposed_potential; out[2] = h->current_acceptance;
    out[3] = h->mean_epsilon; out[4] = h->leapfrog_steps; out[5] = h->step_count; out[6] = h->potential_count;
    out[7] = h->gradient_count; out[8] = h->last_accept; out[9] = h->central_potential;
}
void oracle_hmc_gradient(int kind, int dim, const double* p, const double* params, double* g) {
    switch (kind) {
        case ORACLE_LIKE_ISO: hmc_grad_iso(dim, g, p); break;
        case ORACLE_LIKE_QUADFORM: hmc_grad_quadform(dim, g, p, params); break;
        default: hmc_grad_rosenbrock(dim, g, p, params ? params[0] : 100.0); break;
    }
}
