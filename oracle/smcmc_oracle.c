/* smcmc_oracle.c -- TEST INFRASTRUCTURE ONLY.  C-callable surface of the CPU
 * oracle (see oracle_core.h for the restatement itself and its parity status:
 * PARITY UNPINNED).  Two oracles live here:
 *
 *   oracle_chain_*     one reference chain: TSimpleMCMC<L,TProposeAdaptiveStep>
 *                      exactly as /root/reference/TSimpleMCMC.H runs it, with
 *                      gRandom replaced by the Philox stream of chain c.
 *   oracle_ensemble_*  the many-chain engine semantics the HIP path implements
 *                      (DESIGN.md "Ensemble semantics"): per-chain scalar state
 *                      follows TSimpleMCMC.H:1723-1776 verbatim, the covariance is
 *                      pooled over chains through ordered moment sums.
 */
#include <stdio.h>
#include "oracle_core.h"

/* ======================= single reference chain ========================== */

oracle_chain* oracle_chain_create(int dim, int like_kind, const double* like_params,
                                  int n_like_params, uint64_t seed, uint32_t chain_id) {
    oracle_chain* c = (oracle_chain*)calloc(1, sizeof(oracle_chain));
    c->dim = dim;
    c->like_kind = like_kind;
    if (n_like_params > 0) {
        c->like_params = (double*)malloc(sizeof(double) * (size_t)n_like_params);
        memcpy(c->like_params, like_params, sizeof(double) * (size_t)n_like_params);
    }
    oracle_proposal_init(&c->prop);
    oracle_proposal_set_dim(&c->prop, dim);
    c->stream.seed = seed;
    c->stream.chain = chain_id;
    c->accepted = (double*)calloc((size_t)dim, sizeof(double));
    c->proposed = (double*)calloc((size_t)dim, sizeof(double));
    c->trial_step = (double*)calloc((size_t)dim, sizeof(double));
    c->step_rms = 0.0;            /* TSimpleMCMC.H:221-223 */
    c->step_rms_trials = 0;
    c->step_rms_window = 1000;
    return c;
}

void oracle_chain_destroy(oracle_chain* c) {
    if (!c) return;
    oracle_proposal_free(&c->prop);
    free(c->like_params); free(c->accepted); free(c->proposed); free(c->trial_step);
    free(c);
}

/* setters mirroring TSimpleMCMC.H:786-1003; call before oracle_chain_start */
void oracle_chain_set_gaussian(oracle_chain* c, int d, double sigma) {      /* :855-867 */
    if (d < 0 || d >= c->dim) return;
    c->prop.ptype[d] = 0; c->prop.pparam1[d] = sigma * sigma;
}
void oracle_chain_set_uniform(oracle_chain* c, int d, double lo, double hi) {   /* :833-848 */
    if (d < 0 || d >= c->dim) return;
    c->prop.ptype[d] = 1; c->prop.pparam1[d] = lo; c->prop.pparam2[d] = hi;
}
void oracle_chain_set_correlation(oracle_chain* c, int d1, int d2, double corr) {
    oracle_proposal_set_correlation(&c->prop, d1, d2, corr);
}
void oracle_chain_set_covariance_frozen(oracle_chain* c, int f) { c->prop.cov_frozen = f; }
void oracle_chain_set_covariance_window(oracle_chain* c, int w) { c->prop.cov_window = w; }   /* :914 int arg */
void oracle_chain_set_covariance_deweight(oracle_chain* c, double d) { c->prop.cov_deweight = d; }
void oracle_chain_set_acceptance_window(oracle_chain* c, double w) { c->prop.acceptance_window = w; }
void oracle_chain_set_acceptance_deweight(oracle_chain* c, double d) { c->prop.acceptance_deweight = d; }
void oracle_chain_set_acceptance_rigidity(oracle_chain* c, double r) { c->prop.rigidity = r; }
void oracle_chain_set_target_acceptance(oracle_chain* c, double a) { c->prop.target = a; }
void oracle_chain_set_next_update(oracle_chain* c, double n) { c->prop.next_update = (int)n; }   /* :992 */
void oracle_chain_set_sigma(oracle_chain* c, double s) { c->prop.sigma = s; }
void oracle_chain_set_step_rms_window(oracle_chain* c, int n) { c->step_rms_window = n; }
void oracle_chain_set_scan_dimension(oracle_chain* c, int d) {              /* :820-830 */
    if (d < 0 || d >= c->dim) c->prop.scan_dim = -1; else c->prop.scan_dim = d;
}
void oracle_chain_force_step(oracle_chain* c, const double* p) {            /* :811-817 */
    memcpy(c->prop.forced, p, sizeof(double) * (size_t)c->dim);
    c->prop.has_forced = 1;
}
/* fCurrentCov overwritten (the engine's SetCovariance hook), SetCovarianceTrials :947, SetEstimatedCenterTrials :747 */
void oracle_chain_set_covariance(oracle_chain* c, const double* cov) {
    memcpy(c->prop.cov, cov, sizeof(double) * (size_t)c->dim * (size_t)c->dim);
}
void oracle_chain_set_covariance_trials(oracle_chain* c, double t) { c->prop.cov_trials = t; }
void oracle_chain_set_center_trials(oracle_chain* c, double t) { c->prop.central_trials = t; }
void oracle_chain_update_proposal(oracle_chain* c) { oracle_proposal_update(&c->prop, 0); }
void oracle_chain_reset_proposal(oracle_chain* c) { oracle_proposal_reset(&c->prop); }

int oracle_chain_start_api(oracle_chain* c, const double* start) { return oracle_chain_start(c, start); }
void oracle_chain_restore_api(oracle_chain* c, const double* accepted, double saved_logl, int total_steps,
                              double step_rms, int trials, int successes, int next_update, double acceptance,
                              double acceptance_trials, double sigma, const double* central,
                              double central_trials, const double* cov_packed, double cov_trials) {
    oracle_chain_restore(c, accepted, saved_logl, total_steps, step_rms, trials, successes, next_update,
                         acceptance, acceptance_trials, sigma, central, central_trials, cov_packed, cov_trials);
}
int oracle_chain_step_api(oracle_chain* c, int save, int metropolis) { return oracle_chain_step(c, save, metropolis); }

/* Run n steps, recording per-step observables (any out pointer may be NULL).
 * Scalars (sigma, acceptance, trials, successes) are read after Step() returns,
 * i.e. they carry the reference's one-call lag. */
void oracle_chain_run(oracle_chain* c, int nsteps, int metropolis, uint8_t* accepted_flag,
                      double* logl_proposed, double* logl_accepted, double* sigma,
                      double* acceptance, int32_t* trials, int32_t* successes, double* step_rms) {
    for (int s = 0; s < nsteps; ++s) {
        int a = oracle_chain_step(c, 0, metropolis);
        if (accepted_flag) accepted_flag[s] = (uint8_t)a;
        if (logl_proposed) logl_proposed[s] = c->proposed_logl;
        if (logl_accepted) logl_accepted[s] = c->accepted_logl;
        if (sigma) sigma[s] = c->prop.sigma;
        if (acceptance) acceptance[s] = c->prop.acceptance;
        if (trials) trials[s] = c->prop.trials;
        if (successes) successes[s] = c->prop.successes;
        if (step_rms) step_rms[s] = c->step_rms;
    }
}

/* Run n steps accumulating the running mean/second moment of the accepted point
 * (for the posterior checks), no per-step output. */
void oracle_chain_run_moments(oracle_chain* c, int nsteps, double* sum, double* sumsq_full, int* n_accept) {
    const int n = c->dim;
    int acc = 0;
    for (int s = 0; s < nsteps; ++s) {
        acc += oracle_chain_step(c, 0, 0);
        for (int i = 0; i < n; ++i) {
            sum[i] += c->accepted[i];
            for (int j = 0; j < n; ++j) sumsq_full[i * n + j] += c->accepted[i] * c->accepted[j];
        }
    }
    if (n_accept) *n_accept = acc;
}

void oracle_chain_get_accepted(const oracle_chain* c, double* out) { memcpy(out, c->accepted, sizeof(double) * (size_t)c->dim); }
void oracle_chain_get_proposed(const oracle_chain* c, double* out) { memcpy(out, c->proposed, sizeof(double) * (size_t)c->dim); }
void oracle_chain_get_center(const oracle_chain* c, double* out) { memcpy(out, c->prop.central, sizeof(double) * (size_t)c->dim); }
void oracle_chain_get_covariance(const oracle_chain* c, double* out) { memcpy(out, c->prop.cov, sizeof(double) * (size_t)c->dim * (size_t)c->dim); }
void oracle_chain_get_decomposition(const oracle_chain* c, double* out) { memcpy(out, c->prop.decomp, sizeof(double) * (size_t)c->dim * (size_t)c->dim); }

/* scalars: 0 accepted_logl 1 proposed_logl 2 sigma 3 acceptance 4 acceptance_trials
 * 5 acceptance_window 6 rigidity 7 target 8 sigma_trace 9 cov_trials 10 central_trials
 * 11 cov_window 12 trials 13 successes 14 next_update 15 total_steps 16 like_count
 * 17 step_rms 18 update_count 19 last_update_path 20 failed 21 step_rms_trials */
void oracle_chain_get_scalars(const oracle_chain* c, double* out) {
    const oracle_proposal* p = &c->prop;
    out[0] = c->accepted_logl; out[1] = c->proposed_logl; out[2] = p->sigma; out[3] = p->acceptance;
    out[4] = p->acceptance_trials; out[5] = p->acceptance_window; out[6] = p->rigidity; out[7] = p->target;
    out[8] = p->sigma_trace; out[9] = p->cov_trials; out[10] = p->central_trials; out[11] = p->cov_window;
    out[12] = p->trials; out[13] = p->successes; out[14] = p->next_update; out[15] = c->total_steps;
    out[16] = c->like_count; out[17] = c->step_rms; out[18] = p->update_count; out[19] = p->last_update_path;
    out[20] = p->failed; out[21] = c->step_rms_trials;
}

/* ===================== detmath / helper exports (tests) ================== */

void oracle_log_v(int n, const double* x, double* out) { for (int i = 0; i < n; ++i) out[i] = smcmc_log(x[i]); }
void oracle_exp_v(int n, const double* x, double* out) { for (int i = 0; i < n; ++i) out[i] = smcmc_exp(x[i]); }
void oracle_pow_small_v(int n, const double* x, const double* y, double* out) {
    for (int i = 0; i < n; ++i) out[i] = smcmc_pow_small(x[i], y[i]);
}
void oracle_sincos2pi_v(int n, const double* u, double* s, double* c) {
    for (int i = 0; i < n; ++i) smcmc_sincos2pi(u[i], &s[i], &c[i]);
}
/* words are passed as doubles holding exact 32-bit integers */
void oracle_sincos2pi_u32_v(int n, const double* w, double* s, double* c) {
    for (int i = 0; i < n; ++i) smcmc_sincos2pi_u32((uint32_t)w[i], &s[i], &c[i]);
}
void oracle_u01_v(int n, const double* w, double* out) { for (int i = 0; i < n; ++i) out[i] = smcmc_u01((uint32_t)w[i]); }
void oracle_normal_pair_v(int n, const double* w0, const double* w1, double* n0, double* n1) {
    for (int i = 0; i < n; ++i) smcmc_normal_pair((uint32_t)w0[i], (uint32_t)w1[i], &n0[i], &n1[i]);
}

/* The half-circle form of the pair (SMCMC_NORMAL_PAIR_BODY_HALFCIRCLE, what the step kernels run with a 128-entry angle
 * table in LDS), with the table built the way the kernel builds it: tests/test_detmath.py checks it bit for bit against
 * smcmc_normal_pair. */
static void oracle_normal_pair_halfcircle(uint32_t w0, uint32_t w1, double* n0, double* n1) {
    static double at2[256];
    static int ready = 0;
    if (!ready) {
        for (int k = 0; k < 64; ++k) {
            const double c = smcmc_angle_table_host[2 * k], sn = smcmc_angle_table_host[2 * k + 1];
            at2[2 * k] = c;
            at2[2 * k + 1] = sn;
            at2[128 + 2 * k] = -sn;
            at2[128 + 2 * k + 1] = c;
        }
        ready = 1;
    }
    const uint32_t idx_ = smcmc_normal_angle_index_halfcircle(w1);
#define ORACLE_LT(k, c) smcmc_log_table_host[2u * (k) + (c)]
#define ORACLE_AT2(c) at2[2u * idx_ + (c)]
    SMCMC_NORMAL_PAIR_BODY_HALFCIRCLE(ORACLE_LT, ORACLE_AT2)
#undef ORACLE_LT
#undef ORACLE_AT2
}
void oracle_normal_pair_halfcircle_v(int n, const double* w0, const double* w1, double* n0, double* n1) {
    for (int i = 0; i < n; ++i) oracle_normal_pair_halfcircle((uint32_t)w0[i], (uint32_t)w1[i], &n0[i], &n1[i]);
}
/* `rounds` rounds starting at round `first` of the key schedule (the engine draws with SMCMC_PHILOX_ROUNDS rounds) */
void oracle_philox_rounds(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, int first,
                          int rounds, uint32_t* out) {
    smcmc_u32x4 r = smcmc_philox4x32_rounds(c0, c1, c2, c3, k0, k1, first, rounds);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
int oracle_philox_draw_rounds(void) { return SMCMC_PHILOX_ROUNDS; }
/* the block smcmc_draw_block hands out for (seed, chain, step, block, stream) */
void oracle_draw_block(uint64_t seed, uint32_t chain, uint64_t step, uint32_t block, uint32_t stream, uint32_t* out) {
    smcmc_u32x4 r = smcmc_draw_block(seed, chain, step, block, stream);
    for (int i = 0; i < 4; ++i) out[i] = r.v[i];
}
void oracle_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t* out) {
    smcmc_u32x4 r = smcmc_philox4x32_10(c0, c1, c2, c3, k0, k1);
    memcpy(out, r.v, sizeof(r.v));
}
/* normals of one chain-step: out[i] = N(0,1) for dimension i, *u = Metropolis uniform */
void oracle_step_draws(uint64_t seed, uint32_t chain, uint64_t step, int dim, double* normals, double* u) {
    oracle_stream s; memset(&s, 0, sizeof(s));
    s.seed = seed; s.chain = chain;
    oracle_stream_set_step(&s, step);
    for (int i = 0; i < dim; ++i) normals[i] = oracle_stream_normal(&s, i);
    *u = oracle_stream_uniform(&s, smcmc_accept_word((uint32_t)dim));
}
int oracle_cholesky(int n, const double* A, double* U) { return oracle_cholesky_upper(n, A, U); }
void oracle_eigen(int n, const double* A, double* vec, double* val) { oracle_sym_eigen(n, A, vec, val); }
double oracle_loglike(int kind, int dim, const double* p, const double* params) { return oracle_like(kind, dim, p, params); }

/* TDummyLogLikelihood::Init() (TDummyLogLikelihood.H:44-142) for a general
 * dimension: unit variances, VERY_CORRELATED 0.999999 which the std::abs(d) <
 * GetDim()-1 guard (:82-87) leaves only on the pair (0, D-1); Error = Cov^-1 by
 * Gauss-Jordan with partial pivoting (ROOT's TMatrixD::Invert is unpinned; the
 * same matrix is handed to the oracle and the GPU as an input). */
int oracle_dummy_error_matrix(int dim, double* cov_out, double* err_out) {
    const int n = dim;
    double* cov = cov_out;
    for (int i = 0; i < n * n; ++i) cov[i] = 0.0;
    for (int i = 0; i < n; ++i) { double sigma = 1.0; cov[i * n + i] = sigma * sigma; }
    for (int i = 0; i < n; ++i) {
        for (int j = i + 1; j < n; ++j) {
            double sig1 = sqrt(cov[i * n + i]);
            double sig2 = sqrt(cov[j * n + j]);
            if ((i + j) == n - 1) {
                int d = i - j;
                if (abs(d) < n - 1) continue;
                cov[i * n + j] = 0.999999 * sig1 * sig2 * (double)(j - i) / (n - 1.0);
            }
            cov[j * n + i] = cov[i * n + j];
        }
    }
    /* invert */
    double* a = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n * 2);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            a[i * 2 * n + j] = cov[i * n + j];
            a[i * 2 * n + n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int col = 0; col < n; ++col) {
        int piv = col;
        for (int r = col + 1; r < n; ++r) if (fabs(a[r * 2 * n + col]) > fabs(a[piv * 2 * n + col])) piv = r;
        if (a[piv * 2 * n + col] == 0.0) { free(a); return 0; }
        if (piv != col)
            for (int k = 0; k < 2 * n; ++k) { double t = a[col * 2 * n + k]; a[col * 2 * n + k] = a[piv * 2 * n + k]; a[piv * 2 * n + k] = t; }
        double d = a[col * 2 * n + col];
        for (int k = 0; k < 2 * n; ++k) a[col * 2 * n + k] /= d;
        for (int r = 0; r < n; ++r) {
            if (r == col) continue;
            double f = a[r * 2 * n + col];
            if (f == 0.0) continue;
            for (int k = 0; k < 2 * n; ++k) a[r * 2 * n + k] -= f * a[col * 2 * n + k];
        }
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) err_out[i * n + j] = a[i * 2 * n + n + j];
    free(a);
    return 1;
}
