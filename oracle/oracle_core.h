/* oracle_core.h -- TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference
 * hot path: sMCMC::TSimpleMCMC::Step() with sMCMC::TProposeAdaptiveStep and the
 * three likelihood functors the benchmark configs use.  Every function cites the
 * /root/reference file:line it follows.  Nothing in the product (the package
 * root-simple-mcmc_amd/, include/) may include or link this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as a checker.
 *
 * PARITY UNPINNED: the reference ships no tests, golden vectors or fixed seeds
 * (SURVEY.md section 4), and it cannot be built here (ROOT headers/libraries are
 * absent and writing stand-ins for them is not allowed), so this restatement is
 * pinned only by the known-answer values recorded in SURVEY.md section 8c
 * (defaults after Start, the lag/spurious-reject quirk, analytic posteriors),
 * which tests/test_oracle_known_answers.py checks.
 *
 * Arithmetic: plain IEEE binary64 in the reference's operation order, compiled
 * with -ffp-contract=off (the reference is built with g++ -O2, no -march, no
 * fast-math: mcmc-compile.sh:3-7).  gRandom is replaced by the Philox draw-slot
 * convention of include/smcmc_detmath.h; std::log / std::pow by smcmc_log /
 * smcmc_pow_small from the same header (<= 1 ulp from libm).
 */
#ifndef ORACLE_CORE_H_SEEN
#define ORACLE_CORE_H_SEEN

#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/smcmc_detmath.h"
#include "oracle_linalg.h"

/* 3 is the engine's SMCMC_LIKE_USER (a likelihood compiled in from user source) */
enum { ORACLE_LIKE_ISO = 0, ORACLE_LIKE_QUADFORM = 1, ORACLE_LIKE_ROSENBROCK = 2, ORACLE_LIKE_ASYM = 4,
       ORACLE_LIKE_HORRIFIC = 5, ORACLE_LIKE_CONSTRAINED = 6 };

/* ---- likelihood functors ------------------------------------------------ */

/* README.md:57-66 / TSimpleMCMC.H:111-120: logL += - 0.5*p[i]*p[i], ascending i.
 * "- 0.5*p*p" parses as ((-0.5)*p)*p. */
static double oracle_like_iso(int dim, const double* p) {
    double logl = 0.0;
    for (int i = 0; i < dim; ++i) logl += -0.5 * p[i] * p[i];
    return logl;
}

/* TDummyLogLikelihood.H:21-31: logL -= 0.5*p[i]*Error(j,i)*p[j], i outer, j inner. */
static double oracle_like_quadform(int dim, const double* p, const double* E) {
    double logl = 0.0;
    for (int i = 0; i < dim; ++i)
        for (int j = 0; j < dim; ++j) logl -= 0.5 * p[i] * E[j * dim + i] * p[j];
    return logl;
}

/* THardLogLikelihood.H:57-67: a=(1-p[i]); b=p[i+1]-p[i]*p[i]; logL -= a*a + B*b*b. */
static double oracle_like_rosenbrock(int dim, const double* p, double rosen_b) {
    double logl = 0.0;
    for (int i = 0; i < dim - 1; ++i) {
        double a = (1.0 - p[i]);
        double b = p[i + 1] - p[i] * p[i];
        logl -= a * a + rosen_b * b * b;
    }
    return logl;
}

/* TAsymLogLikelihood.H:20-31: a = p[i]; a *= (a < 0) ? negativeSlope : positiveSlope; logL += a.
 * params = {positiveSlope, negativeSlope}, the reference's constants (-1, 100) by default (:17-18). */
static double oracle_like_asym(int dim, const double* p, const double* params, int nparams) {
    const double positive = (nparams >= 2) ? params[0] : -1.0, negative = (nparams >= 2) ? params[1] : 100.0;
    double logl = 0.0;
    for (int i = 0; i < dim; ++i) {
        double a = p[i];
        if (a < 0.0) a *= negative; else a *= positive;
        logl += a;
    }
    return logl;
}

/* THorrificLogLikelihood.H:26-38: -1E+30 outside the unit box, else -0.5 (sum p / sqrt(D 4/12))^2 / 0.01^2 */
static double oracle_like_horrific(int dim, const double* p) {
    const double sigma = 0.01;
    double logl = 0.0;
    for (int i = 0; i < dim; ++i) {
        if (fabs(p[i]) > 1.0) return -1E+30;
        logl += p[i];
    }
    double natural_sigma = sqrt(dim * 4.0 / 12.0);
    logl /= natural_sigma;
    logl = -0.5 * logl * logl / sigma / sigma;
    return logl;
}

/* example4/TConstrainedLikelihood.H:26-46: the sum constrained to SummedValues +- SummedConstraint, every value to
 * ExpectedValues[i] +- PriorConstraints[i].  params = {SummedValues, SummedConstraint, Expected[D], Prior[D]}. */
static double oracle_like_constrained(int dim, const double* p, const double* params) {
    double logl = 0.0;
    double sum = 0.0;
    for (int i = 0; i < dim; ++i) sum += p[i];
    sum = (sum - params[0]) / params[1];
    logl -= 0.5 * sum * sum;
    for (int i = 0; i < dim; ++i) {
        double v = p[i] - params[2 + i];
        v /= params[2 + dim + i];
        logl -= 0.5 * v * v;
    }
    return logl;
}

static double oracle_like(int kind, int dim, const double* p, const double* params) {
    switch (kind) {
        case ORACLE_LIKE_ISO: return oracle_like_iso(dim, p);
        case ORACLE_LIKE_ASYM: return oracle_like_asym(dim, p, params, params ? 2 : 0);
        case ORACLE_LIKE_HORRIFIC: return oracle_like_horrific(dim, p);
        case ORACLE_LIKE_CONSTRAINED: return oracle_like_constrained(dim, p, params);
        case ORACLE_LIKE_QUADFORM: return oracle_like_quadform(dim, p, params);
        default: return oracle_like_rosenbrock(dim, p, params ? params[0] : 100.0);
    }
}

/* The likelihood in one of the engine's two arithmetic orders: exact = the reference's operation order (oracle_like),
 * otherwise the fused multiply-add order of the HIP kernels' "fast" arithmetic. */
static double oracle_like_order(int kind, int n, const double* p, const double* params, int exact, int quadform_rowwise) {
    /* the stress likelihoods (asymmetric, horrific, constrained) have no fused form: the same arithmetic in both orders */
    if (exact || kind >= ORACLE_LIKE_ASYM) return oracle_like(kind, n, p, params);
    double logl = 0.0;
    switch (kind) {
        case ORACLE_LIKE_ISO:
            for (int i = 0; i < n; ++i) logl = SMCMC_FMA(-0.5 * p[i], p[i], logl);
            return logl;
        case ORACLE_LIKE_QUADFORM:
            if (quadform_rowwise) {
                /* the matrix-pipe kernel's association (smcmc_panel_mfma_kernel.hip.h): row sums of
                 * Error p by fused multiply-adds, j ascending, then the outer sum in dimension order */
                double usum = 0.0;
                for (int i = 0; i < n; ++i) {
                    double s = 0.0;
                    for (int j = 0; j < n; ++j) s = SMCMC_FMA(params[i * n + j], p[j], s);
                    usum += 0.5 * p[i] * s;
                }
                return -usum;
            }
            for (int i = 0; i < n; ++i) {
                double h = 0.5 * p[i];
                for (int j = 0; j < n; ++j) logl = SMCMC_FMA(-(h * params[j * n + i]), p[j], logl);
            }
            return logl;
        default: {
            double rb = params ? params[0] : 100.0;
            for (int i = 0; i < n - 1; ++i) {
                double a = 1.0 - p[i];
                double b = SMCMC_FMA(-p[i], p[i], p[i + 1]);
                double t = SMCMC_FMA(rb * b, b, a * a);
                logl -= t;
            }
            return logl;
        }
    }
}

/* ---- the random stream standing in for gRandom --------------------------- */
typedef struct {
    uint64_t seed;
    uint32_t chain;
    uint64_t step;       /* counter the draws are keyed on (fTotalSteps) */
    uint32_t cached_block;
    int has_block;
    smcmc_u32x4 block;
    uint32_t stream_id;  /* SMCMC_STREAM_STEP (0) unless a restatement says otherwise */
} oracle_stream;

static uint32_t oracle_stream_word(oracle_stream* s, uint32_t w) {
    uint32_t b = w >> 2;
    if (!s->has_block || s->cached_block != b) {
        s->block = smcmc_draw_block(s->seed, s->chain, s->step, b, s->stream_id);
        s->cached_block = b;
        s->has_block = 1;
    }
    return s->block.v[w & 3u];
}
static void oracle_stream_set_step(oracle_stream* s, uint64_t step) {
    s->step = step;
    s->has_block = 0;
}
/* stands in for gRandom->Gaus(0,1) at TSimpleMCMC.H:719 for dimension i */
static double oracle_stream_normal(oracle_stream* s, int i) {
    uint32_t p = (uint32_t)i >> 1;
    double n0, n1;
    smcmc_normal_pair(oracle_stream_word(s, 2u * p), oracle_stream_word(s, 2u * p + 1u), &n0, &n1);
    return (i & 1) ? n1 : n0;
}
/* stands in for gRandom->Uniform() (TSimpleMCMC.H:455) / Uniform(a,b) (:713) */
static double oracle_stream_uniform(oracle_stream* s, uint32_t word) {
    return smcmc_u01(oracle_stream_word(s, word));
}

/* ---- TProposeAdaptiveStep state (TSimpleMCMC.H:1833-1976) ---------------- */
typedef struct {
    int dim;
    double* last_point;       /* fLastPoint */
    double last_value;        /* fLastValue */
    double* central;          /* fCentralPoint */
    double* central_change;   /* fCentralPointChange */
    double central_trials;    /* fCentralPointTrials */
    double* cov;              /* fCurrentCov, dim x dim */
    double cov_trials;        /* fCovarianceTrials */
    double cov_deweight;      /* fCovarianceDeweight */
    int cov_frozen;           /* fCovarianceFrozen */
    double cov_window;        /* fCovarianceWindow */
    double* decomp;           /* fDecomposition, dim x dim */
    int decomp_full;          /* 1 after the eigen fallback filled a full matrix */
    int* ptype;               /* fProposalType[i].type */
    double* pparam1;
    double* pparam2;
    int ncorr;                /* fCorrelations */
    int* corr_d1;
    int* corr_d2;
    double* corr_c;
    double max_corr;          /* fMaxCorrelation */
    int trials;               /* fTrials */
    int successes;            /* fSuccesses */
    int next_update;          /* fNextUpdate */
    double acceptance;        /* fAcceptance */
    double acceptance_trials; /* fAcceptanceTrials */
    double acceptance_deweight;
    double acceptance_window; /* fAcceptanceWindow */
    double rigidity;          /* fAcceptanceRigidity */
    double target;            /* fTargetAcceptance */
    double sigma;             /* fSigma */
    double sigma_trace;       /* fSigmaTrace */
    int state_initialized;    /* fStateInitialized */
    int scan_dim;             /* fScanDimension */
    double* forced;           /* fForcedStep */
    int has_forced;
    int update_count;         /* diagnostics: number of UpdateProposal calls */
    int last_update_path;     /* 0 chol, 1 conditioned chol, 2 eigen, 3 emergency, 4 reset */
    int failed;               /* set instead of throwing */
    double last_sigma_scale;  /* sqrt(old trace / new trace) of the latest UpdateProposal (ensemble bookkeeping) */
} oracle_proposal;

/* constructor TSimpleMCMC.H:642-655 */
static void oracle_proposal_init(oracle_proposal* p) {
    memset(p, 0, sizeof(*p));
    p->last_value = 0.0;
    p->central_trials = 0.0;
    p->cov_trials = 0.0;
    p->cov_deweight = 0.5;
    p->cov_frozen = 0;
    p->cov_window = -1;
    p->trials = 0;
    p->successes = 0;
    p->next_update = -1;
    p->acceptance = 0.0;
    p->acceptance_trials = 0;
    p->acceptance_deweight = 0.5;
    p->acceptance_window = -1;
    p->rigidity = 2.0;
    p->target = -1;
    p->sigma = 0.0;
    p->state_initialized = 0;
    p->scan_dim = -1;
    p->max_corr = DBL_EPSILON;
    p->max_corr = 1.0 - sqrt(p->max_corr);
}

/* SetDim TSimpleMCMC.H:786-795 */
static void oracle_proposal_set_dim(oracle_proposal* p, int dim) {
    if (p->dim > 0) return;
    p->dim = dim;
    size_t n = (size_t)dim;
    p->last_point = (double*)calloc(n, sizeof(double));
    p->central = (double*)calloc(n, sizeof(double));
    p->central_change = (double*)calloc(n, sizeof(double));
    p->cov = (double*)calloc(n * n, sizeof(double));
    p->decomp = (double*)calloc(n * n, sizeof(double));
    p->ptype = (int*)calloc(n, sizeof(int));
    p->pparam1 = (double*)calloc(n, sizeof(double));
    p->pparam2 = (double*)calloc(n, sizeof(double));
    p->forced = (double*)calloc(n, sizeof(double));
}

static void oracle_proposal_free(oracle_proposal* p) {
    free(p->last_point); free(p->central); free(p->central_change); free(p->cov);
    free(p->decomp); free(p->ptype); free(p->pparam1); free(p->pparam2); free(p->forced);
    free(p->corr_d1); free(p->corr_d2); free(p->corr_c);
}

/* SetCorrelation TSimpleMCMC.H:883-904 */
static void oracle_proposal_set_correlation(oracle_proposal* p, int d1, int d2, double c) {
    if (d1 == d2) return;
    if (c < -p->max_corr) c = -p->max_corr;
    if (c > p->max_corr) c = p->max_corr;
    p->corr_d1 = (int*)realloc(p->corr_d1, sizeof(int) * (size_t)(p->ncorr + 1));
    p->corr_d2 = (int*)realloc(p->corr_d2, sizeof(int) * (size_t)(p->ncorr + 1));
    p->corr_c = (double*)realloc(p->corr_c, sizeof(double) * (size_t)(p->ncorr + 1));
    p->corr_d1[p->ncorr] = d1; p->corr_d2[p->ncorr] = d2; p->corr_c[p->ncorr] = c;
    p->ncorr++;
}

/* GetCovarianceTrace TSimpleMCMC.H:961-967 */
static double oracle_proposal_trace(const oracle_proposal* p) {
    double trace = 0.0;
    for (int i = 0; i < p->dim; ++i) trace += p->cov[i * p->dim + i];
    return trace;
}

static void oracle_proposal_reset(oracle_proposal* p);

/* The part of UpdateProposal after the bookkeeping: Cholesky and the fallback
 * ladder, TSimpleMCMC.H:1097-1389.  Returns the path taken. */
static int oracle_decompose_ladder(oracle_proposal* p, int from_reset) {
    const int n = p->dim;
    double* C = p->cov;
    double min_var = DBL_EPSILON;                                   /* :1098 */

    if (oracle_cholesky_upper(n, C, p->decomp)) { p->decomp_full = 0; return 0; }  /* :1103-1119 */

    /* variance conditioning :1134-1183 */
    for (int i = 0; i < n; ++i) {
        double expected = 1.0;
        if (p->ptype[i] == 0) {
            expected = 1.0;
            if (p->pparam1[i] > 0) expected = p->pparam1[i];
        } else if (p->ptype[i] == 1) {
            expected = p->pparam2[i];
            expected -= p->pparam1[i];
            expected = expected * expected / 12.0;
        } else {
            p->failed = 1; return -1;                               /* :1149-1154 throws */
        }
        if (!isfinite(C[i * n + i])) C[i * n + i] = expected;
        if (C[i * n + i] < 0.0) C[i * n + i] = min_var * expected;
        if (C[i * n + i] < min_var * expected) C[i * n + i] = min_var * expected;
        if (C[i * n + i] < min_var) C[i * n + i] = min_var;
    }
    /* correlation conditioning :1187-1217 */
    for (int i = 0; i < n; ++i) {
        for (int j = i + 1; j < n; ++j) {
            double corr = C[i * n + j];
            corr /= sqrt(C[i * n + i]);
            corr /= sqrt(C[j * n + j]);
            if (!isfinite(corr)) corr = 0.0;
            if (fabs(corr) > p->max_corr) {
                if (corr > 0.0) corr = p->max_corr; else corr = -p->max_corr;
            }
            C[i * n + j] = corr;
            C[i * n + j] *= sqrt(C[i * n + i]);
            C[i * n + j] *= sqrt(C[j * n + j]);
            C[j * n + i] = C[i * n + j];
        }
    }
    if (oracle_cholesky_upper(n, C, p->decomp)) { p->decomp_full = 0; return 1; }  /* :1220-1239 */

    /* eigen decomposition :1252-1321 */
    {
        double* vec = (double*)malloc(sizeof(double) * (size_t)n * (size_t)n);
        double* val = (double*)malloc(sizeof(double) * (size_t)n);
        oracle_sym_eigen(n, C, vec, val);
        double eigen_sum = 0.0;
        for (int i = 0; i < n; ++i) {
            if (val[i] < 0.0) continue;
            eigen_sum += val[i];
        }
        double min_axis = 1.0 - p->max_corr;
        if (min_axis < min_var) min_axis = min_var;
        min_axis = min_axis * val[0];
        for (int i = 0; i < n; ++i) {
            double rms = (min_axis > val[i]) ? min_axis : val[i];   /* std::max(minAxis, ev) */
            rms = sqrt(rms);
            for (int j = 0; j < n; ++j) p->decomp[i * n + j] = rms * vec[j * n + i];
        }
        free(vec); free(val);
        if (eigen_sum > 1E-6) { p->decomp_full = 1; return 2; }     /* :1321 */
    }

    /* emergency shrink :1335-1377 */
    {
        double step = DBL_EPSILON;
        for (int i = 0; i < n; ++i) step = (step > C[i * n + i]) ? step : C[i * n + i];
        step *= 1E-4;
        double dec = 1.0;
        for (int trial = 0; trial < 10; ++trial) {
            dec *= 0.84;
            for (int i = 0; i < n; ++i) {
                C[i * n + i] += step;
                for (int j = i + 1; j < n; ++j) C[i * n + j] = C[j * n + i] = dec * C[i * n + j];
            }
            if (oracle_cholesky_upper(n, C, p->decomp)) { p->decomp_full = 0; return 3; }
        }
    }
    if (from_reset) { p->failed = 1; return -1; }                   /* :1383-1386 throws */
    oracle_proposal_reset(p);                                       /* :1389 */
    return 4;
}

/* UpdateProposal TSimpleMCMC.H:1009-1390 */
static void oracle_proposal_update(oracle_proposal* p, int from_reset) {
    const int n = p->dim;
    p->update_count++;
    double current_trace = oracle_proposal_trace(p);                /* :1024 */
    if (current_trace <= 0) { p->failed = 1; return; }              /* :1025-1028 throws */

    const double sigma_scale = sqrt(p->sigma_trace / current_trace);
    p->sigma = p->sigma * sigma_scale;                              /* :1042 */
    p->sigma_trace = current_trace;                                 /* :1043 */

    double max_up = (double)n * (double)n;                          /* :1050 (size_t product) */
    double up = 0.5 * p->successes;                                 /* :1051 */
    p->next_update = (int)(p->acceptance_window + max_up - max_up / (up + 1.0));  /* :1052 */

    if (p->cov_deweight > 0.0) {                                    /* :1056-1067 */
        if (p->cov_deweight > 1.0) p->cov_deweight = 1.0;
        double w = 1.0 - p->cov_deweight;
        p->cov_trials = fmax(1.0, w * p->cov_trials);
        p->cov_trials = fmin(p->cov_trials, w * p->cov_window);
        p->central_trials = fmax(1.0, w * p->central_trials);
        p->central_trials = fmin(p->central_trials, w * p->cov_window);
    }
    if (p->acceptance_deweight > 0.0) {                             /* :1081-1086 */
        if (p->acceptance_deweight > 1.0) p->acceptance_deweight = 1.0;
        double w = 1.0 - p->acceptance_deweight;
        p->acceptance_trials = fmax(1.0, w * p->acceptance_trials);
        p->acceptance_trials = fmin(p->acceptance_trials, w * p->acceptance_window);
    }
    p->last_update_path = oracle_decompose_ladder(p, from_reset);
    p->last_sigma_scale = sigma_scale;   /* after the ladder: a reset on its last rung runs this function again */
}

/* ResetProposal TSimpleMCMC.H:1396-1494 */
static void oracle_proposal_reset(oracle_proposal* p) {
    const int n = p->dim;
    p->trials = 0;
    p->successes = 0;
    if (p->sigma < 0.01 * sqrt(1.0 / n)) p->sigma = sqrt(1.0 / n);  /* :1408-1410 */
    for (int i = 0; i < n; ++i) {                                   /* :1415-1443 */
        for (int j = i; j < n; ++j) {
            if (i == j && p->ptype[i] == 0 && p->pparam1[i] > 0) {
                p->cov[i * n + i] = p->pparam1[i];
            } else if (i == j && p->ptype[i] == 1) {
                double delta = p->pparam1[i];
                delta -= p->pparam2[i];
                p->cov[i * n + i] = delta * delta / 12.0;
            } else if (i == j) {
                p->cov[i * n + i] = 1.0;
            } else {
                p->cov[i * n + j] = p->cov[j * n + i] = 0.0;
            }
        }
    }
    for (int c = 0; c < p->ncorr; ++c) {                            /* :1445-1457 */
        int d1 = p->corr_d1[c], d2 = p->corr_d2[c];
        if (d1 == d2) continue;
        double v1 = p->cov[d1 * n + d1];
        double v2 = p->cov[d2 * n + d2];
        p->cov[d1 * n + d2] = p->cov[d2 * n + d1] = p->corr_c[c] * sqrt(v1) * sqrt(v2);
    }
    p->sigma_trace = oracle_proposal_trace(p);                      /* :1460 */
    int min_window = 100 + 4 * n;                                   /* :1468 */
    if (p->cov_window < min_window) {
        p->cov_window = n;
        p->cov_window *= n;
        p->cov_window *= n;
        p->cov_window += min_window;
        double r = DBL_EPSILON;
        p->cov_window = fmin(p->cov_window, sqrt(1.0 / r));
    }
    if (p->target < 0.0) { p->failed = 1; return; }                 /* :1478-1480 throws */
    p->acceptance = p->target;                                      /* :1481 */
    p->acceptance_trials = fmin(10.0, 0.5 * p->acceptance_window);  /* :1482 */
    memcpy(p->central, p->last_point, sizeof(double) * (size_t)n);  /* :1484-1485 */
    memset(p->central_change, 0, sizeof(double) * (size_t)n);
    p->central_trials = fmax(p->central_trials, 1.0);               /* :1491 */
    oracle_proposal_update(p, 1);                                   /* :1493 */
}

/* InitializeState TSimpleMCMC.H:1679-1714 */
static void oracle_proposal_initialize(oracle_proposal* p, int dim, const double* current, double value) {
    if (p->state_initialized) return;
    p->state_initialized = 1;
    if (p->dim < 1) oracle_proposal_set_dim(p, dim);
    p->last_value = value;
    memcpy(p->last_point, current, sizeof(double) * (size_t)p->dim);
    if (p->acceptance_window < 0) {
        p->acceptance_window = pow(1.0 * p->dim, 1.5) + 1000;       /* :1693-1695 */
    }
    p->next_update = (int)p->acceptance_window;                     /* :1697 */
    if (p->target < 1E-4) {
        if (p->dim > 4) p->target = 0.234; else p->target = 0.44;   /* :1709-1710 */
    }
    oracle_proposal_reset(p);
}

/* The per-chain scalar part of UpdateState, TSimpleMCMC.H:1723-1776.  Split out
 * because the many-chain engine runs exactly this per lane.  Returns `accepted`. */
static int oracle_update_scalars(int* trials, int* successes, double* acceptance,
                                 double* acceptance_trials, double acceptance_window,
                                 double* rigidity, double target, double* sigma,
                                 int moved) {
    ++(*trials);                                                    /* :1723 */
    int accepted = moved;                                           /* :1727-1728 */
    if (accepted) ++(*successes);                                   /* :1731 */
    *acceptance *= *acceptance_trials;                              /* :1734 */
    if (accepted) *acceptance = *acceptance + 1.0;
    *acceptance /= *acceptance_trials + 1.0;
    *acceptance_trials = fmin(acceptance_window, *acceptance_trials + 1.0);   /* :1737 */
    if (*rigidity < 500.0 && *rigidity > 0.0) {                     /* :1745-1762 */
        double asig = target * (1.0 - target);
        asig = sqrt(asig / acceptance_window);
        if (fabs(*acceptance - target) < asig) {
            *rigidity += 0.5 * *rigidity / acceptance_window;
            *rigidity = fmin(200.0, *rigidity);
        }
        if (fabs(*acceptance - target) > 4.0 * asig) {
            *rigidity -= 1.618 * 0.5 * *rigidity / acceptance_window;
            *rigidity = fmax(2.0, *rigidity);
        }
    }
    if (*rigidity > 0 && *rigidity < 100.0) {                       /* :1771-1776 */
        *sigma *= smcmc_pow_small(*acceptance / target,
                                  fmin(1.0 / 500.0, 1.0 / (*rigidity * acceptance_window)));
    }
    return accepted;
}

/* UpdateState TSimpleMCMC.H:1721-1831 */
static void oracle_proposal_update_state(oracle_proposal* p, const double* current, double value) {
    const int n = p->dim;
    oracle_proposal_initialize(p, n, current, value);               /* :1722 */
    int moved = (value != p->last_value || current[0] != p->last_point[0]);
    int accepted = oracle_update_scalars(&p->trials, &p->successes, &p->acceptance,
                                         &p->acceptance_trials, p->acceptance_window,
                                         &p->rigidity, p->target, &p->sigma, moved);
    for (int i = 0; i < n; ++i) {                                   /* :1780-1786 */
        p->central_change[i] = p->central[i];
        p->central[i] *= p->central_trials;
        p->central[i] += current[i];
        p->central[i] /= p->central_trials + 1;
        p->central_change[i] = p->central[i] - p->central_change[i];
    }
    p->central_trials = fmin(p->cov_window, p->central_trials + 1.0);   /* :1787-1788 */
    if (!p->cov_frozen) {                                           /* :1795-1820 */
        for (int i = 0; i < n; ++i) {
            for (int j = 0; j < i + 1; ++j) {
                double v = p->cov[i * n + j];
                double r = (current[i] - p->central[i]) * (current[j] - p->central[j]);
                v *= p->cov_trials;
                v += r;
                v /= p->cov_trials + 1.0;
                if (i == j) p->cov[i * n + j] = v;
                else p->cov[i * n + j] = p->cov[j * n + i] = v;
            }
        }
        p->cov_trials = fmin(p->cov_window, p->cov_trials + 1.0);
    }
    if (accepted && (--p->next_update) < 1) oracle_proposal_update(p, 0);   /* :1824-1826 */
    p->last_value = value;                                          /* :1829-1830 */
    memcpy(p->last_point, current, sizeof(double) * (size_t)n);
}

/* operator() TSimpleMCMC.H:659-725.  `stream` must already be keyed on the
 * step this call belongs to. */
static void oracle_proposal_propose(oracle_proposal* p, oracle_stream* stream,
                                    double* proposal, const double* current, double value) {
    const int n = p->dim;
    if (p->has_forced) {                                            /* :671-678 */
        memcpy(proposal, p->forced, sizeof(double) * (size_t)n);
        p->has_forced = 0;
        return;
    }
    int scan = p->scan_dim;                                         /* :685-704 */
    if (scan >= 0 && scan < n) {
        memcpy(proposal, current, sizeof(double) * (size_t)n);
        if (p->ptype[scan] == 1) {
            double u = oracle_stream_uniform(stream, (uint32_t)scan);
            proposal[scan] = p->pparam1[scan] + (p->pparam2[scan] - p->pparam1[scan]) * u;
            return;
        }
        double sigma = 1.0;
        if (p->pparam1[scan] > 0) sigma = sqrt(p->pparam1[scan]);
        proposal[scan] = p->central[scan] + sigma * oracle_stream_normal(stream, scan);
        return;
    }
    oracle_proposal_update_state(p, current, value);                /* :706 */
    memcpy(proposal, current, sizeof(double) * (size_t)n);          /* :709 */
    for (int i = 0; i < n; ++i) {                                   /* :710-724 */
        if (p->ptype[i] == 1) {
            double u = oracle_stream_uniform(stream, (uint32_t)i);
            proposal[i] = p->pparam1[i] + (p->pparam2[i] - p->pparam1[i]) * u;
            continue;
        }
        double r = oracle_stream_normal(stream, i);
        for (int j = 0; j < n; ++j) {
            if (p->ptype[j] == 1) continue;
            proposal[j] += p->sigma * r * p->decomp[i * n + j];
        }
    }
}

/* ---- TSimpleMCMC engine state (TSimpleMCMC.H:543-589) -------------------- */
typedef struct {
    int dim;
    int like_kind;
    double* like_params;      /* QUADFORM: Error matrix dim x dim; ROSENBROCK: {b} */
    oracle_proposal prop;
    oracle_stream stream;
    int total_steps;          /* fTotalSteps */
    int like_count;           /* fLogLikelihoodCount */
    double* accepted;         /* fAccepted */
    double* proposed;         /* fProposed */
    double* trial_step;       /* fTrialStep */
    double accepted_logl;     /* fAcceptedLogLikelihood */
    double proposed_logl;     /* fProposedLogLikelihood */
    double step_rms;          /* fStepRMS */
    int step_rms_trials;      /* fStepRMSTrials */
    int step_rms_window;      /* fStepRMSWindow */
} oracle_chain;

static double oracle_chain_like(oracle_chain* c, const double* p) {   /* :538-541 */
    ++c->like_count;
    return oracle_like(c->like_kind, c->dim, p, c->like_params);
}

/* Start TSimpleMCMC.H:246-276 (save handled by the caller) */
static int oracle_chain_start(oracle_chain* c, const double* start) {
    size_t n = (size_t)c->dim;
    memcpy(c->proposed, start, sizeof(double) * n);
    memcpy(c->accepted, start, sizeof(double) * n);
    memcpy(c->trial_step, start, sizeof(double) * n);
    c->proposed_logl = oracle_chain_like(c, c->proposed);
    if (!isfinite(c->proposed_logl) || (c->proposed_logl < -0.999999E+10)) return 0;   /* :265-268 */
    c->accepted_logl = c->proposed_logl;
    oracle_proposal_initialize(&c->prop, c->dim, c->accepted, c->accepted_logl);       /* :272 */
    return 1;
}

/* RestoreState TSimpleMCMC.H:1501-1612 with the saved fields of one tree entry
 * (the branches of :1616-1626); cov_packed is the lower triangle, row major
 * (:1585-1598). */
static void oracle_proposal_restore(oracle_proposal* p, const double* current, double value,
                                    int trials, int successes, int next_update, double acceptance,
                                    double acceptance_trials, double sigma, const double* central,
                                    double central_trials, const double* cov_packed, double cov_trials) {
    const int n = p->dim;
    p->state_initialized = 1;                                       /* :1503 */
    p->last_value = value;                                          /* :1514 */
    memcpy(p->last_point, current, sizeof(double) * (size_t)n);
    p->trials = trials;                                             /* :1563-1570 */
    p->successes = successes;
    p->next_update = next_update;
    p->acceptance = acceptance;
    p->acceptance_trials = acceptance_trials;
    p->sigma = sigma;
    memcpy(p->central, central, sizeof(double) * (size_t)n);
    p->central_trials = central_trials;
    const double* cov = cov_packed;                                 /* :1574-1586 */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i + 1; ++j) {
            p->cov[i * n + j] = p->cov[j * n + i] = *cov;
            ++cov;
        }
    p->sigma_trace = oracle_proposal_trace(p);                      /* :1587 */
    p->cov_trials = cov_trials;                                     /* :1588 */
    oracle_proposal_update(p, 0);                                   /* :1612 */
}

/* Restore TSimpleMCMC.H:282-352 from the last entry of a tree (randomize = false):
 * the saved likelihood is kept unless the recomputed one differs by more than 1E-4. */
static void oracle_chain_restore(oracle_chain* c, const double* accepted, double saved_logl,
                                 int total_steps, double step_rms,
                                 int trials, int successes, int next_update, double acceptance,
                                 double acceptance_trials, double sigma, const double* central,
                                 double central_trials, const double* cov_packed, double cov_trials) {
    size_t n = (size_t)c->dim;
    c->total_steps = total_steps;                                   /* :320-323 */
    c->accepted_logl = saved_logl;
    c->step_rms = step_rms;
    memcpy(c->accepted, accepted, sizeof(double) * n);
    memcpy(c->proposed, accepted, sizeof(double) * n);
    memcpy(c->trial_step, accepted, sizeof(double) * n);
    c->proposed_logl = oracle_chain_like(c, c->proposed);           /* :335 */
    double delta = c->proposed_logl - c->accepted_logl;
    if (fabs(delta) > 1E-4) c->accepted_logl = c->proposed_logl;    /* :337-345 */
    oracle_proposal_restore(&c->prop, c->accepted, c->accepted_logl, trials, successes, next_update,
                            acceptance, acceptance_trials, sigma, central, central_trials,
                            cov_packed, cov_trials);                /* :351 */
}

/* Step TSimpleMCMC.H:370-496.  Returns 1 when a new point was accepted. */
static int oracle_chain_step(oracle_chain* c, int save, int metropolis) {
    const int n = c->dim;
    ++c->total_steps;                                               /* :376 */
    oracle_stream_set_step(&c->stream, (uint64_t)(uint32_t)c->total_steps);
    oracle_proposal_propose(&c->prop, &c->stream, c->proposed, c->accepted, c->accepted_logl);   /* :378 */
    if (save || c->step_rms_window > 0) {                           /* :391-406 */
        double sqr = 0.0;
        for (int i = 0; i < n; ++i) {
            c->trial_step[i] = c->proposed[i] - c->accepted[i];
            sqr += c->trial_step[i] * c->trial_step[i];
        }
        if (c->step_rms_window > 0) {
            double ms = c->step_rms * c->step_rms;
            ms *= c->step_rms_trials;
            ms += sqr;
            ms /= c->step_rms_trials + 1.0;
            c->step_rms_trials = (c->step_rms_window < c->step_rms_trials + 1)
                                     ? c->step_rms_window : c->step_rms_trials + 1;
            c->step_rms = sqrt(ms);
        }
    }
    c->proposed_logl = oracle_chain_like(c, c->proposed);           /* :410 */
    if (metropolis == 2) {                                          /* :414-426 */
        memcpy(c->accepted, c->proposed, sizeof(double) * (size_t)n);
        c->accepted_logl = c->proposed_logl;
        return 1;
    }
    if (!isfinite(c->proposed_logl) || (c->proposed_logl < -0.999999E+30)) return 0;   /* :432-436 */
    double delta = c->proposed_logl - c->accepted_logl;             /* :441 */
    if (delta < 0.0) {
        if (metropolis == 1) return 0;                              /* :448 */
        double trial = smcmc_log(oracle_stream_uniform(&c->stream, smcmc_accept_word((uint32_t)n)));   /* :455 */
        if (delta < trial) return 0;                                /* :456-462 */
    }
    c->accepted_logl = c->proposed_logl;                            /* :484 */
    memcpy(c->accepted, c->proposed, sizeof(double) * (size_t)n);   /* :487 */
    return 1;
}

#endif
