// user_likelihood_asym.hip.h -- a user likelihood for the MI355X engine: the reference's
// TASymLogLikelihood (TAsymLogLikelihood.H:20-31: slope -1 above zero, +100 below, summed over the
// dimensions) written as the device function the engine compiles in.
//
//   python root-simple-mcmc_amd/build.py --user-likelihood examples/user_likelihood_asym.hip.h
//
// gives lib/libsmcmc_amd_user.so, in which smcmc_create(..., SMCMC_LIKE_USER, ...) runs this function
// inside the step kernel.  What the reference's `double operator()(const sMCMC::Vector& point)` becomes:
// the point arrives as a register array p[0..D) (entries past D are zero), the functor's data members as
// the parameter array handed to smcmc_set_likelihood_params, and the body keeps its operation order.
#pragma once

template <int DP>
__device__ __forceinline__ double smcmc_user_loglike(const double (&p)[DP], smcmc::cptr_f64 params, int D) {
    const double positiveSlope = params[0];   // -1.0  (TAsymLogLikelihood.H:16)
    const double negativeSlope = params[1];   // 100.0 (:17)
    double logLikelihood = 0.0;
#pragma unroll
    for (int i = 0; i < DP; ++i) {
        if (i < D) {
            double a = p[i];
            if (a < 0.0) a *= negativeSlope;
            else a *= positiveSlope;
            logLikelihood += a;
        }
    }
    return logLikelihood;
}

// The same function for 63 < dim <= 512, where a point no longer fits one lane's registers: p[i] reads coordinate i of
// the chain's point from device memory.  Defining SMCMC_USER_LIKELIHOOD_ANY_DIM tells the build that this form exists.
#define SMCMC_USER_LIKELIHOOD_ANY_DIM 1

template <class Point>
__device__ __forceinline__ double smcmc_user_loglike_at(const Point& p, const double* params, int D) {
    const double positiveSlope = params[0];
    const double negativeSlope = params[1];
    double logLikelihood = 0.0;
    for (int i = 0; i < D; ++i) {
        double a = p[i];
        if (a < 0.0) a *= negativeSlope;
        else a *= positiveSlope;
        logLikelihood += a;
    }
    return logLikelihood;
}
