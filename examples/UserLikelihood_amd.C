// UserLikelihood_amd.C -- a user's own likelihood behind sMCMC::TSimpleMCMC on the MI355X engine.
//
// The reference's TASymLogLikelihood (TAsymLogLikelihood.H:9-31) keeps its host operator() (handy for checks)
// and says how it reaches the device: kDeviceLikelihood = SMCMC_LIKE_USER and its data members as
// DeviceParams(); the device form of operator() is examples/user_likelihood_asym.hip.h, compiled into
// libsmcmc_amd_user.so by `python root-simple-mcmc_amd/build.py --user-likelihood examples/user_likelihood_asym.hip.h`.
//
//   g++ -std=c++17 -O2 -Iinclude examples/UserLikelihood_amd.C -Lroot-simple-mcmc_amd/lib -lsmcmc_amd_user
//       -Wl,-rpath,$PWD/root-simple-mcmc_amd/lib -Wl,-rpath,/opt/rocm/lib -o user_amd.exe
#include <cstdlib>
#include <iostream>

#include "TSimpleMCMC_amd.H"

class TASymLogLikelihood {
public:
    static constexpr int kDeviceLikelihood = SMCMC_LIKE_USER;
    std::size_t GetDim() const { return 10; }
    const double positiveSlope = -1.0;
    const double negativeSlope = 100.0;
    double operator()(const sMCMC::Vector& point) const {           // TAsymLogLikelihood.H:20-31
        double logLikelihood = 0.0;
        for (std::size_t i = 0; i < GetDim(); ++i) {
            double a = point[i];
            if (a < 0.0) a *= negativeSlope;
            else a *= positiveSlope;
            logLikelihood += a;
        }
        return logLikelihood;
    }
    void Init() {}
    sMCMC::Vector DeviceParams() const { return {positiveSlope, negativeSlope}; }
};

int main(int argc, char** argv) {
    const int chains = (argc > 1) ? std::atoi(argv[1]) : 1024;
    try {
        sMCMC::TreeType tree("UserMCMC", "Tree of accepted points");
        sMCMC::TSimpleMCMC<TASymLogLikelihood> mcmc(&tree);
        TASymLogLikelihood& like = mcmc.GetLogLikelihood();
        like.Init();
        mcmc.SetChains(chains);
        mcmc.GetProposeStep().SetDim(like.GetDim());
        sMCMC::Vector p(like.GetDim(), 0.5);
        if (!mcmc.Start(p, true)) return 1;
        if (chains == 1) {
            // the reference's own use: one chain adapting alone, Step() one call at a time (served from recorded launches
            // of the one-chain-per-wavefront kernel, which evaluates the user's function on the whole point)
            for (int i = 0; i < 4000; ++i) mcmc.Step(i % 100 == 0);
            std::cout << "Step() runs ahead: " << (mcmc.GetRunAhead() ? 1 : 0) << std::endl;
        } else {
            for (int w = 0; w < 20; ++w) {
                mcmc.StepMany(200);
                mcmc.GetProposeStep().SyncPooledCovariance();
                mcmc.SaveStep(false);
            }
        }
        // the device's likelihood of chain 0's point against the host functor
        const double host = like(mcmc.GetAccepted());
        std::cout << "chain 0: device logL " << mcmc.GetAcceptedLogLikelihood() << " host logL " << host
                  << (host == mcmc.GetAcceptedLogLikelihood() ? " (identical)" : " (DIFFERENT)") << std::endl;
        sMCMC::Vector all = mcmc.GetAcceptedAll();
        double mean = 0.0;
        for (double v : all) mean += v;
        std::cout << "ensemble mean " << mean / all.size() << " (posterior mean 0.99)" << std::endl;
        return host == mcmc.GetAcceptedLogLikelihood() ? 0 : 3;
    } catch (const std::exception& e) {
        std::cerr << "UserLikelihood_amd: " << e.what() << std::endl;
        return 2;
    }
}
