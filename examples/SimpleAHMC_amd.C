// SimpleAHMC_amd.C -- the reference's approximate-gradient HMC example (SimpleAHMC.C:9-124) on the MI355X engine: the
// header-form TDummyLogLikelihood, a start point uniform in [-1, 1], then the macro's three phases --
//   1. burn-in as a guided random walk: SetAlpha(0.8), SetMeanEpsilon(-0.1), SetLeapFrog(0), Step(false, 5): no gradient,
//      the steps only feed the running covariance;
//   2. second burn-in on the covariant gradient: SetAlpha(0), SetMeanEpsilon(-0.05), SetLeapFrog(5), Step(false, 2);
//   3. the run: SetAlpha(0.75), Step(true, 2).
// Differences: dimension and chain count are arguments (the running covariance is pooled over the chains); the start
// point comes from the counter-based stream instead of gRandom.
//
//   g++ -std=c++17 -O2 -Iinclude examples/SimpleAHMC_amd.C -Lroot-simple-mcmc_amd/lib -lsmcmc_amd
//       -Wl,-rpath,$PWD/root-simple-mcmc_amd/lib -Wl,-rpath,/opt/rocm/lib -o ahmc_amd.exe
//   ./ahmc_amd.exe [trials [output.csv [dim [chains]]]]
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

#include "TSimpleHMC_amd.H"
#include "smcmc_detmath.h"

int SimpleAHMC(int trials, const char* outputName, int dim, int chains) {
    std::cout << "Simple AHMC Loaded (MI355X engine) D=" << dim << " chains=" << chains << std::endl;
    sMCMC::TreeType tree("SimpleAHMC", "Tree of accepted points");                   // SimpleAHMC.C:19
    sMCMC::TSimpleHMC<sMCMC::TDummyLogLikelihood> hmc(&tree);                         // :26
    sMCMC::TDummyLogLikelihood& like = hmc.GetLogLikelihood();
    like.SetDim(dim);
    like.Init();                                                                      // :34
    hmc.SetChains(chains);

    sMCMC::Vector p(like.GetDim());                                                   // :42-43
    for (std::size_t i = 0; i < p.size(); ++i) {
        const smcmc_u32x4 blk = smcmc_draw_block(20240607ull, 0u, 0u, (uint32_t)(i >> 2), SMCMC_STREAM_START);
        p[i] = -1.0 + 2.0 * smcmc_u01(blk.v[i & 3u]);
    }
    hmc.Start(p, true);                                                               // :45

    const int verbose = 500;
    std::cout << "Start burn-in" << std::endl;
    int burnin = 500 + 2 * (int)(p.size() * p.size());                                // :51
    hmc.SetAlpha(0.8);                                                                // :52-54
    hmc.SetMeanEpsilon(-0.1);
    hmc.SetLeapFrog(0);
    for (int i = 0; i < burnin; ++i) {
        if (i % verbose == 0)
            std::cout << "Burn-in: " << i << " Calls: " << hmc.GetPotentialCount() << " Gradients: "
                      << hmc.GetGradientCount() << " Acceptance: " << hmc.GetAcceptanceRate() << std::endl;
        hmc.Step(false, 5);                                                           // :63
    }

    std::cout << "Second burn-in" << std::endl;
    hmc.SetAlpha(0.0);                                                                // :68-70
    hmc.SetMeanEpsilon(-0.05);
    hmc.SetLeapFrog(5);
    burnin = 500 + 2 * (int)(p.size() * p.size());
    for (int i = 0; i < burnin; ++i) {
        if (i % verbose == 0)
            std::cout << "Burn-in: " << i << " Calls: " << hmc.GetPotentialCount() << " Gradients: "
                      << hmc.GetGradientCount() << " Acceptance: " << hmc.GetAcceptanceRate() << std::endl;
        hmc.Step(false, 2);                                                           // :81
    }

    std::cout << "Run chain" << std::endl;
    hmc.SetAlpha(0.75);                                                               // :86-88
    hmc.SetMeanEpsilon(-0.05);
    hmc.SetLeapFrog(5);
    for (int i = 0; i < trials; ++i) {
        if (i % verbose == 0)
            std::cout << "Trials: " << i << " Calls: " << hmc.GetPotentialCount() << " Gradients: "
                      << hmc.GetGradientCount() << " Acceptance: " << hmc.GetAcceptanceRate() << std::endl;
        hmc.Step(true, 2);                                                            // :98
    }
    std::cout << "Finished " << trials << " trials with " << hmc.GetPotentialCount() << " calls and "
              << hmc.GetGradientCount() << " gradients " << " Accepted: " << hmc.GetAcceptanceRate() << std::endl;
    tree.Write();
#ifndef SMCMC_HAVE_ROOT
    tree.WriteCsv(outputName);
    std::cout << "wrote " << tree.GetEntries() << " entries to " << outputName << std::endl;
#endif
    return 0;
}

int main(int argc, char** argv) {
    int trials = 10000, dim = 100, chains = 64;                                       // :110
    std::string outputName("SimpleAHMC_amd.csv");
    if (argc > 1) { std::istringstream in(argv[1]); in >> trials; }
    if (argc > 2) outputName = argv[2];
    if (argc > 3) { std::istringstream in(argv[3]); in >> dim; }
    if (argc > 4) { std::istringstream in(argv[4]); in >> chains; }
    try {
        return SimpleAHMC(trials, outputName.c_str(), dim, chains);
    } catch (const std::exception& e) {
        std::cerr << "SimpleAHMC_amd: " << e.what() << std::endl;
        return 2;
    }
}
