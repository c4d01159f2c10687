// SimpleHMC_amd.C -- the reference's HMC example (SimpleHMC.C:15-91) on the MI355X engine: the
// header-form TDummyLogLikelihood (quadratic form, Error from Init()) with its own gradient, start at
// p = 1, a light burn-in of 100 + D unsaved steps, then `trials` saved steps, same tree schema.
// Differences: dimension, chain count and arithmetic order are arguments; `tune` = 1 (default) leaves the
// step length and the leapfrog count to the chain as SimpleHMC.C does, 0 fixes them (SetMeanEpsilon(<0) +
// SetLeapFrog(20): independent chains, whole launches).  -DUSE_HARD_LIKELIHOOD swaps in THardLogLikelihood as
// SimpleHMC.C:3-5 does.
//
//   g++ -std=c++17 -O2 -Iinclude examples/SimpleHMC_amd.C -Lroot-simple-mcmc_amd/lib -lsmcmc_amd
//       -Wl,-rpath,$PWD/root-simple-mcmc_amd/lib -Wl,-rpath,/opt/rocm/lib -o hmc_amd.exe
//   ./hmc_amd.exe trials [output.csv [dim [chains [fused [tune]]]]]
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

#include "TSimpleHMC_amd.H"

#ifdef USE_HARD_LIKELIHOOD
typedef sMCMC::THardLogLikelihood Likelihood;
#else
typedef sMCMC::TDummyLogLikelihood Likelihood;
#endif

int SimpleHMC(int trials, const char* outputName, int dim, int chains, bool fused, bool tune) {
    std::cout << "Simple HMC (MI355X engine) D=" << dim << " chains=" << chains
              << (fused ? " fused order (matrix pipe)" : " reference order") << std::endl;
    sMCMC::TreeType tree("SimpleHMC", "Tree of accepted points");
    sMCMC::TSimpleHMC<Likelihood, Likelihood> hmc(&tree);
    Likelihood& like = hmc.GetLogLikelihood();
    like.SetDim(dim);
    like.Init();
    hmc.SetChains(chains);
    hmc.SetExactArithmetic(!fused);

    sMCMC::Vector p(like.GetDim(), 1.0);            // SimpleHMC.C:45
    hmc.Start(p, true);
    if (!tune) {
        // Init() couples the first and last coordinate with correlation 0.999999 (TDummyLogLikelihood.H:78-87):
        // the stiff direction has curvature ~1e6, so a fixed leapfrog step must stay below 2e-3
        hmc.SetMeanEpsilon(-0.001);
        hmc.SetLeapFrog(20);
    }

    const int burn = 100 + (int)p.size();           // SimpleHMC.C:51-60
    hmc.StepMany(burn);
    for (int i = 0; i < trials; ++i) {              // :63-73
        if (i % 1000 == 0)
            std::cout << "Steps: " << i << " Likelihood Calls: " << hmc.GetPotentialCount()
                      << " Gradient Calls: " << hmc.GetGradientCount() << std::endl;
        hmc.Step(true);
    }
    std::cout << "Finished " << trials << " trials with " << hmc.GetPotentialCount() << " calls and "
              << hmc.GetGradientCount() << " gradients, acceptance " << hmc.GetAcceptanceRate() << std::endl;
    tree.Write();
#ifndef SMCMC_HAVE_ROOT
    tree.WriteCsv(outputName);
    std::cout << "wrote " << tree.GetEntries() << " entries to " << outputName << std::endl;
#endif
    return 0;
}

int main(int argc, char** argv) {
    int trials = 1000, dim = 50, chains = 64, fused = 0, tune = 1;
    std::string outputName("SimpleHMC_amd.csv");
    if (argc > 1) { std::istringstream in(argv[1]); in >> trials; }
    if (argc > 2) outputName = argv[2];
    if (argc > 3) { std::istringstream in(argv[3]); in >> dim; }
    if (argc > 4) { std::istringstream in(argv[4]); in >> chains; }
    if (argc > 5) { std::istringstream in(argv[5]); in >> fused; }
    if (argc > 6) { std::istringstream in(argv[6]); in >> tune; }
    try {
        return SimpleHMC(trials, outputName.c_str(), dim, chains, fused != 0, tune != 0);
    } catch (const std::exception& e) {
        std::cerr << "SimpleHMC_amd: " << e.what() << std::endl;
        return 2;
    }
}
