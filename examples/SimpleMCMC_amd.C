// SimpleMCMC_amd.C -- the reference's working example (SimpleMCMC.C:45-301) on the
// MI355X engine: same command line (cycles, steps, output, [dim, chains, restore]), same
// schedule (one unsaved burn-in pass, ResetProposal, four burn-in passes with
// UpdateProposal, then cycles x steps with UpdateProposal per cycle, a final
// forced SaveStep), same tree schema.  Differences: the likelihood dimension is
// an argument, `chains` chains run in lock step, and the covariance adapts through
// the pooled update (SyncPooledCovariance) every 256 steps instead of per chain.
//
// build (see tests/test_cpp_host.py):
//   g++ -std=c++17 -O2 -Iinclude examples/SimpleMCMC_amd.C -Lroot-simple-mcmc_amd/lib -lsmcmc_amd
//       -Wl,-rpath,$PWD/root-simple-mcmc_amd/lib -Wl,-rpath,/opt/rocm/lib -o mcmc_amd.exe
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

#include "TSimpleMCMC_amd.H"
#ifdef SMCMC_HAVE_ROOT
#include <TFile.h>
#endif

namespace {

template <typename MCMC>
void RunSteps(MCMC& mcmc, int steps, int window, bool saveLast) {
    int done = 0;
    while (done < steps) {
        const int n = (steps - done < window) ? (steps - done) : window;
        mcmc.StepMany(n);
        done += n;
        if (!mcmc.GetProposeStep().GetCovarianceFrozen()) mcmc.GetProposeStep().SyncPooledCovariance();
        if (saveLast) mcmc.SaveStep(false);      // one tree entry (chain 0) per window
    }
}

}  // namespace

int SimpleMCMC(int cycles, int steps, const char* outputName, int dim, int chains, const char* restoreName) {
    std::cout << "Simple MCMC (MI355X engine) D=" << dim << " chains=" << chains << std::endl;
    sMCMC::TreeType tree("SimpleMCMC", "Tree of accepted points");
    // the likelihood is chosen at compile time, as SimpleMCMC.C:5-39 does: the README's isotropic Gaussian by
    // default, -DUSE_HEADER_TDUMMY for the quadratic form of TDummyLogLikelihood.H, -DUSE_HARD_LIKELIHOOD for Rosenbrock,
    // -DUSE_ASYM_LIKELIHOOD / -DUSE_HORRIFIC_LIKELIHOOD for the two stress targets (SimpleMCMC.C:5-30)
#if defined(USE_HEADER_TDUMMY)
    typedef sMCMC::TDummyLogLikelihood Likelihood;
#elif defined(USE_HARD_LIKELIHOOD)
    typedef sMCMC::THardLogLikelihood Likelihood;
#elif defined(USE_ASYM_LIKELIHOOD)
    typedef sMCMC::TASymLogLikelihood Likelihood;
#elif defined(USE_HORRIFIC_LIKELIHOOD)
    typedef sMCMC::THorrificLogLikelihood Likelihood;
#else
    typedef sMCMC::TIsoGaussLogLikelihood Likelihood;
#endif
    sMCMC::TSimpleMCMC<Likelihood> mcmc(&tree, true);
    Likelihood& like = mcmc.GetLogLikelihood();
    like.SetDim(dim);
    like.Init();
    mcmc.SetChains(chains);
    mcmc.GetProposeStep().SetDim(like.GetDim());

    sMCMC::Vector p(like.GetDim(), 0.0);            // SimpleMCMC.C:149: start at zero
    if (!mcmc.Start(p, false)) {
        std::cout << "bad starting point" << std::endl;
        return 1;
    }
    if (restoreName) {                              // SimpleMCMC.C:50-55, 153-157
        std::cout << "Restore from " << restoreName << std::endl;
#ifdef SMCMC_HAVE_ROOT
        TFile restoreFile(restoreName, "old");
        mcmc.Restore((TTree*)restoreFile.Get("SimpleMCMC"));
#else
        sMCMC::TColumnTree restoreTree = sMCMC::TColumnTree::ReadCsv(restoreName);
        mcmc.Restore(&restoreTree);
#endif
        std::cout << "State Restored: trials " << mcmc.GetProposeStep().GetTrials() << " sigma "
                  << mcmc.GetProposeStep().GetSigma() << std::endl;
    }
    const int window = 256;

    // burn-in, SimpleMCMC.C:163-201
    int aWin = (int)(0.1 * steps);
    if (aWin > 1000) aWin = 1000;
    if (aWin < 100) aWin = 100;
    mcmc.GetProposeStep().SetAcceptanceWindow(aWin);
    mcmc.GetProposeStep().SetCovarianceWindow(steps);
    RunSteps(mcmc, steps, window, false);
    mcmc.GetProposeStep().ResetProposal();
    mcmc.GetProposeStep().SetAcceptanceWindow(aWin);
    mcmc.GetProposeStep().SetCovarianceWindow(2 * steps);
    mcmc.GetProposeStep().SetCovarianceUpdateDeweighting(0.5);
    for (int cycle = 0; cycle < 4; ++cycle) {
        RunSteps(mcmc, steps, window, true);
        mcmc.GetProposeStep().UpdateProposal();
        std::cout << "Finished burnin chain " << cycle << std::endl;
    }

    // main chain, SimpleMCMC.C:204-256
    mcmc.GetProposeStep().SetAcceptanceWindow(1000);
    mcmc.GetProposeStep().SetAcceptanceRigidity(2.0);
    mcmc.GetProposeStep().SetCovarianceWindow(cycles * steps);
    mcmc.GetProposeStep().SetCovarianceUpdateDeweighting(0.20);
    for (int cycle = 0; cycle < cycles; ++cycle) {
        RunSteps(mcmc, steps, window, true);
        std::cout << "Trial " << cycle << ": A: " << mcmc.GetProposeStep().GetAcceptance() << "/"
                  << mcmc.GetProposeStep().GetSuccesses() << " S: " << mcmc.GetProposeStep().GetSigma()
                  << " T: " << mcmc.GetProposeStep().GetCovarianceTrace() << " RMS: " << mcmc.GetStepRMS()
                  << std::endl;
        mcmc.GetProposeStep().UpdateProposal();
        mcmc.GetProposeStep().SetAcceptanceRigidity(2.0);
        mcmc.GetProposeStep().SetCovarianceUpdateDeweighting(0.0);
    }
    std::cout << "Finished with " << mcmc.GetLogLikelihoodCount() << " calls per chain" << std::endl;

    mcmc.SaveStep();                                // SimpleMCMC.C:262: forces the full proposal state out
    tree.Write();
#ifndef SMCMC_HAVE_ROOT
    tree.WriteCsv(outputName);
    std::cout << "wrote " << tree.GetEntries() << " entries to " << outputName << std::endl;
#endif
    return 0;
}

int main(int argc, char** argv) {
    int cycles = 10, steps = 1000, dim = 5, chains = 64;
    std::string outputName("SimpleMCMC_amd.csv");
    if (argc > 1) { std::istringstream in(argv[1]); in >> cycles; }
    if (argc > 2) { std::istringstream in(argv[2]); in >> steps; }
    if (argc > 3) outputName = argv[3];
    if (argc > 4) { std::istringstream in(argv[4]); in >> dim; }
    if (argc > 5) { std::istringstream in(argv[5]); in >> chains; }
    const char* restoreName = (argc > 6) ? argv[6] : NULL;
    try {
        return SimpleMCMC(cycles, steps, outputName.c_str(), dim, chains, restoreName);
    } catch (const std::exception& e) {
        std::cerr << "SimpleMCMC_amd: " << e.what() << std::endl;
        return 2;
    }
}
