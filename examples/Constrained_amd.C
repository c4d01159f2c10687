// Constrained_amd.C -- the reference's example4 (example4/Constrained.C:1-70) on the MI355X engine: a posterior with a
// constraint on the sum of 25 parameters and a prior on each (TConstrainedLikelihood.H), two unsaved burn-in legs each
// followed by UpdateProposal(), then `trials` saved steps.  Differences: the chain count is an argument (the pooled
// covariance of all chains feeds every update), and the burn-in legs are single launches.
//
//   g++ -std=c++17 -O2 -Iinclude examples/Constrained_amd.C -Lroot-simple-mcmc_amd/lib -lsmcmc_amd
//       -Wl,-rpath,$PWD/root-simple-mcmc_amd/lib -Wl,-rpath,/opt/rocm/lib -o constrained_amd.exe
//   ./constrained_amd.exe [trials [output.csv [chains]]]
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

#include "TSimpleMCMC_amd.H"

int Constrained(int trials, const char* outputName, int chains) {
    std::cout << "Simple MCMC Loaded (MI355X engine), " << chains << " chains" << std::endl;
    sMCMC::TreeType tree("Constrained", "Tree of accepted points");                 // Constrained.C:17

    sMCMC::TSimpleMCMC<sMCMC::TConstrainedLikelihood> mcmc(&tree);                   // :20
    sMCMC::TConstrainedLikelihood& like = mcmc.GetLogLikelihood();
    like.Init();                                                                     // :25
    mcmc.SetChains(chains);
    mcmc.GetProposeStep().SetDim((int)like.GetDim());                                // :28

    sMCMC::Vector p(like.GetDim());                                                  // :34: the origin
    mcmc.Start(p, false);                                                            // :36

    const int d = (int)p.size();
    mcmc.StepMany(10000 + d * d);                                                    // :39
    mcmc.GetProposeStep().SyncPooledCovariance();       // ensemble extension: the chains' pooled moments reach the covariance
    std::cout << "Finished burnin chain" << std::endl;
    mcmc.GetProposeStep().UpdateProposal();                                          // :43
    mcmc.StepMany(10000 + 10 * d * d);                                               // :46
    mcmc.GetProposeStep().SyncPooledCovariance();       // ensemble extension: the chains' pooled moments reach the covariance
    std::cout << "Finished burnin chain" << std::endl;
    mcmc.GetProposeStep().UpdateProposal();                                          // :50

    for (int i = 0; i < trials; ++i) mcmc.Step();                                    // :53
    std::cout << "Finished with " << mcmc.GetLogLikelihoodCount() << " calls" << std::endl;

    tree.Write();
#ifndef SMCMC_HAVE_ROOT
    tree.WriteCsv(outputName);
    std::cout << "wrote " << tree.GetEntries() << " entries to " << outputName << std::endl;
#endif
    return 0;
}

int main(int argc, char** argv) {
    int trials = 100000, chains = 256;                                               // :62
    std::string outputName("Constrained_amd.csv");
    if (argc > 1) { std::istringstream in(argv[1]); in >> trials; }
    if (argc > 2) outputName = argv[2];
    if (argc > 3) { std::istringstream in(argv[3]); in >> chains; }
    try {
        return Constrained(trials, outputName.c_str(), chains);
    } catch (const std::exception& e) {
        std::cerr << "Constrained_amd: " << e.what() << std::endl;
        return 2;
    }
}
