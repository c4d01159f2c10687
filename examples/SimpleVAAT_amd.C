// SimpleVAAT_amd.C -- the reference's variable-at-a-time example (SimpleVAAT.C:9-91) on the MI355X engine: the
// header-form TDummyLogLikelihood (100 dimensions, Init()), sMCMC::TProposeVAATStep, a start point uniform in
// [-1, 1], Start(p, false), the explicit UpdateProposal(), then cycles x steps saved steps with the progress line
// (acceptance, successes / trials, mean sigma).  Differences: dimension, chain count and arithmetic order are
// arguments; the start point comes from the counter-based stream instead of gRandom.
//
//   g++ -std=c++17 -O2 -Iinclude examples/SimpleVAAT_amd.C -Lroot-simple-mcmc_amd/lib -lsmcmc_amd
//       -Wl,-rpath,$PWD/root-simple-mcmc_amd/lib -Wl,-rpath,/opt/rocm/lib -o vaat_amd.exe
//   ./vaat_amd.exe [cycles [steps [output.csv [dim [chains [fused]]]]]]
#include <algorithm>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>

#include "TSimpleMCMC_amd.H"
#include "TProposeVAATStep_amd.H"
#include "smcmc_detmath.h"

void SimpleVAAT(int cycles, int steps, std::string outputName, int dim, int chains, bool fused) {
    std::cout << "Simple VAAT Loaded (MI355X engine) D=" << dim << " chains=" << chains
              << (fused ? " fused order" : " reference order") << std::endl;
    sMCMC::TreeType tree("SimpleVAAT", "Tree of accepted points");                  // SimpleVAAT.C:18

    sMCMC::TSimpleMCMC<sMCMC::TDummyLogLikelihood, sMCMC::TProposeVAATStep> mcmc(&tree);   // :21
    sMCMC::TDummyLogLikelihood& like = mcmc.GetLogLikelihood();
    like.SetDim(dim);
    like.Init();                                                                    // :26
    mcmc.SetChains(chains);
    mcmc.SetExactArithmetic(!fused);

    mcmc.GetProposeStep().SetDim(like.GetDim());                                    // :29
    // mcmc.GetProposeStep().SetUniform(1,-0.5,0.5);                                // :32

    sMCMC::Vector p(like.GetDim());                                                 // :40-41
    for (std::size_t i = 0; i < p.size(); ++i) {
        const smcmc_u32x4 blk = smcmc_draw_block(20240607ull, 0u, 0u, (uint32_t)(i >> 2), SMCMC_STREAM_START);
        p[i] = -1.0 + 2.0 * smcmc_u01(blk.v[i & 3u]);
    }

    mcmc.Start(p, false);                                                           // :43
    mcmc.GetProposeStep().UpdateProposal();                                         // :44

    int verbosity = std::max(1, steps * cycles / 100);                              // :47
    int trial = 0;
    for (int cycle = 0; cycle < cycles; ++cycle) {
        for (int step = 0; step < steps; ++step) {
            ++trial;
            mcmc.Step();                                                            // :52
            if (trial % verbosity == 0) {
                std::cout << "Trial " << cycle + 1 << ":" << step + 1 << " Total: " << trial << "/" << cycles * steps
                          << " Acceptance: " << mcmc.GetProposeStep().GetAcceptance() << " ("
                          << mcmc.GetProposeStep().GetSuccesses() << "/" << mcmc.GetProposeStep().GetTrials() << ")"
                          << " Sigma: " << mcmc.GetProposeStep().GetSigma() << std::endl;
            }
        }
    }

    tree.Write();                                                                   // :65
#ifndef SMCMC_HAVE_ROOT
    tree.WriteCsv(outputName.c_str());
    std::cout << "wrote " << tree.GetEntries() << " entries to " << outputName << std::endl;
#endif
    std::cout << "Exit" << std::endl;
}

int main(int argc, char** argv) {
    int cycles = 10, steps = 1000, dim = 100, chains = 64, fused = 0;              // :75-77
    std::string outputName("SimpleVAAT_amd.csv");
    if (argc > 1) { std::istringstream in(argv[1]); in >> cycles; }
    if (argc > 2) { std::istringstream in(argv[2]); in >> steps; }
    if (argc > 3) outputName = argv[3];
    if (argc > 4) { std::istringstream in(argv[4]); in >> dim; }
    if (argc > 5) { std::istringstream in(argv[5]); in >> chains; }
    if (argc > 6) { std::istringstream in(argv[6]); in >> fused; }
    try {
        SimpleVAAT(cycles, steps, outputName, dim, chains, fused != 0);
    } catch (const std::exception& e) {
        std::cerr << "SimpleVAAT_amd: " << e.what() << std::endl;
        return 2;
    }
    return 0;
}
