// The unchanged caller's loop (SimpleMCMC.C:176-243): one chain of sMCMC::TSimpleMCMC<TIsoGaussLogLikelihood> adapting
// as the reference does, `for (...) mcmc.Step(save)` one call at a time, the getters SimpleMCMC.C:209-219 prints read
// every `verbosity` steps, UpdateProposal() + the per-cycle setters (SimpleMCMC.C:245-256) at the end of every cycle.
// argv: dim cycles steps save(0|1) runahead(0|1|2: on, and turned off after the first cycle) [out.csv [chains]]   (chains > 1: an ensemble, each chain adapting alone)
// Prints "steps_per_s <rate>" for the timed loop, and -- for the parity test -- with out.csv the tree (every entry's
// LogLikelihood / Accepted / Adaptive* columns) so that the run-ahead Step() can be diffed against Step() one launch
// at a time: they are the same chain.
#include <chrono>
#include <cstdlib>
#include <iostream>
#include "TSimpleMCMC_amd.H"

int main(int argc, char** argv) {
    if (argc < 6) return 64;
    const int dim = std::atoi(argv[1]), cycles = std::atoi(argv[2]), steps = std::atoi(argv[3]);
    const bool save = std::atoi(argv[4]) != 0;
    const int ahead = std::atoi(argv[5]);
    try {
        sMCMC::TreeType tree("SimpleMCMC", "");
        sMCMC::TSimpleMCMC<sMCMC::TIsoGaussLogLikelihood> mcmc(&tree, true);
        mcmc.SetRunAhead(ahead != 0);
        if (argc > 7) { mcmc.SetChains(std::atoi(argv[7])); mcmc.GetProposeStep().SetPerChainAdaptation(true); }
        mcmc.GetProposeStep().SetDim(dim);
        sMCMC::Vector p((std::size_t)dim, 0.0);
        if (!mcmc.Start(p, false)) return 1;
        mcmc.GetProposeStep().SetAcceptanceWindow(1000);                 // SimpleMCMC.C:196-200
        mcmc.GetProposeStep().SetAcceptanceRigidity(2.0);
        mcmc.GetProposeStep().SetCovarianceWindow(cycles * steps);
        mcmc.GetProposeStep().SetCovarianceUpdateDeweighting(0.20);
        mcmc.GetProposeStep().SetNextUpdate(1E+9);
        const int verbosity = steps / 5 > 0 ? steps / 5 : 1;
        double printed = 0.0;
        int moved = 0, trial = 0;
        const auto t0 = std::chrono::steady_clock::now();
        for (int cycle = 0; cycle < cycles; ++cycle) {
            for (int i = 0; i < steps; ++i) {
                if (++trial % verbosity == 0)                            // what SimpleMCMC.C:209-219 prints
                    printed += mcmc.GetProposeStep().GetAcceptance() + mcmc.GetProposeStep().GetSuccesses() +
                               mcmc.GetProposeStep().GetSigma() + mcmc.GetProposeStep().GetCovarianceTrace() + mcmc.GetStepRMS();
                moved += mcmc.Step(save) ? 1 : 0;
            }
            mcmc.GetProposeStep().UpdateProposal();                      // SimpleMCMC.C:245-256
            mcmc.GetProposeStep().SetAcceptanceRigidity(2.0);
            mcmc.GetProposeStep().SetCovarianceUpdateDeweighting(0.0);
            mcmc.GetProposeStep().SetNextUpdate(10 * steps);
            if (ahead == 2 && cycle == 0) mcmc.SetRunAhead(false);
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        mcmc.SaveStep();
        std::cout << "steps_per_s " << (double)cycles * steps / dt << " moved " << moved << " entries " << tree.GetEntries()
                  << " run_ahead " << (mcmc.GetRunAhead() ? 1 : 0) << " printed " << printed << std::endl;
        if (argc > 6) tree.WriteCsv(argv[6]);
    } catch (const std::exception& e) {
        std::cerr << "step_loop: " << e.what() << std::endl;
        return 2;
    }
    return 0;
}
