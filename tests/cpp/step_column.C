// Test driver: one chain of sMCMC::TSimpleMCMC<TIsoGaussLogLikelihood> with a frozen covariance (every chain of the
// FROZEN engine is a reference chain), Step(true, metropolis) nsteps times with the `Step` branch on, the tree written
// as CSV.  argv: dim nsteps metropolis out.csv [exact=1].  tests/test_cpp_host.py diffs the Step / Accepted / LogLikelihood
// columns against the CPU restatement of TSimpleMCMC.H:370-496 and checks GetProposed() on the way.
#include <cstdlib>
#include <iostream>
#include "TSimpleMCMC_amd.H"

int main(int argc, char** argv) {
    if (argc < 5) return 64;
    const int dim = std::atoi(argv[1]), nsteps = std::atoi(argv[2]), metropolis = std::atoi(argv[3]);
    try {
        sMCMC::TreeType tree("SimpleMCMC", "");
        sMCMC::TSimpleMCMC<sMCMC::TIsoGaussLogLikelihood> mcmc(&tree, true);
        if (argc > 5) mcmc.SetExactArithmetic(std::atoi(argv[5]) != 0);
        mcmc.GetProposeStep().SetDim(dim);
        mcmc.GetProposeStep().SetCovarianceFrozen(true);
        sMCMC::Vector p((std::size_t)dim, 0.25);
        if (!mcmc.Start(p, true)) return 1;
        if (mcmc.GetProposed() != p) { std::cerr << "GetProposed() after Start is not the start point\n"; return 3; }
        int moved = 0;
        for (int s = 0; s < nsteps; ++s) {
            const sMCMC::Vector before = mcmc.GetAccepted();
            const bool took = mcmc.Step(true, metropolis);
            moved += took ? 1 : 0;
            // Step()'s return value, GetProposed() and GetAccepted() hang together (TSimpleMCMC.H:484-491)
            if (took && mcmc.GetAccepted() != mcmc.GetProposed()) { std::cerr << "accepted != proposed after a move\n"; return 3; }
            if (!took && mcmc.GetAccepted() != before) { std::cerr << "the point moved on a rejected step\n"; return 3; }
        }
        std::cout << "moved " << moved << " entries " << tree.GetEntries() << std::endl;
        tree.WriteCsv(argv[4]);
    } catch (const std::exception& e) {
        std::cerr << "step_column: " << e.what() << std::endl;
        return 2;
    }
    return 0;
}
