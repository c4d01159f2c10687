// Test driver: the backwards walk of Restore(tree, randomize) (TSimpleMCMC.H:305-333) on a
// hand-filled TColumnTree with a scripted uniform sequence.  argv: randomize(0/1) n logl... r...
// Prints the TotalSteps of the entry the walk settles on and how many uniforms it drew.
#include <cstdlib>
#include <iostream>
#include "TSimpleMCMC_amd.H"

int main(int argc, char** argv) {
    const bool randomize = std::atoi(argv[1]) != 0;
    const int n = std::atoi(argv[2]);
    sMCMC::TColumnTree tree("t");
    double logl = 0, rms = 0, acc = 0, acct = 0, sig = 0, cpt = 0, covt = 0;
    int total = 0, tr = 0, su = 0, nu = 0;
    std::vector<double> x(2), c(2), cov(3);
    tree.Branch("LogLikelihood", &logl); tree.Branch("TotalSteps", &total); tree.Branch("StepRMS", &rms);
    tree.Branch("Accepted", &x);
    tree.Branch("AdaptiveTrials", &tr); tree.Branch("AdaptiveSuccesses", &su); tree.Branch("AdaptiveNextUpdate", &nu);
    tree.Branch("AdaptiveAcceptance", &acc); tree.Branch("AdaptiveAcceptanceTrials", &acct);
    tree.Branch("AdaptiveSigma", &sig); tree.Branch("AdaptiveCentralPoint", &c);
    tree.Branch("AdaptiveCentralPointTrials", &cpt); tree.Branch("AdaptiveCovariance", &cov);
    tree.Branch("AdaptiveCovarianceTrials", &covt);
    for (int i = 0; i < n; ++i) {
        logl = std::atof(argv[3 + i]); total = i; x[0] = i; x[1] = -i; tr = 100 + i;
        tree.Fill();
    }
    int used = 0;
    auto uniform = [&]() { return std::atof(argv[3 + n + used++]); };
    sMCMC::detail::SavedEntry e = sMCMC::detail::ReadLastEntry(&tree, 3, randomize, uniform);
    std::cout << e.totalSteps << " " << used << " " << e.accepted[0] << " " << e.trials << "\n";
    sMCMC::detail::RestoreUniform u(7);
    double lo = 1, hi = 0;
    for (int i = 0; i < 100000; ++i) { const double r = u(); lo = std::min(lo, r); hi = std::max(hi, r); }
    std::cout << lo << " " << hi << "\n";
    return 0;
}
