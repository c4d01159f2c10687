"""root-simple-mcmc_amd: MI355X-native many-chain adaptive Metropolis engine.

Drop-in for one path of ClarkMcGrew/root-simple-mcmc: the
sMCMC::TSimpleMCMC<L, TProposeAdaptiveStep>::Step() loop.  The directory name has
a hyphen, so import it through the loader at the repo root:

    from smcmc_amd_loader import load_package
    smcmc = load_package()          # module object, registered as "root_simple_mcmc_amd"
"""
from ._capi import (LIKE_ASYM, LIKE_CONSTRAINED, LIKE_HORRIFIC, LIKE_ISO_GAUSS, LIKE_QUADFORM, LIKE_ROSENBROCK, LIKE_USER, MODE_FROZEN, MODE_PER_CHAIN, MODE_POOLED, SmcmcError,  # noqa: F401
                    LIB_PATH, SIGNATURES, load)
from .engine import Autocorrelation, Engine, HmcEngine, PosteriorMoments, VaatEngine, selftest_detmath, selftest_mfma, selftest_mfma_strip  # noqa: F401
from . import build as _build_mod  # noqa: F401
from . import distributed  # noqa: F401


def build(**kw):
    """Compile the HIP library in-tree (hipcc, gfx950)."""
    return _build_mod.build(**kw)


FROZEN_DEFINITION_LIB_PATH = _build_mod.FROZEN_LIB_PATH


def build_frozen_definition():
    """The test library of the frozen-definition golden set (build.py build_frozen_definition); call after build()."""
    return _build_mod.build_frozen_definition()
