"""The C-ABI library loads without a GPU and exports every symbol include/smcmc.h
declares; nothing here launches a kernel."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "smcmc.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(smcmc_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(smcmc):
    declared = _declared_symbols()
    assert len(declared) > 35
    assert sorted(smcmc.SIGNATURES) == declared      # the ctypes binding covers the whole header


def test_library_exports_every_declared_symbol(smcmc):
    assert os.path.exists(smcmc.LIB_PATH), "build it with __graft_entry__.build()"
    lib = ctypes.CDLL(smcmc.LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in include/smcmc.h but not exported"


def test_the_frozen_definition_test_library_is_built_and_whole(smcmc):
    """lib/libsmcmc_amd_frozen_definition.so (build.py build_frozen_definition; tests/test_golden.py's GPU case runs on it)
    carries the same C ABI as the product library."""
    assert os.path.exists(smcmc.FROZEN_DEFINITION_LIB_PATH), "__graft_entry__.build() makes it"
    lib = ctypes.CDLL(smcmc.FROZEN_DEFINITION_LIB_PATH)
    for name in _declared_symbols():
        assert hasattr(lib, name), name


def test_build_facts_without_a_gpu(smcmc):
    lib = smcmc.load()
    assert lib.smcmc_version() >= 100
    assert lib.smcmc_max_register_dim() == 63
    assert lib.smcmc_status_string(0) == b"ok"
    assert b"no HIP device" in lib.smcmc_status_string(7)


def test_no_cpu_fallback(smcmc):
    """Without a device the product path fails loudly instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    with pytest.raises(smcmc.SmcmcError) as err:
        smcmc.Engine(5, 10)
    assert err.value.status == 7                     # SMCMC_ERR_NO_DEVICE
    import numpy as np
    with pytest.raises(smcmc.SmcmcError):
        smcmc.selftest_detmath(0, np.ones(4))
    for cls in (smcmc.HmcEngine, smcmc.VaatEngine):
        with pytest.raises(smcmc.SmcmcError) as err:
            cls(5, 10)
        assert err.value.status == 7


def test_product_does_not_touch_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may use oracle/."""
    pkg = os.path.join(ROOT, "root-simple-mcmc_amd")
    offenders = []
    for base, _, files in os.walk(pkg):
        if os.path.basename(base) in ("build", "lib", "__pycache__"):
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".H")):
                if re.search(r"\boracle\b", open(os.path.join(base, f), errors="ignore").read()):
                    offenders.append(os.path.join(base, f))
    for f in os.listdir(os.path.join(ROOT, "include")):
        if re.search(r'#include\s*"[^"]*oracle', open(os.path.join(ROOT, "include", f)).read()):
            offenders.append(f)
    # comments may name the oracle as the checker; includes/imports may not
    real = []
    for path in offenders:
        text = open(path, errors="ignore").read()
        if re.search(r'(#include\s*"[^"]*oracle|^\s*(from|import)\s+oracle)', text, flags=re.M):
            real.append(path)
    assert not real, real
