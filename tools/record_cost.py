"""What the per-step record of smcmc_step_recorded costs the one-chain-per-wavefront kernel (one chain, D = 50 and 5)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from smcmc_amd_loader import load_package
pkg = load_package()
for dim in (50, 5):
    e = pkg.Engine(dim, 1, mode=pkg.MODE_PER_CHAIN)
    assert e.Start(np.zeros(dim))
    e.Step(2000); e.sync()
    out = {}
    for name, fn in (("Step(2048)", lambda: e.Step(2048)), ("StepRecorded(2048)", lambda: e.StepRecorded(2048))):
        fn(); e.sync()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        e.sync()
        out[name] = (time.perf_counter() - t0) / 10 / 2048 * 1e6
    print("D = %d: us per step " % dim + ", ".join("%s %.3f" % kv for kv in out.items()))
