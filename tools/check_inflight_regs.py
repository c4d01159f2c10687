#!/usr/bin/env python3
"""The in-flight register check of root-simple-mcmc_amd/inflight_check.py (which build.py runs on every step-kernel unit)
on a listing made by hand:
    hipcc <flags of build.py> -DSMCMC_DP=50 -DSMCMC_LIKE=0 -S --cuda-device-only -o inst50.s csrc/smcmc_inst.hip
    python tools/check_inflight_regs.py inst50.s [kernel-name-substring]          exit code 1 on a finding"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "root-simple-mcmc_amd"))
import inflight_check  # noqa: E402


def main():
    bad = 0
    for name, reads, findings in inflight_check.check_listing(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""):
        print(f"{name}: {reads} assembly reads, {len(findings)} finding(s)")
        for idx, text, regs in findings[:12]:
            print(f"    +{idx}: {text}    (in flight: v{regs})")
        bad += 1 if findings else 0
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
