#!/usr/bin/env python3
"""Headline benchmark: chain-steps/s of the TSimpleMCMC::Step() path on MI355X.

Workload (BASELINE.json configs[1]): README-form TDummyLogLikelihood (the
synthetic iso-Gaussian, README.md:57-66), D = 50, 65 536 chains per GPU,
TProposeAdaptiveStep with the covariance pooled over all chains, reference
arithmetic order (EXACT).  One bench "step" = one adaptation window: 256
ensemble steps in ONE kernel launch, then the pooled moment reduction, the
all-reduce over ranks (RCCL, only when --gpus > 1) and UpdateProposal.  Every
chain-step does the full reference work: D normals through U, logL, Metropolis
test, scalar adaptation and the second-moment fold.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM = 50
CHAINS_PER_GPU = 65536
WINDOW = 256                      # ensemble steps per launch / adaptation window
BYTES_PER_CHAIN_STEP = 16 * DIM + 16   # state round-trip model
HBM_PEAK_GBPS = 8000.0


def cpu_baseline(seconds=12.0):
    """The oracle's single reference chain (oracle_chain, D=50 iso-Gaussian,
    adaptive) timed on one host core for a bounded sample."""
    from oracle import oracle as O
    O.build()
    c = O.Chain(DIM)
    c.start(np.zeros(DIM))
    c.run_quiet(20000)            # warm up / first adaptation
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        c.run_quiet(50000)
        n += 50000
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "chain-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} Step() calls of one D={DIM} iso-Gaussian chain, TProposeAdaptiveStep, "
                      f"oracle/oracle_core.h compiled gcc -O2 (no -march), {dt:.1f} s"}


ESS_STEPS, ESS_STRIDE = 8192, 16


def measure_ess(eng, torch, dim):
    """Effective samples per chain-step of the current (adapted) ensemble: the trace of every chain stays on the
    device, smcmc_autocorrelation_sums pools the lagged products (MakeAutocorrelation.C:108-148), the host turns
    64 lags x dim numbers into the integrated autocorrelation time (worst dimension)."""
    slots = ESS_STEPS // ESS_STRIDE
    npad, dpad = eng.nchains_padded, eng.dim_padded
    sx = torch.empty((slots, dpad, npad), dtype=torch.float64, device="cuda")
    sl = torch.empty((slots, npad), dtype=torch.float64, device="cuda")
    eng.StepSave(ESS_STEPS, sx.data_ptr(), sl.data_ptr(), stride=ESS_STRIDE)
    torch.cuda.synchronize()
    tau = eng.AutocorrelationSums(sx.data_ptr(), slots, stream=torch.cuda.current_stream().cuda_stream).tau()
    del sx, sl
    return 1.0 / (float(tau.max()) * ESS_STRIDE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU)
    ap.add_argument("--window", type=int, default=WINDOW)
    ap.add_argument("--fast", action="store_true", help="fused multiply-add arithmetic (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true", help="skip the ESS trace after the timed region")
    ap.add_argument("--frozen", action="store_true", help="diagnostic: frozen covariance, no moment fold")
    ap.add_argument("--dim", type=int, default=DIM, help="diagnostic: other dimension (not the headline)")
    ap.add_argument("--header-tdummy", action="store_true",
                    help="the other C2 likelihood of SURVEY.md 8(d): header-form TDummyLogLikelihood (quadratic form, Error "
                         "from Init(), correlation 0.999999 between the first and last coordinate); not the headline")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from smcmc_amd_loader import load_package
    pkg = load_package()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    stream = torch.cuda.current_stream()
    dim = args.dim
    like, like_params = pkg.LIKE_ISO_GAUSS, None
    if args.header_tdummy:
        # TDummyLogLikelihood::Init() (TDummyLogLikelihood.H:44-142): identity covariance except the (0, D-1) pair
        cov = np.eye(dim)
        cov[0, dim - 1] = cov[dim - 1, 0] = 0.999999
        like, like_params = pkg.LIKE_QUADFORM, np.linalg.inv(cov)
    eng = pkg.Engine(dim, args.chains, likelihood=like, likelihood_params=like_params, seed=20240607,
                     chain_offset=rank * args.chains, device=local,
                     mode=pkg.MODE_FROZEN if args.frozen else pkg.MODE_POOLED,
                     exact=not args.fast, stream=stream.cuda_stream)
    assert eng.Start(np.zeros(dim))
    mbuf = torch.zeros(eng.moments_size, dtype=torch.float64, device="cuda")

    kernel_ms = []

    def window(timed):
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
        eng.Step(args.window)
        if timed:
            e1.record(stream)
            kernel_ms.append((e0, e1))
        if args.frozen:
            return
        eng.reduce_moments()
        if world > 1:
            eng.export_moments(mbuf.data_ptr())
            dist.all_reduce(mbuf)
            eng.import_moments(mbuf.data_ptr())
        eng.apply_moments()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        window(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        window(True)
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    chain_steps = float(args.chains) * world * args.window * args.steps
    value = chain_steps / dt
    kms = float(np.mean([a.elapsed_time(b) for a, b in kernel_ms]))
    per_launch = float(args.chains) * args.window
    bytes_cs = 16 * dim + 16
    achieved = per_launch * bytes_cs / (kms * 1e-3) / 1e9

    # HBM bytes of one launch from the PMC counters (collected in separate rocprofv3 passes,
    # profiles/r01_pmc_traffic.json); only valid for the configuration it was measured on
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if (os.path.exists(pmc) and dim == DIM and args.chains == CHAINS_PER_GPU and args.window == WINDOW
            and not args.fast and not args.frozen):
        traffic = json.load(open(pmc))["hbm_bytes_per_launch"]

    # ESS per chain-step (outside the timed region): a trace of the adapted ensemble saved
    # every ESS_STRIDE steps through the device-side save path, autocorrelation per dimension
    # (definition of MakeAutocorrelation.C:127-148) averaged over a subset of chains, Geyer's
    # initial positive sequence, minimum over dimensions.
    ess_per_chain_step = None
    if rank == 0 and not args.no_ess:
        try:
            ess_per_chain_step = measure_ess(eng, torch, dim)
        except Exception as exc:   # a diagnostic, never a reason to lose the bench line
            print("ESS measurement skipped: %r" % (exc,), file=sys.stderr)

    naccept = eng.lane("naccept").astype(np.float64)
    total_steps = eng.get_param("TOTAL_STEPS")
    out = {
        "metric": "chain-steps/s on D=50 TDummyLogLikelihood, 65 536 chains; ESS/s + accept rate",
        "value": value, "unit": "chain-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("TDummyLogLikelihood header form (quadratic form)" if args.header_tdummy else
                                "TDummyLogLikelihood README form (iso-Gaussian)") +
                               " D=%d, %d chains/GPU, TProposeAdaptiveStep pooled covariance, window=%d steps/launch"
                               % (dim, args.chains, args.window),
                   "dim": dim, "mode": "frozen" if args.frozen else "pooled", "chains_per_gpu": args.chains, "window": args.window,
                   "arithmetic": "fused" if args.fast else "reference-order", "seed": 20240607},
        "accept_rate": float(naccept.sum() / (total_steps * args.chains)),
        "ess_per_chain_step": ess_per_chain_step,
        "ess_per_s": (ess_per_chain_step * value) if ess_per_chain_step else None,
        "mean_sigma": float(eng.lane("sigma").mean()),
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                     "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE)",
                     "algorithmic_bytes_per_launch": per_launch * bytes_cs,
                     "kernel": "step_kernel<%d,%s,%s,tri,moments>" % (dim, "QUADFORM" if args.header_tdummy else "ISO",
                                                                           "fused" if args.fast else "exact"),
                     "kernel_ms": kms, "bytes_per_chain_step": bytes_cs,
                     "chain_steps_per_launch": per_launch},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
