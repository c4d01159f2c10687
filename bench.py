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

After the timed headline (N = 1 only, each with its own warm-up and HIP-event timing, none of it inside
the headline's timed region) the other single-GPU workloads of BASELINE.json go into `extra`: config 2
with the header-form TDummyLogLikelihood, config 3, config 4's per-GPU share, config 5.

Prints ONE JSON line (rank 0).
"""
import argparse
import gc
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

DIM = 50
CHAINS_PER_GPU = 65536
WINDOW = 256                      # ensemble steps per launch / adaptation window
HBM_PEAK_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E peak (spec)
FP64_PEAK_TFLOPS = 78.6           # FP64 vector = FP64 matrix peak (they share one pipe)
def _latest_traffic_file():
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    return files[-1] if files else os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")


TRAFFIC_FILE = _latest_traffic_file()   # the newest round's PMC summary (tools/profile_bench.sh + tools/summarize_pmc.py)


def bytes_per_chain_step(dim):
    """state round-trip model, read x + write x + logL read/write."""
    return 16 * dim + 16


def flops_per_chain_step(dim, like):
    """proposal D^2 (triangular, multiply + add) + covariance moments D^2 + D + likelihood."""
    logl = {"iso": 2 * dim, "quadform": 3 * dim * dim, "rosenbrock": 8 * dim}[like]
    return dim * dim + dim * dim + dim + logl


def tdummy_error(dim):
    """TDummyLogLikelihood::Init() (TDummyLogLikelihood.H:44-142): identity covariance except the (0, D-1) pair."""
    cov = np.eye(dim)
    cov[0, dim - 1] = cov[dim - 1, 0] = 0.999999
    return np.linalg.inv(cov)


def cpu_baseline(seconds=8.0):
    """The oracle's reference chain (oracle_chain = TSimpleMCMC<L, TProposeAdaptiveStep>::Step(false), D=50
    iso-Gaussian, adaptive) timed on the host for a bounded sample: one core, then every core with an
    independent chain each (what continue-chain.sh does with processes)."""
    from oracle import oracle as O
    O.build()

    def run(chain_id, stop_at, out):
        c = O.Chain(DIM, chain_id=chain_id)
        c.start(np.zeros(DIM))
        c.run_quiet(20000)            # warm up / first adaptation
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() < stop_at[0]:
            c.run_quiet(25000)        # (ctypes releases the GIL for the duration of the C call)
            n += 25000
        out[chain_id] = (n, time.perf_counter() - t0)

    res = {}
    stop = [time.perf_counter() + seconds]
    run(0, stop, res)
    n1, dt1 = res[0]
    # the cores this process may use, at most the 16 a one-GPU box shares out
    cores = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 16)
    res, stop = {}, [time.perf_counter() + seconds + 1.0]
    threads = [threading.Thread(target=run, args=(k, stop, res)) for k in range(cores)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    rate_all = sum(n / dt for n, dt in res.values())
    return {"value": n1 / dt1, "unit": "chain-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n1} Step(false) calls of one D={DIM} iso-Gaussian chain, TProposeAdaptiveStep, "
                      f"oracle/oracle_core.h compiled gcc -O2 (no -march), {dt1:.1f} s",
            "all_cores": {"value": rate_all, "unit": "chain-steps/s", "cores": cores,
                          "sample": f"{cores} independent chains, one thread each, {seconds:.0f} s"}}


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


def fractions(rate, dim, like):
    """The two roofline fractions of a Metropolis workload from its chain-steps/s."""
    return {"hbm_model_GBps": rate * bytes_per_chain_step(dim) / 1e9,
            "hbm_model_frac": rate * bytes_per_chain_step(dim) / 1e9 / HBM_PEAK_GBPS,
            "fp64_TFLOPs": rate * flops_per_chain_step(dim, like) / 1e12,
            "fp64_frac": rate * flops_per_chain_step(dim, like) / 1e12 / FP64_PEAK_TFLOPS}


def extra_metropolis(pkg, torch, stream, name, dim, chains, like, like_id, prm, x0, exact, windows, window=WINDOW,
                     stride=1, dense_quadform=None, adapt_windows=4):
    """One of the other BASELINE configs: pooled covariance, `window` steps then a sync, timed with HIP events on the
    engine's stream (device time of the step launches, moment folds included) and with the wall clock (sync included)."""
    eng = pkg.Engine(dim, chains, likelihood=like_id, likelihood_params=prm, seed=20240607, mode=pkg.MODE_POOLED,
                     exact=exact, stream=stream.cuda_stream)
    if dim > 63:
        eng.set_param("MOMENT_STRIDE", stride)
    if dense_quadform is not None:
        eng.set_param("DENSE_QUADFORM", 1.0 if dense_quadform else 0.0)
    assert eng.Start(x0)
    quadform_walk = None
    if like_id == pkg.LIKE_QUADFORM:
        quadform_walk = "dense D^2-term sum" if eng.get_param("DENSE_QUADFORM") else "non-zero entries of Error only (bit for bit the dense sum)"
    gc.collect(); gc.disable()                        # a full collection of a torch-sized heap stalls the host for ~70 ms: here,
                                                      # not next to the timed windows (an idle GPU drops its clocks)
    for _ in range(adapt_windows):                    # adaptation windows before the timed ones (the proposal has settled)
        eng.Step(window); eng.sync()
    torch.cuda.synchronize()
    acc0, steps0 = float(eng.lane("naccept").sum()), eng.get_param("TOTAL_STEPS")
    evs = []
    t0 = time.perf_counter()
    for _ in range(windows):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        eng.Step(window)
        e1.record(stream)
        evs.append((e0, e1))
        eng.sync()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    kms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    rate = chains * window * windows / dt
    out = {"workload": name, "chain_steps_per_s": rate, "ms_per_window": dt / windows * 1e3, "step_launches_ms": kms,
           "ms_per_ensemble_step": kms / window, "windows": windows, "window": window,
           "arithmetic": "reference-order" if exact else "fused (matrix pipe)" if dim > 63 else "fused",
           "moment_stride": stride, "adaptation_windows_before": adapt_windows,
           "accept_rate": float((eng.lane("naccept").sum() - acc0) / ((eng.get_param("TOTAL_STEPS") - steps0) * chains))}
    if quadform_walk:
        out["quadratic_form"] = quadform_walk
    out.update(fractions(rate, dim, like))
    eng.close()
    return out


def extra_perchain(pkg, torch, stream, dim, chains, steps, launches, header_form=False):
    """SMCMC_MODE_PER_CHAIN: every chain adapts its own covariance every step and decomposes it on its own schedule --
    the reference's own mode, the one configuration whose HBM traffic per chain-step is O(D^2): the chain's packed
    decomposition is read, its packed covariance read and written (8 * 3 * D (D + 1) / 2 bytes) on top of the O(D) state."""
    kw = {"likelihood": pkg.LIKE_QUADFORM, "likelihood_params": tdummy_error(dim)} if header_form else {}
    eng = pkg.Engine(dim, chains, seed=20240607, mode=pkg.MODE_PER_CHAIN, stream=stream.cuda_stream, **kw)
    assert eng.Start(np.zeros(dim))
    eng.Step(8)
    torch.cuda.synchronize()
    evs = []
    t0 = time.perf_counter()
    for _ in range(launches):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); eng.Step(steps); e1.record(stream)
        evs.append((e0, e1))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    kms = float(np.mean([a.elapsed_time(b) for a, b in evs]))
    rate = chains * steps * launches / dt
    nbytes = 8 * 3 * dim * (dim + 1) // 2 + 8 * 7 * dim + 16
    out = {"workload": "TDummyLogLikelihood %s form D=%d, %d chains, TProposeAdaptiveStep per chain (own covariance "
                       "every step, own UpdateProposal schedule)" % ("header" if header_form else "README", dim, chains),
           "chain_steps_per_s": rate, "kernel_chain_steps_per_s": chains * steps / (kms * 1e-3),
           "us_per_ensemble_step": kms * 1e3 / steps, "steps_per_launch": steps, "launches": launches,
           "algorithmic_bytes_per_chain_step": nbytes, "hbm_GBps": chains * steps / (kms * 1e-3) * nbytes / 1e9,
           "hbm_frac": chains * steps / (kms * 1e-3) * nbytes / 1e9 / HBM_PEAK_GBPS,
           "adaptive_state_MB": chains * 8 * (dim * (dim + 1) // 2 + dim * dim + 2 * dim) / 1e6,
           "accept_rate": float(eng.lane("naccept").sum() / (eng.get_param("TOTAL_STEPS") * chains))}
    eng.close()
    return out


def hmc_ess_per_trajectory(h, torch, nsteps=256):
    """Effective samples per trajectory of the running ensemble (outside any timed region): nsteps more trajectories,
    the positions of every chain copied into a device trace after each (smcmc_hmc_copy_positions: 8.4 GB at config 5,
    nothing crosses PCIe), the lagged products pooled over the chains on the device (smcmc_autocorrelation_sums: the
    definition of MakeAutocorrelation.C:127-148), Geyer's initial positive sequence, the worst coordinate."""
    npad = h.nchains_padded
    trace = torch.empty((nsteps, h.dim, npad), dtype=torch.float64, device="cuda")
    for k in range(nsteps):
        h.Step(1)
        h.copy_positions(trace[k].data_ptr())
    torch.cuda.synchronize()
    q, _, _ = h.state()
    tau = h.AutocorrelationSums(trace.data_ptr(), nsteps, centre=q.mean(axis=1),
                                stream=torch.cuda.current_stream().cuda_stream).tau()
    del trace
    return 1.0 / float(tau.max())


def extra_hmc(pkg, torch, stream, dim, chains, leapfrog, exact, steps, tuned, burn=2, eps0=None, ess=False, diagonal=False):
    """Config 5: TSimpleHMC, header-form TDummyLogLikelihood with its analytic gradient, start at 1 (SimpleHMC.C:45),
    SetLeapFrog(20).  tuned: the step length is left to the chain (the covariance fold and the pooled UpdateErrorMatrix run
    every step, inside the timed region) -- from the reference's start value 0.05 (TSimpleHMC.H:229: 25 times the
    stability limit of this target, every trajectory is rejected and the step length never moves), or, with eps0, from
    SetMeanEpsilon(eps0 > 0), a start the target can take; otherwise SetMeanEpsilon(< 0) fixes it.  `burn` untimed
    steps first; accept_rate is the acceptance inside the timed region."""
    # diagonal: a quadratic form the reference's own tuning can handle (variances 0.25 .. 4) next to the header form, whose
    # rho = 0.999999 pair needs epsilon < 0.002 while UpdateErrorMatrix never sets less than 0.5 * 0.01 (TSimpleHMC.H:825-839)
    error = np.diag(1.0 / np.linspace(0.25, 4.0, dim)) if diagonal else tdummy_error(dim)
    h = pkg.HmcEngine(dim, chains, likelihood=pkg.LIKE_QUADFORM, likelihood_params=error, seed=20240607,
                      exact=exact, stream=stream.cuda_stream)
    h.Start(np.ones(dim))
    if not tuned:
        h.SetMeanEpsilon(-0.0005)
    elif eps0 is not None:
        h.SetMeanEpsilon(eps0)
    h.SetLeapFrog(leapfrog)
    gc.collect(); gc.disable()
    h.Step(burn)
    torch.cuda.synchronize()
    acc0, tr0 = h.lane("naccept").astype(np.float64).sum(), h.lane("trials").astype(np.float64).sum()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    h.Step(steps)
    e1.record(stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gc.enable()
    rate = chains * steps / dt
    flops = (leapfrog + 1) * 2 * dim * dim + 3 * dim * dim + dim * dim
    acc1, tr1 = h.lane("naccept").astype(np.float64).sum(), h.lane("trials").astype(np.float64).sum()
    eps = h.lane("mean_epsilon")
    out = {"workload": "TSimpleHMC %s D=%d, %d chains x %d leapfrog steps, %s step length" %
                       ("quadratic form with variances 0.25..4" if diagonal else "header-form TDummy", dim, chains, leapfrog,
                        ("self-tuned from SetMeanEpsilon(%g)" % eps0 if eps0 is not None else
                                                "self-tuned from the reference's start value") if tuned else "fixed"),
           "trajectories_per_s": rate, "ms_per_step": dt / steps * 1e3, "device_ms_per_step": e0.elapsed_time(e1) / steps,
           "steps": steps, "burn_in_steps": burn, "arithmetic": "reference-order" if exact else "fused (matrix pipe)",
           "fp64_TFLOPs": rate * flops / 1e12, "fp64_frac": rate * flops / 1e12 / FP64_PEAK_TFLOPS,
           "hbm_model_frac": rate * (32 * dim + 16) / 1e9 / HBM_PEAK_GBPS,
           "accept_rate": float((acc1 - acc0) / max(tr1 - tr0, 1.0)),
           "mean_epsilon_chain0": h.GetMeanEpsilon(), "mean_abs_epsilon": float(np.abs(eps).mean()),
           "covariance_updates": h.tuning["updates"]}
    if ess:
        try:
            per = hmc_ess_per_trajectory(h, torch)
            out["ess_trajectories"] = 256
            out["ess_per_trajectory"] = per
            out["ess_per_s"] = per * rate
        except Exception as exc:
            out["ess_error"] = repr(exc)
    h.close()
    return out


def cpp_step_loop(dim, cycles, steps, runahead, save=False):
    """The unchanged caller: examples/StepLoop_amd.C (the loop of SimpleMCMC.C:176-243 over include/TSimpleMCMC_amd.H, one
    chain, Step() one call at a time, the driver's getters and per-cycle UpdateProposal() + setters) compiled with g++ and
    run as a child process; Step() calls per second of its timed loop."""
    import subprocess
    import tempfile
    libdir = os.path.join(ROOT, "root-simple-mcmc_amd", "lib")
    with tempfile.TemporaryDirectory() as tmp:
        exe = os.path.join(tmp, "step_loop.exe")
        cmd = ["g++", "-std=c++17", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "StepLoop_amd.C"),
               "-L" + libdir, "-lsmcmc_amd", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            return {"error": r.stderr[-400:]}
        r = subprocess.run([exe, str(dim), str(cycles), str(steps), "1" if save else "0", "1" if runahead else "0"],
                           capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            return {"error": (r.stdout + r.stderr)[-400:]}
        words = r.stdout.split()
        return {"workload": "SimpleMCMC.C:176-243 over TSimpleMCMC_amd.H: one chain, D=%d README-form TDummyLogLikelihood, per-chain "
                            "adaptation, %d cycles x %d Step(%s) calls with the driver's prints, UpdateProposal() and setters per cycle"
                            % (dim, cycles, steps, "true" if save else "false"),
                "steps_per_s": float(words[words.index("steps_per_s") + 1]),
                "step_runs_ahead": bool(int(words[words.index("run_ahead") + 1])),
                "moved": int(words[words.index("moved") + 1])}


def bench_c5(args, pkg, torch, dist, rank, world, local, stream, leapfrog=20):
    """BASELINE config 5 over N GPUs: every rank runs its shard of TSimpleHMC chains (quadratic form with variances
    0.25 .. 4, a target the reference's own tuning handles; start at 1, SimpleHMC.C:45; SetLeapFrog(20); every chain
    tuning its own step length from SetMeanEpsilon(0.05)), the covariance fold runs every trajectory, and every `window`
    trajectories the moments are all-reduced and UpdateCovariance / UpdateErrorMatrix run on every rank on the same
    bits (distributed.run_windows over HmcBackend).  A "step" of the contract is one such window."""
    dim, chains = args.dim, args.chains
    h = pkg.HmcEngine(dim, chains, likelihood=pkg.LIKE_QUADFORM, likelihood_params=np.diag(1.0 / np.linspace(0.25, 4.0, dim)),
                      seed=20240607, chain_offset=rank * chains, device=local, exact=not args.fast, stream=stream.cuda_stream)
    h.Start(np.ones(dim))
    h.SetMeanEpsilon(0.05)
    h.SetLeapFrog(leapfrog)
    backend = pkg.distributed.HmcBackend(h, time_steps=True, stream=stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    gc.collect(); gc.disable()    # (before the warm-up: see main())
    pkg.distributed.run_windows(backend, args.warmup, args.window)
    backend.events.clear(); backend.comm_events.clear()
    acc0, tr0 = h.lane("naccept").astype(np.float64).sum(), h.lane("trials").astype(np.float64).sum()
    fence()
    t0 = time.perf_counter()
    pkg.distributed.run_windows(backend, args.steps, args.window)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    acc1, tr1 = h.lane("naccept").astype(np.float64).sum(), h.lane("trials").astype(np.float64).sum()
    kms = float(np.mean([a.elapsed_time(b) for a, b in backend.events]))
    kernel_ms_per_rank = None
    if world > 1:
        mine = torch.zeros(world, dtype=torch.float64, device="cuda")
        mine[rank] = kms
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        kernel_ms_per_rank = [float(v) for v in mine.cpu()]
    rate = float(chains) * world * args.window * args.steps / dt
    flops = (leapfrog + 1) * 2 * dim * dim + 3 * dim * dim + dim * dim      # SURVEY.md 8(d), per trajectory
    per_launch = float(chains) * args.window
    out = {
        "metric": "trajectories/s, TSimpleHMC D=500, 8 192 chains per GPU x 20 leapfrog steps (BASELINE config 5, sharded)",
        "value": rate, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "rccl_ranks": dist.get_world_size() if world > 1 else 1,
        "ms_allreduce": (float(np.mean([a.elapsed_time(b) for a, b in backend.comm_events])) if backend.comm_events else None),
        "kernel_ms_per_rank": kernel_ms_per_rank,
        "comm": "none (one rank)" if world == 1 else "torch.distributed nccl (= RCCL)",
        "vs_baseline": None, "dtype": "f64", "data": "synthetic", "rng": "philox4x32-7",
        "config": {"workload": "BASELINE config 5: TSimpleHMC, quadratic form with variances 0.25..4 and its analytic gradient, "
                               "D=%d, %d chains/GPU x %d leapfrog steps, self-tuned step length from SetMeanEpsilon(0.05), "
                               "pooled covariance retuning every %d trajectories%s" %
                               (dim, chains, leapfrog, args.window,
                                " (%d chains over %d GPUs, one all-reduce of the moments per window)" % (chains * world, world)
                                if world > 1 else ""),
                   "baseline_config": 5, "dim": dim, "chains_per_gpu": chains, "leapfrog": leapfrog, "window": args.window,
                   "arithmetic": "fused (matrix pipe)" if args.fast else "reference-order", "seed": 20240607},
        "accept_rate": float((acc1 - acc0) / max(tr1 - tr0, 1.0)),
        "mean_abs_epsilon": float(np.abs(h.lane("mean_epsilon")).mean()),
        "covariance_updates": h.tuning["updates"],
        "roofline": {"bound": "mfma", "achieved": per_launch * flops / (kms * 1e-3) / 1e12,
                     "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": per_launch * flops / (kms * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                     "traffic": None,
                     "note": "FP64 vector/matrix peak; flops per trajectory = (L + 1) 2 D^2 + 4 D^2 (SURVEY.md 8(d)); kernel time = "
                             "HIP events around the window's launches on the engine's stream"},
    }
    if rank == 0:
        print(json.dumps(out))
    h.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--chains", type=int, default=CHAINS_PER_GPU)
    ap.add_argument("--window", type=int, default=WINDOW)
    ap.add_argument("--fast", action="store_true", help="fused multiply-add arithmetic (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ess", action="store_true", help="skip the ESS trace after the timed region")
    ap.add_argument("--no-extras", action="store_true", help="skip the other BASELINE configs after the headline")
    ap.add_argument("--frozen", action="store_true", help="diagnostic: frozen covariance, no moment fold")
    ap.add_argument("--dim", type=int, default=DIM, help="diagnostic: other dimension (not the headline)")
    ap.add_argument("--header-tdummy", action="store_true",
                    help="diagnostic: the headline loop on the other C2 likelihood (header-form TDummyLogLikelihood)")
    ap.add_argument("--config", choices=("c2", "c4", "c5"), default="c2",
                    help="c2 (default, the headline): BASELINE config 2, D=50, 65 536 chains per GPU.  c4: BASELINE config 4, "
                         "D=500, 32 768 chains per GPU (262 144 over 8), all-reduce of the pooled covariance every 256 steps; "
                         "README-form likelihood, or the header form with --header-tdummy.  c5: BASELINE config 5 sharded, "
                         "TSimpleHMC D=500, 8 192 chains per GPU x 20 leapfrog steps, the chains tuning their own step length, "
                         "the covariance-driven retuning pooled over all ranks every --window trajectories "
                         "(distributed.HmcBackend through run_windows; default window 8)")
    ap.add_argument("--native-comm", action="store_true",
                    help="N > 1: the moment all-reduce through the library's own RCCL communicator (smcmc_comm_init / "
                         "smcmc_allreduce_moments, the C / C++ callers' path) instead of torch.distributed")
    args = ap.parse_args()
    if args.config == "c4":
        # the defaults of config 4 where the command line left the config-2 ones
        if args.dim == DIM: args.dim = 500
        if args.chains == CHAINS_PER_GPU: args.chains = 32768
    if args.config == "c5":
        if args.dim == DIM: args.dim = 500
        if args.chains == CHAINS_PER_GPU: args.chains = 8192
        if args.window == WINDOW: args.window = 8

    import torch
    from smcmc_amd_loader import load_package
    pkg = load_package()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as typed: this process becomes the launcher of N rank processes, one per GPU, and
        # stays off the GPU itself (the GPUs are counted from the KFD topology, not through the HIP runtime)
        try:
            code, out = pkg.distributed.launch_local_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:],
                                                           args.gpus, pkg.distributed.visible_gpus())
        except RuntimeError as exc:
            raise SystemExit("bench.py --gpus %d: %s" % (args.gpus, exc))
        sys.stdout.write(out)
        sys.stdout.flush()
        raise SystemExit(code)

    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (torch.distributed.run --nproc-per-node N), "
                         "or unset WORLD_SIZE and let bench.py launch them" % (args.gpus, world))
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.native_comm:
            # rendezvous, barrier and the max-over-ranks of the timing only (CPU, gloo): the data path is the library's
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))

    stream = torch.cuda.current_stream()
    dim = args.dim
    if args.config == "c5":
        return bench_c5(args, pkg, torch, dist, rank, world, local, stream)
    like_name, like, like_params = "iso", pkg.LIKE_ISO_GAUSS, None
    if args.header_tdummy:
        like_name, like, like_params = "quadform", pkg.LIKE_QUADFORM, tdummy_error(dim)
    eng = pkg.Engine(dim, args.chains, likelihood=like, likelihood_params=like_params, seed=20240607,
                     chain_offset=rank * args.chains, device=local,
                     mode=pkg.MODE_FROZEN if args.frozen else pkg.MODE_POOLED,
                     exact=not args.fast, stream=stream.cuda_stream)
    assert eng.Start(np.zeros(dim))

    # the window loop is the package's own (distributed.run_windows): steps, moment reduction, all-reduce over the
    # ranks when there are several, pooled update; the backend times the step launches with HIP events on its stream
    if world > 1 and args.native_comm:
        ident = [pkg.Engine.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(ident, src=0)
        backend = pkg.distributed.NativeBackend(eng, rank, world, ident[0], frozen=args.frozen, time_steps=True, stream=stream)
    else:
        backend = pkg.distributed.HipBackend(eng, frozen=args.frozen, time_steps=True, stream=stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The collector runs BEFORE the warm-up windows, not between them and the timed region: a full collection of a
    # torch-sized heap takes tens of milliseconds, the idle GPU drops its clocks, and the first timed windows then run
    # 2 - 3 % slow until they are back (20 timed windows: 3.91e9 with the pause there, 60 windows 4.03e9 either way).
    gc.collect(); gc.disable()    # ... and no collector pauses inside the timed region
    pkg.distributed.run_windows(backend, args.warmup, args.window)
    backend.events.clear()
    backend.comm_events.clear()
    fence()
    t0 = time.perf_counter()
    pkg.distributed.run_windows(backend, args.steps, args.window)
    fence()
    dt = time.perf_counter() - t0
    gc.enable()
    rccl_ranks = 1
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if args.native_comm else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # the size of the communicator the moments crossed: the library's own (ncclCommCount) or torch's RCCL group
        rccl_ranks = eng.comm_ranks() if args.native_comm else dist.get_world_size()
    ms_allreduce = (float(np.mean([a.elapsed_time(b) for a, b in backend.comm_events])) if backend.comm_events else None)
    kernel_ms_per_rank = None
    if world > 1:
        # every rank's device time per window of step launches (HIP events on its stream)
        mine = torch.zeros(world, dtype=torch.float64, device="cpu" if args.native_comm else "cuda")
        mine[rank] = float(np.mean([a.elapsed_time(b) for a, b in backend.events]))
        dist.all_reduce(mine, op=dist.ReduceOp.SUM)
        kernel_ms_per_rank = [float(v) for v in mine.cpu()]

    chain_steps = float(args.chains) * world * args.window * args.steps
    value = chain_steps / dt
    kms = float(np.mean([a.elapsed_time(b) for a, b in backend.events]))
    per_launch = float(args.chains) * args.window
    bytes_cs = bytes_per_chain_step(dim)
    achieved = per_launch * bytes_cs / (kms * 1e-3) / 1e9
    flops_cs = flops_per_chain_step(dim, like_name)
    fp64_tflops = per_launch * flops_cs / (kms * 1e-3) / 1e12

    # HBM bytes of one launch from the PMC counters: collected in separate rocprofv3 --pmc passes over this same
    # command (tools/profile_bench.sh) and replayed from the committed summary; only valid for the configuration it was
    # measured on
    traffic, traffic_source = None, None
    if (os.path.exists(TRAFFIC_FILE) and dim == DIM and args.chains == CHAINS_PER_GPU and args.window == WINDOW
            and not args.fast and not args.frozen and not args.header_tdummy):
        traffic = json.load(open(TRAFFIC_FILE))["hbm_bytes_per_launch"]
        traffic_source = "replayed from profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" % os.path.basename(TRAFFIC_FILE)

    # ESS per chain-step (outside the timed region): a trace of the adapted ensemble saved
    # every ESS_STRIDE steps through the device-side save path, autocorrelation per dimension
    # (definition of MakeAutocorrelation.C:127-148) pooled over all chains, Geyer's
    # initial positive sequence, minimum over dimensions.
    ess_per_chain_step = None
    if rank == 0 and not args.no_ess and dim <= 63:   # (the trace of a D = 500 ensemble would not fit)
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
        "rccl_ranks": rccl_ranks, "ms_allreduce": ms_allreduce, "kernel_ms_per_rank": kernel_ms_per_rank,
        "comm": ("none (one rank)" if world == 1 else "library RCCL communicator (smcmc_comm_init / smcmc_allreduce_moments)"
                 if args.native_comm else "torch.distributed nccl (= RCCL)"),
        "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "rng": "philox4x32-7 (counter-based, key = seed, counter = (block, global chain id, step, stream); "
               "include/smcmc_detmath.h SMCMC_PHILOX_ROUNDS)",
        "config": {"workload": ("BASELINE config 4: " if args.config == "c4" else "") +
                               ("TDummyLogLikelihood header form (quadratic form)" if args.header_tdummy else
                                "TDummyLogLikelihood README form (iso-Gaussian)") +
                               " D=%d, %d chains/GPU%s, TProposeAdaptiveStep pooled covariance, window=%d steps/launch"
                               % (dim, args.chains, (" (%d chains sharded over %d GPUs, one all-reduce of the pooled moments "
                                                     "per window)" % (args.chains * world, world)) if args.config == "c4" else "",
                                  args.window),
                   "baseline_config": 4 if args.config == "c4" else 2,
                   "dim": dim, "mode": "frozen" if args.frozen else "pooled", "chains_per_gpu": args.chains, "window": args.window,
                   "arithmetic": "fused" if args.fast else "reference-order", "seed": 20240607, "rng": "philox4x32-7"},
        "accept_rate": float(naccept.sum() / (total_steps * args.chains)),
        "ess_per_chain_step": ess_per_chain_step,
        "ess_per_s": (ess_per_chain_step * value) if ess_per_chain_step else None,
        "mean_sigma": float(eng.lane("sigma").mean()),
        # `achieved` is the north-star's MODEL figure (algorithmic bytes of the state round trip per chain-step / kernel
        # time), not a measured bandwidth: the state stays in LDS / registers for the 256 steps of a launch and the
        # kernel is bound by FP64 instruction issue.  measured_hbm_GBps is what the PMC counters saw.
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBPS,
                     "model": "algorithmic bytes (16 D + 16 per chain-step, SURVEY.md 8d) / kernel time",
                     "traffic": traffic, "traffic_unit": "bytes per launch (2 x FETCH_SIZE + WRITE_SIZE)",
                     "traffic_source": traffic_source,
                     "measured_hbm_GBps": (traffic / (kms * 1e-3) / 1e9) if traffic else None,
                     "fp64_TFLOPs": fp64_tflops, "fp64_issue_frac": fp64_tflops / FP64_PEAK_TFLOPS,
                     "fp64_model": "flops of SURVEY.md 8d (2 D^2 + 3 D) per chain-step / kernel time / 78.6 TFLOP/s",
                     "measured_bound": "instruction issue: one wavefront per SIMD pays ~4 cycles per instruction of any kind and 64 per "
                                       "FP64 matrix instruction; only LDS reads, waits and scalar instructions issue in a matrix "
                                       "instruction's shadow (profiles/r03_notes.md, tools/micro/pipe_overlap.hip)",
                     "limiter": "instruction issue of a lone wavefront per SIMD; see profiles/r03_notes.md",
                     "algorithmic_bytes_per_launch": per_launch * bytes_cs,
                     "kernel": ("step_kernel<%d,%s,%s,tri,moments>" % (dim, "QUADFORM" if args.header_tdummy else "ISO",
                                                                            "fused" if args.fast else "exact")) if dim <= 63 else
                               ("panel_mfma_kernel + fold_ring_kernel" if args.fast else "panel_step_kernel + fold_ring_kernel"),
                     "kernel_ms": kms, "bytes_per_chain_step": bytes_cs, "flops_per_chain_step": flops_cs,
                     "chain_steps_per_launch": per_launch},
    }
    eng.close()
    if rank == 0 and world == 1 and not args.no_extras and args.config == "c2":
        extra = {}
        rng = np.random.default_rng(0)
        try:
            extra["c2_header_tdummy"] = extra_metropolis(
                pkg, torch, stream, "TDummyLogLikelihood header form (quadratic form) D=50, 65 536 chains, pooled", 50,
                CHAINS_PER_GPU, "quadform", pkg.LIKE_QUADFORM, tdummy_error(50), np.zeros(50), True, 20, dense_quadform=True)
            extra["c2_header_tdummy_sparse_walk"] = extra_metropolis(
                pkg, torch, stream, "TDummyLogLikelihood header form D=50, 65 536 chains, pooled; its Error matrix is the "
                "identity plus one correlated pair, and the serial sum walks the 52 non-zero entries", 50,
                CHAINS_PER_GPU, "quadform", pkg.LIKE_QUADFORM, tdummy_error(50), np.zeros(50), True, 20, dense_quadform=False)
            # (first: 2 GB of per-chain state streamed every step is at its best in freshly allocated memory -- after
            # the D = 500 engines below have come and gone the same row reads 395 us per step instead of 370)
            extra["perchain_d50_65536"] = extra_perchain(pkg, torch, stream, 50, 65536, 64, 3)
            extra["perchain_d50_4096"] = extra_perchain(pkg, torch, stream, 50, 4096, 64, 3)
            extra["perchain_d50_65536_header_tdummy"] = extra_perchain(pkg, torch, stream, 50, 65536, 64, 3, header_form=True)
            extra["c2_readme_iso_d50_65536_pooled_fused"] = extra_metropolis(
                pkg, torch, stream, "the headline's workload in the fused order (fma where the reference has multiply + add; "
                "each lane bit for bit the fused-order restatement)", 50, CHAINS_PER_GPU, "iso", pkg.LIKE_ISO_GAUSS, None,
                np.zeros(50), False, 20)
            x3 = rng.uniform(0.5, 1.5, (200, 16384))                  # SimpleMCMC.C:147
            for exact in (True, False):
                extra["c3_rosenbrock_d200_16384_pooled" + ("" if exact else "_fused")] = extra_metropolis(
                    pkg, torch, stream, "THardLogLikelihood (Rosenbrock) D=200, 16 384 chains, pooled", 200, 16384,
                    "rosenbrock", pkg.LIKE_ROSENBROCK, [100.0], x3, exact, 10)
                extra["c4_share_d500_32768_pooled" + ("" if exact else "_fused")] = extra_metropolis(
                    pkg, torch, stream, "TDummyLogLikelihood README form D=500, 32 768 chains (one GPU's share of config 4), "
                    "pooled, sync every 256 steps", 500, 32768, "iso", pkg.LIKE_ISO_GAUSS, None, np.zeros(500), exact, 5)
                extra["c4_share_d500_32768_pooled_stride16" + ("" if exact else "_fused")] = extra_metropolis(
                    pkg, torch, stream, "config 4 share as above with the covariance fed every 16th step (SMCMC_P_MOMENT_STRIDE: "
                    "a thinned running covariance, not the reference's every-step update)", 500, 32768, "iso",
                    pkg.LIKE_ISO_GAUSS, None, np.zeros(500), exact, 5, stride=16)
            extra["c4_share_d500_32768_pooled_header_tdummy"] = extra_metropolis(
                pkg, torch, stream, "TDummyLogLikelihood header form D=500, 32 768 chains, pooled (the likelihood config 4 "
                "names, in the reference's order: one serial D^2-term sum per chain)", 500, 32768, "quadform",
                pkg.LIKE_QUADFORM, tdummy_error(500), np.zeros(500), True, 1, dense_quadform=True, adapt_windows=1)
            extra["c4_share_d500_32768_pooled_header_tdummy_sparse_walk"] = extra_metropolis(
                pkg, torch, stream, "TDummyLogLikelihood header form D=500, 32 768 chains, pooled, reference order; the serial "
                "sum walks the 502 non-zero entries of Error", 500, 32768, "quadform",
                pkg.LIKE_QUADFORM, tdummy_error(500), np.zeros(500), True, 5, dense_quadform=False)
            extra["c4_share_d500_32768_pooled_header_tdummy_fused"] = extra_metropolis(
                pkg, torch, stream, "TDummyLogLikelihood header form D=500, 32 768 chains, pooled", 500, 32768, "quadform",
                pkg.LIKE_QUADFORM, tdummy_error(500), np.zeros(500), False, 5)
            # config 5.  The sampler rows first: a burn-in in which the chains accept, then 200 trajectories timed, ESS from
            # a 256-trajectory device trace.  Fixed step (SetMeanEpsilon(< 0) + SetLeapFrog(20)) on the target config 5
            # names, in both arithmetic orders; the reference's own tuning on a target it can tune.
            extra["c5_hmc_d500_8192_L20_fixed_step_fused"] = extra_hmc(pkg, torch, stream, 500, 8192, 20, False, 200, False,
                                                                       burn=100, ess=True)
            extra["c5_hmc_d500_8192_L20_fixed_step"] = extra_hmc(pkg, torch, stream, 500, 8192, 20, True, 100, False, burn=100)
            extra["c5_hmc_d500_8192_L20_tuned_sampler_diag_target_fused"] = extra_hmc(
                pkg, torch, stream, 500, 8192, 20, False, 200, True, burn=300, ess=True, diagonal=True)
            # As the reference tunes it: from its start value 0.05 on the header form's rho = 0.999999 pair every
            # trajectory is rejected and UpdateErrorMatrix never sets a step below 0.5 x 0.01 (TSimpleHMC.H:825-839) -- the
            # chains stand still, bit for bit as the CPU restatement does.  Throughput rows of that path, nothing more.
            extra["c5_as_reference_tunes"] = {
                "note": "accept_rate 0: the reference's own tuning fails on this target (2.5 x its stability limit); kept as "
                        "the timing of the adaptive path (covariance fold + pooled UpdateErrorMatrix every step)",
                "c5_hmc_d500_8192_L20": extra_hmc(pkg, torch, stream, 500, 8192, 20, True, 100, True),
                "c5_hmc_d500_8192_L20_fused": extra_hmc(pkg, torch, stream, 500, 8192, 20, False, 100, True)}
            # the drop-in loop: Step() one call at a time from a C++ caller, beside cpu_baseline (the same loop on one host core)
            extra["cpp_step_loop_d50"] = cpp_step_loop(50, 4, 20000, True)
            extra["cpp_step_loop_d5"] = cpp_step_loop(5, 4, 50000, True)
            extra["cpp_step_loop_d50_save_every_step"] = cpp_step_loop(50, 4, 20000, True, save=True)
            extra["cpp_step_loop_d50_one_launch_per_call"] = cpp_step_loop(50, 1, 3000, False)
        except Exception as exc:   # never a reason to lose the headline
            extra["error"] = repr(exc)
        out["extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
