"""Summaries of the rocprofv3 passes of tools/profile_bench.sh -> profiles/rNN_*.json / .csv.

usage: python tools/summarize_pmc.py gpurun_out/r03 profiles/r03
Picks the newest run directory of every pass; the first launch after Start is left out."""
import csv
import glob
import json
import os
import shutil
import sys

KERNEL = "smcmc::step_kernel<50, 0, true, false, true, false>"


def newest(pattern):
    files = sorted(glob.glob(pattern), key=os.path.getmtime)
    if not files:
        raise SystemExit(f"no file matches {pattern}")
    return files[-1]


def counters(path, kernel=None):
    """{counter: [per-dispatch value]} for the headline kernel (or `kernel`), in dispatch order"""
    kernel = kernel or KERNEL
    out = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"]:
                out.setdefault(row["Counter_Name"], {}).setdefault(int(row["Dispatch_Id"]), 0.0)
                out[row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
    return {k: [v[d] for d in sorted(v)][1:] for k, v in out.items()}


def main(src, dst):
    shutil.copy(newest(f"{src}/stats/*/*_kernel_stats.csv"), f"{dst}_bench_kernel_stats.csv")
    extras = glob.glob(f"{src}/extras/*/*_kernel_stats.csv")
    if extras:   # the pass with every single-GPU config on the line (bench.py without --no-extras)
        shutil.copy(sorted(extras, key=os.path.getmtime)[-1], f"{dst}_bench_extras_kernel_stats.csv")
    if os.path.exists(f"{src}/bench.json"):
        lines = [ln for ln in open(f"{src}/bench.json") if ln.startswith("{")]
        if lines:
            json.dump(json.loads(lines[-1]), open(f"{dst}_bench_line.json", "w"), indent=1)
    fetch = counters(newest(f"{src}/fetch/*/*_counter_collection.csv"))["FETCH_SIZE"]
    write = counters(newest(f"{src}/write/*/*_counter_collection.csv"))["WRITE_SIZE"]
    fb = 1024.0 * sum(fetch) / len(fetch)
    wb = 1024.0 * sum(write) / len(write)
    traffic = {
        "kernel": KERNEL, "workload": "D=50, 65536 chains, 256 steps per launch",
        "raw": {"FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write},
        "fetch_bytes_reported": fb, "fetch_bytes_corrected": 2 * fb, "write_bytes": wb,
        "hbm_bytes_per_launch": 2 * fb + wb,
        "correction": "FETCH_SIZE counts 128-B requests at 64 B on gfx950 (MI355X_MICROARCH.md, HBM section): doubled. "
                      "Cross-check: the launch reads x (26.2 MB) + per-chain columns (6.3 MB) + the moment tiles it keeps "
                      "(14.7 MB) = 47.2 MB and writes the same.",
        "collection": "two separate rocprofv3 passes (--pmc FETCH_SIZE / --pmc WRITE_SIZE with --kernel-trace only) of "
                      "bench.py --steps 4 --warmup 1 (tools/profile_bench.sh); first launch after Start left out",
    }
    json.dump(traffic, open(f"{dst}_pmc_traffic.json", "w"), indent=1)
    sq = counters(newest(f"{src}/sq/*/*_counter_collection.csv"))
    mean = {k: sum(v) / len(v) for k, v in sq.items()}
    steps, waves = 256.0, 1024.0
    wave_cycles = mean["SQ_WAVE_CYCLES"]
    derived = {
        "wave_quad_cycles_per_step": wave_cycles / waves / steps,
        "issuing_fraction": mean["SQ_ACTIVE_INST_ANY"] / wave_cycles,
        "parked_on_waitcnt_fraction": mean["SQ_WAIT_ANY"] / wave_cycles,
        "issue_stall_fraction_(matrix_pipe_busy)": 1.0 - (mean["SQ_ACTIVE_INST_ANY"] + mean["SQ_WAIT_ANY"]) / wave_cycles,
        "valu_instructions_per_wave_step": mean["SQ_INSTS_VALU"] / waves / steps,
        "lds_instructions_per_wave_step": mean["SQ_INSTS_LDS"] / waves / steps,
        "salu_instructions_per_wave_step": mean["SQ_INSTS_SALU"] / waves / steps,
    }
    json.dump({"kernel": KERNEL, "workload": "D=50, 65536 chains, 256 steps per launch (1024 wavefronts, one per SIMD)",
               f"per_launch_mean_of_{len(next(iter(sq.values())))}": mean, "derived": derived,
               "collection": "one rocprofv3 --pmc pass with --kernel-trace only (tools/profile_bench.sh), bench.py --steps 6 "
                             "--warmup 2; SQ_* cycle counters are in units of 4 cycles, summed over wavefronts; "
                             "GRBM_GUI_ACTIVE summed over the 8 XCDs"},
              open(f"{dst}_pmc_sq.json", "w"), indent=1)
    print(json.dumps({"hbm_bytes_per_launch": 2 * fb + wb, **derived}, indent=1))
    perchain(src, dst)


def perchain(src, dst):
    """SMCMC_MODE_PER_CHAIN: measured HBM bytes per chain-step of perchain_step_kernel against the algorithmic ones."""
    pf, pw = glob.glob(f"{src}/pc_fetch/*/*_counter_collection.csv"), glob.glob(f"{src}/pc_write/*/*_counter_collection.csv")
    if os.path.exists(f"{src}/perchain.json"):
        shutil.copy(f"{src}/perchain.json", f"{dst}_perchain.json")
    for name in ("perchain_wave.json", "fold_bench.txt", "fold_prof.txt", "ordered_sum.txt"):
        if os.path.exists(f"{src}/{name}"):
            shutil.copy(f"{src}/{name}", f"{dst}_{name}")
    if not pf or not pw:
        return
    name = "perchain_step_kernel"
    fetch = counters(sorted(pf, key=os.path.getmtime)[-1], name)["FETCH_SIZE"]
    write = counters(sorted(pw, key=os.path.getmtime)[-1], name)["WRITE_SIZE"]
    dim, chains, steps = 50, 65536, 16
    fb, wb = 1024.0 * sum(fetch) / len(fetch), 1024.0 * sum(write) / len(write)
    alg = 8 * 3 * dim * (dim + 1) // 2 + 8 * 7 * dim + 16
    out = {"kernel": "smcmc::perchain_step_kernel<0>", "workload": f"D={dim}, {chains} chains, {steps} steps per launch",
           "raw": {"FETCH_SIZE_KB_per_launch": fetch, "WRITE_SIZE_KB_per_launch": write},
           "fetch_bytes_corrected": 2 * fb, "write_bytes": wb,
           "hbm_bytes_per_chain_step": (2 * fb + wb) / (chains * steps), "algorithmic_bytes_per_chain_step": alg,
           "read_bytes_per_chain_step": 2 * fb / (chains * steps), "write_bytes_per_chain_step": wb / (chains * steps),
           "algorithmic_read": 8 * 2 * dim * (dim + 1) // 2 + 8 * 5 * dim + 8, "algorithmic_write": 8 * dim * (dim + 1) // 2 + 8 * 2 * dim + 8,
           "correction": "FETCH_SIZE doubled (gfx950: 128-B requests counted at 64 B), as for the headline kernel",
           "collection": "separate rocprofv3 --pmc passes of tools/perchain_time.py --chains 65536 --steps 16 --launches 2; the "
                         "first launch (8 warm-up steps) left out"}
    json.dump(out, open(f"{dst}_perchain_traffic.json", "w"), indent=1)
    print(json.dumps({k: out[k] for k in ("hbm_bytes_per_chain_step", "algorithmic_bytes_per_chain_step")}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
