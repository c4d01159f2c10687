"""Prints the headline and the extras of a bench.py JSON line in a few columns."""
import json, sys
o = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = o["roofline"]
print("headline %.4g chain-steps/s  window %.3f ms  kernel %.3f ms  hbm-model frac %.3f  fp64 frac %.3f" %
      (o["value"], o["ms_per_step"], r["kernel_ms"], r["frac"], r["fp64_issue_frac"]))
for k, v in o.get("extra", {}).items():
    if not isinstance(v, dict):
        print(k, v); continue
    rate = v.get("chain_steps_per_s", v.get("trajectories_per_s"))
    print("%-52s %.4g /s  %8.4f ms/step  fp64 %.3f  hbm-model %.4f  accept %.3f  %s" %
          (k, rate, v.get("ms_per_ensemble_step", v.get("ms_per_step")), v["fp64_frac"], v["hbm_model_frac"], v["accept_rate"], v["arithmetic"]))
if "cpu_baseline" in o:
    c = o["cpu_baseline"]; print("cpu", c["value"], c["cores"], c.get("all_cores"))
