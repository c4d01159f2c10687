"""Step() calls per second of the unchanged caller (examples/StepLoop_amd.C), the rows bench.py reports."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
for dim, cycles, steps, ahead, save in ((50, 20, 5000, True, False), (50, 20, 5000, True, True), (5, 20, 5000, True, False),
                                        (50, 4, 2000, False, False)):
    r = bench.cpp_step_loop(dim, cycles, steps, ahead, save)
    print(json.dumps({k: r[k] for k in r if k != "workload"}), "D=%d %dx%d ahead=%d save=%d" % (dim, cycles, steps, ahead, save))
