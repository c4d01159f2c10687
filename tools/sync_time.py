"""Cost of one pooled update (smcmc_sync = reduce + apply) on the device path and on the host path.
usage: python tools/sync_time.py [out.json]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from smcmc_amd_loader import load_package  # noqa: E402


def run(pkg, torch, dim, chains, device_update, reps=6):
    stream = torch.cuda.Stream()
    e = pkg.Engine(dim, chains, mode=pkg.MODE_POOLED, exact=False, stream=stream.cuda_stream)
    e.set_param("DEVICE_UPDATE", 1 if device_update else 0)
    assert e.Start(np.zeros(dim))
    e.Step(3); e.sync(); e.Step(1)
    torch.cuda.synchronize()
    on_stream, wall = [], []
    for _ in range(reps):
        e.Step(2)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        a.record(stream); e.sync(); b.record(stream)
        t1 = time.perf_counter()                     # the call returned
        e.Step(1)                                    # the next launch waits for the status word in parity mode
        torch.cuda.synchronize()
        on_stream.append(a.elapsed_time(b)); wall.append((t1 - t0) * 1e3)
    out = {"dim": dim, "chains": chains, "path": "device" if device_update else "host",
           "sync_ms_on_stream": float(np.median(on_stream)), "sync_call_ms_host": float(np.median(wall))}
    e.close()
    return out


def main():
    import torch
    pkg = load_package()
    pkg.load()
    rows = []
    for dim, chains in ((50, 65536), (200, 16384), (500, 32768)):
        for dev in (True, False):
            rows.append(run(pkg, torch, dim, chains, dev))
            r = rows[-1]
            print(f"D={r['dim']:4d} N={r['chains']:6d} {r['path']:6s} update: {r['sync_ms_on_stream']:8.3f} ms on the stream, "
                  f"the call returns after {r['sync_call_ms_host']:8.3f} ms")
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
