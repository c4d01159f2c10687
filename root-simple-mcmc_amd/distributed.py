"""Chain ensembles partitioned over ranks (one process per GPU).

The path shards naturally: chains are independent between adaptation points (the
reference runs them as separate OS processes, continue-chain.sh).  The only
exchange step is one sum all-reduce (RCCL over xGMI through torch.distributed's
"nccl" backend) of the packed moment vector M[(D+1)(D+2)/2] per window; every rank
then applies the identical pooled update, so the decompositions stay bit-identical
with no broadcast.  Chain c of rank r draws from the Philox stream of global chain
r * chains_per_rank + c, so results do not depend on how the ensemble is cut.
"""
import numpy as np


class HipBackend:
    """Adapter: one Engine on this rank's GPU, moments exchanged through a device tensor.

    frozen: the covariance does not adapt, so there is nothing to exchange (run_windows only steps).
    time_steps: bracket every step launch with HIP events on `stream` (the torch stream the engine launches on);
    `events` then holds one (start, end) pair per window for the caller to read after a synchronize."""

    def __init__(self, engine, frozen=False, time_steps=False, stream=None):
        import torch
        self.engine = engine
        self.frozen = frozen
        self.buffer = torch.zeros(engine.moments_size, dtype=torch.float64, device="cuda")
        self.time_steps, self.stream, self.events = time_steps, stream, []

    def step(self, nsteps):
        if not self.time_steps:
            self.engine.Step(nsteps)
            return
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(self.stream)
        self.engine.Step(nsteps)
        e1.record(self.stream)
        self.events.append((e0, e1))

    def moments_out(self):
        self.engine.reduce_moments()
        self.engine.export_moments(self.buffer.data_ptr())
        return self.buffer

    def local_update(self):
        """One rank: nothing to exchange, the moments go from the reduction straight into the update."""
        self.engine.reduce_moments()
        self.engine.apply_moments()

    def moments_in(self, tensor):
        if tensor.data_ptr() != self.buffer.data_ptr():
            self.buffer.copy_(tensor)
        self.engine.import_moments(self.buffer.data_ptr())
        self.engine.apply_moments()


def shard(nchains_total, rank, world):
    """(first global chain id, number of chains) of `rank`; chains_per_rank is kept a
    multiple of 64 (one wavefront = 64 chains) except on the last rank."""
    per = -(-nchains_total // world)
    per = -(-per // 64) * 64
    first = min(rank * per, nchains_total)
    return first, max(0, min(per, nchains_total - first))


def run_windows(backend, nwindows, window, group=None):
    """nwindows x { `window` steps of every local chain; pooled moment all-reduce;
    UpdateProposal on every rank }."""
    import torch.distributed as dist
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    for _ in range(nwindows):
        backend.step(window)
        if getattr(backend, "frozen", False):
            continue
        if not distributed and hasattr(backend, "local_update"):
            backend.local_update()
            continue
        m = backend.moments_out()
        if distributed:
            dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)
        backend.moments_in(m)
