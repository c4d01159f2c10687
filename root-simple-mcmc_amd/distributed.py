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
        self.time_steps, self.stream, self.events, self.comm_events = time_steps, stream, [], []

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


class HmcBackend:
    """Adapter for the HMC engine (config 5 sharded): `window` trajectories of every local chain with the covariance
    fold running, then the pooled UpdateCovariance / UpdateErrorMatrix on the all-reduced moments
    (smcmc_hmc_reduce_moments / export / import / apply).  The engine's own sync interval is set beyond any window, so
    the only updates are the ones run_windows makes -- on every rank the same update of the same bits."""

    def __init__(self, engine, time_steps=False, stream=None):
        import torch
        self.engine = engine
        self.frozen = False
        engine.SetSyncInterval(1 << 30)
        self.buffer = torch.zeros(engine.moments_size, dtype=torch.float64, device="cuda")
        self.time_steps, self.stream, self.events, self.comm_events = time_steps, stream, [], []

    step = HipBackend.step
    moments_out = HipBackend.moments_out
    local_update = HipBackend.local_update
    moments_in = HipBackend.moments_in


class NativeBackend:
    """The same window loop with the exchange inside the C library: smcmc_comm_init attaches an RCCL communicator to the
    engine (include/smcmc.h) and the all-reduce of the moments is ncclAllReduce on the engine's stream -- the path a C++
    caller of TSimpleMCMC_amd.H uses (InitComm / SyncPooledCovariance), no torch tensor in the data path."""

    def __init__(self, engine, rank, world, unique_id, frozen=False, time_steps=False, stream=None):
        self.engine, self.frozen = engine, frozen
        self.time_steps, self.stream, self.events, self.comm_events = time_steps, stream, [], []
        self.world = world
        engine.comm_init(unique_id, rank, world)

    native = True
    step = HipBackend.step

    def local_update(self):
        e = self.engine
        e.reduce_moments()
        if self.time_steps:
            import torch
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(self.stream)
            e.allreduce_moments()
            e1.record(self.stream)
            self.comm_events.append((e0, e1))
        else:
            e.allreduce_moments()
        e.apply_moments()


def rank_environments(nranks, port, base_env=None, addr="127.0.0.1"):
    """The environment of each of `nranks` local rank processes (what torch.distributed.run would set): RANK,
    LOCAL_RANK, WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR, MASTER_PORT on top of `base_env`.  HSA_ENABLE_IPC_MODE_LEGACY=0
    is kept / set: RCCL's intra-node transport needs dmabuf IPC on this driver."""
    if nranks < 1:
        raise ValueError("nranks must be positive")
    envs = []
    for r in range(nranks):
        env = dict(base_env or {})
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks), "LOCAL_WORLD_SIZE": str(nranks),
                    "MASTER_ADDR": addr, "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        envs.append(env)
    return envs


def free_port(addr="127.0.0.1"):
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((addr, 0))
        return s.getsockname()[1]


def visible_gpus():
    """GPUs of this node without touching the HIP runtime (the launcher parent must stay off the GPU): the visibility
    variables if they are set, else the KFD topology (nodes with SIMDs are GPUs), else torch's count (which, on this
    image, does not initialise the runtime either)."""
    import os
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    nodes = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for d in os.listdir(nodes):
            with open(os.path.join(nodes, d, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        if n > 0:
            return n
    except (OSError, ValueError):
        pass
    import torch
    return torch.cuda.device_count()


def launch_local_ranks(argv, nranks, visible_devices, timeout=None):
    """Start `nranks` fresh processes of `argv` (one per GPU of this node), rank r on device r, and return
    (exit code, rank 0's stdout).  The caller must not have touched the GPU: the children are plain child processes of
    a GPU-free parent (never a re-exec of a process that holds a HIP context).  visible_devices: how many GPUs the
    node has -- fewer than nranks is refused before anything starts."""
    import os
    import subprocess
    if visible_devices < nranks:
        raise RuntimeError("%d ranks asked for, %d GPU(s) visible on this node: one rank per GPU" % (nranks, visible_devices))
    envs = rank_environments(nranks, free_port(), os.environ)
    import threading
    import time
    procs = []
    for r, env in enumerate(envs):
        procs.append(subprocess.Popen(argv, env=env, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's line is read on the side, while EVERY child is watched: a rank that dies during rendezvous or RCCL
    # initialisation would leave the others waiting in a collective for ever
    out_box = []
    reader = threading.Thread(target=lambda: out_box.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = None if timeout is None else time.monotonic() + timeout
    code = 0
    while True:
        states = [p.poll() for p in procs]
        bad = [c for c in states if c not in (None, 0)]
        if bad:
            code = bad[0]
            break
        if all(c == 0 for c in states):
            break
        if deadline is not None and time.monotonic() > deadline:
            code = 124
            break
        time.sleep(0.05)
    if code != 0:
        for p in procs:          # exactly the processes this function started
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=5)
    out = out_box[0] if out_box else b""
    return code, out.decode() if out else ""


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
    timed = distributed and getattr(backend, "time_steps", False) and getattr(backend, "stream", None) is not None
    for _ in range(nwindows):
        backend.step(window)
        if getattr(backend, "frozen", False):
            continue
        if getattr(backend, "native", False) or (not distributed and hasattr(backend, "local_update")):
            backend.local_update()        # nothing to exchange, or the exchange is the library's own (NativeBackend)
            continue
        m = backend.moments_out()
        if distributed:
            if timed:
                import torch
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(backend.stream)
            dist.all_reduce(m, op=dist.ReduceOp.SUM, group=group)
            if timed:
                e1.record(backend.stream)
                backend.comm_events.append((e0, e1))
        backend.moments_in(m)
