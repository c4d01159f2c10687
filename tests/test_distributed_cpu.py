"""The N > 1 path on the CPU: two gloo ranks drive the sharding / all-reduce logic of
root-simple-mcmc_amd/distributed.py with the CPU oracle as the per-rank engine
(the HIP engine needs a GPU; the exchange logic is the same code)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIM, NCHAINS, WINDOW, NWIN = 6, 128, 12, 4


class OracleBackend:
    def __init__(self, ens):
        self.ens = ens
        self.buffer = torch.zeros(ens.npacked, dtype=torch.float64)

    def step(self, n):
        self.ens.step(n)

    def moments_out(self):
        self.buffer.copy_(torch.from_numpy(self.ens.reduce_moments()))
        return self.buffer

    def moments_in(self, t):
        self.ens.apply_moments(t.numpy().copy())


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, nchains=NCHAINS):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from smcmc_amd_loader import load_package
    pkg = load_package()
    from root_simple_mcmc_amd import distributed as D
    first, count = D.shard(nchains, rank, world)
    ens = O.Ensemble(count, DIM, chain_offset=first, mode=O.MODE_POOLED)
    assert ens.start(np.zeros(DIM))
    D.run_windows(OracleBackend(ens), NWIN, WINDOW)
    ens.step(3)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=ens.x, u=ens.decomposition, cov=ens.covariance,
             first=first, count=count, sigma=ens.lane("sigma"))
    dist.barrier()
    dist.destroy_process_group()
    del pkg


def test_shard_covers_every_chain_once(smcmc):
    from root_simple_mcmc_amd import distributed as D
    for total, world in ((65536, 8), (262144, 8), (1000, 3), (64, 2), (130, 4)):
        spans = [D.shard(total, r, world) for r in range(world)]
        assert sum(c for _, c in spans) == total
        pos = 0
        for first, count in spans:
            if count:
                assert first == pos and first % 64 == 0
                pos += count


def test_two_ranks_equal_one_process(oracle, smcmc, tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # every rank ends with the bit-identical pooled proposal (no broadcast needed)
    assert np.array_equal(r0["u"], r1["u"]) and np.array_equal(r0["cov"], r1["cov"])
    # and the sharded ensemble is the unsharded one (64 chains per rank: the group
    # sums add in the same order as the single-process reduction)
    from root_simple_mcmc_amd import distributed as D
    whole = oracle.Ensemble(NCHAINS, DIM, mode=oracle.MODE_POOLED)
    assert whole.start(np.zeros(DIM))
    D.run_windows(OracleBackend(whole), NWIN, WINDOW)
    whole.step(3)
    assert np.array_equal(np.concatenate([r0["x"], r1["x"]], axis=1), whole.x)
    assert np.array_equal(r0["u"], whole.decomposition)
    assert np.array_equal(np.concatenate([r0["sigma"], r1["sigma"]]), whole.lane("sigma"))


def test_four_ranks_with_an_uneven_last_shard(oracle, smcmc, tmp_path):
    """world_size 4, 232 chains: shards of 64, 64, 64 and 40 (the last rank's group is ragged).  Every rank ends with the
    bit-identical pooled proposal; against the unsharded ensemble the moments are the same sums added in another order
    (gloo's ring instead of the single process's group order), so that comparison carries a rounding tolerance."""
    world, total = 4, 232
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), total), nprocs=world, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(world)]
    assert [int(x["count"]) for x in r] == [64, 64, 64, 40]
    for x in r[1:]:
        assert np.array_equal(x["u"], r[0]["u"]) and np.array_equal(x["cov"], r[0]["cov"])
    from root_simple_mcmc_amd import distributed as D
    whole = oracle.Ensemble(total, DIM, mode=oracle.MODE_POOLED)
    assert whole.start(np.zeros(DIM))
    D.run_windows(OracleBackend(whole), NWIN, WINDOW)
    whole.step(3)
    assert np.allclose(r[0]["u"], whole.decomposition, rtol=1e-11, atol=1e-13)
    x = np.concatenate([k["x"] for k in r], axis=1)
    assert x.shape == whole.x.shape
    assert np.allclose(x, whole.x, rtol=1e-9, atol=1e-11)
