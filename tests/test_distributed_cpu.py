"""CPU tests of the multi-GPU host logic with a real 2-process gloo group: shard ranges tile the batch,
the union of shards equals the single-process batch (RNG keyed by the global sample index), and the
all-reduced moments equal np.mean / np.var of the whole batch.  Local energies come from the oracle here;
on the GPU box the same host code runs with the HIP path and the RCCL transport."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT
from rnnwavefunctions_amd import distributed as D


def test_shard_ranges_tile_the_batch():
    for total in (1, 7, 10000, 262144):
        for world in (1, 2, 3, 8):
            pieces = [D.shard_range(total, r, world) for r in range(world)]
            assert pieces[0][0] == 0
            assert all(pieces[i][0] + pieces[i][1] == pieces[i + 1][0] for i in range(world - 1))
            assert pieces[-1][0] + pieces[-1][1] == total
            assert max(c for _, c in pieces) - min(c for _, c in pieces) <= 1
    with pytest.raises(ValueError):
        D.shard_range(10, 2, 2)


def test_moments_roundtrip():
    e = np.random.RandomState(0).standard_normal(1000) - 80.0
    mean, var = D.moments_to_energy(D.local_moments(e))
    assert np.isclose(mean, e.mean(), rtol=1e-13) and np.isclose(var, e.var(), rtol=1e-9)
    ec = (e + 1j * 0.01 * e).astype(np.complex64)
    mean, var = D.moments_to_energy(D.local_moments(ec))
    assert np.isclose(mean, ec.astype(np.complex128).mean(), rtol=1e-6)
    assert np.isclose(var, np.var(ec.real.astype(np.float64)), rtol=1e-6)
    assert np.array_equal(D.MomentsAllReduce()(np.arange(4.0)), np.arange(4.0))   # no group: identity


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from oracle import estimators as E
    from oracle import models as M
    from oracle import philox
    from rnnwavefunctions_amd import distributed as DD
    from rnnwavefunctions_amd import params as P
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    N, H, total = 12, 8, 101
    prm = P.scale_kernels(P.init_gru_params([H], seed=111), 2.0)
    offset, count = DD.shard_range(total, rank, world)
    u = philox.uniforms(111, 5, offset, count, N)            # global sample indices offset..offset+count
    s, _ = M.prnn_sample(prm, N, u)
    e = E.ising_local_energies(np.ones(N), 1.0, s, lambda x: M.prnn_log_probability(prm, x))
    m = DD.MomentsAllReduce()(DD.local_moments(e))
    comm = DD.ShardComm.from_torch()                        # the transport of the sharded training loop
    assert (comm.rank, comm.world) == (rank, world)
    m2 = comm.allreduce(DD.local_moments(e))
    g = comm.allreduce_grads({"b/kernel": np.full((3, 2), rank + 1.0), "a/bias": np.arange(4.0) * (rank + 1)})
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), s=s, e=e, m=m, m2=m2, offset=offset, gk=g["b/kernel"], gb=g["a/bias"])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_equals_single_process(tmp_path):
    import torch.multiprocessing as mp
    from oracle import estimators as E
    from oracle import models as M
    from oracle import philox
    from rnnwavefunctions_amd import params as P
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [np.load(tmp_path / ("rank%d.npz" % k)) for k in range(2)]
    N, H, total = 12, 8, 101
    prm = P.scale_kernels(P.init_gru_params([H], seed=111), 2.0)
    s_full, _ = M.prnn_sample(prm, N, philox.uniforms(111, 5, 0, total, N))
    e_full = E.ising_local_energies(np.ones(N), 1.0, s_full, lambda x: M.prnn_log_probability(prm, x))
    assert np.array_equal(np.concatenate([r[0]["s"], r[1]["s"]]), s_full)      # union of shards == one batch
    assert np.allclose(np.concatenate([r[0]["e"], r[1]["e"]]), e_full, rtol=1e-6)
    assert np.array_equal(r[0]["m"], r[1]["m"])                               # every rank holds the global sums
    for k in range(2):
        assert np.array_equal(r[k]["m2"], r[0]["m"])
        assert np.array_equal(r[k]["gk"], np.full((3, 2), 3.0)) and np.array_equal(r[k]["gb"], np.arange(4.0) * 3)
    mean, var = D.moments_to_energy(r[0]["m"])
    assert r[0]["m"][2] == total
    assert np.isclose(mean, e_full.mean(), rtol=1e-6) and np.isclose(var, e_full.var(), rtol=1e-5)


def test_shard_comm_single_process_is_identity():
    c = D.ShardComm()
    assert (c.rank, c.world) == (0, 1)
    assert np.array_equal(c.allreduce([1.0, 2.0]), [1.0, 2.0])
    g = {"w": np.ones((2, 2))}
    assert c.allreduce_grads(g) is g
