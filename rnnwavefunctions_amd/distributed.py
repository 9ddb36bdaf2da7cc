"""Data-parallel sharding of the independent sample batch and the ONE collective of the path.

The reference is single-process (no DP/TP/..., no collective).  Every sample of a VMC step is
an independent exact draw, so the batch shards trivially: rank r of G draws the global samples
[offset_r, offset_r + count_r) (the RNG is keyed by the GLOBAL sample index, so the union is independent
of G) and computes their local energies; one all-reduce of four doubles

        (sum Re E, sum (Re E)^2, n, sum Im E)

gives np.mean / np.var of the whole batch (1DTFIM/TrainingRNN_1DTFIM.py:206-207,
J1J2/TrainingRNN_J1J2.py:281-282).  Two transports for that all-reduce:

  * RCCL over xGMI through the C ABI (rnnwf_comm_init / rnnwf_allreduce_moments) - the GPU path;
  * any initialised ``torch.distributed`` process group (gloo on CPU in the tests; "nccl" is RCCL on ROCm).

torch is launcher plumbing only (rendezvous, barrier); it never touches the compute path.
"""
import os

import numpy as np


def shard_range(global_numsamples, rank, world_size):
    """Contiguous block of global sample indices owned by `rank`: (offset, count).
    Blocks differ by at most one sample and tile [0, global_numsamples) exactly."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside [0, %d)" % (rank, world_size))
    base, rem = divmod(int(global_numsamples), int(world_size))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def local_moments(eloc):
    """(sum Re E, sum (Re E)^2, n, sum Im E) of this rank's local energies, float64."""
    e = np.asarray(eloc)
    re = np.real(e).astype(np.float64)
    im = np.imag(e).astype(np.float64) if np.iscomplexobj(e) else np.zeros(0)
    return np.array([re.sum(), (re * re).sum(), float(re.size), im.sum()], dtype=np.float64)


def moments_to_energy(m):
    """mean (complex if an imaginary part was accumulated) and population variance of Re E."""
    s1, s2, n, si = (float(x) for x in m)
    mean_re = s1 / n
    var = max(s2 / n - mean_re * mean_re, 0.0)
    mean = complex(mean_re, si / n) if si != 0.0 else mean_re
    return mean, var


class MomentsAllReduce:
    """Sums the four moments over all ranks, in place semantics, with either transport."""

    def __init__(self, native=None, group=None):
        self.native = native      # _lib.NativeWavefunction with an initialised RCCL communicator
        self.group = group        # torch.distributed group (None: default group), used when native is None

    def __call__(self, moments):
        m = np.ascontiguousarray(moments, dtype=np.float64)
        if self.native is not None:
            return self.native.allreduce_moments(m)
        import torch
        import torch.distributed as dist
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return m
        t = torch.from_numpy(m.copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t.numpy()


class ShardComm:
    """What a sharded training loop needs from the transport: who am I, and sum-all-reduce of float64 arrays.

    ShardComm()                          single process (identity)
    ShardComm.from_torch(group=None)     any initialised torch.distributed group (gloo on CPU tensors; used by the
                                         2-process test on the one-GPU box, where RCCL cannot place two ranks on one device)
    ShardComm.from_rccl(native, r, w)    RCCL through the C ABI (in-step moments, rnnwf_allreduce_grads / rnnwf_allreduce_f64)
    """

    def __init__(self, rank=0, world=1, allreduce=None, native=None):
        self.rank, self.world = int(rank), int(world)
        self._allreduce = allreduce
        self.native = native          # handle whose RCCL communicator reduces gradients inside the library

    @classmethod
    def from_torch(cls, group=None):
        import torch
        import torch.distributed as dist

        def allreduce(a):
            t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
            return t.numpy()
        return cls(dist.get_rank(group), dist.get_world_size(group), allreduce)

    @classmethod
    def from_rccl(cls, native, rank, world):
        """RCCL through the handle's communicator.  The moments of a `vmc_step` are summed over the ranks inside the step
        (rnnwf_comm_reduce_in_step: in-stream all-reduce before the step's one host synchronisation), so `reduce_moments`
        passes them through; `allreduce` / `allreduce_grads` of anything else go through rnnwf_allreduce_f64."""
        comm = cls(rank, world, None, native=native)
        if int(world) > 1:                # a single process never opened a communicator (init_rccl_from_env): nothing to reduce
            native.comm_reduce_in_step(True)
            comm._moments_in_step = True
        return comm

    def reduce_moments(self, m):
        """The four moments of a vmc_step summed over the ranks.  With the RCCL transport the step itself has already
        summed them on the stream (rnnwf_comm_reduce_in_step), so they pass through; every other array goes through
        `allreduce`, which never passes anything through un-reduced."""
        if self.native is not None and getattr(self, "_moments_in_step", False):
            return np.asarray(m, dtype=np.float64)
        return self.allreduce(m)          # (world 1: the identity; a native comm built without from_rccl: rnnwf_allreduce_f64)

    def allreduce(self, a):
        """Sum of a float64 array over the ranks."""
        if self.world == 1:
            return np.asarray(a, dtype=np.float64)
        if self.native is not None:       # RCCL through the C ABI (rnnwf_allreduce_f64)
            return self.native.allreduce_f64(a)
        if self._allreduce is None:
            raise RuntimeError("ShardComm with world_size %d has no transport" % self.world)
        return self._allreduce(a)

    def allreduce_grads(self, grads):
        """{name: array} summed over ranks (one flat all-reduce, names in sorted order on every rank)."""
        if self.world == 1:
            return grads
        if self.native is not None:
            return self.native.allreduce_grads({k: np.ascontiguousarray(v, dtype=np.float64) for k, v in grads.items()})
        if self._allreduce is None:
            raise RuntimeError("ShardComm with world_size %d has no transport" % self.world)
        names = sorted(grads)
        flat = np.concatenate([np.asarray(grads[k], dtype=np.float64).ravel() for k in names])
        flat = self._allreduce(flat)
        out, off = {}, 0
        for k in names:
            n = grads[k].size
            out[k] = flat[off:off + n].reshape(grads[k].shape)
            off += n
        return out


def init_rccl_from_env(native, dist=None):
    """One process per GPU, launched by ``python -m torch.distributed.run``: rank 0 creates the RCCL unique id,
    the launcher's process group (gloo) carries its 128 bytes to the other ranks, every rank joins."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return rank, world
    uid = native.comm_unique_id()         # every rank: loads RCCL now (before torch, if torch is not in yet); rank 0's id is used
    if dist is None:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group(backend="gloo")
    box = [uid if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    native.comm_init(box[0], rank, world)
    return rank, world
