#!/usr/bin/env python3
"""Probe: two RCCL ranks on ONE device (the only multi-rank RCCL run a one-GPU box allows, if RCCL accepts it).
Exercises rnnwf_comm_unique_id -> gloo broadcast -> rnnwf_comm_init -> rnnwf_allreduce_moments exactly as bench.py
and training._resolve_comm do.  Prints what happened; a 'duplicate GPU' refusal from RCCL is an expected outcome."""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      GLOO_SOCKET_IFNAME="lo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    import numpy as np
    from rnnwavefunctions_amd import _lib
    from rnnwavefunctions_amd import params as P
    wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, 10, 1, (10,), device=0)
    wf.set_params(P.init_gru_params([10], seed=1), scope="RNNwavefunction")
    uid = wf.comm_unique_id()
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    box = [uid if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    try:
        wf.comm_init(box[0], rank, world)
        m = wf.allreduce_moments(np.array([1.0 + rank, 2.0, 3.0, 4.0 * rank]))
        print("rank %d: RCCL all-reduce over %d ranks on one device -> %s" % (rank, world, m.tolist()), flush=True)
    except Exception as e:                                   # noqa: BLE001 - a probe reports, it does not judge
        print("rank %d: RCCL refused: %s" % (rank, e), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(worker, args=(2, port), nprocs=2, join=True)
