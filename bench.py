#!/usr/bin/env python3
"""bench.py - samples*sites/sec of the VMC hot path (autoregressive sample + local energy) on MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path over one batch: `numsamples` configurations are drawn by the
RNN wave function and their local energies and energy moments are computed, everything resident in
HBM (BASELINE.json metric).  The headline workload (N=1) is BASELINE config 2:
1D TFIM, pRNN, N=80, num_units=50, numsamples=10000.  With several GPUs every rank runs the same
per-GPU batch (weak scaling) on its own shard of global sample indices and the moments are summed
with ONE RCCL all-reduce per step.  torch.distributed (gloo, CPU) is used only as the launcher's
rendezvous/barrier; no torch.cuda object is ever created - the compute path is ctypes -> C ABI -> HIP.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "cfg2": dict(kind="tfim1d", N=80, H=50, ns=10000, Bx=1.0,
                 desc="1DTFIM pRNN N=80 num_units=50 numsamples=10000 (BASELINE config 2)"),
    "cfg1": dict(kind="tfim1d", N=20, H=20, ns=500, Bx=1.0,
                 desc="1DTFIM pRNN N=20 num_units=20 numsamples=500 (BASELINE config 1)"),
    "cfg3": dict(kind="j1j2", N=40, H=50, ns=10000, J2=0.5,
                 desc="J1J2 cRNN N=40 J2=0.5 U(1) mask num_units=50 numsamples=10000 (BASELINE config 3)"),
    "cfg4": dict(kind="tfim2d", Nx=12, Ny=12, N=144, H=50, ns=10000, Bx=3.0,
                 desc="2DTFIM_2DRNN 12x12 MDRNNcell num_units=50 numsamples=10000, f64 (BASELINE config 4)"),
    "2d1drnn": dict(kind="tfim2d_gru", Nx=12, Ny=12, N=144, H=50, ns=10000, Bx=3.0,
                    desc="2DTFIM_1DRNN 12x12 GRU over the raster path num_units=50 numsamples=10000, f64 (not a BASELINE config)"),
    "cfg5": dict(kind="tfim1d", N=200, H=100, ns=32768, Bx=1.0,
                 desc="1DTFIM pRNN N=200 num_units=100 numsamples=32768 per GPU (BASELINE config 5 shard)"),
}
# /opt/skills/guides/MI355X_MICROARCH.md: dense f32 MFMA = f32 vector peak; f64 matrix = f64 vector peak
PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6}


def f_cell_gru(h):
    """Dense flops of one GRU cell evaluation incl. Dense(2): 6h^2 + 16h."""
    return 6 * h * h + 16 * h


def f_cell(wl):
    h = wl["H"]
    if wl["kind"] == "j1j2":
        return f_cell_gru(h) + 4 * h          # second Dense(2) head
    if wl["kind"] == "tfim2d":
        return 4 * h * h + 12 * h             # MDRNN: 4h^2 + 4dh + 4h
    return f_cell_gru(h)


def make_wavefunction(wl, device):
    from rnnwavefunctions_amd import _lib, params as P
    N, H = wl["N"], wl["H"]
    scale = 3.0 if wl.get("weights") == "trained" else 1.0    # "trained-like" = kernels x 3
    if wl["kind"] == "j1j2":
        prm = P.init_gru_params([H], seed=111, heads=("wf_dense_ampl", "wf_dense_phase"))
        wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,), device=device)
        couplings = np.concatenate([np.ones(N), wl["J2"] * np.ones(N), np.zeros(N), [0.0, 0.0]])
    elif wl["kind"] == "tfim2d_gru":
        prm = P.init_gru_params([H], seed=111, dtype=np.float64)
        wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D_F64, wl["Nx"], wl["Ny"], (H,), device=device)
        couplings = np.append(np.ones(N), wl["Bx"])
    elif wl["kind"] == "tfim2d":
        prm = P.init_mdrnn_params(H, seed=111)
        wf = _lib.NativeWavefunction(_lib.MODEL_MDRNN2D, wl["Nx"], wl["Ny"], (H,), device=device)
        couplings = np.append(np.ones(N), wl["Bx"])
    else:
        prm = P.init_gru_params([H], seed=111)
        wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D, N, 1, (H,), device=device)
        couplings = np.append(np.ones(N), wl["Bx"])
    if scale != 1.0:
        prm = P.scale_kernels(prm, scale)
    wf.set_params(prm, scope="RNNwavefunction")
    return wf, prm, couplings


def cpu_baseline(wl, prm, target_seconds=15.0):
    """The reference formulation (queue of N+1 configurations per sample, every one scored from
    site 0 in <=25000-row chunks) timed on this box's host cores with the C restatement in oracle/."""
    from oracle import cport
    N, ns_full = wl["N"], wl["ns"]
    threads = cport.max_threads()
    rng = np.random.RandomState(0)
    probe = min(64, ns_full)
    s = rng.randint(0, 2, (probe, N)).astype(np.int32)
    cport.ising_local_energies(prm, np.ones(N), wl["Bx"], s[:8], nthreads=threads)   # warm-up (builds the .so)
    t0 = time.perf_counter()
    cport.ising_local_energies(prm, np.ones(N), wl["Bx"], s, nthreads=threads)
    rate = probe / max(time.perf_counter() - t0, 1e-6)
    ns_cpu = int(min(ns_full, max(probe, rate * target_seconds)))
    s = rng.randint(0, 2, (ns_cpu, N)).astype(np.int32)
    t0 = time.perf_counter()
    cport.ising_local_energies(prm, np.ones(N), wl["Bx"], s, nthreads=threads)
    dt = time.perf_counter() - t0
    # also at one thread (a smaller sample, ~5 s), with the host's CPU model and the flags
    ns_1 = int(min(ns_cpu, max(8, rate / threads * 5.0)))
    t0 = time.perf_counter()
    cport.ising_local_energies(prm, np.ones(N), wl["Bx"], s[:ns_1], nthreads=1)
    dt1 = time.perf_counter() - t0
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": ns_cpu * N / dt, "unit": "samples*sites/s", "cores": threads, "kind": "port",
            "sample": "%d of %d samples of the same workload, reference formulation ((N+1)*ns chains from site 0, "
                      "<=25000-row chunks), C/OpenMP restatement oracle/c/rnnwf_oracle.c, %.1f s" % (ns_cpu, ns_full, dt),
            "one_thread": {"value": ns_1 * N / dt1, "samples": ns_1, "seconds": dt1},
            "cpu_model": model, "nproc": os.cpu_count(),
            "compiler": "gcc -O3 -march=native -fopenmp -fno-math-errno -ffp-contract=off (oracle/cport.py)"}


def alt_engine_run(wl, couplings, warmup, steps):
    """The same workload with the flip/swap pass forced onto the f32-input MFMA (RNNWF_ENGINE=f32), so that the
    line carries both engines; reported beside `value`, never instead of it."""
    os.environ["RNNWF_ENGINE"] = "f32"
    try:
        wf, _, _ = make_wavefunction(wl, device=int(os.environ.get("LOCAL_RANK", "0")))
    finally:
        del os.environ["RNNWF_ENGINE"]
    ns, N = wl["ns"], wl["N"]
    for it in range(warmup):
        wf.vmc_step(ns, seed=111, step=it, couplings=couplings)
    wf.timing_enable(True)
    wf.timing_reset()
    wf.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        m = wf.vmc_step(ns, seed=111, step=warmup + it, couplings=couplings)["moments"]
    wf.synchronize()
    dt = (time.perf_counter() - t0) / steps
    k = wf.timing_get(1)
    launches = max(k["launches"], 1)
    ach = k["cell_evals"] / launches * f_cell(wl) / (k["total_ms"] / launches * 1e-3) / 1e12
    return {"engine": wf.engine_name(), "value": ns * N / dt, "ms_per_step": dt * 1e3, "steps": steps,
            "roofline_frac": ach / PEAK_TFLOPS["f32"], "avg_launch_ms": k["total_ms"] / launches, "mean_E": m[0] / m[2]}


def load_traffic(workload):
    """HBM bytes per flip-kernel launch measured with rocprofv3 --pmc (separate pass), if recorded."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--numsamples", type=int, default=0, help="per-GPU batch override")
    ap.add_argument("--weights", default="init", choices=("init", "trained"),
                    help="init: glorot-uniform as initialised; trained: kernels x 3 (sharper conditionals)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-engine", action="store_true", help="skip the extra f32-input-MFMA timing")
    ap.add_argument("--transport", default="rccl", choices=("rccl", "gloo"),
                    help="all-reduce of the moments: RCCL over xGMI (default) or the launcher's gloo group (rehearsal of "
                         "the multi-rank control flow on a box whose GPUs cannot host one rank each)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank on device 0")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    wl = dict(WORKLOADS[args.workload])
    if args.numsamples:
        wl["ns"] = args.numsamples
    wl["weights"] = args.weights

    # Load order matters: the product library (and RCCL, which it dlopens) come first so that the HIP runtime
    # in this process is /opt/rocm's; torch (which bundles its own copies) is imported afterwards and only for
    # the launcher's CPU-side rendezvous.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    wf, prm, couplings = make_wavefunction(wl, device=0 if args.same_device else local_rank)
    dist = None
    gloo_reduce = None
    if world > 1:
        uid = wf.comm_unique_id() if args.transport == "rccl" else None   # every rank: loads librccl now; rank 0's id is used
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        import torch.distributed as dist      # launcher plumbing only: gloo rendezvous + barrier on CPU
        dist.init_process_group(backend="gloo")
        if args.transport == "rccl":
            box = [uid if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            wf.comm_init(box[0], rank, world)
        else:
            from rnnwavefunctions_amd.distributed import ShardComm
            gloo_reduce = ShardComm.from_torch().allreduce

    ns, N = wl["ns"], wl["N"]
    offset = rank * ns                       # global sample indices of this shard

    def step(it):
        out = wf.vmc_step(ns, seed=111, step=it, couplings=couplings, sample_offset=offset)
        m = out["moments"]
        if world > 1:                         # ONE all-reduce: (sum E, sum E^2, n, sum Im E)
            m = gloo_reduce(m) if gloo_reduce else wf.allreduce_moments(m)
        return m

    def barrier():
        wf.synchronize()
        if dist is not None:
            dist.barrier()
        wf.synchronize()

    for it in range(args.warmup):
        step(it)
    wf.timing_enable(True)
    wf.timing_reset()
    barrier()
    t0 = time.perf_counter()
    for it in range(args.steps):
        m = step(args.warmup + it)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])

    flip = wf.timing_get(1)
    base = wf.timing_get(0)
    asm = wf.timing_get(2)
    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * ns * N / (dt / args.steps)
        mean_e = m[0] / m[2]
        var_e = m[1] / m[2] - mean_e ** 2
        launches = max(flip["launches"], 1)
        alg_flops_per_launch = flip["cell_evals"] / launches * f_cell(wl)
        flip_ms = flip["total_ms"] / launches
        achieved = alg_flops_per_launch / (flip_ms * 1e-3) / 1e12 if flip_ms > 0 else 0.0
        dtype = "f64" if wl["kind"] in ("tfim2d", "tfim2d_gru") else "f32"
        peak = PEAK_TFLOPS[dtype]
        engine = wf.engine_name()
        kernel = {"tfim1d": "prnn_flip_split_kernel" if engine == "bf16x3" else "prnn_flip_kernel",
                  "j1j2": "crnn_swap_split_kernel" if engine == "bf16x3" else "crnn_swap_kernel",
                  "tfim2d": "mdrnn_flip_kernel", "tfim2d_gru": "prnn_flip_kernel<double>"}[wl["kind"]]
        traffic = load_traffic(args.workload)
        rec = {
            "metric": "samples*sites/sec (autoregressive sample+local_energy), 1D TFIM N=80 nh=50"
                      if args.workload == "cfg2" else "samples*sites/sec (autoregressive sample+local_energy), " + args.workload,
            "value": value, "unit": "samples*sites/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dtype, "data": "synthetic",
            "config": {"workload": wl["desc"], "numsamples_per_gpu": ns, "global_numsamples": ns * world,
                       "sites": N, "num_units": wl["H"], "parallelism": "dp%d (sample shards, 1 %s all-reduce/step)" % (world, "RCCL" if args.transport == "rccl" else "gloo"),
                       "weights": "glorot-uniform RandomState(111), gate bias 1" + (", kernels x 3 (trained-like)" if args.weights == "trained" else ""),
                       "mean_E": mean_e, "var_E": var_e,
                       "engine": engine},
            "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak,
                         "unit": "TFLOP/s", "frac": achieved / peak,
                         "traffic": (traffic or {}).get("hbm_bytes_per_launch"),
                         "traffic_detail": traffic,
                         "hbm_frac_of_8TBps": ((traffic["hbm_bytes_per_launch"] / (flip_ms * 1e-3) / 8e12)
                                               if traffic and flip_ms > 0 else None),
                         "algorithmic_flops_per_launch": alg_flops_per_launch,
                         "mfma_flops_issued_per_launch": flip["mfma_flops"] / launches,
                         "avg_launch_ms": flip_ms,
                         "note": ("peak = dense f32-input MFMA peak, the rate of an f32 formulation of this path; the "
                                  "bf16x3 engine computes the same f32-accurate products on the bf16 matrix core "
                                  "(six bf16 products per f32 product, bf16 dense peak 2500 TF/s => 417 TF/s "
                                  "f32-equivalent), so frac is relative to the f32 path's roofline")
                                 if engine == "bf16x3" else "peak = dense MFMA peak of the arithmetic type",
                         "frac_of_bf16x6_peak": (achieved / (2500.0 / 6.0)) if engine == "bf16x3" else None,
                         "base_pass_ms": base["total_ms"] / max(base["launches"], 1),
                         "assembly_ms": asm["total_ms"] / max(asm["launches"], 1) * (asm["launches"] / launches)},
        }
        if engine == "bf16x3" and world == 1 and not args.no_alt_engine:
            rec["f32mfma_engine"] = alt_engine_run(wl, couplings, args.warmup, max(args.steps // 2, 3))
        if not args.no_cpu_baseline and world == 1 and wl["kind"] == "tfim1d":
            rec["cpu_baseline"] = cpu_baseline(wl, prm)
        else:
            rec["cpu_baseline"] = None
        print(json.dumps(rec))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
