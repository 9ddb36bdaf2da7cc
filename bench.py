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
    "w64": dict(kind="tfim1d", N=80, H=64, ns=10000, Bx=1.0,
                desc="1DTFIM pRNN N=80 num_units=64 numsamples=10000 (not a BASELINE config: a width between the ping-pong and the streamed form)"),
    "cfg5": dict(kind="tfim1d", N=200, H=100, ns=32768, Bx=1.0,
                 desc="1DTFIM pRNN N=200 num_units=100 numsamples=32768 per GPU (BASELINE config 5 shard)"),
    # SURVEY.md 8 row f4 "at speed": the parity-symmetric class (1DTFIM/RNNwavefunction_paritysym.py:80-145) and stacked layers
    # units=[num_units]*num_layers (1DTFIM/TrainingRNN_1DTFIM.py:98) at config 2's size; not BASELINE configs
    "cfg2_parity": dict(kind="tfim1d", N=80, H=50, ns=10000, Bx=1.0, parity=True,
                        desc="1DTFIM parity-symmetric pRNN N=80 num_units=50 numsamples=10000 (config 2's size; both directions per configuration)"),
    "cfg2_l2": dict(kind="tfim1d", N=80, H=50, ns=10000, Bx=1.0, layers=2,
                    desc="1DTFIM pRNN N=80 units=[50,50] numsamples=10000 (config 2's size, two stacked GRU layers)"),
    "cfg2_l3": dict(kind="tfim1d", N=80, H=50, ns=10000, Bx=1.0, layers=3,
                    desc="1DTFIM pRNN N=80 units=[50,50,50] numsamples=10000 (config 2's size, three stacked GRU layers)"),
    # the complex wave function's default is a two-layer stack (J1J2/ComplexRNNwavefunction.py:16,40): config 3's size with units=[50,50]
    "cfg3_l2": dict(kind="j1j2", N=40, H=50, ns=10000, J2=0.5, layers=2,
                    desc="J1J2 cRNN N=40 J2=0.5 U(1) mask units=[50,50] numsamples=10000 (config 3's size, two stacked GRU layers)"),
}
# /opt/skills/guides/MI355X_MICROARCH.md: dense f32-input MFMA = f32 vector peak; f64 matrix = f64 vector peak; bf16 MFMA
# dense 2 500 TF/s.  The bf16x3 engine spends SIX bf16 products per f32 product (csrc/split_core.h), so the roof of
# that pipe in f32-equivalent flops is 2 500 / 6 TF/s: `roofline.peak` is always the pipe the named kernel issues on.
PEAK_TFLOPS = {"f32": 157.3, "f64": 78.6, "bf16": 2500.0}
BF16X3_PRODUCTS = 6


def engine_peak(engine, dtype):
    """(peak TFLOP/s in the units `achieved` is counted in, description) of the pipe the dominant kernel runs on."""
    if engine == "bf16x3":
        return PEAK_TFLOPS["bf16"] / BF16X3_PRODUCTS, ("bf16 MFMA pipe (2500 TF/s dense) in f32-equivalent flops: the bf16x3 engine "
                                                       "issues 6 bf16 products per f32 product, so peak = 2500/6 TF/s")
    return PEAK_TFLOPS[dtype], "dense %s-input MFMA peak" % dtype


def f_cell_gru(h):
    """Dense flops of one GRU cell evaluation incl. Dense(2): 6h^2 + 16h."""
    return 6 * h * h + 16 * h


def f_cell(wl):
    h = wl["H"]
    if wl["kind"] == "j1j2":
        return f_cell_gru(h) + 4 * h + (wl.get("layers", 1) - 1) * 12 * h * h         # second Dense(2) head; stacked layers as below
    if wl["kind"] == "tfim2d":
        return 4 * h * h + 12 * h             # MDRNN: 4h^2 + 4dh + 4h
    # stacked layers: every layer above the first has an h-wide input instead of the 2-wide one-hot: 6h^2 + 6h^2
    return f_cell_gru(h) + (wl.get("layers", 1) - 1) * 12 * h * h


def make_wavefunction(wl, device):
    from rnnwavefunctions_amd import _lib, params as P
    N, H = wl["N"], wl["H"]
    scale = 3.0 if wl.get("weights") == "trained" else 1.0    # "trained-like" = kernels x 3
    if wl["kind"] == "j1j2":
        L = wl.get("layers", 1)
        prm = P.init_gru_params([H] * L, seed=111, heads=("wf_dense_ampl", "wf_dense_phase"))
        wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,) * L, device=device)
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
        L = wl.get("layers", 1)
        prm = P.init_gru_params([H] * L, seed=111)
        wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D_PARITY if wl.get("parity") else _lib.MODEL_GRU1D, N, 1, (H,) * L, device=device)
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


def alt_engine_run(wl, couplings, warmup, steps, last_step):
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
    wf.timing_enable(2)                   # the dominant pass only
    wf.timing_reset()
    wf.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):               # the last step has the index of the headline run's last step: same batch, comparable mean_E
        m = wf.vmc_step(ns, seed=111, step=last_step - (steps - 1 - it), couplings=couplings)["moments"]
    wf.synchronize()
    dt = (time.perf_counter() - t0) / steps
    k = wf.timing_get(1)
    launches = max(k["launches"], 1)
    ach = k["cell_evals"] / launches * f_cell(wl) / (k["total_ms"] / launches * 1e-3) / 1e12
    return {"engine": wf.engine_name(), "value": ns * N / dt, "ms_per_step": dt * 1e3, "steps": steps,
            "roofline_frac": ach / PEAK_TFLOPS["f32"], "roofline_peak": PEAK_TFLOPS["f32"],
            "avg_launch_ms": k["total_ms"] / launches, "mean_E": m[0] / m[2], "mean_E_at_step": last_step}


def oracle_local_energies(wl, prm, couplings, samples):
    """E_loc of `samples` by the CPU oracle (the checker, never the thing measured): the C restatement of the reference
    formulation for the one-layer positive RNN, the NumPy restatement (oracle/models.py, oracle/estimators.py) otherwise."""
    from oracle import cport
    from oracle import estimators as OE
    from oracle import models as OM
    N, kind = wl["N"], wl["kind"]
    if kind == "tfim1d" and wl.get("layers", 1) == 1 and not wl.get("parity"):
        return cport.ising_local_energies(prm, np.ones(N), wl["Bx"], samples, nthreads=cport.max_threads()), \
            "C restatement of the reference formulation (oracle/c/rnnwf_oracle.c)"
    name = "NumPy restatement (oracle/models.py + oracle/estimators.py)"
    chunks = [samples[k:k + 256] for k in range(0, len(samples), 256)]
    if kind == "tfim1d":
        lp = OM.prnn_paritysym_log_probability if wl.get("parity") else OM.prnn_log_probability
        return np.concatenate([OE.ising_local_energies(np.ones(N), wl["Bx"], c, lambda x: lp(prm, x)) for c in chunks]), name
    if kind == "j1j2":
        return np.concatenate([OE.j1j2_local_energies(np.ones(N), wl["J2"] * np.ones(N), np.zeros(N), c,
                                                      lambda x: OM.crnn_log_amplitude(prm, x)) for c in chunks]), name
    Nx, Ny = wl["Nx"], wl["Ny"]
    if kind == "tfim2d":
        return np.concatenate([OE.ising2d_local_energies(np.ones((Nx, Ny)), wl["Bx"], Nx, Ny, c,
                                                         lambda x: OM.mdrnn_log_probability(prm, x)) for c in chunks]), name
    return np.concatenate([OE.ising2d_local_energies(np.ones((Nx, Ny)), wl["Bx"], Nx, Ny, c,
                                                     lambda x: OM.prnn_log_probability(prm, x, dtype=np.float64)) for c in chunks]), name


# samples of the parity leg per workload: what ~10-15 s of the box's host cores score (config 2: the whole batch)
PARITY_SAMPLES = {"cfg1": 500, "cfg2": 10000, "cfg3": 2048, "cfg4": 256, "cfg5": 512, "w64": 2000, "2d1drnn": 256,
                  "cfg2_parity": 256, "cfg2_l2": 256, "cfg2_l3": 192, "cfg3_l2": 512}


def parity_leg(workload, wl, wf, prm, couplings, step_index, offset, timed_moments):
    """Validates the timed path's own output: the LAST timed step is run once more with its samples and local energies
    returned (same seed / step / offset -> the same batch; its moments must reproduce the timed step's bit for bit), and
    the oracle scores a spread subset (config 2: all) of that very sample matrix."""
    ns, N = wl["ns"], wl["N"]
    out = wf.vmc_step(ns, seed=111, step=step_index, couplings=couplings, sample_offset=offset, want_samples=True, want_eloc=True)
    n = min(ns, PARITY_SAMPLES.get(workload, 256))
    sub = np.unique(np.linspace(0, ns - 1, n).astype(np.int64))
    t0 = time.perf_counter()
    e_ref, oracle_name = oracle_local_energies(wl, prm, couplings, out["samples"][sub])
    secs = time.perf_counter() - t0
    e = np.asarray(out["eloc"])[sub]
    d = np.abs(e - e_ref)
    d_mean = abs(complex(np.mean(e.astype(np.complex128))) - complex(np.mean(np.asarray(e_ref).astype(np.complex128))))
    tol = 1e-4                                # north_star: <E> within 1e-4 per site of the reference
    return {"samples": int(len(sub)), "of": int(ns), "step": int(step_index),
            "max_abs_dE_per_site": float(d.max() / N), "d_meanE_per_site": float(d_mean / N),
            "tolerance_per_site": tol, "pass": bool(d.max() / N < tol and d_mean / N < tol),
            "reproduces_timed_step": bool(np.array_equal(np.asarray(out["moments"]), np.asarray(timed_moments))),
            "oracle": oracle_name, "oracle_seconds": secs}


def train_leg(wl, wf, prm, couplings, steps, offset):
    """the other third of a VMC iteration - rnnwf_vmc_gradient (back-propagation through time on the
    MFMA + the weight-gradient GEMM) on the batch of the step before it, timed alone; then whole iterations resident on the device
    (rnnwf_train_steps: step + gradient + Adam + re-pack of the weight images, ten per host synchronisation)."""
    ns = wl["ns"]
    shapes = {k.split("/", 1)[1]: v.shape for k, v in prm.items()}
    wf.timing_enable(1)
    wf.timing_reset()
    t_step, t_grad = [], []
    for it in range(steps + 1):
        wf.synchronize()
        t0 = time.perf_counter()
        m = wf.vmc_step(ns, seed=111, step=1000 + it, couplings=couplings, sample_offset=offset)["moments"]
        t1 = time.perf_counter()
        mean_e = complex(m[0] / m[2], m[3] / m[2]) if wl["kind"] == "j1j2" else m[0] / m[2]
        g = wf.vmc_gradient(mean_e, m[2], shapes)
        t2 = time.perf_counter()
        if it:                                # the first iteration packs the backward weight image
            t_step.append(t1 - t0)
            t_grad.append(t2 - t1)
    bptt, gemm = wf.timing_get(3), wf.timing_get(4)
    per = max(steps + 1, 1)
    gnorm = float(np.sqrt(sum(float((v * v).sum()) for v in g.values())))
    resident = None
    if wf.device_training_supported():
        lrs = [1e-12] * 10                     # a vanishing rate: the timed work is an iteration's, the weights stay the benchmark's
        wf.train_steps(ns, 111, 2000, couplings, lrs, 0.9, 0.999, 1e-8, sample_offset=offset)
        t0 = time.perf_counter()
        for c in range(max(steps // 10, 1) + 1):
            wf.train_steps(ns, 111, 2010 + 10 * c, couplings, lrs, 0.9, 0.999, 1e-8, sample_offset=offset)
        resident = (time.perf_counter() - t0) / (10 * (max(steps // 10, 1) + 1)) * 1e3
    return {"steps": steps, "device_resident_iteration_ms": resident, "vmc_step_ms_median": float(np.median(t_step) * 1e3),
            "gradient_ms_median": float(np.median(t_grad) * 1e3), "gradient_ms_mean": float(np.mean(t_grad) * 1e3),
            "gradient_includes": "BPTT kernel(s) + weight-gradient GEMM + D2H of the gradient arrays + host unpacking",
            "bptt_kernel_ms_per_iteration": bptt["total_ms"] / per, "bptt_launches_per_iteration": bptt["launches"] / per,
            "tn_gemm_ms_per_iteration": gemm["total_ms"] / per, "tn_gemm_launches_per_iteration": gemm["launches"] / per,
            "grad_l2_norm": gnorm, "finite": bool(np.isfinite(gnorm))}


def dominant_kernel(wl, engine):
    """The dominant kernel as rocprofv3 names it (release library: one kernel per model and width class)."""
    H, L = wl["H"], wl.get("layers", 1)
    if wl["kind"] == "tfim2d":
        return "mdrnn_flip_kernel"
    if wl["kind"] == "tfim2d_gru":
        return "prnn_flip_kernel<double>"
    if engine != "bf16x3":
        return {"tfim1d": "prnn_ml_flip_kernel" if L > 1 else "prnn_flip_kernel", "j1j2": "crnn_ml_swap_kernel" if L > 1 else "crnn_swap_kernel"}[wl["kind"]]
    pp = 37 <= H <= 50                     # the ping-pong form of the bf16x3 engine
    if wl["kind"] == "j1j2":
        return ("crnn_swap_pp_upper_kernel" if L > 1 else "crnn_swap_pp_kernel") if pp else "crnn_swap_split_kernel"
    if L > 1:
        return "prnn_flip_pp_upper_kernel"
    return "prnn_flip_pp_kernel" if pp else "prnn_flip_riders16_asm_kernel" if H > 68 else "prnn_flip_split_kernel"


def assemble_record(*, args, wl, world, dt, step_ms, per_rank_ms, infos, moments, flip, base, asm, engine, transport_fallback,
                    cfg5, traffic, last_step):
    """Rank 0's JSON line from what the ranks measured - a pure function of its arguments (no GPU, no torch), so that
    tests/test_host.py can drive the N = 8 assembly with synthetic per-rank records before the first 8-GPU run does."""
    ns, N = wl["ns"], wl["N"]
    m = moments
    ms_per_step = dt / args.steps * 1e3
    value = world * ns * N / (dt / args.steps)
    mean_e = m[0] / m[2]
    var_e = m[1] / m[2] - mean_e ** 2
    launches = max(flip["launches"], 1)
    alg_flops_per_launch = flip["cell_evals"] / launches * f_cell(wl)
    flip_ms = flip["total_ms"] / launches
    achieved = alg_flops_per_launch / (flip_ms * 1e-3) / 1e12 if flip_ms > 0 else 0.0
    dtype = "f64" if wl["kind"] in ("tfim2d", "tfim2d_gru") else "f32"
    peak, peak_note = engine_peak(engine, dtype)
    kernel = dominant_kernel(wl, engine)
    issued = flip["mfma_flops"] / launches
    step_ms = np.asarray(step_ms, dtype=float)
    rec = {
        "metric": "samples*sites/sec (autoregressive sample+local_energy), 1D TFIM N=80 nh=50"
                  if args.workload == "cfg2" else "samples*sites/sec (autoregressive sample+local_energy), " + args.workload,
        "value": value, "unit": "samples*sites/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # the arithmetic the path computes in.  bf16x3: every f32 operand held EXACTLY as three bf16 parts, the six
        # significant bf16 products accumulated in f32 (csrc/split_core.h) - f32-equivalent (both engines meet the same
        # tolerance against the float64 oracle), not bitwise f32; the pure f32-input-MFMA run is `f32mfma_engine`
        "dtype": dtype + (" (bf16x3 split, f32 accumulate)" if engine == "bf16x3" else ""),
        "arithmetic": ("bf16x3: 3 exact bf16 parts per f32 operand, 6 bf16 MFMA products per f32 product, f32 accumulate, in the "
                       "flip / swap pass and (one layer of 37..52 units) the cooperative base pass that samples; other base passes "
                       "on the f32-input MFMA" if engine == "bf16x3" else "%s-input MFMA, %s accumulate" % (dtype, dtype)),
        "data": "synthetic",
        # SURVEY.md 8(d) asks for the median of >= 20 steps: per-step wall times (each step ends with its own host
        # synchronisation) of THIS rank; `ms_per_step` / `value` stay the barrier-bracketed total over K steps, max over ranks
        "ms_per_step_median": float(np.median(step_ms)), "ms_per_step_min": float(step_ms.min()), "ms_per_step_max": float(step_ms.max()),
        "value_at_median_step": world * ns * N / (float(np.median(step_ms)) * 1e-3),
        "ms_per_step_per_rank": list(per_rank_ms),
        "config": {"workload": wl["desc"], "numsamples_per_gpu": ns, "global_numsamples": ns * world,
                   "sites": N, "num_units": wl["H"], "layers": wl.get("layers", 1),
                   "parallelism": "dp%d (sample shards, 1 %s all-reduce/step)" % (world, "RCCL" if args.transport == "rccl" else "gloo"),
                   "weights": "glorot-uniform RandomState(111), gate bias 1" + (", kernels x 3 (trained-like)" if args.weights == "trained" else ""),
                   "mean_E": mean_e, "var_E": var_e, "mean_E_at_step": last_step,
                   "engine": engine},
        # the communicator's own rank count (ncclCommCount), per rank with its device: N x dp1 cannot pass for dp-N
        "rccl_nranks": min(i["nranks"] for i in infos) if args.transport == "rccl" and not transport_fallback else None,
        "transport_fallback": transport_fallback,
        "ranks": [{"rank": i["pid_rank"], "comm_rank": i["rank"], "comm_nranks": i["nranks"], "device": i["device"]} for i in infos],
        "roofline": {"bound": "mfma", "kernel": kernel, "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak, "peak_is": peak_note,
                     "engine_is_bf16x3": engine == "bf16x3",
                     # the same achieved rate against the f32-input MFMA roof (what an f32 formulation could reach);
                     # above 1 only because the bf16x3 engine left that pipe - informational, never the fraction
                     "frac_of_f32_mfma_peak": achieved / PEAK_TFLOPS[dtype],
                     # MFMA flops the kernel actually issued (padding included) over the peak of their pipe
                     "mfma_issue_frac": (issued / (flip_ms * 1e-3) / 1e12 / (PEAK_TFLOPS["bf16"] if engine == "bf16x3" else PEAK_TFLOPS[dtype]))
                                        if flip_ms > 0 else None,
                     "traffic": (traffic or {}).get("hbm_bytes_per_launch"),
                     "traffic_source": ("replayed from %s (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, build %s); "
                                        "not measured by this run" % (TRAFFIC_FILE, (traffic or {}).get("build", "unknown")))
                                       if traffic else None,
                     "traffic_detail": traffic,
                     "hbm_frac_of_8TBps": ((traffic["hbm_bytes_per_launch"] / (flip_ms * 1e-3) / 8e12)
                                           if traffic and traffic.get("hbm_bytes_per_launch") and flip_ms > 0 else None),
                     # clock the kernel held in an EARLIER profiler pass (GRBM_GUI_ACTIVE; the guide's peaks are quoted at
                     # 2.4 GHz): informational, replayed like `traffic`, never the fraction
                     "held_clock_ghz": (traffic or {}).get("held_clock_ghz"),
                     "held_clock_source": (traffic or {}).get("held_clock_source"),
                     "frac_at_held_clock": (achieved / peak * 2.4 / traffic["held_clock_ghz"]
                                            if traffic and traffic.get("held_clock_ghz") else None),
                     "algorithmic_flops_per_launch": alg_flops_per_launch,
                     "mfma_flops_issued_per_launch": issued,
                     "launches_per_step": flip["launches"] / max(args.steps, 1),
                     "avg_launch_ms": flip_ms,
                     "base_pass_ms": base["total_ms"] / max(base["launches"], 1),
                     "assembly_ms": asm["total_ms"] / 3.0},            # per step (the three extra steps behind the timed region)
    }
    if cfg5 is not None:
        rec["cfg5_sharded"] = cfg5
    return rec


def exit_status(rec, allow_fallback=False):
    """(exit code, reason) of a finished run: a red parity leg or a scaling run that fell back to gloo must not hand the driver an
    rc-0 number.  2: the oracle disagrees with the timed path (or the re-run did not reproduce the timed step); 3: RCCL fallback."""
    par = rec.get("parity")
    if par is not None:
        if not par.get("pass"):
            return 2, "parity leg failed: max |dE_loc|/N %.3g, d<E>/N %.3g against tolerance %.1g" % (
                par.get("max_abs_dE_per_site", float("nan")), par.get("d_meanE_per_site", float("nan")), par.get("tolerance_per_site", 0.0))
        if not par.get("reproduces_timed_step"):
            return 2, "parity leg: the re-run of the last timed step did not reproduce its moments bit for bit"
    if rec.get("transport_fallback") and not allow_fallback:
        return 3, "RCCL communicator could not be created; the moments went over gloo (pass --allow-fallback to accept): " + rec["transport_fallback"]
    if rec.get("n_gpus", 1) > 1 and rec.get("rccl_nranks") is not None and rec["rccl_nranks"] != rec["n_gpus"]:
        return 3, "RCCL communicator spans %s ranks, the job %s" % (rec["rccl_nranks"], rec["n_gpus"])
    return 0, None


TRAFFIC_FILE = os.path.join("profiles", "pmc_traffic.json")


def load_traffic(workload):
    """HBM bytes per flip-kernel launch from an EARLIER rocprofv3 --pmc run (FETCH_SIZE and WRITE_SIZE need separate
    passes and cannot be read from inside this process); the record says which build it belongs to, and the bench
    line labels it as replayed."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            return json.load(f).get(workload)
    except (OSError, ValueError):
        return None


def sharded_cfg5(wf_factory, rank, world, reduce_moments, barrier, steps=2):
    """North-star config 5 (N=200, 100 units, 32 768 samples per GPU) on the ranks of this job: `steps` timed VMC steps
    after one warm-up, one RCCL all-reduce of the moments per step."""
    wl = dict(WORKLOADS["cfg5"])
    wf, _, couplings = wf_factory(wl)
    ns, N = wl["ns"], wl["N"]

    def step(it):
        return reduce_moments(wf, wf.vmc_step(ns, seed=111, step=it, couplings=couplings, sample_offset=rank * ns)["moments"])
    step(0)
    barrier(wf)
    t0 = time.perf_counter()
    for it in range(steps):
        m = step(1 + it)
    barrier(wf)
    return wl, time.perf_counter() - t0, m


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
    ap.add_argument("--no-parity", action="store_true", help="skip the oracle check of the last timed step's output")
    ap.add_argument("--train", type=int, default=0, metavar="K",
                    help="also time K training iterations' gradient (rnnwf_vmc_gradient) behind the timed region")
    ap.add_argument("--no-cfg5", action="store_true",
                    help="multi-GPU runs: skip the extra north-star config 5 leg (N=200, 100 units, 32768 samples per GPU)")
    ap.add_argument("--transport", default="rccl", choices=("rccl", "gloo"),
                    help="all-reduce of the moments: RCCL over xGMI (default) or the launcher's gloo group (rehearsal of "
                         "the multi-rank control flow on a box whose GPUs cannot host one rank each)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank on device 0")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="multi-GPU runs: exit 0 even if the RCCL communicator could not be created and the moments went over gloo "
                         "(without this flag such a run exits 3: a scaling number that silently measured gloo is worse than none)")
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
    device = 0 if args.same_device else local_rank

    # Load order matters: the product library (and RCCL, which it dlopens) come first so that the HIP runtime
    # in this process is /opt/rocm's; torch (which bundles its own copies) is imported afterwards and only for
    # the launcher's CPU-side rendezvous.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    wf, prm, couplings = make_wavefunction(wl, device=device)
    dist = None
    gloo_reduce = None
    if world > 1:
        if args.transport == "rccl":
            wf.comm_unique_id()               # every rank: loads librccl now, before torch brings its own copy
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        import torch.distributed as dist      # launcher plumbing only: gloo rendezvous + barrier on CPU
        dist.init_process_group(backend="gloo")
        if args.transport == "gloo":
            from rnnwavefunctions_amd.distributed import ShardComm
            gloo_reduce = ShardComm.from_torch().allreduce

    transport_note = {"fallback": None}

    def init_comm(w):                         # one RCCL communicator per handle: rank 0's unique id travels over gloo
        nonlocal gloo_reduce
        if world > 1 and args.transport == "rccl" and transport_note["fallback"] is None:
            box = [w.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            err = None
            try:
                w.comm_init(box[0], rank, world)
            except Exception as e:            # noqa: BLE001 - reported in the bench line, never hidden
                err = "%s: %s" % (type(e).__name__, e)
            # every rank must take the same road: if ncclCommInitRank failed anywhere, ALL ranks sum the four moments over
            # the launcher's gloo group instead (32 bytes per step; the data path is unchanged) and the line says so
            import torch
            bad = torch.tensor([1.0 if err else 0.0], dtype=torch.float64)
            dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if float(bad[0]) > 0:
                errs = [None] * world
                dist.all_gather_object(errs, err)
                transport_note["fallback"] = "RCCL communicator could not be created (%s); moments all-reduced over gloo" % \
                    "; ".join("rank %d: %s" % (i, e_) for i, e_ in enumerate(errs) if e_)
                from rnnwavefunctions_amd.distributed import ShardComm
                gloo_reduce = ShardComm.from_torch().allreduce
                return
            w.comm_reduce_in_step(True)       # the step's moments come back already summed over the ranks

    def reduce_moments(w, m):                 # ONE all-reduce per step: (sum E, sum E^2, n, sum Im E)
        if world == 1 or not gloo_reduce:     # RCCL: done inside rnnwf_vmc_step, on the stream (comm_reduce_in_step)
            return m
        return gloo_reduce(m)

    def barrier(w):
        w.synchronize()
        if dist is not None:
            dist.barrier()
        w.synchronize()

    init_comm(wf)
    ns, N = wl["ns"], wl["N"]
    offset = rank * ns                       # global sample indices of this shard

    def step(it):
        return reduce_moments(wf, wf.vmc_step(ns, seed=111, step=it, couplings=couplings, sample_offset=offset)["moments"])

    for it in range(args.warmup):
        step(it)
    # HIP events around the DOMINANT pass only inside the timed region (two stream operations per step; the roofline's launch
    # duration comes from exactly these launches); the other kernel groups are timed in a few extra steps behind it
    wf.timing_enable(2)
    wf.timing_reset()
    barrier(wf)
    t0 = time.perf_counter()
    marks = [t0]
    for it in range(args.steps):
        m = step(args.warmup + it)            # ends with the step's one host synchronisation (the moments' D2H copy)
        marks.append(time.perf_counter())
    barrier(wf)
    dt = time.perf_counter() - t0
    dt_local = dt
    step_ms = np.diff(marks) * 1e3
    last_step = args.warmup + args.steps - 1
    flip = wf.timing_get(1)
    wf.timing_enable(1)
    wf.timing_reset()
    for it in range(3):
        step(args.warmup + args.steps + it)
    barrier(wf)

    def max_over_ranks(x):
        if dist is None:
            return x
        import torch
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    dt = max_over_ranks(dt)
    # every rank's own time per step beside the max that `value` is computed from: a straggler shows
    per_rank_ms = [dt_local / args.steps * 1e3]
    if dist is not None:
        per_rank_ms = [None] * world
        dist.all_gather_object(per_rank_ms, dt_local / args.steps * 1e3)
    # what each rank's communicator says about itself (ncclCommCount / ncclCommUserRank) and the GPU it sits on
    info = wf.comm_info()
    info["pid_rank"] = rank
    infos = [info]
    if dist is not None:
        infos = [None] * world
        dist.all_gather_object(infos, info)

    base = wf.timing_get(0)
    asm = wf.timing_get(2)
    cfg5 = None
    if world > 1 and not args.no_cfg5 and args.workload == "cfg2" and not args.numsamples:
        def factory(w5):
            w5["weights"] = args.weights
            made = make_wavefunction(w5, device=device)
            init_comm(made[0])
            return made
        wl5, dt5, m5 = sharded_cfg5(factory, rank, world, reduce_moments, barrier)
        dt5 = max_over_ranks(dt5)
        cfg5 = {"workload": wl5["desc"], "steps": 2, "warmup": 1, "ms_per_step": dt5 / 2 * 1e3,
                "value": world * wl5["ns"] * wl5["N"] / (dt5 / 2), "unit": "samples*sites/s",
                "global_numsamples": world * wl5["ns"], "mean_E": m5[0] / m5[2],
                "note": "north-star config 5 (sharded over the ranks of this job, one RCCL all-reduce per step), "
                        "reported beside `value`, which stays on the metric's own workload so that it is comparable across N"}
    rc = 0
    if rank == 0:
        rec = assemble_record(args=args, wl=wl, world=world, dt=dt, step_ms=step_ms, per_rank_ms=per_rank_ms, infos=infos, moments=m,
                              flip=flip, base=base, asm=asm, engine=wf.engine_name(), transport_fallback=transport_note["fallback"],
                              cfg5=cfg5, traffic=load_traffic(args.workload), last_step=last_step)
        if rec["roofline"]["engine_is_bf16x3"] and world == 1 and not args.no_alt_engine:
            rec["f32mfma_engine"] = alt_engine_run(wl, couplings, args.warmup, max(args.steps // 2, 3), last_step)
        # the oracle as CHECKER of the timed path's own output (outside the timed region, like the cpu_baseline leg)
        rec["parity"] = parity_leg(args.workload, wl, wf, prm, couplings, last_step, offset, m) if world == 1 and not args.no_parity else None
        if args.train and world == 1:
            rec["train"] = train_leg(wl, wf, prm, couplings, args.train, offset)
        if not args.no_cpu_baseline and world == 1 and wl["kind"] == "tfim1d" and wl.get("layers", 1) == 1 and not wl.get("parity"):
            rec["cpu_baseline"] = cpu_baseline(wl, prm)
        else:
            rec["cpu_baseline"] = None
        rc, why = exit_status(rec, args.allow_fallback)
        rec["exit_code"], rec["exit_reason"] = rc, why
        print(json.dumps(rec))
        if rc:
            print("bench.py: FAILED - " + why, file=sys.stderr)
    if dist is not None:
        box = [rc]
        dist.broadcast_object_list(box, src=0)      # every rank leaves with rank 0's verdict: the launcher sees one exit code
        rc = box[0]
        dist.barrier()
        dist.destroy_process_group()
    sys.exit(rc)


if __name__ == "__main__":
    main()
