"""CPU tests of the host side: the C-ABI library loads and exports everything include/rnnwf.h declares
(no compute without a GPU), the host-only estimator helpers match the reference's golden vectors
bit for bit, the TF stand-ins behave, and the C restatement agrees with the NumPy oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import cport
from oracle import estimators as E
from oracle import models as M
from oracle import philox
from rnnwavefunctions_amd import compat
from rnnwavefunctions_amd import params as P
from rnnwavefunctions_amd.estimators import J1J2MatrixElements, J1J2Slices


@pytest.fixture(scope="module")
def built_lib():
    from rnnwavefunctions_amd import build
    return build.build()


def test_abi_library_exports_every_declared_symbol(built_lib):
    from rnnwavefunctions_amd import _lib
    header = open(os.path.join(ROOT, "include", "rnnwf.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(rnnwf_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = ctypes.CDLL(built_lib)
    for name in declared:
        assert hasattr(lib, name), name
    handle = _lib.load_library()
    assert handle.rnnwf_backend_name() == b"hip-gfx950"
    assert handle.rnnwf_abi_version() == _lib.ABI_VERSION


def test_no_device_fails_loudly(built_lib):
    from rnnwavefunctions_amd import _lib
    lib = _lib.load_library()
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.RnnwfError, match="no HIP device"):
        _lib.NativeWavefunction(_lib.MODEL_GRU1D, 10, 1, (10,))
    assert b"no HIP device" in lib.rnnwf_last_error(None)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "rnnwavefunctions_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(d, f)


def test_host_j1j2_matrix_elements_bit_exact(golden_j1j2):
    g = golden_j1j2
    for c in range(int(g["g1_ncases"])):
        pre = "g1_%d_" % c
        N, J2v, periodic, marshall = g[pre + "meta"]
        N = int(N)
        for k, sig in enumerate(g[pre + "sigma"]):
            sh = np.zeros((2 * N + 2, N), dtype=np.int32)
            me = np.zeros(2 * N + 2, dtype=np.float32)
            num = J1J2MatrixElements(np.ones(N), J2v * np.ones(N), 0.1 * np.arange(N), sig, sh, me,
                                     bool(periodic), bool(marshall))
            assert num == g[pre + "num"][k]
            assert np.array_equal(sh[:num], g[pre + "rows"][k, :num])
            assert np.array_equal(me[:num], g[pre + "elems"][k, :num])


def test_host_j1j2_slices_keep_the_reference_quirk(golden_j1j2):
    g = golden_j1j2
    N, ns = 8, 32
    for flag in (0, 1):
        pre = "g2_%d_" % flag
        sig = np.zeros(((2 * N + 2) * ns, N), dtype=np.int32)
        H = np.zeros((2 * N + 2) * ns, dtype=np.float32)
        sl, tot = J1J2Slices(np.ones(N), 0.5 * np.ones(N), np.zeros(N), g[pre + "samples"], sig, H,
                             np.zeros((2 * N + 2, N), dtype=np.int32), np.zeros(2 * N + 2, dtype=np.float32), bool(flag))
        assert [s.start for s in sl] + [tot] == g[pre + "offsets"].tolist()
        assert np.array_equal(sig[:tot], g[pre + "sigmas"])
        assert np.array_equal(H[:tot], g[pre + "H"])


def test_session_standins():
    sess = compat.Session(graph=compat.Graph(), config=compat.ConfigProto())
    with pytest.raises(TypeError):
        sess.run("not an op")
    ph = compat.placeholder(compat.int32, shape=(None, 5))
    assert ph.shape == (None, 5)
    assert compat.is_gru_cell(compat.CudnnCompatibleGRUCell) and compat.is_gru_cell(None)
    assert not compat.is_gru_cell("LSTMCell")
    with compat.Graph().as_default() as g:
        assert isinstance(g, compat.Graph) and compat.get_default_graph() is g
    assert compat.get_default_graph() is None
    # the handful of further TF-1 names the reference's training scripts touch (TrainingRNN_1DTFIM.py:2,82-123)
    assert compat.contrib.cudnn_rnn.CudnnCompatibleGRUCell is compat.CudnnCompatibleGRUCell
    compat.compat.v1.logging.set_verbosity(compat.compat.v1.logging.ERROR)
    compat.reset_default_graph()
    compat.set_random_seed(3)
    step = compat.Variable(0, trainable=False)
    lr_ph = compat.placeholder(dtype=compat.float64, shape=[])
    lr = compat.train.exponential_decay(lr_ph, global_step=step, decay_steps=100, decay_rate=0.5, staircase=True)
    assert lr.value({lr_ph: 0.01}) == 0.01
    step.value = np.asarray(250)
    assert lr.value({lr_ph: 0.01}) == 0.0025
    opt = compat.train.AdamOptimizer(learning_rate=lr)
    assert (opt.beta1, opt.beta2, opt.epsilon) == (0.9, 0.999, 1e-8)
    assert sess.run(compat.global_variables_initializer()) is None and sess.run(step) == 250
    with compat.variable_scope("RNNwavefunction", reuse=compat.AUTO_REUSE):
        pass
    with pytest.raises(RuntimeError):
        compat.trainable_variables()


def test_drivers_pick_the_launchers_gpu(monkeypatch):
    """comm="env" (one process per GPU under torch.distributed.run): the handle opens LOCAL_RANK's device unless the
    caller names one (round-1 finding: every rank opened GPU 0)."""
    from rnnwavefunctions_amd import training as T
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert T._resolve_device(None, "env") == 5
    assert T._resolve_device(2, "env") == 2
    assert T._resolve_device(None, None) == 0
    monkeypatch.delenv("LOCAL_RANK")
    assert T._resolve_device(None, "env") == 0
    import inspect
    for fn in (T.run_1DTFIM, T.run_J1J2, T.run_2DTFIM_2DRNN, T.run_2DTFIM_1DRNN):
        assert inspect.signature(fn).parameters["device"].default is None


def test_adam_state_round_trips_through_a_checkpoint(tmp_path):
    from rnnwavefunctions_amd import tf_checkpoint as TC
    from rnnwavefunctions_amd.training import Adam
    prm = P.init_gru_params([5], seed=1)
    opt = Adam()
    rng = np.random.RandomState(0)
    for _ in range(7):
        prm = opt.step(prm, {k: rng.standard_normal(v.shape) for k, v in prm.items()}, 1e-2)
    TC.write_checkpoint(str(tmp_path / "m.ckpt"), dict(prm, **opt.state_tensors(prm, "RNNwavefunction")))
    model, state = TC.split_saver_variables(TC.read_checkpoint(str(tmp_path / "m.ckpt")))
    assert set(model) == set(prm) and state["global_step"] == 7
    assert abs(state["beta1_power"] - 0.9 ** 8) < 1e-7        # TF: beta1_power = beta1^(t+1) after t steps
    back = Adam()
    back.load_state(state, list(prm))
    assert back.t == 7
    for k in prm:
        assert np.allclose(back.m[k], opt.m[k], rtol=1e-6, atol=1e-12) and np.allclose(back.v[k], opt.v[k], rtol=1e-6, atol=1e-12)


def test_params_roundtrip(tmp_path):
    prm = P.init_gru_params([7], seed=3, heads=("wf_dense_ampl", "wf_dense_phase"))
    P.save_npz(tmp_path / "w.npz", prm)
    back = P.load_npz(tmp_path / "w.npz")
    assert list(back) == list(prm)
    assert all(np.array_equal(back[k], prm[k]) and back[k].dtype == prm[k].dtype for k in prm)
    assert np.all(prm["RNNwavefunction/multi_rnn_cell/cell_0/cudnn_compatible_gru_cell/gates/bias"] == 1)


def test_c_restatement_matches_numpy_oracle():
    prm = P.randomize_biases(P.scale_kernels(P.init_gru_params([23], seed=5), 2.0), 6)
    s = np.random.RandomState(0).randint(0, 2, (70, 21)).astype(np.int32)
    assert np.allclose(cport.prnn_log_probability(prm, s), M.prnn_log_probability(prm, s), atol=2e-5)
    Jz = 1 + 0.1 * np.arange(21)
    e_c, lp_c = cport.ising_local_energies(prm, Jz, 0.8, s, return_log_probs=True)
    e_n, lp_n = E.ising_local_energies(Jz, 0.8, s, lambda x: M.prnn_log_probability(prm, x), return_log_probs=True)
    assert np.allclose(lp_c, lp_n.ravel(), atol=2e-5)
    assert np.allclose(e_c, e_n, rtol=2e-5)
    u = philox.uniforms(9, 1, 0, 100, 21)
    s_c, l_c = cport.prnn_sample(prm, 21, u)
    s_n, l_n = M.prnn_sample(prm, 21, u)
    assert (s_c != s_n).any(axis=1).sum() <= 1
    assert cport.usable_cores() >= 1


def test_bench_helpers():
    import bench
    assert bench.f_cell_gru(50) == 15800 and bench.f_cell_gru(100) == 61600
    assert bench.WORKLOADS["cfg2"]["N"] == 80 and bench.WORKLOADS["cfg2"]["H"] == 50


def test_c_side_initialiser_reproduces_numpy_randomstate(tmp_path):
    """csrc/init_params.h (MT19937 + genrand_res53 + glorot limits) against numpy.random.RandomState, bit for bit:
    rnnwf_init_params must give a C caller the weights params.init_gru_params gives a Python caller."""
    import ctypes as C
    import subprocess
    src = tmp_path / "shim.cpp"
    src.write_text('#include "init_params.h"\n'
                   'extern "C" void fill(unsigned seed, long rows, long cols, int vec, int f32, int skip, double* out) {\n'
                   '    rnnwf::NumpyRandomState r(seed);\n'
                   '    for (int i = 0; i < skip; ++i) r.next_double();\n'
                   '    std::vector<double> v; rnnwf::glorot_fill(r, rows, cols, vec, f32, v);\n'
                   '    for (size_t i = 0; i < v.size(); ++i) out[i] = v[i];\n}\n')
    so = tmp_path / "shim.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "rnnwavefunctions_amd", "csrc"),
                    str(src), "-o", str(so)], check=True)
    lib = C.CDLL(str(so))
    from rnnwavefunctions_amd import params as P
    for seed in (0, 111, 2 ** 32 - 1):
        prm = P.init_gru_params([7], seed=seed)                      # first tensor: gates/kernel [9, 14], float32
        first = prm["RNNwavefunction/multi_rnn_cell/cell_0/cudnn_compatible_gru_cell/gates/kernel"]
        out = np.empty(first.size)
        lib.fill(C.c_uint(seed), C.c_long(9), C.c_long(14), 0, 1, 0, out.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out.reshape(9, 14).astype(np.float32), first)
        # second drawn tensor starts 9*14 doubles later in the stream
        second = prm["RNNwavefunction/multi_rnn_cell/cell_0/cudnn_compatible_gru_cell/candidate/input_projection/kernel"]
        out2 = np.empty(second.size)
        lib.fill(C.c_uint(seed), C.c_long(2), C.c_long(7), 0, 1, 9 * 14, out2.ctypes.data_as(C.c_void_p))
        assert np.array_equal(out2.reshape(2, 7).astype(np.float32), second)
    b = P.init_mdrnn_params(5, seed=3)["RNNwavefunction/b_rnn_0"]     # 1-D xavier, float64, after 2*(25+10) draws
    out = np.empty(5)
    lib.fill(C.c_uint(3), C.c_long(1), C.c_long(5), 1, 0, 2 * (25 + 10), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, b)


def test_adam_slots_are_matched_by_suffix_and_never_dropped_silently():
    """The reference calls apply_gradients inside tf.variable_scope(wf.scope) (TrainingRNN_1DTFIM.py:149-163), where TF-1
    nests the slot names under the scope again; names are unpinned, so the reader accepts any scope prefix and refuses a
    file whose slots it cannot place."""
    from rnnwavefunctions_amd import tf_checkpoint as TC
    from rnnwavefunctions_amd.training import Adam
    prm = P.init_gru_params([4], seed=2)
    names = list(prm)
    rng = np.random.RandomState(1)
    m = {k: rng.standard_normal(v.shape) for k, v in prm.items()}
    v = {k: rng.random_sample(v.shape) for k, v in prm.items()}
    for style in ("doubled", "plain", "bare", "numbered"):
        def key(k):
            bare = k.split("/", 1)[1]
            return {"doubled": "RNNwavefunction/" + k, "plain": k, "bare": bare, "numbered": "RNNwavefunction_1/" + k}[style]
        dump = dict(prm)
        dump.update({key(k) + "/Adam": m[k].astype(np.float32) for k in names})
        dump.update({key(k) + "/Adam_1": v[k].astype(np.float32) for k in names})
        dump["RNNwavefunction_1/beta1_power"] = np.float32(0.9 ** 501)
        dump["RNNwavefunction_1/beta2_power"] = np.float32(0.999 ** 501)
        dump["RNNwavefunction_1/Variable"] = np.int32(500)
        model, state = TC.split_saver_variables(dump)
        assert set(model) == set(prm)
        opt = Adam()
        opt.load_state(state, names)
        assert opt.t == 500
        for k in names:
            assert np.allclose(opt.m[k], m[k], rtol=1e-6) and np.allclose(opt.v[k], v[k], rtol=1e-6)
    # slots for only some variables: loud
    partial = dict(prm)
    partial[names[0] + "/Adam"] = m[names[0]]
    partial[names[0] + "/Adam_1"] = v[names[0]]
    partial["Variable"] = np.int32(3)
    _, state = TC.split_saver_variables(partial)
    with pytest.raises(TC.CheckpointError, match="Adam slots but none for"):
        Adam().load_state(state, names)
    # no global step: the step count comes from beta2_power (beta1^(t+1) is zero in float32 from t ~ 980 on)
    for t in (0, 7, 2000, 50000):
        dump = dict(prm)
        dump.update({k + "/Adam": m[k] for k in names})
        dump.update({k + "/Adam_1": v[k] for k in names})
        dump["beta1_power"] = np.float32(0.9 ** (t + 1))
        dump["beta2_power"] = np.float32(0.999 ** (t + 1))
        _, state = TC.split_saver_variables(dump)
        assert state["global_step"] is None
        opt = Adam()
        opt.load_state(state, names)
        assert abs(opt.t - t) <= max(1, t // 20000), (t, opt.t)
    dump.pop("beta2_power")
    dump["beta1_power"] = np.float32(0.0)
    _, state = TC.split_saver_variables(dump)
    with pytest.raises(TC.CheckpointError, match="neither a global step"):
        Adam().load_state(state, names)
    # a model-only checkpoint leaves a fresh optimizer
    _, state = TC.split_saver_variables(dict(prm))
    opt = Adam()
    opt.load_state(state, names)
    assert opt.t == 0 and not opt.m


class _FakeNative:
    def __init__(self):
        self.calls = []

    def comm_reduce_in_step(self, on=True):
        self.calls.append(("reduce_in_step", on))

    def comm_unique_id(self):
        self.calls.append(("unique_id",))
        return b"\0" * 128

    def allreduce_f64(self, a):
        self.calls.append(("allreduce_f64", np.size(a)))
        return 2.0 * np.asarray(a, dtype=np.float64)            # "two ranks holding the same values"


def test_comm_env_in_a_single_process_is_the_identity(monkeypatch):
    """run_*(comm="env") without a launcher (or with --nproc-per-node 1): no communicator is opened, so the in-step
    reduce must not be switched on (rnnwf_comm_reduce_in_step would answer 'communicator not initialised')."""
    from rnnwavefunctions_amd import training as T
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    nat = _FakeNative()
    comm = T._resolve_comm("env", nat)
    assert (comm.rank, comm.world) == (0, 1) and nat.calls == []
    m = np.array([1.0, 2.0, 3.0, 0.0])
    assert np.array_equal(comm.reduce_moments(m), m) and np.array_equal(comm.allreduce(m), m)
    g = {"a": np.ones(3)}
    assert comm.allreduce_grads(g) is g and nat.calls == []
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    assert T._resolve_comm("env", nat).world == 1 and nat.calls == []


def test_rccl_shardcomm_never_returns_unreduced_arrays():
    """With the RCCL transport the step's moments arrive already summed (reduce_moments passes them through); every
    other array - gradients included - goes through rnnwf_allreduce_f64."""
    from rnnwavefunctions_amd import distributed as D
    nat = _FakeNative()
    comm = D.ShardComm.from_rccl(nat, 1, 2)
    assert nat.calls == [("reduce_in_step", True)]
    m = np.array([1.0, 2.0, 3.0, 0.0])
    assert np.array_equal(comm.reduce_moments(m), m)
    assert np.array_equal(comm.allreduce(m), 2 * m) and nat.calls[-1] == ("allreduce_f64", 4)
    nat.allreduce_grads = lambda g: {k: 2.0 * v for k, v in g.items()}
    g = comm.allreduce_grads({"w": np.ones((2, 2)), "b": np.arange(3.0)})
    assert np.array_equal(g["w"], 2 * np.ones((2, 2))) and np.array_equal(g["b"], 2 * np.arange(3.0))
    with pytest.raises(RuntimeError, match="no transport"):
        D.ShardComm(0, 2).allreduce(m)
    # a native communicator built through the constructor never switched the in-step reduce on: its moments must be reduced
    bare = D.ShardComm(1, 2, None, native=nat)
    assert np.array_equal(bare.reduce_moments(m), 2 * m) and nat.calls[-1] == ("allreduce_f64", 4)


_SANITIZED_CASES = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from conftest import load_golden, golden_params
from oracle import cport, models as M, estimators as E, philox
from rnnwavefunctions_amd import params as P
assert cport.lib()._name.endswith("librnnwf_oracle_san.so")
g = load_golden("estimators_oracle_driven.npz")
prm = golden_params(g, "g4a")
# G4a: the reference's own Ising_local_energies driven by the oracle (N = 10, 2 400 samples -> two <= 25 000-row chunks)
e, lp = cport.ising_local_energies(prm, g["g4a_Jz"], float(g["g4a_Bx"]), g["g4a_samples"], return_log_probs=True)
assert np.allclose(lp, g["g4a_logp"], rtol=0, atol=3e-5) and np.allclose(e, g["g4a_eloc"], rtol=2e-5, atol=2e-5)
e0 = cport.ising_local_energies(prm, g["g4a_Jz"], 0.0, g["g4a_samples"][:50])
assert np.allclose(e0, g["g4a_eloc_bx0"], atol=1e-12)
# ragged sizes around the 16-row register block, one thread and several, shortest chains, the widest cell
rng = np.random.RandomState(0)
for N, H, ns, nt in ((1, 3, 1, 1), (2, 5, 15, 1), (7, 20, 17, 3), (33, 50, 31, 0), (5, 100, 49, 2)):
    q = P.randomize_biases(P.scale_kernels(P.init_gru_params([H], seed=N), 1.5), N + 1)
    s = rng.randint(0, 2, (ns, N)).astype(np.int32)
    assert np.allclose(cport.prnn_log_probability(q, s, nthreads=nt), M.prnn_log_probability(q, s), atol=2e-5)
    ec = cport.ising_local_energies(q, np.ones(N), 0.7, s, nthreads=nt)
    assert np.allclose(ec, E.ising_local_energies(np.ones(N), 0.7, s, lambda x: M.prnn_log_probability(q, x)), rtol=3e-5, atol=3e-5)
    u = philox.uniforms(3, 1, 0, ns, N)
    sc, lc = cport.prnn_sample(q, N, u, nthreads=nt)
    sr, lr = M.prnn_sample(q, N, u)
    assert (sc != sr).any(axis=1).sum() <= 1
assert cport.prnn_log_probability(prm, np.zeros((0, 10), dtype=np.int32)).shape == (0,)
print("sanitized oracle ok")
"""


def test_c_oracle_under_address_and_undefined_behaviour_sanitizers():
    """sanitizers on the CPU side.  oracle/c/rnnwf_oracle.c built with -fsanitize=address,undefined
    (-fno-sanitize-recover) runs the reference-generated golden case G4a, ragged block sizes and empty input in a child
    process (libasan preloaded); any invalid access aborts it."""
    import subprocess
    import sys
    rt = cport.sanitizer_runtime()
    if rt is None:
        pytest.skip("gcc has no libasan.so here")
    cport.build_sanitized()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LD_PRELOAD=rt, RNNWF_ORACLE_SANITIZED="1", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="4", PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, "-c", _SANITIZED_CASES, root], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "sanitized oracle ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def _synthetic_bench_inputs(world, fallback=None, transport="rccl"):
    import argparse
    import bench
    wl = dict(bench.WORKLOADS["cfg2"], weights="init")
    ns, N = wl["ns"], wl["N"]
    args = argparse.Namespace(workload="cfg2", steps=20, warmup=3, transport=transport, weights="init")
    infos = [{"pid_rank": r, "rank": r, "nranks": world, "device": r} for r in range(world)]
    per_rank = [2.5 + 0.01 * r for r in range(world)]
    dt = max(per_rank) * 1e-3 * args.steps
    cell_evals = ns * N * (N - 1) / 2.0
    flip = {"launches": 20, "total_ms": 20 * 2.36, "cell_evals": 20 * cell_evals, "mfma_flops": 20 * 3.08e12 / 6}
    cfg5 = None
    if world > 1:
        w5 = bench.WORKLOADS["cfg5"]
        cfg5 = {"workload": w5["desc"], "steps": 2, "warmup": 1, "ms_per_step": 171.0, "value": world * w5["ns"] * w5["N"] / 0.171,
                "unit": "samples*sites/s", "global_numsamples": world * w5["ns"], "mean_E": -250.0, "note": "synthetic"}
    return dict(args=args, wl=wl, world=world, dt=dt, step_ms=[2.5] * args.steps, per_rank_ms=per_rank, infos=infos,
                moments=[-1.0e6, 1.1e8, float(world * ns), 0.0], flip=flip, base={"launches": 3, "total_ms": 0.6},
                asm={"launches": 3, "total_ms": 0.03}, engine="bf16x3", transport_fallback=fallback, cfg5=cfg5,
                traffic={"hbm_bytes_per_launch": 1.0e8, "held_clock_ghz": 1.72, "held_clock_source": "synthetic", "build": "test"},
                last_step=22)


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_bench_rank0_record_assembles_for_every_rank_count(world):
    """rank 0's JSON assembly driven with synthetic per-rank records, so that a first 8-GPU run cannot die
    in it: every contract field present, whole-job value, json-serialisable, exit code 0."""
    import json
    import bench
    kw = _synthetic_bench_inputs(world)
    rec = bench.assemble_record(**kw)
    rec["parity"] = None
    rec["cpu_baseline"] = None
    d = json.loads(json.dumps(rec))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    ns, N = kw["wl"]["ns"], kw["wl"]["N"]
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] == pytest.approx(world * ns * N / (d["ms_per_step"] * 1e-3))
    assert d["ms_per_step"] == pytest.approx(max(kw["per_rank_ms"]))          # the slowest rank sets the job's time
    assert len(d["ms_per_step_per_rank"]) == world and len(d["ranks"]) == world
    assert d["rccl_nranks"] == world and d["transport_fallback"] is None
    assert d["config"]["global_numsamples"] == world * ns
    r = d["roofline"]
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"]) and 0.3 < r["frac"] < 0.7
    assert ("cfg5_sharded" in d) == (world > 1)
    assert bench.exit_status(d) == (0, None)


def test_bench_exit_status_is_nonzero_on_red_parity_or_silent_fallback():
    """a red parity leg or a gloo fallback must not hand the driver an rc-0 number."""
    import bench
    rec = bench.assemble_record(**_synthetic_bench_inputs(1))
    good = {"pass": True, "reproduces_timed_step": True, "max_abs_dE_per_site": 1e-7, "d_meanE_per_site": 1e-9, "tolerance_per_site": 1e-4}
    assert bench.exit_status(dict(rec, parity=good))[0] == 0
    assert bench.exit_status(dict(rec, parity=dict(good, **{"pass": False})))[0] == 2
    assert bench.exit_status(dict(rec, parity=dict(good, reproduces_timed_step=False)))[0] == 2
    fb = bench.assemble_record(**_synthetic_bench_inputs(8, fallback="RCCL communicator could not be created (rank 3: x)"))
    assert fb["rccl_nranks"] is None
    assert bench.exit_status(fb)[0] == 3 and bench.exit_status(fb, allow_fallback=True)[0] == 0
    # a communicator that spans fewer ranks than the job (N x dp1 passing for dp-N)
    short = bench.assemble_record(**_synthetic_bench_inputs(8))
    short["rccl_nranks"] = 1
    assert bench.exit_status(short)[0] == 3
