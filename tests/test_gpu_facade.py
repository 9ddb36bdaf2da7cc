"""GPU tests of the drop-in Python layer: the reference's own call sequence
(1DTFIM/TrainingRNN_1DTFIM.py:189-207) against the reference-named modules of this package."""
import numpy as np
import pytest

from oracle import estimators as E
from oracle import models as M
from oracle import philox

pytestmark = pytest.mark.gpu


def test_1dtfim_loop_reads_like_the_reference():
    from rnnwavefunctions_amd import compat as tf
    from rnnwavefunctions_amd.TFIM1D.TrainingRNN_1DTFIM import Ising_local_energies, RNNwavefunction
    N, numsamples, Bx = 20, 500, 1.0
    Jz = +np.ones(N)
    wf = RNNwavefunction(N, units=[20], cell=tf.CudnnCompatibleGRUCell, seed=111)
    assert wf.num_params() == 1442                      # h=20 -> 1 442
    sess = tf.Session(graph=wf.graph, config=tf.ConfigProto())
    with wf.graph.as_default():
        samples_ = wf.sample(numsamples=numsamples, inputdim=2)
        samples_placeholder = tf.placeholder(dtype=tf.int32, shape=(None, N))
        log_probs_tensor = wf.log_probability(samples_placeholder, inputdim=2)
    queue_samples = np.zeros((N + 1, numsamples, N), dtype=np.int32)
    log_probs = np.zeros((N + 1) * numsamples, dtype=np.float64)

    samples = sess.run(samples_)
    assert samples.dtype == np.int64 and samples.shape == (numsamples, N)
    local_energies = Ising_local_energies(Jz, Bx, samples, queue_samples, log_probs_tensor, samples_placeholder,
                                          log_probs, sess)
    meanE, varE = np.mean(local_energies), np.var(local_energies)
    assert np.isfinite(meanE) and varE > 0

    prm = wf.get_params()
    # same stream as the oracle: batch 0 of seed 111
    s_ref, _ = M.prnn_sample(prm, N, philox.uniforms(111, 0, 0, numsamples, N))
    assert (samples != s_ref).any(axis=1).sum() <= 1
    e_ref, lp_ref = E.ising_local_energies(Jz, Bx, samples, lambda x: M.prnn_log_probability(prm, x),
                                           return_log_probs=True)
    assert np.allclose(local_energies, e_ref, rtol=2e-5)
    assert np.allclose(log_probs, lp_ref.ravel(), atol=5e-5)          # the caller's scratch is filled
    assert np.array_equal(queue_samples[0], samples)

    # the reference formulation (host queue + chunked sess.run) through the same objects
    log_probs2 = np.zeros_like(log_probs)
    e2 = Ising_local_energies(Jz, Bx, samples, queue_samples, log_probs_tensor, samples_placeholder, log_probs2,
                              sess, mode="reference")
    assert np.allclose(e2, local_energies, rtol=2e-6)       # same kernels, f32 rounding order only
    assert np.array_equal(queue_samples[3][:, 2], 1 - samples[:, 2])

    # a second sess.run draws the next batch
    assert not np.array_equal(sess.run(samples_), samples)
    # eager conveniences
    assert np.allclose(wf.log_probability(samples[:7], 2), lp_ref[0, :7], atol=5e-5)

    with pytest.raises(TypeError):
        Ising_local_energies(Jz, Bx, samples, queue_samples, object(), samples_placeholder, log_probs, sess)
    with pytest.raises(ValueError):
        wf.sample(10, inputdim=3)


def test_weights_roundtrip_through_npz(tmp_path):
    from rnnwavefunctions_amd.TFIM1D.RNNwavefunction import RNNwavefunction
    a = RNNwavefunction(12, units=[10], seed=1)
    b = RNNwavefunction(12, units=[10], seed=2)
    s = np.random.RandomState(0).randint(0, 2, (20, 12))
    assert not np.allclose(a.log_probability(s, 2), b.log_probability(s, 2))
    a.save(tmp_path / "wf.npz")
    b.restore(tmp_path / "wf.npz")
    assert np.array_equal(a.log_probability(s, 2), b.log_probability(s, 2))
    assert a.num_params() == 422                        # Tutorial_1DTFIM.ipynb cell 15


def test_2dtfim_1drnn_facade(golden_estimators):
    from conftest import golden_params
    from rnnwavefunctions_amd import compat as tf
    from rnnwavefunctions_amd.TFIM2D_1DRNN.Training1DRNN_2DTFIM import Ising2D_local_energies, RNNwavefunction
    g = golden_estimators
    Nx, Ny = (int(v) for v in g["g4c_shape"])
    wf = RNNwavefunction(Nx, Ny, units=[7], cell=tf.CudnnCompatibleGRUCell)
    wf.set_params(golden_params(g, "g4c"))
    ph = tf.placeholder(tf.int32, shape=(None, Nx * Ny))
    t = wf.log_probability(ph, 2)
    s = g["g4c_samples"]
    e = Ising2D_local_energies(g["g4c_Jz"], float(g["g4c_Bx"]), Nx, Ny, s, None, t, ph, None, tf.Session())
    assert np.allclose(e, g["g4c_eloc"], rtol=1e-10)


def test_init_params_in_the_library_equals_the_python_initialiser():
    """rnnwf_init_params (C ABI) and params.init_* (Python) share generator, draw order and rounding."""
    from rnnwavefunctions_amd import _lib
    from rnnwavefunctions_amd import params as P
    cases = [(_lib.MODEL_GRU1D, (10, 1), (12,), lambda s: P.init_gru_params([12], seed=s)),
             (_lib.MODEL_GRU1D, (10, 1), (8, 8), lambda s: P.init_gru_params([8, 8], seed=s)),
             (_lib.MODEL_CRNN_U1, (10, 1), (9,), lambda s: P.init_gru_params([9], seed=s, heads=("wf_dense_ampl", "wf_dense_phase"))),
             (_lib.MODEL_GRU1D_F64, (3, 4), (7,), lambda s: P.init_gru_params([7], seed=s, dtype=np.float64)),
             (_lib.MODEL_MDRNN2D, (3, 3), (6,), lambda s: P.init_mdrnn_params(6, seed=s))]
    for model, (nx, ny), units, make in cases:
        wf = _lib.NativeWavefunction(model, nx, ny, units)
        wf.init_params(111)
        ref = make(111)
        assert wf.num_params() == P.count_params(ref)
        for name, v in ref.items():
            got = wf.get_param(name[len("RNNwavefunction/"):], v.shape, dtype=np.float64)
            assert np.array_equal(got, v.astype(np.float64)), name
        s = wf.sample(8, seed=1, step=0)                       # committed: usable at once
        assert s.shape[0] == 8
    with pytest.raises(ValueError, match="32 bits"):
        wf.init_params(2 ** 33)


def test_graph_mode_surface_counts_samples_scores_and_saves(tmp_path, capsys):
    """What a reference script does before and around its loop, through `rnnwavefunctions_amd.compat` as `tf` and the reference-named
    modules of this package (tests/graph_mode.py makes the calls): the trainable variables and their count (422 at 10 units,
    Tutorial_1DTFIM.ipynb cell 15), a sampling tensor, the estimator with its scratch arrays, and tf.train.Saver - whose file is a
    TF V2 bundle under the TF variable names that a second wave function restores."""
    import rnnwavefunctions_amd.compat as tf
    from graph_mode import GraphModeVMC
    from rnnwavefunctions_amd.TFIM1D.TrainingRNN_1DTFIM import Ising_local_energies, RNNwavefunction
    from rnnwavefunctions_amd import tf_checkpoint as T
    tf.compat.v1.logging.set_verbosity(tf.compat.v1.logging.ERROR)
    tf.reset_default_graph()
    tf.set_random_seed(111)
    N, batch = 12, 40
    wf = RNNwavefunction(N, units=[10], cell=tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell, seed=111)
    gm = GraphModeVMC(tf, wf, batch)
    with wf.graph.as_default():
        names = [v.name for v in tf.trainable_variables()]
        sizes = [tf.reshape(v, [-1]).shape[0] for v in gm.sess.run(names)]
        saver = tf.train.Saver()
    assert len(names) == 8 and sum(sizes) == 422 and gm.optimizer.beta1 == 0.9
    assert gm.schedule.value({gm.lr_in: np.float64(5e-3)}) == 5e-3
    Jz, queue, scratch = np.ones(N), np.zeros((N + 1, batch, N), dtype=np.int32), np.zeros((N + 1) * batch)
    drawn = gm.samples()
    e = Ising_local_energies(Jz, 1.0, drawn, queue, gm.score, gm.any_in, scratch, gm.sess)
    prm = wf.get_params()
    assert drawn.shape == (batch, N) and np.allclose(e, E.ising_local_energies(Jz, 1.0, drawn, lambda x: M.prnn_log_probability(prm, x)), rtol=2e-5)
    path = saver.save(gm.sess, str(tmp_path / "RNNwavefunction.ckpt"))
    stored = [n for n, _, _ in T.list_variables(path)]
    assert "RNNwavefunction/multi_rnn_cell/cell_0/cudnn_compatible_gru_cell/gates/kernel" in stored
    other = RNNwavefunction(N, units=[10], cell=tf.contrib.cudnn_rnn.CudnnCompatibleGRUCell, seed=5, scope="another")
    assert not np.allclose(other.log_probability(drawn[:5], 2), wf.log_probability(drawn[:5], 2))
    with other.graph.as_default():
        tf.train.Saver().restore(tf.Session(graph=other.graph), path)
    assert np.array_equal(other.log_probability(drawn[:5], 2), wf.log_probability(drawn[:5], 2))
    assert [v.name for v in wf.rnn.variables][0].endswith("gates/kernel:0") and wf.dense.count_params() == 22


def test_training_saves_tf_checkpoints_and_resumes(tmp_path):
    """save_dir: energies every 10 steps, a TF checkpoint (model + Adam slots + step) every 500; restore=True is the
    reference's commented restore branch (TrainingRNN_1DTFIM.py:172-183): the resumed run continues the trajectory."""
    from rnnwavefunctions_amd import tf_checkpoint as T
    from rnnwavefunctions_amd.training import run_1DTFIM
    kw = dict(systemsize=8, num_units=6, numsamples=50, learningrate=1e-2, seed=3, verbose=False)
    mE_full, _ = run_1DTFIM(numsteps=520, save_dir=None, **kw)
    d = str(tmp_path)
    mE_a, _ = run_1DTFIM(numsteps=505, save_dir=d, **kw)            # writes the checkpoint of step 500
    ck = [f for f in __import__("os").listdir(d) if f.endswith(".ckpt.index")]
    assert len(ck) == 1
    tensors = T.read_checkpoint(__import__("os").path.join(d, ck[0][:-len(".index")]))
    model, opt = T.split_saver_variables(tensors)
    assert len(model) == 8 and len(opt["m"]) == 8 and opt["global_step"] == 500
    mE_b, _ = run_1DTFIM(numsteps=520, save_dir=d, restore=True, **kw)
    assert len(mE_b) == 521 and np.allclose(mE_b[:500], mE_a[:500], rtol=0, atol=0)
    assert np.allclose(mE_b[500:], mE_full[500:], rtol=1e-5)          # same trajectory as the uninterrupted run
