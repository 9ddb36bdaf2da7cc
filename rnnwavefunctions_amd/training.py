"""VMC training drivers with the reference's call signatures.

    run_1DTFIM        <- 1DTFIM/TrainingRNN_1DTFIM.py:79-229
    run_J1J2          <- J1J2/TrainingRNN_J1J2.py:131-308
    run_2DTFIM_2DRNN  <- 2DTFIM_2DRNN/Training2DRNN_2DTFIM.py:88-231   (exported there as run_2DTFIM)
    run_2DTFIM_1DRNN  <- 2DTFIM_1DRNN/Training1DRNN_2DTFIM.py:85-233   (exported there as run_2DTFIM)

One iteration of the reference loop (:199-227) is: draw samples, local energies, mean/var, print every 10 steps,
Adam step on  cost = mean(log_probs * Eloc) - mean(Eloc) * mean(log_probs)  (:156).  Here the whole iteration but
the 8 102-parameter Adam update runs on the GPU: `vmc_step` (sample + fused local energies + moments) and
`vmc_gradient` (back-propagation through time + weight-gradient GEMM) keep samples, hidden states and local
energies resident in HBM; the host only sees the four moments and the gradient arrays.
"""
import os

import numpy as np

from . import _lib
from . import distributed as D
from . import params as P
from . import tf_checkpoint as T


class Adam:
    """tf.train.AdamOptimizer(learning_rate) with TF-1 defaults (beta1 0.9, beta2 0.999, epsilon 1e-8):
    lr_t = lr sqrt(1 - beta2^t) / (1 - beta1^t);  theta -= lr_t m / (sqrt(v) + epsilon)."""

    def __init__(self, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.b1, self.b2, self.eps = beta1, beta2, epsilon
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, params, grads, lr):
        self.t += 1
        lr_t = lr * np.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for k, g in grads.items():
            g = np.asarray(g, dtype=np.float64)
            if k not in self.m:
                self.m[k] = np.zeros_like(g)
                self.v[k] = np.zeros_like(g)
            self.m[k] = self.b1 * self.m[k] + (1.0 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1.0 - self.b2) * g * g
            params[k] = (params[k].astype(np.float64) - lr_t * self.m[k] / (np.sqrt(self.v[k]) + self.eps)).astype(params[k].dtype)
        return params

    # -- what tf.train.Saver() keeps of the optimizer (TrainingRNN_1DTFIM.py:163-166): slots <var>/Adam, <var>/Adam_1,
    #    beta1_power / beta2_power (beta^(t+1) after t steps, float32 scalars) and the un-named global step `Variable`.
    #    NAMES ARE UNPINNED (no TF-written checkpoint exists here).  The reference enters `tf.variable_scope(wf.scope)`
    #    (:149) BEFORE `with wf.graph.as_default()` (:150): the scope therefore binds to the process default graph, not to
    #    wf.graph (a separate tf.Graph(), RNNwavefunction.py:20), and TF-1 keeps the variable-scope store and the name stack
    #    per graph.  Inside wf.graph both are at root when apply_gradients runs - the same reason the global step is the
    #    bare `Variable` - so the slot creator names the slots after the variable alone, "<var>/Adam" / "<var>/Adam_1", and
    #    the beta powers carry no prefix.  The reader (load_state) accepts any scope prefix, so files written with the
    #    nested names of round 3 ("<scope>/<scope>/.../Adam", "<scope>_2/beta1_power") still load.
    def state_tensors(self, params, scope):
        out = {}
        for k, v in params.items():
            out[k + "/Adam"] = self.m.get(k, np.zeros(v.shape)).astype(v.dtype)
            out[k + "/Adam_1"] = self.v.get(k, np.zeros(v.shape)).astype(v.dtype)
        out["beta1_power"] = np.array(self.b1 ** (self.t + 1), dtype=np.float32)
        out["beta2_power"] = np.array(self.b2 ** (self.t + 1), dtype=np.float32)
        out["Variable"] = np.array(self.t, dtype=np.int32)
        return out

    def load_state(self, opt, names):
        """`opt` as returned by tf_checkpoint.split_saver_variables.  A slot belongs to the model variable whose name it
        ends with once every leading repeat of the scope ("RNNwavefunction/", "RNNwavefunction_1/", ...) is stripped
        from both - so "<scope>/<scope>/x/Adam" (TF-1's nesting), "<scope>/x/Adam" and "x/Adam" all find "<scope>/x".
        Raises tf_checkpoint.CheckpointError when the file holds Adam slots but some variable gets none, or when the step
        count cannot be recovered: a silent restart with zero moments is never taken."""
        import re

        def tail(name):
            parts = name.split("/")
            first = re.sub(r"_\d+$", "", names[0].split("/")[0]) if names else ""
            while len(parts) > 1 and re.sub(r"_\d+$", "", parts[0]) == first:
                parts = parts[1:]
            return "/".join(parts)
        m = {tail(k): v for k, v in opt["m"].items()}
        v2 = {tail(k): v for k, v in opt["v"].items()}
        if not m and not v2:
            return                                             # a model-only checkpoint: fresh optimizer
        missing = [k for k in names if tail(k) not in m or tail(k) not in v2]
        if missing:
            raise T.CheckpointError("the checkpoint holds Adam slots but none for %s (slot names found: %s)" %
                                    (", ".join(missing), ", ".join(sorted(opt["m"])[:4]) + " ..."))
        for k in names:
            self.m[k] = np.asarray(m[tail(k)], dtype=np.float64)
            self.v[k] = np.asarray(v2[tail(k)], dtype=np.float64)
        if opt.get("global_step") is not None:
            self.t = int(opt["global_step"])
            return
        # beta2^(t+1) stays a normal float32 to t ~ 1e5; beta1^(t+1) is zero from t ~ 980 on
        for key, beta in (("beta2_power", self.b2), ("beta1_power", self.b1)):
            p = opt.get(key)
            if p is not None and 0.0 < p < 1.0 and np.log(p) / np.log(beta) < 2 ** 20 and p > 1e-30:
                self.t = max(int(round(np.log(p) / np.log(beta))) - 1, 0)
                return
        raise T.CheckpointError("the checkpoint holds Adam slots but neither a global step nor a usable beta power")


    # -- the optimizer state as the device-resident path keeps it (rnnwf_adam_get_state / rnnwf_adam_set_state): flat, in the
    #    library's tensor order
    def to_flat(self, native, params, scope):
        lay = native._layout()
        n = sum(cnt for _, cnt in lay)
        m, v, off = np.zeros(n), np.zeros(n), 0
        for nm, cnt in lay:
            k = scope + "/" + nm
            if k in self.m:
                m[off:off + cnt] = np.asarray(self.m[k], dtype=np.float64).ravel()
                v[off:off + cnt] = np.asarray(self.v[k], dtype=np.float64).ravel()
            off += cnt
        return m, v

    def from_flat(self, native, params, scope, m, v, t):
        off = 0
        for nm, cnt in native._layout():
            k = scope + "/" + nm
            self.m[k] = m[off:off + cnt].reshape(params[k].shape).copy()
            self.v[k] = v[off:off + cnt].reshape(params[k].shape).copy()
            off += cnt
        self.t = int(t)


# Device-resident iterations (rnnwf_train_steps): on by default where the library supports them (single-layer float32 GRU models);
# DEVICE_TRAINING = False (or RNNWF_HOST_ADAM=1) keeps the optimizer on the host for every model - the two give the same trajectory
# bit for bit (tests/test_gpu_training.py).
DEVICE_TRAINING = os.environ.get("RNNWF_HOST_ADAM", "0") != "1"


def cost_gradient(native, params, scope, mean_energy, norm, allreduce=False):
    """{scoped tf name: gradient} of the reference cost on the batch of the last vmc_step."""
    shapes = {k[len(scope) + 1:]: v.shape for k, v in params.items()}
    g = native.vmc_gradient(mean_energy, norm, shapes, allreduce=allreduce)
    return {scope + "/" + k: v for k, v in g.items()}


def _train(wf, params, scope, couplings, numsteps, numsamples, seed, lr, lr_of_it, opt, complex_energy, comm, verbose,
           on_step=None, history=None):
    """The loop all four drivers share (1DTFIM/TrainingRNN_1DTFIM.py:199-227 and its siblings): sample + local
    energies + moments on the GPU, mean/var, print every 10 steps, gradient of the cost, Adam.

    Sharded over `comm.world` processes (one per GPU): rank r draws the global samples shard_range(numsamples, r, world)
    - the RNG is keyed by the global index, so the union is the single-process batch -, the four moments and the
    gradient partial sums are all-reduced, and every rank applies the identical Adam step."""
    comm = comm or D.ShardComm()
    offset, count = D.shard_range(numsamples, comm.rank, comm.world)
    meanEnergy, varEnergy = history if history is not None else ([], [])
    # the whole iteration on the device, ten at a time (the reference prints and saves every 10 steps, :213-227): single process,
    # or RCCL with the in-step all-reduce of the moments (the gradient then takes one in-stream all-reduce too)
    if DEVICE_TRAINING and (comm.world == 1 or getattr(comm, "_moments_in_step", False)) and wf.device_training_supported():
        return _train_on_device(wf, params, scope, couplings, numsteps, numsamples, seed, lr, lr_of_it, opt, complex_energy, comm,
                                verbose, on_step, meanEnergy, varEnergy, offset, count)
    for it in range(len(meanEnergy), numsteps + 1):       # `for it in range(len(meanEnergy),numsteps+1)` (:199): a restored run resumes
        m = wf.vmc_step(count, seed=seed, step=it, couplings=couplings, sample_offset=offset)["moments"]
        s1, s2, n, si = comm.reduce_moments(m)
        meanE = complex(s1 / n, si / n) if complex_energy else s1 / n
        varE = s2 / n - (s1 / n) ** 2
        meanEnergy.append(np.complex64(meanE) if complex_energy else meanE)
        varEnergy.append(varE)
        if verbose and comm.rank == 0 and it % 10 == 0:
            print("mean(E): {0}, var(E): {1}, #samples {2}, #Step {3} \n\n".format(meanEnergy[-1], varE, numsamples, it))
        if on_step is not None and comm.rank == 0:
            on_step(it, meanEnergy, varEnergy, params, opt)
        native_reduce = comm.native is not None and comm.world > 1
        grads = cost_gradient(wf, params, scope, meanE, n, allreduce=native_reduce)
        if not native_reduce:
            grads = comm.allreduce_grads(grads)
        params = opt.step(params, grads, lr_of_it(lr, it))
        wf.set_params(params, scope=scope)
    return meanEnergy, varEnergy, params


def _train_on_device(wf, params, scope, couplings, numsteps, numsamples, seed, lr, lr_of_it, opt, complex_energy, comm, verbose,
                     on_step, meanEnergy, varEnergy, offset, count):
    """_train's loop with rnnwf_train_steps: chunks that end where the reference prints / saves (every 10 steps), one host
    synchronisation per chunk.  What the loop shows the outside - prints, the histories handed to on_step, the parameters and
    optimizer state a checkpoint holds - is what the per-iteration loop shows at the same iteration."""
    m0, v0 = opt.to_flat(wf, params, scope)
    wf.adam_set_state(m0 if opt.t else None, v0 if opt.t else None, opt.t)
    it = len(meanEnergy)
    while it <= numsteps:
        K = min(10 - it % 10, numsteps + 1 - it)
        if on_step is not None and comm.rank == 0 and it % 500 == 0:      # the checkpoint of iteration `it` holds the state BEFORE its update
            params = wf.get_params_dict(params, scope)
            opt.from_flat(wf, params, scope, *wf.adam_get_state())
        mom = wf.train_steps(count, seed, it, couplings, [float(lr_of_it(lr, j)) for j in range(it, it + K)], opt.b1, opt.b2, opt.eps,
                             sample_offset=offset)
        for j in range(K):
            s1, s2, n, si = mom[j]
            meanE = complex(s1 / n, si / n) if complex_energy else s1 / n
            meanEnergy.append(np.complex64(meanE) if complex_energy else meanE)
            varEnergy.append(s2 / n - (s1 / n) ** 2)
            if verbose and comm.rank == 0 and (it + j) % 10 == 0:
                print("mean(E): {0}, var(E): {1}, #samples {2}, #Step {3} \n\n".format(meanEnergy[-1], varEnergy[-1], numsamples, it + j))
            if on_step is not None and comm.rank == 0:
                on_step(it + j, meanEnergy, varEnergy, params, opt)
        it += K
    params = wf.get_params_dict(params, scope)
    opt.from_flat(wf, params, scope, *wf.adam_get_state())
    return meanEnergy, varEnergy, params


def _resolve_comm(comm, wf):
    """comm=None: single process; a distributed.ShardComm: used as is; "env": one process per GPU launched by
    `python -m torch.distributed.run` - the launcher's (gloo) group carries RCCL's unique id, all-reduces run on RCCL."""
    if comm == "env":
        rank, world = D.init_rccl_from_env(wf)
        return D.ShardComm.from_rccl(wf, rank, world)
    return comm


def _resolve_device(device, comm):
    """device=None (the drivers' default): GPU 0, or - one process per GPU under `python -m torch.distributed.run`
    (comm="env") - the launcher's LOCAL_RANK, so that every rank opens its own GPU (RCCL refuses two ranks on one)."""
    if device is not None:
        return int(device)
    return int(os.environ.get("LOCAL_RANK", "0")) if comm == "env" else 0


def _fit(wf, params, scope, couplings, numsteps, numsamples, seed, lr, lr_of_it, opt, complex_energy, comm, verbose,
         save_dir, tags, restore):
    """_train with the reference's saving cadence and, with restore=True, its restore branch in front."""
    history = None
    if restore:
        if save_dir is None:
            raise ValueError("restore=True needs save_dir (where the checkpoint and the energy histories live)")
        params, history = _restore(save_dir, tags[0], tags[1], tags[2], params, opt, scope)
        wf.set_params(params, scope=scope)
    return _train(wf, params, scope, couplings, numsteps, numsamples, seed, lr, lr_of_it, opt, complex_energy, comm,
                  verbose, _saver(save_dir, tags[0], tags[1], tags[2], scope), history)


def _saver(save_dir, tag_mean, tag_var, tag_model, scope="RNNwavefunction"):
    """The reference's saving cadence (:217-227): energies every 10 steps as .npy, the model every 500 steps as a TF
    checkpoint `<tag_model>.ckpt` (V2 tensor bundle written by tf_checkpoint.py: the model variables under their TF
    names plus what tf.train.Saver() holds of the optimizer - Adam slots, beta powers, global step)."""
    if save_dir is None:
        return None

    def on_step(it, meanEnergy, varEnergy, params, opt=None):
        if it % 500 == 0:
            extra = opt.state_tensors(params, scope) if opt is not None else {}
            T.write_checkpoint(os.path.join(save_dir, tag_model + ".ckpt"), dict(params, **extra))
        if it % 10 == 0:
            np.save(os.path.join(save_dir, tag_mean + ".npy"), meanEnergy)
            np.save(os.path.join(save_dir, tag_var + ".npy"), varEnergy)
    return on_step


def _restore(save_dir, tag_mean, tag_var, tag_model, params, opt, scope):
    """The reference's restore branch (:172-183, commented out there): model + optimizer from the checkpoint, the
    energy histories from the .npy files; training then resumes at iteration len(meanEnergy)."""
    from .wavefunctions import match_checkpoint_names
    model, ostate = T.split_saver_variables(T.read_checkpoint(os.path.join(save_dir, tag_model + ".ckpt")))
    params = match_checkpoint_names(params, model, scope)
    opt.load_state(ostate, list(params))
    meanEnergy = np.load(os.path.join(save_dir, tag_mean + ".npy")).tolist()
    varEnergy = np.load(os.path.join(save_dir, tag_var + ".npy")).tolist()
    # the checkpoint is written every 500 steps, the histories every 10: resume from the checkpoint's step
    keep = min(opt.t, len(meanEnergy))
    return params, (meanEnergy[:keep], varEnergy[:keep])


def run_1DTFIM(numsteps=10 ** 4, systemsize=20, num_units=50, Bx=1, num_layers=1, numsamples=500, learningrate=5e-3,
               seed=111, save_dir=None, device=None, verbose=True, comm=None, restore=False, parity_symmetric=False):
    """Train the 1D pRNN wave function on the open transverse-field Ising chain; returns (meanEnergy, varEnergy)
    lists with one entry per iteration, as the reference's run_1DTFIM.  `comm` (distributed.ShardComm, or "env" under
    torch.distributed.run) shards the batch over one process per GPU; `save_dir` turns on the reference's saving
    (energies every 10 steps, TF checkpoint every 500), `restore=True` its restore branch (:172-183).
    `parity_symmetric=True` is the reference's import switch to RNNwavefunction_paritysym (1DTFIM/TrainingRNN_1DTFIM.py:10):
    P_sym(s) = (P(s) + P(reversed s)) / 2."""
    if not 1 <= num_layers <= 4:
        raise ValueError("num_layers must be 1..4 (stacked layers: num_units <= 100)")
    N = systemsize
    scope = "RNNwavefunction"
    Jz = +np.ones(N)
    units = [num_units] * num_layers
    params = P.init_gru_params(units, seed=seed, scope=scope)
    wf = _lib.NativeWavefunction(_lib.MODEL_GRU1D_PARITY if parity_symmetric else _lib.MODEL_GRU1D, N, 1, tuple(units),
                                 device=_resolve_device(device, comm))
    wf.set_params(params, scope=scope)
    comm = _resolve_comm(comm, wf)
    if verbose and (comm is None or comm.rank == 0):
        for k, v in params.items():
            print(k, (v.size,))
        print("The number of params is {0}".format(P.count_params(params)))
    ending = "_units" + "".join("_{0}".format(u) for u in units)
    tag = "_N" + str(N) + "_samp" + str(numsamples) + "_Jz" + str(Jz[0]) + "_Bx" + str(Bx) + "_GRURNN_OBC_TFIM" + ending
    meanEnergy, varEnergy, params = _fit(
        wf, params, scope, np.append(Jz, float(Bx)), numsteps, numsamples, seed, np.float64(learningrate),
        lambda lr0, it: lr0, Adam(), False, comm, verbose, save_dir,
        ("meanEnergy" + tag, "varEnergy" + tag, "RNNwavefunction" + tag), restore)
    run_1DTFIM.last_params = params
    return meanEnergy, varEnergy


def run_J1J2(numsteps=10 ** 5, systemsize=20, J1_=1.0, J2_=0.0, Marshall_sign=False, num_units=50, num_layers=1,
             numsamples=500, learningrate=2.5 * 1e-4, seed=111, save_dir=None, device=None, verbose=True, comm=None,
             restore=False):
    """Train the complex RNN wave function (U(1) zero magnetisation) on the open J1-J2 chain; returns
    (meanEnergy, varEnergy) as the reference's run_J1J2 (meanEnergy complex, varEnergy = var of the real part).

    As in the reference, `Marshall_sign` reaches J1J2MatrixElements through J1J2Slices' `periodic` slot
    (J1J2/TrainingRNN_J1J2.py:118): Marshall_sign=True therefore selects the PERIODIC chain
    without a Marshall sign - reproduced here on purpose so that runs compare with the reference's."""
    if not 1 <= num_layers <= 4:
        raise ValueError("num_layers must be 1..4 (stacked layers: num_units <= 100)")
    N = systemsize
    scope = "RNNwavefunction"
    lr = np.float64(learningrate)
    units = [num_units] * num_layers
    params = P.init_gru_params(units, seed=seed, scope=scope, heads=("wf_dense_ampl", "wf_dense_phase"))
    wf = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, tuple(units), device=_resolve_device(device, comm))
    wf.set_params(params, scope=scope)
    comm = _resolve_comm(comm, wf)
    if verbose and (comm is None or comm.rank == 0):
        print("The number of params is {0}".format(P.count_params(params)))
    periodic = 1.0 if Marshall_sign else 0.0          # the reference's quirk, see the docstring
    couplings = np.concatenate([J1_ * np.ones(N), J2_ * np.ones(N), np.zeros(N), [periodic, 0.0]])
    ending = "_units" + "".join("_{0}".format(u) for u in units)
    tag = "_N" + str(N) + "_samp" + str(numsamples) + "_lradap" + str(lr) + "_complexGRURNN_J1J2" + str(float(J2_)) + ending + "_zeromag"
    meanEnergy, varEnergy, params = _fit(
        wf, params, scope, couplings, numsteps, numsamples, seed, lr, lambda lr0, it: lr0,
        Adam(beta1=0.9, beta2=0.999, epsilon=1e-8), True, comm, verbose, save_dir,
        ("meanEnergy" + tag, "varEnergy" + tag, "RNNwavefunction" + tag), restore)
    run_J1J2.last_params = params
    return meanEnergy, varEnergy


def _run_2d(model, params, units, Nx, Ny, Bx, numsteps, numsamples, lr, lr_of_it, seed, save_dir, tag, device, verbose,
            comm, restore=False):
    """Shared part of the two 2D drivers (both: cost of TrainingRNN_1DTFIM.py:156 on float64 wave functions,
    default Adam, learning rate adapted per iteration)."""
    scope = "RNNwavefunction"
    wf = _lib.NativeWavefunction(model, Nx, Ny, tuple(units), device=_resolve_device(device, comm))
    wf.set_params(params, scope=scope)
    comm = _resolve_comm(comm, wf)
    if verbose and (comm is None or comm.rank == 0):
        print("The number of params is {0}".format(P.count_params(params)))
    couplings = np.append(np.ones(Nx * Ny), float(Bx))          # Jz = +np.ones((Nx, Ny))
    return _fit(wf, params, scope, couplings, numsteps, numsamples, seed, lr, lr_of_it, Adam(), False, comm, verbose,
                save_dir, ("meanEnergy_" + tag, "varEnergy_" + tag, "RNNwavefunction_" + tag), restore)


def run_2DTFIM_2DRNN(numsteps=2 * 10 ** 4, systemsize_x=5, systemsize_y=5, Bx=+2, num_units=50, numsamples=500,
                     learningrate=5e-3, seed=111, save_dir=None, device=None, verbose=True, comm=None, restore=False):
    """Train the 2D MDRNN (float64, zig-zag path) on the open square-lattice transverse-field Ising model;
    learning rate  lr (1 + it/5000)^-1  (Training2DRNN_2DTFIM.py:228).  The reference builds Jz from Nx, Ny one
    line before it defines them (:96-99, a NameError as published); here the sizes are read first."""
    Nx, Ny = systemsize_x, systemsize_y
    lr = np.float64(learningrate)
    units = [num_units]
    params = P.init_mdrnn_params(num_units, seed=seed)
    tag = "2DVanillaRNN_" + str(Nx) + "x" + str(Ny) + "_Bx" + str(Bx) + "_lradap" + str(lr) + "_samp" + str(numsamples) + \
        "_units" + "".join("_{0}".format(u) for u in units)
    meanE, varE, params = _run_2d(_lib.MODEL_MDRNN2D, params, units, Nx, Ny, Bx, numsteps, numsamples, lr,
                                  lambda lr0, it: lr0 * (1 + it / 5000) ** (-1), seed, save_dir, tag, device, verbose, comm,
                                  restore)
    run_2DTFIM_2DRNN.last_params = params
    return meanE, varE


def run_2DTFIM_1DRNN(numsteps=2 * 10 ** 4, systemsize_x=5, systemsize_y=5, Bx=+2, num_units=50, num_layers=1,
                     numsamples=500, learningrate=1e-3, seed=333, save_dir=None, device=None, verbose=True, comm=None,
                     restore=False):
    """Train the float64 1D GRU wave function over the raster path of the square lattice; learning rate
    1 / (1/lr + it/10)  (Training1DRNN_2DTFIM.py:231).  The reference seeds numpy / TF with `seed` but builds the
    wave function with its class default seed 111 (:104); the initial weights here follow the latter."""
    if not 1 <= num_layers <= 4:
        raise ValueError("num_layers must be 1..4 (stacked float64 layers: num_units <= 68)")
    Nx, Ny = systemsize_x, systemsize_y
    lr = np.float64(learningrate)
    units = [num_units] * num_layers
    params = P.init_gru_params(units, seed=111, dtype=np.float64)
    tag = "GRURNN_" + str(Nx) + "x" + str(Ny) + "_Bx" + str(Bx) + "_lradap" + str(lr) + "_samp" + str(numsamples) + \
        "_units" + "".join("_{0}".format(u) for u in units)
    meanE, varE, params = _run_2d(_lib.MODEL_GRU1D_F64, params, units, Nx, Ny, Bx, numsteps, numsamples, lr,
                                  lambda lr0, it: 1.0 / ((1.0 / lr0) + it / 10), seed, save_dir, tag, device, verbose, comm,
                                  restore)
    run_2DTFIM_1DRNN.last_params = params
    return meanE, varE
