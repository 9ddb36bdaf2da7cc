"""ctypes binding of librnnwf_hip.so (include/rnnwf.h) and a thin NumPy-facing handle class.

There is deliberately no fallback: if the HIP library is not built, or no gfx950 device is
visible, the calls below raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "librnnwf_hip.so")

MODEL_GRU1D, MODEL_GRU1D_PARITY, MODEL_CRNN_U1, MODEL_GRU1D_F64, MODEL_MDRNN2D = range(5)
F32, F64 = 0, 1
ABI_VERSION = 1
MAX_LAYERS = 4
UNIQUE_ID_BYTES = 128


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("model", C.c_int32), ("nx", C.c_int32), ("ny", C.c_int32),
                ("num_layers", C.c_int32), ("units", C.c_int32 * MAX_LAYERS), ("device", C.c_int32),
                ("reserved", C.c_int32 * 6)]


_P = C.c_void_p
_I32P, _F64P, _F32P = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float)
_I64, _U64, _I32, _F64 = C.c_int64, C.c_uint64, C.c_int32, C.c_double

# every symbol include/rnnwf.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "rnnwf_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "rnnwf_destroy": (C.c_int, [_P]),
    "rnnwf_last_error": (C.c_char_p, [_P]),
    "rnnwf_backend_name": (C.c_char_p, []),
    "rnnwf_abi_version": (C.c_int, []),
    "rnnwf_set_param": (C.c_int, [_P, C.c_char_p, _P, _I64, _I32]),
    "rnnwf_get_param": (C.c_int, [_P, C.c_char_p, _P, _I64, _I32]),
    "rnnwf_commit_params": (C.c_int, [_P]),
    "rnnwf_init_params": (C.c_int, [_P, C.c_uint64]),
    "rnnwf_num_params": (_I64, [_P]),
    "rnnwf_sample": (C.c_int, [_P, _I64, _U64, _U64, _I64, _I32P, _F64P]),
    "rnnwf_log_prob": (C.c_int, [_P, _I32P, _I64, _F64P]),
    "rnnwf_log_amp": (C.c_int, [_P, _I32P, _I64, _F32P]),
    "rnnwf_tfim_eloc": (C.c_int, [_P, _I32P, _I64, _F64P, _F64, _F64P, _F64P]),
    "rnnwf_tfim2d_eloc": (C.c_int, [_P, _I32P, _I64, _F64P, _F64, _F64P, _F64P]),
    "rnnwf_j1j2_eloc": (C.c_int, [_P, _I32P, _I64, _F64P, _F64P, _F64P, _I32, _I32, _F32P, C.POINTER(_I64)]),
    "rnnwf_vmc_step": (C.c_int, [_P, _I64, _U64, _U64, _I64, _F64P, _I64, _I32P, _P, _F64P]),
    "rnnwf_vmc_gradient": (C.c_int, [_P, _F64, _F64, _F64]),
    "rnnwf_load_batch": (C.c_int, [_P, _I32P, _I64, _P]),
    "rnnwf_get_grad": (C.c_int, [_P, C.c_char_p, _P, _I64, _I32]),
    "rnnwf_set_params_flat": (C.c_int, [_P, _F64P, _I64]),
    "rnnwf_get_grads_flat": (C.c_int, [_P, _F64P, _I64]),
    "rnnwf_param_name": (C.c_char_p, [_P, _I32, C.POINTER(_I64)]),
    "rnnwf_allreduce_grads": (C.c_int, [_P]),
    "rnnwf_comm_unique_id": (C.c_int, [_P]),
    "rnnwf_comm_init": (C.c_int, [_P, _P, _I32, _I32]),
    "rnnwf_allreduce_moments": (C.c_int, [_P, _F64P, _I32]),
    "rnnwf_allreduce_f64": (C.c_int, [_P, _F64P, _I64]),
    "rnnwf_comm_info": (C.c_int, [_P, _P, _P, _P]),
    "rnnwf_comm_reduce_in_step": (C.c_int, [_P, _I32]),
    "rnnwf_comm_destroy": (C.c_int, [_P]),
    "rnnwf_device_training_supported": (C.c_int, [_P]),
    "rnnwf_adam_step": (C.c_int, [_P, _F64, _F64, _F64, _F64]),
    "rnnwf_train_steps": (C.c_int, [_P, _I32, _I64, C.c_uint64, C.c_uint64, _I64, _F64P, _I64, _F64P, _F64, _F64, _F64, _F64P]),
    "rnnwf_adam_get_state": (C.c_int, [_P, _F64P, _F64P, _I64, C.POINTER(_I64)]),
    "rnnwf_adam_set_state": (C.c_int, [_P, _F64P, _F64P, _I64, _I64]),
    "rnnwf_timing_enable": (C.c_int, [_P, _I32]),
    "rnnwf_timing_reset": (C.c_int, [_P]),
    "rnnwf_timing_get": (C.c_int, [_P, _I32, _F64P, C.POINTER(_I64), _F64P]),
    "rnnwf_engine_name": (C.c_char_p, [_P]),
    "rnnwf_synchronize": (C.c_int, [_P]),
    "rnnwf_device_info": (C.c_int, [_P, C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I64), C.c_char_p]),
}

_lib = None


def load_library(path=None):
    """Load the shared library and attach the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(
            "librnnwf_hip.so not found at %s - build it with `python -m rnnwavefunctions_amd.build` "
            "(hipcc, --offload-arch=gfx950).  There is no CPU fallback." % p)
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)        # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.rnnwf_abi_version() != ABI_VERSION:
        raise ImportError("librnnwf_hip.so ABI %d != binding ABI %d" % (lib.rnnwf_abi_version(), ABI_VERSION))
    if path is None:
        _lib = lib
    return lib


class RnnwfError(RuntimeError):
    pass


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_I32P)


def _f64(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_F64P)


class NativeWavefunction:
    """One rnnwf_handle: a wave function resident on one MI355X."""

    def __init__(self, model, nx, ny=1, units=(10,), device=0):
        self.lib = load_library()
        cfg = Config()
        cfg.abi_version = ABI_VERSION
        cfg.model = model
        cfg.nx, cfg.ny = int(nx), int(ny)
        cfg.num_layers = len(units)
        if len(units) > MAX_LAYERS:
            raise ValueError("at most %d layers" % MAX_LAYERS)
        for i, u in enumerate(units):
            cfg.units[i] = int(u)
        cfg.device = int(device)
        h = _P()
        rc = self.lib.rnnwf_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            msg = self.lib.rnnwf_last_error(None).decode()
            raise (ValueError if rc == -1 else RnnwfError)(msg)
        self.h = h
        self.model = model
        self.N = int(nx) * int(ny)
        self.nx, self.ny = int(nx), int(ny)

    # -- plumbing ---------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            msg = self.lib.rnnwf_last_error(self.h).decode()
            raise (ValueError if rc == -1 else RnnwfError)(msg or "rnnwf error %d" % rc)

    def close(self):
        if getattr(self, "h", None):
            self.lib.rnnwf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- parameters -------------------------------------------------------------------------------
    def _layout(self):
        """[(tf name without scope, element count)] in the library's order (rnnwf_param_name)."""
        lay = getattr(self, "_param_layout", None)
        if lay is None:
            lay, i, cnt = [], 0, _I64()
            while True:
                nm = self.lib.rnnwf_param_name(self.h, i, C.byref(cnt))
                if nm is None:
                    break
                lay.append((nm.decode(), int(cnt.value)))
                i += 1
            self._param_layout = lay
        return lay

    def set_params(self, params, scope=None):
        """params: {tf_name: ndarray}; names may carry the ``<scope>/`` prefix.  A complete set goes over in one call
        (rnnwf_set_params_flat: what a training loop does every iteration), a partial one tensor by tensor."""
        lay = self._layout()
        pre = scope + "/" if scope else ""
        if len(params) == len(lay) and all((pre + nm in params or nm in params) and
                                           np.size(params[pre + nm] if pre + nm in params else params[nm]) == cnt for nm, cnt in lay):
            flat = np.empty(sum(cnt for _, cnt in lay), dtype=np.float64)
            off = 0
            for nm, cnt in lay:
                v = params[pre + nm] if pre + nm in params else params[nm]
                flat[off:off + cnt] = np.asarray(v, dtype=np.float64).ravel()
                off += cnt
            self._check(self.lib.rnnwf_set_params_flat(self.h, flat.ctypes.data_as(_F64P), flat.size))
            return
        for name, v in params.items():
            if scope and name.startswith(scope + "/"):
                name = name[len(scope) + 1:]
            v = np.ascontiguousarray(v)
            if v.dtype == np.float32:
                dt = F32
            else:
                v = np.ascontiguousarray(v, dtype=np.float64)
                dt = F64
            self._check(self.lib.rnnwf_set_param(self.h, name.encode(), v.ctypes.data_as(_P), v.size, dt))
        self._check(self.lib.rnnwf_commit_params(self.h))

    def init_params(self, seed):
        """Glorot-uniform kernels, gate bias 1, other biases 0 - the values of params.init_gru_params /
        init_mdrnn_params for the same seed, generated inside the library (C callers have no NumPy)."""
        self._check(self.lib.rnnwf_init_params(self.h, int(seed)))

    def get_param(self, name, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        dt = F32 if out.dtype == np.float32 else F64
        self._check(self.lib.rnnwf_get_param(self.h, name.encode(), out.ctypes.data_as(_P), out.size, dt))
        return out

    def num_params(self):
        return int(self.lib.rnnwf_num_params(self.h))

    # -- wave function ----------------------------------------------------------------------------
    def _sample_shape(self, n):
        return (n, self.nx, self.ny) if self.model == MODEL_MDRNN2D else (n, self.N)

    def sample(self, numsamples, seed, step=0, sample_offset=0, return_log=False):
        out = np.empty(self._sample_shape(numsamples), dtype=np.int32)
        lg = np.empty(numsamples, dtype=np.float64) if return_log else None
        self._check(self.lib.rnnwf_sample(self.h, numsamples, seed, step, sample_offset, out.ctypes.data_as(_I32P),
                                          lg.ctypes.data_as(_F64P) if return_log else None))
        return (out, lg) if return_log else out

    def log_prob(self, samples):
        s, sp = _i32(samples)
        B = s.shape[0] if s.ndim > 1 else 0
        if s.ndim < 2 or int(np.prod(s.shape[1:])) != self.N:
            raise ValueError("samples must have shape (B, %d) (or (B, Nx, Ny)), got %r" % (self.N, s.shape))
        out = np.empty(B, dtype=np.float64)
        self._check(self.lib.rnnwf_log_prob(self.h, sp, B, out.ctypes.data_as(_F64P)))
        return out

    def log_amp(self, samples):
        s, sp = _i32(samples)
        if s.ndim != 2 or s.shape[1] != self.N:
            raise ValueError("samples must have shape (B, %d), got %r" % (self.N, s.shape))
        out = np.empty((s.shape[0], 2), dtype=np.float32)
        self._check(self.lib.rnnwf_log_amp(self.h, sp, s.shape[0], out.ctypes.data_as(_F32P)))
        return out.view(np.complex64)[:, 0]

    # -- estimators -------------------------------------------------------------------------------
    def tfim_eloc(self, samples, Jz, Bx, log_probs=None):
        s, sp = _i32(samples)
        ns = s.shape[0]
        if s.ndim < 2 or ns < 1 or int(np.prod(s.shape[1:])) != self.N:
            raise ValueError("samples must be non-empty with %d sites per row, got shape %r" % (self.N, s.shape))
        jz, jzp = _f64(Jz)
        if jz.size != self.N:
            raise ValueError("Jz must have %d entries, got %d" % (self.N, jz.size))
        e = np.empty(ns, dtype=np.float64)
        lpp = None
        if log_probs is not None:
            if log_probs.dtype != np.float64 or not log_probs.flags.c_contiguous or log_probs.size < (self.N + 1) * ns:
                raise ValueError("log_probs must be a contiguous float64 array of (N+1)*numsamples entries")
            lpp = log_probs.ctypes.data_as(_F64P)
        two_d = self.model in (MODEL_GRU1D_F64, MODEL_MDRNN2D)
        fn = self.lib.rnnwf_tfim2d_eloc if two_d else self.lib.rnnwf_tfim_eloc
        self._check(fn(self.h, sp, ns, jzp, float(Bx), e.ctypes.data_as(_F64P), lpp))
        return e

    def j1j2_eloc(self, samples, J1, J2, Bz, periodic=False, marshall=False):
        s, sp = _i32(samples)
        if s.ndim != 2 or s.shape[1] != self.N:
            raise ValueError("samples must have shape (ns, %d), got %r" % (self.N, s.shape))
        j1, j1p = _f64(J1)
        j2, j2p = _f64(J2)
        bz, bzp = _f64(Bz)
        if not (j1.size == j2.size == bz.size == self.N):
            raise ValueError("J1, J2, Bz must have %d entries" % self.N)
        e = np.empty((s.shape[0], 2), dtype=np.float32)
        ncon = _I64(0)
        self._check(self.lib.rnnwf_j1j2_eloc(self.h, sp, s.shape[0], j1p, j2p, bzp, int(bool(periodic)),
                                             int(bool(marshall)), e.ctypes.data_as(_F32P), C.byref(ncon)))
        return e.view(np.complex64)[:, 0], int(ncon.value)

    def vmc_step(self, numsamples, seed, step, couplings, sample_offset=0, want_samples=False, want_eloc=False):
        """Fused sample + local energies + moments.  Returns dict(moments=(4,), samples=?, eloc=?)."""
        c, cp = _f64(couplings)
        mom = np.zeros(4, dtype=np.float64)
        smp = np.empty(self._sample_shape(numsamples), dtype=np.int32) if want_samples else None
        if want_eloc:
            el = np.empty((numsamples, 2), dtype=np.float32) if self.model == MODEL_CRNN_U1 else \
                np.empty(numsamples, dtype=np.float64)
        else:
            el = None
        self._check(self.lib.rnnwf_vmc_step(self.h, numsamples, seed, step, sample_offset, cp, c.size,
                                            smp.ctypes.data_as(_I32P) if want_samples else None,
                                            el.ctypes.data_as(_P) if want_eloc else None, mom.ctypes.data_as(_F64P)))
        out = {"moments": mom}
        if want_samples:
            out["samples"] = smp
        if want_eloc:
            out["eloc"] = el.view(np.complex64)[:, 0] if self.model == MODEL_CRNN_U1 else el
        return out

    def load_batch(self, samples, eloc):
        """Make a caller-supplied batch (samples + their local energies) the one vmc_gradient works on."""
        s = np.ascontiguousarray(samples, dtype=np.int32)
        ns = s.shape[0]
        if self.model == MODEL_CRNN_U1:
            e = np.ascontiguousarray(np.asarray(eloc, dtype=np.complex64))
        else:
            e = np.ascontiguousarray(np.asarray(eloc, dtype=np.float64))
        if e.shape != (ns,):
            raise ValueError("load_batch: %d samples but local energies of shape %s" % (ns, e.shape))
        self._check(self.lib.rnnwf_load_batch(self.h, s.ctypes.data_as(_I32P), ns, e.ctypes.data_as(_P)))

    # -- gradient of the VMC cost -----------------------------------------------------------------
    def vmc_gradient(self, mean_energy, norm, shapes, allreduce=False):
        """Gradient of the reference's cost on the batch of the last vmc_step (mean_energy may be complex for
        the complex RNN).  shapes: {tf_name (without scope): shape}; returns {tf_name: float64 ndarray}."""
        me = complex(mean_energy)
        self._check(self.lib.rnnwf_vmc_gradient(self.h, me.real, me.imag, float(norm)))
        if allreduce:
            self._check(self.lib.rnnwf_allreduce_grads(self.h))
        lay = self._layout()
        if len(shapes) == len(lay) and all(nm in shapes and int(np.prod(shapes[nm], dtype=np.int64)) == cnt for nm, cnt in lay):
            flat = np.empty(sum(cnt for _, cnt in lay), dtype=np.float64)          # every tensor in one call
            self._check(self.lib.rnnwf_get_grads_flat(self.h, flat.ctypes.data_as(_F64P), flat.size))
            out, off = {}, 0
            for nm, cnt in lay:
                out[nm] = flat[off:off + cnt].reshape(shapes[nm])
                off += cnt
            return {name: out[name] for name in shapes}
        out = {}
        for name, shape in shapes.items():
            g = np.empty(shape, dtype=np.float64)
            self._check(self.lib.rnnwf_get_grad(self.h, name.encode(), g.ctypes.data_as(_P), g.size, F64))
            out[name] = g
        return out

    # -- device-resident training iteration (rnnwf_train_steps; single-layer float32 GRU models) ------------------
    def device_training_supported(self):
        return bool(self.lib.rnnwf_device_training_supported(self.h))

    def train_steps(self, numsamples, seed, step0, couplings, learning_rates, beta1=0.9, beta2=0.999, epsilon=1e-8, sample_offset=0):
        """len(learning_rates) whole iterations (sample, local energies, gradient, Adam update, re-pack of the weight images) on
        the device with one host synchronisation; returns the (K, 4) moments of the K batches."""
        c, cp = _f64(couplings)
        lr, lrp = _f64(np.atleast_1d(learning_rates))
        mom = np.empty((lr.size, 4), dtype=np.float64)
        self._check(self.lib.rnnwf_train_steps(self.h, lr.size, numsamples, seed, step0, sample_offset, cp, c.size, lrp,
                                               float(beta1), float(beta2), float(epsilon), mom.ctypes.data_as(_F64P)))
        return mom

    def adam_step(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        """One Adam update on the device from the gradient of the last vmc_gradient, then the images' re-pack."""
        self._check(self.lib.rnnwf_adam_step(self.h, float(learning_rate), float(beta1), float(beta2), float(epsilon)))

    def adam_get_state(self):
        """(m_flat, v_flat, t): Adam's moments in the flat order of _layout() and the number of updates applied."""
        n = self.num_params()
        m, v, t = np.empty(n), np.empty(n), _I64(0)
        self._check(self.lib.rnnwf_adam_get_state(self.h, m.ctypes.data_as(_F64P), v.ctypes.data_as(_F64P), n, C.byref(t)))
        return m, v, int(t.value)

    def adam_set_state(self, m_flat, v_flat, t):
        if m_flat is None:
            self._check(self.lib.rnnwf_adam_set_state(self.h, None, None, self.num_params(), int(t)))
            return
        m, mp = _f64(m_flat)
        v, vp = _f64(v_flat)
        self._check(self.lib.rnnwf_adam_set_state(self.h, mp, vp, m.size, int(t)))

    def get_params_dict(self, like, scope=None):
        """{name: array} with the names, shapes and dtypes of `like`, read back from the library (after device-resident updates)."""
        pre = scope + "/" if scope else ""
        out = type(like)()
        for k, v in like.items():
            nm = k[len(pre):] if pre and k.startswith(pre) else k
            out[k] = self.get_param(nm, v.shape, v.dtype)
        return out

    # -- multi-GPU --------------------------------------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        rc = self.lib.rnnwf_comm_unique_id(buf)
        if rc != 0:
            raise RnnwfError("rnnwf_comm_unique_id failed (%d)" % rc)
        return buf.raw

    def comm_init(self, unique_id, rank, nranks):
        self._check(self.lib.rnnwf_comm_init(self.h, C.c_char_p(unique_id), rank, nranks))

    def comm_reduce_in_step(self, on=True):
        """vmc_step returns the moments summed over all ranks (one in-stream RCCL all-reduce, one host sync per step)."""
        self._check(self.lib.rnnwf_comm_reduce_in_step(self.h, int(on)))

    def comm_info(self):
        """What the RCCL communicator itself reports (ncclCommCount / ncclCommUserRank) plus the handle's device."""
        n, r, d = _I32(0), _I32(0), _I32(0)
        self._check(self.lib.rnnwf_comm_info(self.h, C.byref(n), C.byref(r), C.byref(d)))
        return {"nranks": n.value, "rank": r.value, "device": d.value}

    def allreduce_moments(self, moments):
        m, mp = _f64(np.array(moments, dtype=np.float64))
        self._check(self.lib.rnnwf_allreduce_moments(self.h, mp, m.size))
        return m

    def allreduce_f64(self, a):
        """Sum over the ranks of a float64 array of any shape (rnnwf_allreduce_f64; identity without a communicator)."""
        m = np.array(a, dtype=np.float64, order="C")
        flat, fp = _f64(m.reshape(-1))
        self._check(self.lib.rnnwf_allreduce_f64(self.h, fp, flat.size))
        return flat.reshape(m.shape)

    def allreduce_grads(self, grads):
        """{name: array} summed over the ranks in ONE all-reduce (names in sorted order on every rank)."""
        names = sorted(grads)
        flat = self.allreduce_f64(np.concatenate([np.asarray(grads[k], dtype=np.float64).ravel() for k in names]))
        out, off = {}, 0
        for k in names:
            n = int(np.asarray(grads[k]).size)
            out[k] = flat[off:off + n].reshape(np.shape(grads[k]))
            off += n
        return out

    # -- measurement ------------------------------------------------------------------------------
    def timing_enable(self, on=True):
        self._check(self.lib.rnnwf_timing_enable(self.h, int(on)))

    def timing_reset(self):
        self._check(self.lib.rnnwf_timing_reset(self.h))

    def timing_get(self, kernel_id):
        ms, n = _F64(0), _I64(0)
        work = (C.c_double * 2)()
        self._check(self.lib.rnnwf_timing_get(self.h, kernel_id, C.byref(ms), C.byref(n), work))
        return {"total_ms": ms.value, "launches": n.value, "cell_evals": work[0], "mfma_flops": work[1]}

    def engine_name(self):
        return self.lib.rnnwf_engine_name(self.h).decode()

    def synchronize(self):
        self._check(self.lib.rnnwf_synchronize(self.h))

    def device_info(self):
        cu, mhz, hbm = _I32(0), _I32(0), _I64(0)
        name = C.create_string_buffer(64)
        self._check(self.lib.rnnwf_device_info(self.h, C.byref(cu), C.byref(mhz), C.byref(hbm), name))
        return {"cu_count": cu.value, "clock_mhz": mhz.value, "hbm_bytes": hbm.value, "name": name.value.decode()}
