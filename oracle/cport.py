"""ctypes access to oracle/c/rnnwf_oracle.c (the C restatement).  TEST INFRASTRUCTURE ONLY.

Built on demand with gcc into oracle/_build/ (git-ignored, but it travels to the GPU box with
the snapshot).  ``-march=native`` code built in one place may not run in another, so the library
is rebuilt whenever the host CPU flags recorded next to it differ.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "rnnwf_oracle.c")
OUT = os.path.join(HERE, "_build", "librnnwf_oracle.so")
STAMP = OUT + ".stamp"
# AddressSanitizer + UndefinedBehaviorSanitizer build of the same source (tests/test_host.py runs the golden cases through
# it in a child process with libasan preloaded; SURVEY.md 5: sanitizers on the CPU side only)
OUT_SAN = os.path.join(HERE, "_build", "librnnwf_oracle_san.so")
GRU = "multi_rnn_cell/cell_0/cudnn_compatible_gru_cell/"


def _host_tag():
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    with open(SRC, "rb") as f:
        return hashlib.sha1(flags.encode() + f.read()).hexdigest()


def build(force=False):
    tag = _host_tag()
    if not force and os.path.exists(OUT) and os.path.exists(STAMP) and open(STAMP).read() == tag:
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["gcc", "-O3", "-march=native", "-fopenmp", "-fPIC", "-std=c11", "-fno-math-errno",
           "-ffp-contract=off", "-shared", "-o", OUT, SRC, "-lm"]
    subprocess.run(cmd, check=True, capture_output=True)
    with open(STAMP, "w") as f:
        f.write(tag)
    return OUT


def build_sanitized():
    """gcc -fsanitize=address,undefined -fno-sanitize-recover: any out-of-bounds access, misaligned load, signed
    overflow or shift error in the C restatement aborts the process that loaded it."""
    if os.path.exists(OUT_SAN) and os.path.getmtime(OUT_SAN) > os.path.getmtime(SRC):
        return OUT_SAN
    os.makedirs(os.path.dirname(OUT_SAN), exist_ok=True)
    cmd = ["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fopenmp", "-fPIC", "-std=c11", "-ffp-contract=off", "-shared", "-o", OUT_SAN, SRC, "-lm"]
    subprocess.run(cmd, check=True, capture_output=True)
    return OUT_SAN


def sanitizer_runtime():
    """Path of libasan.so for LD_PRELOAD (an ASan-instrumented .so cannot be dlopen'ed into a plain interpreter)."""
    r = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True)
    path = r.stdout.strip()
    return path if r.returncode == 0 and os.path.isabs(path) and os.path.exists(path) else None


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_sanitized() if os.environ.get("RNNWF_ORACLE_SANITIZED") == "1" else build())
        _lib.rnnwf_oracle_max_threads.restype = C.c_int
    return _lib


def usable_cores():
    """Cores this process may actually use: min(affinity mask, cgroup CPU quota, OpenMP default)."""
    n = int(lib().rnnwf_oracle_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                q = int(parts[0])
                if q > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                        n = min(n, max(1, q // int(f.read())))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def max_threads():
    return usable_cores()


def _wargs(params, scope):
    def g(name):
        return np.ascontiguousarray(params[scope + "/" + name], dtype=np.float32)
    arrs = [g(GRU + "gates/kernel"), g(GRU + "gates/bias"), g(GRU + "candidate/input_projection/kernel"),
            g(GRU + "candidate/input_projection/bias"), g(GRU + "candidate/hidden_projection/kernel"),
            g(GRU + "candidate/hidden_projection/bias"), g("wf_dense/kernel"), g("wf_dense/bias")]
    H = arrs[4].shape[0]
    return H, arrs, [a.ctypes.data_as(C.c_void_p) for a in arrs]


def prnn_log_probability(params, samples, scope="RNNwavefunction", nthreads=0):
    s = np.ascontiguousarray(samples, dtype=np.int32)
    B, N = s.shape
    H, keep, w = _wargs(params, scope)
    out = np.empty(B, dtype=np.float64)
    lib().rnnwf_oracle_prnn_log_prob(C.c_int(H), C.c_int(N), *w, s.ctypes.data_as(C.c_void_p), C.c_int64(B),
                                     out.ctypes.data_as(C.c_void_p), C.c_int(nthreads))
    return out


def prnn_sample(params, N, u, scope="RNNwavefunction", nthreads=0):
    u = np.ascontiguousarray(u, dtype=np.float64)
    ns = u.shape[0]
    H, keep, w = _wargs(params, scope)
    s = np.empty((ns, N), dtype=np.int32)
    lp = np.empty(ns, dtype=np.float64)
    lib().rnnwf_oracle_prnn_sample(C.c_int(H), C.c_int(N), *w, u.ctypes.data_as(C.c_void_p), C.c_int64(ns),
                                   s.ctypes.data_as(C.c_void_p), lp.ctypes.data_as(C.c_void_p), C.c_int(nthreads))
    return s, lp


def ising_local_energies(params, Jz, Bx, samples, scope="RNNwavefunction", nthreads=0, return_log_probs=False):
    """Reference formulation (queue materialised, <= 25000-row chunks) entirely in C."""
    s = np.ascontiguousarray(samples, dtype=np.int32)
    ns, N = s.shape
    H, keep, w = _wargs(params, scope)
    jz = np.ascontiguousarray(Jz, dtype=np.float64)
    queue = np.empty((N + 1, ns, N), dtype=np.int32)
    lp = np.empty((N + 1) * ns, dtype=np.float64)
    e = np.empty(ns, dtype=np.float64)
    lib().rnnwf_oracle_tfim_local_energies(C.c_int(H), C.c_int(N), *w, jz.ctypes.data_as(C.c_void_p),
                                           C.c_double(float(Bx)), s.ctypes.data_as(C.c_void_p), C.c_int64(ns),
                                           queue.ctypes.data_as(C.c_void_p), lp.ctypes.data_as(C.c_void_p),
                                           e.ctypes.data_as(C.c_void_p), C.c_int(nthreads))
    return (e, lp) if return_log_probs else e
