"""Philox4x32-10 counter-based RNG, NumPy restatement.  TEST INFRASTRUCTURE ONLY.

The reference draws its spins with ``tf.multinomial`` / ``tf.random.categorical``
(1DTFIM/RNNwavefunction.py:68, J1J2/ComplexRNNwavefunction.py:95), i.e. with
TensorFlow's Philox stream, which is not reproducible outside TensorFlow
("parity unpinned").  The build therefore defines its own stream,
identical in this oracle and in the HIP kernels (csrc/philox.h):

    key     = (seed_lo32, seed_hi32)
    counter = (g_lo32, g_hi32, site // 4, step_lo32)
    u       = (out[site % 4] >> 8) * 2**-24            in [0, 1)

with g the GLOBAL sample index, so the union of samples drawn by G shards is
independent of G.
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """All arguments broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def uniforms(seed, step, sample_offset, numsamples, nsites):
    """(numsamples, nsites) float64 array of the uniforms the sampler consumes.

    Every value is a multiple of 2**-24, hence exact in float32 as well.
    """
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    g = np.arange(sample_offset, sample_offset + numsamples, dtype=np.uint64)[:, None]
    blk = np.arange((nsites + 3) // 4, dtype=np.uint32)[None, :]
    out = philox4x32_10((g & _MASK).astype(np.uint32), (g >> np.uint64(32)).astype(np.uint32),
                        blk, np.uint32(int(step) & 0xFFFFFFFF),
                        seed & 0xFFFFFFFF, seed >> 32)
    words = np.stack(out, axis=-1).reshape(numsamples, -1)[:, :nsites]
    return (words >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
