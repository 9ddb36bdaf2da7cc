"""Build librnnwf_hip.so (hipcc, gfx950) in-tree: ``python -m rnnwavefunctions_amd.build``.

The shared library is the whole product below the Python facade; it links only against the
HIP runtime (RCCL is dlopen'ed at communicator creation).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "librnnwf_hip.so")
SOURCES = ["rnnwf_api.hip", "prnn.hip", "crnn.hip", "split.hip", "split_stream.hip", "mdrnn.hip", "grad.hip", "comm.hip", "train.hip", "grad_wide.hip"]
# no SLP packing of adjacent f32 adds / fmas into v_pk_*_f32 in the bf16x3 engine's translation unit: packed-f32 (and v_dot2)
# instructions stall behind bf16 MFMAs - their own wave's AND the SIMD partner's (measured: tools/microbench/issue_model,
# a VALU segment with packed ops beside an MFMA partner 5 170 vs 3 337 cycles).  The f32-input-MFMA kernels keep it.
# train.hip (device-side Adam and image re-pack) must round every operation on its own, as NumPy and the host packers do: no
# contraction into fused multiply-adds (HIP's __dmul_rn / __dadd_rn are plain operators and were fused under -ffp-contract=fast;
# found by tests/test_gpu_training.py: test_device_adam_step_and_checkpointed_state)
# prnn.hip / crnn.hip: the streamed-weight products of the 133..260-unit kernels (gru_core.h: mfma_streamed, 160 k-groups x 5 tiles) must
# unroll completely - their fragment ring is indexed by the loop counter.  Past clang's default size budget for `#pragma unroll` the loop
# stayed rolled, the ring went to scratch memory and the 260-unit flip pass ran at 587 ms per config-2-sized step instead of 95
# (profiles/r04_m_wide_widths.txt); the budget is raised for these two translation units.
WIDE_UNROLL = ["-mllvm", "-pragma-unroll-threshold=400000"]
PER_SOURCE_FLAGS = {"split.hip": ["-fno-slp-vectorize"], "split_stream.hip": ["-fno-slp-vectorize"], "train.hip": ["-ffp-contract=off"],
                    "prnn.hip": WIDE_UNROLL, "crnn.hip": WIDE_UNROLL, "grad_wide.hip": WIDE_UNROLL}
# the 100-unit bf16x3 kernel keeps its 160 accumulator registers in AGPRs (a wave addresses 256 VGPRs + 256 AGPRs; its
# other live values need ~210 VGPRs): no -amdgpu-mfma-vgpr-form for its translation unit
AGPR_FORM_SOURCES = {"split_stream.hip", "grad_wide.hip"}     # grad_wide.hip: see its header (a compiler crash in the VGPR form)
MFMA_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result",
         "-ffp-contract=fast"]
# MFMA_VGPR_FORM (all sources but AGPR_FORM_SOURCES): keep MFMA accumulators in VGPRs: no v_accvgpr_read/write around the
# gate arithmetic and 4 waves/SIMD at num_units=50 (the option exists for every target of this clang, so the host pass
# accepts it too)


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "rnnwf.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, extra, objdir=None):
    obj = os.path.join(objdir or OBJDIR, os.path.splitext(src)[0] + ".o")
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), _newest_header()) and (not extra or objdir):
        return obj, ""
    cmd = [HIPCC] + FLAGS + ([] if src in AGPR_FORM_SOURCES else MFMA_VGPR_FORM) + PER_SOURCE_FLAGS.get(src, []) + extra + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, " ".join(cmd), r.stderr[-6000:]))
    return obj, r.stderr


def build(verbose=False, extra_flags=(), jobs=None, variant=None):
    """variant=None: the product library.  variant="diag": librnnwf_hip_diag.so, compiled with -DRNNWF_DIAGNOSTICS
    (timing-only ablation switches and in-kernel cycle stamps; used by tools/ only, never loaded by the package)."""
    extra = list(extra_flags)
    objdir, lib = OBJDIR, LIB
    if variant == "diag":
        extra.append("-DRNNWF_DIAGNOSTICS")
        objdir, lib = os.path.join(LIBDIR, "obj_diag"), os.path.join(LIBDIR, "librnnwf_hip_diag.so")
    elif variant:                      # ad-hoc A/B builds of tools/: extra_flags into lib/librnnwf_hip_<variant>.so
        objdir, lib = os.path.join(LIBDIR, "obj_" + variant), os.path.join(LIBDIR, "librnnwf_hip_%s.so" % variant)
    os.makedirs(objdir, exist_ok=True)
    with ThreadPoolExecutor(max_workers=jobs or min(len(SOURCES), os.cpu_count() or 4)) as ex:
        results = list(ex.map(lambda s: _compile(s, extra, objdir), SOURCES))
    objs = [o for o, _ in results]
    if verbose:
        for _, log in results:
            if log.strip():
                print(log, file=sys.stderr)
    if (not os.path.exists(lib)) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
    return lib


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv, variant="diag" if "--diag" in sys.argv else None))
