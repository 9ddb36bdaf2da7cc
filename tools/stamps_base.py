#!/usr/bin/env python3
"""In-kernel cycle stamps of the bf16 cooperative base pass (diagnostics build only: python -m rnnwavefunctions_amd.build --diag).

    python tools/stamps_base.py [cfg2|cfg1] [steps]

Per wave role (gate waves / the remainder wave that also runs head + draw) the median over waves of the cycles spent per launch in:
products (operand reads + MFMAs), waiting at barrier B, gates + state / checkpoint / operand writes, waiting at barrier A, site()."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RNNWF_STAMPS_BASE"] = "1"
from rnnwavefunctions_amd import _lib   # noqa: E402

_lib._lib = _lib.load_library(os.path.join(ROOT, "rnnwavefunctions_amd", "lib", "librnnwf_hip_diag.so"))
assert _lib._lib.rnnwf_backend_name() == b"hip-gfx950-diagnostics"
import bench   # noqa: E402

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
wf, prm, couplings = bench.make_wavefunction(wl, device=0)
for it in range(steps):
    wf.vmc_step(wl["ns"], seed=111, step=it, couplings=couplings)
