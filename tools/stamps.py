#!/usr/bin/env python3
"""In-kernel cycle stamps of the ping-pong flip kernel (diagnostics build only: python -m rnnwavefunctions_amd.build --diag).

    python tools/stamps.py [cfg2|cfg3] [steps]

Loads lib/librnnwf_hip_diag.so (never the product library), runs a few VMC steps of the workload with RNNWF_STAMPS=1 and
prints, per flip launch, the median / min / max over waves of the cycles spent in the MFMA segment, at the barrier behind
it, in the VALU segment, at the barrier behind that, in tile switches, in total, and the 100 MHz real-time ticks (so that
clock = total_cycles / ticks * 100 MHz).  Stamps cost ~10 % themselves; the numbers rank segments, they are not timings."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("RNNWF_ABLATE") is None:
    os.environ["RNNWF_STAMPS"] = "1"          # with RNNWF_ABLATE=bits: timing only (1 no MFMA segment, 2 no gates, 4 no head, 8 no split)
from rnnwavefunctions_amd import _lib   # noqa: E402

diag = os.path.join(ROOT, "rnnwavefunctions_amd", "lib", "librnnwf_hip_diag.so")
_lib._lib = _lib.load_library(diag)
assert _lib._lib.rnnwf_backend_name() == b"hip-gfx950-diagnostics"
import bench   # noqa: E402

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wf, prm, couplings = bench.make_wavefunction(wl, device=0)
wf.timing_enable(True)
for it in range(steps):
    m = wf.vmc_step(wl["ns"], seed=111, step=it, couplings=couplings)["moments"]
k = wf.timing_get(1)
print("ablate", os.environ.get("RNNWF_ABLATE"), "flip kernel avg ms", k["total_ms"] / max(k["launches"], 1), "mean_E", m[0] / m[2])
