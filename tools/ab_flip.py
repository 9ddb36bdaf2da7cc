#!/usr/bin/env python3
"""A/B timing of the flip / swap pass for an ad-hoc library variant built by
    python -c "from rnnwavefunctions_amd import build; build.build(variant='x', extra_flags=['-DRNNWF_AB_...'])"
usage: python tools/ab_flip.py <variant|product> [cfg2|cfg3] [steps]   -> average kernel ms of timing slot 1 (flip / swap pass)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnnwavefunctions_amd import _lib   # noqa: E402

variant = sys.argv[1]
if variant != "product":
    _lib._lib = _lib.load_library(os.path.join(ROOT, "rnnwavefunctions_amd", "lib", "librnnwf_hip_%s.so" % variant))
import bench   # noqa: E402

wl = dict(bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "cfg2"])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
wf, prm, couplings = bench.make_wavefunction(wl, device=0)
for it in range(5):
    wf.vmc_step(wl["ns"], seed=111, step=it, couplings=couplings)
wf.timing_enable(True)
for it in range(steps):
    m = wf.vmc_step(wl["ns"], seed=111, step=5 + it, couplings=couplings)["moments"]
k = wf.timing_get(1)
print("%-10s %s flip/swap pass avg %.4f ms over %d launches   <E> %.6f" % (variant, sys.argv[2] if len(sys.argv) > 2 else "cfg2",
                                                                          k["total_ms"] / max(k["launches"], 1), k["launches"], m[0] / m[2]))
