import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rnnwavefunctions_amd import _lib, params as P
if len(sys.argv) > 1:
    _lib._lib = _lib.load_library(os.path.join(ROOT, "rnnwavefunctions_amd", "lib", "librnnwf_hip_%s.so" % sys.argv[1]))
def trained_like(H, seed, heads):
    return P.randomize_biases(P.scale_kernels(P.init_gru_params([H], seed=seed, heads=heads), 2.0), seed + 1)
H, N = 20, 12
prm = trained_like(H, 5, ("wf_dense_ampl", "wf_dense_phase"))
w1 = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,))
w1.set_params(prm, scope="RNNwavefunction")
J = (np.ones(N), 0.5 * np.ones(N), np.zeros(N))
sc = w1.sample(3000, seed=2, step=0)
es = [w1.j1j2_eloc(sc, *J)[0] for _ in range(3)]
print(sys.argv[1:], "random batch: run-to-run max", np.abs(es[0] - es[1]).max(), np.abs(es[0] - es[2]).max())
# identical configurations: every sample must give the same E_loc within one run
one = np.tile(sc[:1], (64, 1))
e = w1.j1j2_eloc(one, *J)[0]
print("  64 copies of one configuration: spread within the run", np.abs(e - e[0]).max(), " distinct values", len(set(e.tolist())))
e_b = w1.j1j2_eloc(one, *J)[0]
print("  second run spread", np.abs(e_b - e_b[0]).max(), " first-vs-second", np.abs(e - e_b).max())
os.environ["RNNWF_ENGINE"] = "f32"
w3 = _lib.NativeWavefunction(_lib.MODEL_CRNN_U1, N, 1, (H,)); del os.environ["RNNWF_ENGINE"]
w3.set_params(prm, scope="RNNwavefunction")
e3 = w3.j1j2_eloc(one, *J)[0]
print("  f32 engine: spread", np.abs(e3 - e3[0]).max(), " bf16x3 - f32:", np.abs(e - e3).max(), "values", e[:3], e3[:1])
