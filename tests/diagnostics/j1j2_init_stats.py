"""E_0 and var Re(E_loc) of the untrained complex RNN at the notebook's size over 40 initial draws (CPU, the oracle): is the notebook's
step-0 line (2.3466, 0.0785) inside the distribution of OUR initial states?   python tests/diagnostics/j1j2_init_stats.py"""
import sys, numpy as np
sys.path.insert(0, __import__('os').getcwd())
from rnnwavefunctions_amd import params as P
from oracle import models as M, estimators as E
N=10; J1=np.ones(N); J2=0.2*np.ones(N); Bz=np.zeros(N)
res=[]
for seed in range(1,41):
    prm=P.init_gru_params([10], seed=seed, heads=("wf_dense_ampl","wf_dense_phase"))
    rng=np.random.RandomState(1000+seed)
    ns=2000
    s=M.crnn_sample(prm, N, rng.random_sample((ns,N)))
    s = s[0] if isinstance(s, tuple) else s
    el=E.j1j2_local_energies(J1,J2,Bz,s,lambda x: M.crnn_log_amplitude(prm,x))
    res.append((np.mean(el.real), np.var(el.real), np.var(el.imag)))
    print(seed, "E0 %.3f varRe %.4f varIm %.4f" % res[-1], flush=True)
r=np.array(res)
print("E0 mean %.3f sd %.3f ; varRe mean %.4f sd %.4f min %.4f max %.4f" % (r[:,0].mean(), r[:,0].std(), r[:,1].mean(), r[:,1].std(), r[:,1].min(), r[:,1].max()))
