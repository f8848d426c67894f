"""GPU: the `extract` eigen-decompositions at a BASELINE size -- B_l (L+1 real symmetric Nq x Nq matrices of the synthetic
particle) through Engine.hermitian_eig: wall time per call, kernel time, against numpy.   usage: python scripts/bench_extract.py [cfg=3]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
np.seterr(all='ignore')
from xframe_amd.fxs import synthetic as S          # noqa: E402
from xframe_amd.fxs.engine import Engine           # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, L = S._SIZES[cfg]
e = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, max_q=S.data_cutoff(N))
rho = S.ball_density(e.rs, e.thetas, e.phis)
F = e.ft(rho)
Ilm = e.forward_l(F * F.conj())
B = np.stack([(np.asarray(Il) @ np.asarray(Il).conj().T).real / 4 for Il in Ilm])
B = (B + np.swapaxes(B, -1, -2)) / 2
e.hermitian_eig(B)
e.profile(True)
t0 = time.perf_counter()
reps = 5
for _ in range(reps):
    w, v = e.hermitian_eig(B)
t1 = time.perf_counter()
for fam in ('sym_eig', 'herm_eig'):
    ms, n = e.profile_get(fam)
    if n:
        print('%s kernel: %.3f ms per call (%d matrices of %d x %d)' % (fam, ms / n, len(B), N, N))
print('Engine.hermitian_eig wall: %.2f ms per call (incl. PCIe both ways and host sorting)' % (1e3 * (t1 - t0) / reps))
t0 = time.perf_counter(); wn, vn = np.linalg.eigh(B); t1 = time.perf_counter()
print('numpy.linalg.eigh: %.1f ms' % (1e3 * (t1 - t0)))
err = np.abs(w - wn[:, ::-1]).max(1) / np.abs(wn).max(1)
print('eigenvalues vs numpy, max |dw| / |w|_max per order: %.1e' % err.max())
e.close()
