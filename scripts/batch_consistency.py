"""Diagnostic (GPU): the same initial density in every slot of batches of different sizes must give the same error
trajectory in every slot and for every batch size (restarts never interact; kernels pick their tiling by batch size)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
np.seterr(all='ignore')
from xframe_amd.fxs import synthetic as S, hostsetup as hs
from xframe_amd.fxs.engine import Engine

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, L = S._SIZES[cfg]
ed = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, max_q=S.data_cutoff(N))
data, _ = S.make_invariants(ed, N, L)
ed.close()
ref = None
for B in (2, 8, 9, 11, 12, 16):
    e = Engine(S.config_overrides(cfg), data, n_batch=B)
    rho0 = hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000), e.rsetup.integrated_intensity, e.int_wr, e.int_wt)
    for b in range(B):
        e.set_density(b, rho0)
    e.init_state()
    err, _ = e.run('HIO', True, np.full(6, 0.45))
    spread = np.abs(err / err[:, :1] - 1).max()
    if ref is None:
        ref = err[:, 0].copy()
    print('B = %2d: max rel spread between slots %.2e, vs B = 2 %.2e   errors[:3] %s' % (B, spread, np.abs(err[:, 0] / ref - 1).max(), err[:3, 0]))
    e.close()
