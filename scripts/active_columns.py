"""GPU: Jacobi sweeps and active (non-deflated) columns per order along the tutorial schedule, for the projection path selected
by the environment (MTIP_PROJ_REAL=0: general complex kernels).   usage: python scripts/active_columns.py [cfg=3]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
np.seterr(all='ignore')
from xframe_amd.fxs import synthetic as S, hostsetup as hs      # noqa: E402
from xframe_amd.fxs.engine import Engine                        # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, L = S._SIZES[cfg]
ed = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, max_q=S.data_cutoff(N))
data, _ = S.make_invariants(ed, N, L)
ed.close()
e = Engine(S.config_overrides(cfg), data, n_batch=1)
e.set_density(0, hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000),
                                 e.rsetup.integrated_intensity, e.int_wr, e.int_wt))
e.init_state()
e.profile(True)
def report(tag):
    sw, ac = e.jacobi_sweeps()[0], e.jacobi_active_columns()[0]
    print(tag, 'sweeps', [int(x) for x in sw[2::2]], 'active columns', [int(x) for x in ac[2::2]])
for i in range(3):
    e.run('HIO', True, np.full(20, 0.45)); report('after %3d HIO steps:' % (20 * (i + 1)))
e.shrinkwrap(20.0, 0.09, 6e-3) if hasattr(e, 'shrinkwrap') else None
for i in range(2):
    e.run('ER', True, np.full(20, 0.45)); report('after %3d ER steps: ' % (20 * (i + 1)))
ms, n = e.profile_get('polar')
print('polar: %.1f us per call over %d calls' % (1e3 * ms / n, n))
e.close()
