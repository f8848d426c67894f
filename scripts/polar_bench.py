"""GPU micro-benchmark of the reciprocal projection (X_l GEMM + polar factor + V_l U_l) at a BASELINE config:
realistic I_lm (intensity of a few phasing steps of a seeded restart), hipEvent time per call, optional in-kernel phase
timers of the Newton polar-factor kernel.   usage: python scripts/polar_bench.py [cfg=3] [B=8] [reps=20]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
np.seterr(all='ignore')
from xframe_amd.fxs import synthetic as S, hostsetup as hs, _lib      # noqa: E402
from xframe_amd.fxs.engine import Engine                              # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
N, L = S._SIZES[cfg]
ed = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, max_q=S.data_cutoff(N))
data, _ = S.make_invariants(ed, N, L)
ed.close()
e = Engine(S.config_overrides(cfg), data, n_batch=B)
for b in range(B):
    e.set_density(b, hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000 + b),
                                     e.rsetup.integrated_intensity, e.int_wr, e.int_wt))
e.init_state()
e.run('HIO', True, np.full(5, 0.45))
F = np.stack([e.reciprocal_density(b) for b in range(B)])
Ilm = e.sht_forward(F, 1)
e.project_coefficients(Ilm)
timers = os.environ.get('POLAR_TIMERS', '0') == '1'
if timers:
    e.lib.mtip_debug_polar_timing(e.ctx, None)
e.profile(True)
for _ in range(reps):
    e.project_coefficients(Ilm)
ms, n = e.profile_get('proj')
print('variant %s polar %s: proj %.1f us per call (B = %d, %d calls)' % (os.environ.get('MTIP_POLAR_VARIANT', '0'), os.environ.get('MTIP_POLAR', 'jacobi'), 1e3 * ms / n, B, n))
print('iterations per order (restart 0):', list(e.jacobi_sweeps()[0]))
if not timers:
    sys.exit(0)
out = np.zeros((B, L + 1, 8, 4), np.int64)
e.lib.mtip_debug_polar_timing(e.ctx, _lib.ptr(out))
for l in (L, L - 2, 16, 8):
    t = out[0, l]
    print('l = %2d: per wave  produce %s | wait %s | consume %s | kernel %s  (k cycles)' % (
        l, (t[:, 0] // 1000).tolist(), (t[:, 1] // 1000).tolist(), (t[:, 2] // 1000).tolist(), (t[:, 3] // 1000).tolist()))
