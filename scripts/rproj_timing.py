"""GPU micro-benchmark of the real-arithmetic projection kernel k_rproj at a BASELINE config: realistic I_lm (intensity after a
few phasing steps of seeded restarts), hipEvent time per call and the in-kernel phase timers (mtip_debug_polar_timing).
usage: python scripts/rproj_timing.py [cfg=3] [B=3] [reps=20] [n_hio=5]"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
np.seterr(all='ignore')
from xframe_amd.fxs import synthetic as S, hostsetup as hs, _lib      # noqa: E402
from xframe_amd.fxs.engine import Engine                              # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n_hio = int(sys.argv[4]) if len(sys.argv) > 4 else 5
N, L = S._SIZES[cfg]
ed = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, max_q=S.data_cutoff(N))
data, _ = S.make_invariants(ed, N, L)
ed.close()
e = Engine(S.config_overrides(cfg), data, n_batch=B)
for b in range(B):
    e.set_density(b, hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000 + b),
                                     e.rsetup.integrated_intensity, e.int_wr, e.int_wt))
e.init_state()
e.run('HIO', True, np.full(n_hio, 0.45))
F = np.stack([e.reciprocal_density(b) for b in range(B)])
Ilm = e.sht_forward(F, 1)
real = os.environ.get('RPROJ_COMPLEX', '0') != '1'
e.project_coefficients(Ilm, real_intensity=real)
e.lib.mtip_debug_polar_timing(e.ctx, None)
e.profile(True)
for _ in range(reps):
    e.project_coefficients(Ilm, real_intensity=real)
ms, n = e.profile_get('proj')
print('%s projection: %.1f us per call (B = %d, %d calls, warm start on the same input)' % ('real' if real else 'complex', 1e3 * ms / n, B, n))
print('sweeps per order (restart 0):', list(e.jacobi_sweeps()[0]))
out = np.zeros((B, L + 1, 40), np.int64)      # MTIP_POLAR_TIMING_SLOTS
e.lib.mtip_debug_polar_timing(e.ctx, _lib.ptr(out))
t_first = out[:, :, 6][out[:, :, 6] > 0].min() if (out[:, :, 6] > 0).any() else 0
print('order: cycles of phase A (X~ product) | W (warm start) | J (Jacobi) | U | E (apply);  rounds; cycles per round; start / end (k cycles after the first workgroup); hw id')
for b in range(min(B, 2)):
    for l in range(L, 0, -1):
        t = out[b, l]
        if t[5] == 0 and t[2] == 0:
            continue
        print('b %d l %2d: A %6d  W %6d  J %8d  U %6d  E %6d   rounds %4d  %5.0f cyc/round   start %6d end %6d   hw 0x%x' % (
            b, l, t[0], t[1], t[2], t[3], t[4], t[5], t[2] / max(t[5], 1), (t[6] - t_first) // 1000, (t[7] - t_first) // 1000, t[8]))
for b in range(min(B, 1)):
    for l in range(L, 0, -1):
        t = out[b, l]
        if t[10:30].any():
            r = max(t[5], 1)
            print('b %d l %2d per round: busy per wave %s | LDS drain + barrier per wave %s' % (b, l, (t[10:18] // r).tolist(), (t[18:26] // r).tolist()))
            names = 'operands | Gram | lane sums | parameters | rotations + stores issued | LDS drain | barrier'
            print('      wave 0: %s = %s;   wave 5: %s' % (names, (t[26:33] // r).tolist(), (t[33:40] // r).tolist()))
e.close()
