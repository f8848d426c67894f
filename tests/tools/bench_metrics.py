"""What the optional reciprocal metrics cost per step on the GPU (128 x L32, one engine of 3 restarts), next to the oracle's routines
on one host core (a study script: imports the oracle).   python tests/tools/bench_metrics.py [cfg=3] [B=3] [steps=40]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(HERE, '..'))
np.seterr(all='ignore')
from oracle import metrics as M, mtip as OM                              # noqa: E402
from xframe_amd.fxs import hostsetup as hs, synthetic as S               # noqa: E402
from xframe_amd.fxs.engine import Engine                                 # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
N, L = S._SIZES[cfg]
ed = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, max_q=S.data_cutoff(N))
data, _ = S.make_invariants(ed, N, L)
ed.close()
data['xray_wavelength'] = S.XRAY_WAVELENGTH
cases = [('no reciprocal metric', []), ('deg2_invariant_l2_diff', ['deg2_invariant_l2_diff']),
         ('+ deg2_ranked, l2_projection_diff', ['deg2_invariant_l2_diff', 'deg2_ranked_invariant_l2_diff', 'l2_projection_diff']),
         ('II_error', ['II_error']), ('ccd_diff', ['ccd_diff']), ('fqc_error', ['fqc_error']),
         ('all six', ['deg2_invariant_l2_diff', 'deg2_ranked_invariant_l2_diff', 'l2_projection_diff', 'II_error', 'ccd_diff', 'fqc_error'])]
base = None
Ilm_host = None
for name, calc in cases:
    opt = OM.deep_update(S.config_overrides(cfg), {'main_loop': {'error': {'methods': {'reciprocal': {'calculate': calc, 'ccd_diff': {'C_order': 2}}}}}})
    e = Engine(opt, data, n_batch=B)
    for b in range(B):
        e.set_density(b, hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000 + b), e.rsetup.integrated_intensity,
                                         e.int_wr, e.int_wt))
    e.init_state()
    e.run('HIO', True, np.full(10, 0.45))
    t = time.perf_counter()
    e.run('HIO', True, np.full(steps, 0.45))
    dt = (time.perf_counter() - t) / steps
    base = dt if base is None else base
    print(f'{name:38s} {dt * 1e3:7.3f} ms per step of {B} restarts ({(dt - base) * 1e6:+7.1f} us)')
    if Ilm_host is None:
        F = np.stack([e.reciprocal_density(b) for b in range(B)])
        Ilm_host = e.sht_forward(F, 1)[0]
        pm = [np.asarray(e.rsetup.projection_matrices[l]) for l in range(L + 1)]
        rmask = np.asarray(e.rsetup.radial_mask)
        qs = np.asarray(e.qs)
    e.close()
# the oracle's routines on the same coefficients, one core
Il = [Ilm_host[:, l * l:(l + 1) ** 2] for l in range(L + 1)]
used = {l: l for l in range(L + 1)}
inv = rmask[:, :, None] * rmask[:, None, :]
ref = np.array([p @ p.conj().T for p in pm])
for name, make in (('II_error', lambda: M.II_error_routine(qs, ref, used, inv)), ('ccd_diff', lambda: M.ccd_diff_routine(qs, ref, used, 1.0, inv, 2, S.XRAY_WAVELENGTH)),
                   ('fqc_error', lambda: M.fqc_error_routine(qs, ref, used, inv, S.XRAY_WAVELENGTH))):
    t = time.perf_counter()
    f = make()
    t_setup = time.perf_counter() - t
    t = time.perf_counter()
    for _ in range(3):
        f(Il)
    print(f'oracle {name:10s}: {(time.perf_counter() - t) / 3 * 1e3:8.1f} ms per evaluation of one restart (set-up {t_setup:.2f} s), one core')
