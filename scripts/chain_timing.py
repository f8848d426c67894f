"""GPU: phase stamps of the chained inverse -> forward SHT kernels (k_sht_chain, mtip_debug_chain_timing) during a phasing step
at a BASELINE config, per kind (store+square / modulus / real-space), mean over the shells of the launch, in s_memtime ticks
(100 MHz on gfx950: 10 ns each).
usage: python scripts/chain_timing.py [cfg=3] [B=3] [n_hio=5]"""
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
n_hio = int(sys.argv[3]) if len(sys.argv) > 3 else 5
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
if os.environ.get('CHAIN_STAMPS', '1') != '0':
    e.lib.mtip_debug_chain_timing(e.ctx, None)
e.profile(True)
e.run('HIO', True, np.full(4, 0.45))
for fam in ('sht_chain', 'sht_chain_modulus', 'sht_chain_real', 'hankel', 'polar'):
    ms, n = e.profile_get(fam)
    if n:
        print('%-18s %7.1f us per launch (hipEvent bracket, %d launches, B = %d)' % (fam, 1e3 * ms / n, n, B))
if os.environ.get('CHAIN_STAMPS', '1') == '0':
    e.close()
    sys.exit(0)
SL = 18
out = np.zeros((3, B * N, SL), np.int64)
e._ck(e.lib.mtip_debug_chain_timing(e.ctx, _lib.ptr(out)))
for k, kind in enumerate(('store + |.|^2', 'modulus', 'real-space')):
    t = out[k]
    ok = t[:, 17] > 0
    if not ok.any():
        continue
    t = t[ok].astype(float)
    tot = t[:, 17] - t[:, 0]
    rows = [('tables staged', t[:, 1] - t[:, 0]), ('Legendre synthesis (wave 0)', t[:, 2] - t[:, 1]), ('   its barrier', t[:, 3] - t[:, 2]),
            ('row groups of wave 0: inverse step 1', t[:, 4]), ('row groups of wave 0: step 2 + epilogue + forward phase 1', t[:, 6]),
            ('row groups of wave 0: forward phase 2', t[:, 8]), ('   barrier (the other waves finish their groups)', t[:, 9]),
            ('Legendre sums (+ error sums), wave 0', t[:, 16] - t[:, 3] - t[:, 4:10].sum(1)), ('reduce + store (incl. waiting for the other waves)', t[:, 17] - t[:, 16])]
    print('\n%s: %d shells; workgroup lifetime mean %.0f ticks (min %.0f, max %.0f)' % (kind, len(t), tot.mean(), tot.min(), tot.max()))
    for nm, col in rows:
        print('   %-52s %8.0f  (%4.1f %%)' % (nm, col.mean(), 100 * col.mean() / tot.mean()))
    st = np.sort(t[:, 0] - t[:, 0].min())
    print('   start offsets (ticks after the first workgroup): median %.0f, 75 %% %.0f, max %.0f' % (np.median(st), st[int(0.75 * len(st))], st[-1]))
e.close()
