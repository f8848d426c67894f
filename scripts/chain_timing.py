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
names = ['tables staged', 'synthesis (wave 0)', 'synthesis barrier']
for p in range(2):
    names += ['p%d inverse step 1' % p, 'p%d barrier' % p, 'p%d step 2 + epilogue + fwd phase 1' % p, 'p%d barrier' % p, 'p%d fwd phase 2' % p, 'p%d barrier' % p]
names += ['Legendre sums (+ error sums)', 'reduce + store']
idx = list(range(1, 18))
for k, kind in enumerate(('store + |.|^2', 'modulus', 'real-space')):
    t = out[k]
    ok = t[:, 17] > 0
    if not ok.any():
        continue
    t = t[ok]
    d = np.diff(t[:, [0] + idx], axis=1)
    tot = t[:, 17] - t[:, 0]
    span = t[:, 17].max() - t[:, 0].min()
    print('\n%s: %d shells; workgroup lifetime mean %.0f ticks (min %.0f, max %.0f); launch span %.0f ticks' % (kind, len(t), tot.mean(), tot.min(), tot.max(), span))
    for nm, col in zip(names, d.T):
        print('   %-40s %8.0f  (%4.1f %%)' % (nm, col.mean(), 100 * col.mean() / tot.mean()))
    # how many workgroups started late (second round on their CU)
    st = t[:, 0] - t[:, 0].min()
    print('   start offsets: %d workgroups within 200 ticks of the first, median of the rest %.0f ticks' % ((st < 200).sum(), np.median(st[st >= 200]) if (st >= 200).any() else 0))
e.close()
