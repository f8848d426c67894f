"""Probe: Jacobi sweeps / projection time along the tutorial schedule at a BASELINE config (GPU box)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
np.seterr(all='ignore')
from xframe_amd.fxs import hostsetup as hs, synthetic as S
from xframe_amd.fxs.engine import Engine
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N, L = S._SIZES[cfg]
ed = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, max_q=S.data_cutoff(N))
data, _ = S.make_invariants(ed, N, L); ed.close()
e = Engine(S.config_overrides(cfg), data, n_batch=B, fused=True)
for b in range(B):
    e.set_density(b, hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000 + b),
                                     e.rsetup.integrated_intensity, e.int_wr, e.int_wt))
e.init_state()
ramp = hs.ExponentialRamp(0.5, 0.4, -1 / 250, 500)
step = 0
def run(kind, n):
    global step
    for i in range(n):
        e.synchronize(); t = time.perf_counter()
        err, _ = e.run(kind, True, [ramp.eval(step)])
        dt = time.perf_counter() - t
        step += 1
        if i in (0, 1, 2, 5, 10, 20, n - 1):
            sw = e.jacobi_sweeps()[0][::2]
            print(kind, step, 'err %.3e' % err[0, 0], 'ms %.2f' % (dt * 1e3), 'sweeps(even l)', [int(x) for x in sw],
                  'active cols', [int(x) for x in e.jacobi_active_columns()[0][::2]])
run('HIO', 60)
print('SW', e.shrinkwrap(20.0, 0.09, 6e-3))
run('ER', 40)
run('HIO', 10)
