"""GPU: BASELINE config 5 (256 shells x L_max = 48, dense B_l metric on) on one GPU -- ms per step and the family timers; run
under rocprofv3 (--kernel-trace --stats, or --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 with
scripts/pmc_mfma.py) for the kernel statistics and the MFMA utilisation north_star asks for.
usage: python scripts/config5_steps.py [restarts=2] [steps=6]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
np.seterr(all='ignore')
from xframe_amd.fxs import synthetic as S, hostsetup as hs      # noqa: E402
from xframe_amd.fxs.engine import Engine                        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
N, L = S._SIZES[5]
ed = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, max_q=S.data_cutoff(N))
data, _ = S.make_invariants(ed, N, L)
ed.close()
e = Engine(S.config_overrides(5), data, n_batch=B)
for b in range(B):
    e.set_density(b, hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000 + b),
                                     e.rsetup.integrated_intensity, e.int_wr, e.int_wt))
e.init_state()
e.run('HIO', True, np.full(2, 0.45)); e.synchronize()
t = time.time(); e.run('HIO', True, np.full(steps, 0.45)); e.synchronize(); dt = time.time() - t
print('config 5: %.2f ms per step (%d restarts in one engine, %d x L%d, grid %s)' % (dt / steps * 1e3, B, N, L, e.shape))
e.profile(True); e.run('HIO', True, np.full(3, 0.45)); e.synchronize()
for f in ('sht_fwd', 'sht_inv', 'sht_inv_modulus', 'sht_inv_real', 'hankel', 'proj', 'polar', 'deg2_metric'):
    ms, n = e.profile_get(f)
    if n:
        print('%-16s %8.3f ms per launch (%d launches)' % (f, ms / n, n))
print('sweeps per order (restart 0):', [int(x) for x in e.jacobi_sweeps()[0][2::2]])
e.close()
