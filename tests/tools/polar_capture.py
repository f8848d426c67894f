"""Study helper (CPU only): capture the polar-factor inputs X_l = PD_l I_l and the V_l of a running oracle
reconstruction (config 3 by default) at chosen steps, into an .npz for scripts/polar_algorithms_study.py."""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
np.seterr(all='ignore')
from helpers import OracleTransforms                                   # noqa: E402
from oracle import mtip as OM                                          # noqa: E402
from oracle.fourier import FourierPair                                 # noqa: E402
from oracle.sht import SHT                                             # noqa: E402
from xframe_amd.fxs import synthetic as S                              # noqa: E402

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out = sys.argv[2] if len(sys.argv) > 2 else '/tmp/polar_capture_cfg%d.npz' % cfg
steps = [int(s) for s in sys.argv[3].split(',')] if len(sys.argv) > 3 else [1, 2, 3, 10, 30, 59, 61, 62, 70, 99, 101, 130]
N, L = S._SIZES[cfg]
fpd = FourierPair(SHT(L), N, S.data_cutoff(N), 2.0)
data, _ = S.make_invariants(OracleTransforms(fpd), N, L)
opt = OM.deep_update(OM.default_settings(), S.config_overrides(cfg))
om = OM.MTIP(opt, data)
captured = {}
orig = om.rp.approximate_unknowns
step = [0]


def spy(Ilm):
    step[0] += 1
    if step[0] in steps:
        for PD, oid in zip(om.rp.PDs, om.rp.used_orders.values()):
            captured['X_s%d_l%d' % (step[0], oid)] = (PD @ Ilm[oid]).copy()
    if step[0] > max(steps):
        raise StopIteration
    return orig(Ilm)


om.rp.approximate_unknowns = spy
rho0 = om.density_guess(np.random.default_rng(1000))
try:
    om.phasing_loop(rho0=rho0)
except StopIteration:
    pass
for i, oid in enumerate(om.rp.used_orders.values()):
    captured['V_l%d' % oid] = om.rp.projection_matrices[i]
captured['steps'] = np.array(steps)
np.savez(out, **captured)
print('saved', out, len(captured))
