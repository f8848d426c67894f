"""Per-order deviations of the real-arithmetic projection (k_rproj) from the oracle's complex SVD route and from the general
complex kernel, at one size (GPU box; uses the oracle as the checker)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
np.seterr(all='ignore')
from helpers import OracleTransforms, golden_settings, rel_l2
from oracle.fourier import FourierPair
from oracle.sht import SHT
from oracle import mtip as OM
from xframe_amd.fxs import synthetic as S
from xframe_amd.fxs.engine import Engine

N, L = int(sys.argv[1]), int(sys.argv[2])
sht = SHT(L)
fpd = FourierPair(sht, N, S.data_cutoff(N), 2.0)
data, _ = S.make_invariants(OracleTransforms(fpd), N, L)
opt = golden_settings(N, L)
e = Engine(opt, data, n_batch=1, lib_path=(sys.argv[3] if len(sys.argv) > 3 else None))
om = OM.MTIP(opt, data)
rng = np.random.default_rng(1)
for rep in range(2):
    grid = rng.uniform(0.0, 1.0, (1, N, sht.n_theta, sht.n_phi)) * rng.uniform(0.5, 2.0, (1, N, 1, 1))
    Ilm = np.stack([np.concatenate(sht.forward_l(g.astype(complex)), axis=1) for g in grid])
    proj = e.project_coefficients(Ilm, real_intensity=True)
    unk_r = e.unknowns(0)
    sw_r = e.jacobi_sweeps()[0]
    proj_c = e.project_coefficients(Ilm)
    unk_c = e.unknowns(0)
    sw_c = e.jacobi_sweeps()[0]
    Il = [Ilm[0][:, l * l:(l + 1) ** 2] for l in range(L + 1)]
    unk = om.rp.approximate_unknowns(Il)
    ref = np.concatenate(om.rp.mtip_projection(Il, unk), axis=1)
    print('rep', rep, 'proj real vs oracle %.2e, complex vs oracle %.2e' % (rel_l2(proj[0], ref), rel_l2(proj_c[0], ref)))
    for i, l in enumerate(om.rp.used_orders.values()):
        V = om.rp.projection_matrices[l]
        if np.abs(V).max() == 0 or l == 0:
            continue
        s = np.linalg.svd(om.rp.PDs[i] @ Il[l], compute_uv=False)
        print('  l %2d  V U: real %.2e  complex %.2e   coefficients of the order: real %.2e complex %.2e   sweeps real %d complex %d   sigma min/max %.1e'
              % (l, rel_l2(V @ unk_r[l], V @ unk[i]), rel_l2(V @ unk_c[l], V @ unk[i]),
                 rel_l2(proj[0][:, l * l:(l + 1) ** 2], ref[:, l * l:(l + 1) ** 2]), rel_l2(proj_c[0][:, l * l:(l + 1) ** 2], ref[:, l * l:(l + 1) ** 2]),
                 sw_r[l] & 255, sw_c[l] & 255, s[-1] / s[0]))
e.close()
