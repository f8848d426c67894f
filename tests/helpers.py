"""Shared helpers for the test-suite (oracle side)."""
import numpy as np


def rel_l2(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    n = np.linalg.norm(b.ravel())
    d = np.linalg.norm((a - b).ravel())
    return d / n if n > 0 else d


class OracleTransforms:
    """Adapter giving ``xframe_amd.fxs.synthetic.make_invariants`` the oracle's transforms."""

    def __init__(self, fp):
        self.fp, self.rs, self.thetas, self.phis = fp, fp.rs, fp.sht.theta, fp.sht.phi

    def ft(self, x):
        return self.fp.ft(x)

    def forward_l(self, x):
        return self.fp.sht.forward_l(x)

    def hermitian_eig(self, mats):
        """numpy eigensolver in Engine.hermitian_eig's layout (descending eigenvalues, eigenvectors in columns): lets the synthetic
        front half build inputs for the CPU oracle without a device (test infrastructure only)"""
        w, v = np.linalg.eigh(np.asarray(mats))
        return w[:, ::-1].copy(), np.ascontiguousarray(v[:, :, ::-1])


def data_from_golden(g, L, prefix='data_'):
    pms = np.empty(L + 1, dtype=object)
    for l in range(L + 1):
        pms[l] = g[f'{prefix}pm{l}']
    return {'dimensions': 3, 'xray_wavelength': 1.23984, 'average_intensity': g[f'{prefix}aint'],
            'data_radial_points': g[f'{prefix}q'], 'data_angular_points': np.zeros(1), 'max_order': L,
            'data_projection_matrices': pms}


def golden_settings(N, L, extra=None):
    from oracle import mtip as OM
    from xframe_amd.fxs import synthetic as S
    o = OM.deep_update(OM.default_settings(), S.config_overrides(1))
    o = OM.deep_update(o, {'grid': {'n_radial_points': N, 'max_order': L},
                           'projections': {'reciprocal': {'used_order_ids': np.arange(L + 1)}},
                           'GPU': {'use': True}, 'multi_process': {'use': False}})
    if extra:
        o = OM.deep_update(o, extra)
    return o


# ---- rotation / inversion / translation invariant summaries of a reconstruction (converged-run comparison) ------------
def radial_profile(rho):
    """sqrt of the angular mean of |rho|^2 per shell (Gauss-Legendre weights in theta, uniform in phi)"""
    from scipy.special import roots_legendre
    wt = roots_legendre(rho.shape[1])[1]
    return np.sqrt((np.abs(rho) ** 2 * wt[None, :, None]).sum((1, 2)) / (wt.sum() * rho.shape[2]))


def bl_error(Bl, projection_matrices, radial_mask, used_orders, n_particles):
    """sum_l |B_l - B_l^data|^2 / sum_l |B_l^data|^2 on the masked shells (the metric of fxs_IO_methods.py:408-447 summed
    over the orders); projection_matrices / radial_mask are indexed like used_orders"""
    num = den = 0.0
    for i, l in enumerate(used_orders):
        V = projection_matrices[i]
        m = radial_mask[i]
        Bd = (V @ V.conj().T)[np.ix_(m, m)]
        if l == 0:
            Bd = Bd / n_particles
        num += (np.abs(Bl[l][np.ix_(m, m)] - Bd) ** 2).sum()
        den += (np.abs(Bd) ** 2).sum()
    return num / den
