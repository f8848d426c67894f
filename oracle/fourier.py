"""Oracle spherical Fourier transform pair + grids (TEST INFRASTRUCTURE).

* radial grids: ``ft_grid_pairs.py:282-291`` (midpoint), 274-281 (trapz/zernike)
* ``generate_ft``: ``fourier_transforms.py:49-86``  ft = iSHT o Hankel o SHT
* ``SphericalIntegrator``: ``xframe/library/mathLibrary.py:1212-1240``
"""
import numpy as np
from scipy.special import roots_legendre
from . import hankel as _hk


def radial_grid_midpoint(max_q, n_radial_points, reciprocity_coefficient):
    """ft_grid_pairs.py:282-291."""
    N = n_radial_points
    r_max = _hk.reciprocal_cutoff(max_q, N, reciprocity_coefficient)
    dr = r_max / N
    dq = max_q / N
    rs = np.linspace(0 + dr / 2, r_max - dr / 2, num=N, endpoint=True)
    qs = np.linspace(0 + dq / 2, max_q - dq / 2, num=N, endpoint=True)
    return rs, qs


def radial_grid_gauss(max_q, n_radial_points, reciprocity_coefficient):
    """ft_grid_pairs.py:293-300: Gauss-Legendre nodes mapped to [0, R] and [0, Q]."""
    from scipy.special import roots_legendre
    r_max = reciprocity_coefficient * n_radial_points / max_q
    xs, _ = roots_legendre(n_radial_points)
    return r_max / 2 * xs + r_max / 2, max_q / 2 * xs + max_q / 2


def radial_grid_trapz(max_q, n_radial_points, reciprocity_coefficient):
    """ft_grid_pairs.py:274-281 (uniformGrid_func with endpoint)."""
    N = n_radial_points
    r_max = _hk.reciprocal_cutoff(max_q, N, reciprocity_coefficient)
    return np.linspace(0, r_max, N), np.linspace(0, max_q, N)


class GridPair:
    """Plain stand-in for FTGridPair: radial points + angular nodes; grid arrays (N,nt,np,3)."""

    def __init__(self, rs, qs, thetas, phis):
        self.rs, self.qs, self.thetas, self.phis = rs, qs, thetas, phis
        self.shape = (len(rs), len(thetas), len(phis))

    def real_grid(self):
        return np.stack(np.meshgrid(self.rs, self.thetas, self.phis, indexing='ij'), -1)

    def reciprocal_grid(self):
        return np.stack(np.meshgrid(self.qs, self.thetas, self.phis, indexing='ij'), -1)


class FourierPair:
    """generate_ft (fourier_transforms.py:49-86) for dimensions=3."""

    def __init__(self, sht, n_radial_points, max_q, reciprocity_coefficient=2.0, mode='midpoint'):
        self.sht = sht
        self.mode = mode
        self.kappa = reciprocity_coefficient
        N = n_radial_points
        if mode == 'midpoint':
            self.rs, self.qs = radial_grid_midpoint(max_q, N, reciprocity_coefficient)
            wraw = _hk.spherical_mid_weights(sht.l_max, N, reciprocity_coefficient)
        elif mode == 'trapz':
            self.rs, self.qs = radial_grid_trapz(max_q, N, reciprocity_coefficient)
            wraw = _hk.spherical_trapz_weights(sht.l_max, N, reciprocity_coefficient)
        elif mode == 'Zernike':                     # the trapz grid (ft_grid_pairs.py:545), its own weights
            self.rs, self.qs = radial_grid_trapz(max_q, N, reciprocity_coefficient)
            wraw = _hk.zernike_weights_as_loaded(sht.l_max, N, reciprocity_coefficient)
        elif mode == 'gauss':
            self.rs, self.qs = radial_grid_gauss(max_q, N, reciprocity_coefficient)
            wraw = _hk.spherical_gauss_weights(sht.l_max, N, reciprocity_coefficient)
        else:
            raise AssertionError(mode)
        # reconstruct.py:329 r_max = max(real_radial_points) -- NOT the cutoff R:
        self.r_max = np.max(self.rs)
        self.raw_weights = wraw
        self.w = _hk.assemble_weights_mode(wraw, self.r_max, reciprocity_coefficient, mode)
        self.trapz = mode in ('trapz', 'Zernike')       # sums skip shell 0 of the input (hankel_transforms.py:647-652: ht_modes[:2])
        self.grid = GridPair(self.rs, self.qs, sht.theta, sht.phi)

    def hankel(self, c):
        return _hk.apply_direct(self.w['forward'], c, self.trapz)

    def ihankel(self, c):
        return _hk.apply_direct(self.w['inverse'], c, self.trapz)

    def ft(self, data):
        return self.sht.inverse_d(self.hankel(self.sht.forward_d(data)))

    def ift(self, data):
        return self.sht.inverse_d(self.ihankel(self.sht.forward_d(data)))


class SphericalIntegrator:
    """mathLibrary.py:1212-1240: int f = trapz_r[ r^2 (pi/n_theta) sum_theta w_theta sum_phi f ]."""

    def __init__(self, rs, n_theta):
        self.rs = np.asarray(rs)
        self.n_theta = n_theta
        self.gauss_weights = roots_legendre(n_theta)[1]
        self.max_r = np.max(rs)
        self.norm = 4 / 3 * np.pi * self.max_r ** 3

    def integrate(self, values):
        w = self.gauss_weights
        s2 = np.pi / self.n_theta * np.sum(w[None, :] * np.sum(values, axis=2), axis=1)
        f = s2 * self.rs ** 2
        # np.trapz(f, x=rs): sum 0.5*(x[i+1]-x[i])*(f[i+1]+f[i])
        d = np.diff(self.rs)
        return np.sum(d * (f[1:] + f[:-1]) / 2.0)

    def integrate_normed(self, values):
        return self.integrate(values) / self.norm
