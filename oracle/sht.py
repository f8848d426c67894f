"""Oracle spherical-harmonic transform (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates the convention of the reference's shtns wrapper
``xframe/externalLibraries/shtns_plugin.py`` (class ``sh``):

* ``shtns.sht(l_max)`` default = orthonormal Y_lm with Condon-Shortley phase
  (shtns_plugin.py:20), complex transforms ``analys_cplx`` / ``synth_cplx`` with
  coefficient index ``l*(l+1)+m`` and length ``(l_max+1)**2``
  (shtns_plugin.py:24,105-114).
* Gauss grid, ``theta = arccos(cos_theta)`` running north->south,
  ``phi_j = 2 pi j / n_phi`` (shtns_plugin.py:130-133).
* default angular grid size ``n_phi = 2**(int(log2((N+1)*l_max))+1)``,
  ``n_theta = n_phi//2`` with anti-aliasing degree N=2 (shtns_plugin.py:94-101).
* layouts: 'direct' ``(Nq,(L+1)^2)`` (250-261); 'lm' = list over l of
  ``(Nq,2l+1)``, m ascending (166-170, 181-184); 'ml' = list over
  ``m in (0,1..L,-L..-1)`` of ``(Nq, L-|m|+1)``, l ascending (105-110, 171-194).

shtns itself is third-party and absent: PARITY UNPINNED for the SHT values;
pinned by analytic known answers in tests/test_oracle_sht.py.
"""
import numpy as np
from scipy.special import roots_legendre

try:  # scipy >= 1.15
    from scipy.special import sph_harm_y as _sph_harm_y

    def _ylm(l, m, theta, phi):
        return _sph_harm_y(l, m, theta, phi)
except ImportError:  # pragma: no cover
    from scipy.special import sph_harm as _sph_harm

    def _ylm(l, m, theta, phi):
        return _sph_harm(m, l, phi, theta)


def angular_grid_size(l_max, anti_aliazing_degree=2):
    """shtns_plugin.py:94-101 (n_angular_step_from_max_order)."""
    n_phi = 2 ** (int(np.log2((anti_aliazing_degree + 1) * l_max)) + 1)
    return n_phi // 2, n_phi


class SHT:
    """Complex orthonormal SHT on a Gauss-Legendre x uniform-phi grid."""

    def __init__(self, l_max, n_theta=None, n_phi=None, anti_aliazing_degree=2):
        l_max = int(l_max)
        nt, nphi = angular_grid_size(l_max, anti_aliazing_degree)
        # shtns_plugin.py:121-124: non-int / bool -> formula value
        if (not isinstance(n_theta, (int, np.integer))) or isinstance(n_theta, bool) or n_theta == 0:
            n_theta = nt
        if (not isinstance(n_phi, (int, np.integer))) or isinstance(n_phi, bool) or n_phi == 0:
            n_phi = nphi
        self.l_max = l_max
        self.n_theta = int(n_theta)
        self.n_phi = int(n_phi)
        assert self.n_phi > 2 * l_max and self.n_theta > l_max
        self.n_coeff = (l_max + 1) ** 2
        x, w = roots_legendre(self.n_theta)
        self.cos_theta = x[::-1].copy()          # north -> south
        self.weights = w[::-1].copy()
        self.theta = np.arccos(self.cos_theta)
        self.phi = 2 * np.pi * np.arange(self.n_phi) / self.n_phi
        # index bookkeeping (shtns_plugin.py:105-114, 268-274)
        ls = np.arange(l_max + 1, dtype=int)
        self.l = ls
        self.m = np.concatenate((ls, -ls[:0:-1]))
        self.cplx_m_indices = [ls[abs(m):] * (ls[abs(m):] + 1) + m for m in self.m]
        self.cplx_l_indices = [slice(l ** 2, l ** 2 + 2 * l + 1) for l in range(l_max + 1)]
        self.cplx_l_split_indices = np.arange(1, l_max + 1) ** 2
        # per-m Legendre matrices  Y_lm(theta, 0)  (real), shape (L-|m|+1, n_theta)
        self._ylm0 = {}
        for m in range(-l_max, l_max + 1):
            lv = np.arange(abs(m), l_max + 1)
            self._ylm0[m] = np.real(_ylm(lv[:, None], m, self.theta[None, :], 0.0))

    # ---- 'direct' layout -------------------------------------------------
    def forward_d(self, data):
        """analys_cplx per shell: f_lm = sum_theta w (2pi/nphi) sum_phi f conj(Y_lm)."""
        data = np.asarray(data, dtype=complex)
        lead = data.shape[:-2]
        g = np.fft.fft(data, axis=-1) * (2 * np.pi / self.n_phi)
        out = np.zeros(lead + (self.n_coeff,), dtype=complex)
        for m in range(-self.l_max, self.l_max + 1):
            lv = np.arange(abs(m), self.l_max + 1)
            A = self._ylm0[m] * self.weights[None, :]
            out[..., lv * (lv + 1) + m] = g[..., :, m % self.n_phi] @ A.T
        return out

    def inverse_d(self, coeff):
        """synth_cplx per shell: f = sum_lm f_lm Y_lm."""
        coeff = np.asarray(coeff, dtype=complex)
        lead = coeff.shape[:-1]
        G = np.zeros(lead + (self.n_theta, self.n_phi), dtype=complex)
        for m in range(-self.l_max, self.l_max + 1):
            lv = np.arange(abs(m), self.l_max + 1)
            G[..., :, m % self.n_phi] = coeff[..., lv * (lv + 1) + m] @ self._ylm0[m]
        return np.fft.ifft(G, axis=-1) * self.n_phi

    # ---- 'lm' layout: list over l of (Nq, 2l+1) --------------------------
    def forward_l(self, data):
        c = self.forward_d(data)
        return [np.array(c[..., s]) for s in self.cplx_l_indices]

    def inverse_l(self, coeff_list):
        return self.inverse_d(np.concatenate(coeff_list, axis=-1))

    # ---- 'ml' layout: list over m in (0..L,-L..-1) of (Nq, L-|m|+1) ------
    def forward_m(self, data):
        c = self.forward_d(data)
        return [np.array(c[..., idx]) for idx in self.cplx_m_indices]

    def inverse_m(self, coeff_list):
        lead = coeff_list[0].shape[:-1]
        c = np.zeros(lead + (self.n_coeff,), dtype=complex)
        for m_id, idx in enumerate(self.cplx_m_indices):
            c[..., idx] = coeff_list[m_id]
        return self.inverse_d(c)

    def test(self, data):
        return self.inverse_d(self.forward_d(np.asarray(data) + 0.j))
