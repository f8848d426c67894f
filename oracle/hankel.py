"""Oracle radial Hankel transform (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows ``xframe/projects/fxs/projectLibrary/hankel_transforms.py``:
* ``calc_spherical_mid_weights``  399-410   w[l,p,k] = (p+1/2)^2 j_l(kappa (k+1/2)(p+1/2)/N)
* ``calc_spherical_trapz_weights`` 322-333  w[l,p,k] = p^2 j_l(kappa k p/N), p=1..N-1, k=0..N-1
* ``assemble_weights_mid`` 426-452 / ``assemble_weights_trapz`` 349-375:
  moveaxis -> (p,k,l); forward x (-i)^l (R/N)^3 sqrt(2/pi); inverse x (+i)^l (Q/N)^3 sqrt(2/pi),
  Q = kappa N / R (``mathLibrary.py:1169-1176``)
* ``generate_spherical_ht`` 642-658 (CPU, 'ml' lists) and the OpenCL kernel
  ``apply_weights`` 702-731 (midpoint) / 671-700 (trapz), 'direct' layout:
  out[k,lm] = sum_p w[p,k,l(lm)] f[p,lm]   (trapz: f[p+1], p < N-1)
"""
import numpy as np
from scipy.special import eval_jacobi, roots_legendre, spherical_jn


def reciprocal_cutoff(cutoff, n_points, reciprocity_coefficient=np.pi):
    """mathLibrary.py:1169-1176 polar_spherical_dft_reciprocity_relation_radial_cutoffs."""
    return reciprocity_coefficient * n_points / cutoff


def spherical_mid_weights(l_max, n_radial_points, reciprocity_coefficient):
    """hankel_transforms.py:399-410. Returns real (L+1, N, N) indexed [l, p, k]."""
    N = n_radial_points
    ps = np.arange(N) + 0.5
    ks = np.arange(N) + 0.5
    ls = np.arange(l_max + 1)
    arg = ks[None, :] * ps[:, None] * reciprocity_coefficient / N
    jl = spherical_jn(ls[:, None, None], arg[None, :, :])
    return ps[None, :, None] ** 2 * jl


def spherical_trapz_weights(l_max, n_radial_points, reciprocity_coefficient):
    """hankel_transforms.py:322-333. Returns real (L+1, N-1, N) indexed [l, p-1, k]."""
    N = n_radial_points
    ps = np.arange(1, N)
    ks = np.arange(N)
    ls = np.arange(l_max + 1)
    arg = ks[None, :] * ps[:, None] * reciprocity_coefficient / N
    jl = spherical_jn(ls[:, None, None], arg[None, :, :])
    return ps[None, :, None] ** 2 * jl


def spherical_gauss_weights(l_max, n_radial_points, reciprocity_coefficient):
    """hankel_transforms.py:477-490 (calc_spherical_gauss_weights): Gauss-Legendre nodes x_i on [-1, 1], p = k = x + 1,
    w[l,p,k] = p^2 j_l(k p kappa N / 4) w_p.  Real (L+1, N, N) indexed [l, p, k]."""
    N = n_radial_points
    xi, wgauss = roots_legendre(N)              # (mathLibrary.gauss_legendre = scipy's roots_legendre)
    ps = xi + 1
    ks = xi + 1
    ls = np.arange(l_max + 1)
    jl = spherical_jn(ls[:, None, None], (ks[None, :] * ps[:, None] * reciprocity_coefficient * N / 4)[None, :, :])
    return ps[None, :, None] ** 2 * jl * wgauss[None, :, None]


def nd_zernike_radial(l, s_max, points, dimension=3):
    """mathLibrary.py:805-819 (eval_ND_zernike_polynomials): R^l_s(p) for s = l, l+2, ... <= s_max, rows s."""
    s = np.arange(l, s_max + 1, 2)
    return np.array([((-1) ** ((si - l) / 2)) * (points ** l) * eval_jacobi((si - l) / 2, l + dimension / 2 - 1, 0, 1 - 2 * (points ** 2))
                     for si in s])


def spherical_zernike_weights(l_max, n_radial_points, expansion_limit, reciprocity_coefficient):
    """hankel_transforms.py:88-131 (calc_spherical_zernike_weights): p = 1..N-1, k = 0..N-1,
    w[l,p,k] = (p^2 / k) sum_s (-1)^((s-l)/2) (2s+3) R^l_s(p/N) j_{s+1}(k kappa), the k = 0 column p^2 kappa delta_{l0} (s = l = 0 only).
    Real (L+1, N-1, N) indexed [l, p-1, k]."""
    N = n_radial_points
    ps = np.arange(1, N)
    ks = np.arange(N)
    out = []
    for l in range(l_max + 1):
        Zk = nd_zernike_radial(l, expansion_limit, ps / N, 3)                      # (n_s, N-1)
        s = np.arange(l, expansion_limit + 1, 2)
        pref = (-1) ** ((s - l) / 2) * (2 * s + 3)
        jp = spherical_jn((s + 1)[:, None], (ks[1:] * reciprocity_coefficient)[None, :])   # (n_s, N-1)
        summ = np.zeros((len(s), N - 1, N))
        summ[:, :, 1:] = pref[:, None, None] * Zk[:, :, None] * jp[:, None, :]
        if l == 0:
            summ[0, :, 0] = reciprocity_coefficient
        out.append(summ.sum(axis=0))
    w = np.array(out)
    c_kp = np.zeros((N - 1, N))
    c_kp[:, 1:] = np.square(ps)[:, None] / ks[None, 1:]
    c_kp[:, 0] = np.square(ps)
    return w * c_kp[None, :, :]


def zernike_weights_as_loaded(l_max, n_radial_points, reciprocity_coefficient):
    """The Zernike weights the reference's loader ends up with: load_fourier_transform_weights (fourier_transforms.py:17-35) ->
    generate_weightDict(max_order, n, reciprocity_coefficient=rc, mode='Zernike') -> generate_weightDict_zernike(max_order, n, rc,
    ...) (hankel_transforms.py:26): rc lands in the third positional parameter, `expansion_limit` (52), so the limit is
    max(rc, max_order) (62) and the weights' own reciprocity coefficient keeps its default pi."""
    lim = max(reciprocity_coefficient, l_max)
    return spherical_zernike_weights(l_max, n_radial_points, lim, np.pi)


def assemble_weights_mode(weights, r_max, reciprocity_coefficient, mode):
    """hankel_transforms.py:36-48: midpoint / trapz (426-452, 349-375), gauss (509-535: (R/2)^3, (Q/2)^3), Zernike
    (270-300: (R/N)^3 sqrt(2/pi^3), (Q/N)^3 sqrt(2/pi^3))."""
    if mode in ('midpoint', 'trapz'):
        return assemble_weights(weights, r_max, reciprocity_coefficient)
    n = weights.shape[-1]
    orders = np.arange(weights.shape[0])
    q_max = reciprocal_cutoff(r_max, n, reciprocity_coefficient)
    if mode == 'gauss':
        fp, ip = (r_max / 2) ** 3 * np.sqrt(2 / np.pi), (q_max / 2) ** 3 * np.sqrt(2 / np.pi)
    elif mode == 'Zernike':
        fp, ip = (r_max / n) ** 3 * np.sqrt(2 / np.pi ** 3), (q_max / n) ** 3 * np.sqrt(2 / np.pi ** 3)
    else:
        raise AssertionError(mode)
    w = np.moveaxis(weights, 0, 2)
    return {'forward': w * ((-1.j) ** orders[None, None, :] * fp), 'inverse': w * ((1.j) ** orders[None, None, :] * ip)}


def assemble_weights(weights, r_max, reciprocity_coefficient):
    """hankel_transforms.py:426-452 (3-D branch; trapz 349-375 is identical arithmetic).

    Returns dict with complex 'forward' / 'inverse' arrays of shape (Np, Nk, L+1)."""
    n_radial_points = weights.shape[-1]
    orders = np.arange(weights.shape[0])
    q_max = reciprocal_cutoff(r_max, n_radial_points, reciprocity_coefficient)
    fwd_pref = (-1.j) ** (orders[None, None, :]) * (r_max / n_radial_points) ** 3 * np.sqrt(2 / np.pi)
    inv_pref = (1.j) ** (orders[None, None, :]) * (q_max / n_radial_points) ** 3 * np.sqrt(2 / np.pi)
    w = np.moveaxis(weights, 0, 2)
    return {'forward': w * fwd_pref, 'inverse': w * inv_pref}


def l_of_lm(l_max):
    """Order l of every 'direct' index j = l(l+1)+m (the kernel uses floor(sqrt(j)), 712)."""
    return np.floor(np.sqrt(np.arange((l_max + 1) ** 2) + 0.5)).astype(int)


def apply_direct(w, coeff, trapz=False):
    """The OpenCL ``apply_weights`` kernel restated (hankel_transforms.py:702-731 / 671-700).

    w: (Np, Nk, L+1) complex; coeff: (..., Nq, nlm) complex -> (..., Nk, nlm)."""
    l_max = w.shape[-1] - 1
    src = coeff[..., 1:, :] if trapz else coeff
    out = np.empty(coeff.shape[:-2] + (w.shape[1], coeff.shape[-1]), dtype=complex)
    for l in range(l_max + 1):              # all j with l(j) = l share the weight matrix w[:, :, l]
        sl = slice(l * l, (l + 1) ** 2)
        out[..., sl] = np.einsum('pk,...pj->...kj', w[:, :, l], src[..., sl])
    return out


def apply_ml(w, coeff_list, trapz=False):
    """CPU path ``generate_spherical_ht`` (hankel_transforms.py:642-658), 'ml' lists."""
    l_max = w.shape[-1] - 1
    m_orders = np.concatenate((np.arange(l_max + 1, dtype=int), -np.arange(l_max, 0, -1, dtype=int)))
    out = []
    for m in m_orders:
        c = coeff_list[m]                 # list index m (negative wraps, as the reference does)
        src = c[1:, None, :l_max - abs(m) + 1] if trapz else c[:, None, :l_max - abs(m) + 1]
        out.append(np.sum(w[:, :, abs(m):] * src, axis=0))
    return tuple(out)
