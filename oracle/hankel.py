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
from scipy.special import spherical_jn


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
