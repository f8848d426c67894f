"""The 2-D (polar) variant of the hot path on the MI355X, operator level: the circular harmonic transforms
(``xframe/library/mathLibrary.py:469-496``), the polar Hankel pair with midpoint weights (``hankel_transforms.py:411-424, 300-362,
602-640``), the Fourier pair ``generate_ft`` builds from them (``fourier_transforms.py:49-88``) on the polar midpoint grid pair
(``ft_grid_pairs.py:282-291, 325-336``) and the 2-D reciprocal projection (``fxs_Projections.py:723-745, 803-826, 855-863``),
through the ``mtip2d_*`` entry points of ``include/mtip_hip.h`` (``csrc/k_polar2d.hip``); the radial rules besides midpoint (trapz,
gauss, Zernike: ``hankel_transforms.py:133-176, 335-375, 492-535``) are host weight tables for the same device contraction.  The 2-D
phasing loop on these operators is ``reconstruct2d.py``; no CPU fallback."""
import ctypes as C

import numpy as np
from scipy.special import jv

from . import _lib


def polar_mid_weights(orders, n_radial_points, reciprocity_coefficient):
    """calc_polar_mid_weights (hankel_transforms.py:411-424)"""
    N = n_radial_points
    ps = np.arange(N) + 0.5
    ks = np.arange(N) + 0.5
    ms = np.asarray(orders, dtype=float)
    return ps[None, :, None] * jv(ms[:, None, None], ks[None, None, :] * ps[None, :, None] * reciprocity_coefficient / N)


def assemble_weights_mid(weights, orders, r_max, reciprocity_coefficient):
    """assemble_weights_mid, 2-D branch (hankel_transforms.py:300-362)"""
    orders = np.asarray(orders)
    N = weights.shape[-1]
    q_max = reciprocity_coefficient * N / r_max
    all_orders = np.concatenate((orders, -orders[:0:-1]))
    w = np.concatenate((weights, (-1.0) ** orders[:0:-1, None, None] * weights[:0:-1]), axis=0)
    w = np.moveaxis(w, 0, 2)
    return (w * ((-1.j) ** all_orders[None, None, :] * (r_max / N) ** 2), w * ((1.j) ** all_orders[None, None, :] * (q_max / N) ** 2))


def polar_raw_weights(orders, n, kappa, mode='midpoint'):
    """raw polar Hankel weights [m, p, k] of a radial rule, as the reference's loader builds them for dimensions = 2: `midpoint`
    (hankel_transforms.py:411-424), `trapz` (335-347) and `Zernike` (133-176) sum over p = 1..N-1, `gauss` (492-507) over the
    Gauss-Legendre nodes.  Zernike: the loader passes the reciprocity coefficient on as `expansion_limit` (26, 52-62), so the expansion
    stops at max(kappa, M) and the Bessel arguments use pi (the same quirk as in 3-D, hostsetup._zernike_raw_weights)"""
    ms = np.asarray(orders, dtype=float)
    if mode == 'midpoint':
        return polar_mid_weights(orders, n, kappa)
    if mode == 'trapz':
        ps, ks = np.arange(1, n), np.arange(n)
        return ps[None, :, None] * jv(ms[:, None, None], ks[None, None, :] * ps[None, :, None] * kappa / n)
    if mode == 'gauss':
        from scipy.special import roots_legendre
        xi, wg = roots_legendre(n)
        node = xi + 1
        return node[None, :, None] * wg[None, :, None] * jv(ms[:, None, None], node[None, None, :] * node[None, :, None] * kappa * n / 4)
    if mode == 'Zernike':
        from scipy.special import eval_jacobi
        limit = max(kappa, ms.max())
        ps, ks = np.arange(1, n), np.arange(n)
        x = ps / n
        out = np.zeros((len(ms), n - 1, n))
        for i, m in enumerate(ms):
            terms = np.arange(m, limit + 1, 2)
            j = (terms - m) / 2
            radial = (x ** m)[None, :] * eval_jacobi(j[:, None], m, 0, (1 - 2 * x * x)[None, :])        # (-1)^j R^m_s(x), mathLibrary.py:805-819 with D = 2
            out[i, :, 1:] = np.einsum('s,sp,sk->pk', 2 * terms + 2, radial, jv((terms + 1)[:, None], (ks[1:] * np.pi)[None, :]))
            if m == 0:
                out[i, :, 0] = np.pi
        out[:, :, 1:] *= (ps[:, None] / ks[None, 1:])[None]
        out[:, :, 0] *= ps[None, :]
        return out
    raise NotImplementedError(f"2-D fourier_transform.type {mode!r}: 'midpoint', 'trapz', 'gauss', 'Zernike'")


def assemble_weights_2d(weights, orders, r_max, kappa, mode='midpoint'):
    """(forward, inverse) complex weights (p, k, all orders 0..M, -M..-1), N x N x (2M + 1): assemble_weights for dimensions = 2
    (midpoint 426-452, trapz 349-375, gauss 509-535, Zernike 270-300); the rules that leave out shell 0 of their input get a zero row
    for it, so that the device contraction is the same N x N sum for every rule"""
    orders = np.asarray(orders)
    n = weights.shape[-1]
    q_max = kappa * n / r_max
    signed = np.concatenate((orders, orders[:0:-1] if mode == 'Zernike' else -orders[:0:-1]))       # (Zernike's sign vector has no signs)
    fs, iv = {'gauss': ((r_max / 2) ** 2, (q_max / 2) ** 2),
              'Zernike': ((r_max / n) ** 2 / np.pi, (q_max / n) ** 2 / np.pi)}.get(mode, ((r_max / n) ** 2, (q_max / n) ** 2))
    w = np.concatenate((weights, (-1.0) ** orders[:0:-1, None, None] * weights[:0:-1]), axis=0)
    if w.shape[1] == n - 1:
        w = np.concatenate((np.zeros((w.shape[0], 1, n)), w), axis=1)
    w = np.moveaxis(w, 0, 2)
    return w * ((-1.j) ** signed[None, None, :] * fs), w * ((1.j) ** signed[None, None, :] * iv)


class Engine2D:
    """transforms (and, after ``set_projection``, the reciprocal projection) of the 2-D variant for batches of ``n_batch`` grids"""

    def __init__(self, n_radial_points, max_order, max_q, reciprocity_coefficient=2.0, n_batch=1, device=0, lib_path=None, used_orders=None,
                 weights_r_max=None, mode='midpoint'):
        self.lib = _lib.load(lib_path)
        self.N, self.M, self.B = int(n_radial_points), int(max_order), int(n_batch)
        self.n_phi = 2 * self.M + 1                              # harmonic_transforms.py:44-47
        self.kappa = float(reciprocity_coefficient)
        self.q_max = float(max_q)
        self.r_max = self.kappa * self.N / self.q_max            # mathLibrary.py:1169-1176
        from .hostsetup import radial_grids
        self.mode = mode
        self.rs, self.qs = radial_grids(self.q_max, self.N, self.kappa, mode)      # the polar pairs use the spherical pairs' radial functions (ft_grid_pairs.py:312-349)
        self.phis = np.arange(self.n_phi) / self.n_phi * 2 * np.pi
        self.shape = (self.N, self.n_phi)
        if self.lib.mtip_device_count() <= 0:
            raise _lib.MtipError('no HIP device visible: the 2-D operators need an MI355X (no CPU fallback)')
        self.ctx = self.lib.mtip2d_create(self.N, self.n_phi, self.B, int(device))
        if not self.ctx:
            raise _lib.MtipError('mtip2d_create failed (sizes: n_phi odd, 3..2047; a visible device)')
        orders = np.arange(self.M + 1)
        # (the phasing loop hands generate_ft max(r_p) instead of the cutoff, reconstruct.py:329: `weights_r_max`)
        fw, iw = assemble_weights_2d(polar_raw_weights(orders, self.N, self.kappa, mode), orders,
                                     self.r_max if weights_r_max is None else float(weights_r_max), self.kappa, mode)
        all_abs = np.concatenate((orders, orders[:0:-1]))
        unused = ~np.isin(all_abs, orders if used_orders is None else np.asarray(used_orders))     # hankel_transforms.py:611-613
        self._ck(self.lib.mtip2d_set_hankel_weights(self.ctx, _lib.ptr(_lib.as_c128(fw)), _lib.ptr(_lib.as_c128(iw)), _lib.ptr(_lib.as_u8(unused))))
        self.n_used = 0

    def _ck(self, rc):
        if rc != 0:
            raise _lib.MtipError(f'libmtip_hip (2-D) error {rc}: ' + self.lib.mtip2d_last_error(self.ctx).decode())

    def close(self):
        if getattr(self, 'ctx', None):
            self.lib.mtip2d_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _grid(self, a, last=None):
        a = _lib.as_c128(a)
        shape = (self.N, self.n_phi if last is None else last)
        if a.shape == shape:
            a = np.broadcast_to(a, (self.B,) + shape)
        assert a.shape == (self.B,) + shape, (a.shape, shape)
        return np.ascontiguousarray(a)

    def harmonic(self, grid, inverse=False):
        g = self._grid(grid)
        out = np.empty_like(g)
        self._ck(self.lib.mtip2d_op_harmonic(self.ctx, _lib.ptr(g), _lib.ptr(out), int(inverse)))
        return out

    def real_harmonic_forward(self, grid):
        g = self._grid(grid)
        out = np.empty((self.B, self.N, self.M + 1), complex)
        self._ck(self.lib.mtip2d_op_real_harmonic_forward(self.ctx, _lib.ptr(g), _lib.ptr(out)))
        return out

    def real_harmonic_inverse(self, coeff):
        c = self._grid(coeff, self.M + 1)
        out = np.empty((self.B, self.N, self.n_phi))
        self._ck(self.lib.mtip2d_op_real_harmonic_inverse(self.ctx, _lib.ptr(c), _lib.ptr(out)))
        return out

    def hankel(self, coeff, inverse=False):
        c = self._grid(coeff)
        out = np.empty_like(c)
        self._ck(self.lib.mtip2d_op_hankel(self.ctx, _lib.ptr(c), _lib.ptr(out), int(inverse)))
        return out

    def fourier_transform(self, grid, inverse=False):
        g = self._grid(grid)
        out = np.empty_like(g)
        self._ck(self.lib.mtip2d_op_fourier_transform(self.ctx, _lib.ptr(g), _lib.ptr(out), int(inverse)))
        return out

    def set_projection(self, projection_vectors, used_orders, radial_mask, number_of_particles=1.0):
        """projection_vectors (n_used, Nq): one per used order; used_orders {order: id} (ascending ids, order 0 included);
        radial_mask (n_orders, Nq) indexed by id as upstream"""
        ids = np.ascontiguousarray(list(used_orders.values()), dtype=np.int32)
        pm = _lib.as_c128(projection_vectors)
        assert pm.shape == (len(ids), self.N)
        rm = _lib.as_u8(np.asarray(radial_mask, dtype=bool)[ids])
        self._ck(self.lib.mtip2d_set_projection(self.ctx, len(ids), _lib.ptr(ids), _lib.ptr(pm), _lib.ptr(rm), _lib.ptr(_lib.as_f64(self.qs)),
                                                float(number_of_particles)))
        self.n_used = len(ids)

    def set_so_freedom(self, position):
        """SO_freedom (dimensions == 2): the unknown at this position among the used orders is 1 in every projection; None: off"""
        self._ck(self.lib.mtip2d_set_so_freedom(self.ctx, -1 if position is None else int(position)))

    def project(self, I):
        """approximate_unknowns + mtip_projection + number-of-particles rule: (projected coefficients, unknowns)"""
        c = self._grid(I, self.M + 1)
        out = np.empty_like(c)
        unk = np.empty((self.B, self.n_used), complex)
        self._ck(self.lib.mtip2d_op_project(self.ctx, _lib.ptr(c), _lib.ptr(out), _lib.ptr(unk)))
        return out, unk

    # ---- operators of the phasing loop
    def set_real_constraints(self, flags, lo, hi, imag_thr, hio_flags):
        self._ck(self.lib.mtip2d_set_real_constraints(self.ctx, int(flags), float(lo), float(hi), float(imag_thr), int(hio_flags)))

    def set_error_weights(self, weights):
        w = _lib.as_f64(weights)
        assert w.shape == (self.N, self.n_phi)
        self._ck(self.lib.mtip2d_set_error_weights(self.ctx, _lib.ptr(w)))

    def step(self, method, ft_stab, beta, rho, support, fixed_intensity=None, want_inputs=False):
        """one HIO / ER (or HIO_non_FXS / ER_non_FXS: `fixed_intensity` (B, Nq, n_phi)) step for the batch:
        (F_new, rho_new, errors (B,), unknowns (B, n_used) or None); with want_inputs also (F = FT(rho), I_m of |F|^2 or None),
        the arguments of the reciprocal error metrics"""
        r = self._grid(rho)
        sup = np.ascontiguousarray(np.broadcast_to(np.asarray(support, dtype=np.uint8), (self.B,) + self.shape))
        F_new, rho_new = np.empty_like(r), np.empty_like(r)
        err = np.empty(self.B)
        mid = {'HIO': 0, 'ER': 1, 'HIO_non_FXS': 2, 'ER_non_FXS': 3}[method]
        fxs = mid < 2
        unk = np.empty((self.B, self.n_used), complex) if fxs else None
        fixed = None
        if not fxs:
            fixed = np.ascontiguousarray(np.broadcast_to(np.asarray(fixed_intensity, dtype=np.float64), (self.B,) + self.shape))
        F_in = np.empty_like(r) if want_inputs else None
        I_in = np.empty((self.B, self.N, self.M + 1), complex) if (want_inputs and fxs) else None
        self._ck(self.lib.mtip2d_op_step_ex(self.ctx, mid, int(bool(ft_stab)), float(beta), _lib.ptr(r), _lib.ptr(sup), _lib.ptr(fixed),
                                            _lib.ptr(F_new), _lib.ptr(rho_new), _lib.ptr(err), _lib.ptr(unk), _lib.ptr(F_in), _lib.ptr(I_in)))
        if want_inputs:
            return F_new, rho_new, err, unk, F_in, I_in
        return F_new, rho_new, err, unk

    def shrinkwrap(self, rho, sigma, threshold):
        r = self._grid(rho)
        mask = np.empty((self.B,) + self.shape, np.uint8)
        self._ck(self.lib.mtip2d_op_shrinkwrap(self.ctx, _lib.ptr(r), float(sigma), float(threshold), _lib.ptr(mask)))
        return mask.astype(bool)
