"""CPU restatement (TEST INFRASTRUCTURE) of the 2-D (polar) variant of the hot path, SURVEY section 8 f-4: the circular harmonic
transforms, the polar Hankel transform with midpoint weights, the Fourier pair built from them and the 2-D branches of the reciprocal
projection.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.

Follows (file:line of /root/reference/xframe):
  library/mathLibrary.py:469-496                         circularHarmonicTransform_{complex,real}_{forward,inverse}
  projects/fxs/projectLibrary/harmonic_transforms.py:36-58   2-D HarmonicTransform: n_phi = 2 max_order + 1
  projects/fxs/projectLibrary/hankel_transforms.py:411-424   calc_polar_mid_weights
  projects/fxs/projectLibrary/hankel_transforms.py:300-362   assemble_weights_mid (2-D branch)
  projects/fxs/projectLibrary/hankel_transforms.py:602-640   generate_polar_ht (midpoint branch)
  projects/fxs/projectLibrary/fourier_transforms.py:49-88    generate_ft
  projects/fxs/projectLibrary/ft_grid_pairs.py:282-291, 325-336   midpoint radial grids, polar grid pair
  projects/fxs/projectLibrary/fxs_Projections.py:723-745, 803-826, 855-863   approximate_unknowns / mtip_projection / fixed_projection, dim == 2

**Parity status**: pinned by fixture G18 (tests/golden/polar2d_ops.npz, `make_golden.py polar2d`): outputs of the reference's own
functions on seeded inputs -- nothing third party is involved in the 2-D path (numpy FFT, scipy Bessel functions)."""
import numpy as np
from scipy.special import jv


# ---------------------------------------------------------------------------------------------- harmonic transforms (phi)
def harmonic_forward(x):
    """circularHarmonicTransform_complex_forward (mathLibrary.py:469-476): orders 0..M, -M..-1 along axis 1"""
    return np.fft.fft(np.array(x), axis=1) / x.shape[-1]


def harmonic_inverse(c):
    """circularHarmonicTransform_complex_inverse (478-483)"""
    return np.fft.ifft(np.array(c) * c.shape[-1], axis=1)


def real_harmonic_forward(x):
    """circularHarmonicTransform_real_forward (485-491): orders 0..M of the real part"""
    d = np.array(np.asarray(x).real)
    return np.fft.rfft(d) / d.shape[-1]


def real_harmonic_inverse(c, size):
    """circularHarmonicTransform_real_inverse (493-496)"""
    return np.fft.irfft(np.array(c) * size, size)


# ---------------------------------------------------------------------------------------------- polar Hankel transform, midpoint rule
def polar_mid_weights(orders, n_radial_points, reciprocity_coefficient):
    """calc_polar_mid_weights (hankel_transforms.py:411-424): weights[m, p, k] = (p + 1/2) J_m((k + 1/2)(p + 1/2) kappa / N)"""
    N = n_radial_points
    ps = np.arange(N) + 0.5
    ks = np.arange(N) + 0.5
    ms = np.asarray(orders)
    J = jv(ms[:, None, None] * np.ones((1, N, N)), ks[None, None, :] * ps[None, :, None] * reciprocity_coefficient / N)
    return ps[None, :, None] * J


def assemble_weights_mid(weights, orders, r_max, reciprocity_coefficient):
    """assemble_weights_mid, 2-D branch (300-362): (summed radial index p, new radial index k, order) for the orders
    0..M, -M..-1, with the prefactors (-+i)^m (cutoff / N)^2; w_{-m} = (-1)^m w_m"""
    orders = np.asarray(orders)
    N = weights.shape[-1]
    q_max = reciprocity_coefficient * N / r_max
    all_orders = np.concatenate((orders, -orders[:0:-1]))
    fpre = (-1.j) ** (all_orders[None, None, :]) * (r_max / N) ** 2
    ipre = (1.j) ** (all_orders[None, None, :]) * (q_max / N) ** 2
    w = np.concatenate((weights, (-1.0) ** orders[:0:-1, None, None] * weights[:0:-1]), axis=0)
    w = np.moveaxis(w, 0, 2)
    return {'forward': w * fpre, 'inverse': w * ipre}


def polar_raw_weights(orders, n_radial_points, reciprocity_coefficient, mode='midpoint'):
    """the workers behind generate_weightDict for dimensions = 2 (hankel_transforms.py): midpoint 411-424, trapz 335-347 (sum over
    p = 1..N-1), gauss 492-507 (Gauss-Legendre nodes and weights on p = k = x + 1), Zernike 133-176 (sum over p = 1..N-1) with the
    arguments the loader's call chain gives it -- generate_weightDict hands the reciprocity coefficient on as `expansion_limit` (26,
    52-62): the expansion stops at max(kappa, M) and the Bessel arguments use the default pi"""
    N = n_radial_points
    ms = np.asarray(orders)
    if mode == 'midpoint':
        return polar_mid_weights(orders, N, reciprocity_coefficient)
    if mode == 'trapz':
        ps, ks = np.arange(1, N), np.arange(N)
        return ps[None, :, None] * jv(ms[:, None, None] * np.ones((1, N - 1, N)), ks[None, None, :] * ps[None, :, None] * reciprocity_coefficient / N)
    if mode == 'gauss':
        from scipy.special import roots_legendre
        xi, wg = roots_legendre(N)
        ps = ks = xi + 1
        return ps[None, :, None] * jv(ms[:, None, None] * np.ones((1, N, N)), ks[None, None, :] * ps[None, :, None] * reciprocity_coefficient * N / 4) * wg[None, :, None]
    if mode == 'Zernike':
        from scipy.special import eval_jacobi
        lim, rc = max(reciprocity_coefficient, ms.max()), np.pi
        ps, ks = np.arange(1, N), np.arange(N)
        x = ps / N
        w = np.zeros((len(ms), N - 1, N))
        for i, m in enumerate(ms):
            sv = np.arange(m, lim + 1, 2)
            half = (sv - m) / 2
            Z = ((-1) ** half)[:, None] * (x ** m)[None, :] * eval_jacobi(half[:, None], m, 0, (1 - 2 * x ** 2)[None, :])   # mathLibrary.py:805-819, D = 2
            pref = (-1) ** half * (2 * sv + 2)
            Jk = jv((sv + 1)[:, None] * np.ones((1, N - 1)), (ks[1:] * rc)[None, :])
            w[i, :, 1:] = np.einsum('s,sp,sk->pk', pref, Z, Jk)
            if m == 0:
                w[i, :, 0] = rc
        c_kp = np.empty((N - 1, N))
        c_kp[:, 1:] = ps[:, None] / ks[None, 1:]
        c_kp[:, 0] = ps
        return w * c_kp[None]
    raise NotImplementedError(mode)


def assemble_weights_2d(weights, orders, r_max, reciprocity_coefficient, mode='midpoint'):
    """assemble_weights for dimensions = 2: midpoint 426-452 / trapz 349-375 (cutoff / N)^2, gauss 509-535 (cutoff / 2)^2, Zernike
    270-300 (cutoff / N)^2 / pi with the sign vector (-+i)^|m| for the negative orders too (its `all_orders` is built without the sign)"""
    orders = np.asarray(orders)
    N = weights.shape[-1]
    q_max = reciprocity_coefficient * N / r_max
    all_orders = np.concatenate((orders, orders[:0:-1] if mode == 'Zernike' else -orders[:0:-1]))
    if mode == 'gauss':
        fs, qs_ = (r_max / 2) ** 2, (q_max / 2) ** 2
    elif mode == 'Zernike':
        fs, qs_ = (r_max / N) ** 2 / np.pi, (q_max / N) ** 2 / np.pi
    else:
        fs, qs_ = (r_max / N) ** 2, (q_max / N) ** 2
    w = np.concatenate((weights, (-1.0) ** orders[:0:-1, None, None] * weights[:0:-1]), axis=0)
    w = np.moveaxis(w, 0, 2)
    return {'forward': w * ((-1.j) ** (all_orders[None, None, :]) * fs), 'inverse': w * ((1.j) ** (all_orders[None, None, :]) * qs_),
            'skip_first': mode in ('trapz', 'Zernike')}


def radial_grids_2d(max_q, n, kappa, mode):
    """the radial points of the polar grid pairs (ft_grid_pairs.py:312-349: the same radial functions as the spherical pairs)"""
    r_cut = kappa * n / max_q
    if mode == 'midpoint':
        dr, dq = r_cut / n, max_q / n
        return np.linspace(dr / 2, r_cut - dr / 2, num=n, endpoint=True), np.linspace(dq / 2, max_q - dq / 2, num=n, endpoint=True)
    if mode in ('trapz', 'Zernike'):
        return np.linspace(0, r_cut, n), np.linspace(0, max_q, n)
    if mode == 'gauss':
        from scipy.special import roots_legendre
        xs = roots_legendre(n)[0]
        return r_cut / 2 * xs + r_cut / 2, max_q / 2 * xs + max_q / 2
    raise NotImplementedError(mode)


def polar_ht(w, used_orders):
    """generate_polar_ht, midpoint branch (629-640): HT_m(f_m)(k) = sum_p f_m(p) w_pkm; orders outside `used_orders` are zeroed"""
    fw, iw = w['forward'], w['inverse']
    n_orders = (fw.shape[-1] + 1) // 2
    all_abs = np.concatenate((np.arange(n_orders), np.arange(n_orders)[:0:-1]))
    unused = ~np.isin(all_abs, used_orders)

    first = 1 if w.get('skip_first', False) else 0       # trapz / Zernike: the sums leave out shell 0 (ht_modes[:2], 619-628)

    def zht(c):
        out = np.sum(fw * c[first:, None, :], axis=0)
        out[:, unused] = 0
        return out

    def izht(c):
        out = np.sum(iw * c[first:, None, :], axis=0)
        out[:, unused] = 0
        return out
    return zht, izht, unused


class PolarFourierPair:
    """generate_ft for dimensions = 2 (fourier_transforms.py:49-88) on the midpoint grid pair (ft_grid_pairs.py:282-291, 325-336)"""

    def __init__(self, n_radial_points, max_order, max_q, reciprocity_coefficient=2.0, used_orders=None, weights_r_max=None, mode='midpoint'):
        self.N, self.M, self.kappa = n_radial_points, max_order, reciprocity_coefficient
        self.n_phi = 2 * max_order + 1                          # harmonic_transforms.py:44-47
        self.q_max = float(max_q)
        self.r_max = reciprocity_coefficient * self.N / self.q_max
        self.mode = mode
        self.rs, self.qs = radial_grids_2d(self.q_max, self.N, reciprocity_coefficient, mode)
        self.phis = np.arange(self.n_phi) / self.n_phi * 2 * np.pi
        self.orders = np.arange(max_order + 1)
        self.raw_weights = polar_raw_weights(self.orders, self.N, reciprocity_coefficient, mode)
        # the phasing loop hands generate_ft max(r_p), not the cutoff (reconstruct.py:329): `weights_r_max`
        self.weights = assemble_weights_2d(self.raw_weights, self.orders, self.r_max if weights_r_max is None else weights_r_max,
                                           reciprocity_coefficient, mode)
        self.zht, self.izht, self.unused = polar_ht(self.weights, self.orders if used_orders is None else np.asarray(used_orders))

    def ft(self, data):
        return harmonic_inverse(self.zht(harmonic_forward(data)))

    def ift(self, data):
        return harmonic_inverse(self.izht(harmonic_forward(data)))


# ---------------------------------------------------------------------------------------------- reciprocal projection, dim == 2
class ReciprocalProjection2D:
    """the 2-D branches of fxs_Projections.py's ReciprocalProjection on prepared inputs: `projection_matrices` (n_used, Nq) -- one
    vector per used order (after regridding / modify_projection_matrices) --, `used_orders` {order: id}, `radial_mask` (n_orders, Nq)"""

    def __init__(self, projection_matrices, used_orders, radial_mask, radial_points, n_harmonic_orders, number_of_particles=1.0):
        self.pm = np.asarray(projection_matrices, dtype=complex)
        self.used_orders = dict(used_orders)
        self.order_ids = tuple(self.used_orders.values())
        self.radial_mask = np.asarray(radial_mask, dtype=bool)
        self.radial_points = np.asarray(radial_points, dtype=float)
        self.n_orders = int(n_harmonic_orders)                  # (n_phi + 1) // 2 columns of the real harmonic transform
        self.number_of_particles = float(number_of_particles)

    def approximate_unknowns(self, I):
        """723-741: u_m = <I_m, v_m>_q / |.| with the radial weight q; 1 where the scalar product vanishes"""
        pmT = self.pm.T
        sp = np.sum(I[:, self.order_ids] * np.conjugate(pmT) * self.radial_points[:, None], axis=0)
        u = np.ones(len(self.order_ids), dtype=complex)
        nz = sp != 0
        u[nz] = sp[nz] / np.abs(sp[nz])
        return u

    def mtip_projection(self, I, unknowns):
        """803-826 + 855-863: I'_m(q) = v_m(q) u_m on the masked shells of the used orders, I'_0 = v_0 there, column 0 / sqrt(n_particles)"""
        order_array = np.array(list(self.used_orders.values()))
        new = np.array(I, dtype=complex)
        mask = np.zeros((len(self.radial_points), self.n_orders), dtype=bool)
        for o_id in order_array:
            mask[:, o_id] = self.radial_mask[o_id]
        rm2 = self.radial_mask[order_array].T
        pmT = self.pm.T
        new[mask] = (pmT * unknowns[None, :])[rm2]
        zero_id = self.used_orders.get(0, False)
        if not isinstance(zero_id, bool):
            zero_pos = int(np.argmax(np.array(tuple(self.used_orders.keys())) == 0))
            new[self.radial_mask[zero_id], zero_id] = pmT[self.radial_mask[zero_id], zero_pos]
            new[:, zero_id] /= np.sqrt(self.number_of_particles)
        return new
