"""One-off host-side setup of the MTIP engine (numpy/scipy, exactly where the reference does it on the host).

Each helper mirrors a reference routine (file:line relative to the reference checkout, under xframe/):
  angular grid        externalLibraries/shtns_plugin.py:94-101, 121-133
  radial grids        projects/fxs/projectLibrary/ft_grid_pairs.py:282-291 (midpoint), 274-281 (trapz, Zernike), 293-300 (gauss)
  Hankel weights      projects/fxs/projectLibrary/hankel_transforms.py:399-410, 322-333, 426-452
  projection matrices projects/fxs/projectLibrary/fxs_Projections.py:471-537, 578-714
  initial support     fxs_Projections.py:133-155
  error weights       library/mathLibrary.py:1223-1235 + projects/fxs/projectLibrary/fxs_IO_methods.py:97-151, 287-300
  ramps               library/mathLibrary.py:1033-1129; reconstruct.py:1212-1258
  density guess       projects/fxs/reconstruct.py:1115-1174; library/mathLibrary.py:1456-1466
None of this is on the per-iteration path; the iteration itself runs in libmtip_hip.so.
"""
import numpy as np
from scipy.interpolate import griddata
from scipy.special import roots_legendre, spherical_jn


# ---------------------------------------------------------------------------------------------- grids
def angular_grid_size(l_max, n_theta=0, n_phi=0, anti_aliazing_degree=2):
    f_phi = 2 ** (int(np.log2((anti_aliazing_degree + 1) * l_max)) + 1) if l_max > 0 else 4
    f_theta = f_phi // 2
    if (not isinstance(n_theta, (int, np.integer))) or isinstance(n_theta, bool) or n_theta == 0:
        n_theta = f_theta
    if (not isinstance(n_phi, (int, np.integer))) or isinstance(n_phi, bool) or n_phi == 0:
        n_phi = f_phi
    return int(n_theta), int(n_phi)


def gauss_grid(n_theta, n_phi):
    x, w = roots_legendre(n_theta)
    cos_theta = x[::-1].copy()          # north -> south
    weights = w[::-1].copy()
    return cos_theta, weights, np.arccos(cos_theta), 2 * np.pi * np.arange(n_phi) / n_phi


def radial_grids(max_q, n, kappa, mode='midpoint'):
    r_cut = kappa * n / max_q
    if mode == 'midpoint':
        dr, dq = r_cut / n, max_q / n
        return (np.linspace(dr / 2, r_cut - dr / 2, num=n, endpoint=True),
                np.linspace(dq / 2, max_q - dq / 2, num=n, endpoint=True))
    if mode in ('trapz', 'Zernike'):                    # ft_grid_pairs.py:274-281, selected for both at 545
        return np.linspace(0, r_cut, n), np.linspace(0, max_q, n)
    if mode == 'gauss':                                 # ft_grid_pairs.py:293-300 (dim == 3: 551-552): Gauss-Legendre nodes on [0, R], [0, Q]
        xs = roots_legendre(n)[0]
        return r_cut / 2 * xs + r_cut / 2, max_q / 2 * xs + max_q / 2
    raise NotImplementedError(f'fourier_transform.type {mode!r} is not supported (midpoint, trapz, gauss, Zernike)')


_HANKEL_W = {}


def hankel_raw_weights(l_max, n, kappa, mode='midpoint'):
    """real (L+1, Np, N) array indexed [l, p, k] (summation index p, output index k); read only (the engines of one worker
    share it: 0.06 s of Bessel evaluations at 128 x L32)."""
    key = (int(l_max), int(n), float(kappa), mode)
    if key not in _HANKEL_W:
        if len(_HANKEL_W) >= 4:
            _HANKEL_W.pop(next(iter(_HANKEL_W)))
        w = _hankel_raw_weights(l_max, n, kappa, mode)
        w.setflags(write=False)
        _HANKEL_W[key] = w
    return _HANKEL_W[key]


def _hankel_raw_weights(l_max, n, kappa, mode='midpoint'):
    ls = np.arange(l_max + 1)
    if mode == 'midpoint':
        ps = np.arange(n) + 0.5
        ks = np.arange(n) + 0.5
    elif mode == 'trapz':
        ps = np.arange(1, n)
        ks = np.arange(n)
    elif mode == 'gauss':
        # calc_spherical_gauss_weights, hankel_transforms.py:477-490: Gauss-Legendre nodes and weights, p = k = x + 1
        xi, wg = roots_legendre(n)
        ps = ks = xi + 1
        arg = ks[None, :] * ps[:, None] * kappa * n / 4
        return np.ascontiguousarray(ps[None, :, None] ** 2 * spherical_jn(ls[:, None, None], arg[None, :, :]) * wg[None, :, None])
    elif mode == 'Zernike':
        return _zernike_raw_weights(l_max, n, kappa)
    else:
        raise NotImplementedError(f'fourier_transform.type {mode!r} is not supported (midpoint, trapz, gauss, Zernike)')
    arg = ks[None, :] * ps[:, None] * kappa / n
    return np.ascontiguousarray(ps[None, :, None] ** 2 * spherical_jn(ls[:, None, None], arg[None, :, :]))


def _zernike_raw_weights(l_max, n, kappa):
    """calc_spherical_zernike_weights (hankel_transforms.py:88-131) with the arguments the reference's loader gives it: its
    generate_weightDict hands the reciprocity coefficient on as the third POSITIONAL argument of generate_weightDict_zernike (26),
    which is `expansion_limit` (52) -- so the Zernike expansion stops at max(kappa, l_max) (62) and the Bessel arguments use that
    function's default reciprocity coefficient pi, whatever the settings say; only the assembly (hankel_scales) sees kappa.
    w[l,p,k] = (p^2 / k) sum_{s = l, l+2, ..} (-1)^((s-l)/2) (2s+3) R^l_s(p/n) j_{s+1}(pi k) for p = 1..n-1, k = 1..n-1, with the
    3-D Zernike radial polynomials R^l_s(x) = (-1)^((s-l)/2) x^l P^{(l+1/2, 0)}_{(s-l)/2}(1 - 2 x^2) (mathLibrary.py:805-819); column
    k = 0: p^2 pi for l = 0, else 0."""
    from scipy.special import eval_jacobi
    lim = max(kappa, l_max)
    rc = np.pi
    ps = np.arange(1, n)
    ks = np.arange(n)
    x = ps / n
    w = np.zeros((l_max + 1, n - 1, n))
    for l in range(l_max + 1):
        s = np.arange(l, lim + 1, 2)
        half = (s - l) / 2
        R = ((-1) ** half)[:, None] * (x ** l)[None, :] * eval_jacobi(half[:, None], l + 0.5, 0, (1 - 2 * x ** 2)[None, :])
        pref = (-1) ** half * (2 * s + 3)
        jp = spherical_jn((s + 1)[:, None], (ks[1:] * rc)[None, :])
        w[l, :, 1:] = np.einsum('s,sp,sk->pk', pref, R, jp)
        if l == 0:
            w[0, :, 0] = rc
    c_kp = np.empty((n - 1, n))
    c_kp[:, 1:] = np.square(ps)[:, None] / ks[None, 1:]
    c_kp[:, 0] = np.square(ps)
    return np.ascontiguousarray(w * c_kp[None, :, :])


def hankel_skips_first_shell(mode):
    """the sums of the trapz and Zernike rules leave out shell 0 of their input (hankel_transforms.py:647-652, 671-700: ht_modes[:2])"""
    return mode in ('trapz', 'Zernike')


def hankel_scales(r_max, n, kappa, mode='midpoint'):
    """(forward, inverse) real prefactors; the (-/+ i)^l phases are applied by the kernel.  midpoint / trapz:
    hankel_transforms.py:426-452, 349-375; gauss 509-535; Zernike 270-300."""
    q_max = kappa * n / r_max
    if mode == 'gauss':
        c = np.sqrt(2 / np.pi)
        return (r_max / 2) ** 3 * c, (q_max / 2) ** 3 * c
    c = np.sqrt(2 / np.pi ** 3) if mode == 'Zernike' else np.sqrt(2 / np.pi)
    return (r_max / n) ** 3 * c, (q_max / n) ** 3 * c


# ---------------------------------------------------------------------------------------------- ramps
def _is_number(v):
    return np.issubdtype(np.array(v).dtype, np.number)


class ExponentialRamp:
    def __init__(self, start, stop, exponent, stop_argument=1):
        self.start, self.stop = start, stop
        exponent = -abs(exponent) if stop < start else abs(exponent)
        self.exponent = exponent
        self.A = (start - stop) / (1 - np.exp(exponent * stop_argument))
        self.B = start - self.A

    def eval(self, x):
        v = self.A * np.exp(x * self.exponent) + self.B
        return np.maximum(v, self.stop) if self.start > self.stop else np.minimum(v, self.stop)

    __call__ = eval


class LinearRamp:
    def __init__(self, start, stop=False, slope=False, default_start=False, default_stop=False):
        self.start = tuple(start) if isinstance(start, (list, tuple)) else (start, 0)
        self.undefined = False
        if not _is_number(self.start[0]):
            if isinstance(default_start, bool) and default_start is False:
                self.undefined = True
            else:
                self.start = (default_start, 0)
        stop_ok = False
        if isinstance(stop, (list, tuple)):
            stop = list(stop)
            if not _is_number(stop[0]) and _is_number(default_stop):
                stop[0] = default_stop
            if _is_number(stop[0]) and _is_number(stop[1]) and stop[1] >= self.start[1]:
                stop_ok = True
        slope_ok = not isinstance(slope, bool)
        self.C = np.nan
        if self.undefined:
            return
        if not stop_ok and not slope_ok:
            self.A = 0
        elif stop_ok:
            self.C = stop[0]
            self.A = 0 if (stop[1] - self.start[1]) == 0 else (stop[0] - self.start[0]) / (stop[1] - self.start[1])
            if slope_ok:
                self.A = slope
        elif slope == 0:
            self.A = slope
        else:
            self.C = np.sign(slope) * np.inf
            self.A = slope
        self.B = self.start[0] - self.A * self.start[1]

    def eval(self, x):
        if self.undefined:
            return np.nan
        v = self.A * x + self.B
        if self.A < 0:
            v = max(v, self.C)
        elif self.A > 0:
            v = min(v, self.C)
        return v

    __call__ = eval


# ---------------------------------------------------------------------------------------------- reciprocal projection data
def _regrid(values, old, new, interpolation):
    """the reference's 1-D regridding call (fxs_Projections.py:639-676); `values` may hold several columns (len(old), k): the
    spline of every column is the one a call per column gives, in one scipy call instead of k"""
    out = griddata(old[:, None], values, new[:, None], method=interpolation, fill_value=0.0, rescale=False)
    return out.reshape((len(new),) + np.shape(values)[1:])


def reciprocal_radial_mask(qs, q_d, max_order, mopt, data):
    """generate_radial_mask (fxs_Projections.py:578-629): (n_orders, Nq) bool, data range AND the q_mask option"""
    qs = np.asarray(qs, dtype=float)
    n = len(qs)
    data_mask = (qs >= q_d.min()) & (qs <= q_d.max())
    mask = np.ones((max_order + 1, n), dtype=bool)
    
    if isinstance(mopt, dict):
        mtype = mopt['type']
        if mtype == 'manual' and mopt['manual']['type'] == 'region':
            lo, hi = mopt['manual']['region']
            lo_set = not (isinstance(lo, bool) and lo is False)
            hi_set = not (isinstance(hi, bool) and hi is False)
            if not lo_set and hi_set:
                mask[:] = (qs < hi)[None, :]
            elif lo_set and not hi_set:
                mask[:] = (qs >= lo)[None, :]
            elif lo_set and hi_set:
                mask[:] = ((qs >= lo) & (qs < hi))[None, :]
        elif mtype == 'manual' and mopt['manual']['type'] == 'order_dependent_line':
            # fxs_Projections.py:619-624, mathLibrary.py:1131-1137: side of the line through two (order, q) points
            p1, p2 = np.asarray(mopt['manual']['order_dependent_line'], dtype=float)
            rot = np.array([p2[1] - p1[1], -(p2[0] - p1[0])])
            oq = np.stack(np.meshgrid(np.arange(max_order + 1, dtype=float), qs, indexing='ij'), axis=-1)
            mask = (-1 * np.sum((oq - p1) * rot[None, None, :], axis=-1)) >= 0
        elif mtype == 'from_projection_matrices':
            # 592-597: per order the open q interval covered by the data matrices
            lims = data.get('data_projection_matrices_q_id_limits', False)
            if isinstance(lims, dict):
                lims = lims['I1I1']
            if isinstance(lims, bool):
                raise ValueError("q_mask 'from_projection_matrices' needs data_projection_matrices_q_id_limits")
            mask = np.zeros((max_order + 1, n), dtype=bool)
            for row, lim in zip(mask, lims):
                row[:] = (qs > q_d[int(lim[0])]) & (qs < q_d[int(lim[1]) - 1])
        elif mtype != 'none':
            raise NotImplementedError(f'q_mask type {mtype!r}')

    return mask & data_mask[None, :]


class ReciprocalSetup:
    """Everything ``ReciprocalProjection.__init__`` prepares on the host (fxs_Projections.py:471-537)."""

    def __init__(self, qs, data, max_order, opt):
        # variants of the reference that are not on the accelerated path must not be ignored silently
        if opt.get('number_of_particles', {}).get('estimate', False):
            raise NotImplementedError('projections.reciprocal.number_of_particles.estimate = True (marked "NOT WORKING" upstream, '
                                      'default_0.01.yaml:135-137)')
        q_d = np.asarray(data['data_radial_points'], dtype=float)
        aint = data['average_intensity']
        aint = np.asarray(getattr(aint, 'data', aint), dtype=float)
        self.qs = np.asarray(qs, dtype=float)
        n = len(self.qs)
        self.integrated_intensity = (q_d[1] - q_d[0]) * np.sum(aint * q_d ** 2) * 2 * np.sqrt(np.pi)
        orders = np.arange(max_order + 1)
        used_ids = np.asarray(opt.get('used_orders', opt['used_order_ids']))
        self.used_orders = {int(o): int(i) for o, i in zip(orders, used_ids)}
        self.number_of_particles = float(opt['number_of_particles']['initial'])
        interp = opt['regrid']['interpolation']
        dpm = data['data_projection_matrices']
        self.average_intensity = _regrid(aint, q_d, self.qs, interp)
        pm = {}
        for o, oid in self.used_orders.items():
            m = np.asarray(dpm[oid])
            if m.ndim < 2:
                m = m[:, None]
            pm[oid] = _regrid(m, q_d, self.qs, interp).astype(complex)
        self.full_projection_matrices = [pm.get(l, np.zeros((n, min(n, 2 * l + 1)), complex)) for l in range(max_order + 1)]
        # modify_projection_matrices, 679-714
        proj = {oid: m.copy() for oid, m in pm.items()}
        if opt.get('odd_orders_to_0', False):
            for o, oid in self.used_orders.items():
                if o % 2 == 1:
                    proj[oid][:] = 0
        if opt.get('use_averaged_intensity', False) and 0 in self.used_orders:
            proj[self.used_orders[0]] = (self.average_intensity[:, None] * 2 * np.sqrt(np.pi)).astype(complex)
        for oid in proj:
            proj[oid] = proj[oid] * 2
        self.projection_matrices = proj                     # keyed by order id
        # SO_freedom (fxs_Projections.py:493, 768-780): element [4, 2] of the unknowns of the best ranked even order is made real
        self.so_order = -1
        if opt.get('SO_freedom', {}).get('use', False):
            if sorted(self.used_orders.values()) != list(range(max_order + 1)):
                # upstream walks the coefficient list from order 0 next to the per-used-order lists (771): only defined for all orders
                raise NotImplementedError('SO_freedom with a subset of the orders (fxs_Projections.py:771 pairs the lists by position)')
            ids = rank_projection_matrices_3d([proj[i] for i in range(max_order + 1)], orders, self.qs, opt['SO_freedom']['radial_high_pass'])
            self.so_order = int(ids[0])
            if proj[self.so_order].shape[1] < 5:
                raise ValueError('SO_freedom: the chosen order has fewer than 5 unknown rows (upstream indexes row 4)')
        self.radial_mask = reciprocal_radial_mask(self.qs, q_d, max_order, opt.get('q_mask', None), data)
        self.max_order = max_order


def sph_plm(l, m, x):
    """gsl_sf_legendre_sphPlm(l, m, x) = sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!) P_l^m(x) = Y_l^m(arccos x, 0) (the reference gets it from
    pygsl, gsl_plugin.py:8-69; restated from the published definition)"""
    from scipy.special import sph_harm_y
    return sph_harm_y(np.asarray(l), np.asarray(m), np.arccos(np.clip(x, -1.0, 1.0)), 0.0).real


def invariant_metric_tables(which, qs, projection_matrices, radial_mask, xray_wavelength, C_order=None):
    """Constant tables of the non-default reciprocal metrics for mtip_set_invariant_metrics (fxs_IO_methods.py:507-550 fqc_error,
    587-627 II_error, 651-683 ccd_diff; Legendre products fxs_invariant_tools.py:23-33, 48-58).  projection_matrices: list over all
    orders 0..L of the modified V_l (the reference's rp.projection_matrices; its reference invariants are V_l V_l^+, 631-637)."""
    qs = np.asarray(qs, dtype=float)
    N, L = len(qs), len(projection_matrices) - 1
    ref = np.array([np.asarray(p) @ np.asarray(p).conj().T for p in projection_matrices])
    rm = np.asarray(radial_mask, dtype=bool)
    zero = ~(rm[:, :, None] & rm[:, None, :])                     # entries of B_l outside the invariant mask
    ref[zero] = 0
    out = {'zero_mask': zero.astype(np.uint8)}
    orders = np.arange(L + 1)
    x = qs * xray_wavelength / (4 * np.pi)                        # cos(theta) of ewald_sphere_theta_pi (physicsLibrary.py:94-95)
    if 'II_error' in which:
        out['II_reference'] = np.sum(ref[1:], axis=0)
        out['qq'] = (qs[:, None] * qs[None, :]) ** 2
    if 'ccd_diff' in which:
        if C_order is None:
            raise KeyError("main_loop.error.methods.reciprocal.ccd_diff.C_order is required (fxs_IO_methods.py:637)")
        pl = sph_plm(orders[None, :], np.full((1, L + 1), int(C_order)), x[:, None])                  # (q, l); 0 for l < C_order
        pl = np.where(orders[None, :] >= int(C_order), pl, 0.0)
        T = np.moveaxis(pl[None, :, :] * pl[:, None, :] / (2 * orders + 1)[None, None, :], -1, 0).copy()    # (l, q, q')
        T[0] = 0
        T[orders < int(C_order)] = 0
        T[np.isnan(T)] = 0
        refC = np.sum(ref * T, axis=0)
        norm = np.sum(refC * refC.conj())
        if norm == 0:
            raise ValueError('ccd_diff: the reference C_m vanishes (fxs_IO_methods.py:669)')
        out['ccd_weights'], out['ccd_reference'], out['ccd_norm'] = T, refC, float(np.real(norm))
    if 'fqc_error' in which:
        qm = np.zeros((N, L + 1, L + 1))                          # (q, m, l)
        for m in range(L + 1):
            ls = np.arange(m, L + 1)
            qm[:, m, ls] = sph_plm(ls[None, :], np.full((1, len(ls)), m), x[:, None])
        P = np.moveaxis(qm[None, :] * qm[:, None] / (2 * orders + 1)[None, None, None, :], -1, 0)     # (l, q, q', m)
        ccn = np.sum(ref[1:, ..., None] * P[1:], axis=0)
        out['fqc_reference_average'] = (ccn[..., 0] * ccn[..., 0]).real + 2 * np.sum(ccn[..., 1:] * ccn[..., 1:].conj(), axis=-1).real
        rw = np.zeros((L + 1, N, N))
        rw[1:] = (P[1:, ..., 0] * ccn[None, ..., 0]).real + 2 * np.sum(P[1:, ..., 1:] * ccn[None, ..., 1:].conj(), axis=-1).real
        out['fqc_P'], out['fqc_reference_weights'] = np.ascontiguousarray(P), rw
    return out


def rank_projection_matrices_3d(projection_matrices, orders, radial_points, radial_high_pass=0.15):
    """fxs_invariant_tools.py:1467-1486 (RadialIntegrator(., 2), mathLibrary.py:1270-1294): ids of the even non-zero orders, ranked by
    the radial L2 norm of B_l = Re(V_l V_l^+) beyond the high-pass radius, largest first"""
    orders = np.asarray(orders)
    hp = int((len(radial_points) - 1) * radial_high_pass)
    r = np.asarray(radial_points, dtype=float)[hp:]
    ids = np.nonzero((orders % 2 == 0) & (orders != 0))[0]
    w = np.zeros(len(r))                                    # trapezoid weights times the radial weight r^(2-1)
    w[:-1] += np.diff(r) / 2
    w[1:] += np.diff(r) / 2
    w = w * r
    metrics = []
    for i in ids:
        pm = np.asarray(projection_matrices[i])
        bl = (pm @ pm.conj().T).real[hp:, hp:]
        inner = (bl * bl) @ w
        metrics.append(float((inner * inner) @ w))
    return ids[np.argsort(np.array(metrics))[::-1]]


def initial_support(rs, shape, opt, particle_radius, auto_correlation=None):
    sup = opt['support']['initial_support']
    r = np.broadcast_to(np.asarray(rs)[:, None, None], shape)
    if sup['type'] == 'max_radius':
        return np.ascontiguousarray(r < sup['max_radius'])
    if sup['type'] == 'auto_correlation':
        m = auto_correlation >= sup['auto_correlation']['threshold'] * np.max(auto_correlation)
        m = np.array(m)
        m[r > particle_radius] = False
        return m
    raise AssertionError(f"Initial support type {sup['type']!r} is not known.")


def real_constraint_flags(opt, considered):
    """-> (flags, lo, hi, imag_thr, hio_flags) for mtip_set_real_constraints."""
    SUP, LO, HI, IM = 1, 2, 4, 8
    flags, lo, hi, thr = 0, 0.0, 0.0, 0.0
    bits = {}
    for key in opt['apply']:
        if key == 'support':
            flags |= SUP
            bits[key] = SUP
        elif key == 'value_threshold':
            t = opt['value_threshold'].get('threshold', 0.0)
            num = [isinstance(v, (float, int)) and not isinstance(v, bool) for v in t]
            b = 0
            if num[0]:
                b |= LO
                lo = float(t[0])
            if num[1]:
                b |= HI
                hi = float(t[1])
            flags |= b
            bits[key] = b
        elif key == 'limit_imag':
            flags |= IM
            thr = float(opt['limit_imag'].get('threshold', 0.0))
            bits[key] = IM
        # unknown names (e.g. 'assert_real') have no generator in the reference and are ignored (113-118)
    if not isinstance(considered, (list, tuple)) or len(considered) == 0 or list(considered) == ['all']:
        hio = flags
    else:
        hio = 0
        for name in considered:
            hio |= flags if name == 'all' else bits.get(name, 0)
    return flags, lo, hi, thr, hio


def error_weights(rs, n_theta, shape, inside_initial_support, cache_aware=True, l2_cache_kb=512):
    """Weights of the l2_projection_diff metric such that E = sum m wr wt |w-P|^2 / sum m wr wt |w|^2.

    Returns (wr, wt, use_initial_support_mask).  Reproduces which variant the reference really runs:
    the cache-aware routine drops the initial-support mask when the grid fits into L2_cache/2 and then
    ``square[~True] = 0`` zeroes radial shell N-2 (fxs_IO_methods.py:113-116, 131-151, 203-205)."""
    rs = np.asarray(rs, dtype=float)
    n = len(rs)
    t = np.zeros(n)
    d = np.diff(rs)
    t[:-1] += d / 2
    t[1:] += d / 2
    wr = t * rs ** 2
    wt = roots_legendre(n_theta)[1][::-1] * np.pi / n_theta
    use_mask = bool(inside_initial_support)
    units = (l2_cache_kb / 2) * 1024 / 16
    fits = not (np.prod(shape) > units)
    if cache_aware and fits:
        use_mask = False
    if not use_mask:
        wr = wr.copy()
        wr[n - 2] = 0.0                # the ~True == -2 quirk
    return wr, wt, use_mask


def bump_density(rs, shape, radius, slope, snr, rng, integrated_intensity, wr_plain, wt):
    """reconstruct.py:1155-1174 with a seeded generator (the reference seeds from os.urandom)."""
    amp = 1 + 1 / snr * rng.random(shape)
    r = np.broadcast_to(np.asarray(rs)[:, None, None], shape)
    inside = (r > -radius) & (r < radius)
    env = np.zeros(shape)
    env[inside] = np.exp(-slope * radius ** 2 / (radius ** 2 - r[inside] ** 2))
    density = amp * env
    total_sq = np.einsum('q,t,qtp->', wr_plain, wt, density * density)
    return (density * np.sqrt(integrated_intensity / total_sq)).astype(complex)


def ball_density(rs, shape, radius, snr, rng, integrated_intensity, wr_plain, wt):
    """reconstruct.py:1136-1153 ('ball': get_disk_function, mathLibrary.py:124-167): random amplitude for r < radius."""
    r = np.broadcast_to(np.asarray(rs)[:, None, None], shape)
    inside = r < radius
    density = np.zeros(shape)
    density[inside] = 1 + 1 / snr * rng.random(int(inside.sum()))
    total_sq = np.einsum('q,t,qtp->', wr_plain, wt, density * density)
    return (density * np.sqrt(integrated_intensity / total_sq)).astype(complex)


def autocorrelation_density(autocorrelation, rs, shape, particle_radius, snr, rng, integrated_intensity, wr_plain, wt):
    """reconstruct.py:1186-1203: low-resolution autocorrelation (already transformed) -> guess."""
    ac = np.array(autocorrelation, dtype=float)
    ac[ac < 0] = 0
    density = ac * (1 + 1 / snr * rng.random(shape))
    density[density < 0] = 0
    r = np.broadcast_to(np.asarray(rs)[:, None, None], shape)
    inside = (r > -particle_radius) & (r < particle_radius)
    env = np.zeros(shape)
    env[inside] = np.exp(-0.1 * particle_radius ** 2 / (particle_radius ** 2 - r[inside] ** 2))
    density = density * env
    total_sq = np.einsum('q,t,qtp->', wr_plain, wt, density * density)
    return (density * np.sqrt(integrated_intensity / total_sq)).astype(complex)


def integrator_weights(rs, n_theta):
    """plain SphericalIntegrator weights: int f = sum wr[q] wt[t] sum_phi f."""
    rs = np.asarray(rs, dtype=float)
    t = np.zeros(len(rs))
    d = np.diff(rs)
    t[:-1] += d / 2
    t[1:] += d / 2
    return t * rs ** 2, roots_legendre(n_theta)[1][::-1] * np.pi / n_theta


# ----------------------------------------------------------------------------- output modifier 'shift_to_center'
def spherical_to_cartesian(grid):
    """mathLibrary.py:673-698, 3-D."""
    g = np.asarray(grid, dtype=float)
    xy = g[..., 0] * np.sin(g[..., 1])
    return np.stack((np.cos(g[..., 2]) * xy, np.sin(g[..., 2]) * xy, g[..., 0] * np.cos(g[..., 1])), axis=-1)


def cartesian_to_spherical(v):
    """mathLibrary.py:629-665, 3-D."""
    v = np.asarray(v, dtype=float)
    r = np.sqrt(np.sum(v * v, axis=-1))
    th = np.zeros(r.shape)
    nz = r != 0
    if np.any(nz):
        th[nz] = np.arccos(v[..., 2][nz] / r[nz])
    ph = np.arctan2(v[..., 1], v[..., 0])
    return np.stack((r, th, np.where(ph < 0, ph + 2 * np.pi, ph)), axis=-1)


def calc_center(rs, theta, phi, density):
    """generate_calc_center (misk.py:295-312): centre of mass of Re(rho) with the plain SphericalIntegrator
    (mathLibrary.py:1223-1232), in spherical coordinates."""
    wr, wt = integrator_weights(rs, len(theta))
    d = np.asarray(density).real
    total = np.einsum('q,t,qtp->', wr, wt, d)
    if total == 0:
        total = 1
    st, ct = np.sin(theta), np.cos(theta)
    rd = np.asarray(rs)[:, None, None] * d
    cx = np.einsum('q,t,qtp->', wr, wt * st, rd * np.cos(phi)[None, None, :])
    cy = np.einsum('q,t,qtp->', wr, wt * st, rd * np.sin(phi)[None, None, :])
    cz = np.einsum('q,t,qtp->', wr, wt * ct, rd)
    return cartesian_to_spherical(np.array([cx, cy, cz]) / total)


def shift_phases(qs, theta, phi, vector, opposite_direction=False):
    """generate_shift_by_operator (fxs_Projections.py:1419-1444): exp(-i s k.c) on the (q, theta, phi) grid."""
    pre = -1 if opposite_direction else 1
    c = spherical_to_cartesian(np.asarray(vector, dtype=float))
    st, ct = np.sin(theta)[None, :, None], np.cos(theta)[None, :, None]
    q = np.asarray(qs)[:, None, None]
    kc = q * (st * (np.cos(phi)[None, None, :] * c[0] + np.sin(phi)[None, None, :] * c[1]) + ct * c[2])
    return np.exp(-1.j * pre * kc)


# ---------------------------------------------------------------------------------------------- SO(3) tables (alignment)
def wigner_d(l, betas):
    """d^l_mn(beta) = <l m| exp(-i beta J_y) |l n>, (len(betas), 2l+1, 2l+1), m, n = -l..l, from the eigen-decomposition of
    J_y (unitary by construction).  The reference gets these from pysofft (``soft_plugin.py:30-36``, ``genWigAll``)."""
    betas = np.atleast_1d(np.asarray(betas, dtype=float))
    w, v = _jy_eig(l)
    return np.ascontiguousarray(np.einsum('ik,bk,jk->bij', v, np.exp(-1j * betas[:, None] * w[None, :]), v.conj()).real)


_JY_EIG = {}


def _jy_eig(l):
    """eigen-decomposition of J_y in the |l m> basis (depends on l only: kept, every rotation of coefficients asks for it)"""
    hit = _JY_EIG.get(l)
    if hit is None:
        m = np.arange(-l, l)
        c = 0.5 * np.sqrt(l * (l + 1) - m * (m + 1))
        jy = np.zeros((2 * l + 1, 2 * l + 1), complex)
        idx = np.arange(2 * l)
        jy[idx + 1, idx] = -1j * c
        jy[idx, idx + 1] = 1j * c
        hit = _JY_EIG[l] = np.linalg.eigh(jy)
    return hit


def euler_grid(bw):
    """Euler angle samples of the SO(3) grid of bandwidth bw (``soft_plugin.py:55-58``: alpha, beta, gamma)"""
    j = np.arange(2 * bw)
    return 2 * np.pi * j / (2 * bw), np.pi * (2 * j + 1) / (4 * bw), 2 * np.pi * j / (2 * bw)


def so3_table_offsets(L):
    return np.array([l * (4 * l * l - 1) // 3 for l in range(L + 2)])


def wigner_d_table(L):
    """(2bw, sum_l (2l+1)^2) table of d^l_mn(beta_b) for the device (mtip_set_so3_tables)"""
    bw = L + 1
    off = so3_table_offsets(L)
    out = np.empty((2 * bw, off[-1]))
    be = euler_grid(bw)[1]
    for l in range(L + 1):
        out[:, off[l]:off[l + 1]] = wigner_d(l, be).reshape(2 * bw, -1)
    return out


def wigner_D_flat(L, euler):
    """D^l_mn(alpha, beta, gamma) = e^{-i m alpha} d^l_mn(beta) e^{-i n gamma} for l = 0..L in the table layout"""
    a, b, g = euler
    off = so3_table_offsets(L)
    out = np.empty(off[-1], complex)
    for l in range(L + 1):
        m = np.arange(-l, l + 1)
        out[off[l]:off[l + 1]] = (np.exp(-1j * m * a)[:, None] * wigner_d(l, [b])[0] * np.exp(-1j * m * g)[None, :]).reshape(-1)
    return out


_RSETUP_CACHE = {}
_RSETUP_LOCK = None


def reciprocal_setup(qs, data, l_max, ropt):
    """ReciprocalSetup(qs, data, l_max, ropt), built once per (invariants object, grid, options): the engines of one worker
    (GPU.n_gpu_workers restart groups) share the same invariants, and the per-column cubic regridding of the projection matrices
    (the same scipy call the reference makes, 1089 columns at L = 32) is ~0.3 s of pure Python per engine otherwise.  The result is
    read only.  The cache holds the last few setups and keeps `data` alive with them, so ids cannot be recycled under it."""
    global _RSETUP_LOCK
    import threading
    if _RSETUP_LOCK is None:
        _RSETUP_LOCK = threading.Lock()
    key = (id(data), int(l_max), np.asarray(qs, dtype=float).tobytes(), repr(ropt))
    with _RSETUP_LOCK:                                   # engines are created from worker threads: build once, the others wait
        hit = _RSETUP_CACHE.get(key)
        if hit is None:
            hit = (ReciprocalSetup(qs, data, l_max, ropt), data)
            while len(_RSETUP_CACHE) >= 4:
                _RSETUP_CACHE.pop(next(iter(_RSETUP_CACHE)))
            _RSETUP_CACHE[key] = hit
    return hit[0]
