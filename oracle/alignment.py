"""CPU restatement (TEST INFRASTRUCTURE) of the alignment + averaging of reconstructions, SURVEY section 8 f-1:
``xframe/projects/fxs/average.py`` (``ProjectWorker.run_3d`` 359-627, ``Alignment`` 729-1111),
``xframe/externalLibraries/soft_plugin.py:17-99``, ``projectLibrary/resolution_metrics.py:62-110``.

**Parity status.**  The FLOW is pinned by the reference's own code: ``tests/golden/average_flow.npz`` (G17) holds what
``ProjectWorker.run_3d`` / ``Alignment`` of the imported reference do with two seeded sets of reconstructions (centring,
normalisation, reference choice, alignment of each reconstruction and of its point inverse, error limit, selection, averages,
PRTF variants, centred average), and this restatement reproduces it; so do fixtures G12 (centring) and G14 (``PRTF``,
``integrate_normed``).  **Unpinned** remain the two calls into the third-party ``pysofft`` (``calc_mean_C_array``,
``rotate_coeff_multi``, ``get_euler_angles``; not vendored, not installed): the fixture run used a double for them that is
built on THIS file (tests/golden/make_golden.py ``install_pysofft_double``), so their grid / sign / normalisation conventions
are those restated below from the published definitions the plugin quotes (soft_plugin.py:64-99): ``C(R) = <f, g o R>`` on the
(2 bw)^3 Euler grid (``alpha_j = 2 pi j / 2bw``, ``beta_k = pi (2k+1) / 4bw``, ``gamma_j`` like alpha, bw = L + 1) and
``f_lm -> sum_n D^l_nm f_ln``, ZYZ convention ``D^l_mn(alpha, beta, gamma) = e^{-i m alpha} d^l_mn(beta) e^{-i n gamma}``, pinned
by known answers only (a density rotated by a grid rotation is found again, Wigner matrices are unitary and compose).  What the
reference's averaged density depends on is the *composition* find_rotation -> rotate, which is convention free: the rotation
that maximises the overlap with the reference is applied.

One behaviour of the reference is kept literally because its results depend on it: ``find_rotation`` edits the Euler-angle grid
IN PLACE (average.py:938-940: the grid entry it reads is a view, ``alpha -> 2 pi - alpha``, ``gamma -> 2 pi - gamma``), so a grid
point that is found a second time by the same ``Alignment`` object hands out the un-flipped angles, and the angles it stored for
the first find change with it.  Set A of the fixture holds such a pair.
"""
import numpy as np

from . import projections as P
from .fourier import SphericalIntegrator


# ---------------------------------------------------------------------------------------------- Wigner matrices
def _jy(l):
    """matrix of J_y in the basis |l, m>, m = -l..l (Condon-Shortley phases)"""
    m = np.arange(-l, l)
    c = 0.5 * np.sqrt(l * (l + 1) - m * (m + 1))          # <m+1| J_+ |m> / 2
    jy = np.zeros((2 * l + 1, 2 * l + 1), complex)
    idx = np.arange(2 * l)
    jy[idx + 1, idx] = -1j * c                                # J_y = (J_+ - J_-) / 2i
    jy[idx, idx + 1] = 1j * c
    return jy


def wigner_d(l, betas):
    """d^l_{mn}(beta) = <l m| exp(-i beta J_y) |l n>, (len(betas), 2l+1, 2l+1) real, rows / columns m, n = -l..l.
    From the eigen-decomposition of J_y: unitary by construction, stable for any l."""
    betas = np.atleast_1d(np.asarray(betas, dtype=float))
    w, v = np.linalg.eigh(_jy(l))
    d = np.einsum('ik,bk,jk->bij', v, np.exp(-1j * betas[:, None] * w[None, :]), v.conj())
    return np.ascontiguousarray(d.real)


def wigner_D(l, euler):
    """D^l_{mn}(alpha, beta, gamma) = e^{-i m alpha} d^l_{mn}(beta) e^{-i n gamma} (ZYZ, active)"""
    a, b, g = euler
    m = np.arange(-l, l + 1)
    return np.exp(-1j * m * a)[:, None] * wigner_d(l, [b])[0] * np.exp(-1j * m * g)[None, :]


def euler_grid(bw):
    """(alpha_j, beta_k, gamma_j) samples of the SO(3) grid of bandwidth bw (soft_plugin.py:55-58)"""
    j = np.arange(2 * bw)
    return 2 * np.pi * j / (2 * bw), np.pi * (2 * j + 1) / (4 * bw), 2 * np.pi * j / (2 * bw)


# ---------------------------------------------------------------------------------------------- rotation and correlation
def rotate_coeff(coeff, euler, L):
    """(R f)(x) = f(R^-1 x):  f_lm -> sum_n D^l_mn(R) f_ln for every shell; coeff (Nq, (L+1)^2), index l(l+1)+m"""
    out = np.empty_like(coeff)
    for l in range(L + 1):
        D = wigner_D(l, euler)
        out[:, l * l:(l + 1) ** 2] = coeff[:, l * l:(l + 1) ** 2] @ D.T
    return out


def correlation(ref, sig, L, r_limit_ids=None):
    """C(alpha_i, beta_j, gamma_k) = mean over the shells r_lo <= r < r_hi of Re <ref_r, R(alpha, beta, gamma) sig_r>
    = Re sum_l sum_mn T^l_mn d^l_mn(beta) e^{-i m alpha} e^{-i n gamma},  T^l_mn = mean_r conj(ref_lm(r)) sig_ln(r).
    Returns (2bw, 2bw, 2bw) indexed [alpha, beta, gamma]."""
    bw = L + 1
    lo, hi = (0, ref.shape[0]) if r_limit_ids is None else r_limit_ids
    al, be, ga = euler_grid(bw)
    m_all = np.arange(-L, L + 1)
    S = np.zeros((len(be), 2 * L + 1, 2 * L + 1), complex)
    for l in range(L + 1):
        T = ref[lo:hi, l * l:(l + 1) ** 2].conj().T @ sig[lo:hi, l * l:(l + 1) ** 2] / (hi - lo)
        d = wigner_d(l, be)
        S[:, L - l:L + l + 1, L - l:L + l + 1] += T[None] * d
    Ea = np.exp(-1j * al[:, None] * m_all[None, :])          # (alpha, m)
    Eg = np.exp(-1j * ga[:, None] * m_all[None, :])          # (gamma, n)
    C = np.einsum('am,bmn,gn->abg', Ea, S, Eg)
    return C.real


def find_rotation(ref, sig, L, r_limit_ids=None):
    C = correlation(ref, sig, L, r_limit_ids)
    a, b, g = np.unravel_index(np.argmax(C), C.shape)
    al, be, ga = euler_grid(L + 1)
    return np.array([al[a], be[b], ga[g]]), C


def mean_C_layout(C):
    """The correlation in the layout average.py:936-940 reads it in: indexed [beta, alpha, gamma] (the grid is read at
    [argmax[1], argmax[0], argmax[2]]) and tabulated at the angles whose flip (alpha -> 2 pi - alpha, gamma -> 2 pi - gamma) is
    the rotation that maps the signal onto the reference; the arg-max (first maximum in THIS order) decides ties."""
    n = C.shape[0]
    flip = (-np.arange(n)) % n
    return C[flip][:, :, flip].transpose(1, 0, 2)


# ---------------------------------------------------------------------------------------------- the alignment loop (average.py:1043-1110)
class Alignment:
    def __init__(self, fp, opt=None):
        """fp: oracle FourierPair of the reconstruction grid"""
        self.fp, self.sht, self.L = fp, fp.sht, fp.sht.l_max
        self.opt = {'max_iterations': 1, 'alignment_error_limit': 0.3, 'find_rotation': {}}
        self.opt.update(opt or {})
        self.integrator = SphericalIntegrator(fp.rs, fp.sht.n_theta)
        self.real_grid = fp.grid.real_grid()
        self.reciprocal_grid = fp.grid.reciprocal_grid()
        self.soft_grid = np.stack(np.meshgrid(*euler_grid(self.L + 1), indexing='ij'), -1)     # make_SO3_grid; edited in place below

    def shift_to_center(self, density, ft_density):
        """assemble_shift_to_center (average.py:1007-1020): (IFT(FT(rho) e^{i k c}), F e^{i k c}, c)"""
        c = P.calc_center(self.fp.rs, self.sht.n_theta, self.real_grid, density)
        ph = P.shift_phases(self.reciprocal_grid, c, opposite_direction=True)
        return self.fp.ift(self.fp.ft(density) * ph), ft_density * ph, c

    def find_rotation(self, cr, cs):
        """find_rotation (average.py:922-946), literally: the returned angles are a VIEW of the grid entry, flipped in place"""
        r_lim = self.opt['find_rotation'].get('r_limit_ids', [0, len(self.fp.rs)])
        C = correlation(cr, cs, self.L, [int(r_lim[0]), int(r_lim[1])])
        mean_C = mean_C_layout(C)
        am = np.unravel_index(np.argmax(mean_C), mean_C.shape)
        euler = self.soft_grid[am[1], am[0], am[2]]
        euler[0] = 2 * np.pi - euler[0]
        euler[2] = 2 * np.pi - euler[2]
        return euler, mean_C

    def align(self, ref, sig, ft_sig):
        """rotate_signal sketch (average.py:970-975): the rotation found on the densities is applied to both halves"""
        cr, cs, cf = self.sht.forward_d(ref), self.sht.forward_d(sig), self.sht.forward_d(ft_sig)
        euler, C = self.find_rotation(cr, cs)
        return (self.sht.inverse_d(rotate_coeff(cs, euler, self.L)), self.sht.inverse_d(rotate_coeff(cf, euler, self.L)),
                euler, C)

    def alignment_loop(self, reference, signal):
        ref = np.array(reference)
        sig, ft_sig = signal
        norm = self.integrator.integrate_normed(ref.real ** 2)
        norm = norm if norm != 0 else 1
        errors, angles = [], []
        for _ in range(self.opt['max_iterations']):
            sig, ft_sig, euler, _ = self.align(ref, sig, ft_sig)
            errors.append(self.integrator.integrate_normed((ref.real - sig.real) ** 2) / norm)
            angles.append(euler)
            break                                # the loop ends when the last shift is 0: the shift step is disabled upstream (979)
        return {'densities': [sig, ft_sig], 'errors': errors, 'rotation_angles': angles}

    def apply_to(self, reference, signal):
        """alignment_routine (1089-1109): align the signal, then its point inverse, keep the better one"""
        inverted = (self.fp.ift(self.fp.ft(signal[0]).conj()), signal[1].conj())
        out, out_inv = self.alignment_loop(reference, signal), self.alignment_loop(reference, inverted)
        norm = self.integrator.integrate_normed(np.asarray(reference).real ** 2)
        e = self.integrator.integrate_normed((reference.real - out['densities'][0].real) ** 2) / norm
        e_inv = self.integrator.integrate_normed((reference.real - out_inv['densities'][0].real) ** 2) / norm
        best = out if e < e_inv else out_inv
        best['inverted'] = not (e < e_inv)
        return best


def normalize_density(d, d_min=False):
    """average.py:721-727"""
    if isinstance(d_min, bool):
        d_min = d.real.min()
    return (d - d_min) / (np.max(d.real) - d_min)


def PRTF(a1, a2, b1, b2):
    """resolution_metrics.py:62-78"""
    axes = tuple(range(1, a1.ndim))
    nd = np.ones(a1.shape, dtype=complex)
    nz = (b1 != 0) & (b2 != 0)
    nd[nz] = (a1[nz] * a2[nz].conj()) / (b1[nz] * b2[nz].conj())
    nd[~nz & (a1 != 0) & (a2 != 0)] = 0
    nd = np.sqrt(nd)
    return np.average(nd, axis=axes), np.std(nd, axis=axes)


def average_reconstructions(fp, reconstructions, errors, opt=None):
    """run_3d (average.py:359-627) for reconstructions = [(real_density, reciprocal_density), ...]: centre, normalise, reference =
    lowest selection error (optionally point inverted, 457-464), align the rest (incl. the point-inversion test), mean of those
    below the alignment error limit, the four PRTF variants (546-561), the centred average."""
    o = {'center_reconstructions': True, 'normalize_reconstructions': {'use': True, 'mode': 'max'}, 'max_iterations': 1,
         'alignment_error_limit': 0.3, 'find_rotation': {}, 'n_reconstructions': len(reconstructions), 'pointinvert_reference': False,
         'average_normalization_min': False}
    o.update(opt or {})
    if isinstance(o.get('selection'), dict) and isinstance(o['selection'].get('n_reconstructions'), int):
        o['n_reconstructions'] = min(o['selection']['n_reconstructions'], len(reconstructions))     # average.py:113-115
    al = Alignment(fp, o)
    recs = [[np.array(r[0]), np.array(r[1])] for r in reconstructions]
    if o['center_reconstructions']:
        recs = [list(al.shift_to_center(*r)[:2]) for r in recs]
    scales = np.ones(len(recs))
    if o['normalize_reconstructions']['use']:
        for i, r in enumerate(recs):
            if o['normalize_reconstructions']['mode'] == 'max':
                if np.max(r[0]).real <= 0:
                    continue
                scale = np.max(r[0][r[0] > 0].real)
            else:
                scale = np.mean(r[0][r[0] > 0])               # (432-435: complex; only its real part reaches scaling_factors)
            scales[i] = np.real(scale)
            recs[i] = [r[0] / scale, r[1] / scale]
    ref_arg = int(np.argmin(errors))
    reference = recs.pop(ref_arg)
    if o.get('pointinvert_reference', False):
        ri = reference[1].conj()
        reference = [fp.ift(ri), ri]
    valid, valid_err, all_err, angles, inverted, valid_ids = [reference], [], [], [], [], [0]
    for r_id, r in enumerate(recs):
        out = al.apply_to(reference[0].copy(), r)
        all_err.append(out['errors'][-1])
        angles.append(out['rotation_angles'][-1])          # views of the angle grid, as upstream: later finds may still change them
        inverted.append(out['inverted'])
        if out['errors'][-1] < o['alignment_error_limit']:
            valid.append(out['densities'])
            valid_err.append(out['errors'][-1])
            valid_ids.append(r_id)
    # average.py:519-524, literally: `valid` starts with the reference but `valid_err` does not, and the argsort of the errors
    # indexes `valid` -- so the reference is always in, and the last valid alignment (in processing order) never is
    aligned = [valid[i] for i in np.argsort(valid_err)]
    if len(aligned) >= o['n_reconstructions']:
        aligned = aligned[:o['n_reconstructions']]
    if not aligned:                                      # (no valid alignment: np.mean of an empty list upstream) keep the reference
        aligned = [reference]
    average = [np.mean([a[0] for a in aligned], axis=0), np.mean([a[1] for a in aligned], axis=0)]
    ftd = [fp.ft(a[0]) for a in aligned]
    I_ft = np.mean([(a[1] * a[1].conj()).real for a in aligned], axis=0)
    I_d = np.mean([(f * f.conj()).real for f in ftd], axis=0)
    # average.py:538: the averaged pair is centred BEFORE the metrics, and the reference's shift operator multiplies its argument
    # in place (fxs_Projections.py:1442) -- the averaged reciprocal density that is saved and that enters 'PRTF' is the shifted one
    cen = al.shift_to_center(*average)
    average[1] = cen[1]
    fa = fp.ft(average[0])
    metrics = {}
    for name, args in (('PRTF', (fa, average[1], np.sqrt(I_d), np.sqrt(I_ft))), ('PRTF_from_density', (fa, fa, np.sqrt(I_d), np.sqrt(I_d))),
                       ('PRTF_from_ft_density', (average[1], average[1], np.sqrt(I_ft), np.sqrt(I_ft))),
                       ('PRTF_ftI', (fa, fa, np.sqrt(I_ft), np.sqrt(I_ft)))):
        metrics[name], metrics[name + '_std'] = PRTF(*args)
    dmin = o.get('average_normalization_min', False)
    return {'average': {'real_density': average[0], 'normalized_real_density': normalize_density(average[0], dmin),
                        'reciprocal_density': average[1], 'intensity_from_densities': I_d, 'intensity_from_ft_densities': I_ft},
            'centered_average': {'real_density': cen[0], 'normalized_real_density': normalize_density(cen[0], dmin),
                                 'reciprocal_density': cen[1]},
            'resolution_metrics': metrics, 'aligned': aligned, 'average_ids': valid_ids,
            'alignment_errors': np.array(all_err), 'rotation_angles': np.array(angles), 'inverted': inverted,
            'scaling_factors': scales, 'reference_arg': ref_arg, 'so3_grid': al.soft_grid}
