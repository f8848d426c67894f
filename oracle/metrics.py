"""CPU restatement (TEST INFRASTRUCTURE) of the reference's non-default reciprocal error metrics, SURVEY section 8 f-4:
``fqc_error``, ``II_error`` and ``ccd_diff`` of ``xframe/projects/fxs/projectLibrary/fxs_IO_methods.py`` (472-551, 552-627, 628-683) with
the normalised associated Legendre matrices of ``fxs_invariant_tools.py:23-33, 48-58`` and ``ewald_sphere_theta_pi``
(``library/physicsLibrary.py:94-95``).

**Parity status.**  The metric formulas are pinned by fixture G19 (tests/golden/metrics_ops.npz): the reference's own
``_generate_fqc_3d`` / ``_generate_II_3d`` / ``_generate_ccd_diff_3d`` on seeded invariants.  **Unpinned**: the values of GSL's
``gsl_sf_legendre_sphPlm`` (pygsl is absent; ``gsl_plugin.py:8-69`` is the only call site) -- restated from its published definition
``sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!) P_l^m(x)`` (Condon-Shortley phase included), i.e. ``Y_l^m(theta, 0)``, through scipy; the fixture
run used a double for ``mathLibrary.gsl`` built on THIS function (``II_error`` does not use the values it requests)."""
import numpy as np
from scipy.special import sph_harm_y

from .projections import harmonic_coeff_to_deg2_invariants_3d


def sphPlm(l, m, x):
    """gsl_sf_legendre_sphPlm(l, m, x) for arrays l, m (broadcast against x)"""
    return sph_harm_y(np.asarray(l), np.asarray(m), np.arccos(np.clip(x, -1.0, 1.0)), 0.0).real


def legendre_sphPlm_array(l_max, m_max, xs):
    """gsl_plugin.py:38-48 (ordered by m): values (n_lm, n_x), ls, ms"""
    ms = np.arange(m_max + 1)
    ls = np.concatenate([np.arange(m, l_max + 1) for m in ms])
    rms = np.concatenate([np.full(max(0, l_max + 1 - m), m) for m in ms])
    xs = np.atleast_1d(xs)
    return sphPlm(ls[:, None], rms[:, None], xs[None, :]), ls, rms


def legendre_sphPlm_array_single_m(l_max, m, xs):
    """gsl_plugin.py:50-59"""
    ls = np.arange(l_max + 1)
    xs = np.atleast_1d(xs)
    return sphPlm(ls[:, None], np.full(len(ls), m)[:, None], xs[None, :]), ls, np.full(len(ls), m)


def ewald_sphere_theta_pi(wavelength, qs):
    return np.arccos(qs * wavelength / (4 * np.pi))


def ccd_associated_legendre_matrices(thetas, l_max, m_max):
    """fxs_invariant_tools.py:23-33: (q, q', m, l) = P^m_l(q) P^m_l(q') / (2l + 1)"""
    qm = np.zeros((len(thetas), m_max + 1, l_max + 1))
    values, ls, ms = legendre_sphPlm_array(l_max, m_max, np.cos(thetas))
    qm[:, ms, ls] = values.T
    return qm[None, :] * qm[:, None] / (2 * np.arange(l_max + 1) + 1)[None, None, None, :]


def ccd_associated_legendre_matrices_single_m(thetas, l_max, m):
    """fxs_invariant_tools.py:48-58: (q, q', l)"""
    qm = np.zeros((len(thetas), l_max + 1))
    values, ls, _ = legendre_sphPlm_array_single_m(l_max, m, np.cos(thetas))
    qm[:, ls] = values.T
    return qm[None, :] * qm[:, None] / (2 * np.arange(l_max + 1) + 1)[None, None, :]


def _masked_reference(reference_invariant, used_orders, invariant_mask, by='values'):
    ids = np.array(tuple(used_orders.values()))
    ref = np.array(reference_invariant)
    mask = np.zeros(ref.shape, dtype=bool)
    mask[:] = ~invariant_mask[ids]
    ref[mask] = 0
    return ref, mask


def fqc_error_routine(radial_points, reference_invariant, used_orders, invariant_mask, xray_wavelength):
    """_generate_fqc_3d (fxs_IO_methods.py:507-550); reference_invariant already indexed by the used orders (481)"""
    order_array = np.array(tuple(used_orders.keys())).astype(int)
    max_order = int(np.max(order_array))
    thetas = ewald_sphere_theta_pi(xray_wavelength, radial_points)
    P = np.moveaxis(ccd_associated_legendre_matrices(thetas, max_order, max_order), -1, 0)          # (l, q, q', m)
    ref, mask = _masked_reference(reference_invariant, used_orders, invariant_mask)

    def calc_ccn(bl):
        return np.sum(bl[1:, ..., None] * P[1:], axis=0)

    def avg2(c1, c2):
        return (c1[..., 0] * c2[..., 0]).real + 2 * np.sum(c1[..., 1:] * c2[..., 1:].conj(), axis=-1).real
    ref_ccn = calc_ccn(ref)
    ref_avg = avg2(ref_ccn, ref_ccn)
    ref_w = (P[1:, ..., 0] * ref_ccn[None, ..., 0]).real + 2 * np.sum(P[1:, ..., 1:] * ref_ccn[None, ..., 1:].conj(), axis=-1).real

    def fqc_error(Ims):
        Bl = harmonic_coeff_to_deg2_invariants_3d(Ims)
        Bl[mask] = 0
        ccn = calc_ccn(Bl)
        average = avg2(ccn, ccn)
        with np.errstate(all='ignore'):
            norm = np.sqrt(average * ref_avg)
            pos = norm >= 0
            control = np.sum(Bl[1:] * ref_w, axis=0)
            fqc = np.ones_like(ref_avg)
            fqc[pos] = (control[pos] / norm[pos]).real          # (the reference assigns the complex quotient to a real array)
        return np.array([1 - np.mean(fqc[i, :i + 1]) for i in range(len(radial_points))])
    fqc_error.tables = {'P': P, 'ref_avg': ref_avg, 'ref_w': ref_w, 'mask': mask}
    return fqc_error


def II_error_routine(radial_points, reference_invariant, used_orders, invariant_mask):
    """_generate_II_3d (587-627)"""
    ref, mask = _masked_reference(reference_invariant, used_orders, invariant_mask)
    ref_II = np.sum(ref[1:], axis=0)
    qq = (radial_points[:, None] * radial_points[None, :]) ** 2

    def II_error(Ims):
        Bl = harmonic_coeff_to_deg2_invariants_3d(Ims)
        Bl[mask] = 0
        cur = np.sum(Bl[1:], axis=0)
        return 1 - np.sum(cur * ref_II * qq) / np.sqrt(np.sum(cur ** 2 * qq) * np.sum(ref_II ** 2 * qq))
    II_error.tables = {'ref_II': ref_II, 'qq': qq, 'mask': mask}
    return II_error


def ccd_diff_routine(radial_points, reference_invariant, used_orders, n_particles, invariant_mask, C_order, xray_wavelength):
    """_generate_ccd_diff_3d (651-683)"""
    order_array = np.array(tuple(used_orders.values()))
    zero_id = used_orders[0]
    relevant = order_array >= C_order
    thetas = ewald_sphere_theta_pi(xray_wavelength, radial_points)
    PP = np.moveaxis(np.squeeze(ccd_associated_legendre_matrices_single_m(thetas, int(np.max(order_array)), C_order)), -1, 0)[order_array]
    PP[zero_id] = 0
    PP = PP[relevant]
    PP[np.isnan(PP)] = 0
    ref = np.array(reference_invariant)
    mask = np.zeros(ref.shape, dtype=bool)
    mask[:] = ~invariant_mask[order_array]
    ref[mask] = 0
    ref_C = np.sum(ref[relevant] * PP, axis=0)
    norm = np.sum(ref_C * ref_C.conj())
    assert norm != 0

    def ccd_error(Ims):
        Bl = harmonic_coeff_to_deg2_invariants_3d(Ims)[order_array]
        Bl[mask] = 0
        Bl[zero_id] *= np.sqrt(n_particles)
        d = np.sum(Bl[relevant] * PP, axis=0) - ref_C
        return np.sum((d * d.conj()).real) / norm
    ccd_error.tables = {'PP': PP, 'ref_C': ref_C, 'norm': norm, 'mask': mask, 'relevant': relevant}
    return ccd_error
