"""Synthetic single-particle inputs for benchmarks and parity tests (SURVEY.md section 8 d).

Mirrors the maths of the reference's ``simulate_ccd`` -> ``extract`` front half only as an input
generator: density -> I = |FT rho|^2 -> I_lm -> B_l = I_l I_l^+ -> V_l = eigvecs sqrt(eigvals)
(``xframe/projects/fxs/projectLibrary/fxs_invariant_tools.py:1133-1207``,
``xframe/projects/fxs/simulate_ccd.py:196-233``).  The transforms are supplied by the caller
(``transforms.ft(grid)``, ``transforms.forward_l(grid)``) so the same generator runs on the HIP
engine (bench) and on the oracle (tests).
"""
import numpy as np

XRAY_WAVELENGTH = 1.23984
PARTICLE_RADIUS = 250.0
OVERSAMPLING = 4
KAPPA = 2.0


def data_cutoff(n_radial_points, kappa=KAPPA, particle_radius=PARTICLE_RADIUS, oversampling=OVERSAMPLING):
    """Q_d such that R = kappa*N/Q_d = oversampling * particle_radius."""
    return kappa * n_radial_points / (oversampling * particle_radius)


def midpoint_points(cutoff, n):
    d = cutoff / n
    return np.linspace(d / 2, cutoff - d / 2, num=n, endpoint=True)


def ball_density(rs, thetas, phis, seed=20241020, n_balls=6, particle_radius=PARTICLE_RADIUS):
    """Sum of uniform balls: radii U(.15,.3) R_p, centres uniform in the ball of radius .5 R_p,
    densities U(1,2)."""
    rng = np.random.default_rng(seed)
    radii = rng.uniform(0.15, 0.3, n_balls) * particle_radius
    # uniform in ball: direction normal, radius ~ u^(1/3)
    d = rng.normal(size=(n_balls, 3))
    d /= np.linalg.norm(d, axis=1)[:, None]
    centres = d * (0.5 * particle_radius * rng.uniform(0, 1, n_balls) ** (1 / 3))[:, None]
    dens = rng.uniform(1, 2, n_balls)
    r, t, p = np.meshgrid(rs, thetas, phis, indexing='ij')
    x = r * np.sin(t) * np.cos(p)
    y = r * np.sin(t) * np.sin(p)
    z = r * np.cos(t)
    rho = np.zeros(r.shape)
    for c, a, v in zip(centres, radii, dens):
        rho += v * (((x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2) < a * a)
    return rho.astype(complex)


def invariants_from_intensity_coefficients(Ilm, data_radial_points, max_order, eigh=None):
    """I_lm (list over l of (N,2l+1)) -> the dict ``load_invariants`` would hand to the worker
    (``xframe/projects/fxs/_database_.py:566-609``).  Stored in the reference's on-disk convention:
    V_l halved (``fxs_Projections.py:710-713`` multiplies by 2), average_intensity from B_0.
    ``eigh``: an object with ``hermitian_eig`` (an Engine: the ``extract`` step on the device with the reference's rules of
    ``xframe_amd.fxs.extract``, fxs_invariant_tools.py:1114-1207, sort_mode 0).  Required: the product has no CPU eigensolver (the
    test infrastructure hands in a numpy adapter when it builds inputs for the CPU oracle)."""
    N = len(data_radial_points)
    bls = []
    for l in range(max_order + 1):
        Il = np.asarray(Ilm[l])
        # B_l of a real intensity is real (the +-m pairs are conjugate); `Il @ Il^+` leaves rounding residue in the
        # imaginary part, dropped here as the reference's own B_l-from-coefficients routine does (fxs_invariant_tools.py:1255),
        # so that the projection matrices are exactly real like those of its cross-correlation route
        B = (Il @ Il.conj().T).real / 4.0                  # stored convention: (V/2)(V/2)^+
        bls.append((B + B.T) / 2)
    bls = np.stack(bls)
    from . import extract as X
    pms_t, _ = X.deg2_invariant_to_projection_matrices(eigh, bls)
    pms = np.empty(max_order + 1, dtype=object)
    for l in range(max_order + 1):
        pms[l] = pms_t[l]
    aint = np.sqrt(np.diag(bls[0]).real / (4 * np.pi))
    return {'dimensions': 3, 'xray_wavelength': XRAY_WAVELENGTH, 'average_intensity': aint,
            'data_radial_points': np.asarray(data_radial_points), 'data_angular_points': np.zeros(1),
            'max_order': max_order, 'data_projection_matrices': pms}


def make_invariants(transforms, n_radial_points, max_order, seed=20241020, eigh=None):
    """transforms must be built on the *data* grid (max_q = data_cutoff(N)) and expose
    ``rs``, ``thetas``, ``phis``, ``ft(grid)->grid`` and ``forward_l(grid)->list``."""
    rho = ball_density(transforms.rs, transforms.thetas, transforms.phis, seed)
    F = transforms.ft(rho)
    I = F * F.conj()
    Ilm = transforms.forward_l(I)
    q_d = midpoint_points(data_cutoff(n_radial_points), n_radial_points)
    # the eigensolver: given explicitly, else the transforms object itself when it is an Engine (it has `hermitian_eig`)
    return invariants_from_intensity_coefficients(Ilm, q_d, max_order, eigh if eigh is not None else transforms), rho


# ---- BASELINE.json configs (SURVEY section 8 d) ---------------------------------------------
TUTORIAL_OVERRIDES = {
    'structure_name': 'synthetic', 'particle_radius': PARTICLE_RADIUS,
    'density_guess': {'type': 'bump', 'bump': {'slope': 0.3}, 'radius': PARTICLE_RADIUS,
                      'amplitude_function': 'random', 'random': {'SNR': 2}},
    'projections': {
        'real': {
            'shrink_wrap': {'sigmas': [[20, [False, 5], -2], False], 'thresholds': [0.09, 0.09]},
            'HIO': {'beta': [[0.5, 0.4, -1 / 250, 500], [0.01, 0.002, -1 / 200, 200]]},
            'projections': {'apply': ['support', 'value_threshold', 'limit_imag'],
                            'support': {'initial_support': {'type': 'max_radius', 'max_radius': PARTICLE_RADIUS},
                                        'enforce_initial_support': {'apply': True, 'if_error_bigger_than': 6e-3}},
                            'value_threshold': {'threshold': [0, False]},
                            'limit_imag': {'threshold': 2}}},
        'reciprocal': {'number_of_particles': {'initial': 1}, 'use_averaged_intensity': True,
                       'q_mask': {'type': 'none'}}},
}

_SIZES = {1: (32, 8), 2: (64, 16), 3: (128, 32), 4: (128, 32), 5: (256, 48)}


def config_overrides(cfg):
    """Settings overrides (on top of the reference defaults) for BASELINE.json config 1..5."""
    N, L = _SIZES[cfg]
    o = {k: v for k, v in TUTORIAL_OVERRIDES.items()}
    o['grid'] = {'n_radial_points': N, 'max_order': L, 'max_q': False, 'n_phi': 0, 'n_theta': 0}
    o['fourier_transform'] = {'type': 'midpoint', 'reciprocity_coefficient': KAPPA}
    rec = dict(o['projections']['reciprocal'])
    rec['used_order_ids'] = np.arange(L + 1)
    o['projections'] = {'real': o['projections']['real'], 'reciprocal': rec}
    hio, er = {'iterations': 60, 'ft_stab': True}, {'iterations': 40, 'ft_stab': True}
    if cfg == 1:
        loops = {'order': ['main'],
                 'main': {'methods': {'HIO': hio, 'ER': er, 'SW': 1}, 'order': ['HIO', 'SW', 'ER'],
                          'iterations': 1, 'best_density_not_in_first_n_iterations': np.inf}}
        err = None
    elif cfg in (2, 5):
        loops = {'order': ['main'],
                 'main': {'methods': {'HIO': hio, 'ER': er}, 'order': ['HIO', 'ER'],
                          'iterations': 2, 'best_density_not_in_first_n_iterations': np.inf}}
        err = {'methods': {'reciprocal': {'calculate': ['deg2_invariant_l2_diff'],
                                          'deg2_invariant_l2_diff': {'order': 2}}}}
    else:
        loops = {'order': ['main', 'refinement'],
                 'main': {'methods': {'HIO': hio, 'ER': er, 'SW': 1}, 'order': ['HIO', 'SW', 'ER'],
                          'iterations': 5, 'best_density_not_in_first_n_iterations': np.inf},
                 'refinement': {'methods': {'ER': {'iterations': 100, 'ft_stab': True}, 'SW': 1},
                                'order': ['SW', 'ER'], 'iterations': 1,
                                'best_density_not_in_first_n_iterations': np.inf}}
        err = None
    o['main_loop'] = {'sub_loops': loops}
    if err is not None:
        o['main_loop']['error'] = err
    return o
