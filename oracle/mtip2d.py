"""CPU restatement (TEST INFRASTRUCTURE) of the 2-D (polar) phasing loop, SURVEY section 8 f-4: the `dimensions == 2` branches of
``xframe/projects/fxs/reconstruct.py`` (263-266, 347-350, 421-423, 1126-1129) and ``projectLibrary/fxs_Projections.py`` (473-476,
506-511, 631-637, 679-714, 723-745, 803-826, 855-863, 189-203) on top of the 3-D restatement ``oracle/mtip.py`` (the loop, the
sketches, the real-space projections, the shrink-wrap and the ramps are the same code upstream) and the operators of
``oracle/polar2d.py``.

**Parity status**: pinned by fixture G20 (tests/golden/mtip2d_N12_M6.npz, `make_golden.py mtip2d`): single steps, the shrink-wrap
mask and a trajectory of the reference's own ``reconstruct.MTIP`` with ``dimensions: 2`` (its data grid differs from the internal one
by one ulp, so the regridding of fxs_Projections.py:651-662 runs -- per order vector, cubic, zero outside the data range -- and
the end points fall outside)."""
import numpy as np

from . import mtip as OM
from . import polar2d as P2
from . import projections as P


class PolarIntegrator:
    """mathLibrary.py:1242-1267: trapezoid in phi (over the samples, without closing the circle) and in r with weight r"""

    def __init__(self, rs, phis):
        self.rs, self.phis = np.asarray(rs), np.asarray(phis)
        self.max_r = np.max(self.rs)
        self.norm = np.pi * self.max_r ** 2

    def integrate(self, values):
        dp = np.diff(self.phis)
        s_int = np.sum(dp[None, :] * (values[:, 1:] + values[:, :-1]) / 2.0, axis=1)
        f = s_int * self.rs
        return np.sum(np.diff(self.rs) * (f[1:] + f[:-1]) / 2.0)

    def integrate_normed(self, values):
        return self.integrate(values) / self.norm


class _RealHarmonic:
    """the `harmonic_transform` pair of the 2-D loop (reconstruct.py:347-350: HarmonicTransform('real', ...)) under the names the
    3-D loop uses"""

    def __init__(self, n_phi):
        self.n_phi = n_phi

    def forward_l(self, grid):
        return P2.real_harmonic_forward(grid)

    def inverse_l(self, coeff):
        return P2.real_harmonic_inverse(coeff, self.n_phi)


class ReciprocalSetup2D(P2.ReciprocalProjection2D):
    """ReciprocalProjection.__init__ for dimensions == 2 (fxs_Projections.py:471-537)"""

    def __init__(self, radial_points, data, max_order, opt):
        q_d = np.asarray(data['data_radial_points'], dtype=float)
        aint = np.asarray(getattr(data['average_intensity'], 'data', data['average_intensity']), dtype=float)
        qs = np.asarray(radial_points, dtype=float)
        needs_regridding = q_d.shape != qs.shape or not (q_d == qs).all()                                   # 642-648
        interp = opt['regrid']['interpolation']
        self.opt = opt
        self.data_radial_points = q_d
        self.data_min_q, self.data_max_q = np.min(q_d), np.max(q_d)
        self.integrated_intensity = (q_d[1] - q_d[0]) * np.sum(aint * q_d, axis=0) * 2 * np.sqrt(np.pi)      # 473-474
        self.positive_orders = np.arange(max_order + 1)
        self.used_order_ids = np.asarray(opt['used_order_ids'])
        used_orders = {int(o): int(i) for o, i in zip(self.positive_orders, self.used_order_ids)}
        pm = np.array(np.asarray(data['data_projection_matrices'])[list(used_orders.values())], dtype=complex)
        if needs_regridding:                                                                                # 655-662: one vector at a time
            aint = P.regrid_1d(aint, q_d, qs, interp)
            pm = np.array([P.regrid_1d(v, q_d, qs, interp) for v in pm])
        self.average_intensity = aint
        self.full_projection_matrices = np.zeros((max_order + 1, len(qs)), dtype=complex)                   # 508-511
        for oid, p in zip(self.used_order_ids, pm):
            self.full_projection_matrices[oid] = p
        # modify_projection_matrices 679-714, 2-D branches: no factor 2, the zero order is the average intensity itself
        proj = pm.copy()
        keys = np.array(tuple(used_orders))
        if opt.get('odd_orders_to_0', False):
            proj[keys % 2 == 1, :] = 0
        if opt.get('use_averaged_intensity', False):
            proj[used_orders[0]] = aint.astype(complex)
        self.radial_points = qs
        self.used_orders = used_orders
        self.data_q_id_limits = False
        radial_mask = P.ReciprocalProjection._radial_mask(self, opt.get('q_mask', None))
        self.number_of_particles_list = [opt['number_of_particles']['initial']]
        super().__init__(proj, used_orders, radial_mask, qs, max_order + 1, self.number_of_particles_list[0])
        self.projection_matrices = proj
        self.number_of_particles = self.number_of_particles_list
        self.deg2_invariants = np.array([v[:, None] * v[None, :].conj() for v in proj])                     # 631-633, fxs_invariant_tools.py:906-914
        self.fixed_intensity = None
        self.SO_order_id = None
        self.use_SO_freedom = bool(opt.get('SO_freedom', {}).get('use', False))
        if self.use_SO_freedom:
            # generate_approximate_unknowns, 744-750: the unknown of the best ranked even order is set to 1 in every step
            self.SO_order_id = int(so_ranking_2d(proj, keys, qs, opt['SO_freedom'].get('radial_high_pass', 0.2))[0][0])

    def approximate_unknowns(self, I):
        u = super().approximate_unknowns(I)
        if self.SO_order_id is not None:
            u[self.SO_order_id] = 1
        return u

    def mtip_projection(self, I, unknowns):
        self.n_particles = float(self.number_of_particles_list[0])
        self.number_of_particles = float(self.number_of_particles_list[0])
        out = super().mtip_projection(I, unknowns)
        self.number_of_particles = self.number_of_particles_list
        return out

    project_to_modified_intensity = P.ReciprocalProjection.project_to_modified_intensity
    project_to_fixed_intensity = P.ReciprocalProjection.project_to_fixed_intensity


def so_ranking_2d(projection_vectors, orders, radial_points, radial_high_pass=0.2):
    """rank_projection_matrix_orders_2d, fxs_Projections.py:933-962: the even non-zero used orders by mean_q |v_m(q) q| above the
    radial high pass, best first -> (positions among the used orders, the orders, the sort permutation)"""
    radial_points = np.asarray(radial_points)
    idx = int((len(radial_points) - 1) * radial_high_pass)
    orders = np.asarray(orders)
    order_mask = (orders % 2 == 0) & (orders != 0)
    pv = np.asarray(projection_vectors)[order_mask, idx:].T
    metric = np.mean(np.abs(pv * radial_points[idx:, None]), axis=0)
    sorted_indices = np.argsort(metric)[::-1]
    so_indices = order_mask.nonzero()[0][sorted_indices]
    return so_indices, orders[so_indices], sorted_indices


def remaining_so_projection_2d(projection_vectors, used_orders, radial_points, n_phi, radial_high_pass=0.2):
    """generate_remaining_SO_projection_2D, fxs_Projections.py:1022-1095: the rotation that is left free after the best ranked order
    fixed its phase -- a ladder over the ranked even orders, each using the rotations the ones before it left (gcd of their orders) --
    as apply(harmonic_coefficients (Nq, n_phi), fxs_unknowns) -> coefficients * exp(i m rotation_phase)"""
    projection_orders = np.concatenate((np.arange(int(n_phi / 2) + 1), -1 * np.arange(int(n_phi / 2) + n_phi % 2)[:0:-1]))
    pos_orders = np.array(tuple(used_orders.keys()))
    order_mask = (pos_orders % 2 == 0) & (pos_orders != 0)
    harmonic_orders = pos_orders[order_mask]
    max_order = np.max(harmonic_orders)
    so_indices, so_orders, sorted_order_indices = so_ranking_2d(projection_vectors, pos_orders, radial_points, radial_high_pass)
    first_order = so_orders[0]
    remaining_rotations = first_order
    current_order = first_order
    free_orders_mask = True
    angle_coeffs, angles, order_indices, gcds = (), (), (), ()
    while remaining_rotations > 2:
        order_multiples = np.arange(current_order, max_order + 1, current_order)
        multiple_indices = np.where(np.isin(harmonic_orders, order_multiples))
        free_orders_mask = free_orders_mask * ~np.isin(sorted_order_indices, multiple_indices)
        if not free_orders_mask.any():
            break
        current_order_index = sorted_order_indices[free_orders_mask][0]
        current_order = harmonic_orders[current_order_index]
        gcd = np.gcd(remaining_rotations, current_order)
        n_independent_rotations = remaining_rotations / gcd
        smallest_angle = 2 * np.pi / n_independent_rotations
        smallest_angle_coeff = np.argmin((np.arange(1, n_independent_rotations) * current_order / gcd) % n_independent_rotations) + 1
        order_indices += (current_order_index,)
        angle_coeffs += (smallest_angle_coeff,)
        angles += (smallest_angle,)
        gcds += (gcd,)
        remaining_rotations = gcd

    def apply(harmonic_coefficients, fxs_unknowns):
        phases = (-1.j * np.log(np.asarray(fxs_unknowns)[order_mask])).real
        rotation_phase = 0
        for oi, angle, ac, g in zip(order_indices, angles, angle_coeffs, gcds):
            rotation_phase -= (phases[oi] // angle) * ac * angle / g
        return harmonic_coefficients * np.exp(1.j * projection_orders * rotation_phase)
    apply.ladder = (order_indices, angles, angle_coeffs, gcds)
    return apply


class ShrinkWrap2D(P.ShrinkWrap):
    def __init__(self, qs, shape, threshold=0.06):
        self.qgrid = np.broadcast_to(np.asarray(qs)[:, None], shape)
        self.default_sigma = np.pi / np.max(qs)                                  # fxs_Projections.py:189-190
        self._threshold = threshold
        self._sigma = self.default_sigma
        self.gaussian_values = P.gaussian_fourier_transformed_spherical(self.qgrid, self._sigma)


class Deg2InvariantDiff2D:
    """_generate_deg2_invariant_diff_2d, fxs_IO_methods.py:370-400: per used order sum |B_ref - B_m|^2 / sum |B_ref|^2 with
    B_m = I_m (x) I_m^* (fxs_invariant_tools.py:906-914), -1 where the reference invariant is zero; the zero order's reference is
    divided by the number of particles; no q weights and no mask (both are commented out upstream)"""

    def __init__(self, reference_invariant, used_orders, n_particles):
        self.ref = np.array(reference_invariant)[np.array(list(used_orders.keys())).astype(int)].copy()
        self.reference = self.ref.copy()
        self.norm = np.sum(self.ref * self.ref.conj(), axis=(1, 2)).real
        self.order_array = np.array(tuple(used_orders.values()))
        self.zero_id = used_orders[0]
        self.n_particles = n_particles

    def __call__(self, Im):
        Im = np.asarray(Im)                                            # (Nq, M + 1)
        Bm = np.array([v[:, None] * v[None, :].conj() for v in Im.T])[self.order_array]
        self.reference[self.zero_id] = self.ref[self.zero_id] / self.n_particles[0]
        diff = self.reference - Bm
        nd = np.sum((diff * diff.conj()).real, axis=(1, 2))
        err = np.full(len(self.norm), -1.0)
        nz = self.norm != 0
        err[nz] = nd[nz] / self.norm[nz]
        return err


class MTIP2D(OM.MTIP):
    def __init__(self, opt, data):
        self.opt = opt
        self.data = data
        g = opt['grid']
        M = int(g['max_order'])
        N = int(g['n_radial_points'])
        self.kappa = OM.reciprocity_coefficient(opt['fourier_transform'])
        max_q = g['max_q']
        if not isinstance(max_q, float):
            max_q = float(np.max(data['data_radial_points']))
        self.max_q = max_q
        mode = opt['fourier_transform']['type']
        r_top = float(np.max(P2.radial_grids_2d(max_q, N, self.kappa, mode)[0]))
        self.fp = P2.PolarFourierPair(N, M, max_q, self.kappa, weights_r_max=r_top, mode=mode)              # r_max = max(r_p), reconstruct.py:329
        self.sht = _RealHarmonic(self.fp.n_phi)
        self.shape = (N, self.fp.n_phi)
        self.rp = ReciprocalSetup2D(self.fp.qs, data, M, opt['projections']['reciprocal'])
        r_opt = opt['projections']['real']
        self.real_r = np.broadcast_to(self.fp.rs[:, None], self.shape)
        auto = None
        if r_opt['projections']['support']['initial_support']['type'] == 'auto_correlation':
            auto = self.autocorrelation_guess()
        self.real_pr = P.RealProjection(r_opt['projections'], self.real_r, opt['particle_radius'], auto)
        self.sw = ShrinkWrap2D(self.fp.qs, self.shape)
        self.integrator = PolarIntegrator(self.fp.rs, self.fp.phis)
        self.hio_considered = r_opt['HIO'].get('considered_projections', ['all']) or ['all']
        self.beta = r_opt['HIO']['beta'][0][0]
        em = opt['main_loop']['error']['methods']
        self.real_metrics = list(em['real']['calculate'])
        self.reciprocal_metrics = list(em['reciprocal']['calculate'])
        for name in self.reciprocal_metrics:
            if name not in ('deg2_invariant_l2_diff', 'l2_projection_diff'):
                raise NotImplementedError('2-D reciprocal metric %r' % (name,))
        self.inside_initial = em['real'].get('l2_projection_diff', {}).get('inside_initial_support', False)
        self.initial_mask = self.real_pr.initial_support
        gen = opt.get('general', {})
        self.real_error_mask = P.select_real_error_mask(self.shape, self.inside_initial, self.initial_mask, gen.get('cache_aware', True),
                                                        gen.get('L2_cache', 512))
        self.deg2_diff = None
        self._ranked_id = None
        self._so_apply = None
        if 'deg2_invariant_l2_diff' in self.reciprocal_metrics:
            self.deg2_diff = Deg2InvariantDiff2D(self.rp.deg2_invariants, self.rp.used_orders, self.rp.number_of_particles_list)
        self.results = {}
        self._init_sw_ramps()

    def autocorrelation_guess(self):
        """reconstruct.py:400-403, 421-423: ift(icht(pr.T)).real, pr = the (n_orders, Nq) `full_projection_matrices` (before the
        odd-order / average-intensity modifications), icht = the inverse of the REAL harmonic transform (the loop's
        'inverse_harmonic_transform', reconstruct.py:370), ift the polar inverse Fourier transform"""
        return self.fp.ift(self.sht.inverse_l(np.array(self.rp.full_projection_matrices).T).astype(complex)).real

    def output_modifier(self, pair):
        """assemble_output_modifier, reconstruct.py:721-755 with the 2-D operators (454, misk.py:295-312, fxs_Projections.py:1419-1432):
        (reciprocal, real) -> (reciprocal * phases, IFT(FT(real) * phases)), phases = exp(+i k.c), c = centre of mass of Re(real) by
        the PolarIntegrator, returned in polar coordinates with phi in [0, 2 pi)"""
        om = self.opt.get('output_density_modifiers', {})
        fix = bool(om.get('fix_orientation', False)) and self.rp.use_SO_freedom       # 746-752: shift_center + fix_orientation
        if not om.get('shift_to_center', False) and not fix:
            return pair
        recip, real = np.array(pair[0]), np.array(pair[1])
        ft = self.fp.ft(real)
        r, ph = np.meshgrid(self.fp.rs, self.fp.phis, indexing='ij')
        cart = np.stack((r * np.cos(ph), r * np.sin(ph)), -1)
        integral = self.integrator.integrate(real.real)
        if integral == 0:
            integral = 1
        c = np.array([self.integrator.integrate(cart[..., i] * real.real) for i in range(2)]) / integral
        phi_c = np.arctan2(c[1], c[0])
        center = np.array([np.hypot(c[0], c[1]), phi_c + 2 * np.pi if phi_c < 0 else phi_c])
        self.results['neg_center_pos'] = center
        cv = np.array([center[0] * np.cos(center[1]), center[0] * np.sin(center[1])])                     # spherical_to_cartesian(vector)
        q, pq = np.meshgrid(self.fp.qs, self.fp.phis, indexing='ij')
        phases = np.exp(1j * (q * np.cos(pq) * cv[0] + q * np.sin(pq) * cv[1]))                            # opposite_direction = True
        out = (recip * phases, self.fp.ift(ft * phases))
        if fix:
            # 736-741: complex harmonic transform of both, the rotation left free by the SO order taken out, back
            if self._so_apply is None:
                self._so_apply = remaining_so_projection_2d(self.rp.projection_matrices, self.rp.used_orders, self.rp.radial_points, self.fp.n_phi,
                                                            self.opt['projections']['reciprocal']['SO_freedom'].get('radial_high_pass', 0.2))
            unk = self.results['fxs_unknowns']
            out = tuple(P2.harmonic_inverse(self._so_apply(P2.harmonic_forward(a), unk)) for a in out)
        return out

    def _reciprocal_errors(self, F, F_new, Im):
        for name in self.reciprocal_metrics:
            if name == 'deg2_invariant_l2_diff':
                val = self.deg2_diff(Im)
            else:                                                     # l2_projection_diff: fxs_IO_methods.py:301-310 (see oracle/mtip.py)
                val = P.l2_rel_diff_error(self.integrator, np.array(F), np.array(F_new), True)
            self.errors['reciprocal'][name].append(val)

    def phasing_loop(self, rho0=None, rng=None, step_hook=None):
        if rho0 is None:
            rho0 = self.density_guess(rng if rng is not None else np.random.default_rng())
        state = self.create_initial_state(np.array(rho0, dtype=complex))
        initial_densities = tuple(d.copy() for d in state['best_density_pair'])
        initial_mask = state['mask'].copy()
        iterations = []
        for lid, name in enumerate(self.opt['main_loop']['sub_loops']['order']):
            state, it = self.run_sub_loop(name, lid, state, step_hook)
            iterations.append(it)
        best = self.output_modifier(state['best_density_pair'])
        last = self.output_modifier(state['density_pair_history'][-1])
        I_last = P2.real_harmonic_forward(np.abs(self.fp.ft(np.array(last[1]))) ** 2)                      # calc_deg2_invariant, 757-765, 993
        last_deg2 = np.array([v[:, None] * v[None, :].conj() for v in I_last.T])
        masked = np.array(self.rp.projection_matrices)
        masked[~self.rp.radial_mask] = 0                                                                     # 997-1001
        n_steps = len(self.errors['main'])
        grids = {'real_grid': np.stack(np.meshgrid(self.fp.rs, self.fp.phis, indexing='ij'), -1),
                 'reciprocal_grid': np.stack(np.meshgrid(self.fp.qs, self.fp.phis, indexing='ij'), -1)}
        err = {'main': np.array(self.errors['main']), 'real': {k: np.array(v) for k, v in self.errors['real'].items()},
               'reciprocal': {k: np.array(v) for k, v in self.errors['reciprocal'].items()}}
        return {'real_density': best[1], 'last_real_density': last[1], 'reciprocal_density': best[0], 'last_reciprocal_density': last[0],
                'final_error': state['best_error'], 'initial_density': initial_densities[1], 'initial_support': initial_mask,
                'error_dict': err, 'support_mask': state['best_mask'], 'last_support_mask': state['mask'],
                'loop_iterations': np.sum(iterations) + 1, 'fxs_unknowns': self.results.get('fxs_unknowns'),
                'n_particles': np.full((n_steps, 1), self.rp.number_of_particles_list[0]), 'n_particles_gradients': np.array([]),
                'n_particles_fraction': np.array([]), 'grid_pair': grids, 'projection_matrices': masked, 'last_deg2_invariant': last_deg2}
