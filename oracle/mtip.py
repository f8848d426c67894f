"""Oracle MTIP phasing loop (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Restates ``xframe/projects/fxs/reconstruct.py``:
* operator assembly 303-485, sketches ``MTIP_start`` 518-528, ``HIO``/``ER`` 576-583,
  ``*_ft_stab`` 584-593, ``SW`` 598-605, ``calc_deg2_invariant`` 757-765
* loop state machine ``assemble_phasing_loop`` 768-1036 (beta ramp 911, history 924-926,
  best tracking 934-938, SW / enforce_initial_support 877-885, ft_stab switch 836-850,
  non-FXS variants 899-904, best reselection 945-949)
* initial state 957-979, density guess 1115-1210, SW ramps 1212-1258, output dict 980-1022.

Settings are the *resolved* ``settings.project`` tree as a plain nested dict with the
reference's key names (``settings/reconstruct/default_0.01.yaml``).
"""
import copy as _copy
import numpy as np
from .sht import SHT
from .fourier import FourierPair, SphericalIntegrator
from . import projections as P


def default_settings():
    """Resolved defaults of settings/reconstruct/default_0.01.yaml (3-D), tutorial.yaml overrides
    are applied by the caller."""
    return {
        'dimensions': 3,
        'structure_name': 'default_structure',
        'particle_radius': 150,
        'grid': {'max_q': False, 'max_order': 63, 'n_phi': 0, 'n_theta': 0, 'n_radial_points': 128},
        'fourier_transform': {'type': 'midpoint', 'reciprocity_coefficient': 2.0,
                              'allow_weight_calculation': True, 'allow_weight_saving': True},
        'density_guess': {'type': 'bump', 'bump': {'slope': 0.3}, 'radius': 150,
                          'amplitude_function': 'random', 'random': {'SNR': 2}},
        'projections': {
            'real': {
                'projections': {
                    'apply': ['support', 'value_threshold', 'assert_real'],
                    'value_threshold': {'threshold': [0, False]},
                    'limit_imag': {'threshold': 2},
                    'support': {'initial_support': {'type': 'max_radius', 'max_radius': 150,
                                                    'auto_correlation': {'threshold': 0.01}},
                                'enforce_initial_support': {'apply': True, 'if_error_bigger_than': 6e-3}},
                },
                'shrink_wrap': {'sigmas': [[False, [False, False], False], [False, [False, False], False]],
                                'thresholds': [[0.08, [0, 0], 0], [0.08, [0, 0], 0]]},
                'HIO': {'beta': [[0.5, 0.4, -1 / 700, 1600], [0.01, 0.002, -1 / 200, 200]],
                        'considered_projections': ['all']},
            },
            'reciprocal': {
                'number_of_particles': {'initial': 1.0, 'estimate': False},
                'regrid': {'interpolation': 'cubic'},
                'used_order_ids': np.arange(64),
                'odd_orders_to_0': True,
                'use_averaged_intensity': True,
                'q_mask': {'type': 'none'},
                'SO_freedom': {'use': False},
            },
        },
        'output_density_modifiers': {'shift_to_center': False},
        'main_loop': {
            'error': {'methods': {
                'real': {'calculate': ['l2_projection_diff'],
                         'l2_projection_diff': {'inside_initial_support': True}},
                'reciprocal': {'calculate': [], 'deg2_invariant_l2_diff': {'order': 2}},
                'main': {'metrics': {'real': ['l2_projection_diff'], 'reciprocal': []}, 'type': 'mean'}}},
            'sub_loops': {
                'order': ['main', 'refinement'],
                'main': {'methods': {'HIO': {'iterations': 60, 'ft_stab': True},
                                     'ER': {'iterations': 40, 'ft_stab': True}, 'SW': 1},
                         'order': ['HIO', 'SW', 'ER'], 'iterations': 5,
                         'best_density_not_in_first_n_iterations': np.inf},
                'refinement': {'methods': {'ER': {'iterations': 100, 'ft_stab': True}, 'SW': 1},
                               'order': ['SW', 'ER'], 'iterations': 2,
                               'best_density_not_in_first_n_iterations': np.inf},
            },
        },
        'GPU': {'use': True, 'n_gpu_workers': 1},
        'multi_process': {'use': True, 'n_parallel_reconstructions': False},
        'profiling': {'enable': False, 'reconstruction_process_id': 1, 'gpu_worker_id': -1},
    }


def deep_update(base, upd):
    out = _copy.deepcopy(base)
    for k, v in upd.items():
        if isinstance(v, dict) and isinstance(out.get(k), dict):
            out[k] = deep_update(out[k], v)
        else:
            out[k] = _copy.deepcopy(v)
    return out


def reciprocity_coefficient(ft_opt):
    """misk.py:387-394 with mathLibrary.py:1178-1182."""
    pi_in_q = ft_opt.get('pi_in_q', None)
    if isinstance(pi_in_q, bool):
        return np.pi if pi_in_q else 1 / 2
    return ft_opt.get('reciprocity_coefficient', np.pi)


class MTIP:
    def __init__(self, opt, data):
        self.opt = opt
        self.data = data
        g = opt['grid']
        L = int(g['max_order'])
        N = int(g['n_radial_points'])
        self.kappa = reciprocity_coefficient(opt['fourier_transform'])
        max_q = g['max_q']
        if not isinstance(max_q, float):                       # reconstruct.py:258-261
            max_q = float(np.max(data['data_radial_points']))
        self.max_q = max_q
        self.sht = SHT(L, g.get('n_theta', 0), g.get('n_phi', 0))
        self.fp = FourierPair(self.sht, N, max_q, self.kappa, opt['fourier_transform']['type'])
        self.shape = (N, self.sht.n_theta, self.sht.n_phi)
        self.rp = P.ReciprocalProjection(self.fp.qs, data, L, opt['projections']['reciprocal'])
        r_opt = opt['projections']['real']
        self.real_r = np.broadcast_to(self.fp.rs[:, None, None], self.shape)
        auto = None
        if r_opt['projections']['support']['initial_support']['type'] == 'auto_correlation':
            auto = self.autocorrelation_guess()
        self.real_pr = P.RealProjection(r_opt['projections'], self.real_r, opt['particle_radius'], auto)
        self.sw = P.ShrinkWrap(self.fp.qs, self.shape)
        self.integrator = SphericalIntegrator(self.fp.rs, self.sht.n_theta)
        self.hio_considered = r_opt['HIO'].get('considered_projections', ['all']) or ['all']
        self.beta = r_opt['HIO']['beta'][0][0]
        em = opt['main_loop']['error']['methods']
        self.real_metrics = list(em['real']['calculate'])
        self.reciprocal_metrics = list(em['reciprocal']['calculate'])
        self.inside_initial = em['real'].get('l2_projection_diff', {}).get('inside_initial_support', False)
        self.initial_mask = self.real_pr.initial_support
        gen = opt.get('general', {})
        self.real_error_mask = P.select_real_error_mask(self.shape, self.inside_initial, self.initial_mask,
                                                        gen.get('cache_aware', True), gen.get('L2_cache', 512))
        self.deg2_diff = None
        self._ranked_id = None
        if 'deg2_invariant_l2_diff' in self.reciprocal_metrics or 'deg2_ranked_invariant_l2_diff' in self.reciprocal_metrics:
            inv_mask = self.rp.radial_mask[:, :, None] * self.rp.radial_mask[:, None, :]
            self.deg2_diff = P.Deg2InvariantDiff(self.rp.deg2_invariants, self.rp.used_orders,
                                                 self.rp.number_of_particles, inv_mask)
        self.results = {}
        self._init_sw_ramps()

    # -- reconstruct.py:400-420
    def autocorrelation_guess(self):
        pr_padded = []
        for l, p in enumerate(self.rp.full_projection_matrices):
            n_ms = 2 * l + 1
            if p.shape[1] != n_ms:
                pp = np.zeros((p.shape[0], n_ms), dtype=p.dtype)
                pp[:, :p.shape[1]] = p
                pr_padded.append(pp)
            else:
                pr_padded.append(p)
        return self.fp.ift(self.sht.inverse_l(pr_padded)).real

    # -- reconstruct.py:1212-1258
    def _init_sw_ramps(self):
        sw_opt = self.opt['projections']['real']['shrink_wrap']
        order = self.opt['main_loop']['sub_loops']['order']
        self.sw_sigma_ramps, self.sw_thresh_ramps = [], []
        for lid in range(len(order)):
            sig = sw_opt['sigmas'][lid] if len(sw_opt['sigmas']) - 1 >= lid else False
            if not isinstance(sig, (list, tuple)):
                sig = [sig]
            self.sw_sigma_ramps.append(P.LinearRamp(*sig, default_start=self.sw.default_sigma,
                                                    default_stop=self.sw.default_sigma))
            thr = sw_opt['thresholds'][lid] if len(sw_opt['thresholds']) - 1 >= lid else 0.1
            if not isinstance(thr, (list, tuple)):
                thr = [thr]
            self.sw_thresh_ramps.append(P.LinearRamp(*thr))

    def update_shrink_wrap(self, iteration, loop_number):
        r = self.sw_sigma_ramps[loop_number]
        if not r.undefined:
            self.sw.gaussian_sigma = r(iteration)
        t = self.sw_thresh_ramps[loop_number]
        if not t.undefined:
            self.sw.threshold = t(iteration)

    # -- sketch MTIP_start, reconstruct.py:518-528
    def mtip_start(self, F):
        Fc = np.array(F)
        I = P.square_grid(F)
        Ilm = self.sht.forward_l(I)
        unknowns = self.rp.approximate_unknowns(Ilm)
        self.results['fxs_unknowns'] = unknowns
        Ilm_new = self.rp.mtip_projection(Ilm, unknowns)
        I_new = self.sht.inverse_l(Ilm_new)
        F_new = self.rp.project_to_modified_intensity(Fc, I, I_new)
        self.results.setdefault('n_particles', []).append(list(self.rp.number_of_particles))
        self._reciprocal_errors(Fc, F_new, Ilm)
        return F_new

    # -- sketch MTIP_start_non_FXS, 530-535
    def mtip_start_non_fxs(self, F):
        Fc = np.array(F)
        I = P.square_grid(F)
        F_new = self.rp.project_to_fixed_intensity(Fc, I)
        self.results.setdefault('n_particles', []).append(list(self.rp.number_of_particles))
        return F_new

    def _reciprocal_errors(self, F, F_new, Ilm):
        for name in self.reciprocal_metrics:
            if name == 'deg2_invariant_l2_diff':
                val = self.deg2_diff(Ilm)
            elif name == 'deg2_ranked_invariant_l2_diff':
                # fxs_IO_methods.py:330-366: the entry of the per-order metric at the best ranked even order, or at the order the option names
                if self._ranked_id is None:
                    order = self.opt['main_loop']['error']['methods']['reciprocal'].get(name, {}).get('order', False)
                    if isinstance(order, (int, np.integer)) and not isinstance(order, (bool, np.bool_)):
                        self._ranked_id = self.rp.used_orders[int(order)]
                    else:
                        orders = np.array(list(self.rp.used_orders.keys())).astype(int)
                        self._ranked_id = P.rank_projection_matrices_3d(self.rp.projection_matrices, orders, self.rp.radial_points)[0][0]
                val = self.deg2_diff(Ilm)[self._ranked_id]
            elif name == 'l2_projection_diff':
                # 301-310: the cache-aware branch asks for type 'reziprocal' -> the REAL grid's integrator (131-140); the plain branch takes
                # the reciprocal grid's, whose radial points are proportional -- the same ratio; mask True in both (shell N - 2 drops out)
                val = P.l2_rel_diff_error(self.integrator, np.array(F), np.array(F_new), True)
            else:
                raise NotImplementedError(name)
            self.errors['reciprocal'][name].append(val)

    # -- sketches HIO/ER(+_non_FXS)(+_ft_stab), 576-593
    def step(self, method, rho, ft_stab):
        fxs = '_non_FXS' not in method
        F = self.fp.ft(rho)
        F_new = self.mtip_start(F) if fxs else self.mtip_start_non_fxs(F)
        rho_p = self.fp.ift(F_new)
        if ft_stab:
            rho_rt = self.fp.ift(F)
            rho_p = P.add_above_zero_index(rho_p, rho - rho_rt)
        w = np.array(rho_p)
        proj_out = self.real_pr.projection(rho_p)
        if method.startswith('HIO'):
            rho_new = P.hybrid_input_output(w, proj_out, rho, self.beta, self.hio_considered)
        else:
            rho_new = P.error_reduction(w, proj_out, rho)
        for name in self.real_metrics:
            if name == 'l2_projection_diff':
                val = P.l2_rel_diff_error(self.integrator, w, proj_out[0], self.real_error_mask)
            else:
                raise NotImplementedError(name)
            self.errors['real'][name].append(val)
        return F_new, rho_new

    # -- sketch SW, 598-605
    def sw_step(self, rho):
        a = P.abs_value(np.array(rho))
        c = self.fp.ift(self.sw.multiply_with_ft_gaussian(self.fp.ft(a)))
        return self.sw.get_new_mask(c)

    # -- generate_main_error_routine, fxs_IO_methods.py:746-765
    def main_error(self):
        em = self.opt['main_loop']['error']['methods']['main']
        method = {'mean': np.mean, 'min': np.min, 'max': np.max, 'prod': np.prod}[em['type']]
        try:
            vals = [self.errors['real'][n][-1] for n in em['metrics']['real']]
            vals += [self.errors['reciprocal'][n][-1] for n in em['metrics']['reciprocal']]
            return method(np.array(vals))
        except IndexError:
            return -1

    # -- generate_density_guess_method, reconstruct.py:1115-1210
    def density_guess(self, rng):
        dg = self.opt['density_guess']
        radius = dg['radius']
        if isinstance(radius, bool):
            radius = self.opt['particle_radius']
        if radius < 0:
            radius = np.max(self.fp.rs)
        if dg['type'] == 'low_resolution_autocorrelation':
            # 1175-1205: IFT of the inverse harmonic transform of the (modified) projection matrices, negative values cut,
            # times the random amplitude and a bump of slope 0.1 over the particle radius
            pr = [np.array(p) for p in self.rp.projection_matrices]
            ac = self.fp.ift(self.sht.inverse_l(pr)).real
            ac[ac < 0] = 0
            amp = (1 + 1 / dg['random']['SNR'] * rng.random(self.shape)).astype(complex)
            density = ac * amp
            density[density < 0] = 0
            pr_radius = self.opt['particle_radius']
            density = density * P.get_test_function([-pr_radius, pr_radius], 0.1)(np.array(self.real_r))
            total_sq = self.integrator.integrate((density * density.conj()).real)
            return density * np.sqrt(self.rp.integrated_intensity / total_sq)
        if dg['type'] == 'ball':
            # 1136-1153 with get_disk_function / get_shape_function (mathLibrary.py:124-167): amplitude inside r < radius
            r = np.array(self.real_r)
            inside = r < radius
            density = np.zeros(self.shape)
            density[inside] = 1 + 1 / dg['random']['SNR'] * rng.random(int(inside.sum()))
        else:
            assert dg['type'] == 'bump'
            amp = 1 + 1 / dg['random']['SNR'] * rng.random(self.shape)
            bump = P.get_test_function([-radius, radius], dg['bump']['slope'])
            density = amp * bump(np.array(self.real_r))
        total_sq = self.integrator.integrate((density * density.conj()).real)
        density = density * np.sqrt(self.rp.integrated_intensity / total_sq)
        return density.astype(complex)

    # -- create_initial_state, 957-979
    def create_initial_state(self, rho0):
        F0 = self.fp.ft(rho0)
        rho0 = self.fp.ift(F0)
        hl = self.opt['main_loop'].get('history_length', 3)
        pairs = ((F0, rho0),) * hl
        self.errors = {'real': {n: [] for n in self.real_metrics},
                       'reciprocal': {n: [] for n in self.reciprocal_metrics}, 'main': []}
        self.results['errors'] = self.errors
        init_sup = self.real_pr.initial_support
        return {'density_pair_history': pairs, 'error_dict': self.errors, 'mask': init_sup,
                'best_density_pair': pairs[-1], 'best_error': np.inf, 'best_iteration': 0,
                'best_mask': init_sup}

    # -- generate_loop_method / loop, 814-952
    def run_sub_loop(self, loop_name, loop_number, state, step_hook=None):
        loop_opt = self.opt['main_loop']['sub_loops'][loop_name]
        hio_opt = self.opt['projections']['real']['HIO']
        order = loop_opt['order']
        methods = {}
        for key in order:
            mo = loop_opt['methods'][key]
            if isinstance(mo, dict):
                methods[key] = {'iterations': mo.get('iterations', 0), 'options': mo}
            else:
                methods[key] = {'iterations': mo, 'options': {}}
        if len(hio_opt['beta']) - 1 < loop_number:
            hio_beta = [0.5, 0.5, -1 / 700, 1600]
        else:
            hio_beta = hio_opt['beta'][loop_number]
        ramp = P.ExponentialRamp(*hio_beta)
        eis_opt = self.opt['projections']['real']['projections']['support']['enforce_initial_support']
        limit = [eis_opt['if_error_bigger_than']] if eis_opt['apply'] else [np.inf]

        if 'SW' in methods:
            self.update_shrink_wrap(0, loop_number)
        error_dict = state['error_dict']
        # reconstruct.py:859: `hist` is a local that is re-read from the state only at the top of every phasing step (913);
        # SW_center (893) and the *_non_FXS intensity (901) read it as it was left there, i.e. the history BEFORE the most
        # recent step (or the loop's initial history when no step has run in this call yet)
        hist = state['density_pair_history']
        eis_list = state.get('enforce_initial_support_list', [])
        iteration = 0
        step = 0
        latest_intensity = False
        sw_step = 0
        for iteration in range(1, loop_opt['iterations'] + 1):
            for key in order:
                repeats = methods[key]['iterations']
                popt = methods[key]['options']
                if key == 'SW':
                    support = self.sw_step(state['density_pair_history'][-1][1])
                    enforce = error_dict['main'][-1:] > limit
                    eis_list.append(enforce)
                    self.real_pr.enforce_initial_support = enforce
                    self.real_pr.support = support
                    state['mask'] = self.real_pr.support
                    sw_step += 1
                    self.update_shrink_wrap(sw_step, loop_number)
                    continue
                if key == 'SW_center':
                    # reconstruct.py:606-613, 886-897.  Two quirks of the reference are restated literally:
                    #  * the sketch's last stage [(0,1,2,3), ['calculate_support_mask','id','id']] feeds the one-argument
                    #    get_new_mask (fxs_Projections.py:247) input 0 and the two 'id's inputs 1 and 2, i.e. the process
                    #    returns (support, copy(rho), FT(rho)); the loop unpacks that as (support, ft_density, density), so
                    #    the pair appended to the history is (rho, FT(rho)): the "real" half of the last pair is FT(rho);
                    #  * the new history is built from the stale `hist` (893): the pair of the most recent step is dropped.
                    enforce = error_dict['main'][-1] > limit
                    self.real_pr.enforce_initial_support = enforce
                    eis_list.append(enforce)
                    for _ in range(repeats):
                        rho = np.array(state['density_pair_history'][-1][1])
                        support = self.sw_step(rho)
                        self.real_pr.support = support
                        state['mask'] = self.real_pr.support
                        state['density_pair_history'] = hist[1:] + ((np.array(rho), self.fp.ft(np.array(rho))),)
                        sw_step += 1
                        self.update_shrink_wrap(sw_step, loop_number)
                    continue
                if key in ('ER_non_FXS', 'HIO_non_FXS'):
                    if isinstance(latest_intensity, bool):
                        latest_intensity = np.abs(hist[-1][0]).real          # stale hist: reconstruct.py:901
                        self.rp.fixed_intensity = latest_intensity
                else:
                    latest_intensity = False
                ft_stab = self._change_to_ft_stab(popt, key, eis_list)
                for _ in range(repeats):
                    self.beta = ramp.eval(step)
                    hist = state['density_pair_history']
                    new_pair = self.step(key, hist[-1][1], ft_stab)
                    copied = tuple(np.array(a) for a in new_pair)
                    state['density_pair_history'] = hist[1:] + (copied,)
                    main_error = self.main_error()
                    error_dict['main'].append(main_error)
                    if state['best_error'] > main_error:
                        state['best_error'] = main_error
                        state['best_density_pair'] = copied
                        state['best_iteration'] = iteration
                        state['best_mask'] = state['mask']
                    if step_hook is not None:
                        step_hook(key, step, copied, main_error)
                    step += 1
        if state['best_iteration'] > loop_opt.get('best_density_not_in_first_n_iterations', np.inf):
            state['density_pair_history'] = state['density_pair_history'][1:] + (state['best_density_pair'],)
            self.real_pr.support = state['best_mask']
            state['mask'] = state['best_mask']
        state['enforce_initial_support_list'] = eis_list
        return state, iteration

    @staticmethod
    def _change_to_ft_stab(popt, name, eis_list):
        """reconstruct.py:836-850."""
        apply = False
        if name[-8:] != '_ft_stab' and 'ft_stab' in popt:
            v = popt['ft_stab']
            if isinstance(v, bool):
                apply = v
            elif v == 'link_to_enforce_initial_support':
                delay = max(int(popt['link_to_enforce_initial_support']['delay']), 1)
                if len(eis_list) >= delay:
                    apply = not (np.array(eis_list[-delay:]) == True).any()  # noqa: E712
        return apply

    # -- main_loop + generate_output, 980-1035
    def output_modifier(self, pair):
        """assemble_output_modifier, reconstruct.py:721-755 (3-D: identity, or 'shift_center' 728-734): the pair
        (reciprocal, real) becomes (reciprocal * phases, IFT(FT(real) * phases)) with phases = exp(+i k.c), c the centre
        of mass of Re(real); the centre is kept as results['neg_center_pos']."""
        if not self.opt.get('output_density_modifiers', {}).get('shift_to_center', False):
            return pair
        recip, real = np.array(pair[0]), np.array(pair[1])
        ft = self.fp.ft(real)
        center = P.calc_center(self.fp.rs, self.sht.n_theta, self.fp.grid.real_grid(), real)
        self.results['neg_center_pos'] = center
        phases = P.shift_phases(self.fp.grid.reciprocal_grid(), center, opposite_direction=True)
        return (recip * phases, self.fp.ift(ft * phases))

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
        F_last = self.fp.ft(last[1])
        last_deg2 = P.harmonic_coeff_to_deg2_invariants_3d(self.sht.forward_l(P.square_grid(F_last)))
        err = {'main': np.array(self.errors['main']),
               'real': {k: np.array(v) for k, v in self.errors['real'].items()},
               'reciprocal': {k: np.array(v) for k, v in self.errors['reciprocal'].items()}}
        masked_pm = []
        for mask, matrix in zip(self.rp.radial_mask, self.rp.projection_matrices):
            tmp = np.array(matrix)
            tmp[~mask] = 0
            masked_pm.append(tmp)
        return {'real_density': best[1], 'last_real_density': last[1],
                'reciprocal_density': best[0], 'last_reciprocal_density': last[0],
                'final_error': state['best_error'], 'initial_density': initial_densities[1],
                'initial_support': initial_mask, 'error_dict': err,
                'support_mask': state['best_mask'], 'last_support_mask': state['mask'],
                'loop_iterations': np.sum(iterations) + 1,
                'fxs_unknowns': self.results.get('fxs_unknowns'),
                'n_particles': np.array(self.results.get('n_particles', [])),
                'n_particles_gradients': np.array([]), 'n_particles_fraction': np.array([]),
                'grid_pair': {'real_grid': self.fp.grid.real_grid(), 'reciprocal_grid': self.fp.grid.reciprocal_grid()},
                'projection_matrices': masked_pm, 'last_deg2_invariant': last_deg2}
