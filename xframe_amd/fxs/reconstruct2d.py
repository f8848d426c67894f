"""The 2-D (polar) phasing loop of ``fxs reconstruct`` (``dimensions: 2``) on the MI355X -- host mirror of the `dimensions == 2`
branches of ``xframe/projects/fxs/reconstruct.py`` (263-266, 347-350, 421-423, 1126-1129) and
``projectLibrary/fxs_Projections.py`` (473-476, 506-511, 631-637, 651-662, 679-714, 189-203) around the loop of 814-952.

A 2-D restart is 260 KB at 128 x 129, so unlike the 3-D engine the loop is orchestrated from the host, as the reference's is:
every phasing step (Fourier pair, real harmonic transform of |F|^2, unknowns + projection, modulus replacement, inverse transform
with the ft_stab add-back, real-space projection + HIO / ER, error sums) is ONE call into the device operators for the whole
batch of restarts (`mtip2d_op_step`, csrc/k_polar2d.hip), the shrink-wrap another (`mtip2d_op_shrinkwrap`); schedule, beta and
shrink-wrap ramps, support bookkeeping, history and best tracking are the host logic of the reference.  Pinned by fixture G20 (the
reference's own 2-D `MTIP` run) and, for the loop's sub-variants -- `SW_center`, `HIO_non_FXS` / `ER_non_FXS`, the reciprocal metrics
`deg2_invariant_l2_diff` / `l2_projection_diff` (and main errors over them), the auto-correlation initial support, `shift_to_center`
-- and `SO_freedom` with the `fix_orientation` output modifier, by `tests/golden/mtip2d_variants_N12_M6.npz` (the reference's own
2-D runs of each); the radial rules `trapz`, `gauss`, `Zernike` are host weight tables for the same device contraction (fixture G23 from
the reference's own functions).  Not built for 2-D: the other reciprocal metrics; the `low_resolution_autocorrelation` guess raises upstream in 2-D (reconstruct.py:1186 iterates over
`low_resolution_intensity_coefficients`, which is False for dimensions == 2) and raises here."""
import numpy as np

from . import hostsetup as hs
from .polar2d import Engine2D
from .settings import reciprocity_coefficient, resolve


def polar_integrator_weights(rs, phis):
    """PolarIntegrator.integrate (mathLibrary.py:1254-1262) as weights: trapezoid over the phi samples (the circle is not closed) and
    over r with the weight r"""
    def trapz_w(x):
        w = np.zeros(len(x))
        d = np.diff(x)
        w[:-1] += d / 2
        w[1:] += d / 2
        return w
    return (trapz_w(np.asarray(rs)) * np.asarray(rs))[:, None] * trapz_w(np.asarray(phis))[None, :]


def so_order_ranking(vectors, orders, qs, radial_high_pass):
    """rank_projection_matrix_orders_2d (fxs_Projections.py:933-962): positions (among the used orders) of the even non-zero orders,
    strongest first by mean_q |v_m(q)| q above the radial high pass; also those orders and the sorting permutation"""
    qs, orders = np.asarray(qs), np.asarray(orders)
    start = int((len(qs) - 1) * radial_high_pass)
    candidates = np.flatnonzero((orders % 2 == 0) & (orders != 0))
    strength = np.mean(np.abs(np.asarray(vectors)[candidates, start:] * qs[None, start:]), axis=1)
    perm = np.argsort(strength)[::-1]
    return candidates[perm], orders[candidates[perm]], perm


class RemainingRotation2D:
    """generate_remaining_SO_projection_2D (fxs_Projections.py:1022-1095), the `fix_remaining_SO_freedom` operator of the output
    modifier: fixing the phase of the strongest order m0 leaves m0 rotations free; the next strongest order that is not a multiple of
    the ones used so far takes gcd-many of them away, and so on until at most two are left.  The rotation chosen from the unknowns of
    the last step multiplies the complex harmonic coefficients (Nq, n_phi) by exp(i m phase)."""

    def __init__(self, vectors, used_orders, qs, n_phi, radial_high_pass):
        self.m = np.concatenate((np.arange(n_phi // 2 + 1), -np.arange(n_phi // 2 + n_phi % 2)[:0:-1]))    # FFT order of the columns
        orders = np.array(tuple(used_orders.keys()))
        self.even = (orders % 2 == 0) & (orders != 0)
        even_orders = orders[self.even]
        _, ranked_orders, perm = so_order_ranking(vectors, orders, qs, radial_high_pass)
        left, current = ranked_orders[0], ranked_orders[0]
        free = np.ones(len(perm), bool)
        self.rungs = []                                               # (index among the even orders, angle, coefficient, gcd)
        while left > 2:
            multiples = np.arange(current, even_orders.max() + 1, current)
            free &= ~np.isin(perm, np.flatnonzero(np.isin(even_orders, multiples)))
            if not free.any():
                break
            idx = perm[free][0]
            current = even_orders[idx]
            g = np.gcd(left, current)
            n_rot = left / g
            coeff = np.argmin((np.arange(1, n_rot) * current / g) % n_rot) + 1
            self.rungs.append((idx, 2 * np.pi / n_rot, coeff, g))
            left = g

    def phase(self, unknowns):
        ph = (-1j * np.log(np.asarray(unknowns)[self.even])).real
        total = 0.0
        for idx, angle, coeff, g in self.rungs:
            total -= (ph[idx] // angle) * coeff * angle / g
        return total

    def __call__(self, coefficients, unknowns):
        return coefficients * np.exp(1j * self.m * self.phase(unknowns))


class ReciprocalSetup2D:
    """ReciprocalProjection.__init__ for dimensions == 2 (fxs_Projections.py:471-537) on the host"""

    def __init__(self, qs, data, max_order, opt):
        q_d = np.asarray(data['data_radial_points'], dtype=float)
        aint = np.asarray(getattr(data['average_intensity'], 'data', data['average_intensity']), dtype=float)
        self.qs = np.asarray(qs, dtype=float)
        self.integrated_intensity = (q_d[1] - q_d[0]) * np.sum(aint * q_d) * 2 * np.sqrt(np.pi)              # 473-474
        orders = np.arange(max_order + 1)
        used_ids = np.asarray(opt.get('used_orders', opt['used_order_ids']))
        self.used_orders = {int(o): int(i) for o, i in zip(orders, used_ids)}
        self.number_of_particles = float(opt['number_of_particles']['initial'])
        pm = np.array(np.asarray(data['data_projection_matrices'])[list(self.used_orders.values())], dtype=complex)
        if q_d.shape != self.qs.shape or not (q_d == self.qs).all():                                        # 642-662: one vector at a time
            interp = opt['regrid']['interpolation']
            aint = hs._regrid(aint, q_d, self.qs, interp)
            pm = np.array([hs._regrid(v, q_d, self.qs, interp) for v in pm])
        self.average_intensity = aint
        self.full_projection_matrices = np.zeros((max_order + 1, len(self.qs)), dtype=complex)              # 508-511
        for oid, p in zip(used_ids, pm):
            self.full_projection_matrices[int(oid)] = p
        proj = pm.copy()                                                                                    # 679-714, 2-D branches
        keys = np.array(tuple(self.used_orders))
        if opt.get('odd_orders_to_0', False):
            proj[keys % 2 == 1, :] = 0
        if opt.get('use_averaged_intensity', False) and 0 in self.used_orders:
            proj[self.used_orders[0]] = aint.astype(complex)
        self.projection_matrices = proj
        self.deg2_invariants = np.array([v[:, None] * v[None, :].conj() for v in proj])                     # 631-633, fxs_invariant_tools.py:906-914
        self.radial_mask = hs.reciprocal_radial_mask(self.qs, q_d, max_order, opt.get('q_mask', None), data)
        so = opt.get('SO_freedom', {})
        self.use_SO_freedom = bool(so.get('use', False))
        self.radial_high_pass = so.get('radial_high_pass', 0.2)
        # generate_approximate_unknowns, 744-750: the unknown of the strongest even order is 1 in every step
        self.so_position = int(so_order_ranking(proj, keys, self.qs, self.radial_high_pass)[0][0]) if self.use_SO_freedom else None


class MTIP2D:
    """the 2-D loop for a batch of restarts; settings as for the 3-D worker (`dimensions: 2`), data = the 2-D invariants
    (`data_projection_matrices` (n_orders, Nq), `average_intensity`, `data_radial_points`)"""

    def __init__(self, settings, data, n_restarts=1, initial_densities=None, seeds=None, device=0, lib_path=None):
        opt = self.opt = resolve(settings)
        if opt.get('dimensions', 3) != 2:
            raise ValueError('MTIP2D is the dimensions == 2 loop')
        g = opt['grid']
        self.N, self.M = int(g['n_radial_points']), int(g['max_order'])
        mode = opt['fourier_transform']['type']                       # midpoint, trapz, gauss, Zernike (polar2d.polar_raw_weights raises otherwise)
        kappa = reciprocity_coefficient(opt['fourier_transform'])
        max_q = g['max_q']
        if not isinstance(max_q, float):
            max_q = float(np.max(data['data_radial_points']))
        self.B = int(n_restarts)
        r_top = float(np.max(hs.radial_grids(max_q, self.N, kappa, mode)[0]))
        self.engine = e = Engine2D(self.N, self.M, max_q, kappa, n_batch=self.B, device=device, lib_path=lib_path,
                                   weights_r_max=r_top, mode=mode)                                          # r_max = max(r_p), reconstruct.py:329
        self.shape = e.shape
        self.rsetup = rs_ = ReciprocalSetup2D(e.qs, data, self.M, opt['projections']['reciprocal'])
        e.set_projection(rs_.projection_matrices, rs_.used_orders, rs_.radial_mask, rs_.number_of_particles)
        e.set_so_freedom(rs_.so_position)
        popt = opt['projections']['real']['projections']
        considered = opt['projections']['real']['HIO'].get('considered_projections', ['all'])
        e.set_real_constraints(*hs.real_constraint_flags(popt, considered))
        self.initial_support = np.ascontiguousarray(self._initial_support(popt['support']['initial_support']))
        em = opt['main_loop']['error']['methods']
        self.reciprocal_metrics = list(em['reciprocal'].get('calculate', []))
        for name in self.reciprocal_metrics:
            if name not in ('deg2_invariant_l2_diff', 'l2_projection_diff'):
                raise NotImplementedError('2-D reciprocal metric %r' % (name,))
        if list(em['real'].get('calculate', [])) != ['l2_projection_diff']:
            raise NotImplementedError('2-D real error metrics other than l2_projection_diff')
        main = em.get('main', {'metrics': {'real': ['l2_projection_diff'], 'reciprocal': []}, 'type': 'mean'})
        self.main_real = list(main['metrics'].get('real', []))
        self.main_reciprocal = list(main['metrics'].get('reciprocal', []))
        if any(n != 'l2_projection_diff' for n in self.main_real) or any(n not in self.reciprocal_metrics for n in self.main_reciprocal):
            raise NotImplementedError('2-D main error over metrics that are not calculated')
        self.main_type = main.get('type', 'mean')
        if 'deg2_invariant_l2_diff' in self.reciprocal_metrics:
            ids = np.array(list(rs_.used_orders.keys())).astype(int)
            self._deg2_ref = rs_.deg2_invariants[ids].copy()
            self._deg2_norm = np.sum(self._deg2_ref * self._deg2_ref.conj(), axis=(1, 2)).real
        # l2_projection_diff (fxs_IO_methods.py:97-128) with the PolarIntegrator; which mask the reference really uses: hostsetup.error_weights
        inside = em['real'].get('l2_projection_diff', {}).get('inside_initial_support', False)
        gen = opt.get('general', {})
        units = (gen.get('L2_cache', 512) / 2) * 1024 / 16
        use_mask = bool(inside) and not (gen.get('cache_aware', True) and not (np.prod(self.shape) > units))
        W = polar_integrator_weights(e.rs, e.phis)
        if use_mask:
            W = W * self.initial_support
        else:
            W[self.N - 2, :] = 0.0                                    # `square[~True] = 0`: shell N - 2 (fxs_IO_methods.py:113-116)
        e.set_error_weights(W)
        self.default_sigma = np.pi / np.max(e.qs)                     # fxs_Projections.py:189-190
        self.initial_densities = initial_densities
        self.seeds = seeds

    # -- density guess (reconstruct.py:1115-1174 with the PolarIntegrator, 1126-1127)
    def _initial_density(self, i):
        if self.initial_densities is not None:
            return np.asarray(self.initial_densities[i], dtype=complex)
        dg = self.opt['density_guess']
        if dg['type'] == 'low_resolution_autocorrelation':
            raise NotImplementedError("2-D density guess 'low_resolution_autocorrelation': the reference raises for dimensions == 2 "
                                      '(reconstruct.py:1186: low_resolution_intensity_coefficients is False)')
        rng = np.random.default_rng(None if self.seeds is None else self.seeds[i])
        e = self.engine
        radius = dg['radius']
        if isinstance(radius, bool):
            radius = self.opt['particle_radius']
        if radius < 0:
            radius = np.max(e.rs)
        r = np.broadcast_to(e.rs[:, None], self.shape)
        if dg['type'] == 'ball':
            density = np.zeros(self.shape)
            inside = r < radius
            density[inside] = 1 + 1 / dg['random']['SNR'] * rng.random(int(inside.sum()))
        elif dg['type'] == 'bump':
            amp = 1 + 1 / dg['random']['SNR'] * rng.random(self.shape)
            inside = (r > -radius) & (r < radius)
            env = np.zeros(self.shape)
            env[inside] = np.exp(-dg['bump']['slope'] * radius ** 2 / (radius ** 2 - r[inside] ** 2))
            density = amp * env
        else:
            raise NotImplementedError('2-D density guess %r' % (dg['type'],))
        total_sq = np.sum(polar_integrator_weights(e.rs, e.phis) * density * density)
        return (density * np.sqrt(self.rsetup.integrated_intensity / total_sq)).astype(complex)

    def _initial_support(self, sup):
        """RealProjection's initial support (fxs_Projections.py:60-93): a radius, or the thresholded auto-correlation
        ift(icht(pr.T)).real of the full projection vectors (reconstruct.py:400-403, 421-423; icht = the inverse REAL harmonic
        transform, ift the polar inverse Fourier transform -- both on the device)"""
        e = self.engine
        if sup['type'] == 'max_radius':
            return np.broadcast_to(e.rs[:, None] < sup['max_radius'], self.shape)
        if sup['type'] == 'auto_correlation':
            pr = np.array(self.rsetup.full_projection_matrices).T                                       # (Nq, M + 1)
            auto = e.fourier_transform(e.real_harmonic_inverse(pr)[0].astype(complex), True)[0].real
            m = np.array(auto >= sup['auto_correlation']['threshold'] * np.max(auto))                  # fxs_Projections.py:77-84
            m[np.broadcast_to(e.rs[:, None], self.shape) > self.opt['particle_radius']] = False
            return m
        raise NotImplementedError('2-D initial support %r' % (sup['type'],))

    def _reciprocal_errors(self, F, F_new, Im):
        """the reciprocal metrics of one step per restart: {name: (B,) or (B, n_used)} (fxs_IO_methods.py:301-310, 370-400)"""
        out = {}
        for name in self.reciprocal_metrics:
            if name == 'deg2_invariant_l2_diff':
                ref = self._deg2_ref.copy()
                zero_id = self.rsetup.used_orders[0]
                ref[zero_id] = self._deg2_ref[zero_id] / self.rsetup.number_of_particles
                order_array = np.array(tuple(self.rsetup.used_orders.values()))
                vals = np.full((self.B, len(self._deg2_norm)), -1.0)
                nz = self._deg2_norm != 0
                for b in range(self.B):
                    Bm = np.einsum('qm,pm->mqp', Im[b], Im[b].conj())[order_array]
                    diff = ref - Bm
                    nd = np.sum((diff * diff.conj()).real, axis=(1, 2))
                    vals[b, nz] = nd[nz] / self._deg2_norm[nz]
                out[name] = vals
            else:
                # l2_projection_diff of (F, F'): the cache-aware branch asks for type 'reziprocal' and gets the REAL grid's integrator
                # (fxs_IO_methods.py:131-140), `square[~True] = 0` drops shell N - 2
                W = polar_integrator_weights(self.engine.rs, self.engine.phis)
                W[self.N - 2, :] = 0.0
                num = np.sum(W * np.abs(F - F_new) ** 2, axis=(1, 2))
                den = np.sum(W * np.abs(F) ** 2, axis=(1, 2))
                out[name] = np.where(den != 0, num / np.where(den != 0, den, 1), np.inf)
        return out

    def _main_error(self, err_real, recip):
        """generate_main_error_routine (fxs_IO_methods.py:746-765): mean / min / max / prod of np.array(last values of the chosen
        metrics); a scalar metric next to a per-order one makes that array ragged upstream (ValueError) and here"""
        method = {'mean': np.mean, 'min': np.min, 'max': np.max, 'prod': np.prod}[self.main_type]
        out = np.empty(self.B)
        for b in range(self.B):
            vals = [err_real[b] for _ in self.main_real] + [recip[n][b] for n in self.main_reciprocal]
            if len({np.shape(v) for v in vals}) > 1:
                raise ValueError('main error over metrics of different shapes (inhomogeneous array upstream, fxs_IO_methods.py:758)')
            out[b] = method(np.array(vals))
        return out

    def _shift_to_center(self, F, rho):
        """assemble_output_modifier's shift_center for dimensions == 2 (reconstruct.py:454, 721-735; misk.py:295-312;
        fxs_Projections.py:1419-1432): (F, rho) -> (F phases, IFT(FT(rho) phases)), phases = exp(+i k.c), c = centre of mass of
        Re(rho) by the PolarIntegrator; transforms on the device, the moments are weighted sums on the host"""
        e = self.engine
        W = polar_integrator_weights(e.rs, e.phis)
        r, ph = np.meshgrid(e.rs, e.phis, indexing='ij')
        q, pq = np.meshgrid(e.qs, e.phis, indexing='ij')
        ft = e.fourier_transform(rho)
        phases = np.empty(rho.shape, complex)
        centers = []
        for b in range(self.B):
            re = rho[b].real
            integral = np.sum(W * re)
            if integral == 0:
                integral = 1
            cx, cy = np.sum(W * r * np.cos(ph) * re) / integral, np.sum(W * r * np.sin(ph) * re) / integral
            rad, phi_c = np.hypot(cx, cy), np.arctan2(cy, cx)
            phi_c = phi_c + 2 * np.pi if phi_c < 0 else phi_c
            centers.append(np.array([rad, phi_c]))
            phases[b] = np.exp(1j * (q * np.cos(pq) * rad * np.cos(phi_c) + q * np.sin(pq) * rad * np.sin(phi_c)))
        return F * phases, e.fourier_transform(ft * phases, True), centers

    def _sw_ramps(self):
        sw_opt = self.opt['projections']['real']['shrink_wrap']
        order = self.opt['main_loop']['sub_loops']['order']
        sig, thr = [], []
        for lid in range(len(order)):
            s = sw_opt['sigmas'][lid] if len(sw_opt['sigmas']) - 1 >= lid else False
            sig.append(hs.LinearRamp(*(s if isinstance(s, (list, tuple)) else [s]), default_start=self.default_sigma, default_stop=self.default_sigma))
            t = sw_opt['thresholds'][lid] if len(sw_opt['thresholds']) - 1 >= lid else 0.1
            thr.append(hs.LinearRamp(*(t if isinstance(t, (list, tuple)) else [t])))
        return sig, thr

    def _update_shrink_wrap(self, iteration, loop_number):
        r = self._sig_ramps[loop_number]
        if not r.undefined:
            v = r(iteration)
            ok = np.issubdtype(np.array(v).dtype, np.number) and not isinstance(v, bool) and v > 0
            self.sw_sigma = v if ok else self.default_sigma             # fxs_Projections.py:233-243
        t = self._thr_ramps[loop_number]
        if not t.undefined:
            v = t(iteration)
            self.sw_threshold = 0 if v < 0 else (1 if v >= 1 else v)    # 218-227

    @staticmethod
    def _change_to_ft_stab(popt, name, eis_list):
        """reconstruct.py:836-850: one decision per restart; a bool when they agree"""
        if name[-8:] == '_ft_stab' or 'ft_stab' not in popt:
            return False
        v = popt['ft_stab']
        if isinstance(v, bool):
            return v
        if v == 'link_to_enforce_initial_support':
            delay = max(int(popt['link_to_enforce_initial_support']['delay']), 1)
            if len(eis_list) >= delay:
                flags = ~(np.array(eis_list[-delay:]) == True).any(axis=0)          # noqa: E712
                if flags.all() != flags.any():
                    return flags                                    # restarts of one batch disagree: per restart (see _step)
                return bool(flags.all())
        return False

    def _step(self, key, ft_stab, beta, rho, support, fixed, want):
        """Engine2D.step for the batch; when the restarts disagree on ft_stab (the reference decides per reconstruction process) the
        step runs once with and once without the add-back and every restart takes its own"""
        e = self.engine
        if not isinstance(ft_stab, np.ndarray):
            return e.step(key, ft_stab, beta, rho, support, fixed_intensity=fixed, want_inputs=want)
        on = e.step(key, True, beta, rho, support, fixed_intensity=fixed, want_inputs=want)
        off = e.step(key, False, beta, rho, support, fixed_intensity=fixed, want_inputs=want)
        out = []
        for a, b in zip(on, off):
            if a is None:
                out.append(None)
            else:
                sel = ft_stab.reshape((-1,) + (1,) * (a.ndim - 1))
                out.append(np.where(sel, a, b))
        return tuple(out)

    def phasing_loop(self):
        """create_initial_state + the sub-loops + generate_output (reconstruct.py:957-1035) for the batch: list of result dicts"""
        e, B, opt = self.engine, self.B, self.opt
        rho0 = np.stack([self._initial_density(b) for b in range(B)])
        F0 = e.fourier_transform(rho0)
        rho0 = e.fourier_transform(F0, True)                          # 962-963: the state starts from IFT(FT(guess))
        hl = opt['main_loop'].get('history_length', 3)
        hist = [(F0.copy(), rho0.copy())] * hl                        # per entry: (F (B, ...), rho (B, ...))
        init_sup = np.broadcast_to(self.initial_support, (B,) + self.shape).copy()
        support = init_sup.copy()                                     # effective support (what RealProjection's mask holds)
        best = {'pair': (F0.copy(), rho0.copy()), 'err': np.full(B, np.inf), 'iter': np.zeros(B, int), 'mask': init_sup.copy()}
        err_real, err_main, unknowns = [], [], None
        err_recip = {n: [] for n in self.reciprocal_metrics}
        self._sig_ramps, self._thr_ramps = self._sw_ramps()
        self.sw_sigma, self.sw_threshold = self.default_sigma, 0.06
        hio_opt = opt['projections']['real']['HIO']
        eis_opt = opt['projections']['real']['projections']['support']['enforce_initial_support']
        limit = eis_opt['if_error_bigger_than'] if eis_opt['apply'] else np.inf
        loops = opt['main_loop']['sub_loops']
        eis_list, iterations = [], []
        want = bool(self.reciprocal_metrics)
        for loop_number, loop_name in enumerate(loops['order']):
            lo = loops[loop_name]
            methods = {}
            for key in lo['order']:
                mo = lo['methods'][key]
                methods[key] = ({'iterations': mo.get('iterations', 0), 'options': mo} if isinstance(mo, dict) else {'iterations': mo, 'options': {}})
                if key not in ('HIO', 'ER', 'SW', 'SW_center', 'HIO_non_FXS', 'ER_non_FXS'):
                    raise NotImplementedError('2-D loop method %r' % (key,))
                if key.endswith('_non_FXS') and self.reciprocal_metrics:
                    # (upstream the *_non_FXS start sketch hands the reciprocal metrics a grid instead of I_m and raises)
                    raise NotImplementedError('2-D %s with reciprocal metrics enabled: the reference raises' % key)
            beta_cfg = hio_opt['beta'][loop_number] if len(hio_opt['beta']) - 1 >= loop_number else [0.5, 0.5, -1 / 700, 1600]
            ramp = hs.ExponentialRamp(*beta_cfg)
            if 'SW' in methods:
                self._update_shrink_wrap(0, loop_number)
            step = sw_step = iteration = 0
            # reconstruct.py:859: `hist` is a local that is re-read from the state only at the top of every phasing step (913); SW_center
            # (893) and the *_non_FXS intensity (901) read it as it was left there -- the history BEFORE the most recent step
            stale = hist
            latest_intensity = None
            for iteration in range(1, lo['iterations'] + 1):
                for key in lo['order']:
                    if key == 'SW':                                   # 877-885
                        new_sup = e.shrinkwrap(hist[-1][1], self.sw_sigma, self.sw_threshold)
                        last = np.asarray(err_main[-1]) if err_main else np.full(B, np.nan)
                        enforce = last > limit if err_main else np.zeros(B, bool)      # (`error_dict['main'][-1:] > limit` of an empty list is empty: falsy)
                        eis_list.append(enforce)
                        support = np.where(enforce[:, None, None], new_sup & init_sup, new_sup)    # support setter, fxs_Projections.py:53-58
                        sw_step += 1
                        self._update_shrink_wrap(sw_step, loop_number)
                        continue
                    if key == 'SW_center':
                        # reconstruct.py:606-613, 886-897: one enforce decision, then `iterations` support updates; the sketch shifts
                        # nothing and hands its outputs back in swapped order -- the pair appended to the (stale) history is (rho, FT(rho))
                        if not err_main:
                            raise IndexError("SW_center before any phasing step: error_dict['main'][-1] of an empty list (reconstruct.py:887)")
                        enforce = np.asarray(err_main[-1]) > limit
                        eis_list.append(enforce)
                        for _ in range(methods[key]['iterations']):
                            rho = hist[-1][1]
                            new_sup = e.shrinkwrap(rho, self.sw_sigma, self.sw_threshold)
                            support = np.where(enforce[:, None, None], new_sup & init_sup, new_sup)
                            hist = stale[1:] + [(rho.copy(), e.fourier_transform(rho))]
                            sw_step += 1
                            self._update_shrink_wrap(sw_step, loop_number)
                        continue
                    if key.endswith('_non_FXS'):
                        if latest_intensity is None:
                            latest_intensity = np.abs(stale[-1][0])       # |F| of the stale pair, used as the intensity (899-902)
                    else:
                        latest_intensity = None
                    ft_stab = self._change_to_ft_stab(methods[key]['options'], key, eis_list)
                    for _ in range(methods[key]['iterations']):
                        stale = hist
                        res = self._step(key, ft_stab, ramp.eval(step), hist[-1][1], support, latest_intensity, want)
                        F_new, rho_new, err, unk = res[:4]
                        if unk is not None:
                            unknowns = unk
                        hist = hist[1:] + [(F_new, rho_new)]
                        err_real.append(err)
                        recip = self._reciprocal_errors(res[4], F_new, res[5]) if want else {}
                        for n_, v_ in recip.items():
                            err_recip[n_].append(v_)
                        main_err = self._main_error(err, recip)
                        err_main.append(main_err)
                        better = best['err'] > main_err
                        if better.any():
                            sel = better[:, None, None]
                            best['pair'] = (np.where(sel, F_new, best['pair'][0]), np.where(sel, rho_new, best['pair'][1]))
                            best['mask'] = np.where(sel, support, best['mask'])
                            best['err'] = np.where(better, main_err, best['err'])
                            best['iter'] = np.where(better, iteration, best['iter'])
                        step += 1
            n_first = lo.get('best_density_not_in_first_n_iterations', np.inf)
            res = best['iter'] > n_first                              # 945-949
            if np.any(res):
                sel = res[:, None, None]
                hist = hist[1:] + [(np.where(sel, best['pair'][0], hist[-1][0]), np.where(sel, best['pair'][1], hist[-1][1]))]
                support = np.where(sel, best['mask'], support)
            iterations.append(iteration)
        best_pair, last_pair = best['pair'], hist[-1]
        self.neg_center_pos = None
        om = opt.get('output_density_modifiers', {})
        # assemble_output_modifier (reconstruct.py:721-755): shift_center, and with SO_freedom in use + fix_orientation the sketch
        # shift_center + fix_orientation whatever shift_to_center says (746-752; 2-D only)
        fix = bool(om.get('fix_orientation', False)) and self.rsetup.use_SO_freedom
        if om.get('shift_to_center', False) or fix:
            bF, brho, _ = self._shift_to_center(*best_pair)
            lF, lrho, self.neg_center_pos = self._shift_to_center(*last_pair)
            best_pair, last_pair = (bF, brho), (lF, lrho)
        if fix:
            rot = RemainingRotation2D(self.rsetup.projection_matrices, self.rsetup.used_orders, e.qs, e.n_phi, self.rsetup.radial_high_pass)

            def turn(grids):                                          # complex harmonic transforms on the device, the phases on the host
                c = e.harmonic(grids)
                return e.harmonic(np.stack([rot(c[b], unknowns[b]) for b in range(B)]), True)
            best_pair = (turn(best_pair[0]), turn(best_pair[1]))
            last_pair = (turn(last_pair[0]), turn(last_pair[1]))
        err_real, err_main = np.array(err_real), np.array(err_main)
        # calc_deg2_invariant of the last density (reconstruct.py:757-765, 993): B_m = I_m (x) I_m^* (fxs_invariant_tools.py:906-914)
        F_last = e.fourier_transform(last_pair[1])
        I_last = e.real_harmonic_forward((F_last * F_last.conj()).real)
        masked = np.array(self.rsetup.projection_matrices)           # 997-1001 (one vector per used order in 2-D)
        masked[~self.rsetup.radial_mask[list(self.rsetup.used_orders.values())]] = 0
        grids = {'real_grid': np.stack(np.meshgrid(e.rs, e.phis, indexing='ij'), -1),
                 'reciprocal_grid': np.stack(np.meshgrid(e.qs, e.phis, indexing='ij'), -1)}
        n_steps = len(err_main)
        out = []
        for b in range(B):
            out.append({'real_density': best_pair[1][b], 'last_real_density': last_pair[1][b], 'reciprocal_density': best_pair[0][b],
                        'last_reciprocal_density': last_pair[0][b], 'final_error': float(best['err'][b]), 'initial_density': rho0[b],
                        'initial_support': self.initial_support.copy(),
                        'error_dict': {'main': err_main[:, b].copy(), 'real': {'l2_projection_diff': err_real[:, b].copy()},
                                       'reciprocal': {n_: np.array(v_)[:, b].copy() for n_, v_ in err_recip.items()}},
                        'support_mask': best['mask'][b], 'last_support_mask': support[b], 'loop_iterations': int(np.sum(iterations) + 1),
                        'fxs_unknowns': None if unknowns is None else unknowns[b],
                        'n_particles': np.full((n_steps, 1), self.rsetup.number_of_particles), 'n_particles_gradients': np.array([]),
                        'n_particles_fraction': np.array([]), 'grid_pair': grids, 'projection_matrices': masked,
                        'last_deg2_invariant': np.einsum('qm,pm->mqp', I_last[b], I_last[b].conj())})
        return out

    def close(self):
        self.engine.close()
