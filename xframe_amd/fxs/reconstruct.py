"""``fxs reconstruct`` worker on the MI355X engine -- host-side mirror of
``xframe/projects/fxs/reconstruct.py`` (``ProjectWorker`` 89-209, ``MTIP`` 211-1278).

Same call shape as the reference::

    worker = ProjectWorker(settings, invariants)        # reference: settings.project / db.load('invariants')
    result, _ = worker.run()                            # object array of per-restart result dicts (1003-1021)

    MTIP.preinit(settings, invariants); m = MTIP(process_factory); m.generate_phasing_loop(); m.phasing_loop()

What differs by design: the reference forks one OS process per restart and round-trips every Hankel
transform through a GPU daemon (reconstruct.py:141-157, Multiprocessing.py:1033-1117); here all restarts of a
rank are one batch resident on one GPU and the whole loop runs on the device; the Python side only walks the
schedule (sub-loops, SW steps, ramps) and collects results.  Restarts are sharded over ranks / GPUs by
``parallel.shard_restarts``; there is no collective on the data path.
"""
import time

import numpy as np

from . import hostsetup as hs
from .engine import Engine
from .operators import RecipeFactory, build_operators
from .settings import DictNamespace, resolve


class MTIP:
    # set by MTIP.preinit (reference: class attributes filled from settings.project / the database, 241-276)
    settings = None
    mtip_data = None
    preinit_was_called = False
    loops_str = 'Loops:\n'

    @classmethod
    def preinit(cls, settings=None, mtip_data=None):
        cls.settings = resolve(settings)
        cls.mtip_data = mtip_data
        cls.dimensions = cls.settings['dimensions']
        if cls.dimensions not in (2, 3):
            raise NotImplementedError('dimensions must be 2 or 3')
        q = np.asarray(mtip_data['data_radial_points'])
        cls.data_q_limits = [q.min(), q.max()]
        cls.data_number_of_radial_points = len(q)
        loops = cls.settings['main_loop']['sub_loops']
        s = 'Loops:\n'
        width = max(len(n) for n in loops['order'])
        for name in loops['order']:
            lo = loops.get(name, {})
            ms = ''
            for method in lo.get('order', []):
                mo = lo['methods'].get(method, {})
                ms += f"{mo['iterations'] if isinstance(mo, dict) else mo}x{method} "
            s += f"\t{name}:" + ' ' * (width - len(name)) + f"\t {lo.get('iterations', '')}x( {ms})\n"
        cls.loops_str = s
        cls.preinit_was_called = True

    def __init__(self, process_factory=None, n_restarts=1, device=0, fused=True, seeds=None, initial_densities=None,
                 lib_path=None):
        if not MTIP.preinit_was_called:
            raise RuntimeError('MTIP.preinit(settings, invariants) must be called first')
        self.process_factory = process_factory if process_factory is not None else RecipeFactory({})
        self.opt = MTIP.settings
        self.n_restarts = int(n_restarts)
        self.device = device
        self.fused = fused
        self.seeds = seeds
        self.initial_densities = initial_densities
        self.lib_path = lib_path
        self.results = {}
        self.engine = None
        self.phasing_loop = False
        self.timing = {}

    # ------------------------------------------------------------------ assembly (reference 1269-1278)
    def generate_phasing_loop(self):
        t_engine = time.perf_counter()
        if MTIP.dimensions == 2:
            # the polar loop: host-orchestrated on the batched device operators (reconstruct2d.py)
            from .reconstruct2d import MTIP2D
            self.loop2d = MTIP2D(self.opt, MTIP.mtip_data, n_restarts=self.n_restarts, initial_densities=self.initial_densities,
                                 seeds=self.seeds, device=self.device, lib_path=self.lib_path)
            self.engine = self.loop2d.engine
            self._engine_seconds = time.perf_counter() - t_engine
            self.rprojection = self.loop2d.rsetup

            def loop2d(*args, **kwargs):
                t0 = time.perf_counter()
                res = self.loop2d.phasing_loop()
                self.timing = {'engine_seconds': self._engine_seconds, 'loop_seconds': time.perf_counter() - t0}
                out = np.empty(len(res), dtype=object)
                out[:] = res
                return out
            self.phasing_loop = loop2d
            return
        self.engine = Engine(self.opt, MTIP.mtip_data, n_batch=self.n_restarts, device=self.device, fused=self.fused,
                             lib_path=self.lib_path)
        self._engine_seconds = time.perf_counter() - t_engine
        self.rprojection = self.engine.rsetup
        self.process_factory.addOperators(build_operators(self.engine))
        self.phasing_loop = self._main_loop

    # ------------------------------------------------------------------ ramps (reference 1212-1258)
    def _sw_ramps(self):
        e = self.engine
        sw_opt = self.opt['projections']['real']['shrink_wrap']
        if sw_opt.get('mode', 'threshold') != 'threshold':
            raise NotImplementedError("shrink_wrap.mode = %r: only 'threshold' is on the accelerated path "
                                      '(fixed_volume: fxs_Projections.py:260-291)' % (sw_opt.get('mode'),))
        order = self.opt['main_loop']['sub_loops']['order']
        sig, thr = [], []
        for lid in range(len(order)):
            s = sw_opt['sigmas'][lid] if len(sw_opt['sigmas']) - 1 >= lid else False
            s = s if isinstance(s, (list, tuple)) else [s]
            sig.append(hs.LinearRamp(*s, default_start=e.default_sigma, default_stop=e.default_sigma))
            t = sw_opt['thresholds'][lid] if len(sw_opt['thresholds']) - 1 >= lid else 0.1
            t = t if isinstance(t, (list, tuple)) else [t]
            thr.append(hs.LinearRamp(*t))
        return sig, thr

    def _update_shrink_wrap(self, iteration, loop_number):
        r = self._sig_ramps[loop_number]
        if not r.undefined:
            v = r(iteration)
            ok = np.issubdtype(np.array(v).dtype, np.number) and not isinstance(v, bool) and v > 0
            self.sw_sigma = v if ok else self.engine.default_sigma           # fxs_Projections.py:233-243
        t = self._thr_ramps[loop_number]
        if not t.undefined:
            v = t(iteration)
            self.sw_threshold = 0 if v < 0 else (1 if v >= 1 else v)          # 218-227

    # ------------------------------------------------------------------ initial densities (1115-1174, 957-979)
    def _initial_densities(self):
        """guesses of all restarts of this batch; the seed-independent autocorrelation is transformed once"""
        self._autocorrelation = None
        return [self._initial_density(b) for b in range(self.n_restarts)]

    def _initial_density(self, i):
        if self.initial_densities is not None:
            return np.asarray(self.initial_densities[i], dtype=complex)
        dg = self.opt['density_guess']
        if dg['type'] not in ('bump', 'ball', 'low_resolution_autocorrelation'):
            raise AssertionError('density type "{}" is not known.'.format(dg['type']))     # reconstruct.py:1206-1209
        seed = None if self.seeds is None else self.seeds[i]
        rng = np.random.default_rng(seed)       # seed None = OS entropy, like the reference's os.urandom seeding
        e = self.engine
        radius = dg['radius']
        if isinstance(radius, bool):
            radius = self.opt['particle_radius']
        if radius < 0:
            radius = np.max(e.rs)
        if dg['type'] == 'low_resolution_autocorrelation':
            # reconstruct.py:1175-1205: transforms on the device, the rest on the host as in the reference
            if getattr(self, '_autocorrelation', None) is None:
                coeff = np.zeros((e.N, e.nlm), complex)
                for l, pm in e.rsetup.projection_matrices.items():
                    coeff[:, l * l:l * l + pm.shape[1]] = pm
                self._autocorrelation = e.fourier_transform(e.sht_inverse(coeff)[0], True)[0].real
            ac = self._autocorrelation
            return hs.autocorrelation_density(ac, e.rs, e.shape, self.opt['particle_radius'], dg['random']['SNR'], rng,
                                              e.rsetup.integrated_intensity, e.int_wr, e.int_wt)
        if dg['type'] == 'ball':
            return hs.ball_density(e.rs, e.shape, radius, dg['random']['SNR'], rng, e.rsetup.integrated_intensity,
                                   e.int_wr, e.int_wt)
        return hs.bump_density(e.rs, e.shape, radius, dg['bump']['slope'], dg['random']['SNR'], rng,
                               e.rsetup.integrated_intensity, e.int_wr, e.int_wt)

    @staticmethod
    def _change_to_ft_stab(popt, name, eis_list):
        """reconstruct.py:836-850: the reference decides per reconstruction process -- a bool when the restarts agree, else one per restart."""
        if name[-8:] == '_ft_stab' or 'ft_stab' not in popt:
            return False
        v = popt['ft_stab']
        if isinstance(v, bool):
            return v
        if v == 'link_to_enforce_initial_support':
            delay = max(int(popt['link_to_enforce_initial_support']['delay']), 1)
            if len(eis_list) >= delay:
                recent = np.array(eis_list[-delay:])            # (delay, B)
                flags = ~(recent == True).any(axis=0)          # noqa: E712
                if flags.all() != flags.any():
                    return flags                                # the restarts disagree: a flag per restart (Engine.run takes it)
                return bool(flags.all())
        return False

    # ------------------------------------------------------------------ the loop (854-951, 1023-1035)
    def _main_loop(self, *args, **kwargs):
        e = self.engine
        B = self.n_restarts
        opt = self.opt
        t_setup = time.perf_counter()
        # all guesses first: the autocorrelation guess runs transforms on the engine, whose single-operator entry
        # points use the same device scratch that mtip_set_density stages the guesses in
        guesses = self._initial_densities()
        for b in range(B):
            e.set_density(b, guesses[b])
        e.init_state()
        initial_density = [e.density(b) for b in range(B)]
        initial_mask = e.initial_support.copy()
        self._sig_ramps, self._thr_ramps = self._sw_ramps()
        self.sw_sigma, self.sw_threshold = e.default_sigma, 0.06
        hio_opt = opt['projections']['real']['HIO']
        eis_opt = opt['projections']['real']['projections']['support']['enforce_initial_support']
        limit = eis_opt['if_error_bigger_than'] if eis_opt['apply'] else np.inf
        loops = opt['main_loop']['sub_loops']
        eis_list = []
        iterations = []
        e.synchronize()
        t0 = time.perf_counter()
        n_steps = 0
        # best_error / best_iteration per restart (reconstruct.py:934-938), only followed on the host when some loop
        # reselects the best density at its end (945-949)
        track_best = any(np.isfinite(loops[name].get('best_density_not_in_first_n_iterations', np.inf))
                         for name in loops['order'])
        best_err_h = np.full(B, np.inf)
        best_iter_h = np.zeros(B, int)
        for loop_number, loop_name in enumerate(loops['order']):
            lo = loops[loop_name]
            e.begin_sub_loop()
            loop_first_step = n_steps
            step_iteration = []
            methods = {}
            for key in lo['order']:
                mo = lo['methods'][key]
                methods[key] = ({'iterations': mo.get('iterations', 0), 'options': mo} if isinstance(mo, dict)
                                else {'iterations': mo, 'options': {}})
            beta_cfg = hio_opt['beta'][loop_number] if len(hio_opt['beta']) - 1 >= loop_number else [0.5, 0.5, -1 / 700, 1600]
            ramp = hs.ExponentialRamp(*beta_cfg)
            if 'SW' in methods:
                self._update_shrink_wrap(0, loop_number)
            step = 0
            sw_step = 0
            iteration = 0
            for iteration in range(1, lo['iterations'] + 1):
                for key in lo['order']:
                    if key == 'SW':
                        enforced = e.shrinkwrap(self.sw_sigma, self.sw_threshold, limit)
                        eis_list.append(enforced)
                        sw_step += 1
                        self._update_shrink_wrap(sw_step, loop_number)
                        continue
                    if key == 'SW_center':
                        # reconstruct.py:886-897: one enforce decision, then `iterations` support updates, each replacing
                        # the last pair (the sketch, 606-613, shifts nothing and hands its outputs back in swapped order)
                        for i in range(methods[key]['iterations']):
                            enforced = e.shrinkwrap(self.sw_sigma, self.sw_threshold, limit)
                            if i == 0:
                                eis_list.append(enforced)
                            e.refresh_reciprocal_density()
                            sw_step += 1
                            self._update_shrink_wrap(sw_step, loop_number)
                        continue
                    repeats = methods[key]['iterations']
                    ft_stab = self._change_to_ft_stab(methods[key]['options'], key, eis_list)
                    betas = np.array([ramp.eval(step + i) for i in range(repeats)], dtype=float)
                    e.run(key, ft_stab, betas, fetch=False)
                    step += repeats
                    n_steps += repeats
                    step_iteration += [iteration] * repeats
            if track_best and n_steps > loop_first_step:
                errs = e.fetch_main_errors(loop_first_step, n_steps - loop_first_step)
                for i, it in enumerate(step_iteration):
                    better = best_err_h > errs[i]
                    best_err_h = np.where(better, errs[i], best_err_h)
                    best_iter_h = np.where(better, it, best_iter_h)
            n_first = lo.get('best_density_not_in_first_n_iterations', np.inf)
            if np.isfinite(n_first):
                reselect = best_iter_h > n_first
                if reselect.any():
                    e.select_best(reselect)
            iterations.append(iteration)
        e.synchronize()
        t1 = time.perf_counter()
        out = self._generate_output(iterations, initial_density, initial_mask, n_steps)
        # where a run spends its time: engine (host setup of weights / projection matrices + uploads), initial densities and
        # state, the loop itself (device resident), the result dicts (grids back over PCIe, centring, output transforms)
        self.timing = {'engine_seconds': getattr(self, '_engine_seconds', float('nan')), 'setup_seconds': t0 - t_setup,
                       'loop_seconds': t1 - t0, 'output_seconds': time.perf_counter() - t1, 'steps_per_restart': n_steps,
                       'iterations_per_second': n_steps * B / (t1 - t0) if t1 > t0 else float('nan')}
        return out

    # ------------------------------------------------------------------ output dict (980-1022)
    def _generate_output(self, iterations, initial_density, initial_mask, n_steps):
        e = self.engine
        real_err, deg2 = e.fetch_errors(0, n_steps)
        rec_l2 = e.fetch_reciprocal_l2(0, n_steps)
        inv_metrics = e.fetch_invariant_metrics(0, n_steps)         # II_error / ccd_diff / fqc_error, when enabled
        main_err = e.fetch_main_errors(0, n_steps)
        best_err, _ = e.best_error()
        masked_pm = []
        for l in range(e.L + 1):
            m = np.array(e.rsetup.projection_matrices.get(l, np.zeros((e.N, min(e.N, 2 * l + 1)), complex)))
            m[~e.rsetup.radial_mask[l]] = 0
            masked_pm.append(m)
        r, t, p = np.meshgrid(e.rs, e.theta, e.phi, indexing='ij')
        real_grid = np.stack((r, t, p), -1)
        q, t, p = np.meshgrid(e.qs, e.theta, e.phi, indexing='ij')
        reciprocal_grid = np.stack((q, t, p), -1)
        out = np.empty(self.n_restarts, dtype=object)
        # the reference's metric has one entry per used order (fxs_IO_methods.py:413, 425-427)
        order_array = np.array(tuple(e.rsetup.used_orders.values()), dtype=int)
        shift = bool(self.opt.get('output_density_modifiers', {}).get('shift_to_center', False))
        # grids of the result dicts that are the engine's state as it stands (no output modifier touched them): the end-of-run
        # gather sends those from HBM (device_source)
        self._state_keys = {} if shift else {'real_density': ('density', True), 'last_real_density': ('density', False),
                                             'reciprocal_density': ('reciprocal_density', True),
                                             'last_reciprocal_density': ('reciprocal_density', False)}
        self._state_keys.update({'support_mask': ('support', True), 'last_support_mask': ('support', False)})
        B = self.n_restarts
        recip = {best: np.stack([e.reciprocal_density(b, best=best) for b in range(B)]) for best in (True, False)}
        real = {best: np.stack([e.density(b, best=best) for b in range(B)]) for best in (True, False)}
        last_deg2 = [e.last_deg2_invariant(b) for b in range(B)]
        if shift:
            # assemble_output_modifier 'shift_center' (reconstruct.py:728-734), all restarts at once: transforms on the
            # device, centre of mass and phase ramp on the host as the reference's operators do
            for best in (True, False):
                centers = [hs.calc_center(e.rs, e.theta, e.phi, real[best][b]) for b in range(B)]
                self.results['neg_center_pos'] = centers[-1]
                phases = np.stack([hs.shift_phases(e.qs, e.theta, e.phi, c, opposite_direction=True) for c in centers])
                recip[best] = recip[best] * phases
                real[best] = e.fourier_transform(e.fourier_transform(real[best]) * phases, True)
            # last_deg2_invariant of the modified last density (reconstruct.py:993)
            last_deg2 = e.deg2_invariants(e.sht_forward(e.fourier_transform(real[False]), 1))
        for b in range(self.n_restarts):
            err = {'main': main_err[:, b].copy(),
                   'real': {'l2_projection_diff': real_err[:, b].copy()},
                   'reciprocal': {}}
            for name in self.opt['main_loop']['error']['methods']['reciprocal']['calculate']:      # in the order they are listed, as upstream
                if name == 'deg2_invariant_l2_diff':
                    err['reciprocal'][name] = deg2[:, b][:, order_array].copy()
                elif name == 'deg2_ranked_invariant_l2_diff':
                    err['reciprocal'][name] = deg2[:, b][:, order_array][:, e.deg2_ranked_id].copy()
                elif name == 'l2_projection_diff':
                    err['reciprocal'][name] = rec_l2[:, b].copy()
            for name, hist in inv_metrics.items():
                err['reciprocal'][name] = hist[:, b].copy()
            out[b] = {
                'real_density': real[True][b], 'last_real_density': real[False][b],
                'reciprocal_density': recip[True][b], 'last_reciprocal_density': recip[False][b],
                'final_error': float(best_err[b]), 'initial_density': initial_density[b],
                'initial_support': initial_mask, 'error_dict': err,
                'support_mask': e.support(b, best=True), 'last_support_mask': e.support(b),
                'loop_iterations': int(np.sum(iterations) + 1), 'fxs_unknowns': e.unknowns(b),
                'n_particles': np.full((n_steps, 1), e.rsetup.number_of_particles),
                'n_particles_gradients': np.array([]), 'n_particles_fraction': np.array([]),
                'grid_pair': {'real_grid': real_grid, 'reciprocal_grid': reciprocal_grid},
                'projection_matrices': masked_pm, 'last_deg2_invariant': last_deg2[b]}
        return out


def _device_source(mtip, batch):
    """key -> device tensor of restart `batch` of that MTIP instance's engine, or None when the dict's array is not engine state"""
    def get(key):
        spec = getattr(mtip, '_state_keys', {}).get(key)
        if spec is None or mtip.engine is None:
            return None
        t = mtip.engine.t_state(spec[0], batch, best=spec[1])
        return t.bool() if spec[0] == 'support' else t
    return get


class ProjectWorker:
    """reconstruct.py:89-209.  ``run()`` returns ``(result, locals())`` like ProjectWorkerInterface.run
    (xframe/interfaces.py:9-20)."""

    def __init__(self, settings=None, invariants=None, device=None, rank=None, world_size=None, seeds=None, lib_path=None,
                 n_gather_full=8):
        """rank / world_size / device default to the launcher's RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run: one
        process per GPU), else to a single process on device 0.  n_gather_full: restarts (best final errors first) whose
        grid-sized arrays are brought to rank 0 at the end of a multi-rank run."""
        import os
        self.opt = DictNamespace.dict_to_dictnamespace(resolve(settings))
        MTIP.preinit(self.opt, invariants)
        self.mtip = MTIP
        self.process_factory = RecipeFactory({})
        rank = int(os.environ.get('RANK', 0)) if rank is None else rank
        world_size = int(os.environ.get('WORLD_SIZE', 1)) if world_size is None else world_size
        device = int(os.environ.get('LOCAL_RANK', 0)) if device is None else device
        self.device, self.rank, self.world_size = device, rank, world_size
        self.n_gather_full = n_gather_full
        self.seeds = seeds
        self.lib_path = lib_path
        self.results = {'stats': {}}
        if not self.opt['GPU']['use']:
            raise RuntimeError('GPU.use = False: this worker has no CPU path; use the reference for CPU-only runs')

    def n_restarts_total(self):
        mp = self.opt['multi_process']
        n = mp.get('n_parallel_reconstructions', 1)
        if not mp.get('use', True) or isinstance(n, bool) or n is None:
            return 1
        return int(n)

    def run(self):
        from .parallel import shard_restarts, gather_results
        start = time.time()
        total = self.n_restarts_total()
        mine = shard_restarts(total, self.rank, self.world_size)
        seeds = None if self.seeds is None else [self.seeds[i] for i in mine]
        result = np.empty(0, dtype=object)
        if len(mine):
            # GPU.n_gpu_workers (reference: number of GPU daemon processes, reconstruct.py:104) = restart groups that run
            # concurrently on this GPU, each with its own engine / HIP stream, driven from its own host thread.  The
            # per-restart chain of a step is serial (the polar-factor kernel alone is half of it and fills half of the
            # CUs), so 2-3 groups overlap it with the streaming kernels of the others; with more than 3 every kernel boundary
            # gets ~40 us slower (more than four active queues of one process).
            n_eng = int(self.opt['GPU'].get('n_gpu_workers', 1) or 1)
            n_eng = max(1, min(n_eng, 3, len(mine) // 2))
            groups = [list(range(g, len(mine), n_eng)) for g in range(n_eng)]

            # the groups' engines step through mtip_run_group_async (EngineGroup): every group runs the reference's loop in its
            # own thread, their `run` calls meet and are enqueued in turn order -- each group's results are those of its own
            # mtip_run_async; GPU.take_turns = False leaves the groups to themselves
            turns = None
            if n_eng > 1 and self.opt['GPU'].get('take_turns', True):
                import threading
                from .engine import EngineGroup
                turns = EngineGroup()
                ready = threading.Barrier(n_eng)

            def run_group(local_ids):
                gseeds = None if seeds is None else [seeds[i] for i in local_ids]
                m = None
                try:
                    m = MTIP(self.process_factory, n_restarts=len(local_ids), device=self.device, seeds=gseeds,
                             lib_path=self.lib_path)
                    m.generate_phasing_loop()
                    if turns is not None and m.engine is not None and MTIP.dimensions != 2:
                        turns.attach(m.engine)
                finally:
                    if turns is not None:
                        try:
                            ready.wait(120.0)            # nobody steps before everybody is attached (or has failed)
                        except threading.BrokenBarrierError:
                            pass
                try:
                    return m, m.phasing_loop()
                finally:
                    if turns is not None and m is not None and m.engine is not None:
                        turns.leave(m.engine)

            if n_eng == 1:
                outs = [run_group(groups[0])]
            else:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(n_eng) as pool:             # ctypes calls release the GIL
                    outs = list(pool.map(run_group, groups))
            result = np.empty(len(mine), dtype=object)
            sources = {}
            for local_ids, (m, res) in zip(groups, outs):
                for j, i in enumerate(local_ids):
                    result[i] = res[j]
                    sources[mine[i]] = _device_source(m, j)
            self.mtip_instance = outs[0][0]
            self.mtip_instances = [m for m, _ in outs]
            self.results['stats']['groups'] = [dict(m.timing, restarts=len(g)) for (m, _), g in zip(outs, groups)]
            self.results['stats']['turn_calls'] = None if turns is None else dict(turns.calls)
        result = gather_results(result, mine, total, self.rank, self.world_size, n_full=self.n_gather_full,
                                device=self._torch_device(), device_source=sources if len(mine) else None)
        self.results['MTIP'] = result
        self.results['stats']['run_time'] = time.time() - start
        self.post_processing()
        return result, locals()

    def _torch_device(self):
        """device of the staging tensors of the final gather: the GPU with the nccl (= RCCL) backend, None otherwise"""
        if self.world_size == 1:
            return None
        try:
            import torch
            import torch.distributed as dist
            if dist.is_initialized() and dist.get_backend() == 'nccl':
                return torch.device('cuda', self.device)
            if dist.is_initialized() and self.lib_path is not None:     # CPU emulation of the kernels (tests): same gather path, CPU tensors
                return torch.device('cpu')
        except Exception:
            pass
        return None

    def post_processing(self):
        """reconstruct.py:160-183: sort restarts by their last main error (rank 0 holds everything)."""
        res = self.results.get('MTIP')
        if res is None or len(res) == 0:
            return
        errors = [r['error_dict']['main'][-1] for r in res]
        order = np.argsort(errors)
        self.results['sorted_ids'] = order
        self.results['reconstruction_results'] = {str(i): res[i] for i in order}

    def database_tree(self, xray_wavelength=None, reciprocity_coefficient=None):
        """the dict tree the reference's worker hands to ``database.project.save('reconstructions', ...)`` after a run
        (reconstruct.py:160-185), from this run's results: see xframe_amd/fxs/io.py (HDF5 layout, G16 fixtures)"""
        from . import io as IO
        res = self.results.get('MTIP')
        if res is None or len(res) == 0:
            raise RuntimeError('no results: run() first (rank 0 holds them)')
        dicts = {}
        for i, r in enumerate(res):
            d = dict(r)
            gp = d.get('grid_pair')
            if isinstance(gp, dict):
                d['grid_pair'] = IO.GridPair(gp['real_grid'], gp['reciprocal_grid'])
            dicts[i] = d
        if reciprocity_coefficient is None:                     # misk._get_reciprocity_coefficient: pi in q, or the coefficient given
            ft = self.opt['fourier_transform']
            reciprocity_coefficient = np.pi if ft.get('pi_in_q', False) else ft.get('reciprocity_coefficient', np.pi)
        if xray_wavelength is None:
            xray_wavelength = MTIP.mtip_data.get('xray_wavelength', 0.0)
        return IO.reconstruction_tree(dicts, float(xray_wavelength), float(reciprocity_coefficient), self.results.get('stats', {}))
