"""Device engine: a thin Python object around one ``mtip_ctx`` (one GPU, one stream, ``n_batch`` restarts).

Does the reference's one-off host setup (``hostsetup``), uploads it through the C ABI and exposes the loop
primitives (`run`, `shrinkwrap`, getters) plus the single operators used by ``operators.py`` and the tests.
"""
import ctypes as C

import numpy as np

from . import _lib, hostsetup as hs
from .settings import reciprocity_coefficient, resolve

METHOD_ID = {'HIO': 0, 'ER': 1, 'HIO_non_FXS': 2, 'ER_non_FXS': 3}


class Engine:
    def __init__(self, settings=None, data=None, n_batch=1, device=0, fused=True, lib_path=None,
                 n_radial=None, l_max=None, max_q=None):
        """settings: resolved (or override) settings dict; data: invariants dict (may be None for a
        transforms-only engine, then ``max_q`` must be given)."""
        self.lib = _lib.load(lib_path)
        self.device_index = int(device)
        self.emulated = hasattr(self.lib, 'mtip_emulated')          # the CPU emulation of tests/emul: its device memory is host memory
        opt = resolve(settings)
        self.opt = opt
        g = opt['grid']
        self.N = int(n_radial if n_radial is not None else g['n_radial_points'])
        self.L = int(l_max if l_max is not None else g['max_order'])
        self.kappa = float(reciprocity_coefficient(opt['fourier_transform']))
        self.mode = opt['fourier_transform']['type']
        if max_q is None:
            max_q = g['max_q']
            if not isinstance(max_q, float):                        # reconstruct.py:258-261
                if data is None:
                    raise ValueError('max_q is needed when no invariants are given')
                max_q = float(np.max(data['data_radial_points']))
        self.max_q = float(max_q)
        self.n_theta, self.n_phi = hs.angular_grid_size(self.L, g.get('n_theta', 0), g.get('n_phi', 0))
        self.B = int(n_batch)
        self.shape = (self.N, self.n_theta, self.n_phi)
        self.nlm = (self.L + 1) ** 2
        self.fused = bool(fused)
        if self.lib.mtip_device_count() <= 0:
            raise _lib.MtipError('no HIP device visible: the MTIP engine needs an MI355X (no CPU fallback)')
        cfg = _lib.MtipCfg(self.N, self.L, self.n_theta, self.n_phi, self.B, 1 if hs.hankel_skips_first_shell(self.mode) else 0,
                           1 if fused else 0, 0)
        self.ctx = self.lib.mtip_create(C.byref(cfg), int(device))
        self.group = None                                            # EngineGroup this engine steps with (side-by-side restart groups)
        if not self.ctx:
            raise _lib.MtipError('mtip_create: ' + self.lib.mtip_last_error(None).decode())
        # ---- transforms
        self.cos_theta, self.gauss_w, self.theta, self.phi = hs.gauss_grid(self.n_theta, self.n_phi)
        self._ck(self.lib.mtip_set_angular_grid(self.ctx, _lib.ptr(self.cos_theta), _lib.ptr(self.gauss_w)))
        self.rs, self.qs = hs.radial_grids(self.max_q, self.N, self.kappa, self.mode)
        self._ck(self.lib.mtip_set_radial_grid(self.ctx, _lib.ptr(self.rs), _lib.ptr(self.qs)))
        self.r_max = float(np.max(self.rs))                          # reconstruct.py:329
        self.raw_weights = hs.hankel_raw_weights(self.L, self.N, self.kappa, self.mode)
        fs, ivs = hs.hankel_scales(self.r_max, self.N, self.kappa, self.mode)
        self._ck(self.lib.mtip_set_hankel_weights(self.ctx, _lib.ptr(self.raw_weights), fs, ivs))
        self.int_wr, self.int_wt = hs.integrator_weights(self.rs, self.n_theta)
        self.default_sigma = np.pi / np.max(self.qs)                 # fxs_Projections.py:189-193
        self.rsetup = None
        if data is not None:
            self._setup_projections(data)

    # ------------------------------------------------------------------ setup
    def _ck(self, rc):
        if rc != 0:
            raise _lib.MtipError(f'libmtip_hip error {rc}: ' + self.lib.mtip_last_error(self.ctx).decode())

    def _setup_projections(self, data):
        opt = self.opt
        self.xray_wavelength = data.get('xray_wavelength', None)
        ropt = opt['projections']['reciprocal']
        rs_ = hs.reciprocal_setup(self.qs, data, self.L, ropt)      # (shared by the engines of one worker: read only)
        self.rsetup = rs_
        used_ids = set(rs_.used_orders.values())
        for l in range(self.L + 1):
            if l in used_ids:
                V = _lib.as_c128(rs_.projection_matrices[l])
                mask = _lib.as_u8(rs_.radial_mask[l])
                self._ck(self.lib.mtip_set_projection_matrix(self.ctx, l, _lib.ptr(V), V.shape[1], _lib.ptr(mask), 1))
            else:
                self._ck(self.lib.mtip_set_projection_matrix(self.ctx, l, None, 1, None, 0))
        self._ck(self.lib.mtip_set_number_of_particles(self.ctx, rs_.number_of_particles))
        self._ck(self.lib.mtip_set_so_freedom(self.ctx, int(rs_.so_order)))
        popt = opt['projections']['real']['projections']
        considered = opt['projections']['real']['HIO'].get('considered_projections', ['all'])
        flags, lo, hi, thr, hio = hs.real_constraint_flags(popt, considered)
        self._ck(self.lib.mtip_set_real_constraints(self.ctx, flags, lo, hi, thr, hio))
        auto = None
        if popt['support']['initial_support']['type'] == 'auto_correlation':
            auto = self.autocorrelation_guess()
        self.initial_support = hs.initial_support(self.rs, self.shape, popt, opt['particle_radius'], auto)
        s0 = _lib.as_u8(self.initial_support)
        self._ck(self.lib.mtip_set_initial_support(self.ctx, _lib.ptr(s0)))
        em = opt['main_loop']['error']['methods']
        inside = em['real'].get('l2_projection_diff', {}).get('inside_initial_support', False)
        gen = opt.get('general', {})
        wr, wt, use_mask = hs.error_weights(self.rs, self.n_theta, self.shape, inside, gen.get('cache_aware', True),
                                            gen.get('L2_cache', 512))
        self._ck(self.lib.mtip_set_error_weights(self.ctx, _lib.ptr(_lib.as_f64(wr)), _lib.ptr(_lib.as_f64(wt)), int(use_mask)))
        # metrics that are not on the accelerated path must not be dropped silently (fxs_IO_methods.py:690-703 lists them)
        for cat, known in (('real', ('l2_projection_diff',)), ('reciprocal', ('deg2_invariant_l2_diff', 'II_error', 'ccd_diff', 'fqc_error', 'l2_projection_diff',
                                                                                      'deg2_ranked_invariant_l2_diff'))):
            for name in em.get(cat, {}).get('calculate', []) or []:
                if name not in known:
                    if (cat, name) == ('real', 'support_size'):
                        raise NotImplementedError("real metric 'support_size' raises upstream as well (fxs_IO_methods.py:687 is handed the "
                                                  "projection's output list: AttributeError on .real)")
                    raise NotImplementedError('main_loop.error.methods.%s.calculate: %r is not built (DESIGN section 6); built: %s'
                                              % (cat, name, ', '.join(known)))
        rec_calc = list(em['reciprocal']['calculate'])
        # deg2_ranked_invariant_l2_diff (fxs_IO_methods.py:330-366): the entry of deg2_invariant_l2_diff of the best ranked even order
        # (fxs_invariant_tools.py:1467-1486) or of the order the option names -- a column of the same per-step history
        self.deg2_ranked_id = None
        if 'deg2_ranked_invariant_l2_diff' in rec_calc:
            order = em['reciprocal'].get('deg2_ranked_invariant_l2_diff', {}).get('order', False)
            if isinstance(order, (int, np.integer)) and not isinstance(order, (bool, np.bool_)):
                self.deg2_ranked_id = int(self.rsetup.used_orders[int(order)])
            else:
                orders = np.array(list(self.rsetup.used_orders.keys())).astype(int)
                self.deg2_ranked_id = int(hs.rank_projection_matrices_3d(self.rsetup.projection_matrices, orders, self.qs)[0])
        self.deg2_listed = 'deg2_invariant_l2_diff' in rec_calc
        self.deg2_enabled = self.deg2_listed or self.deg2_ranked_id is not None
        self._ck(self.lib.mtip_set_deg2_metric(self.ctx, int(self.deg2_enabled)))
        # reciprocal l2_projection_diff (fxs_IO_methods.py:301-310): the real metric's integrator without a support mask, i.e. with shell
        # N - 2 zeroed (the cache-aware branch integrates over the real grid, the plain one over the proportional reciprocal grid)
        self.reciprocal_l2 = 'l2_projection_diff' in rec_calc
        if self.reciprocal_l2:
            wr2, wt2, _ = hs.error_weights(self.rs, self.n_theta, self.shape, False)
            self._ck(self.lib.mtip_set_reciprocal_l2_metric(self.ctx, _lib.ptr(_lib.as_f64(wr2)), _lib.ptr(_lib.as_f64(wt2))))
        # II_error / ccd_diff / fqc_error (fxs_IO_methods.py:587-627, 651-683, 507-550): per step on the device from B_l
        self.invariant_metrics = [n for n in ('II_error', 'ccd_diff', 'fqc_error') if n in em['reciprocal']['calculate']]
        if self.invariant_metrics:
            if sorted(self.rsetup.used_orders.values()) != list(range(self.L + 1)):
                raise NotImplementedError('II_error / ccd_diff / fqc_error with a subset of the orders (upstream masks B_l of ALL orders with '
                                          'arrays shaped by the used ones, fxs_IO_methods.py:594-597, 610)')
            if self.xray_wavelength is None:
                raise KeyError("II_error / ccd_diff / fqc_error need data['xray_wavelength'] (fxs_IO_methods.py:511, 590, 657)")
            t = hs.invariant_metric_tables(self.invariant_metrics, self.qs, [self.rsetup.projection_matrices[l] for l in range(self.L + 1)],
                                           self.rsetup.radial_mask, float(self.xray_wavelength),
                                           em['reciprocal'].get('ccd_diff', {}).get('C_order', None))
            flags = sum(f for f, n in ((1, 'II_error'), (2, 'ccd_diff'), (4, 'fqc_error')) if n in self.invariant_metrics)

            def pp(key, conv):
                return _lib.ptr(conv(t[key])) if key in t else None
            self._im_keep = t                                         # (the arrays must outlive the call)
            self._ck(self.lib.mtip_set_invariant_metrics(
                self.ctx, flags, _lib.ptr(_lib.as_u8(t['zero_mask'])), pp('II_reference', _lib.as_c128), pp('qq', _lib.as_f64),
                pp('ccd_weights', _lib.as_f64), pp('ccd_reference', _lib.as_c128), float(t.get('ccd_norm', 0.0)), pp('fqc_P', _lib.as_f64),
                pp('fqc_reference_average', _lib.as_f64), pp('fqc_reference_weights', _lib.as_f64)))
        # generate_main_error_routine, fxs_IO_methods.py:746-765
        main = em.get('main', {'metrics': {'real': ['l2_projection_diff'], 'reciprocal': []}, 'type': 'mean'})
        real_m, rec_m = list(main['metrics'].get('real', [])), list(main['metrics'].get('reciprocal', []))
        types = {'mean': 0, 'min': 1, 'max': 2, 'prod': 3}
        if main.get('type', 'mean') not in types:
            raise ValueError('main error type %r (mean / min / max / prod)' % (main.get('type'),))
        if real_m == ['l2_projection_diff'] and not rec_m:
            self.main_is_reciprocal = False
        elif not real_m and rec_m == ['deg2_invariant_l2_diff']:
            if not self.deg2_listed:
                raise KeyError("main error uses 'deg2_invariant_l2_diff' but main_loop.error.methods.reciprocal.calculate does not list it")
            self.main_is_reciprocal = True
        elif real_m == ['l2_projection_diff'] and rec_m == ['deg2_invariant_l2_diff']:
            raise NotImplementedError('main error over l2_projection_diff (a scalar) AND deg2_invariant_l2_diff (one value per order): the '
                                      'reference itself raises there (np.array of an inhomogeneous list, fxs_IO_methods.py:760)')
        else:
            raise NotImplementedError('main error metrics %r / %r: only l2_projection_diff and deg2_invariant_l2_diff are on the '
                                      'accelerated path' % (real_m, rec_m))
        self._ck(self.lib.mtip_set_main_error(self.ctx, int(self.main_is_reciprocal), types[main.get('type', 'mean')]))

    def autocorrelation_guess(self):
        """reconstruct.py:400-420: ift(icht(V_l padded)).real"""
        c = np.zeros((self.B, self.N, self.nlm), complex)
        for l, p in enumerate(self.rsetup.full_projection_matrices):
            c[:, :, l * l:l * l + p.shape[1]] = p[None]
        return self.fourier_transform(self.sht_inverse(c), inverse=True)[0].real

    def close(self):
        if getattr(self, 'group', None) is not None:
            self.group.leave(self)
        if getattr(self, 'ctx', None):
            self.lib.mtip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ batch helpers
    def _bgrid(self, a):
        a = _lib.as_c128(a)
        if a.shape == self.shape:
            a = np.broadcast_to(a, (self.B,) + self.shape)
        assert a.shape == (self.B,) + self.shape, (a.shape, self.shape)
        return np.ascontiguousarray(a)

    def _bcoef(self, a):
        a = _lib.as_c128(a)
        if a.shape == (self.N, self.nlm):
            a = np.broadcast_to(a, (self.B, self.N, self.nlm))
        assert a.shape == (self.B, self.N, self.nlm), a.shape
        return np.ascontiguousarray(a)

    # ------------------------------------------------------------------ single operators (host arrays)
    def sht_forward(self, grid, prologue=0):
        g = self._bgrid(grid)
        out = np.empty((self.B, self.N, self.nlm), complex)
        self._ck(self.lib.mtip_op_sht_forward(self.ctx, _lib.ptr(g), _lib.ptr(out), prologue))
        return out

    def sht_inverse(self, coeff):
        c = self._bcoef(coeff)
        out = np.empty((self.B,) + self.shape, complex)
        self._ck(self.lib.mtip_op_sht_inverse(self.ctx, _lib.ptr(c), _lib.ptr(out)))
        return out

    def sht_inverse_forward(self, coeff, prologue=0):
        """grid = iSHT(coeff) and SHT(grid) (prologue 0) / SHT(|grid|^2) (prologue 1) in one kernel per shell."""
        c = self._bcoef(coeff)
        grid = np.empty((self.B,) + self.shape, complex)
        out = np.empty((self.B, self.N, self.nlm), complex)
        self._ck(self.lib.mtip_op_sht_inverse_forward(self.ctx, _lib.ptr(c), _lib.ptr(grid), _lib.ptr(out), prologue))
        return grid, out

    def hankel(self, coeff, inverse=False):
        c = self._bcoef(coeff)
        out = np.empty_like(c)
        self._ck(self.lib.mtip_op_hankel(self.ctx, _lib.ptr(c), _lib.ptr(out), int(inverse)))
        return out

    def fourier_transform(self, grid, inverse=False):
        g = self._bgrid(grid)
        out = np.empty_like(g)
        self._ck(self.lib.mtip_op_fourier_transform(self.ctx, _lib.ptr(g), _lib.ptr(out), int(inverse)))
        return out

    def project_coefficients(self, Ilm, real_intensity=False):
        """approximate_unknowns + mtip_projection.  real_intensity: the coefficients are those of a real grid (as in the
        phasing loop, SHT of |F|^2); only their m >= 0 half is read and real projection matrices take the real-arithmetic kernel."""
        c = self._bcoef(Ilm)
        out = np.empty_like(c)
        fn = self.lib.mtip_op_project_real_intensity if real_intensity else self.lib.mtip_op_project_coefficients
        self._ck(fn(self.ctx, _lib.ptr(c), _lib.ptr(out)))
        return out

    def apply_unknowns(self, Ilm, unknowns):
        """mtip_projection(Ilm, unknowns) (fxs_Projections.py:832-849): `unknowns` = per restart a sequence over l of
        (k_l, 2l+1) arrays, or one such sequence for all restarts."""
        c = self._bcoef(Ilm)
        if len(unknowns) == self.L + 1 and np.ndim(unknowns[0]) == 2:
            unknowns = [unknowns] * self.B
        flat = np.stack([np.concatenate([_lib.as_c128(u).reshape(-1) for u in per]) for per in unknowns])
        flat = np.ascontiguousarray(flat)
        out = np.empty_like(c)
        self._ck(self.lib.mtip_op_apply_unknowns(self.ctx, _lib.ptr(c), _lib.ptr(flat), _lib.ptr(out)))
        return out

    def modulus_replacement(self, F, I_new):
        f, i = self._bgrid(F), self._bgrid(I_new)
        out = np.empty_like(f)
        self._ck(self.lib.mtip_op_modulus_replacement(self.ctx, _lib.ptr(f), _lib.ptr(i), _lib.ptr(out)))
        return out

    def real_space_update(self, w, rho_prev, method, beta):
        w_, p_ = self._bgrid(w), self._bgrid(rho_prev)
        out = np.empty_like(w_)
        err = np.empty(self.B)
        self._ck(self.lib.mtip_op_real_space_update(self.ctx, _lib.ptr(w_), _lib.ptr(p_), METHOD_ID[method], float(beta),
                                                    _lib.ptr(out), _lib.ptr(err)))
        return out, err

    def deg2_invariants(self, Ilm):
        c = self._bcoef(Ilm)
        out = np.empty((self.B, self.L + 1, self.N, self.N), complex)
        self._ck(self.lib.mtip_op_deg2_invariants(self.ctx, _lib.ptr(c), _lib.ptr(out)))
        return out

    def apply_matrix(self, matrix, vects):
        m, v = _lib.as_f64(matrix), _lib.as_f64(vects)
        squeeze = v.ndim == 1
        if squeeze:
            v = np.ascontiguousarray(v[:, None])
        out = np.empty((m.shape[0], v.shape[1]))
        self._ck(self.lib.mtip_op_apply_matrix(self.ctx, _lib.ptr(m), _lib.ptr(v), _lib.ptr(out), m.shape[0], m.shape[1], v.shape[1]))
        return out[:, 0] if squeeze else out

    # ------------------------------------------------------------------ extract: B_l -> V_l (fxs_invariant_tools.py:1079-1207)
    def hermitian_eig(self, mats):
        """eigen-decomposition of a stack of Hermitian matrices (K, n, n) on the device: eigenvalues (K, n) in descending order,
        eigenvectors (K, n, n) with eigenvector i in [:, :, i] (numpy.linalg.eigh's layout, reversed order)"""
        m = _lib.as_c128(mats)
        K, n = m.shape[0], m.shape[1]
        if n <= 288 and not np.any(m.imag):
            # real symmetric (B_l of a real intensity): LDS-resident solver up to 128, column blocks over workgroups up to 288
            sym = np.ascontiguousarray((m.real + np.swapaxes(m.real, -1, -2)) / 2)
            vals = np.empty((K, n))
            vecs = np.empty((K, n, n))
            self._ck(self.lib.mtip_op_symmetric_eig(self.ctx, n, K, _lib.ptr(sym), _lib.ptr(vals), _lib.ptr(vecs)))
            order = np.argsort(vals, axis=1)[:, ::-1]
            vals = np.take_along_axis(vals, order, axis=1)
            vecs = np.take_along_axis(vecs, order[:, :, None], axis=1)          # rows = eigenvectors
            return vals, np.ascontiguousarray(np.swapaxes(vecs, -1, -2)).astype(complex)
        herm = np.ascontiguousarray((m + np.conj(np.swapaxes(m, -1, -2))) / 2)
        # the kernel rotates the columns of its column-major work matrix: hand it B^T = conj(B), whose eigenvectors are the
        # conjugates of B's
        vals = np.empty((K, n))
        vecs = np.empty((K, n, n), complex)
        self._ck(self.lib.mtip_op_hermitian_eig(self.ctx, n, K, _lib.ptr(np.ascontiguousarray(np.swapaxes(herm, -1, -2))),
                                                _lib.ptr(vals), _lib.ptr(vecs)))
        order = np.argsort(vals, axis=1)[:, ::-1]
        vals = np.take_along_axis(vals, order, axis=1)
        vecs = np.take_along_axis(vecs, order[:, :, None], axis=1)          # rows = eigenvectors
        return vals, np.ascontiguousarray(np.swapaxes(vecs, -1, -2))

    def extract_projection_matrices(self, Bl, orders=None):
        """deg2_invariant_to_projection_matrices_3d (fxs_invariant_tools.py:1171-1207) for B_l (L+1, Nq, Nq): V_l = the top
        min(2l+1, Nq) eigenvectors scaled by sqrt(eigenvalue), negative eigenvalues clipped.  Returns (list of V_l, list of
        eigenvalues)."""
        Bl = _lib.as_c128(Bl)
        orders = range(Bl.shape[0]) if orders is None else orders
        vals, vecs = self.hermitian_eig(Bl)
        pms, evs = [], []
        for i, l in enumerate(orders):
            k = min(Bl.shape[1], 2 * l + 1)
            w, v = vals[i, :k].copy(), vecs[i][:, :k].copy()
            neg = w < 0
            w[neg] = 0
            v[:, neg] = 0
            pms.append((v * np.sqrt(w)[None, :]).astype(complex))
            evs.append(w)
        return pms, evs

    # ------------------------------------------------------------------ rotational alignment (average.py:920-960)
    def _so3_setup(self):
        if not getattr(self, '_so3_ready', False):
            tab = np.ascontiguousarray(hs.wigner_d_table(self.L))
            self._ck(self.lib.mtip_set_so3_tables(self.ctx, self.L + 1, _lib.ptr(tab)))
            self._so3_ready = True

    def so3_correlation(self, ref_coeff, sig_coeff, r_limit_ids=None):
        """C[b, alpha, beta, gamma] = mean_r Re <ref_r, R sig_r> on the (2 bw)^3 Euler grid (soft.calc_mean_C)"""
        self._so3_setup()
        ref = np.ascontiguousarray(_lib.as_c128(ref_coeff).reshape(self.N, self.nlm))
        sig = self._bcoef(sig_coeff)
        lo, hi = (0, self.N) if r_limit_ids is None else (int(r_limit_ids[0]), int(r_limit_ids[1]))
        nb = 2 * (self.L + 1)
        out = np.empty((self.B, nb, nb, nb))
        self._ck(self.lib.mtip_op_so3_correlation(self.ctx, _lib.ptr(ref), _lib.ptr(sig), lo, hi, _lib.ptr(out)))
        return out

    def rotate_coefficients(self, coeff, eulers):
        """f_lm -> sum_n D^l_mn(euler_b) f_ln per restart (soft.rotate_coeff); eulers: (3,) or (B, 3)"""
        self._so3_setup()
        c = self._bcoef(coeff)
        eulers = np.broadcast_to(np.asarray(eulers, dtype=float), (self.B, 3))
        D = np.ascontiguousarray(np.stack([hs.wigner_D_flat(self.L, e) for e in eulers]))
        out = np.empty_like(c)
        self._ck(self.lib.mtip_op_rotate_coefficients(self.ctx, _lib.ptr(c), _lib.ptr(D), _lib.ptr(out)))
        return out

    # ------------------------------------------------------------------ the same operators on device-resident batches
    # torch tensors hold the memory (complex128, contiguous, leading dimension n_batch) on the context's device -- CPU tensors
    # under the emulation; the operators read and write them in place in HBM, nothing crosses PCIe.
    def torch_device(self):
        import torch
        return torch.device('cpu') if self.emulated else torch.device('cuda', self.device_index)

    @staticmethod
    def _tp(t):
        assert t.is_contiguous()
        return C.c_void_p(t.data_ptr())

    def t_fourier_transform(self, g, inverse=False):
        import torch
        assert tuple(g.shape) == (self.B,) + self.shape and g.dtype == torch.complex128
        out = torch.empty_like(g)
        self._ck(self.lib.mtip_op_fourier_transform(self.ctx, self._tp(g), self._tp(out), int(inverse)))
        return out

    def t_sht_forward(self, g):
        import torch
        assert tuple(g.shape) == (self.B,) + self.shape and g.dtype == torch.complex128
        out = torch.empty((self.B, self.N, self.nlm), dtype=torch.complex128, device=g.device)
        self._ck(self.lib.mtip_op_sht_forward(self.ctx, self._tp(g), self._tp(out), 0))
        return out

    def t_sht_inverse(self, c):
        import torch
        assert tuple(c.shape) == (self.B, self.N, self.nlm) and c.dtype == torch.complex128
        out = torch.empty((self.B,) + self.shape, dtype=torch.complex128, device=c.device)
        self._ck(self.lib.mtip_op_sht_inverse(self.ctx, self._tp(c), self._tp(out)))
        return out

    # ---- grid arithmetic of the averaging worker on device stacks (csrc/k_average.hip): torch owns the memory, nothing else
    def t_grid_stats(self, X, ref=None):
        """(n, 12) host array per grid of the stack X (n, Nq, n_theta, n_phi): integrals, moments, extrema (mtip_op_grid_stats)"""
        import torch
        assert X.dtype == torch.complex128 and tuple(X.shape[1:]) == self.shape and X.is_contiguous()
        out = np.zeros((X.shape[0], 12))
        wr, wt = _lib.as_f64(self.int_wr), _lib.as_f64(self.int_wt)
        self._ck(self.lib.mtip_op_grid_stats(self.ctx, self._tp(X), int(X.shape[0]), self._tp(ref) if ref is not None else None,
                                             _lib.ptr(wr), _lib.ptr(wt), _lib.ptr(out)))
        return out

    def t_phase_ramp(self, X, centers_cartesian, sign):
        """X[b] *= exp(-i sign k . c_b) in place (generate_shift_by_operator, fxs_Projections.py:1419-1444)"""
        import torch
        assert X.dtype == torch.complex128 and tuple(X.shape[1:]) == self.shape and X.is_contiguous()
        cc = np.ascontiguousarray(centers_cartesian, dtype=np.float64)
        assert cc.shape == (X.shape[0], 3)
        self._ck(self.lib.mtip_op_grid_phase_ramp(self.ctx, self._tp(X), int(X.shape[0]), _lib.ptr(cc), float(sign)))
        return X

    def t_combine(self, op, A, scalars=None):
        """mtip_op_grid_combine on a stack A (n, ...): 'conj', 'scale' (complex per grid), 'sum', 'abs2sum' (one grid out), 'affine'"""
        import torch
        code = {'conj': 0, 'scale': 1, 'sum': 2, 'abs2sum': 3, 'affine': 4}[op]
        assert A.dtype == torch.complex128 and tuple(A.shape[1:]) == self.shape and A.is_contiguous()
        out = torch.empty(self.shape if code in (2, 3) else tuple(A.shape), dtype=torch.complex128, device=A.device)
        sc = None if scalars is None else np.ascontiguousarray(scalars, dtype=np.complex128)
        self._ck(self.lib.mtip_op_grid_combine(self.ctx, code, self._tp(out), self._tp(A), int(A.shape[0]), _lib.ptr(sc) if sc is not None else None))
        return out

    def t_prtf(self, a1, a2, I1, I2):
        """resolution_metrics.py:62-78: per-shell mean (complex) and standard deviation of sqrt(a1 conj(a2) / sqrt(I1 I2))"""
        mean = np.zeros(self.N, complex)
        std = np.zeros(self.N)
        for x in (a1, a2, I1, I2):
            assert tuple(x.shape) == self.shape and x.is_contiguous()
        self._ck(self.lib.mtip_op_prtf(self.ctx, self._tp(a1), self._tp(a2), self._tp(I1), self._tp(I2), _lib.ptr(mean), _lib.ptr(std)))
        return mean, std

    def t_state(self, kind, batch, best=False):
        """a grid of the loop state -- 'density', 'reciprocal_density' (complex128) or 'support' (uint8), current or best pair -- as a
        tensor ON THE ENGINE'S DEVICE (device-to-device copy out of the slot arrays; the getters of the C ABI take a device
        destination as well as a host one).  What the end-of-run gather sends from (parallel.gather_results)."""
        import torch
        dev = self.torch_device()
        if kind == 'support':
            out = torch.empty(self.shape, dtype=torch.uint8, device=dev)
            self._ck(self.lib.mtip_get_support(self.ctx, int(batch), int(best), self._tp(out)))
            return out
        fn = {'density': self.lib.mtip_get_density, 'reciprocal_density': self.lib.mtip_get_reciprocal_density}[kind]
        out = torch.empty(self.shape, dtype=torch.complex128, device=dev)
        self._ck(fn(self.ctx, int(batch), int(best), self._tp(out)))
        return out

    def t_find_rotation(self, ref_coeff, sig_coeff, r_limit_ids=None, keep_metric=False):
        """find_rotation (average.py:920-947) for a batch: (arg-max in the reference's reading order as (i_beta, i_alpha, i_gamma)
        per restart, the maxima, the correlation (B, nb, nb, nb) [alpha, beta, gamma] on the device or None)"""
        import torch
        self._so3_setup()
        assert tuple(ref_coeff.shape) == (self.N, self.nlm) and tuple(sig_coeff.shape) == (self.B, self.N, self.nlm)
        lo, hi = (0, self.N) if r_limit_ids is None else (int(r_limit_ids[0]), int(r_limit_ids[1]))
        nb = 2 * (self.L + 1)
        arg = np.empty(self.B, np.int64)
        vmax = np.empty(self.B)
        Cm = torch.empty((self.B, nb, nb, nb), dtype=torch.float64, device=sig_coeff.device) if keep_metric else None
        self._ck(self.lib.mtip_op_so3_find_rotation(self.ctx, self._tp(ref_coeff), self._tp(sig_coeff), lo, hi, _lib.ptr(arg),
                                                    _lib.ptr(vmax), self._tp(Cm) if keep_metric else None))
        return np.stack(np.unravel_index(arg, (nb, nb, nb)), axis=1), vmax, Cm

    def t_rotate_grid(self, coeff, beta_index, alpha, gamma):
        """rotate (average.py:948-960) by Euler angles (alpha[b], beta sample beta_index[b], gamma[b])"""
        import torch
        self._so3_setup()
        assert tuple(coeff.shape) == (self.B, self.N, self.nlm) and coeff.dtype == torch.complex128
        bi = np.ascontiguousarray(beta_index, dtype=np.int32)
        al, ga = _lib.as_f64(alpha), _lib.as_f64(gamma)
        assert bi.shape == (self.B,) and al.shape == (self.B,) and ga.shape == (self.B,)
        out = torch.empty_like(coeff)
        self._ck(self.lib.mtip_op_rotate_coefficients_grid(self.ctx, self._tp(coeff), _lib.ptr(bi), _lib.ptr(al), _lib.ptr(ga), self._tp(out)))
        return out

    # ------------------------------------------------------------------ state and loop
    def set_density(self, batch, rho):
        r = _lib.as_c128(rho)
        assert r.shape == self.shape
        self._ck(self.lib.mtip_set_density(self.ctx, batch, _lib.ptr(r)))

    def init_state(self):
        self._ck(self.lib.mtip_init_state(self.ctx))

    def reset_support(self):
        """back to the initial support, enforced, for every restart (the state a fresh reconstruction starts from,
        fxs_Projections.py:34, 133-155)"""
        s0 = _lib.as_u8(self.initial_support)
        self._ck(self.lib.mtip_set_initial_support(self.ctx, _lib.ptr(s0)))

    def density(self, batch, best=False):
        out = np.empty(self.shape, complex)
        self._ck(self.lib.mtip_get_density(self.ctx, batch, int(best), _lib.ptr(out)))
        return out

    def reciprocal_density(self, batch, best=False):
        out = np.empty(self.shape, complex)
        self._ck(self.lib.mtip_get_reciprocal_density(self.ctx, batch, int(best), _lib.ptr(out)))
        return out

    def support(self, batch, best=False):
        out = np.empty(self.shape, np.uint8)
        self._ck(self.lib.mtip_get_support(self.ctx, batch, int(best), _lib.ptr(out)))
        return out.astype(bool)

    def set_support(self, batch, support, enforce_initial_support):
        s = _lib.as_u8(support)
        self._ck(self.lib.mtip_set_support(self.ctx, batch, _lib.ptr(s), int(bool(enforce_initial_support))))

    def unknowns(self, batch):
        out = []
        for l in range(self.L + 1):
            k = min(2 * l + 1, self.N)
            u = np.empty((k, 2 * l + 1), complex)
            self._ck(self.lib.mtip_get_unknowns(self.ctx, batch, l, _lib.ptr(u)))
            out.append(u)
        return tuple(out)

    def best_error(self):
        out = np.empty(self.B)
        n = C.c_int64(0)
        self._ck(self.lib.mtip_get_best_error(self.ctx, _lib.ptr(out), C.byref(n)))
        return out, int(n.value)

    def select_best(self, where=None):
        """reconstruct.py:945-949 for all restarts, or for those flagged in `where` (bool per restart)."""
        if where is None:
            self._ck(self.lib.mtip_select_best(self.ctx))
        else:
            w = np.ascontiguousarray(np.asarray(where, dtype=np.uint8))
            assert w.shape == (self.B,)
            self._ck(self.lib.mtip_select_best_where(self.ctx, _lib.ptr(w)))

    def set_ft_stab_mask(self, mask):
        """ft_stab per restart for the runs that follow (bool per restart), None = every restart (mtip_set_ft_stab_mask)"""
        m = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=bool).astype(np.uint8))
        assert m is None or m.shape == (self.B,)
        self._ck(self.lib.mtip_set_ft_stab_mask(self.ctx, _lib.ptr(m)))
        self._ft_mask_set = m is not None

    def run(self, method, ft_stab, betas, fetch=True):
        """n = len(betas) steps of `method`; ft_stab: bool, or a bool per restart (the reference links ft_stab to the shrink-wrap's
        enforce decision per reconstruction process: restarts of a batch may disagree)"""
        betas = _lib.as_f64(np.atleast_1d(betas))
        n = len(betas)
        mixed = None
        if isinstance(ft_stab, np.ndarray):
            flags = np.asarray(ft_stab, dtype=bool).reshape(self.B)
            if flags.all() or not flags.any():
                ft_stab = bool(flags.all())
            else:
                mixed, ft_stab = flags, True
        if mixed is not None:
            self.set_ft_stab_mask(mixed)
        elif getattr(self, '_ft_mask_set', False):
            self.set_ft_stab_mask(None)                              # a plain flag again: every restart
        if self.group is not None:
            # side by side with other engines of this GPU: the group enqueues everybody's steps in turn order (EngineGroup); a
            # per-restart ft_stab mask is state of this engine's context, the call itself says ft_stab = 1
            best = np.empty(self.B)
            first = C.c_int64(0)
            if fetch:
                self._ck(self.lib.mtip_get_best_error(self.ctx, _lib.ptr(best), C.byref(first)))
            self.group.member_run(self, method, ft_stab, betas)
            if not fetch:
                return None, None
            return self.fetch_errors(first.value, n)
        if not fetch:
            self._ck(self.lib.mtip_run_async(self.ctx, METHOD_ID[method], int(bool(ft_stab)), n, _lib.ptr(betas)))
            return None, None
        err = np.empty((n, self.B))
        deg2 = np.empty((n, self.B, self.L + 1)) if self.deg2_enabled else None
        self._ck(self.lib.mtip_run(self.ctx, METHOD_ID[method], int(bool(ft_stab)), n, _lib.ptr(betas), _lib.ptr(err), _lib.ptr(deg2)))
        return err, deg2

    def fetch_errors(self, first, n):
        err = np.empty((n, self.B))
        deg2 = np.empty((n, self.B, self.L + 1)) if self.deg2_enabled else None
        self._ck(self.lib.mtip_fetch_errors(self.ctx, first, n, _lib.ptr(err), _lib.ptr(deg2)))
        return err, deg2

    def fetch_reciprocal_l2(self, first, n):
        """reciprocal l2_projection_diff of the steps [first, first + n): (n, B), or None when the metric is off"""
        if not self.reciprocal_l2:
            return None
        out = np.empty((n, self.B))
        self._ck(self.lib.mtip_fetch_reciprocal_l2_metric(self.ctx, first, n, _lib.ptr(out)))
        return out

    def fetch_invariant_metrics(self, first, n):
        """{'II_error': (n, B), 'ccd_diff': (n, B), 'fqc_error': (n, B, Nq)} for the enabled ones"""
        out = {}
        if not self.invariant_metrics:
            return out
        II = np.empty((n, self.B)) if 'II_error' in self.invariant_metrics else None
        ccd = np.empty((n, self.B)) if 'ccd_diff' in self.invariant_metrics else None
        fqc = np.empty((n, self.B, self.N)) if 'fqc_error' in self.invariant_metrics else None
        self._ck(self.lib.mtip_fetch_invariant_metrics(self.ctx, first, n, _lib.ptr(II), _lib.ptr(ccd), _lib.ptr(fqc)))
        for k, v in (('II_error', II), ('ccd_diff', ccd), ('fqc_error', fqc)):
            if v is not None:
                out[k] = v
        return out

    def invariant_metrics_of(self, Ilm):
        """the enabled metrics of given intensity coefficients: {'II_error': (B,), 'ccd_diff': (B,), 'fqc_error': (B, Nq)}"""
        c = self._bcoef(Ilm)
        II, ccd, fqc = np.empty(self.B), np.empty(self.B), np.empty((self.B, self.N))
        self._ck(self.lib.mtip_op_invariant_metrics(self.ctx, _lib.ptr(c), _lib.ptr(II), _lib.ptr(ccd), _lib.ptr(fqc)))
        return {k: v for k, v in (('II_error', II), ('ccd_diff', ccd), ('fqc_error', fqc)) if k in self.invariant_metrics}

    def fetch_main_errors(self, first, n):
        err = np.empty((n, self.B))
        self._ck(self.lib.mtip_fetch_main_errors(self.ctx, first, n, _lib.ptr(err)))
        return err

    def shrinkwrap(self, sigma, threshold, error_limit):
        enforced = np.empty(self.B, np.uint8)
        self._ck(self.lib.mtip_shrinkwrap(self.ctx, float(sigma), float(threshold), float(error_limit), _lib.ptr(enforced)))
        return enforced.astype(bool)

    def begin_sub_loop(self):
        """top of a sub-loop call (reconstruct.py:859, 866): stale `hist` and latest_intensity are reset"""
        self._ck(self.lib.mtip_begin_sub_loop(self.ctx))

    def refresh_reciprocal_density(self):
        """'SW_center' tail (reconstruct.py:891-894), literally: the last pair becomes (reciprocal, real) = (rho, FT(rho))."""
        self._ck(self.lib.mtip_refresh_reciprocal_density(self.ctx))

    def last_deg2_invariant(self, batch):
        out = np.empty((self.L + 1, self.N, self.N), complex)
        self._ck(self.lib.mtip_last_deg2_invariant(self.ctx, batch, _lib.ptr(out)))
        return out

    def synchronize(self):
        self._ck(self.lib.mtip_synchronize(self.ctx))

    # ------------------------------------------------------------------ profiling
    def profile(self, enable=True):
        self._ck(self.lib.mtip_profile(self.ctx, int(enable)))
        self._ck(self.lib.mtip_profile_reset(self.ctx))

    def profile_get(self, name):
        ms, n = C.c_double(0), C.c_int64(0)
        self._ck(self.lib.mtip_profile_get(self.ctx, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def debug_spin(self, microseconds):
        """enqueue a kernel that spins for `microseconds` on this engine's stream (asynchronous; see streams_side_by_side)"""
        self._ck(self.lib.mtip_debug_spin(self.ctx, float(microseconds)))

    def jacobi_sweeps(self):
        out = np.zeros((self.B, self.L + 1), np.int32)
        self._ck(self.lib.mtip_debug_jacobi_sweeps(self.ctx, _lib.ptr(out)))
        return out & 0xff

    def jacobi_active_columns(self):
        """columns of X_l the last polar-factor call still rotated in its final sweep (numerical rank estimate)"""
        out = np.zeros((self.B, self.L + 1), np.int32)
        self._ck(self.lib.mtip_debug_jacobi_sweeps(self.ctx, _lib.ptr(out)))
        return (out >> 8) & 0xffff

    def projection_slots(self):
        """workgroups per restart of the real projection kernel in the last call (0: general complex kernels)"""
        n = self.lib.mtip_debug_projection_slots(self.ctx)
        if n < 0:
            self._ck(n)
        return int(n)

    def jacobi_closing_step(self):
        """how the real-arithmetic projection closed its sweeps in the last call, per (restart, order): 0 = classic confirming sweep,
        1 = first-order, 2 = second-order polar step on the Gram matrix (k_projr.hip)"""
        out = np.zeros((self.B, self.L + 1), np.int32)
        self._ck(self.lib.mtip_debug_jacobi_sweeps(self.ctx, _lib.ptr(out)))
        return (out >> 24) & 3

    # ------------------------------------------------------------------ helpers shared with synthetic.py
    @property
    def thetas(self):
        return self.theta

    @property
    def phis(self):
        return self.phi

    def ft(self, grid):
        """transforms adapter for synthetic.make_invariants (batch element 0)."""
        return self.fourier_transform(grid)[0]

    def forward_l(self, grid):
        c = self.sht_forward(grid)[0]
        return [np.array(c[:, l * l:(l + 1) ** 2]) for l in range(self.L + 1)]


class EngineGroup:
    """Engines of one GPU that step side by side (the reference's `GPU.n_gpu_workers` restart groups, reconstruct.py:104), run
    through mtip_run_group_async: the same steps per engine, enqueued so that the engines take turns at the chip-filling
    transforms while the others' projections run beside them (mtip_api.hip).

    ``run(method, ft_stab, betas)`` is the direct form (one host thread drives all engines: bench.py).  A worker whose restart
    groups run the reference's loop each in their own host thread attaches the engines (``attach``): every ``Engine.run`` then
    meets the others here, and the last one to arrive enqueues for all that asked for the same steps (engines that asked for
    something else -- a data-dependent ft_stab decision, say -- are enqueued on their own).  A member that is done ``leave``s;
    nobody waits for a member that has left, and a wait that lasts longer than `patience` seconds stops waiting and enqueues alone."""

    def __init__(self, engines=(), patience=30.0):
        import threading
        self.engines = list(engines)
        self.cond = threading.Condition()
        self.members = []
        self.pending = {}
        self.generation = 0
        self.patience = patience
        self.calls = {'group': 0, 'single': 0}

    # -- direct form
    def run(self, method, ft_stab, betas):
        betas = _lib.as_f64(np.atleast_1d(betas))
        self._enqueue(self.engines, method, ft_stab, betas)

    def _enqueue(self, engines, method, ft_stab, betas):
        e0 = engines[0]
        if len(engines) == 1:
            e0._ck(e0.lib.mtip_run_async(e0.ctx, METHOD_ID[method], int(bool(ft_stab)), len(betas), _lib.ptr(betas)))
            self.calls['single'] += 1
            return
        import ctypes as C
        arr = (C.c_void_p * len(engines))(*[e.ctx for e in engines])
        rc = e0.lib.mtip_run_group_async(arr, len(engines), METHOD_ID[method], int(bool(ft_stab)), len(betas), _lib.ptr(betas))
        self.calls['group'] += 1
        if rc != 0:
            for e in engines:                       # the context that failed holds the message
                msg = e.lib.mtip_last_error(e.ctx).decode()
                if msg:
                    raise RuntimeError('mtip_run_group_async: %s' % msg)
            raise RuntimeError('mtip_run_group_async failed with code %d' % rc)

    # -- rendezvous form (one host thread per engine)
    def attach(self, engine):
        with self.cond:
            self.members.append(engine)
            engine.group = self

    def leave(self, engine):
        with self.cond:
            if engine in self.members:
                self.members.remove(engine)
            engine.group = None
            self.pending.pop(id(engine), None)
            self._flush_if_complete()

    def _flush_if_complete(self):
        if not self.pending or len(self.pending) < len(self.members):
            return
        self._flush()

    def _flush(self):
        todo = list(self.pending.values())
        self.pending.clear()
        self.generation += 1
        errors = []
        while todo:
            e, method, ft_stab, betas = todo[0]
            same = [t for t in todo if t[1] == method and t[2] == ft_stab and np.array_equal(t[3], betas)]
            todo = [t for t in todo if not any(t is u for u in same)]
            try:
                self._enqueue([t[0] for t in same], method, ft_stab, betas)
            except Exception as ex:                 # delivered to every waiting member: nobody is left hanging
                errors.append(ex)
        self.error = errors[0] if errors else None
        self.cond.notify_all()

    def member_run(self, engine, method, ft_stab, betas):
        import time
        with self.cond:
            self.pending[id(engine)] = (engine, method, bool(ft_stab), betas)
            gen = self.generation
            if len(self.pending) >= len(self.members):
                self._flush()
            else:
                deadline = time.monotonic() + self.patience
                while self.generation == gen:
                    left = deadline - time.monotonic()
                    if left <= 0:                    # the others are busy elsewhere: go alone
                        self.pending.pop(id(engine), None)
                        self._enqueue([engine], method, ft_stab, betas)
                        return
                    self.cond.wait(left)
            if getattr(self, 'error', None) is not None:
                raise self.error


def streams_side_by_side(engines, microseconds=2000.0, reps=3):
    """How many of the engines' streams execute at the same time: every engine gets `reps` spin kernels, the ratio of the
    serial time to the measured wall time is the number of chains that ran side by side.  HIP maps streams onto a pool of
    hardware queues (GPU_MAX_HW_QUEUES); two engines on one queue serialise, which no kernel timing shows directly."""
    import time
    for e in engines:
        e.debug_spin(10.0)
    for e in engines:
        e.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for e in engines:
            e.debug_spin(microseconds)
    for e in engines:
        e.synchronize()
    wall = time.perf_counter() - t0
    return len(engines) * reps * microseconds * 1e-6 / wall
