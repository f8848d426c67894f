"""``fxs average`` on the MI355X engine -- host-side mirror of ``xframe/projects/fxs/average.py`` (``ProjectWorker.run_3d``
359-627, ``Alignment`` 729-1111): centre the reconstructions, normalise them, align every one (and its point inverse) to
the reference by the SO(3) correlation of their harmonic coefficients, average the aligned ones and compute the PRTF.

Where the reference forks one process per reconstruction and runs numpy / shtns / pysofft in each, here the batch of
reconstructions stays in HBM from the first upload to the last download: torch tensors OWN THE MEMORY (allocation, stacking,
slicing, the collectives) and every piece of arithmetic on it is a HIP kernel of the engine behind the C ABI, working on those
tensors in place (``Engine.t_*``): the transforms (FT, SHT), the SO(3) correlation with its arg-max and the rotation of the
coefficients (``csrc/k_align.hip``), and the glue between them -- centre-of-mass and error integrals, extrema and the
normalisation factors (``mtip_op_grid_stats``), phase ramps (``mtip_op_grid_phase_ramp``), conjugation / scaling / the sums over the
selected alignments / normalisation (``mtip_op_grid_combine``), the PRTF shell statistics (``mtip_op_prtf``; ``csrc/k_average.hip``).
What crosses PCIe: the inputs once (unless they are device tensors already), a few scalars per
reconstruction (centres, maxima, arg-max indices, errors -- the decisions are host logic, as they depend on each other in the
reference's order), and the results.  With several ranks (``torch.distributed``: one process per GPU) every rank aligns its own
restarts against the reference, which its owner broadcasts, and the sums of the aligned densities are all-reduced from the
device buffers (RCCL over xGMI) -- the only collective of the ``reconstruct -> average`` pipeline that moves grid-sized data.

pysofft (the reference's SO(3) library) is not available: conventions of the correlation / rotation are those of
``oracle/alignment.py`` (see its parity note); only their composition enters the result.  The flow around those two calls is
held to the reference's own ``run_3d`` / ``Alignment`` (fixture G17, tests/golden/average_flow.npz), including three things the
reference does that change its results: ``find_rotation`` flips the angles of the Euler-grid entry it found IN PLACE
(average.py:938-940), so a grid point found twice by one ``Alignment`` object hands out un-flipped angles the second time; the
list of valid alignments starts with the reference while the list of their errors does not (519-524); and the averaged pair is
centred before the metrics with a shift operator that works in place (538, fxs_Projections.py:1442), so the saved averaged
reciprocal density and the 'PRTF' metric see the shifted one.
"""
import numpy as np

from . import hostsetup as hs

DEFAULTS = {
    'center_reconstructions': True, 'normalize_reconstructions': {'use': True, 'mode': 'max'}, 'pointinvert_reference': False,
    'alignment_error_limit': 0.5, 'max_iterations': 1, 'find_rotation': {}, 'resolution_metrics': {'PRTF': True},
    'selection': {'n_reconstructions': 100}, 'average_normalization_min': 0,
}


class Alignment:
    """average.py:729-1111 on a transforms engine (``Engine(settings, None, n_batch=B, max_q=...)`` of the reconstruction grid).
    Works on torch tensors of the engine's device: stacks (n, Nq, n_theta, n_phi) complex128."""

    def __init__(self, engine, opt=None, device=None):
        import torch
        self.torch = torch
        self.e = e = engine
        self.opt = dict(DEFAULTS)
        self.opt.update(opt or {})
        self.L = e.L
        self.dev = torch.device(device) if device is not None else e.torch_device()
        self.soft_grid = np.stack(np.meshgrid(*hs.euler_grid(self.L + 1), indexing='ij'), -1)     # make_SO3_grid; edited in place
        self.results = {}
        # integrals, centres of mass, phase ramps, sums and the PRTF statistics are kernels of the engine (csrc/k_average.hip,
        # Engine.t_grid_stats / t_phase_ramp / t_combine / t_prtf): torch only owns the memory of the stacks
        self.volume = 4 / 3 * np.pi * np.max(e.rs) ** 3

    # -- helpers
    def to_device(self, x):
        t = self.torch
        if not t.is_tensor(x):
            x = t.from_numpy(np.ascontiguousarray(x, dtype=np.complex128))
        return x.to(self.dev, dtype=t.complex128)

    def _batched(self, fn, X):
        """X (n, ...) -> fn over chunks of the engine's batch size (the last chunk padded with its last item)"""
        t, B, n = self.torch, self.e.B, X.shape[0]
        out = []
        for i in range(0, n, B):
            chunk = X[i:i + B]
            m = chunk.shape[0]
            if m < B:
                chunk = t.cat([chunk, chunk[-1:].expand(B - m, *chunk.shape[1:])])
            out.append(fn(chunk.contiguous())[:m])
        return t.cat(out) if len(out) > 1 else out[0]

    def ft(self, X):
        return self._batched(lambda g: self.e.t_fourier_transform(g), X)

    def ift(self, X):
        return self._batched(lambda g: self.e.t_fourier_transform(g, True), X)

    def sht(self, X):
        return self._batched(lambda g: self.e.t_sht_forward(g), X)

    def isht(self, Cf):
        return self._batched(lambda c: self.e.t_sht_inverse(c), Cf)

    def stats(self, X, ref=None):
        """mtip_op_grid_stats of a stack (any length): (n, 12) on the host"""
        return self.e.t_grid_stats(X.contiguous(), ref)

    def centers(self, R):
        """generate_calc_center (misk.py:295-312): centres of mass of Re(rho), spherical coordinates, (n, 3) on the host"""
        m = self.stats(R)[:, :4]
        total = np.where(m[:, 0] == 0, 1.0, m[:, 0])
        return np.stack([hs.cartesian_to_spherical(m[i, 1:] / total[i]) for i in range(len(m))])

    def shift(self, X, vectors, opposite_direction=False):
        """generate_shift_by_operator (fxs_Projections.py:1419-1444): X[b] *= exp(-i s k.c_b) for spherical vectors (n, 3), in place"""
        c = np.stack([hs.spherical_to_cartesian(np.asarray(v, dtype=float)) for v in vectors])
        return self.e.t_phase_ramp(X, c, -1.0 if opposite_direction else 1.0)

    def shift_to_center(self, R, F):
        """assemble_shift_to_center (1007-1020) for stacks: (IFT(FT(rho) e^{i k c}), F e^{i k c}, c)"""
        c = self.centers(R)
        return self.ift(self.shift(self.ft(R), c, True)), self.shift(F.clone(), c, True), c

    def pick_rotation(self, am):
        """find_rotation (936-946), literally: the grid entry at the arg-max (i_beta, i_alpha, i_gamma) is a VIEW and is flipped in place"""
        euler = self.soft_grid[am[1], am[0], am[2]]
        euler[0] = 2 * np.pi - euler[0]
        euler[2] = 2 * np.pi - euler[2]
        return euler

    def metric_layout(self, C):
        """a correlation of the device ([alpha, beta, gamma]) in the layout the reference stores it in (oracle mean_C_layout)"""
        t = self.torch
        n = C.shape[0]
        flip = t.from_numpy((-np.arange(n)) % n).to(C.device)
        return C.index_select(0, flip).index_select(2, flip).permute(1, 0, 2)

    def apply_to(self, reference, S_rho, S_F):
        """alignment_routine (1089-1109) for stacks of signals: each signal and its point inverse are aligned (rotate_signal sketch
        970-975: the rotation found on the densities is applied to both halves), the one with the smaller difference to the
        reference is kept.  Transforms, correlations and rotations run batched over all signals and their inverses; the angles are
        picked in the reference's order (signal 0, its inverse, signal 1, ...) because picking edits the grid.  Returns a list of
        dicts like the reference's (tensors on the device)."""
        t, e = self.torch, self.e
        n = S_rho.shape[0]
        keep_metric = bool(self.opt.get('keep_rotation_metrics', True))
        inv_d = self.ift(e.t_combine('conj', self.ft(S_rho)))
        both_rho = t.cat([S_rho, inv_d])
        both_F = t.cat([S_F, e.t_combine('conj', S_F.contiguous())])
        norm = float(self.stats(reference[None])[0, 4]) / self.volume       # integrate_normed(Re(reference)^2)
        norm = norm if norm != 0 else 1
        ref_c = self.sht(reference[None])[0].contiguous()
        sig_c = self.sht(both_rho)
        ft_c = self.sht(both_F)
        r_lim = self.opt['find_rotation'].get('r_limit_ids', [0, e.N])
        r_lim = [int(r_lim[0]), int(r_lim[1])]                      # the reference reads entries 0 and 1 (soft_plugin.py:92-94)
        args, metrics = [], []
        B = e.B
        for i in range(0, 2 * n, B):
            chunk = sig_c[i:i + B]
            m = chunk.shape[0]
            if m < B:
                chunk = t.cat([chunk, chunk[-1:].expand(B - m, *chunk.shape[1:])])
            am, _, Cm = e.t_find_rotation(ref_c, chunk.contiguous(), r_lim, keep_metric)
            args.extend(am[:m])
            metrics.extend([Cm[j] for j in range(m)] if keep_metric else [None] * m)
        eulers, at_pick, beta_ids = [None] * (2 * n), np.zeros((2 * n, 3)), np.zeros(2 * n, np.int32)
        for i in range(n):
            for j in (i, n + i):
                eulers[j] = self.pick_rotation(args[j])          # the view the reference stores: a later pick of the same point edits it
                at_pick[j] = eulers[j]                           # what the reference rotates with (it rotates right after the pick)
                beta_ids[j] = args[j][0]

        def rotate(Cf):
            out = []
            for i in range(0, 2 * n, B):
                chunk, idx = Cf[i:i + B], np.arange(i, min(i + B, 2 * n))
                m = chunk.shape[0]
                if m < B:
                    chunk = t.cat([chunk, chunk[-1:].expand(B - m, *chunk.shape[1:])])
                    idx = np.concatenate([idx, np.full(B - m, idx[-1])])
                out.append(e.t_rotate_grid(chunk.contiguous(), beta_ids[idx], at_pick[idx, 0], at_pick[idx, 2])[:m])
            return t.cat(out) if len(out) > 1 else out[0]
        dens = self.isht(rotate(sig_c))
        fts = self.isht(rotate(ft_c))
        errs = self.stats(dens, reference.contiguous())[:, 5] / self.volume / norm          # integrate_normed((Re ref - Re d)^2) / norm
        res = []
        for i in range(n):
            k = i if errs[i] < errs[n + i] else n + i
            res.append({'densities': [dens[k], fts[k]], 'errors': [float(errs[k])], 'rotation_angles': [eulers[k]],
                        'rotation_metrics': [metrics[k]], 'inverted': k >= n})
        return res


def integrate_normed_weights(rs, n_theta):
    """weights w[q, t] with integrate_normed(f) = sum_qtp w[q, t] f[q, t, p] (SphericalIntegrator.integrate_normed,
    mathLibrary.py:1223-1237: Gauss weights in theta, plain sum in phi, trapezoid in r with r^2, over the ball's volume)"""
    wr, wt = hs.integrator_weights(rs, n_theta)
    return wr[:, None] * wt[None, :] / (4 / 3 * np.pi * np.max(rs) ** 3)


def _normalize(al, d, d_min=False):
    """average.py:721-727 on a device grid: (d - min) / (max - min) of the real parts"""
    st = al.stats(d[None])[0]
    if isinstance(d_min, bool):
        d_min = float(st[7])
    return al.e.t_combine('affine', d[None].contiguous(), [d_min, 1.0 / (float(st[6]) - d_min)])[0]


def average_reconstructions(engine, reconstructions, errors, opt=None, dist=None, device=None):
    """run_3d (average.py:359-570).  reconstructions: list of (real_density, reciprocal_density) of THIS rank -- numpy arrays, or
    torch tensors already on the engine's device -- errors: their selection errors.  dist: torch.distributed module of an
    initialised multi-rank job (None: single process); then the reference is the globally best reconstruction and the aligned
    sums are all-reduced.  Returns the result dict of the reference (``average``, ``resolution_metrics``, ``centered_average``,
    ``aligned`` (local ones), ...) as numpy arrays; with ``opt['keep_on_device']`` the grid-sized entries of ``aligned`` and
    ``rotation_metric`` stay torch tensors on the device (the averages are always downloaded)."""
    import torch as t
    o = dict(DEFAULTS)
    o.update(opt or {})
    al = Alignment(engine, o, device)
    e, dv = engine, al.dev
    n_in = len(reconstructions)
    errors = np.asarray(errors, dtype=float)
    if n_in:
        R = t.stack([al.to_device(r[0]) for r in reconstructions])
        F = t.stack([al.to_device(r[1]) for r in reconstructions])
    else:
        R = F = t.zeros((0,) + e.shape, dtype=t.complex128, device=dv)
    if o['center_reconstructions'] and n_in:
        R, F, _ = al.shift_to_center(R, F)
    scales = np.ones(n_in)
    if o['normalize_reconstructions']['use'] and n_in:
        fac = np.ones(n_in, complex)
        st = al.stats(R)
        if o['normalize_reconstructions']['mode'] == 'max':
            # 424-429: skipped when np.max (lexicographic on complex) has a real part <= 0; else the largest real part
            mx = st[:, 6]
            fac = np.where(mx > 0, mx, 1.0).astype(complex)
        else:
            # 432-435: the mean over the entries > 0 (numpy compares complex numbers lexicographically); complex, and only its
            # real part reaches scaling_factors
            fac = (st[:, 8] + 1j * st[:, 9]) / st[:, 10]
        scales = np.real(fac).astype(float)
        R, F = e.t_combine('scale', R.contiguous(), 1.0 / fac), e.t_combine('scale', F.contiguous(), 1.0 / fac)
    # ---- reference: the reconstruction with the lowest error (of all ranks)
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    if world > 1:
        best_local = float(errors.min()) if len(errors) else np.inf
        tb = t.tensor([best_local], dtype=t.float64, device=dv)
        allb = [t.empty_like(tb) for _ in range(world)]
        dist.all_gather(allb, tb)
        owner = int(np.argmin([float(x.item()) for x in allb]))
    else:
        owner = 0
    ref_arg = int(np.argmin(errors)) if (rank == owner and len(errors)) else -1
    keep = [i for i in range(n_in) if i != ref_arg]
    if world > 1:
        buf = t.empty((2,) + e.shape + (2,), dtype=t.float64, device=dv)
        if rank == owner:
            buf.copy_(t.view_as_real(t.stack([R[ref_arg], F[ref_arg]])))
        dist.broadcast(buf, src=owner)
        ref_pair = t.view_as_complex(buf)
        reference = [ref_pair[0].contiguous(), ref_pair[1].contiguous()]
    else:
        reference = [R[ref_arg].clone(), F[ref_arg].clone()]
    S_rho, S_F = R[keep], F[keep]
    if o.get('pointinvert_reference', False):
        ri = e.t_combine('conj', reference[1][None].contiguous())[0]
        reference = [al.ift(ri[None])[0], ri]
    # ---- align
    outs = al.apply_to(reference[0], S_rho, S_F) if len(keep) else []
    limit = o['alignment_error_limit']
    loc_err = np.array([x['errors'][-1] for x in outs])
    # average.py:519-524, literally: the list of valid alignments starts with the reference but the list of their errors does
    # not, and the argsort of the errors indexes the former: the reference is always in, the last valid alignment (in
    # processing order: rank by rank, restart by restart) never is; then the list is cut to n_reconstructions
    if world > 1:
        n_loc = t.tensor([len(loc_err)], dtype=t.int64, device=dv)
        counts = [t.empty_like(n_loc) for _ in range(world)]
        dist.all_gather(counts, n_loc)
        counts = [int(c.item()) for c in counts]
        pad = t.full((max(counts + [1]),), np.inf, dtype=t.float64, device=dv)
        pad[:len(loc_err)] = t.from_numpy(loc_err).to(dv)
        alle = [t.empty_like(pad) for _ in range(world)]
        dist.all_gather(alle, pad)
        glob = [(rk, i, float(alle[rk][i].item())) for rk in range(world) for i in range(counts[rk])]
    else:
        glob = [(0, i, float(x)) for i, x in enumerate(loc_err)]
    valid = [('ref', -1)] + [(rk, i) for rk, i, x in glob if x < limit]
    valid_err = [x for _, _, x in glob if x < limit]
    chosen = [valid[i] for i in np.argsort(valid_err)]
    n_rec = o['selection'].get('n_reconstructions', 'all')         # average.py:113-115: anything but an int means all
    if not isinstance(n_rec, int) or isinstance(n_rec, bool):
        n_rec = len(glob) + 1
    if len(chosen) >= n_rec:
        chosen = chosen[:n_rec]
    if not chosen:
        chosen = [('ref', -1)]
    aligned = []                                           # this rank's part, in the reference's order (sorted by alignment error)
    for rk, i in chosen:
        if rk == 'ref':
            if rank == owner:
                aligned.append(reference)
        elif rk == rank:
            aligned.append(outs[i]['densities'])
    # ---- sums over the selected alignments (all-reduced over the ranks from the device buffer), then the means
    sums = t.zeros((4,) + e.shape, dtype=t.complex128, device=dv)
    if aligned:
        A_rho, A_F = t.stack([a[0] for a in aligned]), t.stack([a[1] for a in aligned])
        ftd = al.ft(A_rho)
        sums[0], sums[1] = e.t_combine('sum', A_rho), e.t_combine('sum', A_F)
        sums[2] = e.t_combine('abs2sum', A_F)
        sums[3] = e.t_combine('abs2sum', ftd.contiguous())
    count = float(len(aligned))
    if world > 1:
        tr = t.view_as_real(sums)
        nn = t.tensor([count], dtype=t.float64, device=dv)
        dist.all_reduce(tr)
        dist.all_reduce(nn)
        count = float(nn.cpu()[0])
    mean4 = e.t_combine('scale', sums, np.full(4, 1.0 / count))          # the four sums over the selected alignments -> means
    average = [mean4[0], mean4[1]]
    I_ft, I_d = mean4[2], mean4[3]                                       # (real values in complex grids)
    # average.py:538: the averaged pair is centred BEFORE the metrics and the reference's shift operator multiplies its argument in
    # place (fxs_Projections.py:1442): the averaged reciprocal density that is saved and that enters 'PRTF' is the shifted one
    cen = al.shift_to_center(average[0][None], average[1][None])
    average[1] = cen[1][0]
    ft_avg = al.ft(average[0][None])[0]
    metrics = {}
    if o['resolution_metrics'].get('PRTF', False):
        a_ft, a_d, i_d, i_f = average[1].contiguous(), ft_avg.contiguous(), I_d.contiguous(), I_ft.contiguous()
        for name, args in (('PRTF', (a_d, a_ft, i_d, i_f)), ('PRTF_from_density', (a_d, a_d, i_d, i_d)),
                           ('PRTF_from_ft_density', (a_ft, a_ft, i_f, i_f)), ('PRTF_ftI', (a_d, a_d, i_f, i_f))):
            metrics[name], metrics[name + '_std'] = e.t_prtf(*args)       # b = sqrt(I) inside the kernel
    dmin = o.get('average_normalization_min', False)
    on_dev = bool(o.get('keep_on_device', False))

    def out(x):
        return x if on_dev else x.cpu().numpy()

    def host(x):
        return x.cpu().numpy()
    return {
        'average': {'real_density': host(average[0]), 'normalized_real_density': host(_normalize(al, average[0], dmin)),
                    'reciprocal_density': host(average[1]), 'intensity_from_densities': host(I_d).real.copy(),
                    'intensity_from_ft_densities': host(I_ft).real.copy()},
        'resolution_metrics': metrics,
        'centered_average': {'real_density': host(cen[0][0]), 'normalized_real_density': host(_normalize(al, cen[0][0], dmin)),
                             'reciprocal_density': host(cen[1][0])},
        'aligned': {str(i): {'real_density': out(a[0]), 'reciprocal_density': out(a[1])} for i, a in enumerate(aligned)},
        'n_averaged': int(count), 'alignment_errors': loc_err, 'reference_owner': owner, 'reference_arg': ref_arg,
        'rotation_angles': {str(i + 1): x['rotation_angles'] for i, x in enumerate(outs)},
        'rotation_metric': {str(i + 1): [out(al.metric_layout(m)) for m in x['rotation_metrics'] if m is not None] for i, x in enumerate(outs)},
        'inverted': [x['inverted'] for x in outs],
        # average.py:479, 520: 0 for the reference, then the position (in this rank's list without the reference) of every valid one
        'average_ids': [0] + [i for i, x in enumerate(loc_err) if x < limit],
        'input_meta': {'scaling_factors': scales}, 'so3_grid': al.soft_grid,
    }
