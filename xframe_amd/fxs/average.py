"""``fxs average`` on the MI355X engine -- host-side mirror of ``xframe/projects/fxs/average.py`` (``ProjectWorker.run_3d``
359-627, ``Alignment`` 729-1111): centre the reconstructions, normalise them, align every one (and its point inverse) to
the reference by the SO(3) correlation of their harmonic coefficients, average the aligned ones and compute the PRTF.

Where the reference forks one process per reconstruction and runs numpy / shtns / pysofft in each, here all transforms
(FT, SHT), the SO(3) correlation and the rotation of the coefficients run on the device for a whole batch of
reconstructions per call (``csrc/k_align.hip``); centre of mass, arg-max over the Euler grid, the error integrals and the
final mean are host numpy on the downloaded grids, as cheap as the downloads themselves.  With several ranks
(``torch.distributed``: one process per GPU) every rank aligns its own restarts against the reference, which its owner
broadcasts, and the sums of the aligned densities are all-reduced (RCCL over xGMI) -- the only collective of the
``reconstruct -> average`` pipeline that moves grid-sized data.

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


def integrate_normed(rs, n_theta, values):
    """SphericalIntegrator.integrate_normed (mathLibrary.py:1223-1237)"""
    from scipy.special import roots_legendre
    w = roots_legendre(n_theta)[1]
    s2 = np.pi / n_theta * np.sum(w[None, :] * np.sum(values, axis=2), axis=1)
    f = s2 * rs ** 2
    return np.sum(np.diff(rs) * (f[1:] + f[:-1]) / 2.0) / (4 / 3 * np.pi * np.max(rs) ** 3)


def PRTF(a1, a2, b1, b2):
    """resolution_metrics.py:62-78"""
    axes = tuple(range(1, a1.ndim))
    nd = np.ones(a1.shape, dtype=complex)
    nz = (b1 != 0) & (b2 != 0)
    nd[nz] = (a1[nz] * a2[nz].conj()) / (b1[nz] * b2[nz].conj())
    nd[~nz & (a1 != 0) & (a2 != 0)] = 0
    nd = np.sqrt(nd)
    return np.average(nd, axis=axes), np.std(nd, axis=axes)


def normalize_density(d, d_min=False):
    """average.py:721-727"""
    if isinstance(d_min, bool):
        d_min = d.real.min()
    return (d - d_min) / (np.max(d.real) - d_min)


class Alignment:
    """average.py:729-1111 on a transforms engine (``Engine(settings, None, n_batch=B, max_q=...)`` of the reconstruction grid)."""

    def __init__(self, engine, opt=None):
        self.e = engine
        self.opt = dict(DEFAULTS)
        self.opt.update(opt or {})
        self.L = engine.L
        self.soft_grid = np.stack(np.meshgrid(*hs.euler_grid(self.L + 1), indexing='ij'), -1)     # make_SO3_grid; edited in place
        self.results = {}

    # -- batched transforms: lists of grids -> lists, in chunks of the engine's batch size
    def _batched(self, fn, arrays, out_shape=None):
        B, out = self.e.B, []
        for i in range(0, len(arrays), B):
            chunk = list(arrays[i:i + B])
            pad = B - len(chunk)
            res = fn(np.stack(chunk + [chunk[-1]] * pad))
            out.extend(res[:len(chunk)])
        return out

    def ft(self, grids):
        return self._batched(lambda g: self.e.fourier_transform(g), grids)

    def ift(self, grids):
        return self._batched(lambda g: self.e.fourier_transform(g, True), grids)

    def sht(self, grids):
        return self._batched(lambda g: self.e.sht_forward(g), grids)

    def isht(self, coeffs):
        return self._batched(lambda c: self.e.sht_inverse(c), coeffs)

    def shift_to_center(self, densities, ft_densities):
        """assemble_shift_to_center (1007-1020) for a list of reconstructions: (IFT(FT(rho) e^{i k c}), F e^{i k c}, c)"""
        e = self.e
        centers = [hs.calc_center(e.rs, e.theta, e.phi, d) for d in densities]
        phases = [hs.shift_phases(e.qs, e.theta, e.phi, c, opposite_direction=True) for c in centers]
        shifted = self.ift([f * p for f, p in zip(self.ft(densities), phases)])
        return shifted, [f * p for f, p in zip(ft_densities, phases)], centers

    def correlations(self, ref_coeff, sig_coeffs):
        """mean_C of find_rotation (920-935) for a list of signals, in the layout the reference reads it in: [beta, alpha, gamma],
        tabulated at the angles whose flip is the aligning rotation (oracle/alignment.py mean_C_layout)"""
        r_lim = self.opt['find_rotation'].get('r_limit_ids', [0, self.e.N])
        r_lim = [int(r_lim[0]), int(r_lim[1])]                      # the reference reads entries 0 and 1 (soft_plugin.py:92-94)
        Cs = self._batched(lambda c: self.e.so3_correlation(ref_coeff, c, r_lim), sig_coeffs)
        n = 2 * (self.L + 1)
        flip = (-np.arange(n)) % n
        return [C[flip][:, :, flip].transpose(1, 0, 2) for C in Cs]

    def pick_rotation(self, mean_C):
        """find_rotation (936-946), literally: arg-max in the reference's order, the grid entry is a VIEW and is flipped in place"""
        am = np.unravel_index(np.argmax(mean_C), mean_C.shape)
        euler = self.soft_grid[am[1], am[0], am[2]]
        euler[0] = 2 * np.pi - euler[0]
        euler[2] = 2 * np.pi - euler[2]
        return euler

    def rotate(self, coeffs, eulers):
        B, out = self.e.B, []
        for i in range(0, len(coeffs), B):
            cc, ee = list(coeffs[i:i + B]), list(eulers[i:i + B])
            pad = B - len(cc)
            res = self.e.rotate_coefficients(np.stack(cc + [cc[-1]] * pad), np.stack(ee + [ee[-1]] * pad))
            out.extend(res[:len(cc)])
        return out

    def apply_to(self, reference, signals):
        """alignment_routine (1089-1109) for a list of signals: each signal and its point inverse are aligned (rotate_signal
        sketch 970-975: the rotation found on the densities is applied to both halves), the one with the smaller difference to the
        reference is kept.  Transforms, correlations and rotations run batched over all signals and their inverses; the angles are
        picked in the reference's order (signal 0, its inverse, signal 1, ...) because picking edits the grid.  Returns a list of
        dicts like the reference's."""
        e = self.e
        inv_d = self.ift([f.conj() for f in self.ft([s[0] for s in signals])])
        both = [[s[0], s[1]] for s in signals] + [[d, s[1].conj()] for d, s in zip(inv_d, signals)]
        n = len(signals)
        norm = integrate_normed(e.rs, e.n_theta, reference.real ** 2)
        norm = norm if norm != 0 else 1
        ref_c = self.e.sht_forward(np.stack([reference] * self.e.B))[0]
        sig_c = self.sht([s[0] for s in both])
        ft_c = self.sht([s[1] for s in both])
        Cs = self.correlations(ref_c, sig_c)
        eulers, at_pick = [None] * (2 * n), [None] * (2 * n)
        for i in range(n):
            for j in (i, n + i):
                eulers[j] = self.pick_rotation(Cs[j])           # the view the reference stores: a later pick of the same point edits it
                at_pick[j] = np.array(eulers[j])                # what the reference rotates with (it rotates right after the pick)
        dens = self.isht(self.rotate(sig_c, at_pick))
        fts = self.isht(self.rotate(ft_c, at_pick))
        errs = [integrate_normed(e.rs, e.n_theta, (reference.real - d.real) ** 2) / norm for d in dens]
        res = []
        for i in range(n):
            k = i if errs[i] < errs[n + i] else n + i
            res.append({'densities': [dens[k], fts[k]], 'errors': [errs[k]], 'rotation_angles': [eulers[k]],
                        'rotation_metrics': [Cs[k]], 'inverted': k >= n})
        return res


def average_reconstructions(engine, reconstructions, errors, opt=None, dist=None, device=None):
    """run_3d (average.py:359-570).  reconstructions: list of (real_density, reciprocal_density) of THIS rank, errors: their
    selection errors.  dist: torch.distributed module of an initialised multi-rank job (None: single process); then the
    reference is the globally best reconstruction and the aligned sums are all-reduced.  Returns the result dict of the
    reference (``average``, ``resolution_metrics``, ``centered_average``, ``aligned`` (local ones), ...)."""
    o = dict(DEFAULTS)
    o.update(opt or {})
    al = Alignment(engine, o)
    e = engine
    recs = [[np.array(r[0], dtype=complex), np.array(r[1], dtype=complex)] for r in reconstructions]
    errors = np.asarray(errors, dtype=float)
    if o['center_reconstructions'] and recs:
        d, f, _ = al.shift_to_center([r[0] for r in recs], [r[1] for r in recs])
        recs = [[a, b] for a, b in zip(d, f)]
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
    # ---- reference: the reconstruction with the lowest error (of all ranks)
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    if world > 1:
        import torch
        best_local = float(errors.min()) if len(errors) else np.inf
        t = torch.tensor([best_local], dtype=torch.float64, device=device if device is not None else 'cpu')
        allb = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(allb, t)
        owner = int(np.argmin([float(x.item()) for x in allb]))
    else:
        owner = 0
    ref_arg = int(np.argmin(errors)) if (rank == owner and len(errors)) else -1
    if rank == owner:
        reference = recs.pop(ref_arg)
    if world > 1:
        import torch
        buf = torch.empty((2,) + e.shape + (2,), dtype=torch.float64, device=device if device is not None else 'cpu')
        if rank == owner:
            buf.copy_(torch.view_as_real(torch.from_numpy(np.stack(reference))))
        dist.broadcast(buf, src=owner)
        ref_arr = torch.view_as_complex(buf.cpu().contiguous()).numpy()
        reference = [ref_arr[0].copy(), ref_arr[1].copy()]
    if o.get('pointinvert_reference', False):
        ri = reference[1].conj()
        reference = [al.ift([ri])[0], ri]
    # ---- align
    outs = al.apply_to(reference[0].copy(), recs) if recs else []
    limit = o['alignment_error_limit']
    loc_err = np.array([x['errors'][-1] for x in outs])
    # average.py:519-524, literally: the list of valid alignments starts with the reference but the list of their errors does
    # not, and the argsort of the errors indexes the former: the reference is always in, the last valid alignment (in
    # processing order: rank by rank, restart by restart) never is; then the list is cut to n_reconstructions
    if world > 1:
        import torch
        n_loc = torch.tensor([len(loc_err)], dtype=torch.int64, device=device if device is not None else 'cpu')
        counts = [torch.empty_like(n_loc) for _ in range(world)]
        dist.all_gather(counts, n_loc)
        counts = [int(c.item()) for c in counts]
        pad = torch.full((max(counts + [1]),), np.inf, dtype=torch.float64, device=device if device is not None else 'cpu')
        pad[:len(loc_err)] = torch.from_numpy(loc_err)
        alle = [torch.empty_like(pad) for _ in range(world)]
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
    # ---- sums over the selected alignments (all-reduced over the ranks), then the means
    ftd = al.ft([a[0] for a in aligned]) if aligned else []
    sums = np.zeros((4,) + e.shape, complex)
    for a, fd in zip(aligned, ftd):
        sums[0] += a[0]
        sums[1] += a[1]
        sums[2] += (a[1] * a[1].conj()).real
        sums[3] += (fd * fd.conj()).real
    count = float(len(aligned))
    if world > 1:
        import torch
        t = torch.view_as_real(torch.from_numpy(sums)).contiguous()
        n = torch.tensor([count], dtype=torch.float64)
        if device is not None:
            t, n = t.to(device), n.to(device)
        dist.all_reduce(t)
        dist.all_reduce(n)
        sums = torch.view_as_complex(t.cpu().contiguous()).numpy()
        count = float(n.cpu()[0])
    average = [sums[0] / count, sums[1] / count]
    I_ft, I_d = (sums[2] / count).real, (sums[3] / count).real
    # average.py:538: the averaged pair is centred BEFORE the metrics and the reference's shift operator multiplies its argument in
    # place (fxs_Projections.py:1442): the averaged reciprocal density that is saved and that enters 'PRTF' is the shifted one
    cen = al.shift_to_center([average[0]], [average[1]])
    average[1] = cen[1][0]
    ft_avg = al.ft([average[0]])[0]
    metrics = {}
    if o['resolution_metrics'].get('PRTF', False):
        for name, args in (('PRTF', (ft_avg, average[1], np.sqrt(I_d), np.sqrt(I_ft))),
                           ('PRTF_from_density', (ft_avg, ft_avg, np.sqrt(I_d), np.sqrt(I_d))),
                           ('PRTF_from_ft_density', (average[1], average[1], np.sqrt(I_ft), np.sqrt(I_ft))),
                           ('PRTF_ftI', (ft_avg, ft_avg, np.sqrt(I_ft), np.sqrt(I_ft)))):
            p = PRTF(*args)
            metrics[name], metrics[name + '_std'] = p
    dmin = o.get('average_normalization_min', False)
    return {
        'average': {'real_density': average[0], 'normalized_real_density': normalize_density(average[0], dmin),
                    'reciprocal_density': average[1], 'intensity_from_densities': I_d, 'intensity_from_ft_densities': I_ft},
        'resolution_metrics': metrics,
        'centered_average': {'real_density': cen[0][0], 'normalized_real_density': normalize_density(cen[0][0], dmin),
                             'reciprocal_density': cen[1][0]},
        'aligned': {str(i): {'real_density': a[0], 'reciprocal_density': a[1]} for i, a in enumerate(aligned)},
        'n_averaged': int(count), 'alignment_errors': loc_err, 'reference_owner': owner, 'reference_arg': ref_arg,
        'rotation_angles': {str(i + 1): x['rotation_angles'] for i, x in enumerate(outs)},
        'inverted': [x['inverted'] for x in outs],
        # average.py:479, 520: 0 for the reference, then the position (in this rank's list without the reference) of every valid one
        'average_ids': [0] + [i for i, x in enumerate(loc_err) if x < limit],
        'input_meta': {'scaling_factors': scales}, 'so3_grid': al.soft_grid,
    }
