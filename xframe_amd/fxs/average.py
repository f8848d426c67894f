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
``oracle/alignment.py`` (see its parity note); only their composition enters the result.
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
        self.soft_grid = hs.euler_grid(self.L + 1)
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

    def find_rotation(self, ref_coeff, sig_coeffs):
        """find_rotation (920-947) for a list of signals: Euler angles maximising the mean correlation + the metric arrays"""
        r_lim = self.opt['find_rotation'].get('r_limit_ids', [0, self.e.N])
        r_lim = [int(r_lim[0]), int(r_lim[1])]                      # the reference reads entries 0 and 1 (soft_plugin.py:92-94)
        al, be, ga = self.soft_grid
        Cs = self._batched(lambda c: self.e.so3_correlation(ref_coeff, c, r_lim), sig_coeffs)
        eulers = []
        for C in Cs:
            a, b, g = np.unravel_index(np.argmax(C), C.shape)
            eulers.append(np.array([al[a], be[b], ga[g]]))
        return eulers, Cs

    def rotate(self, coeffs, eulers):
        B, out = self.e.B, []
        for i in range(0, len(coeffs), B):
            cc, ee = list(coeffs[i:i + B]), list(eulers[i:i + B])
            pad = B - len(cc)
            res = self.e.rotate_coefficients(np.stack(cc + [cc[-1]] * pad), np.stack(ee + [ee[-1]] * pad))
            out.extend(res[:len(cc)])
        return out

    def align(self, reference, signals):
        """rotate_signal sketch (970-975) for a list of (density, ft_density): the rotation found on the densities is applied
        to both halves"""
        ref_c = self.e.sht_forward(np.stack([reference] * self.e.B))[0]
        sig_c = self.sht([s[0] for s in signals])
        ft_c = self.sht([s[1] for s in signals])
        eulers, Cs = self.find_rotation(ref_c, sig_c)
        dens = self.isht(self.rotate(sig_c, eulers))
        fts = self.isht(self.rotate(ft_c, eulers))
        return [[d, f] for d, f in zip(dens, fts)], eulers, Cs

    def apply_to(self, reference, signals):
        """alignment_routine (1089-1109) for a list of signals: each signal and its point inverse are aligned, the one with
        the smaller difference to the reference is kept.  Returns a list of dicts like the reference's."""
        e = self.e
        inv_d = self.ift([f.conj() for f in self.ft([s[0] for s in signals])])
        inverted = [[d, s[1].conj()] for d, s in zip(inv_d, signals)]
        norm = integrate_normed(e.rs, e.n_theta, reference.real ** 2)
        norm = norm if norm != 0 else 1
        outs = []
        for variant in (signals, inverted):
            aligned, eulers, Cs = self.align(reference, variant)
            errs = [integrate_normed(e.rs, e.n_theta, (reference.real - a[0].real) ** 2) / norm for a in aligned]
            outs.append((aligned, eulers, Cs, errs))
        res = []
        for i in range(len(signals)):
            k = 0 if outs[0][3][i] < outs[1][3][i] else 1
            res.append({'densities': outs[k][0][i], 'errors': [outs[k][3][i]], 'rotation_angles': [outs[k][1][i]],
                        'rotation_metrics': [outs[k][2][i]], 'inverted': bool(k)})
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
            if np.max(r[0]).real <= 0:
                continue
            pos = r[0][r[0] > 0]
            scales[i] = np.max(pos.real) if o['normalize_reconstructions']['mode'] == 'max' else np.mean(pos)
            recs[i] = [r[0] / scales[i], r[1] / scales[i]]
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
    n_rec = o['selection'].get('n_reconstructions', 100)
    if len(chosen) >= n_rec:
        chosen = chosen[:n_rec]
    if not chosen:
        chosen = [('ref', -1)]
    mine = [i for rk, i in chosen if rk == rank]
    use_ref = ('ref', -1) in chosen and rank == owner
    aligned = ([reference] if use_ref else []) + [outs[i]['densities'] for i in mine]
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
    ft_avg = al.ft([average[0]])[0]
    metrics = {}
    if o['resolution_metrics'].get('PRTF', False):
        for name, args in (('PRTF', (ft_avg, average[1], np.sqrt(I_d), np.sqrt(I_ft))),
                           ('PRTF_from_density', (ft_avg, ft_avg, np.sqrt(I_d), np.sqrt(I_d))),
                           ('PRTF_from_ft_density', (average[1], average[1], np.sqrt(I_ft), np.sqrt(I_ft))),
                           ('PRTF_ftI', (ft_avg, ft_avg, np.sqrt(I_ft), np.sqrt(I_ft)))):
            p = PRTF(*args)
            metrics[name], metrics[name + '_std'] = p
    cen = al.shift_to_center([average[0]], [average[1]])
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
        'input_meta': {'scaling_factors': scales}, 'so3_grid': np.stack(np.meshgrid(*al.soft_grid, indexing='ij'), -1),
    }
