"""Timing of the `fxs average` step on the device: R synthetic reconstructions at N shells x L_max -- one
band-limited density, rotated by grid rotations, one copy point-inverted, scaled and noisy -- aligned against the best one and
averaged (xframe_amd/fxs/average.py: centring, FT / SHT of the batch, SO(3) correlation over the (2 L + 2)^3 Euler grid, rotation of
the coefficients, inversion test, sums, PRTF).  Prints wall times; run under rocprofv3 --kernel-trace --stats for the kernels.
usage: bench_average.py [N L R]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xframe_amd.fxs import average as AV            # noqa: E402
from xframe_amd.fxs import hostsetup as hs          # noqa: E402
from xframe_amd.fxs import synthetic as S           # noqa: E402
from xframe_amd.fxs.engine import Engine            # noqa: E402

N, L, R = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 32, 8)
max_q = float(np.max(S.midpoint_points(S.data_cutoff(N), N)))
e = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=R, max_q=max_q)
rng = np.random.default_rng(7)
al, be, ga = hs.euler_grid(L + 1)
nlm = (L + 1) ** 2
c = (rng.normal(size=(N, nlm)) + 1j * rng.normal(size=(N, nlm))) * np.exp(-(np.arange(N)[:, None] / (0.35 * N)) ** 2) / (1 + np.arange(nlm)[None, :]) ** 0.5
c[:, 1:4] = 0
base = e.sht_inverse(np.broadcast_to(c, (R, N, nlm)))[0].real
base = base - base.min() + 0.05
rs = e.rs
base = (base * np.exp(-(rs[:, None, None] / (0.5 * rs.max())) ** 4)).astype(complex)
coeff = e.sht_forward(np.broadcast_to(base, (R,) + base.shape))[0]
eulers = np.array([[al[rng.integers(len(al))], be[rng.integers(len(be))], ga[rng.integers(len(ga))]] if i != 3 else [0.0, 0.0, 0.0] for i in range(R)])
rot = e.rotate_coefficients(np.broadcast_to(coeff, (R,) + coeff.shape), eulers)
dens = e.sht_inverse(rot)
recs, errs = [], []
for i in range(R):
    d = (1.0 + 0.3 * i) * dens[i] + 1e-5 * rng.normal(size=dens[i].shape)
    recs.append(d)
    errs.append(0.01 * (1 + ((i + 2) % R)))
F = e.fourier_transform(np.stack(recs))
recs = [(recs[i], F[i]) for i in range(R)]
opt = {'alignment_error_limit': 0.5, 'find_rotation': {'r_limit_ids': [0, N]}}
import torch                                          # noqa: E402


def timed(label, inputs, o, reps=3):
    best = None
    for rep in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = AV.average_reconstructions(e, inputs, errs, o)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if rep == 0:
            print(f'{label}: first call {1e3 * dt:.1f} ms', end='')
        best = dt if best is None else min(best, dt)
    print(f', best of the next {reps - 1}: {1e3 * best:.1f} ms wall; reference {got["reference_arg"]}, '
          f'alignment errors {np.array2string(np.asarray(got["alignment_errors"]), precision=2)}')
    return got


print(f'average_reconstructions: {R} reconstructions at {N} x L{L}, Euler grid {2 * L + 2}^3')
timed('host arrays in, every array of the result on the host (incl. aligned pairs and rotation metrics)', recs, opt)
dv = e.torch_device()
recs_dev = [(torch.from_numpy(a).to(dv), torch.from_numpy(b).to(dv)) for a, b in recs]
timed('device tensors in (as a reconstruct worker on the same GPU hands them over), full result on the host', recs_dev, opt)
timed('device tensors in, aligned pairs and rotation metrics stay on the device, averages and metrics on the host', recs_dev,
      dict(opt, keep_on_device=True))
timed('the same without keeping the rotation metrics', recs_dev, dict(opt, keep_on_device=True, keep_rotation_metrics=False))
