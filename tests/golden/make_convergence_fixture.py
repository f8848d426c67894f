"""Converged-run fixture (SURVEY section 8 d, last tolerance row): N oracle restarts of the full
tutorial schedule 5 x (60 HIO, SW, 40 ER) + (SW, 100 ER) = 600 steps at BASELINE config 3 (128 shells x L = 32), one
process per restart like the reference (reconstruct.py:141-157), reduced to rotation / inversion / translation
invariant summaries:

  final_error[i]     last value of error_dict['main']
  best_error[i]      final_error of the result dict (best pair)
  bl_err[i]          sum_l |B_l(rho_last) - B_l^data|^2 / sum_l |B_l^data|^2 on the masked shells (fxs_IO_methods.py:408-447)
  profile[i]         radial profile sqrt(<|rho|^2>_angles)(r) of the last density after the reference's own centring
                     (output_density_modifiers.shift_to_center: calc_center misk.py:295-312 + phase ramp
                     fxs_Projections.py:1419-1444)
  first_errors[i]    first 20 values of error_dict['main'] (same-seed trajectories agree tightly before chaos sets in)
  true_profile       the same profile of the centred synthetic density the invariants were made from

Written to tests/golden/convergence_cfg3_oracle.npz.  Runs in the build container only (about 2 minutes of 8 cores per
8 restarts); the GPU test (tests/test_gpu_convergence.py) reads the .npz.
usage: python tests/golden/make_convergence_fixture.py [n_restarts=8] [cfg=3] [first_restart=0]
(first_restart > 0 appends restarts first_restart .. first_restart + n - 1 to the existing file)"""
import os
import sys
import time

os.environ.setdefault('OMP_NUM_THREADS', '1')
os.environ.setdefault('OPENBLAS_NUM_THREADS', '1')
os.environ.setdefault('MKL_NUM_THREADS', '1')
import numpy as np                                                     # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
np.seterr(all='ignore')
from helpers import OracleTransforms, bl_error, radial_profile         # noqa: E402
from oracle import mtip as OM                                          # noqa: E402
from oracle import projections as OP                                   # noqa: E402
from oracle.fourier import FourierPair                                 # noqa: E402
from oracle.sht import SHT                                             # noqa: E402
from xframe_amd.fxs import synthetic as S                              # noqa: E402

SEED0 = 1000


def settings(cfg):
    opt = OM.deep_update(OM.default_settings(), S.config_overrides(cfg))
    return OM.deep_update(opt, {'output_density_modifiers': {'shift_to_center': True}})


def problem(cfg):
    N, L = S._SIZES[cfg]
    fpd = FourierPair(SHT(L), N, S.data_cutoff(N), 2.0)
    data, rho_true = S.make_invariants(OracleTransforms(fpd), N, L)
    return data, rho_true


def one_restart(args):
    cfg, i = args
    data, _ = problem(cfg)
    om = OM.MTIP(settings(cfg), data)
    rho0 = om.density_guess(np.random.default_rng(SEED0 + i))
    t0 = time.time()
    r = om.phasing_loop(rho0=rho0)
    rp = om.rp
    used = list(rp.used_orders.values())
    out = {'final_error': r['error_dict']['main'][-1], 'best_error': r['final_error'],
           'bl_err': bl_error(r['last_deg2_invariant'], rp.projection_matrices, rp.radial_mask, used, rp.number_of_particles),
           'profile': radial_profile(r['last_real_density']),
           'first_errors': np.asarray(r['error_dict']['main'][:20]), 'seconds': time.time() - t0}
    print('restart %d: %.0f s, final error %.3e, B_l err %.3e' % (i, out['seconds'], out['final_error'], out['bl_err']), flush=True)
    return out


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    cfg = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    import multiprocessing as mp
    with mp.Pool(min(n, os.cpu_count() or 1)) as pool:
        outs = pool.map(one_restart, [(cfg, i) for i in range(first, first + n)])
    data, rho_true = problem(cfg)
    om = OM.MTIP(settings(cfg), data)
    # centre the true density with the same operators the loop output uses
    grid = om.fp.grid.real_grid()
    c = OP.calc_center(om.fp.rs, len(om.sht.theta), grid, rho_true)
    ph = OP.shift_phases(om.fp.grid.reciprocal_grid(), c, True)
    true_c = om.fp.ift(om.fp.ft(rho_true) * ph)
    res = {'n_restarts': n, 'cfg': cfg, 'seed0': SEED0,
           'final_error': np.array([o['final_error'] for o in outs]), 'best_error': np.array([o['best_error'] for o in outs]),
           'bl_err': np.array([o['bl_err'] for o in outs]), 'profile': np.stack([o['profile'] for o in outs]),
           'first_errors': np.stack([o['first_errors'] for o in outs]), 'seconds': np.array([o['seconds'] for o in outs]),
           'true_profile': radial_profile(true_c), 'rs': om.fp.rs}
    path = os.path.join(HERE, 'convergence_cfg%d_oracle.npz' % cfg)
    if first > 0:
        old = dict(np.load(path))
        assert int(old['n_restarts']) == first and int(old['seed0']) == SEED0
        for k in ('final_error', 'best_error', 'bl_err', 'profile', 'first_errors', 'seconds'):
            res[k] = np.concatenate([old[k], res[k]])
        res['n_restarts'] = first + n
    np.savez_compressed(path, **res)
    print('saved', path)
