"""Checker (test infrastructure, uses the oracle): the 128 x L32 short trajectory against the oracle under the polar-factor modes --
MTIP_JAC_CONC 0 / 1, every order split, serial rotation log.  Run on the GPU box: python tests/tools/debug_conc_trajectory.py"""
import os, sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
np.seterr(all='ignore')
import parity_cases as PC
from oracle import mtip as OM
from xframe_amd.fxs import reconstruct as R, synthetic as S, hostsetup as hs
from xframe_amd.fxs.engine import Engine
cfg = 3
N, L = S._SIZES[cfg]
data, _ = PC.synthetic_problem(cfg)
opt = OM.deep_update(OM.default_settings(), S.config_overrides(cfg))
opt = OM.deep_update(opt, {'grid': {'n_radial_points': N, 'max_order': L}, 'projections': {'reciprocal': {'used_order_ids': np.arange(L + 1)}}})
loops = opt['main_loop']['sub_loops']; loops['order'] = ['main']; main = loops['main']
main['methods']['HIO']['iterations'] = 10; main['methods']['ER']['iterations'] = 10; main['iterations'] = 1
e = Engine(opt, data, n_batch=1)
rho0 = hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000), e.rsetup.integrated_intensity, e.int_wr, e.int_wt)
e.close()
ref = OM.MTIP(opt, data).phasing_loop(rho0=rho0)
for env in ({'MTIP_JAC_CONC': '0'}, {'MTIP_JAC_CONC': '1'}, {'MTIP_JAC_CONC': '1', 'MTIP_JAC_CONC_MIN_K': '2'}, {'MTIP_JAC_CONC': '1'}, {'MTIP_JAC_REPLAY': '2', 'MTIP_JAC_CONC': '0'}):
    for k in ('MTIP_JAC_CONC', 'MTIP_JAC_CONC_MIN_K', 'MTIP_JAC_REPLAY'):
        os.environ.pop(k, None)
    os.environ.update(env)
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=2, initial_densities=[rho0, rho0], fused=True)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    dev = [np.abs(r['error_dict']['main'] / ref['error_dict']['main'] - 1) for r in res]
    print(env, 'max dev vs oracle', [float(d.max()) for d in dev], 'first bad step', [int(np.argmax(d > 1e-6)) if (d > 1e-6).any() else -1 for d in dev],
          'restarts equal', bool(np.array_equal(res[0]['error_dict']['main'], res[1]['error_dict']['main'])), flush=True)
    m.engine.close()
