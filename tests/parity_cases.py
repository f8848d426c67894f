"""Parity checks of the HIP path (through the C ABI) against the oracle and the golden fixtures.

Each function takes ``lib_path``: ``None`` = the real libmtip_hip.so (``-m gpu`` tests on the MI355X box);
the CPU pre-flight suite passes tests/emul/libmtip_emul.so (same kernel sources compiled for the host).
Tolerances (SURVEY section 8 d / BASELINE.md section 5, fp64 end to end):
  per operator rel-L2 <= 1e-12 (Hankel, elementwise, GEMMs), <= 1e-10 (SHT; polar factor via V_l U_l),
  one full step <= 1e-9, 20-step trajectory <= 1e-6.
"""
import os

import numpy as np

from helpers import rel_l2, data_from_golden, golden_settings, OracleTransforms
from oracle import mtip as OM
from oracle import projections as OP
from oracle.fourier import FourierPair
from oracle.sht import SHT
from xframe_amd.fxs import reconstruct as R
from xframe_amd.fxs import synthetic as S
from xframe_amd.fxs.engine import Engine

TOL_OP = 1e-12
TOL_SHT = 1e-10
TOL_STEP = 1e-9
TOL_TRAJ = 1e-6


def cplx(rng, shape):
    return rng.normal(size=shape) + 1j * rng.normal(size=shape)


def transforms_engine(N, L, lib_path, n_batch=2, mode='midpoint'):
    max_q = float(np.max(S.midpoint_points(S.data_cutoff(N), N)))
    e = Engine({'grid': {'n_radial_points': N, 'max_order': L}, 'fourier_transform': {'type': mode}}, None,
               n_batch=n_batch, lib_path=lib_path, max_q=max_q)
    fp = FourierPair(SHT(L), N, max_q, 2.0, mode)
    return e, fp


def check_transforms(N, L, lib_path, seed=0, mode='midpoint', expect_chain=None):
    e, fp = transforms_engine(N, L, lib_path, mode=mode)
    sht = fp.sht
    assert (e.n_theta, e.n_phi) == (sht.n_theta, sht.n_phi)
    rng = np.random.default_rng(seed)
    g = cplx(rng, (2,) + e.shape)
    co = cplx(rng, (2, N, e.nlm))
    assert rel_l2(e.sht_forward(g), sht.forward_d(g)) < TOL_SHT
    assert rel_l2(e.sht_forward(g, 1), sht.forward_d(g * g.conj())) < TOL_SHT
    assert rel_l2(e.sht_forward(g, 2), sht.forward_d(np.abs(g))) < TOL_SHT
    assert rel_l2(e.sht_inverse(co), sht.inverse_d(co)) < TOL_SHT
    # the chained inverse -> forward kernel: the grid is that of the inverse transform, the coefficients those of the oracle's
    # two transforms one after the other
    e.profile(True)
    for pro in (0, 1):
        gi, ci = e.sht_inverse_forward(co, pro)
        ref_g = sht.inverse_d(co)
        assert rel_l2(gi, ref_g) < TOL_SHT
        assert rel_l2(ci, sht.forward_d(ref_g * ref_g.conj() if pro else ref_g)) < TOL_SHT
    if expect_chain is not None:
        assert (e.profile_get('sht_chain')[1] > 0) == expect_chain, e.profile_get('sht_chain')
    e.profile(False)
    assert rel_l2(e.hankel(co), fp.hankel(co)) < TOL_OP
    assert rel_l2(e.hankel(co, True), fp.ihankel(co)) < TOL_OP
    assert rel_l2(e.fourier_transform(g), fp.ft(g)) < TOL_SHT
    assert rel_l2(e.fourier_transform(g, True), fp.ift(g)) < TOL_SHT
    # size-independent properties: SHT round trip on band-limited data, linearity, Friedel symmetry of FT(real)
    band = e.sht_inverse(co)
    assert rel_l2(e.sht_forward(band), co) < TOL_SHT
    a, b = 0.3 - 1.1j, -2.0 + 0.5j
    lin = e.fourier_transform(a * g + b * g[::-1])
    assert rel_l2(lin, a * e.fourier_transform(g) + b * e.fourier_transform(g[::-1])) < TOL_SHT
    e.close()


def check_transforms_golden(golden_ops, lib_path):
    g = golden_ops
    N, L = 16, 4
    max_q = float(np.max(g['D16_q']))
    e = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, lib_path=lib_path, max_q=max_q)
    assert rel_l2(e.rs, g['G2_rs']) < 1e-14 and rel_l2(e.qs, g['G2_qs']) < 1e-14
    assert rel_l2(e.theta, g['G2_theta']) < 1e-14 and rel_l2(e.phi, g['G2_phi']) < 1e-14
    assert rel_l2(e.hankel(g['G2_in'])[0], g['G2_fwd']) < TOL_OP
    assert rel_l2(e.hankel(g['G2_in'], True)[0], g['G2_inv']) < TOL_OP
    assert rel_l2(e.fourier_transform(g['G2_grid_in'])[0], g['G2_ft']) < TOL_SHT
    assert rel_l2(e.fourier_transform(g['G2_grid_in'], True)[0], g['G2_ift']) < TOL_SHT
    e.close()
    e = Engine({'grid': {'n_radial_points': N, 'max_order': L}, 'fourier_transform': {'type': 'trapz'}}, None,
               n_batch=1, lib_path=lib_path, max_q=max_q)
    # golden trapz weights were assembled with the midpoint grid's r_max
    import xframe_amd.fxs.hostsetup as hs
    fs, ivs = hs.hankel_scales(float(np.max(g['G2_rs'])), N, 2.0)
    e._ck(e.lib.mtip_set_hankel_weights(e.ctx, e.raw_weights.ctypes.data, fs, ivs))
    assert rel_l2(e.hankel(g['G2_in'])[0], g['G2_trapz_fwd']) < TOL_OP
    e.close()


def check_radial_rules_golden(g, lib_path):
    """G21: the `gauss` and `Zernike` radial rules -- grids, raw weights (host setup), Hankel pair and Fourier pair of the device
    against the outputs of the reference's own functions (tests/golden/radial_rules.npz), and against the oracle at a second size."""
    import xframe_amd.fxs.hostsetup as hs
    N, L, kappa, max_q = int(g['N']), int(g['L']), float(g['kappa']), float(g['max_q'])
    for mode in ('gauss', 'Zernike'):
        e = Engine({'grid': {'n_radial_points': N, 'max_order': L}, 'fourier_transform': {'type': mode, 'reciprocity_coefficient': kappa}},
                   None, n_batch=1, lib_path=lib_path, max_q=max_q)
        assert rel_l2(e.rs, g[mode + '_rs']) < 1e-14 and rel_l2(e.qs, g[mode + '_qs']) < 1e-14
        assert rel_l2(e.raw_weights, g[mode + '_raw']) < 1e-13
        fs, ivs = hs.hankel_scales(e.r_max, N, kappa, mode)
        orders = np.arange(L + 1)
        assert rel_l2(np.moveaxis(e.raw_weights, 0, 2) * ((-1j) ** orders * fs), g[mode + '_fwd']) < 1e-13
        assert rel_l2(np.moveaxis(e.raw_weights, 0, 2) * ((1j) ** orders * ivs), g[mode + '_inv']) < 1e-13
        assert rel_l2(e.hankel(g['coeff_in'])[0], g[mode + '_hankel']) < TOL_OP
        assert rel_l2(e.hankel(g['coeff_in'], True)[0], g[mode + '_ihankel']) < TOL_OP
        assert rel_l2(e.fourier_transform(g['grid_in'])[0], g[mode + '_ft']) < TOL_SHT
        assert rel_l2(e.fourier_transform(g['grid_in'], True)[0], g[mode + '_ift']) < TOL_SHT
        e.close()
        check_transforms(10, 7, lib_path, seed=4, mode=mode)


def _engine_and_oracle(g, lib_path, prefix='data_', n_batch=1, fused=False, extra=None):
    N, L = (int(g['N']), int(g['L'])) if 'N' in g else (16, 4)
    data = data_from_golden(g, L, prefix=prefix)
    opt = golden_settings(N, L, extra)
    e = Engine(opt, data, n_batch=n_batch, lib_path=lib_path, fused=fused)
    om = OM.MTIP(opt, data)
    return e, om, opt, data


def check_operators_golden(golden_ops, lib_path):
    """G3 (projection, through V_l U_l), G4 (modulus replacement incl. zero / negative points),
    G5 (real projection + HIO + ER + error metric incl. masks), G9 (B_l)."""
    g = golden_ops
    N, L = 16, 4
    extra = {'projections': {'reciprocal': {'q_mask': {'type': 'manual', 'manual': {
        'type': 'region', 'region': [False, float(g['G2_qs'][N - 3])]}}}}}
    e, om, opt, data = _engine_and_oracle(g, lib_path, prefix='D16_', extra=extra)
    assert np.isclose(e.rsetup.integrated_intensity, g['G3_integrated_intensity'], rtol=1e-13)
    assert (e.rsetup.radial_mask == g['G3_radial_mask']).all()
    for l in range(L + 1):
        assert rel_l2(e.rsetup.projection_matrices[l], g[f'G3_pm{l}']) < 1e-12
    Ilm = np.concatenate([g[f'G3_Ilm{l}'] for l in range(L + 1)], axis=1)
    proj = e.project_coefficients(Ilm)[0]
    ref = np.concatenate([g[f'G3_proj{l}'] for l in range(L + 1)], axis=1)
    assert rel_l2(proj, ref) < TOL_SHT
    U = e.unknowns(0)
    for l in range(L + 1):
        assert rel_l2(e.rsetup.projection_matrices[l] @ U[l], g[f'G3_VU{l}']) < TOL_SHT
    assert rel_l2(e.deg2_invariants(Ilm)[0], g['G9_Bl']) < TOL_OP
    out = e.modulus_replacement(g['G4_F'], g['G4_Inew'])[0]
    fin = np.isfinite(g['G4_Fnew'])
    assert (np.isfinite(out) == fin).all()
    assert rel_l2(out[fin], g['G4_Fnew'][fin]) < TOL_OP
    e.close()
    e, om, opt, data = _engine_and_oracle(g, lib_path, prefix='D16_')
    assert (e.initial_support == g['G5_initial_support']).all()
    for enforce in (True, False):
        e.set_support(0, g['G5_support'], enforce)
        tag = f'G5_enf{int(enforce)}'
        new, err = e.real_space_update(g['G5_rho_in'], g['G5_rho_prev'], 'HIO', 0.37)
        assert rel_l2(new[0], g[tag + '_hio']) < TOL_OP
        assert np.isclose(err[0], g[tag + '_err'][0], rtol=1e-11)       # the variant the reference selects
        new, _ = e.real_space_update(g['G5_rho_in'], g['G5_rho_prev'], 'ER', 0.37)
        assert rel_l2(new[0], g[tag + '_er']) < TOL_OP
    e.close()


def check_steps_golden(golden_mtip16, lib_path, fused):
    """G10: single HIO / ER (+ft_stab) steps and one SW update from a stored state, vs the reference's sketches."""
    g = golden_mtip16
    extra = {'main_loop': {'error': {'methods': {'reciprocal': {
        'calculate': ['deg2_invariant_l2_diff'], 'deg2_invariant_l2_diff': {'order': 2}}}}}}
    e, om, opt, data = _engine_and_oracle(g, lib_path, fused=fused, extra=extra)
    for enforce in (True, False):
        for meth in ('HIO', 'ER', 'HIO_ft_stab', 'ER_ft_stab'):
            e.set_density(0, g['rho0'])
            e.init_state()
            assert rel_l2(e.density(0), g['step_rho_in']) < TOL_STEP
            e.set_support(0, g['step_support'], enforce)
            err, deg2 = e.run(meth.replace('_ft_stab', ''), meth.endswith('_ft_stab'), [0.45])
            tag = f'step_{meth}_enf{int(enforce)}'
            assert rel_l2(e.reciprocal_density(0), g[tag + '_F']) < TOL_STEP, tag
            assert rel_l2(e.density(0), g[tag + '_rho']) < TOL_STEP, tag
            assert np.isclose(err[0, 0], g[tag + '_err'], rtol=1e-8), tag
            assert np.allclose(deg2[0, 0], g[tag + '_deg2'], rtol=1e-7), tag
    e.set_density(0, g['rho0'])
    e.init_state()
    e.shrinkwrap(20.0, 0.09, np.inf)
    assert (e.support(0) != g['step_SW_mask']).sum() == 0
    e.close()


def check_trajectory_golden(g, lib_path, fused, n_restarts=1, max_steps=None):
    """G10: whole trajectory from the stored initial density vs the reference's own loop."""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L, {'main_loop': {'error': {'methods': {'reciprocal': {
        'calculate': ['deg2_invariant_l2_diff'], 'deg2_invariant_l2_diff': {'order': 2}}}}}})
    main = opt['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = int(g['n_hio'])
    main['methods']['ER']['iterations'] = int(g['n_er'])
    main['iterations'] = int(g['loop_iterations_main'])
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=n_restarts, initial_densities=[g['rho0']] * n_restarts, lib_path=lib_path, fused=fused)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    n = len(g['traj_main'])
    for b in range(n_restarts):
        r = res[b]
        assert len(r['error_dict']['main']) == n
        assert int(r['loop_iterations']) == int(g['traj_loop_iterations'])
        assert rel_l2(r['initial_density'], g['traj_initial_density']) < TOL_STEP
        k = min(20, n)
        assert np.allclose(r['error_dict']['main'][:k], g['traj_main'][:k], rtol=TOL_TRAJ)
        assert np.allclose(r['error_dict']['main'], g['traj_main'], rtol=1e-3)
        assert np.allclose(r['error_dict']['reciprocal']['deg2_invariant_l2_diff'][:k], g['traj_deg2'][:k], rtol=1e-5)
        assert rel_l2(r['last_real_density'], g['traj_last_real_density']) < 1e-4
        assert rel_l2(r['last_reciprocal_density'], g['traj_last_reciprocal_density']) < 1e-4
        assert rel_l2(r['real_density'], g['traj_real_density']) < 1e-4
        assert (r['last_support_mask'] != g['traj_last_support_mask']).mean() < 1e-3
        assert (r['support_mask'] != g['traj_support_mask']).mean() < 1e-3
        assert np.isclose(r['final_error'], g['traj_final_error'], rtol=1e-3)
        assert rel_l2(r['last_deg2_invariant'], g['traj_last_deg2_invariant']) < 1e-4
        L_ = int(g['L'])
        for l in range(L_ + 1):
            assert r['fxs_unknowns'][l].shape == g[f'traj_unk{l}'].shape
            vu = m.engine.rsetup.projection_matrices[l] @ r['fxs_unknowns'][l]
            ref = g[f'traj_VU{l}']
            assert np.linalg.norm(vu - ref) <= 1e-4 * max(np.linalg.norm(ref), 1e-30) + 1e-12
        assert r['n_particles'].shape == g['traj_n_particles'].shape
    m.engine.close()
    return res


def check_short_trajectory_vs_oracle(g, lib_path, fused, n_hio=3, n_er=2, n_restarts=2, reciprocal_opt=None):
    """A few HIO + SW + ER steps against the oracle (cheap enough for the CPU emulation).  `reciprocal_opt` overrides
    projections.reciprocal keys (odd orders kept, V_0 from the data instead of <I>, order subsets)."""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L, {'main_loop': {'error': {'methods': {'reciprocal': {
        'calculate': ['deg2_invariant_l2_diff'], 'deg2_invariant_l2_diff': {'order': 2}}}}}})
    if reciprocal_opt:
        opt['projections']['reciprocal'].update(reciprocal_opt)
    main = opt['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = n_hio
    main['methods']['ER']['iterations'] = n_er
    main['iterations'] = 2
    ref = OM.MTIP(opt, data).phasing_loop(rho0=g['rho0'])
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=n_restarts, initial_densities=[g['rho0']] * n_restarts, lib_path=lib_path, fused=fused)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    for b in range(n_restarts):
        r = res[b]
        assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-8)
        # orders whose reference invariant is rounding noise (odd l of a real density, kept by odd_orders_to_0 = False)
        # give a 0/0 metric in both implementations: compare the orders that carry signal
        n_used = len(opt['projections']['reciprocal']['used_order_ids'])
        sig = np.array([np.linalg.norm(np.asarray(data['data_projection_matrices'][l])) for l in range(min(L + 1, n_used))])
        sig = sig > 1e-9 * sig.max()
        assert np.allclose(np.asarray(r['error_dict']['reciprocal']['deg2_invariant_l2_diff'])[:, sig],
                           np.asarray(ref['error_dict']['reciprocal']['deg2_invariant_l2_diff'])[:, sig], rtol=1e-7)
        for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density',
                  'initial_density', 'last_deg2_invariant'):
            assert rel_l2(r[k], ref[k]) < 1e-8, k
        assert (r['support_mask'] != ref['support_mask']).sum() == 0
        assert (r['last_support_mask'] != ref['last_support_mask']).sum() == 0
        assert np.isclose(r['final_error'], ref['final_error'], rtol=1e-8)
        assert r['loop_iterations'] == ref['loop_iterations']
    m.engine.close()


def check_ft_stab_disagreement(g, lib_path=None):
    """3-D loop, two restarts of ONE engine that disagree on `ft_stab: link_to_enforce_initial_support` (one has its initial support
    enforced by the shrink-wrap, the other not -- the reference decides per reconstruction process, reconstruct.py:836-850): the
    engine takes ft_stab per restart (mtip_set_ft_stab_mask) and each restart follows the oracle's own run of it"""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    link = {'ft_stab': 'link_to_enforce_initial_support', 'link_to_enforce_initial_support': {'delay': 1}}
    opt = golden_settings(N, L, {'main_loop': {'sub_loops': {'main': {'iterations': 3, 'order': ['HIO', 'SW', 'ER'], 'methods': {
        'HIO': dict(iterations=3, **link), 'SW': 1, 'ER': dict(iterations=2, **link)}}}}})
    rho_a = np.asarray(g['rho0'])
    rng = np.random.default_rng(9)
    rho_b = rho_a * (1.0 + 2.0 * rng.random(rho_a.shape)) + 0.3 * np.abs(rho_a).max() * rng.random(rho_a.shape)
    refs = [OM.MTIP(opt, data).phasing_loop(rho0=r) for r in (rho_a, rho_b)]
    e3 = [r['error_dict']['main'][2] for r in refs]
    assert max(e3) > 1.2 * min(e3), e3
    eis = opt['projections']['real']['projections']['support']['enforce_initial_support']
    eis['apply'], eis['if_error_bigger_than'] = True, float(np.sqrt(e3[0] * e3[1]))
    refs = [OM.MTIP(opt, data).phasing_loop(rho0=r) for r in (rho_a, rho_b)]
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=2, initial_densities=[rho_a, rho_b], lib_path=lib_path, fused=True)
    m.generate_phasing_loop()
    seen = []
    orig = m.engine.run

    def spy(key, ft_stab, betas, **kw):
        seen.append(ft_stab)
        return orig(key, ft_stab, betas, **kw)
    m.engine.run = spy
    res = m.phasing_loop()
    m.engine.close()
    assert any(isinstance(f, np.ndarray) for f in seen)              # the restarts did disagree in some block
    for r, ref in zip(res, refs):
        assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-8)
        for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density'):
            assert rel_l2(r[k], ref[k]) < 1e-8, k
        assert (r['last_support_mask'] != ref['last_support_mask']).sum() == 0


def check_group_run_identical(g, lib_path, fused=True, sizes=(2, 1, 1)):
    """mtip_run_group_async (contexts of one GPU taking turns at the transforms of a step) against mtip_run_async per context:
    HIO + SW + ER + a non-FXS block, every density, error history and support bit-identical; the direct form of EngineGroup."""
    from xframe_amd.fxs.engine import EngineGroup
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L)
    rng = np.random.default_rng(5)
    rho0 = [g['rho0'] * (1.0 + 0.1 * rng.standard_normal(g['rho0'].shape)) for _ in range(sum(sizes))]
    out = {}
    for mode in ('group', 'single'):
        engines = [Engine(opt, data, n_batch=b, lib_path=lib_path, fused=fused) for b in sizes]
        i = 0
        for e in engines:
            for b in range(e.B):
                e.set_density(b, rho0[i])
                i += 1
            e.init_state()
        grp = EngineGroup(engines)

        def run(kind, ft_stab, betas):
            if mode == 'group':
                grp.run(kind, ft_stab, betas)
            else:
                for e in engines:
                    e.run(kind, ft_stab, betas, fetch=False)
        run('HIO', True, np.full(3, 0.45))
        run('HIO', True, np.full(1, 0.5))                        # one step per call: prologue and last turn in one
        for e in engines:
            e.shrinkwrap(e.default_sigma, 0.09, 1e9)
        run('ER', True, np.full(2, 0.5))
        run('ER', False, np.full(2, 0.5))
        run('HIO_non_FXS', True, np.full(2, 0.4))
        n = 3 + 1 + 2 + 2 + 2
        out[mode] = [(e.fetch_errors(0, n)[0], [e.density(b) for b in range(e.B)], [e.support(b) for b in range(e.B)])
                     for e in engines]
        assert grp.calls['group'] == (5 if mode == 'group' else 0)
        for e in engines:
            e.close()
    for (ea, da, sa), (eb, db, sb) in zip(out['group'], out['single']):
        assert np.array_equal(ea, eb)
        for x, y in zip(da, db):
            assert np.array_equal(x, y)
        for x, y in zip(sa, sb):
            assert np.array_equal(x, y)


def check_engine_group_rendezvous(g, lib_path):
    """EngineGroup's rendezvous form (one host thread per engine, as ProjectWorker's restart groups): same steps asked by all ->
    one group call; different steps -> each on its own; a member that leaves is not waited for; a member that never arrives is not
    waited for longer than `patience`.  Every engine ends bit-identical to an engine that ran the same calls alone."""
    import threading
    from xframe_amd.fxs.engine import EngineGroup
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L)

    def make(scale):
        e = Engine(opt, data, n_batch=1, lib_path=lib_path, fused=True)
        e.set_density(0, g['rho0'] * scale)
        e.init_state()
        return e
    plans = [[('HIO', True, [0.45, 0.45]), ('ER', True, [0.5]), ('HIO', True, [0.4])],           # engine 0
             [('HIO', True, [0.45, 0.45]), ('ER', False, [0.5]), None]]                          # engine 1: differs at call 2, leaves before call 3
    alone = []
    for i, plan in enumerate(plans):
        e = make(1.0 + 0.2 * i)
        for c in plan:
            if c is not None:
                e.run(c[0], c[1], np.array(c[2]), fetch=False)
        alone.append(e.density(0))
        e.close()
    engines = [make(1.0), make(1.2)]
    grp = EngineGroup(patience=20.0)
    for e in engines:
        grp.attach(e)
    errors = []

    def worker(e, plan):
        try:
            for c in plan:
                if c is None:
                    grp.leave(e)
                    return
                e.run(c[0], c[1], np.array(c[2]), fetch=False)
        except Exception as ex:                                      # noqa: BLE001
            errors.append(ex)
    th = [threading.Thread(target=worker, args=(e, p)) for e, p in zip(engines, plans)]
    for t in th:
        t.start()
    for t in th:
        t.join(60.0)
    assert not errors and not any(t.is_alive() for t in th), errors
    assert grp.calls == {'group': 1, 'single': 3}, grp.calls          # call 1 together; call 2 apart (2 singles); call 3 alone
    for e, want in zip(engines, alone):
        assert np.array_equal(e.density(0), want)
    # patience: engine 1 never arrives
    grp2 = EngineGroup(patience=0.3)
    for e in engines:
        e.group = None
        grp2.attach(e)
    engines[0].run('ER', True, np.array([0.5]), fetch=False)
    assert grp2.calls == {'group': 0, 'single': 1}
    for e in engines:
        e.close()


def check_group_run_identical_synthetic(cfg, lib_path=None, sizes=(2, 1, 1), n_hio=4, n_er=3):
    """check_group_run_identical at a BASELINE size on synthetic invariants (GPU: the chained kernels and k_rproj of the metric)"""
    import xframe_amd.fxs.hostsetup as hs
    from xframe_amd.fxs.engine import EngineGroup
    data, _ = synthetic_problem(cfg, lib_path)
    out = {}
    for mode in ('group', 'single'):
        engines = [Engine(S.config_overrides(cfg), data, n_batch=b, lib_path=lib_path) for b in sizes]
        i = 0
        for e in engines:
            for b in range(e.B):
                e.set_density(b, hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000 + i),
                                                 e.rsetup.integrated_intensity, e.int_wr, e.int_wt))
                i += 1
            e.init_state()
        grp = EngineGroup(engines)
        for kind, betas in (('HIO', np.full(n_hio, 0.45)), ('SW', None), ('ER', np.full(n_er, 0.5))):
            if kind == 'SW':
                for e in engines:
                    e.shrinkwrap(e.default_sigma, 0.09, 1e9)
            elif mode == 'group':
                grp.run(kind, True, betas)
            else:
                for e in engines:
                    e.run(kind, True, betas, fetch=False)
        out[mode] = [(e.fetch_errors(0, n_hio + n_er)[0], [e.density(b) for b in range(e.B)]) for e in engines]
        for e in engines:
            e.close()
    for (ea, da), (eb, db) in zip(out['group'], out['single']):
        assert np.array_equal(ea, eb) and np.isfinite(ea).all()
        for x, y in zip(da, db):
            assert np.array_equal(x, y)


def check_split_shell_steps_vs_oracle(lib_path, N=6, L=44, fused=True):
    """Angular size of config 5 (128 x 256, L > 40) with few shells: the inverse SHT shares a shell between two
    workgroups there (two error partial sums per shell in the fused real-space epilogue).  2 HIO + SW + 1 ER against
    the oracle from a seeded bump guess."""
    fpd = FourierPair(SHT(L), N, S.data_cutoff(N), 2.0)
    data, _ = S.make_invariants(OracleTransforms(fpd), N, L)
    opt = golden_settings(N, L)
    main = opt['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = 2
    main['methods']['ER']['iterations'] = 1
    main['iterations'] = 1
    om = OM.MTIP(opt, data)
    rho0 = om.density_guess(np.random.default_rng(3))
    ref = om.phasing_loop(rho0=rho0)
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=2, initial_densities=[rho0] * 2, lib_path=lib_path, fused=fused)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    for b in range(2):
        r = res[b]
        assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-8)
        for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density'):
            assert rel_l2(r[k], ref[k]) < 1e-9, k
        assert (r['last_support_mask'] != ref['last_support_mask']).sum() == 0
    m.engine.close()


SETTINGS_VARIANTS = {
    'limit_imag': {'projections': {'real': {'projections': {
        'apply': ['support', 'value_threshold', 'limit_imag'], 'limit_imag': {'threshold': 1e-3}}}}},
    'value_lo_hi': {'projections': {'real': {'projections': {
        'apply': ['support', 'value_threshold'], 'value_threshold': {'threshold': [0.01, 0.2]}}}}},
    'value_hi_only': {'projections': {'real': {'projections': {
        'apply': ['support', 'value_threshold'], 'value_threshold': {'threshold': [False, 0.15]}}}}},
    'support_only': {'projections': {'real': {'projections': {'apply': ['support']}}}},
    'no_enforce': {'projections': {'real': {'projections': {'support': {
        'enforce_initial_support': {'apply': False, 'if_error_bigger_than': 6e-3}}}}}},
    'hio_considers_support_only': {'projections': {'real': {'HIO': {'considered_projections': ['support']}}}},
    'trapz': {'fourier_transform': {'type': 'trapz'}},
    'gauss': {'fourier_transform': {'type': 'gauss'}},                  # hankel_transforms.py:477-535, ft_grid_pairs.py:293-300, 551-552
    'pi_in_q': {'fourier_transform': {'pi_in_q': True}},
    'history5': {'main_loop': {'history_length': 5}},
}


def check_settings_variant_vs_oracle(g, lib_path, name, fused=True):
    """Settings switches of the real-space projection (fxs_Projections.py:72-130, pythonLibrary.py:1289-1320), of HIO's
    considered projections (fxs_IO_methods.py:40-64), of the radial rule / reciprocity coefficient (misk.py:387-394)
    and the history length: 2 x (3 HIO, SW, 2 ER) against the oracle."""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L, SETTINGS_VARIANTS[name])
    main = opt['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = 3
    main['methods']['ER']['iterations'] = 2
    main['iterations'] = 2
    ref = OM.MTIP(opt, data).phasing_loop(rho0=g['rho0'])
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=1, initial_densities=[g['rho0']], lib_path=lib_path, fused=fused)
    m.generate_phasing_loop()
    r = m.phasing_loop()[0]
    assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-8)
    for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density'):
        assert rel_l2(r[k], ref[k]) < 1e-8, k
    assert (r['support_mask'] != ref['support_mask']).sum() == 0
    assert (r['last_support_mask'] != ref['last_support_mask']).sum() == 0
    m.engine.close()


def check_shift_to_center_vs_oracle(g, lib_path, n_restarts=2):
    """output_density_modifiers.shift_to_center (assemble_output_modifier, reconstruct.py:721-755; calc_center
    misk.py:295-312; shift_by fxs_Projections.py:1419-1444) after a short loop, against the oracle."""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L, {'output_density_modifiers': {'shift_to_center': True}})
    main = opt['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = 3
    main['methods']['ER']['iterations'] = 2
    main['iterations'] = 1
    om = OM.MTIP(opt, data)
    ref = om.phasing_loop(rho0=g['rho0'])
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=n_restarts, initial_densities=[g['rho0']] * n_restarts, lib_path=lib_path, fused=True)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    assert np.allclose(m.results['neg_center_pos'], om.results['neg_center_pos'], rtol=1e-7, atol=1e-9)
    assert om.results['neg_center_pos'][0] > 0          # a real shift
    plain = OM.MTIP(golden_settings(N, L), data)
    for b in range(n_restarts):
        r = res[b]
        for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density',
                  'last_deg2_invariant'):
            assert rel_l2(r[k], ref[k]) < 1e-8, k
    m.engine.close()
    del plain


def check_best_reselection_vs_oracle(g, lib_path, fused, n_restarts=2):
    """End-of-loop reselection of the best pair (reconstruct.py:945-949): loop 'main' ends on HIO steps with a large
    beta, so its last error is above its best one; the best density (found in iteration 3 > 1) is what loop
    'refinement' continues from."""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L)
    loops = opt['main_loop']['sub_loops']
    loops['main']['methods'] = {'ER': {'iterations': 2, 'ft_stab': True}, 'HIO': {'iterations': 2, 'ft_stab': True}}
    loops['main']['order'] = ['ER', 'HIO']
    loops['main']['iterations'] = 3
    loops['main']['best_density_not_in_first_n_iterations'] = 1
    loops['refinement']['methods'] = {'ER': {'iterations': 2, 'ft_stab': True}, 'SW': 1}
    loops['refinement']['order'] = ['SW', 'ER']
    loops['refinement']['iterations'] = 1
    loops['refinement']['best_density_not_in_first_n_iterations'] = 0
    loops['order'] = ['main', 'refinement']
    opt['projections']['real']['HIO']['beta'] = [[1.5, 1.5, -1 / 250, 500], [0.01, 0.002, -1 / 200, 200]]
    ref_m = OM.MTIP(opt, data)
    ref = ref_m.phasing_loop(rho0=g['rho0'])
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=n_restarts, initial_densities=[g['rho0']] * n_restarts, lib_path=lib_path, fused=fused)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    main_err = np.asarray(ref['error_dict']['main'])
    assert len(main_err) == 14 and np.argmin(main_err[:12]) < 11, 'case does not exercise the reselection'
    for b in range(n_restarts):
        r = res[b]
        assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-8)
        for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density'):
            assert rel_l2(r[k], ref[k]) < 1e-8, k
        assert (r['support_mask'] != ref['support_mask']).sum() == 0
        assert (r['last_support_mask'] != ref['last_support_mask']).sum() == 0
    m.engine.close()


def check_sw_center_trajectory_vs_oracle(g, lib_path, fused, n_restarts=2):
    """'SW_center' in the schedule (reconstruct.py:606-613, 886-897): HIO -> SW -> ER -> SW_center (2 repeats) ->
    HIO_non_FXS (which reads the refreshed reciprocal half of the last pair, 899-904) against the oracle."""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L)
    main = opt['main_loop']['sub_loops']['main']
    main['methods'] = {'HIO': {'iterations': 3, 'ft_stab': True}, 'SW': 1, 'ER': {'iterations': 2, 'ft_stab': True},
                       'SW_center': 2, 'HIO_non_FXS': {'iterations': 2, 'ft_stab': False}}
    main['order'] = ['HIO', 'SW', 'ER', 'SW_center', 'HIO_non_FXS']
    main['iterations'] = 2
    ref = OM.MTIP(opt, data).phasing_loop(rho0=g['rho0'])
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=n_restarts, initial_densities=[g['rho0']] * n_restarts, lib_path=lib_path, fused=fused)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    for b in range(n_restarts):
        r = res[b]
        assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-8)
        for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density'):
            assert rel_l2(r[k], ref[k]) < 1e-8, k
        assert (r['support_mask'] != ref['support_mask']).sum() == 0
        assert (r['last_support_mask'] != ref['last_support_mask']).sum() == 0
        assert r['loop_iterations'] == ref['loop_iterations']
    m.engine.close()


def check_non_fxs_trajectory_vs_oracle(g, lib_path, fused, n_restarts=2):
    """The *_non_FXS variants (reconstruct.py:899-904, sketches 530-535, 565-593): after some FXS steps the intensity
    is frozen to |F'|^2 of the latest pair and the reciprocal projection becomes F sqrt(fixed / |F|^2)
    (fxs_Projections.py:911-923).  HIO -> HIO_non_FXS -> ER_non_FXS -> ER against the oracle."""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L)
    main = opt['main_loop']['sub_loops']['main']
    main['methods'] = {'HIO': {'iterations': 3, 'ft_stab': True}, 'HIO_non_FXS': {'iterations': 2, 'ft_stab': True},
                       'ER_non_FXS': {'iterations': 2, 'ft_stab': False}, 'ER': {'iterations': 2, 'ft_stab': True}}
    main['order'] = ['HIO', 'HIO_non_FXS', 'ER_non_FXS', 'ER']
    main['iterations'] = 2
    ref = OM.MTIP(opt, data).phasing_loop(rho0=g['rho0'])
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=n_restarts, initial_densities=[g['rho0']] * n_restarts, lib_path=lib_path, fused=fused)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    for r in res:
        assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-8)
        for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density'):
            assert rel_l2(r[k], ref[k]) < 1e-8, k
        assert np.isclose(r['final_error'], ref['final_error'], rtol=1e-8)
    m.engine.close()


def check_zernike_rule_trajectories(g, lib_path, fused=True):
    """The `Zernike` radial rule (hankel_transforms.py:88-131, 270-300) inside the loop.  Its transform pair is numerically singular
    (condition of the weight matrices 1e20 .. 1e36 at 16 x L4), and with it the Procrustes matrices V_l^+ D^2 I_l of the very first
    step are rank deficient to rounding (singular-value ratios 5e-17 at l = 2, 3e-20 at l = 4, measured on this problem): their
    polar factor is not unique and LAPACK's SVD (oracle) and the Jacobi kernel complete it differently (3e-5 in V_l U_l) -- DESIGN
    section 1, intrinsic limit.  So: (1) a schedule WITHOUT the B_l projection (*_non_FXS steps with and without ft_stab: Fourier
    pair, modulus step, real-space stage, error metric, shrink wrap) at the usual 1e-8, and (2) the FXS schedule with the error trace
    to 1e-3 over its first six steps and 5e-2 over all ten (the two completions drift apart: 1e-4 after three steps, 1e-2 after nine)."""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    for fxs in (False, True):
        opt = golden_settings(N, L, {'fourier_transform': {'type': 'Zernike'}})
        main = opt['main_loop']['sub_loops']['main']
        if fxs:
            main['methods']['HIO']['iterations'] = 3
            main['methods']['ER']['iterations'] = 2
        else:
            main['methods'] = {'HIO_non_FXS': {'iterations': 3, 'ft_stab': True}, 'ER_non_FXS': {'iterations': 2, 'ft_stab': False},
                               'SW': main['methods']['SW']}
            main['order'] = ['HIO_non_FXS', 'SW', 'ER_non_FXS']
        main['iterations'] = 2
        ref = OM.MTIP(opt, data).phasing_loop(rho0=g['rho0'])
        R.MTIP.preinit(opt, data)
        m = R.MTIP(n_restarts=1, initial_densities=[g['rho0']], lib_path=lib_path, fused=fused)
        m.generate_phasing_loop()
        r = m.phasing_loop()[0]
        if fxs:
            assert np.allclose(r['error_dict']['main'][:6], ref['error_dict']['main'][:6], rtol=1e-3)
            assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=5e-2)
        else:
            assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-8)
            for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density'):
                assert rel_l2(r[k], ref[k]) < 1e-8, k
            assert (r['support_mask'] != ref['support_mask']).sum() == 0
        m.engine.close()


def synthetic_problem(cfg, lib_path=None, N=None, L=None):
    """Synthetic invariants for a BASELINE config, generated with the HIP transforms (product path)."""
    n, l = S._SIZES[cfg]
    N, L = N or n, L or l
    eng = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, lib_path=lib_path,
                 max_q=S.data_cutoff(N))
    data, rho = S.make_invariants(eng, N, L)
    eng.close()
    return data, rho


def check_projection_vs_oracle(N, L, lib_path=None, n_batch=2, seed=1):
    """mtip_projection (X_l, polar factor, V_l U_l, masks, l = 0 rules) of random coefficients against the oracle's
    numpy-SVD path, at sizes chosen to hit a particular polar-factor kernel: 2l+1 up to 89 does not fit LDS and takes the
    global-memory Jacobi fallback (config 5 has L = 48)."""
    from oracle.fourier import FourierPair
    from oracle.sht import SHT
    from helpers import OracleTransforms
    fpd = FourierPair(SHT(L), N, S.data_cutoff(N), 2.0)
    data, _ = S.make_invariants(OracleTransforms(fpd), N, L)
    opt = golden_settings(N, L)
    e = Engine(opt, data, n_batch=n_batch, lib_path=lib_path)
    om = OM.MTIP(opt, data)
    rng = np.random.default_rng(seed)
    Ilm = cplx(rng, (n_batch, N, e.nlm))
    for rep in range(2):                                   # second call: warm start from the first one's V_r
        proj = e.project_coefficients(Ilm)
        for b in range(n_batch):
            Il = [Ilm[b][:, l * l:(l + 1) ** 2] for l in range(L + 1)]
            unk = om.rp.approximate_unknowns(Il)
            ref = np.concatenate(om.rp.mtip_projection(Il, unk), axis=1)
            assert rel_l2(proj[b], ref) < TOL_SHT, (rep, b)
    e.close()


def check_prtf_golden(lib_path=None):
    """G14: the device PRTF (mtip_op_prtf, resolution_metrics.py:62-78) against values of the reference's own function -- per-shell
    complex mean and standard deviation, the general case and the single-input case"""
    import torch
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'average_ops.npz'))
    N, nt, nph = g['G14_a1'].shape
    e = Engine({'grid': {'n_radial_points': int(N), 'max_order': 1, 'n_theta': int(nt), 'n_phi': int(nph)}}, None, n_batch=1,
               lib_path=lib_path, max_q=1.0)
    dev = e.torch_device()

    def T(x):
        return torch.from_numpy(np.ascontiguousarray(x, dtype=complex)).to(dev)
    a1, a2, I1, I2 = T(g['G14_a1']), T(g['G14_a2']), T(g['G14_I1']), T(g['G14_I2'])
    p, sd = e.t_prtf(a1, a2, I1, I2)
    assert np.allclose(p, g['G14_prtf'], rtol=1e-12) and np.allclose(sd, g['G14_prtf_std'], rtol=1e-12)
    p, sd = e.t_prtf(a1, a1, I1, I1)
    assert np.allclose(p, g['G14_prtf_single'], rtol=1e-12) and np.allclose(sd, g['G14_prtf_single_std'], rtol=1e-12)
    # the zero rules (b = 0 with both a non-zero -> 0, b = 0 with an a = 0 -> 1) against the oracle's restatement
    from oracle import alignment as OA
    z = torch.zeros_like(I1)
    zero = np.zeros(g['G14_a1'].shape)
    for x1, n1 in ((a1, g['G14_a1']), (torch.zeros_like(a1), zero.astype(complex))):
        p, sd = e.t_prtf(x1, a2, z, z)
        po, so = OA.PRTF(n1, g['G14_a2'], zero, zero)
        assert np.allclose(p, po, rtol=1e-12, atol=1e-15) and np.allclose(sd, so, rtol=1e-12, atol=1e-15)
    e.close()


def check_find_rotation_nan(lib_path=None, N=6, L=4):
    """arg-max of the SO(3) correlation as numpy's (average.py:936): a NaN is the maximum, the first one in reading order wins --
    a restart whose coefficients hold a NaN comes back with index (0, 0, 0) and a NaN maximum, the others are untouched"""
    import torch
    e, _ = transforms_engine(N, L, lib_path, n_batch=2)
    rng = np.random.default_rng(11)
    dev = e.torch_device()
    ref = torch.from_numpy(cplx(rng, (N, e.nlm))).to(dev)
    sig = cplx(rng, (2, N, e.nlm))
    arg0, v0, _ = e.t_find_rotation(ref, torch.from_numpy(sig).to(dev))
    sig[1, 2, 5] = np.nan
    arg1, v1, _ = e.t_find_rotation(ref, torch.from_numpy(sig).to(dev))
    assert (arg1[0] == arg0[0]).all() and v1[0] == v0[0] and np.isfinite(v0).all()
    assert (arg1[1] == 0).all() and np.isnan(v1[1])
    e.close()


def check_polar_timing_records(lib_path=None, N=24, L=10):
    """mtip_debug_polar_timing: every solved (restart, order) has a record of its own (MTIP_POLAR_TIMING_SLOTS wide; the kernel
    clears and fills slots up to 39 -- with 32-wide records it zeroed the head of its neighbour's and wrote past the buffer)."""
    from xframe_amd.fxs import _lib
    sht = SHT(L)
    fpd = FourierPair(sht, N, S.data_cutoff(N), 2.0)
    data, _ = S.make_invariants(OracleTransforms(fpd), N, L)
    B = 2
    e = Engine(golden_settings(N, L), data, n_batch=B, lib_path=lib_path)
    rng = np.random.default_rng(3)
    grid = rng.uniform(0.0, 1.0, (B, N, sht.n_theta, sht.n_phi))
    Ilm = np.stack([np.concatenate(sht.forward_l(g.astype(complex)), axis=1) for g in grid])
    e._ck(e.lib.mtip_debug_polar_timing(e.ctx, None))
    e.project_coefficients(Ilm, real_intensity=True)
    e.project_coefficients(Ilm, real_intensity=True)
    W = 40
    out = np.zeros((B, L + 1, W), np.int64)
    e._ck(e.lib.mtip_debug_polar_timing(e.ctx, _lib.ptr(out)))
    solved = (e.jacobi_sweeps() & 0xff) > 0
    assert solved.any()
    for b in range(B):
        for l in range(L + 1):
            t = out[b, l]
            if solved[b, l] and l > 0:
                assert t[6] > 0 and t[7] >= t[6] and t[5] > 0, (b, l, t[:9])      # start / end stamps, rounds
                assert (t[:5] >= 0).all() and t[:5].sum() > 0, (b, l, t[:5])       # phase times of its own
            elif l > 0 and not solved[b, l]:
                assert not t.any(), (b, l, t)
    e.close()


def check_projection_real_vs_oracle(N, L, lib_path=None, n_batch=2, seed=1, reciprocal_opt=None, imag_residue=None, so_order=None,
                                    closing=None, expect_real=True):
    """The real-arithmetic form of the projection (k_projr.hip: real V_l, coefficients of a real intensity) against the
    oracle's complex numpy-SVD route (fxs_Projections.py:752-767, 832-871): projected coefficients, unknowns, and the
    general (complex) kernel on the same input; second call = warm start.  closing: what jacobi_closing_step() must have
    reported over the calls -- 'none' (classic confirming sweep everywhere, MTIP_RP_CORR=0) or 'some' (a closing polar step ran)."""
    from oracle.fourier import FourierPair
    from oracle.sht import SHT
    from helpers import OracleTransforms
    sht = SHT(L)
    fpd = FourierPair(sht, N, S.data_cutoff(N), 2.0)
    data, _ = S.make_invariants(OracleTransforms(fpd), N, L)
    assert all(np.all(np.asarray(p).imag == 0) for p in data['data_projection_matrices'])
    opt = golden_settings(N, L, {'projections': {'reciprocal': reciprocal_opt}} if reciprocal_opt else None)
    if imag_residue:
        # rounding residue in Im V_l, as the reference's `density` route of extract leaves it (complex-typed B_l = Il Il^+):
        # the real kernel is taken only under the opt-in MTIP_PROJ_REAL_TOL, which drops it
        rng_v = np.random.default_rng(seed + 77)
        data = dict(data)
        data['data_projection_matrices'] = [np.asarray(p) + 1j * imag_residue * np.abs(p).max() * rng_v.normal(size=np.shape(p))
                                            for p in data['data_projection_matrices']]
        os.environ['MTIP_PROJ_REAL_TOL'] = str(100 * imag_residue)
    try:
        e = Engine(opt, data, n_batch=n_batch, lib_path=lib_path)
    finally:
        os.environ.pop('MTIP_PROJ_REAL_TOL', None)
    om = OM.MTIP(opt, data)
    if so_order is not None:
        # SO_freedom on a given order (the ranking of the synthetic particle picks l = 2, whose column 2 is m = 0: real anyway)
        e._ck(e.lib.mtip_set_so_freedom(e.ctx, int(so_order)))
        om.rp.SO_order_id = int(so_order)
    rng = np.random.default_rng(seed)
    closing_seen, sweeps_seen = set(), set()
    for rep in range(3):                                   # later calls: warm start from the previous V_r
        grid = rng.uniform(0.0, 1.0, (n_batch, N, sht.n_theta, sht.n_phi)) * rng.uniform(0.5, 2.0, (n_batch, N, 1, 1))
        Ilm = np.stack([np.concatenate(sht.forward_l(g.astype(complex)), axis=1) for g in grid])
        # (the general kernel first, and only once: the real kernel's later calls then warm-start from its own V_r)
        proj_c = e.project_coefficients(Ilm) if rep == 0 else None
        proj = e.project_coefficients(Ilm, real_intensity=True)
        if expect_real:
            solved = e.jacobi_sweeps() > 0                  # (raises if an order's LDS layout did not fit its launch)
            closing_seen |= set(np.unique(e.jacobi_closing_step()[solved]).tolist())
            sweeps_seen |= set(np.unique(e.jacobi_sweeps()[solved]).tolist())
        unk_hip = [e.unknowns(b) for b in range(n_batch)]
        for b in range(n_batch):
            Il = [Ilm[b][:, l * l:(l + 1) ** 2] for l in range(L + 1)]
            unk = om.rp.approximate_unknowns(Il)
            ref = np.concatenate(om.rp.mtip_projection(Il, unk), axis=1)
            assert rel_l2(proj[b], ref) < TOL_SHT, (rep, b, rel_l2(proj[b], ref))
            assert proj_c is None or rel_l2(proj[b], proj_c[b]) < TOL_SHT, (rep, b)
            for i, l in enumerate(om.rp.used_orders.values()):
                # compared through V_l U_l (the unknowns themselves are only defined up to the null space of V_l)
                V = om.rp.projection_matrices[l]
                if np.abs(V).max() == 0:
                    continue
                assert rel_l2(V @ unk_hip[b][l], V @ unk[i]) < TOL_SHT, (rep, b, l)
    if closing == 'none':
        assert closing_seen <= {0}, closing_seen
    elif closing == 'some':
        assert closing_seen & {1, 2}, (closing_seen, sweeps_seen)
    # which kernel ran: the real one reads only the m >= 0 half of the coefficients
    junk = Ilm.copy()
    for l in range(1, L + 1):
        junk[:, :, l * l:l * l + l] = 1.0 + 2.0j
    same = rel_l2(e.project_coefficients(junk, real_intensity=True), proj) < TOL_SHT
    assert same == expect_real, 'the real-arithmetic projection was expected to run' if expect_real else 'the general kernels were expected to run'
    if so_order is not None:
        U = e.unknowns(0)[so_order]
        assert U[4, 2].imag == 0 and np.abs(U[4, 1].imag) > 1e-6       # the one element lost its imaginary part, its neighbours kept theirs
    if imag_residue:
        e2 = Engine(opt, data, n_batch=n_batch, lib_path=lib_path)      # without the opt-in: the general kernels
        assert rel_l2(e2.project_coefficients(junk, real_intensity=True), proj) > 1e-3
        assert rel_l2(e2.project_coefficients(Ilm, real_intensity=True), proj) < TOL_SHT
        e2.close()
    sw = e.jacobi_sweeps() if hasattr(e, 'jacobi_sweeps') else None
    e.close()
    return sw


def check_config_trajectory_vs_oracle(cfg, lib_path=None, fused=True, n_hio=10, n_er=10):
    """BASELINE config sizes the oracle still walks in seconds (config 2: 64 x L16): n_hio HIO + SW + n_er ER ft_stab
    steps of the product worker against the oracle's phasing loop on the same synthetic invariants and the same seeded
    initial density.  This is the size at which the wide inverse SHT (n_phi = 64), the 16-lane resident-column Jacobi
    (3 row slots) and the workgroup-tiled Hankel kernel run as they do in the benchmark."""
    import xframe_amd.fxs.hostsetup as hs
    data, _ = synthetic_problem(cfg, lib_path)
    N, L = S._SIZES[cfg]
    opt = S.config_overrides(cfg)
    opt = OM.deep_update(OM.default_settings(), opt)
    opt = OM.deep_update(opt, {'main_loop': {'error': {'methods': {'reciprocal': {
        'calculate': ['deg2_invariant_l2_diff'], 'deg2_invariant_l2_diff': {'order': 2}}}}}})
    loops = opt['main_loop']['sub_loops']
    name = loops['order'][0]
    loops['order'] = [name]
    main = loops[name]
    main['order'] = ['HIO', 'SW', 'ER'] if 'SW' in main['methods'] else ['HIO', 'ER']
    main['methods']['HIO']['iterations'] = n_hio
    main['methods']['ER']['iterations'] = n_er
    main['iterations'] = 1
    e = Engine(opt, data, n_batch=1, lib_path=lib_path)
    rho0 = hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000),
                           e.rsetup.integrated_intensity, e.int_wr, e.int_wt)
    e.close()
    ref = OM.MTIP(opt, data).phasing_loop(rho0=rho0)
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=2, initial_densities=[rho0, rho0], lib_path=lib_path, fused=fused)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    for r in res:
        assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-6)
        assert np.allclose(r['error_dict']['reciprocal']['deg2_invariant_l2_diff'],
                           ref['error_dict']['reciprocal']['deg2_invariant_l2_diff'], rtol=1e-5)
        for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density'):
            assert rel_l2(r[k], ref[k]) < TOL_TRAJ, k
        assert rel_l2(r['last_deg2_invariant'], ref['last_deg2_invariant']) < TOL_TRAJ
        assert (r['support_mask'] != ref['support_mask']).mean() < 1e-5
    m.engine.close()
    return ref['error_dict']['main']


def check_full_size_properties(cfg, lib_path=None, n_steps=12):
    """At BASELINE sizes the oracle is too slow for step-by-step comparison: check size-independent properties.
    * FT round trip of a band-limited density, SHT(iSHT(c)) == c
    * fused step == reference-order step (<= 1e-9) from the same state
    * identical restarts in one batch stay bit-identical; all errors are finite and positive; projected coefficients
      reproduce the data B_l on the masked shells
    """
    data, rho_true = synthetic_problem(cfg, lib_path)
    N, L = S._SIZES[cfg]
    opt = S.config_overrides(cfg)
    rng = np.random.default_rng(7)
    out = {}
    eng = {}
    for fused in (False, True):
        e = Engine(opt, data, n_batch=2, lib_path=lib_path, fused=fused)
        eng[fused] = e
        import xframe_amd.fxs.hostsetup as hs
        rho0 = hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000),
                               e.rsetup.integrated_intensity, e.int_wr, e.int_wt)
        for b in range(2):
            e.set_density(b, rho0)
        e.init_state()
        betas = np.full(n_steps, 0.45)
        err_h, _ = e.run('HIO', True, betas[:n_steps // 2])
        err_e, _ = e.run('ER', True, betas[:n_steps - n_steps // 2])
        out[fused] = (np.concatenate([err_h, err_e]), e.density(0), e.reciprocal_density(0), e.density(1))
    errs_a, rho_a, F_a, rho_a1 = out[False]
    errs_b, rho_b, F_b, _ = out[True]
    assert np.array_equal(rho_a, rho_a1)                       # identical restarts stay identical
    assert np.array_equal(errs_a[:, 0], errs_a[:, 1])
    assert rel_l2(rho_b, rho_a) < 1e-7 and rel_l2(F_b, F_a) < 1e-7     # fused == reference order (12 steps)
    assert np.allclose(errs_b[0], errs_a[0], rtol=TOL_STEP)             # first step tight
    assert np.isfinite(errs_a).all() and (errs_a > 0).all()
    e = eng[False]
    c = cplx(rng, (2, N, e.nlm))
    assert rel_l2(e.sht_forward(e.sht_inverse(c)), c) < TOL_SHT
    band = e.sht_inverse(c)
    # projected coefficients: B_l of the projection equals the data B_l on masked shells (U_l unitary)
    Ilm = e.sht_forward(e.fourier_transform(rho_true), 1)
    proj = e.project_coefficients(Ilm)[0]
    l = 2
    m = e.rsetup.radial_mask[l]
    V = e.rsetup.projection_matrices[l]
    Pl = proj[:, l * l:(l + 1) ** 2]
    B_proj = (Pl @ Pl.conj().T)[np.ix_(m, m)]
    B_ref = (V @ V.conj().T)[np.ix_(m, m)]
    assert rel_l2(B_proj, B_ref) < 1e-9
    for x in eng.values():
        x.close()
    return errs_a


def check_initial_density_batch(g, lib_path, kind='low_resolution_autocorrelation', seeds=(77, 78, 79)):
    """Seeded density guesses of a batch of restarts through MTIP.phasing_loop (reconstruct.py:957-979, 1115-1210): the
    autocorrelation guess runs transforms on the engine while the guesses are being staged, so every restart of the
    batch -- not only the last one -- must come out as the oracle's IFT(FT(guess)) for its own seed."""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L, {'density_guess': {'type': kind}})
    main = opt['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = 1
    main['methods']['ER']['iterations'] = 1
    main['iterations'] = 1
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=len(seeds), seeds=list(seeds), lib_path=lib_path)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    for b, seed in enumerate(seeds):
        om = OM.MTIP(opt, data)
        ref = om.phasing_loop(rho0=om.density_guess(np.random.default_rng(seed)))
        assert rel_l2(res[b]['initial_density'], ref['initial_density']) < 1e-8, (kind, b)
        assert np.allclose(res[b]['error_dict']['main'], ref['error_dict']['main'], rtol=1e-7), (kind, b)
    assert rel_l2(res[0]['initial_density'], res[1]['initial_density']) > 1e-3      # the seeds really differ
    m.engine.close()


def check_apply_unknowns(g, lib_path):
    """Registry operator mtip_projection(Ilm, unknowns) (reconstruct.py:391, fxs_Projections.py:832-849, 866-871) with
    caller-supplied unknowns: the engine's own U_l reproduce its projection, random unitary-free U_l follow the oracle's
    V_l U_l with the mask and l = 0 rules."""
    from xframe_amd.fxs.operators import build_operators
    N, L = int(g['N']), int(g['L'])
    e, om, opt, data = _engine_and_oracle(g, lib_path)
    ops = build_operators(e)
    rng = np.random.default_rng(11)
    Ilm = [cplx(rng, (N, 2 * l + 1)) for l in range(L + 1)]
    unk = ops['approximate_unknowns'](Ilm)
    a = ops['mtip_projection'](Ilm, unk)
    b = ops['mtip_projection'](Ilm, None)
    ref = om.rp.mtip_projection(Ilm, om.rp.approximate_unknowns(Ilm))
    for l in range(L + 1):
        assert rel_l2(a[l], b[l]) < 1e-13 and rel_l2(a[l], ref[l]) < TOL_SHT, l
    rand = [cplx(rng, (min(2 * l + 1, N), 2 * l + 1)) for l in range(L + 1)]
    got = ops['mtip_projection'](Ilm, rand)
    want = om.rp.mtip_projection(Ilm, rand)
    for l in range(L + 1):
        assert rel_l2(got[l], want[l]) < TOL_OP, l
    e.close()


def check_config4_worker(lib_path=None, cfg=4, n_restarts=8, n_workers=3, n_hio=10, n_er=10, oracle_restarts=(0, 5),
                         sizes=None):
    """BASELINE config 4 on one GPU as the benchmark runs it: ProjectWorker with multi_process.n_parallel_reconstructions = 8
    distinct seeds and GPU.n_gpu_workers = 3 (three engines / HIP streams driven from three host threads,
    reconstruct.py:104, 141-157), n_hio HIO + SW + n_er ER ft_stab steps.  Every restart must come out bit-identical to a
    single-engine run holding all eight (restarts never interact), and the restarts in `oracle_restarts` (one per engine
    group) must follow the oracle from the same seeded guess (20-step trajectory tolerance)."""
    import xframe_amd.fxs.hostsetup as hs
    N, L = sizes if sizes is not None else S._SIZES[cfg]
    data, _ = synthetic_problem(cfg, lib_path, N, L)
    opt = OM.deep_update(OM.default_settings(), S.config_overrides(cfg))
    opt = OM.deep_update(opt, {'grid': {'n_radial_points': N, 'max_order': L},
                               'projections': {'reciprocal': {'used_order_ids': np.arange(L + 1)}},
                               'multi_process': {'use': True, 'n_parallel_reconstructions': n_restarts}})
    loops = opt['main_loop']['sub_loops']
    loops['order'] = ['main']
    main = loops['main']
    main['methods']['HIO']['iterations'] = n_hio
    main['methods']['ER']['iterations'] = n_er
    main['iterations'] = 1
    seeds = [1000 + i for i in range(n_restarts)]
    results = {}
    for workers in (n_workers, 1):
        o = OM.deep_update(opt, {'GPU': {'use': True, 'n_gpu_workers': workers}})
        w = R.ProjectWorker(o, data, seeds=seeds, lib_path=lib_path)
        res, _ = w.run()
        assert len(res) == n_restarts and len(w.mtip_instances) == workers
        # the groups step through mtip_run_group_async (their `run` calls meet in EngineGroup): every block of the loop was
        # enqueued by group calls -- and the restarts still come out bit-identical to the single engine below
        tc = w.results['stats']['turn_calls']
        assert (tc is None) == (workers == 1) and (tc is None or (tc['group'] >= 2 and tc['single'] == 0)), tc
        groups = w.results['stats']['groups']                     # where the run spent its time, per engine group
        assert len(groups) == workers and sum(g['restarts'] for g in groups) == n_restarts
        assert all(g[k] >= 0 for g in groups for k in ('engine_seconds', 'setup_seconds', 'loop_seconds', 'output_seconds'))
        results[workers] = res
        engine = w.mtip_instances[0].engine
        if workers == n_workers:
            # the tree the reference's worker hands to its database, and the HDF5 layout of it (xframe_amd/fxs/io.py, f-2)
            from xframe_amd.fxs import io as IO
            tree = w.database_tree()
            assert sorted(tree) == ['configuration', 'projection_matrices', 'reconstruction_results', 'stats']
            ids = list(tree['reconstruction_results'])
            errs = [tree['reconstruction_results'][i]['error_dict']['main'][-1] for i in ids]
            assert errs == sorted(errs) and sorted(int(i) for i in ids) == list(range(n_restarts))
            lay = {e['path']: e for e in IO.hdf5_layout(tree)}
            assert lay['/configuration/internal_grid/real_grid']['type'] == 'NestedArray'
            assert lay[f'/reconstruction_results/{ids[0]}/real_density']['dtype'] == 'complex128'
            assert lay['/projection_matrices']['type'] in ('list', 'tuple')
            rs, shape = engine.rs, engine.shape
            ii, wr, wt = engine.rsetup.integrated_intensity, engine.int_wr, engine.int_wt
        for m in w.mtip_instances:
            m.engine.close()
    for a, b in zip(results[n_workers], results[1]):
        for k in ('initial_density', 'real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density'):
            assert np.array_equal(a[k], b[k]), k
        assert np.array_equal(a['error_dict']['main'], b['error_dict']['main'])
        assert np.array_equal(a['last_support_mask'], b['last_support_mask'])
    finals = [r['error_dict']['main'][-1] for r in results[n_workers]]
    assert len(set(finals)) == n_restarts                           # eight different reconstructions
    for i in oracle_restarts:
        rho0 = hs.bump_density(rs, shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(seeds[i]), ii, wr, wt)
        ref = OM.MTIP(opt, data).phasing_loop(rho0=rho0)
        r = results[n_workers][i]
        assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=TOL_TRAJ), i
        for k in ('initial_density', 'real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density'):
            assert rel_l2(r[k], ref[k]) < TOL_TRAJ, (i, k)
        assert (r['last_support_mask'] != ref['last_support_mask']).mean() < 1e-5
    return finals


# ---- loop variants pinned by trajectories of the reference's own MTIP class (tests/golden/make_golden.py variants) -------
def variant_settings(N, L, name):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location('make_golden', os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'make_golden.py'))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)                       # only the VARIANTS table is used (no reference import happens at import time)
    opt = golden_settings(N, L, {'main_loop': {'error': {'methods': {'reciprocal': {
        'calculate': ['deg2_invariant_l2_diff'], 'deg2_invariant_l2_diff': {'order': 2}}}}}})
    main = opt['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = 4
    main['methods']['ER']['iterations'] = 3
    main['iterations'] = 2
    return OM.deep_update(opt, mg.VARIANTS[name])


VARIANT_NAMES = ('nonfxs', 'swcenter', 'main_recip_mean', 'main_recip_max', 'main_recip_min', 'main_recip_prod', 'so_freedom', 'extra_metrics',
                 'extra_metrics_plain')


def check_variant_golden(g, v, name, lib_path=None, use_oracle=False, n_restarts=2, fused=True):
    """`name` in VARIANT_NAMES: the *_non_FXS / SW_center schedules (incl. the reference's stale `hist` and swapped
    SW_center outputs, reconstruct.py:859-913, 606-613) and the main error over the reciprocal deg2 metric
    (fxs_IO_methods.py:746-765), against trajectories recorded from the reference's own loop."""
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = variant_settings(N, L, name)
    if use_oracle:
        res = [OM.MTIP(opt, data).phasing_loop(rho0=g['rho0'])]
        tol_e, tol_d = 1e-12, 1e-12
    else:
        R.MTIP.preinit(opt, data)
        m = R.MTIP(n_restarts=n_restarts, initial_densities=[g['rho0']] * n_restarts, lib_path=lib_path, fused=fused)
        m.generate_phasing_loop()
        res = m.phasing_loop()
        m.engine.close()
        tol_e, tol_d = 1e-7, 1e-7
    for r in res:
        main_ref = v[name + '/traj_main']
        assert len(r['error_dict']['main']) == len(main_ref)
        assert np.allclose(r['error_dict']['main'], main_ref, rtol=tol_e, atol=1e-300), name
        assert np.allclose(r['error_dict']['real']['l2_projection_diff'], v[name + '/traj_real_err'], rtol=tol_e), name
        for k in ('last_real_density', 'real_density', 'last_reciprocal_density', 'reciprocal_density'):
            assert rel_l2(r[k], v[name + '/traj_' + k]) < tol_d, (name, k)
        assert (r['support_mask'] != v[name + '/traj_support_mask']).sum() == 0
        assert (r['last_support_mask'] != v[name + '/traj_last_support_mask']).sum() == 0
        assert np.isclose(r['final_error'], float(v[name + '/traj_final_error']), rtol=tol_e)
        assert int(r['loop_iterations']) == int(v[name + '/traj_loop_iterations'])
        for key in [k for k in v.files if k.startswith(name + '/traj_metric_')]:                 # every other metric the reference recorded
            cat, mname = key.split('traj_metric_')[1].split('_', 1)
            assert np.allclose(r['error_dict'][cat][mname], v[key], rtol=10 * tol_e), key


# ---- alignment + averaging of reconstructions ----------------------------------------------------
def check_average_vs_oracle(lib_path=None, N=12, L=6, n_rec=5, seed=3):
    """xframe/projects/fxs/average.py run_3d on synthetic reconstructions: one band-limited real density, rotated by
    different rotations of the Euler grid (one copy point-inverted, all with a little noise and different scales and
    centres), has to come back aligned.  The HIP path (transforms, SO(3) correlation and coefficient rotation on the
    device) against the oracle restatement: alignment errors, Euler angles, the inversion decisions, the averaged density
    and the PRTF; and the size-independent property that the aligned copies agree with the reference."""
    from oracle import alignment as OA
    from xframe_amd.fxs import average as AV
    from xframe_amd.fxs import hostsetup as hs
    max_q = float(np.max(S.midpoint_points(S.data_cutoff(N), N)))
    e = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=2, lib_path=lib_path, max_q=max_q)
    fp = FourierPair(SHT(L), N, max_q, 2.0)
    sht = fp.sht
    rng = np.random.default_rng(seed)
    al, be, ga = OA.euler_grid(L + 1)

    def make(centred):
        """a smooth real density, band limited in angle, concentrated at small radii (off centre unless `centred`: without
        l = 1 components the centre of mass is the origin), and rotated / inverted / scaled / noisy copies of it"""
        c = cplx(rng, (N, (L + 1) ** 2)) * np.exp(-(np.arange(N)[:, None] / (0.35 * N)) ** 2) / (1 + np.arange((L + 1) ** 2)[None, :]) ** 0.5
        if centred:
            c[:, 1:4] = 0
        base = sht.inverse_d(c).real
        base = base - base.min() + 0.05
        base = (base * np.exp(-(fp.rs[:, None, None] / (0.5 * fp.rs.max())) ** 4)).astype(complex)
        recs, errs = [], []
        for i in range(n_rec):
            # copy i is rotated by a grid rotation relative to copy 3, the one with the lowest error (the reference)
            euler = np.array([al[rng.integers(len(al))], be[rng.integers(len(be))], ga[rng.integers(len(ga))]]) if i != 3 else np.zeros(3)
            d = sht.inverse_d(OA.rotate_coeff(sht.forward_d(base), euler, L))
            if i == 2:
                d = fp.ift(fp.ft(d).conj())                          # point inverse
            d = (1.0 + 0.3 * i) * d + 1e-5 * rng.normal(size=d.shape)
            recs.append((d, fp.ft(d)))
            errs.append(0.01 * (1 + ((i + 2) % n_rec)))
        return recs, errs

    recs, errs = make(False)
    opt = {'alignment_error_limit': 0.5, 'find_rotation': {'r_limit_ids': [0, N]}}
    ref = OA.average_reconstructions(fp, recs, errs, opt)
    got = AV.average_reconstructions(e, recs, errs, opt)
    assert got['reference_arg'] == ref['reference_arg']
    assert np.allclose(got['alignment_errors'], ref['alignment_errors'], rtol=1e-6, atol=1e-12)
    for i in range(n_rec - 1):
        assert np.allclose(got['rotation_angles'][str(i + 1)][-1], ref['rotation_angles'][i]), i
    assert rel_l2(got['average']['real_density'], ref['average']['real_density']) < 1e-8
    assert rel_l2(got['average']['reciprocal_density'], ref['average']['reciprocal_density']) < 1e-8
    assert rel_l2(got['average']['intensity_from_densities'], ref['average']['intensity_from_densities']) < 1e-8
    assert np.allclose(got['resolution_metrics']['PRTF'], ref['resolution_metrics']['PRTF'], rtol=1e-6, atol=1e-9)
    assert np.allclose(got['input_meta']['scaling_factors'], ref['scaling_factors'])
    # property (without the centring step, whose phase-ramp shift is only approximate on the truncated spherical grid): every
    # copy comes back onto the reference -- they are exact grid rotations of one density -- with errors at the noise level
    recs, errs = make(True)
    got = AV.average_reconstructions(e, recs, errs, dict(opt, center_reconstructions=False))
    # (what is left is the 'max' normalisation: the maximum over the SAMPLES of a rotated copy differs from the reference's by
    # the interpolation error of the grid, a scale mismatch of ~1e-2, squared in the metric; the point-inverted copy
    # additionally goes through IFT(conj(FT(.))), which is only as exact as the FT round trip of this small grid)
    assert got['reference_arg'] == 3 and np.delete(got['alignment_errors'], 2).max() < 5e-3, got['alignment_errors']
    assert got['alignment_errors'][2] < 0.5
    order = [i for i in range(n_rec) if i != got['reference_arg']]
    inv_expected = [(i == 2) != (got['reference_arg'] == 2) for i in order]
    assert got['inverted'] == inv_expected
    assert got['n_averaged'] == n_rec - 1                           # the reference's selection quirk drops the last valid alignment
    e.close()
    return got


def _average_flow_sets(golden_flow):
    import json
    g = golden_flow
    N, L = int(g['G17_N']), int(g['G17_L'])
    for s in g['G17_sets']:
        pre = f'G17_{s}_'
        n = int(g[pre + 'n'])
        recs = [[g[pre + f'in{i}_real'], g[pre + f'in{i}_recip']] for i in range(n)]
        yield str(s), pre, N, L, recs, g[pre + 'selection_errors'], json.loads(str(g[pre + 'settings']))


def _compare_average_flow(res, g, pre, tol, aligned, angles, inverted, errors, scales):
    """a result of the averaging flow against what the reference's own run_3d saved / decided (fixture G17)"""
    assert int(res['reference_arg']) == int(g[pre + 'reference_arg'])
    assert np.allclose(errors, g[pre + 'alignment_errors'], rtol=1e-6, atol=1e-12), (errors, g[pre + 'alignment_errors'])
    assert list(inverted) == [bool(x) for x in g[pre + 'inverted']]
    assert np.allclose(scales, g[pre + 'scaling_factors'], rtol=1e-10)
    assert list(res['average_ids']) == [int(x) for x in g[pre + 'average_ids']]
    assert len(aligned) == int(g[pre + 'n_aligned'])
    for i, a in enumerate(aligned):
        assert rel_l2(a[0], g[pre + f'aligned{i}_real']) < tol and rel_l2(a[1], g[pre + f'aligned{i}_recip']) < tol, i
    # angles modulo 2 pi (2 pi - 0 is stored as 2 pi); compared on the circle
    d = np.asarray(angles) - g[pre + 'rotation_angles']
    assert np.abs(np.exp(1j * d) - 1).max() < 1e-9
    for k in ('real_density', 'normalized_real_density', 'reciprocal_density', 'intensity_from_densities', 'intensity_from_ft_densities'):
        assert rel_l2(res['average'][k], g[pre + 'average_' + k]) < tol, k
    for k in ('real_density', 'normalized_real_density', 'reciprocal_density'):
        assert rel_l2(res['centered_average'][k], g[pre + 'centered_' + k]) < tol, k
    for k in ('PRTF', 'PRTF_from_density', 'PRTF_from_ft_density', 'PRTF_ftI'):
        assert np.allclose(res['resolution_metrics'][k], g[pre + 'metric_' + k], rtol=1e-6, atol=1e-9), k
        assert np.allclose(res['resolution_metrics'][k + '_std'], g[pre + 'metric_' + k + '_std'], rtol=1e-6, atol=1e-8), k
    assert np.abs(np.exp(1j * (res['so3_grid'] - g[pre + 'so3_grid'])) - 1).max() < 1e-9          # the in-place flips, both sides


def check_average_flow_golden_oracle(golden_flow):
    """oracle/alignment.py against the reference's own averaging flow (fixture G17: ProjectWorker.run_3d + Alignment of the
    imported average.py on two seeded sets, pysofft doubled on the oracle's correlation / rotation)"""
    from oracle import alignment as OA
    for s, pre, N, L, recs, sel_err, o in _average_flow_sets(golden_flow):
        fp = FourierPair(SHT(L), N, float(golden_flow['G17_max_q']), 2.0)
        res = OA.average_reconstructions(fp, recs, sel_err, o)
        _compare_average_flow(res, golden_flow, pre, 1e-12, res['aligned'], res['rotation_angles'], res['inverted'],
                              res['alignment_errors'], res['scaling_factors'])


def check_average_flow_golden_hip(golden_flow, lib_path=None):
    """the product's averaging (xframe_amd/fxs/average.py on the engine's device transforms / SO(3) correlation / rotation)
    against the same fixture: every decision of the reference's flow and every array it saves"""
    from xframe_amd.fxs import average as AV
    for s, pre, N, L, recs, sel_err, o in _average_flow_sets(golden_flow):
        e = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=3, lib_path=lib_path,
                   max_q=float(golden_flow['G17_max_q']))
        res = AV.average_reconstructions(e, recs, sel_err, o)
        n = len(recs) - 1
        aligned = [[res['aligned'][str(i)]['real_density'], res['aligned'][str(i)]['reciprocal_density']] for i in range(len(res['aligned']))]
        _compare_average_flow(res, golden_flow, pre, 1e-9, aligned, [res['rotation_angles'][str(i + 1)][-1] for i in range(n)],
                              res['inverted'], res['alignment_errors'], res['input_meta']['scaling_factors'])
        e.close()


def check_metrics_golden_oracle(g):
    """oracle/metrics.py against the reference's own _generate_fqc_3d / _generate_II_3d / _generate_ccd_diff_3d (fixture G19)"""
    from oracle import metrics as M
    N, L = int(g['G19_N']), int(g['G19_L'])
    used = {l: l for l in range(L + 1)}
    rm = g['G19_radial_mask']
    inv = rm[:, :, None] * rm[:, None, :]
    fq = M.fqc_error_routine(g['G19_qs'], g['G19_ref'], used, inv, float(g['G19_wavelength']))
    ii = M.II_error_routine(g['G19_qs'], g['G19_ref'], used, inv)
    cc = M.ccd_diff_routine(g['G19_qs'], g['G19_ref'], used, float(g['G19_n_particles'][0]), inv, int(g['G19_C_order']), float(g['G19_wavelength']))
    for tag, pre in (('', 'I'), ('2', 'J')):
        Ims = [g[f'G19_{pre}{l}'] for l in range(L + 1)]
        assert np.allclose(fq(Ims), g['G19_fqc' + tag], rtol=1e-13, atol=1e-15)
        assert np.isclose(ii(Ims), g['G19_II' + tag], rtol=1e-13) and np.isclose(cc(Ims), g['G19_ccd' + tag], rtol=1e-13)


def check_invariant_metrics_vs_oracle(g, lib_path=None, fused=True):
    """II_error / ccd_diff / fqc_error of the product's loop (k_metrics.hip: per step on the device from B_l) against the oracle's
    metric routines evaluated on the oracle's own trajectory of the same steps (same data, same initial density)."""
    from oracle import metrics as M
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    data['xray_wavelength'] = 1.23984
    names = ['II_error', 'ccd_diff', 'fqc_error']
    opt = golden_settings(N, L, {'main_loop': {'error': {'methods': {'reciprocal': {'calculate': names, 'ccd_diff': {'C_order': 2}}}}}})
    main = opt['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = 3
    main['methods']['ER']['iterations'] = 2
    main['iterations'] = 1
    # oracle trajectory with a spy on the coefficients that enter the projection (what the metrics are evaluated on)
    oopt = OM.deep_update(dict(opt), {})
    oopt['main_loop']['error']['methods']['reciprocal'] = dict(oopt['main_loop']['error']['methods']['reciprocal'], calculate=[])
    om = OM.MTIP(oopt, data)
    seen = []
    orig = om.rp.approximate_unknowns

    def spy(Ilm):
        seen.append([np.array(a) for a in Ilm])
        return orig(Ilm)
    om.rp.approximate_unknowns = spy
    om.phasing_loop(rho0=g['rho0'])
    used = {l: l for l in range(L + 1)}
    rm = om.rp.radial_mask
    inv = rm[:, :, None] * rm[:, None, :]
    ref = np.array([p @ p.conj().T for p in om.rp.projection_matrices])
    fq = M.fqc_error_routine(om.rp.radial_points, ref, used, inv, 1.23984)
    ii = M.II_error_routine(om.rp.radial_points, ref, used, inv)
    cc = M.ccd_diff_routine(om.rp.radial_points, ref, used, float(om.rp.number_of_particles[0]), inv, 2, 1.23984)
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=2, initial_densities=[g['rho0']] * 2, lib_path=lib_path, fused=fused)
    m.generate_phasing_loop()
    res = m.phasing_loop()
    m.engine.close()
    n = len(seen)
    assert n == 5
    for r in res:
        e = r['error_dict']['reciprocal']
        assert e['II_error'].shape == (n,) and e['fqc_error'].shape == (n, N)
        for s_ in range(n):
            assert np.isclose(e['II_error'][s_], ii(seen[s_]).real, rtol=1e-6, atol=1e-12), s_
            assert np.isclose(e['ccd_diff'][s_], cc(seen[s_]).real, rtol=1e-6, atol=1e-12), s_
            # (with shell 0 outside the radial mask -- always, in 3-D -- fqc is 0 / 0 at q' = 0 and every row mean is NaN, upstream too)
            assert np.allclose(e['fqc_error'][s_], fq(seen[s_]), rtol=1e-6, atol=1e-9, equal_nan=True), s_


def check_invariant_metrics_golden_hip(g, lib_path=None):
    """the three metrics on the device at operator level against the reference's own routines (fixture G19): seeded invariants with a
    random mask, two sets of coefficients"""
    from xframe_amd.fxs import hostsetup as hs
    from xframe_amd.fxs import _lib
    N, L = int(g['G19_N']), int(g['G19_L'])
    e = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=2, lib_path=lib_path, max_q=1.0)
    # the tables from "projection matrices" whose V V^+ is the fixture's reference invariant: its Cholesky-like factor I_ref itself
    # is not stored, so hand the tables the invariants directly through a thin shim of the same code path
    ref = g['G19_ref']
    w, v = np.linalg.eigh(ref)
    pms = [v[l] * np.sqrt(np.clip(w[l], 0, None))[None, :] for l in range(L + 1)]
    t = hs.invariant_metric_tables(['II_error', 'ccd_diff', 'fqc_error'], g['G19_qs'], pms, g['G19_radial_mask'], float(g['G19_wavelength']),
                                   int(g['G19_C_order']))
    e.invariant_metrics = ['II_error', 'ccd_diff', 'fqc_error']
    e._ck(e.lib.mtip_set_invariant_metrics(e.ctx, 7, _lib.ptr(_lib.as_u8(t['zero_mask'])), _lib.ptr(_lib.as_c128(t['II_reference'])),
                                           _lib.ptr(_lib.as_f64(t['qq'])), _lib.ptr(_lib.as_f64(t['ccd_weights'])),
                                           _lib.ptr(_lib.as_c128(t['ccd_reference'])), float(t['ccd_norm']), _lib.ptr(_lib.as_f64(t['fqc_P'])),
                                           _lib.ptr(_lib.as_f64(t['fqc_reference_average'])), _lib.ptr(_lib.as_f64(t['fqc_reference_weights']))))
    Ilm = np.stack([np.concatenate([g[f'G19_{pre}{l}'] for l in range(L + 1)], axis=1) for pre in ('I', 'J')])
    got = e.invariant_metrics_of(Ilm)
    for b, tag in enumerate(('', '2')):
        assert np.isclose(got['II_error'][b], g['G19_II' + tag].real, rtol=1e-9), (got['II_error'][b], g['G19_II' + tag])
        assert np.isclose(got['ccd_diff'][b], g['G19_ccd' + tag].real, rtol=1e-9)
        assert np.allclose(got['fqc_error'][b], g['G19_fqc' + tag], rtol=1e-9, atol=1e-12, equal_nan=True)
    e.close()


def check_polar2d_rules_golden(g, lib_path=None, device=True):
    """the 2-D radial rules trapz / gauss / Zernike against the reference's own functions (fixture G23): grids, raw and assembled
    weights, the Hankel pair and the Fourier pair -- the oracle, the product's host tables and (device=True) the device transforms"""
    from oracle import polar2d as P2
    from xframe_amd.fxs import polar2d as X2
    N, M, kappa, max_q = int(g['N']), int(g['M']), float(g['kappa']), float(g['max_q'])
    x = g['x']
    for mode in ('trapz', 'gauss', 'Zernike'):
        fp = P2.PolarFourierPair(N, M, max_q, kappa, mode=mode)
        assert np.array_equal(fp.rs, g[mode + '_rs']) and np.array_equal(fp.qs, g[mode + '_qs'])
        assert rel_l2(fp.raw_weights, g[mode + '_raw']) < 1e-14
        assert rel_l2(fp.weights['forward'], g[mode + '_forward']) < 1e-14 and rel_l2(fp.weights['inverse'], g[mode + '_inverse']) < 1e-14
        assert rel_l2(fp.zht(x), g[mode + '_hankel_fwd']) < 1e-13 and rel_l2(fp.izht(x), g[mode + '_hankel_inv']) < 1e-13
        assert rel_l2(fp.ft(x), g[mode + '_ft']) < 1e-13 and rel_l2(fp.ift(x), g[mode + '_ift']) < 1e-13
        raw = X2.polar_raw_weights(np.arange(M + 1), N, kappa, mode)
        assert rel_l2(raw, g[mode + '_raw']) < 1e-14
        fw, iw = X2.assemble_weights_2d(raw, np.arange(M + 1), kappa * N / max_q, kappa, mode)
        skip = 1 if g[mode + '_forward'].shape[0] == N - 1 else 0
        assert rel_l2(fw[skip:], g[mode + '_forward']) < 1e-14 and rel_l2(iw[skip:], g[mode + '_inverse']) < 1e-14
        assert not skip or (np.abs(fw[0]).max() == 0 and np.abs(iw[0]).max() == 0)
        if device:
            e = X2.Engine2D(N, M, max_q, kappa, n_batch=2, lib_path=lib_path, mode=mode)
            assert np.array_equal(e.rs, g[mode + '_rs']) and np.array_equal(e.qs, g[mode + '_qs'])
            for b in range(2):
                assert rel_l2(e.hankel(x)[b], g[mode + '_hankel_fwd']) < 1e-12 and rel_l2(e.hankel(x, True)[b], g[mode + '_hankel_inv']) < 1e-12
                assert rel_l2(e.fourier_transform(x)[b], g[mode + '_ft']) < 1e-12 and rel_l2(e.fourier_transform(x, True)[b], g[mode + '_ift']) < 1e-12
            e.close()


def mtip2d_problem(g):
    """data dict and settings of the 2-D loop fixture G20"""
    N, M = int(g['N']), int(g['M'])
    data = {'dimensions': 2, 'xray_wavelength': 1.23984, 'average_intensity': g['data_aint'], 'data_radial_points': g['data_q'], 'max_order': M,
            'data_projection_matrices': g['data_pm']}
    o = OM.deep_update(OM.default_settings(), S.config_overrides(1))
    o = OM.deep_update(o, {'dimensions': 2, 'grid': {'n_radial_points': N, 'max_order': M, 'max_q': float(g['max_q'])},
                           'projections': {'reciprocal': {'used_order_ids': np.arange(M + 1)}}})
    main = o['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = int(g['n_hio'])
    main['methods']['ER']['iterations'] = int(g['n_er'])
    main['iterations'] = int(g['loop_iterations_main'])
    return data, o


def _compare_mtip2d_trajectory(res, g, tol_e, tol_d):
    assert len(res['error_dict']['main']) == len(g['traj_main'])
    assert np.allclose(res['error_dict']['main'], g['traj_main'], rtol=tol_e, atol=1e-300)
    assert np.allclose(res['error_dict']['real']['l2_projection_diff'], g['traj_real_err'], rtol=tol_e)
    for k in ('last_real_density', 'real_density', 'last_reciprocal_density', 'reciprocal_density'):
        assert rel_l2(res[k], g['traj_' + k]) < tol_d, k
    assert (res['support_mask'] != g['traj_support_mask']).sum() == 0 and (res['last_support_mask'] != g['traj_last_support_mask']).sum() == 0
    assert np.isclose(res['final_error'], float(g['traj_final_error']), rtol=tol_e)
    assert int(res['loop_iterations']) == int(g['traj_loop_iterations'])
    assert rel_l2(res['fxs_unknowns'], g['traj_unknowns']) < tol_d
    assert rel_l2(res['last_deg2_invariant'], g['traj_last_deg2_invariant']) < tol_d
    assert rel_l2(res['projection_matrices'], g['traj_projection_matrices']) < 1e-14
    assert np.array_equal(np.asarray(res['n_particles'], dtype=float), np.asarray(g['traj_n_particles'], dtype=float))
    assert np.allclose(res['grid_pair']['real_grid'], g['traj_real_grid'], rtol=1e-14, atol=1e-15)
    assert np.allclose(res['grid_pair']['reciprocal_grid'], g['traj_reciprocal_grid'], rtol=1e-14, atol=1e-15)


def check_mtip2d_golden_oracle(g):
    """oracle/mtip2d.py against the reference's own 2-D MTIP run (fixture G20): prepared fields, single steps, shrink-wrap, trajectory"""
    from oracle import mtip2d as O2
    data, o = mtip2d_problem(g)
    m = O2.MTIP2D(o, data)
    assert rel_l2(m.rp.projection_matrices, g['rp_projection_matrices']) < 1e-14 and (m.rp.radial_mask != g['rp_radial_mask']).sum() == 0
    assert np.isclose(m.rp.integrated_intensity, float(g['rp_integrated_intensity']), rtol=1e-14)
    assert (m.real_pr.initial_support != g['initial_support']).sum() == 0
    assert rel_l2(m.fp.ft(g['rho0']), g['step_F0']) < 1e-13 and rel_l2(m.fp.ift(g['step_F0']), g['step_rho_in']) < 1e-13
    for enf in (1, 0):
        for meth in ('HIO', 'ER', 'HIO_ft_stab', 'ER_ft_stab'):
            m.real_pr.enforce_initial_support = bool(enf)
            m.real_pr.support = g['step_support']
            m.beta = 0.45
            m.errors = {'real': {'l2_projection_diff': []}, 'reciprocal': {}, 'main': []}
            Fn, rn = m.step(meth.replace('_ft_stab', ''), np.array(g['step_rho_in']), meth.endswith('ft_stab'))
            tag = f'step_{meth}_enf{enf}'
            assert rel_l2(Fn, g[tag + '_F']) < 1e-12 and rel_l2(rn, g[tag + '_rho']) < 1e-12, tag
            assert np.isclose(m.errors['real']['l2_projection_diff'][-1], float(g[tag + '_err']), rtol=1e-10), tag
    m.sw.gaussian_sigma, m.sw.threshold = 20.0, 0.09
    assert (m.sw_step(np.array(g['step_rho_in'])) != g['step_SW_mask']).sum() == 0
    m = O2.MTIP2D(o, data)
    _compare_mtip2d_trajectory(m.phasing_loop(rho0=g['rho0']), g, 1e-10, 1e-10)


def check_mtip2d_golden_hip(g, lib_path=None):
    """the product's 2-D loop (xframe_amd/fxs/reconstruct2d.py on mtip2d_op_step / mtip2d_op_shrinkwrap) against the reference's
    own 2-D MTIP run (fixture G20); two restarts per call, the second one a scaled copy"""
    from xframe_amd.fxs.reconstruct2d import MTIP2D
    data, o = mtip2d_problem(g)
    m = MTIP2D(o, data, n_restarts=2, initial_densities=[g['rho0'], g['rho0']], lib_path=lib_path)
    e = m.engine
    assert rel_l2(m.rsetup.projection_matrices, g['rp_projection_matrices']) < 1e-14 and (m.rsetup.radial_mask != g['rp_radial_mask']).sum() == 0
    assert np.isclose(m.rsetup.integrated_intensity, float(g['rp_integrated_intensity']), rtol=1e-14)
    assert (m.initial_support != g['initial_support']).sum() == 0
    assert rel_l2(e.fourier_transform(g['rho0'])[0], g['step_F0']) < 1e-12 and rel_l2(e.fourier_transform(g['step_F0'], True)[1], g['step_rho_in']) < 1e-12
    for enf in (1, 0):
        sup = g['step_support'] & m.initial_support if enf else g['step_support']
        for meth in ('HIO', 'ER', 'HIO_ft_stab', 'ER_ft_stab'):
            Fn, rn, err, unk = e.step(meth.replace('_ft_stab', ''), meth.endswith('ft_stab'), 0.45, g['step_rho_in'], sup)
            tag = f'step_{meth}_enf{enf}'
            for b in range(2):
                assert rel_l2(Fn[b], g[tag + '_F']) < 1e-11 and rel_l2(rn[b], g[tag + '_rho']) < 1e-11, tag
                assert np.isclose(err[b], float(g[tag + '_err']), rtol=1e-9), tag
    assert (e.shrinkwrap(g['step_rho_in'], 20.0, 0.09)[1] != g['step_SW_mask']).sum() == 0
    res = m.phasing_loop()
    for r in res:
        _compare_mtip2d_trajectory(r, g, 1e-8, 1e-8)
    m.close()


MTIP2D_VARIANTS = ('nonfxs', 'swcenter', 'shift', 'recip_deg2', 'recip_l2', 'autocorr_support', 'so_freedom', 'so_freedom_fix', 'so_freedom_fix_hp')


def mtip2d_variant_problem(g, gv, name):
    """settings of the 2-D sub-variant `name` (tests/golden/make_golden.py VARIANTS_2D; same data and rho0 as G20) and its arrays"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import variants2d as mg
    data, o = mtip2d_problem(g)
    o = OM.deep_update(o, {k: w for k, w in mg.VARIANTS_2D[name].items() if not k.startswith('_')})
    ref = {k[len(name) + 1:]: w for k, w in gv.items() if k.startswith(name + '/')}
    return data, o, ref


def _compare_mtip2d_variant(res, ref, tol_e, tol_d):
    assert len(res['error_dict']['main']) == len(ref['traj_main'])
    assert np.allclose(res['error_dict']['main'], ref['traj_main'], rtol=tol_e, atol=1e-300)
    assert np.allclose(res['error_dict']['real']['l2_projection_diff'], ref['traj_real_err'], rtol=tol_e)
    for k in [k for k in ref if k.startswith('traj_reciprocal_') and k != 'traj_reciprocal_grid' and k != 'traj_reciprocal_density']:
        got = np.asarray(res['error_dict']['reciprocal'][k[len('traj_reciprocal_'):]])
        assert got.shape == ref[k].shape and np.allclose(got, ref[k], rtol=tol_e * 10, atol=1e-300), k
    for k in ('last_real_density', 'real_density', 'last_reciprocal_density', 'reciprocal_density', 'initial_density'):
        assert rel_l2(res[k], ref['traj_' + k]) < tol_d, k
    assert (res['support_mask'] != ref['traj_support_mask']).sum() == 0 and (res['last_support_mask'] != ref['traj_last_support_mask']).sum() == 0
    assert np.isclose(res['final_error'], float(ref['traj_final_error']), rtol=tol_e)
    assert int(res['loop_iterations']) == int(ref['traj_loop_iterations'])
    assert rel_l2(res['last_deg2_invariant'], ref['traj_last_deg2_invariant']) < tol_d


def check_mtip2d_variant_golden_oracle(g, gv, name):
    """oracle/mtip2d.py on a sub-variant of the 2-D loop against the reference's own run of it"""
    from oracle import mtip2d as O2
    data, o, ref = mtip2d_variant_problem(g, gv, name)
    m = O2.MTIP2D(o, data)
    _compare_mtip2d_variant(m.phasing_loop(rho0=g['rho0']), ref, 1e-10, 1e-10)
    if 'so_apply_in' in ref:                                           # the fix_remaining_SO_freedom operator on seeded inputs
        ap = O2.remaining_so_projection_2d(m.rp.projection_matrices, m.rp.used_orders, m.rp.radial_points, m.fp.n_phi,
                                           o['projections']['reciprocal']['SO_freedom']['radial_high_pass'])
        for c, u, want in zip(ref['so_apply_in'], ref['so_apply_unknowns'], ref['so_apply_out']):
            assert rel_l2(ap(c, u), want) < 1e-14
        if name.endswith('_hp'):
            assert any(np.abs(c - w).max() > 1 for c, w in zip(ref['so_apply_in'], ref['so_apply_out']))     # a real rotation is in the set


def check_mtip2d_variant_golden_hip(g, gv, name, lib_path=None):
    """the product's 2-D loop on a sub-variant (SW_center, *_non_FXS, reciprocal metrics, auto-correlation support, shift_to_center)
    against the reference's own run of it; two restarts per call"""
    from xframe_amd.fxs.reconstruct2d import MTIP2D
    data, o, ref = mtip2d_variant_problem(g, gv, name)
    m = MTIP2D(o, data, n_restarts=2, initial_densities=[g['rho0'], g['rho0']], lib_path=lib_path)
    res = m.phasing_loop()
    for r in res:
        _compare_mtip2d_variant(r, ref, 1e-8, 1e-8)
    if 'so_apply_in' in ref:
        from xframe_amd.fxs.reconstruct2d import RemainingRotation2D
        rot = RemainingRotation2D(m.rsetup.projection_matrices, m.rsetup.used_orders, m.engine.qs, m.engine.n_phi, m.rsetup.radial_high_pass)
        for c, u, want in zip(ref['so_apply_in'], ref['so_apply_unknowns'], ref['so_apply_out']):
            assert rel_l2(rot(c, u), want) < 1e-14
    m.close()


def check_mtip2d_unbuildable_variants(g, gv, lib_path=None):
    """what the reference cannot run in 2-D raises here too: the low-resolution auto-correlation guess (recorded upstream exception)"""
    from xframe_amd.fxs.reconstruct2d import MTIP2D
    import pytest
    assert 'TypeError' in str(gv['autocorr_guess/raises'])
    data, o = mtip2d_problem(g)
    o = OM.deep_update(o, {'density_guess': {'type': 'low_resolution_autocorrelation'}})
    m = MTIP2D(o, data, n_restarts=1, lib_path=lib_path)
    with pytest.raises(NotImplementedError):
        m.phasing_loop()
    m.close()


def check_mtip2d_worker_vs_oracle(g, lib_path=None, N=None, M=None, n_restarts=3):
    """`fxs reconstruct` with `dimensions: 2` through ProjectWorker: seeded density guesses (bump), each restart against the oracle's
    loop run from the same generator; sizes of the fixture unless N, M are given (then data interpolated from the fixture's)"""
    from oracle import mtip2d as O2
    data, o = mtip2d_scaled_problem(g, N, M)
    o = OM.deep_update(o, {'multi_process': {'use': True, 'n_parallel_reconstructions': n_restarts}, 'GPU': {'use': True, 'n_gpu_workers': 1}})
    seeds = [77 + i for i in range(n_restarts)]
    w = R.ProjectWorker(o, data, seeds=seeds, lib_path=lib_path)
    res, _ = w.run()
    assert len(res) == n_restarts and set(w.results['reconstruction_results']) == {str(i) for i in range(n_restarts)}
    for b in range(n_restarts):
        ref = O2.MTIP2D(o, data).phasing_loop(rng=np.random.default_rng(seeds[b]))
        assert rel_l2(res[b]['initial_density'], ref['initial_density']) < 1e-12
        assert np.allclose(res[b]['error_dict']['main'], ref['error_dict']['main'], rtol=1e-7)
        for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density', 'last_deg2_invariant', 'fxs_unknowns'):
            assert rel_l2(res[b][k], ref[k]) < 1e-7, k
        assert (res[b]['support_mask'] != ref['support_mask']).sum() == 0 and (res[b]['last_support_mask'] != ref['last_support_mask']).sum() == 0
    for m in w.mtip_instances:
        m.engine.close()


def check_mtip2d_ft_stab_disagreement(g, lib_path=None):
    """two restarts of one batch that DISAGREE on the ft_stab link (one has its initial support enforced by the shrink-wrap, the other
    not: the reference decides per reconstruction process, reconstruct.py:836-850): every restart against the oracle's own run of it"""
    from oracle import mtip2d as O2
    from xframe_amd.fxs.reconstruct2d import MTIP2D
    data, o = mtip2d_problem(g)
    o = OM.deep_update(o, {'main_loop': {'sub_loops': {'main': {'iterations': 3, 'order': ['HIO', 'SW', 'ER'], 'methods': {
        'HIO': {'iterations': 3, 'ft_stab': 'link_to_enforce_initial_support', 'link_to_enforce_initial_support': {'delay': 1}},
        'SW': 1, 'ER': {'iterations': 2, 'ft_stab': 'link_to_enforce_initial_support', 'link_to_enforce_initial_support': {'delay': 1}}}}}}})
    rho_a = np.asarray(g['rho0'])
    rho_b = rho_a * (1.0 + 2.0 * np.random.default_rng(9).random(rho_a.shape)) + 0.3 * np.abs(rho_a).max() * np.random.default_rng(10).random(rho_a.shape)
    refs = [O2.MTIP2D(o, data).phasing_loop(rho0=r) for r in (rho_a, rho_b)]
    e3 = [r['error_dict']['main'][2] for r in refs]
    assert max(e3) > 1.3 * min(e3)
    eis = o['projections']['real']['projections']['support']['enforce_initial_support']
    eis['apply'], eis['if_error_bigger_than'] = True, float(np.sqrt(e3[0] * e3[1]))
    refs = [O2.MTIP2D(o, data).phasing_loop(rho0=r) for r in (rho_a, rho_b)]
    m = MTIP2D(o, data, n_restarts=2, initial_densities=[rho_a, rho_b], lib_path=lib_path)
    seen = []
    orig = m._step

    def spy(key, ft_stab, *a):
        seen.append(ft_stab)
        return orig(key, ft_stab, *a)
    m._step = spy
    res = m.phasing_loop()
    m.close()
    assert any(isinstance(f, np.ndarray) for f in seen)              # the restarts did disagree in some block
    for r, ref in zip(res, refs):
        assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-8)
        for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density', 'fxs_unknowns'):
            assert rel_l2(r[k], ref[k]) < 1e-8, k
        assert (r['last_support_mask'] != ref['last_support_mask']).sum() == 0


SETTINGS_VARIANTS_2D = dict(
    {k: SETTINGS_VARIANTS[k] for k in ('limit_imag', 'value_lo_hi', 'value_hi_only', 'support_only', 'no_enforce', 'hio_considers_support_only',
                                       'pi_in_q', 'history5')},
    n_particles={'projections': {'reciprocal': {'number_of_particles': {'initial': 7}}}},
    odd_orders_to_0={'projections': {'reciprocal': {'odd_orders_to_0': True}}},
    q_mask_region={'projections': {'reciprocal': {'q_mask': {'type': 'manual', 'manual': {'type': 'region', 'region': [0.02, 0.09]}}}}},
    error_inside_support={'general': {'cache_aware': False},
                          'main_loop': {'error': {'methods': {'real': {'l2_projection_diff': {'inside_initial_support': True}}}}}},
    ft_stab_linked={'main_loop': {'sub_loops': {'main': {'methods': {
        'HIO': {'iterations': 3, 'ft_stab': 'link_to_enforce_initial_support', 'link_to_enforce_initial_support': {'delay': 1}},
        'ER': {'iterations': 2, 'ft_stab': False}}}}}},
    er_only={'main_loop': {'sub_loops': {'main': {'order': ['ER'], 'methods': {'ER': {'iterations': 4, 'ft_stab': True}}}}}},
    rule_trapz={'fourier_transform': {'type': 'trapz'}}, rule_gauss={'fourier_transform': {'type': 'gauss'}},
    best_reselected={'main_loop': {'sub_loops': {'main': {'best_density_not_in_first_n_iterations': 0}}}})


def check_mtip2d_settings_vs_oracle(g, lib_path, name):
    """settings switches of the 2-D loop -- the real-space projections and HIO's considered ones, the reciprocity coefficient, history
    length, number of particles, odd orders, q mask, the masked error metric, the ft_stab link, an ER-only loop, the end-of-loop
    reselection of the best density -- product against oracle/mtip2d.py (itself at 0.0 from the reference's run, G20), two restarts"""
    from oracle import mtip2d as O2
    from xframe_amd.fxs.reconstruct2d import MTIP2D
    data, o = mtip2d_problem(g)
    main = o['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = 3
    main['methods']['ER']['iterations'] = 2
    main['iterations'] = 2
    o = OM.deep_update(o, SETTINGS_VARIANTS_2D[name])
    rho0 = np.asarray(g['rho0'])
    ref = O2.MTIP2D(o, data).phasing_loop(rho0=rho0)
    m = MTIP2D(o, data, n_restarts=2, initial_densities=[rho0, 1.5 * rho0], lib_path=lib_path)
    r = m.phasing_loop()[0]
    m.close()
    assert len(r['error_dict']['main']) == len(ref['error_dict']['main'])
    assert np.allclose(r['error_dict']['main'], ref['error_dict']['main'], rtol=1e-8), name
    for k in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density', 'fxs_unknowns', 'last_deg2_invariant'):
        assert rel_l2(r[k], ref[k]) < 1e-8, (name, k)
    assert (r['support_mask'] != ref['support_mask']).sum() == 0 and (r['last_support_mask'] != ref['last_support_mask']).sum() == 0
    assert np.isclose(r['final_error'], ref['final_error'], rtol=1e-8)


def mtip2d_scaled_problem(g, N=None, M=None):
    """the 2-D problem of fixture G20, or a larger one of the same kind: its invariants interpolated to N shells, the orders beyond the
    fixture's repeated with falling weight"""
    data, o = mtip2d_problem(g)
    if N is not None:
        qd = np.linspace(data['data_radial_points'][0], data['data_radial_points'][-1], N)
        m0 = int(g['M'])
        pm = np.zeros((M + 1, N), complex)
        for i in range(M + 1):
            src = data['data_projection_matrices'][min(i, m0)]
            pm[i] = (np.interp(qd, data['data_radial_points'], src.real) + 1j * np.interp(qd, data['data_radial_points'], src.imag)) / (1 + max(i - m0, 0))
        data = dict(data, data_radial_points=qd, data_projection_matrices=pm, max_order=M,
                    average_intensity=np.interp(qd, data['data_radial_points'], np.asarray(data['average_intensity'], dtype=float)))
        o = OM.deep_update(o, {'grid': {'n_radial_points': N, 'max_order': M}, 'projections': {'reciprocal': {'used_order_ids': np.arange(M + 1)}}})
    return data, o


def check_polar2d_golden_oracle(g):
    """oracle/polar2d.py against the reference's own 2-D functions (fixture G18)"""
    from oracle import polar2d as P2
    N, M, kappa, max_q = int(g['G18_N']), int(g['G18_M']), float(g['G18_kappa']), float(g['G18_max_q'])
    x = g['G18_x']
    assert rel_l2(P2.harmonic_forward(x), g['G18_cht_fwd']) < 1e-14 and rel_l2(P2.harmonic_inverse(x), g['G18_cht_inv']) < 1e-14
    assert rel_l2(P2.real_harmonic_forward(x), g['G18_rht_fwd']) < 1e-14
    assert rel_l2(P2.real_harmonic_inverse(g['G18_xr'], 2 * M + 1), g['G18_rht_inv']) < 1e-14
    fp = P2.PolarFourierPair(N, M, max_q, kappa)
    assert fp.n_phi == int(g['G18_ht_n_phi'])
    assert np.allclose(fp.rs, g['G18_rs'], rtol=1e-14) and np.allclose(fp.qs, g['G18_qs'], rtol=1e-14) and np.allclose(fp.phis, g['G18_phis'], atol=1e-15)
    assert rel_l2(fp.raw_weights, g['G18_weights_raw']) < 1e-14
    assert rel_l2(fp.weights['forward'], g['G18_weights_forward']) < 1e-14 and rel_l2(fp.weights['inverse'], g['G18_weights_inverse']) < 1e-14
    assert rel_l2(fp.zht(x), g['G18_hankel_fwd_all']) < 1e-14 and rel_l2(fp.izht(x), g['G18_hankel_inv_all']) < 1e-14
    assert rel_l2(fp.ft(x), g['G18_ft_all']) < 1e-13 and rel_l2(fp.ift(x), g['G18_ift_all']) < 1e-13
    used = g['G18_used_sub']
    rp = P2.ReciprocalProjection2D(g['G18_proj_pm'], {int(o): int(o) for o in used}, g['G18_proj_mask'], g['G18_qs'], M + 1,
                                   float(g['G18_proj_n_particles']))
    u = rp.approximate_unknowns(g['G18_proj_I'])
    assert rel_l2(u, g['G18_proj_unknowns']) < 1e-14
    assert rel_l2(rp.mtip_projection(g['G18_proj_I'], u), g['G18_proj_out']) < 1e-14


def check_polar2d_golden_hip(g, lib_path=None):
    """the mtip2d_* operators (k_polar2d.hip) against the same fixture, two grids per call"""
    from xframe_amd.fxs.polar2d import Engine2D
    N, M, kappa, max_q = int(g['G18_N']), int(g['G18_M']), float(g['G18_kappa']), float(g['G18_max_q'])
    e = Engine2D(N, M, max_q, kappa, n_batch=2, lib_path=lib_path)
    assert np.allclose(e.rs, g['G18_rs'], rtol=1e-14) and np.allclose(e.qs, g['G18_qs'], rtol=1e-14)
    x = g['G18_x']
    xb = np.stack([x, 2.0 * x[::-1]])
    for got, ref in ((e.harmonic(xb), g['G18_cht_fwd']), (e.harmonic(xb, True), g['G18_cht_inv']), (e.hankel(xb), g['G18_hankel_fwd_all']),
                     (e.hankel(xb, True), g['G18_hankel_inv_all']), (e.fourier_transform(xb), g['G18_ft_all']),
                     (e.fourier_transform(xb, True), g['G18_ift_all'])):
        assert rel_l2(got[0], ref) < 1e-12
    assert rel_l2(e.real_harmonic_forward(xb)[0], g['G18_rht_fwd']) < 1e-12
    xr = g['G18_xr']
    assert rel_l2(e.real_harmonic_inverse(np.stack([xr, xr]))[1], g['G18_rht_inv']) < 1e-12
    used = g['G18_used_sub']
    e.set_projection(g['G18_proj_pm'], {int(o): int(o) for o in used}, g['G18_proj_mask'], float(g['G18_proj_n_particles']))
    out, unk = e.project(np.stack([g['G18_proj_I'], g['G18_proj_I']]))
    assert rel_l2(unk[0], g['G18_proj_unknowns']) < 1e-12 and rel_l2(out[1], g['G18_proj_out']) < 1e-12
    e.close()


def check_polar2d_vs_oracle(N, M, lib_path=None, seed=0):
    """the 2-D Fourier pair and projection at other sizes against the oracle; round trip ift(ft(x)) as the size-independent property"""
    from oracle import polar2d as P2
    from xframe_amd.fxs.polar2d import Engine2D
    rng = np.random.default_rng(seed)
    max_q = 0.7
    fp = P2.PolarFourierPair(N, M, max_q, 2.0)
    e = Engine2D(N, M, max_q, 2.0, n_batch=2, lib_path=lib_path)
    x = cplx(rng, (2, N, 2 * M + 1))
    F = e.fourier_transform(x)
    for b in range(2):
        assert rel_l2(F[b], fp.ft(x[b])) < 1e-11
        assert rel_l2(e.fourier_transform(x, True)[b], fp.ift(x[b])) < 1e-11
    # size-independent property: the pair is linear (the midpoint pair is NOT an exact inverse pair upstream either: a smooth function
    # comes back from ift(ft(.)) with 8 % error at 128 x M64 and kappa = 2, oracle and device alike)
    y = cplx(rng, (2, N, 2 * M + 1))
    assert rel_l2(e.fourier_transform(2.0 * x - 0.5j * y), 2.0 * F - 0.5j * e.fourier_transform(y)) < 1e-12
    r, phi = fp.rs[:, None], fp.phis[None, :]
    smooth = (np.exp(-(r / (0.25 * fp.r_max)) ** 2) * (1.0 + 0.5 * np.cos(2 * phi) + 0.2 * np.sin(3 * phi))).astype(complex)
    back = e.fourier_transform(e.fourier_transform(np.stack([smooth, smooth])), True)[0]
    assert rel_l2(back, fp.ift(fp.ft(smooth))) < 1e-11
    used = {int(o): int(o) for o in range(0, M + 1) if o != 3}
    pm = cplx(rng, (len(used), N))
    mask = rng.random((M + 1, N)) > 0.2
    rp = P2.ReciprocalProjection2D(pm, used, mask, fp.qs, M + 1, 2.5)
    e.set_projection(pm, used, mask, 2.5)
    I = cplx(rng, (2, N, M + 1))
    out, unk = e.project(I)
    for b in range(2):
        u = rp.approximate_unknowns(I[b])
        assert rel_l2(unk[b], u) < 1e-12 and rel_l2(out[b], rp.mtip_projection(I[b], u)) < 1e-12
    e.close()


def check_symmetric_eig(lib_path=None, n=130, K=3, seed=0):
    """mtip_op_symmetric_eig against LAPACK on what `extract` feeds it (fxs_invariant_tools.py:1114-1131): a rank-deficient
    semi-definite matrix (B_l of 2l+1 coefficients), an indefinite one, the zero matrix; n > 128 takes the solver that cuts the
    columns into blocks over workgroups.  Eigenvalues to eps |A|, eigenvectors through A = V diag(w) V^T and orthonormality."""
    e = Engine({'grid': {'n_radial_points': 8, 'max_order': 2}}, None, n_batch=1, lib_path=lib_path, max_q=1.0)
    rng = np.random.default_rng(seed)
    B = np.empty((K, n, n))
    for k in range(K):
        if k % 3 == 0:
            A = rng.normal(size=(n, min(n, 2 * k + 7))) * np.logspace(0, -6, min(n, 2 * k + 7))[None, :]
            B[k] = A @ A.T
        elif k % 3 == 1:
            M = rng.normal(size=(n, n))
            B[k] = (M + M.T) / 2
        else:
            B[k] = 0
    vals, vecs = e.hermitian_eig(B.astype(complex))
    for k in range(K):
        w = np.linalg.eigvalsh(B[k])[::-1]
        scale = max(np.abs(w).max(), 1e-300)
        assert np.abs(vals[k] - w).max() / scale < 1e-12, (k, np.abs(vals[k] - w).max() / scale)
        assert np.linalg.norm((vecs[k] * vals[k]) @ vecs[k].conj().T - B[k]) <= 1e-12 * max(np.linalg.norm(B[k]), 1e-300), k
        assert np.abs(vecs[k].conj().T @ vecs[k] - np.eye(n)).max() < 1e-12, k
    e.close()


def check_extract_vs_numpy(lib_path=None, N=24, L=6):
    """`extract` (fxs_invariant_tools.py:1079-1207): B_l -> V_l with the device eigensolver against numpy's eigh -- compared
    through eigenvalues and through V_l V_l^+ (eigenvectors are only defined up to phases / rotations inside degenerate
    spaces), incl. a matrix with negative eigenvalues (clipped) and the rank 2l+1 < Nq structure of real B_l."""
    fpd = FourierPair(SHT(L), N, S.data_cutoff(N), 2.0)
    tr = OracleTransforms(fpd)
    e = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, lib_path=lib_path, max_q=S.data_cutoff(N))
    d_np, _ = S.make_invariants(tr, N, L)
    d_dev, _ = S.make_invariants(tr, N, L, eigh=e)
    for l in range(L + 1):
        a, b = d_np['data_projection_matrices'][l], d_dev['data_projection_matrices'][l]
        assert a.shape == b.shape == (N, min(N, 2 * l + 1))
        assert rel_l2(b @ b.conj().T, a @ a.conj().T) < 1e-11, l
    assert np.allclose(d_np['average_intensity'], d_dev['average_intensity'], rtol=1e-13)
    rng = np.random.default_rng(5)
    m = cplx(rng, (3, N, N))
    m = m + np.conj(np.swapaxes(m, -1, -2))                         # indefinite Hermitian matrices
    w, v = e.hermitian_eig(m)
    for k in range(3):
        assert np.allclose(w[k], np.linalg.eigvalsh(m[k])[::-1], rtol=1e-12, atol=1e-12 * np.abs(w[k]).max())
        assert rel_l2(m[k] @ v[k], v[k] * w[k][None, :]) < 1e-11
        assert np.abs(v[k].conj().T @ v[k] - np.eye(N)).max() < 1e-12
    # eigenvalue pairs +x / -x share a singular value: the case the shifted repeat of the solver exists for
    q, _ = np.linalg.qr(cplx(rng, (N, N)))
    lam = np.concatenate([np.arange(1, N // 2 + 1), -np.arange(1, N - N // 2 + 1)]).astype(float)
    mp = (q * lam[None, :]) @ q.conj().T
    w, v = e.hermitian_eig(np.stack([mp, m[0]]))
    assert np.allclose(w[0], np.sort(lam)[::-1], atol=1e-11 * N)
    assert rel_l2(mp @ v[0], v[0] * w[0][None, :]) < 1e-11
    assert rel_l2(m[0] @ v[1], v[1] * w[1][None, :]) < 1e-11
    pms, evs = e.extract_projection_matrices(m, orders=[1, 2, 30])
    for k, l in enumerate([1, 2, 30]):
        kk = min(N, 2 * l + 1)
        ww, vv = np.linalg.eigh(m[k])
        ww, vv = ww[::-1][:kk], vv[:, ::-1][:, :kk]
        ww = np.where(ww < 0, 0, ww)
        assert pms[k].shape == (N, kk) and np.allclose(evs[k], ww, rtol=1e-12, atol=1e-12)
        assert rel_l2(pms[k] @ pms[k].conj().T, (vv * ww[None, :]) @ vv.conj().T) < 1e-11
    e.close()


def _extract_fixture():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'extract_ops.npz'))


def _outer(v):
    return v @ np.conj(v).T


def check_extract_rules_golden(impl, tol=1e-10):
    """G15: the reference's own `extract` numerics (tests/golden/extract_ops.npz: deg2_invariant_eigenvalues,
    deg2_invariant_to_projection_matrices_3d, nearest_positive_semidefinite_matrix run on seeded matrices by make_golden.py
    extract) against `impl` = (eigenvalues(B, sort_mode), projection_matrix(B, limits, order, sort_mode), psd(A, floor)).
    Eigenvectors are only defined up to signs / rotations inside degenerate spaces: compared through eigenvalues and through
    V V^+ (what the phasing uses of V_l)."""
    g = _extract_fixture()
    f_eig, f_pm, f_psd = impl
    n = 12
    for name in g['G15_names']:
        B = g[f'G15_{name}_B']
        scale = max(np.abs(B).max(), 1e-300)
        for sm in (0, 1):
            w, v = f_eig(B.copy(), sm)
            w_ref, v_ref = g[f'G15_{name}_eigvals_s{sm}'], g[f'G15_{name}_eigvecs_s{sm}']
            assert np.abs(np.sort(w) - np.sort(w_ref)).max() <= tol * scale * n, (name, sm)
            if sm == 0:
                assert np.abs(w - w_ref).max() <= tol * scale * n, (name, sm)        # descending eigenvalues: the order is pinned too
                # spectral projectors of the well separated top eigenvalues
                Bh = (B + B.conj().T) / 2
                assert np.abs((v * w[None, :]) @ np.conj(v).T - Bh).max() <= tol * scale * n or np.isclose(Bh, 0).all(), (name, sm)
            for order in (1, 2, 4, 7):
                for lname, lim in (('full', np.array([[0, n], [0, n]])), ('sub', np.array([[2, 10], [2, 10]]))):
                    pm, ev = f_pm(B.copy(), lim, order, sm)
                    pm_ref, ev_ref = g[f'G15_{name}_pm_l{order}_{lname}_s{sm}'], g[f'G15_{name}_ev_l{order}_{lname}_s{sm}']
                    assert pm.shape == pm_ref.shape and pm.dtype == pm_ref.dtype, (name, order, lname, sm)
                    if sm == 0:
                        assert np.abs(ev - ev_ref).max() <= tol * scale * n, (name, order, lname, sm)
                        # V V^+ is unique when the cut does not split a cluster of eigenvalues: true for the seeded cases at sort_mode 0
                        # except the indefinite / noisy full-rank matrices, whose cut falls between simple eigenvalues as well
                        assert np.abs(_outer(pm) - _outer(pm_ref)).max() <= 1e-8 * scale, (name, order, lname, sm)
                    else:
                        assert np.abs(np.sort(ev) - np.sort(ev_ref)).max() <= tol * scale * n, (name, order, lname, sm)
        for key, flag in (('psd', False), ('psd_floor', True)):
            a, ref = f_psd(B.copy(), flag), g[f'G15_{name}_{key}']
            assert np.abs(a - ref).max() <= 1e-9 * scale, (name, key)
    stack = np.stack([g['G15_psd_rank5_B'], g['G15_indefinite_B'], g['G15_zero_B']])
    assert np.abs(f_psd(stack, False) - g['G15_stack_psd']).max() <= 1e-9 * np.abs(stack).max()


def check_extract_rules_hip(lib_path=None):
    """the product's extract module (device eigensolver, reference rules) against G15"""
    from xframe_amd.fxs import extract as X
    e = Engine({'grid': {'n_radial_points': 12, 'max_order': 4}}, None, n_batch=1, lib_path=lib_path, max_q=S.data_cutoff(12))

    def eig(B, sm):
        w, v = X.deg2_invariant_eigenvalues(e, B[None], sm)
        return w[0], v[0]

    def pm(B, lim, order, sm):
        b = np.zeros((order + 1,) + B.shape, dtype=B.dtype)
        b[order] = B
        lims = np.broadcast_to(lim, (order + 1, 2, 2)).copy()
        p, ev = X.deg2_invariant_to_projection_matrices(e, b, lims, sm)
        return p[order], ev[order]

    check_extract_rules_golden((eig, pm, lambda A, flag: X.nearest_positive_semidefinite_matrix(e, A, flag)))
    e.close()
