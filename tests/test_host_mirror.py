"""Host-side mirror of the reference API (worker schema, operator registry + sketches, GPU-process boundary),
run on the CPU emulation build of the kernels."""
import os
import subprocess

import numpy as np
import pytest

import boundary_cases as BC
from helpers import data_from_golden, golden_settings, rel_l2

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL_DIR = os.path.join(HERE, 'emul')
EMUL_LIB = os.path.join(EMUL_DIR, 'libmtip_emul.so')


@pytest.fixture(scope='module')
def emul_lib():
    r = subprocess.run(['make', '-C', EMUL_DIR, '-j6'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return EMUL_LIB


def test_worker_result_schema(emul_lib, golden_mtip16):
    """The schema the reference's integration test pins (tests/test_fxs_integration.py:388-423)."""
    from xframe_amd.fxs import reconstruct as R
    g = golden_mtip16
    N, L = int(g['N']), int(g['L'])
    opt = golden_settings(N, L, {'multi_process': {'use': True, 'n_parallel_reconstructions': 2}})
    main = opt['main_loop']['sub_loops']['main']
    main['methods']['HIO']['iterations'] = 2
    main['methods']['ER']['iterations'] = 2
    main['iterations'] = 2
    w = R.ProjectWorker(opt, data_from_golden(g, L), seeds=[1, 2], lib_path=emul_lib)
    result, _ = w.run()
    assert result.dtype == object and len(result) == 2
    n_steps = 2 * (2 + 2)
    shape = (N, 8, 16)
    for r in result:
        for key in ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density', 'initial_density'):
            assert r[key].shape == shape and r[key].dtype == np.complex128 and not np.isnan(r[key]).any(), key
        for key in ('initial_support', 'support_mask', 'last_support_mask'):
            assert r[key].shape == shape and r[key].dtype == bool, key
        assert r['error_dict']['main'].shape == (n_steps,)
        assert r['error_dict']['real']['l2_projection_diff'].shape == (n_steps,)
        assert len(r['fxs_unknowns']) == L + 1
        for l, u in enumerate(r['fxs_unknowns']):
            assert u.shape == (min(2 * l + 1, N), 2 * l + 1) and u.dtype == np.complex128
        assert r['last_deg2_invariant'].shape == (L + 1, N, N)
        assert r['loop_iterations'] == 3 and np.isfinite(r['final_error'])
        assert r['grid_pair']['real_grid'].shape == shape + (3,)
        assert len(r['projection_matrices']) == L + 1
        assert r['n_particles'].shape == (n_steps, 1)
    assert set(w.results['reconstruction_results']) == {'0', '1'}
    with pytest.raises(RuntimeError):
        R.ProjectWorker(dict(opt, GPU={'use': False}), data_from_golden(g, L), lib_path=emul_lib)


def test_jacobi_pairing_schedule_is_a_valid_sweep(emul_lib):
    """The resident-column ordering of the polar-factor kernel is generated and verified on the host for every column
    count (every pair once per sweep, no column twice in a round); a failing verification would silently fall back
    to the round-robin ordering, so check it here for all sizes the kernel accepts."""
    from xframe_amd.fxs.engine import Engine
    e = Engine({'grid': {'n_radial_points': 4, 'max_order': 2}}, None, n_batch=1, max_q=1.0, lib_path=emul_lib)
    assert e.lib.mtip_debug_check_jacobi_schedule(e.ctx, 127) == 0
    assert e.lib.mtip_debug_check_jacobi_schedule(e.ctx, 1) != 0
    e.close()


def test_worker_restart_groups_match_single_engine(emul_lib, golden_mtip16):
    """GPU.n_gpu_workers > 1 splits the restarts of a rank into concurrently driven engines (host threads, one HIP stream
    each); restarts are independent, so every restart must come out exactly as from one engine holding them all."""
    from xframe_amd.fxs import reconstruct as R
    g = golden_mtip16
    N, L = int(g['N']), int(g['L'])
    outs = []
    for workers in (1, 2):
        opt = golden_settings(N, L, {'multi_process': {'use': True, 'n_parallel_reconstructions': 4},
                                     'GPU': {'use': True, 'n_gpu_workers': workers}})
        main = opt['main_loop']['sub_loops']['main']
        main['methods']['HIO']['iterations'] = 2
        main['methods']['ER']['iterations'] = 1
        main['iterations'] = 1
        w = R.ProjectWorker(opt, data_from_golden(g, L), seeds=[11, 12, 13, 14], lib_path=emul_lib)
        result, _ = w.run()
        assert len(result) == 4
        assert len(getattr(w, 'mtip_instances')) == workers
        outs.append(result)
    for a, b in zip(*outs):
        assert np.array_equal(a['initial_density'], b['initial_density'])
        assert np.array_equal(a['real_density'], b['real_density'])
        assert np.array_equal(a['error_dict']['main'], b['error_dict']['main'])


def test_reference_sketch_on_registry(emul_lib, golden_mtip16):
    BC.check_reference_sketch_on_registry(golden_mtip16, emul_lib)


def test_gpu_process_boundary(emul_lib):
    BC.check_gpu_process_boundary(emul_lib)


def test_gpu_process_from_child_processes(emul_lib):
    BC.check_gpu_process_from_child_processes(emul_lib, n_processes=2)


@pytest.mark.parametrize('name', ['none', 'region_lo', 'region_both', 'line', 'from_pm'])
def test_radial_mask_types_golden(golden_ops, name):
    """generate_radial_mask (fxs_Projections.py:578-629), every q_mask type: the reference's own masks (fixture
    G3m_*, tests/golden/make_golden.py) against the oracle restatement and the host mirror."""
    from oracle import projections as OP
    from xframe_amd.fxs import hostsetup as hs
    g = golden_ops
    N, L = 16, 4
    data = dict(data_from_golden(g, L, prefix='D16_'))
    data['data_projection_matrices_q_id_limits'] = {'I1I1': g['G3m_q_id_limits']}
    q_mask = {
        'none': {'type': 'none'},
        'region_lo': {'type': 'manual', 'manual': {'type': 'region', 'region': [float(g['G3m_region_lo_pt']), False]}},
        'region_both': {'type': 'manual', 'manual': {'type': 'region', 'region': [float(x) for x in g['G3m_region_both_pts']]}},
        'line': {'type': 'manual', 'manual': {'type': 'order_dependent_line',
                                              'order_dependent_line': g['G3m_line_points'].tolist()}},
        'from_pm': {'type': 'from_projection_matrices'},
    }[name]
    opt = golden_settings(N, L, {'projections': {'reciprocal': {'q_mask': q_mask}}})['projections']['reciprocal']
    qs = np.asarray(g['G2_qs'])
    want = g['G3m_' + name]
    assert (OP.ReciprocalProjection(qs, data, L, opt).radial_mask == want).all()
    assert (hs.ReciprocalSetup(qs, data, L, opt).radial_mask == want).all()


@pytest.mark.parametrize('kind', ['bump', 'ball', 'low_resolution_autocorrelation'])
def test_density_guess_matches_oracle(emul_lib, golden_mtip16, kind):
    """generate_density_guess_method (reconstruct.py:1115-1174): the worker's seeded guess against the oracle's for
    the same generator state, and the normalisation int |rho|^2 = integrated intensity."""
    from oracle import mtip as OM
    from xframe_amd.fxs import reconstruct as R
    g = golden_mtip16
    N, L = int(g['N']), int(g['L'])
    data = data_from_golden(g, L)
    opt = golden_settings(N, L, {'density_guess': {'type': kind}})
    om = OM.MTIP(opt, data)
    want = om.density_guess(np.random.default_rng(77))
    R.MTIP.preinit(opt, data)
    m = R.MTIP(n_restarts=1, seeds=[77], lib_path=emul_lib)
    m.generate_phasing_loop()
    got = m._initial_density(0)
    assert rel_l2(got, want) < (1e-9 if kind == 'low_resolution_autocorrelation' else 1e-13)
    assert np.isclose(om.integrator.integrate((want * want.conj()).real), om.rp.integrated_intensity, rtol=1e-12)
    if kind == 'ball':
        assert (got == 0).any() and (got != 0).any()
    m.engine.close()


def test_calc_center_and_shift_golden(golden_ops):
    """calc_center (misk.py:295-312) and the shift operator (fxs_Projections.py:1419-1444): the reference's own centre
    and phase ramps (fixture G12_*) against the oracle restatement and the host mirror."""
    from oracle import projections as OP
    from xframe_amd.fxs import hostsetup as hs
    g = golden_ops
    rs, qs, theta, phi = g['G2_rs'], g['G2_qs'], g['G2_theta'], g['G2_phi']
    dens, want = g['G12_density'], g['G12_center']
    r3, t3, p3 = np.meshgrid(rs, theta, phi, indexing='ij')
    c_or = OP.calc_center(rs, len(theta), np.stack((r3, t3, p3), -1), dens)
    c_hs = hs.calc_center(rs, theta, phi, dens)
    assert np.allclose(c_or, want, rtol=1e-12) and np.allclose(c_hs, want, rtol=1e-12)
    q3, t3, p3 = np.meshgrid(qs, theta, phi, indexing='ij')
    qgrid = np.stack((q3, t3, p3), -1)
    for name, opp in (('G12_phases_neg', True), ('G12_phases_pos', False)):
        assert rel_l2(OP.shift_phases(qgrid, want, opp), g[name]) < 1e-13
        assert rel_l2(hs.shift_phases(qs, theta, phi, want, opp), g[name]) < 1e-13


def test_reference_sw_and_shift_sketches_on_registry(emul_lib, golden_mtip16):
    BC.check_reference_sw_and_shift_sketches_on_registry(golden_mtip16, emul_lib)


def test_unsupported_radial_rules_raise():
    """the four rules of the reference exist (midpoint, trapz, gauss, Zernike: hankel_transforms.py:15); anything else -- e.g. the
    lower-case 'zernike' of default_0.01.yaml:25, which the reference's mode test does not know either -- raises"""
    from xframe_amd.fxs import hostsetup as hs
    for mode in ('midpoint', 'trapz', 'gauss', 'Zernike'):
        assert len(hs.radial_grids(1.0, 8, 2.0, mode)[0]) == 8 and hs.hankel_raw_weights(2, 8, 2.0, mode).shape[0] == 3
    for mode in ('zernike', 'PNAS', 'simpson'):
        with pytest.raises(NotImplementedError):
            hs.radial_grids(1.0, 8, 2.0, mode)
        with pytest.raises(NotImplementedError):
            hs.hankel_raw_weights(2, 8, 2.0, mode)
