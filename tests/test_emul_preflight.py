"""CPU pre-flight of the HIP kernel sources: the unchanged xframe_amd/csrc/*.hip files are compiled for the
host against tests/emul (fibers emulate the GPU threads of a block) and pushed through the same C ABI and
the same parity cases as the -m gpu suite, at toy sizes.  This catches index / barrier / out-of-bounds bugs
before any GPU minute is spent; it is NOT the parity claim (that is tests/test_gpu_parity.py on the MI355X)."""
import os
import subprocess

import numpy as np
import pytest

import parity_cases as PC

HERE = os.path.dirname(os.path.abspath(__file__))
EMUL_DIR = os.path.join(HERE, 'emul')
EMUL_LIB = os.path.join(EMUL_DIR, 'libmtip_emul.so')


@pytest.fixture(scope='session')
def emul_lib():
    r = subprocess.run(['make', '-C', EMUL_DIR, '-j6'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return EMUL_LIB


# n_phi = 16, 32, 8, 64, 128, 128, 256; the chained inverse -> forward kernel exists where the register FFTs do and a shell fits one
# CU's LDS (k_sht_chain.hip: run-time table variants at 16 / 32 / 64, register-table variants <2, 8> and <3, 8> at 128)
@pytest.mark.parametrize('N,L,chain', [(16, 4, True), (10, 7, True), (8, 2, False), (6, 12, True), (4, 24, True), (3, 32, True),
                                       (3, 44, False)])
def test_transforms(emul_lib, N, L, chain):
    PC.check_transforms(N, L, emul_lib, seed=N + L, expect_chain=chain)


def test_transforms_trapz(emul_lib):
    PC.check_transforms(12, 3, emul_lib, seed=3, mode='trapz')


def test_radial_rules_golden(emul_lib, golden_radial):
    PC.check_radial_rules_golden(golden_radial, emul_lib)


def test_transforms_golden(emul_lib, golden_ops):
    PC.check_transforms_golden(golden_ops, emul_lib)


def test_operators_golden(emul_lib, golden_ops):
    PC.check_operators_golden(golden_ops, emul_lib)


@pytest.mark.parametrize('fused', [False, True])
def test_single_steps_golden(emul_lib, golden_mtip16, fused):
    PC.check_steps_golden(golden_mtip16, emul_lib, fused)


@pytest.mark.parametrize('fused', [False, True])
def test_short_trajectory_vs_oracle(emul_lib, golden_mtip16, fused):
    PC.check_short_trajectory_vs_oracle(golden_mtip16, emul_lib, fused)


def test_engine_group_rendezvous(emul_lib, golden_mtip16):
    PC.check_engine_group_rendezvous(golden_mtip16, emul_lib)


def test_ft_stab_disagreement(emul_lib, golden_mtip16):
    PC.check_ft_stab_disagreement(golden_mtip16, emul_lib)


@pytest.mark.parametrize('fused', [False, True])
def test_group_run_identical(emul_lib, golden_mtip16, fused):
    PC.check_group_run_identical(golden_mtip16, emul_lib, fused)


@pytest.mark.parametrize('fused', [False, True])
def test_non_fxs_variants_vs_oracle(emul_lib, golden_mtip16, fused):
    PC.check_non_fxs_trajectory_vs_oracle(golden_mtip16, emul_lib, fused)


@pytest.mark.parametrize('ropt', [{'odd_orders_to_0': False}, {'use_averaged_intensity': False},
                                  {'odd_orders_to_0': False, 'use_averaged_intensity': False},
                                  {'used_order_ids': np.arange(3)}])
def test_reciprocal_option_variants_vs_oracle(emul_lib, golden_mtip16, ropt):
    """modify_projection_matrices switches (fxs_Projections.py:679-714): odd orders kept (all L+1 polar factors, some of
    zero matrices), V_0 taken from the data, a subset of the orders (the others pass through unprojected)."""
    PC.check_short_trajectory_vs_oracle(golden_mtip16, emul_lib, True, reciprocal_opt=ropt)


@pytest.mark.parametrize('name', sorted(PC.SETTINGS_VARIANTS))
def test_settings_variants_vs_oracle(emul_lib, golden_mtip16, name):
    PC.check_settings_variant_vs_oracle(golden_mtip16, emul_lib, name)


@pytest.mark.parametrize('fused', [True, False])
def test_zernike_rule_trajectories(emul_lib, golden_mtip16, fused):
    PC.check_zernike_rule_trajectories(golden_mtip16, emul_lib, fused)


def test_split_shell_steps_vs_oracle(emul_lib):
    PC.check_split_shell_steps_vs_oracle(emul_lib)


def test_shift_to_center_vs_oracle(emul_lib, golden_mtip16):
    PC.check_shift_to_center_vs_oracle(golden_mtip16, emul_lib)


def test_best_reselection_vs_oracle(emul_lib, golden_mtip16):
    PC.check_best_reselection_vs_oracle(golden_mtip16, emul_lib, True)


@pytest.mark.parametrize('fused', [False, True])
def test_sw_center_vs_oracle(emul_lib, golden_mtip16, fused):
    PC.check_sw_center_trajectory_vs_oracle(golden_mtip16, emul_lib, fused)


def test_projection_general_kernels_large(emul_lib):
    """the general (complex) projection at 2l+1 = 69: X_l and V_r still share one CU's LDS (unpadded columns)"""
    PC.check_projection_vs_oracle(70, 34, emul_lib, n_batch=1)


@pytest.mark.parametrize('kind', ['bump', 'low_resolution_autocorrelation'])
def test_initial_density_batch(emul_lib, golden_mtip16, kind):
    PC.check_initial_density_batch(golden_mtip16, emul_lib, kind)


def test_apply_unknowns_operator(emul_lib, golden_mtip16):
    PC.check_apply_unknowns(golden_mtip16, emul_lib)


def test_worker_engines_vs_single_and_oracle(emul_lib):
    """the config-4 worker case of the GPU suite at a toy size: 4 restarts on 2 engines"""
    PC.check_config4_worker(emul_lib, cfg=1, n_restarts=4, n_workers=2, n_hio=3, n_er=2, oracle_restarts=(0, 1), sizes=(12, 4))


@pytest.mark.parametrize('name', PC.VARIANT_NAMES)
def test_loop_variants_golden(emul_lib, golden_mtip16, golden_variants, name):
    PC.check_variant_golden(golden_mtip16, golden_variants, name, emul_lib, n_restarts=1)


@pytest.mark.parametrize('name', ['extra_metrics', 'so_freedom', 'swcenter'])
def test_loop_variants_golden_unfused(emul_lib, golden_mtip16, golden_variants, name):
    """the same trajectories of the reference with the step assembled from the separate operators (fused = False)"""
    PC.check_variant_golden(golden_mtip16, golden_variants, name, emul_lib, n_restarts=1, fused=False)


def test_average_vs_oracle(emul_lib):
    """alignment + averaging of reconstructions (average.py run_3d): device SO(3) correlation / coefficient rotation / transforms
    against the oracle restatement"""
    PC.check_average_vs_oracle(emul_lib)


def test_average_flow_golden(emul_lib, golden_flow):
    """the product's averaging against what the reference's own run_3d / Alignment did with two seeded sets (fixture G17)"""
    PC.check_average_flow_golden_hip(golden_flow, emul_lib)


def test_invariant_metrics_golden(emul_lib, golden_metrics):
    PC.check_invariant_metrics_golden_hip(golden_metrics, emul_lib)


@pytest.mark.parametrize('fused', [True, False])
def test_invariant_metrics_vs_oracle(emul_lib, golden_mtip16, fused):
    """II_error / ccd_diff / fqc_error recorded by the loop against the oracle's routines on the oracle's trajectory"""
    PC.check_invariant_metrics_vs_oracle(golden_mtip16, emul_lib, fused)


@pytest.mark.parametrize('cat,name', [('real', 'support_size'), ('reciprocal', 'no_such_metric')])
def test_unknown_error_metrics_are_rejected(emul_lib, golden_mtip16, cat, name):
    """a metric this build does not record raises instead of being dropped silently: `support_size` (fxs_IO_methods.py:685-688 raises
    upstream too: it is handed the projection's output list) and unknown names"""
    from helpers import data_from_golden, golden_settings
    from xframe_amd.fxs.engine import Engine
    g = golden_mtip16
    N, L = int(g['N']), int(g['L'])
    opt = golden_settings(N, L)
    calc = opt['main_loop']['error']['methods'][cat]
    calc['calculate'] = list(calc.get('calculate', [])) + [name]
    with pytest.raises(NotImplementedError):
        Engine(opt, data_from_golden(g, L), n_batch=1, lib_path=emul_lib)


def test_polar2d_golden(emul_lib, golden_polar2d):
    """the 2-D operators against the reference's own functions (fixture G18)"""
    PC.check_polar2d_golden_hip(golden_polar2d, emul_lib)


def test_polar2d_vs_oracle(emul_lib):
    PC.check_polar2d_vs_oracle(10, 6, emul_lib)


def test_mtip2d_loop_golden(emul_lib, golden_mtip2d):
    """the 2-D phasing loop on the device operators against the reference's own 2-D run (fixture G20)"""
    PC.check_mtip2d_golden_hip(golden_mtip2d, emul_lib)


@pytest.mark.parametrize('name', PC.MTIP2D_VARIANTS)
def test_mtip2d_variants_golden(emul_lib, golden_mtip2d, golden_mtip2d_variants, name):
    PC.check_mtip2d_variant_golden_hip(golden_mtip2d, golden_mtip2d_variants, name, emul_lib)


def test_polar2d_radial_rules(emul_lib, golden_polar2d_rules):
    PC.check_polar2d_rules_golden(golden_polar2d_rules, emul_lib)


def test_mtip2d_ft_stab_disagreement(emul_lib, golden_mtip2d):
    PC.check_mtip2d_ft_stab_disagreement(golden_mtip2d, emul_lib)


def test_mtip2d_unbuildable_variants(emul_lib, golden_mtip2d, golden_mtip2d_variants):
    PC.check_mtip2d_unbuildable_variants(golden_mtip2d, golden_mtip2d_variants, emul_lib)


def test_mtip2d_worker_vs_oracle(emul_lib, golden_mtip2d):
    PC.check_mtip2d_worker_vs_oracle(golden_mtip2d, emul_lib)


@pytest.mark.parametrize('name', sorted(PC.SETTINGS_VARIANTS_2D))
def test_mtip2d_settings_vs_oracle(emul_lib, golden_mtip2d, name):
    PC.check_mtip2d_settings_vs_oracle(golden_mtip2d, emul_lib, name)


def test_symmetric_eig_blocked(emul_lib):
    """n > 128: column blocks over workgroups (k_sym_eig_block), 5 blocks -> an empty sixth pads the tournament"""
    PC.check_symmetric_eig(emul_lib, n=130, K=3)


def test_extract_vs_numpy(emul_lib):
    PC.check_extract_vs_numpy(emul_lib, N=12, L=4)


@pytest.mark.parametrize('N,L,ropt', [(16, 4, None), (24, 10, None), (16, 4, {'odd_orders_to_0': False, 'use_averaged_intensity': False}),
                                      (16, 4, {'used_order_ids': np.arange(3)}),
                                      (16, 4, {'SO_freedom': {'use': True, 'radial_high_pass': 0.2}})])
def test_projection_real_vs_oracle(emul_lib, N, L, ropt):
    """the real-arithmetic projection kernel (k_projr.hip) on the emulator"""
    PC.check_projection_real_vs_oracle(N, L, emul_lib, n_batch=1, reciprocal_opt=ropt)


@pytest.mark.parametrize('env,closing', [({'MTIP_RP_CORR': '0'}, 'none'), ({}, 'some'),
                                         ({'MTIP_RP_EARLY': '0.2', 'MTIP_RP_CORR2_MAX': '5e-4'}, 'some'),
                                         ({'MTIP_RP_EARLY': '0.2', 'MTIP_RP_CORR2_MAX': '1e-6'}, None),
                                         ({'MTIP_RP_EARLY': '7', 'MTIP_RP_CORR2_MAX': '1'}, 'some')])
def test_projection_real_switches(emul_lib, env, closing, monkeypatch):
    """the closing-step switches of k_rproj: classic confirming sweep (MTIP_RP_CORR=0), the two ends of the tested threshold
    range, and values beyond it (clamped by mtip_create: the operator keeps its 1e-10)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    PC.check_projection_real_vs_oracle(24, 10, emul_lib, n_batch=1, closing=closing)


def test_prtf_golden(emul_lib):
    PC.check_prtf_golden(emul_lib)


def test_find_rotation_nan_is_the_maximum(emul_lib):
    PC.check_find_rotation_nan(emul_lib)


def test_polar_timing_records_do_not_overlap(emul_lib):
    PC.check_polar_timing_records(emul_lib)


def test_so_freedom_on_a_higher_order(emul_lib):
    """SO_freedom (fxs_Projections.py:768-780) forced onto l = 4, where column 2 is m = -2 and the correction is not a no-op"""
    PC.check_projection_real_vs_oracle(16, 6, emul_lib, n_batch=1, so_order=4)


def test_projection_real_tolerance_opt_in(emul_lib):
    """Im V_l at rounding level (the reference's `density` route): general kernels by default, the real one under MTIP_PROJ_REAL_TOL"""
    PC.check_projection_real_vs_oracle(16, 6, emul_lib, n_batch=1, imag_residue=1e-15)


def test_wide_projection_matrices(emul_lib):
    """k_l = Nq < 2l+1 (the reference's integration test uses 8 radial points with max_order 15,
    tests/test_fxs_integration.py:326-355): polar factor of a wide matrix, compared through V_l U_l."""
    from helpers import rel_l2
    from oracle.fourier import FourierPair
    from oracle.sht import SHT
    from oracle import mtip as OM
    from helpers import OracleTransforms, golden_settings
    from xframe_amd.fxs import synthetic as S
    from xframe_amd.fxs.engine import Engine
    N, L = 8, 7
    fpd = FourierPair(SHT(L), N, S.data_cutoff(N), 2.0)
    data, _ = S.make_invariants(OracleTransforms(fpd), N, L)
    opt = golden_settings(N, L)
    e = Engine(opt, data, n_batch=1, lib_path=emul_lib)
    om = OM.MTIP(opt, data)
    rng = np.random.default_rng(1)
    Ilm = PC.cplx(rng, (1, N, e.nlm))
    proj = e.project_coefficients(Ilm)[0]
    Il = [Ilm[0][:, l * l:(l + 1) ** 2] for l in range(L + 1)]
    unk = om.rp.approximate_unknowns(Il)
    ref = np.concatenate(om.rp.mtip_projection(Il, unk), axis=1)
    assert rel_l2(proj, ref) < 1e-10
    U = e.unknowns(0)
    for l in range(L + 1):
        assert U[l].shape == (min(2 * l + 1, N), 2 * l + 1)
    e.close()


def test_extract_rules_golden(emul_lib):
    PC.check_extract_rules_hip(emul_lib)
