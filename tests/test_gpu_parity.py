"""Parity of the HIP path on a real MI355X, through the C ABI, against the oracle and the golden fixtures
captured from the reference (run with `pytest -m gpu`)."""
import numpy as np
import pytest

import parity_cases as PC

pytestmark = pytest.mark.gpu


def test_library_loaded_and_device_present():
    from xframe_amd.fxs import _lib
    lib = _lib.load()
    assert lib.mtip_device_count() >= 1


@pytest.mark.parametrize('N,L,chain', [(16, 4, True), (10, 7, True), (8, 2, False), (32, 8, True), (64, 16, True), (20, 24, True),
                                       (128, 32, True), (12, 44, False)])
def test_transforms(N, L, chain):
    PC.check_transforms(N, L, None, seed=N + L, expect_chain=chain)


def test_transforms_trapz():
    PC.check_transforms(12, 3, None, seed=3, mode='trapz')


def test_radial_rules_golden(golden_radial):
    """`gauss` and `Zernike` radial rules: device Hankel / Fourier pairs against the reference's own outputs (G21) and the oracle"""
    PC.check_radial_rules_golden(golden_radial, None)


def test_transforms_golden(golden_ops):
    PC.check_transforms_golden(golden_ops, None)


def test_operators_golden(golden_ops):
    PC.check_operators_golden(golden_ops, None)


@pytest.mark.parametrize('fused', [False, True])
def test_single_steps_golden(golden_mtip16, fused):
    PC.check_steps_golden(golden_mtip16, None, fused)


@pytest.mark.parametrize('fused', [False, True])
def test_short_trajectory_vs_oracle(golden_mtip16, fused):
    PC.check_short_trajectory_vs_oracle(golden_mtip16, None, fused)


@pytest.mark.parametrize('ropt', [{'odd_orders_to_0': False}, {'odd_orders_to_0': False, 'use_averaged_intensity': False}])
def test_reciprocal_option_variants_vs_oracle(golden_mtip16, ropt):
    PC.check_short_trajectory_vs_oracle(golden_mtip16, None, True, reciprocal_opt=ropt)


@pytest.mark.parametrize('name', ['limit_imag', 'value_lo_hi', 'value_hi_only', 'hio_considers_support_only', 'trapz', 'pi_in_q'])
def test_settings_variants_vs_oracle(golden_mtip16, name):
    PC.check_settings_variant_vs_oracle(golden_mtip16, None, name)


@pytest.mark.parametrize('fused', [False, True])
def test_split_shell_steps_vs_oracle(fused):
    PC.check_split_shell_steps_vs_oracle(None, N=12, L=48, fused=fused)


def test_shift_to_center_vs_oracle(golden_mtip16):
    PC.check_shift_to_center_vs_oracle(golden_mtip16, None)


def test_best_reselection_vs_oracle(golden_mtip16):
    PC.check_best_reselection_vs_oracle(golden_mtip16, None, True)


@pytest.mark.parametrize('fused', [False, True])
def test_sw_center_vs_oracle(golden_mtip16, fused):
    PC.check_sw_center_trajectory_vs_oracle(golden_mtip16, None, fused)


@pytest.mark.parametrize('fused', [False, True])
def test_non_fxs_variants_vs_oracle(golden_mtip16, fused):
    PC.check_non_fxs_trajectory_vs_oracle(golden_mtip16, None, fused)


@pytest.mark.parametrize('fused', [False, True])
def test_trajectory_golden_16(golden_mtip16, fused):
    PC.check_trajectory_golden(golden_mtip16, None, fused, n_restarts=2)


@pytest.mark.parametrize('fused', [False, True])
def test_trajectory_golden_cfg1(golden_cfg1, fused):
    """BASELINE config 1 (32 shells x L=8, 60 HIO + SW + 40 ER): the reference's own loop, 100 steps."""
    PC.check_trajectory_golden(golden_cfg1, None, fused, n_restarts=1)


@pytest.mark.parametrize('fused', [False, True])
def test_config2_trajectory_vs_oracle(fused):
    """64 x L16 (BASELINE config 2 size): 20 steps + shrink-wrap against the oracle, both step orders."""
    PC.check_config_trajectory_vs_oracle(2, fused=fused)


@pytest.mark.parametrize('N,L', [(40, 18), (66, 32), (96, 44)])
def test_projection_vs_oracle_sizes(N, L):
    """the general (complex) projection kernels on random coefficients without any symmetry -- (40, 18): LDS Jacobi with 3 row
    slots and odd/even k mixes; (66, 32): 65 x 65 matrices; (96, 44): X_l and V_r (2l+1 = 89, complex) do not share one CU's
    LDS -> the global-memory Jacobi fallback"""
    PC.check_projection_vs_oracle(N, L)


@pytest.mark.parametrize('N,L', [(16, 4), (40, 18), (66, 32), (128, 32), (100, 48)])
def test_projection_real_vs_oracle(N, L):
    """k_rproj (real V_l, coefficients of a real intensity: what the phasing loop runs) against the oracle's complex SVD route
    and against the general complex kernel; (128, 32): the benchmark's matrices, (100, 48): config 5's 97 x 97 (768 threads,
    pairing table read from L2)"""
    PC.check_projection_real_vs_oracle(N, L)


@pytest.mark.parametrize('env,closing', [({'MTIP_RP_CORR': '0'}, 'none'), ({}, 'some'),
                                         ({'MTIP_RP_EARLY': '0.2', 'MTIP_RP_CORR2_MAX': '5e-4'}, 'some'),
                                         ({'MTIP_RP_EARLY': '0.2', 'MTIP_RP_CORR2_MAX': '1e-6'}, None),
                                         ({'MTIP_RP_EARLY': '7', 'MTIP_RP_CORR2_MAX': '1'}, 'some')])
@pytest.mark.parametrize('N,L', [(66, 32), (100, 48)])
def test_projection_real_switches(N, L, env, closing, monkeypatch):
    """the closing-step switches of k_rproj on the padded (65-column) and the tight (97-column, Gram matrix in V_r's place) layout:
    classic confirming sweep, both ends of the tested threshold range, values beyond it (clamped by mtip_create)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    PC.check_projection_real_vs_oracle(N, L, closing=closing)


def test_projection_order_49_takes_the_general_kernels():
    """l = 49 (99 columns) needs 832 threads in the real kernel's pairing -- beyond its 768-thread instantiation: the launcher must
    refuse it (rproj_supported) and the general kernels run; odd orders active"""
    PC.check_projection_real_vs_oracle(100, 49, n_batch=1, reciprocal_opt={'odd_orders_to_0': False}, expect_real=False)


def test_prtf_golden():
    PC.check_prtf_golden(None)


def test_find_rotation_nan_is_the_maximum():
    PC.check_find_rotation_nan(None)


def test_polar_timing_records_do_not_overlap():
    PC.check_polar_timing_records(None)


@pytest.mark.parametrize('ropt', [{'odd_orders_to_0': False}, {'use_averaged_intensity': False}, {'used_order_ids': np.arange(3)},
                                  {'SO_freedom': {'use': True, 'radial_high_pass': 0.2}}])
def test_projection_real_option_variants(ropt):
    PC.check_projection_real_vs_oracle(24, 10, reciprocal_opt=ropt)


@pytest.mark.parametrize('N,L,so', [(16, 6, 4), (40, 18, 10)])
def test_so_freedom_on_a_higher_order(N, L, so):
    """SO_freedom (fxs_Projections.py:768-780) forced onto an order where column 2 is m = 2 - l != 0: the correction kernel after
    the real-arithmetic projection and after the general kernels, against the oracle"""
    PC.check_projection_real_vs_oracle(N, L, so_order=so)


def test_projection_real_tolerance_opt_in():
    """Im V_l at rounding level (the reference's `density` route of extract): general kernels by default, k_rproj under the
    opt-in MTIP_PROJ_REAL_TOL; the check tells the two apart by which half of the coefficients is read"""
    PC.check_projection_real_vs_oracle(40, 18, imag_residue=1e-15)


def test_config3_short_trajectory_vs_oracle():
    """128 x L32 (the benchmark size): 10 HIO + shrink-wrap + 10 ER ft_stab steps against the oracle."""
    PC.check_config_trajectory_vs_oracle(3, fused=True, n_hio=10, n_er=10)


@pytest.mark.gpu
def test_ft_stab_disagreement(golden_mtip16):
    """restarts of one engine that disagree on the ft_stab link: ft_stab per restart, each follows the oracle's run of it"""
    PC.check_ft_stab_disagreement(golden_mtip16)


@pytest.mark.gpu
@pytest.mark.parametrize('fused', [False, True])
def test_group_run_identical(golden_mtip16, fused):
    """mtip_run_group_async (engines of one GPU taking turns at the transforms) == mtip_run_async per engine, bit for bit"""
    PC.check_group_run_identical(golden_mtip16, None, fused)


@pytest.mark.gpu
@pytest.mark.parametrize('lag', ['1', '2'])
def test_group_run_identical_metric_grid(lag, monkeypatch):
    """the same at the benchmark's grid (chained kernels, k_rproj): three engines {2, 1, 1} of 128 x L32; both turn orders
    (MTIP_TURN_LAG: strict turns / at most two contexts in their transforms)"""
    monkeypatch.setenv('MTIP_TURN_LAG', lag)
    PC.check_group_run_identical_synthetic(3, sizes=(2, 1, 1))


@pytest.mark.gpu
def test_config4_worker_three_engines_vs_single_and_oracle():
    """128 x L32, eight distinct restarts on three engines (BASELINE config 4 as bench.py runs it on one GPU): bit-equal to
    one engine holding all eight, restarts 0 and 5 (different engine groups) against the oracle."""
    PC.check_config4_worker(None)


@pytest.mark.parametrize('kind', ['bump', 'low_resolution_autocorrelation'])
def test_initial_density_batch(golden_mtip16, kind):
    PC.check_initial_density_batch(golden_mtip16, None, kind)


def test_apply_unknowns_operator(golden_mtip16):
    PC.check_apply_unknowns(golden_mtip16, None)


@pytest.mark.parametrize('name', PC.VARIANT_NAMES)
def test_loop_variants_golden(golden_mtip16, golden_variants, name):
    PC.check_variant_golden(golden_mtip16, golden_variants, name, None)


@pytest.mark.parametrize('N,L', [(12, 6), (24, 10)])
def test_average_vs_oracle(N, L):
    """alignment + averaging of reconstructions (xframe/projects/fxs/average.py run_3d) against the oracle"""
    PC.check_average_vs_oracle(None, N=N, L=L)


def test_average_flow_golden(golden_flow):
    """the product's averaging against what the reference's own ProjectWorker.run_3d / Alignment did with two seeded sets of
    reconstructions (fixture G17, tests/golden/average_flow.npz): decisions, aligned pairs, averages, PRTF variants"""
    PC.check_average_flow_golden_hip(golden_flow)


def test_invariant_metrics_golden(golden_metrics):
    """II_error / ccd_diff / fqc_error on the device against the reference's own routines (fixture G19, operator level)"""
    PC.check_invariant_metrics_golden_hip(golden_metrics)


@pytest.mark.parametrize('fused', [True, False])
def test_invariant_metrics_vs_oracle(golden_mtip16, fused):
    """the non-default reciprocal metrics II_error / ccd_diff / fqc_error (fxs_IO_methods.py:587-627, 651-683, 507-550) as the loop
    records them per step (k_metrics.hip) against the oracle's routines evaluated on the oracle's trajectory"""
    PC.check_invariant_metrics_vs_oracle(golden_mtip16, None, fused)


def test_polar2d_golden(golden_polar2d):
    """the 2-D (polar) operators -- circular harmonic transforms, polar Hankel pair, Fourier pair, reciprocal projection -- against
    the reference's own functions (fixture G18, tests/golden/polar2d_ops.npz)"""
    PC.check_polar2d_golden_hip(golden_polar2d)


@pytest.mark.parametrize('N,M', [(10, 6), (64, 30), (128, 64)])
def test_polar2d_vs_oracle(N, M):
    PC.check_polar2d_vs_oracle(N, M)


def test_mtip2d_loop_golden(golden_mtip2d):
    """the 2-D phasing loop (reconstruct2d.MTIP2D on mtip2d_op_step / mtip2d_op_shrinkwrap): single HIO / ER steps with and without
    ft_stab, the shrink-wrap mask and the 14-step trajectory of the reference's own `dimensions: 2` run (fixture G20)"""
    PC.check_mtip2d_golden_hip(golden_mtip2d)


@pytest.mark.parametrize('name', PC.MTIP2D_VARIANTS)
def test_mtip2d_variants_golden(golden_mtip2d, golden_mtip2d_variants, name):
    """the 2-D loop's sub-variants on the device operators against the reference's own 2-D runs (fixture G22)"""
    PC.check_mtip2d_variant_golden_hip(golden_mtip2d, golden_mtip2d_variants, name)


def test_polar2d_radial_rules(golden_polar2d_rules):
    """2-D trapz / gauss / Zernike: device Hankel and Fourier pairs against the reference's own functions (fixture G23)"""
    PC.check_polar2d_rules_golden(golden_polar2d_rules)


def test_mtip2d_ft_stab_disagreement(golden_mtip2d):
    """restarts of one batch that disagree on the ft_stab link: each one follows the oracle's run of it"""
    PC.check_mtip2d_ft_stab_disagreement(golden_mtip2d)


def test_mtip2d_unbuildable_variants(golden_mtip2d, golden_mtip2d_variants):
    PC.check_mtip2d_unbuildable_variants(golden_mtip2d, golden_mtip2d_variants)


@pytest.mark.parametrize('N,M', [(None, None), (64, 30)])
def test_mtip2d_worker_vs_oracle(golden_mtip2d, N, M):
    """`dimensions: 2` through ProjectWorker with seeded guesses, every restart against the oracle's loop"""
    PC.check_mtip2d_worker_vs_oracle(golden_mtip2d, None, N, M)


@pytest.mark.parametrize('name', ['limit_imag', 'value_lo_hi', 'no_enforce', 'n_particles', 'q_mask_region', 'error_inside_support', 'ft_stab_linked',
                                  'best_reselected', 'rule_trapz', 'rule_gauss'])
def test_mtip2d_settings_vs_oracle(golden_mtip2d, name):
    """settings switches of the 2-D loop against the oracle (pinned at 0.0 by the reference's own 2-D run, G20)"""
    PC.check_mtip2d_settings_vs_oracle(golden_mtip2d, None, name)


@pytest.mark.parametrize('n,K', [(100, 6), (128, 33), (130, 5), (200, 7), (256, 49), (288, 4)])
def test_symmetric_eig(n, K):
    """the eigensolvers of `extract` against LAPACK: LDS-resident up to 128, column blocks over workgroups up to 288
    ((256, 49) = config 5's B_l)"""
    PC.check_symmetric_eig(None, n=n, K=K)


@pytest.mark.parametrize('N,L', [(24, 6), (128, 32)])
def test_extract_vs_numpy(N, L):
    """B_l -> V_l on the device (the `extract` step) against numpy eigh, small and at the benchmark size"""
    PC.check_extract_vs_numpy(None, N=N, L=L)


def test_extract_rules_golden():
    """G15: the product's `extract` rules on the device eigensolvers against the reference's own outputs"""
    PC.check_extract_rules_hip(None)


def test_config2_properties():
    PC.check_full_size_properties(2)


def test_zernike_rule_trajectories(golden_mtip16):
    PC.check_zernike_rule_trajectories(golden_mtip16, None)


def test_config3_properties_full_size():
    """128 shells x L_max = 32 (the metric's configuration)."""
    errs = PC.check_full_size_properties(3, n_steps=8)
    assert np.isfinite(errs).all()


def test_config5_properties_full_size():
    """256 shells x L_max = 48 (BASELINE config 5, 128 x 256 angular grid): the inverse SHT shares a shell between two
    workgroups (fused epilogues, two error partial sums per shell; no chained inverse -> forward kernel: a shell of this grid
    does not fit one CU), the forward SHT takes the pass-wise table kernel with prefetched rows, and the 97-column polar factors
    run in k_rproj's tight layout (768 threads, pairing table from L2, the closing step's Gram matrix in V_r's place); same
    size-independent properties as at the metric's size (fused == reference order, round trips, B_l of the projection ==
    data B_l).  Its oracle coverage: test_projection_real_vs_oracle[100-48], test_projection_real_switches[100-48-*],
    test_split_shell_steps_vs_oracle."""
    errs = PC.check_full_size_properties(5, n_steps=4)
    assert np.isfinite(errs).all()


# ---- every MTIP_* switch that selects another kernel of the shipping library gets a forced parity case ------------------
_TRAJ_SWITCHES = [('MTIP_SHT_MODE', '0'), ('MTIP_SHT_MODE', '1'), ('MTIP_SHT_WIDE', '0'), ('MTIP_FUSE_REAL', '0'),
                  ('MTIP_DEG2_SIMPLE', '1'), ('MTIP_SHT_FWD_PAIR', '0'),
                  ('MTIP_PROJ_FUSE', '0'), ('MTIP_PROJ_REAL', '0'), ('MTIP_SHT_CHAIN', '0'), ('MTIP_RP_CORR', '0')]


@pytest.mark.parametrize('name,value', _TRAJ_SWITCHES)
def test_switch_short_trajectory_and_transforms(golden_mtip16, golden_ops, name, value, monkeypatch):
    """2 x (3 HIO, SW, 2 ER) with the B_l metric against the oracle, the golden single steps and the transforms at a
    second size, with the alternative kernel forced"""
    monkeypatch.setenv(name, value)
    PC.check_short_trajectory_vs_oracle(golden_mtip16, None, True)
    PC.check_steps_golden(golden_mtip16, None, True)
    PC.check_transforms(32, 8, None, seed=5)


@pytest.mark.parametrize('env', [{'MTIP_JAC_RESIDENT': '0'}, {'MTIP_JAC_TG': '8'}, {'MTIP_PROJ_FUSE': '0'}, {'MTIP_POLAR_ABS_TOL': '1e-14'}])
def test_switch_projection_vs_oracle(env, monkeypatch):
    """polar-factor / projection-GEMM switches at a size with several row slots (2l+1 up to 37)"""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    PC.check_projection_vs_oracle(40, 18)
