"""Drop-in boundary checks (SURVEY section 8 b): the reference's sketches on the operator registry and the GPU-process
look-alike.  ``emul_lib`` = None runs on the real libmtip_hip.so (``-m gpu``), else on the CPU emulation build."""
import numpy as np

from helpers import data_from_golden, golden_settings, rel_l2


def check_reference_sketch_on_registry(golden_mtip16, emul_lib=None):
    """The reference's HIO_ft_stab sketch (reconstruct.py:584-593, MTIP_start 518-528) executed through
    RecipeFactory on the HIP-backed operators == the device-resident step."""
    from xframe_amd.fxs.engine import Engine
    from xframe_amd.fxs.operators import RecipeFactory, build_operators
    g = golden_mtip16
    N, L = int(g['N']), int(g['L'])
    e = Engine(golden_settings(N, L), data_from_golden(g, L), n_batch=1, lib_path=emul_lib, fused=False)
    f = RecipeFactory({})
    ops = build_operators(e)
    f.addOperators(ops)
    f.addOperators({'save_to_dict': lambda *a: None, 'save_number_of_particles': lambda: None,
                    'calc_reciprocal_errors': [lambda a, b, c: {}, 3], 'calc_real_errors': [lambda a, b: {}, 2]})
    mtip_start = [
        [(0, 0), ['copy', 'square_grid']],
        [(0, 1, 1), ['id', 'harmonic_transform', 'copy']],
        [(0, 1, 1, 2), ['id', 'id', 'approximate_unknowns', 'id']],
        [(0, 1, 2, 1, 3), ['id', 'mtip_projection', 'id', 'id']],
        [(0, 1, 2, 3), ['id', 'inverse_harmonic_transform', 'id', 'id']],
        [(0, 0, 3, 1, 2), ['id', 'project_to_modified_intensity', 'save_number_of_particles', 'id']],
        [(0, 1, 2, 1), ['calc_reciprocal_errors', 'id']],
        [(1,), ['id']]]
    f.addOperators({'MTIP_start': f.buildProcessFromSketch(mtip_start)})
    sketch = [
        [(1, 1), ['fourier_transform', 'id']],
        [(0, 0, 0, 1), ['MTIP_start', 'inverse_fourier_transform', 'id']],
        [(0, 2, 1, 2, 0), ['inverse_fourier_transform', 'diff', 'id', 'id']],
        [(0, 1, 2, 3), ['add_above_zero_index', 'id', 'id']],
        [(0, 0, 1, 2), ['copy', 'real_projection', 'id', 'id']],
        [(0, 1, 2, 0, 1, 3), ['hybrid_input_output', 'calc_real_errors', 'id']],
        [(2, 0), ['id', 'id']]]
    proc = f.buildProcessFromSketch(sketch)
    e.set_density(0, g['rho0'])
    e.init_state()
    rho = e.density(0)
    e.hio_beta = 0.45
    F_new, rho_new = proc.run(np.zeros_like(rho), np.array(rho))
    e.run('HIO', True, [0.45])
    assert rel_l2(F_new, e.reciprocal_density(0)) < 1e-10
    assert rel_l2(rho_new, e.density(0)) < 1e-10
    e.close()


def check_gpu_process_boundary(emul_lib=None):
    """ClProcess / add_gpu_process look-alike: the reference's test_GPU contract gpu_func(vects) == matrix @ vects
    (tests/test_framework_integration.py:230-400) and the spherical Hankel kernel_dict of
    hankel_transforms.py:733-759."""
    from oracle import hankel as OH
    from xframe_amd.fxs.gpu_process import ClProcess, _GpuProcessManager
    mgr = _GpuProcessManager()
    rng = np.random.default_rng(0)
    nq, nvec = 12, 5
    matrix = rng.integers(-4, 5, (nq, nq)).astype(float)
    vects = rng.integers(-4, 5, (nq, nvec)).astype(float)
    kd = {'kernel': 'apply_matrix', 'name': 'matmul',
          'functions': ({'name': 'apply_matrix', 'dtypes': (float, float, float, np.int64, np.int64),
                         'shapes': ((nq, nvec), (nq, nq), (nq, nvec), None, None),
                         'arg_roles': ('output', 'const_input', 'input', 'const_input', 'const_input'),
                         'const_inputs': (None, matrix, None, np.int64(nq), np.int64(nvec)),
                         'global_range': (nq, nvec), 'local_range': None},)}
    gpu_func = mgr.add_gpu_process(ClProcess(kd), lib_path=emul_lib)
    assert (gpu_func(vects) == matrix @ vects).all()            # exact on small integers, like the reference test
    L, kappa = 3, 2.0
    w = OH.assemble_weights(OH.spherical_mid_weights(L, nq, kappa), 37.0, kappa)
    nlm = (L + 1) ** 2
    for key in ('forward', 'inverse'):
        kdh = {'kernel': '__kernel void apply_weights(...) { /* OpenCL source is ignored */ }', 'name': key + '_hankel',
               'functions': ({'name': 'apply_weights', 'dtypes': (complex, complex, complex, np.int64, np.int64, np.int64),
                              'shapes': ((nq, nlm), w[key].shape, (nq, nlm), None, None, None),
                              'arg_roles': ('output', 'const_input', 'input', 'const_input', 'const_input', 'const_input'),
                              'const_inputs': (None, w[key], None, np.int64(nq), np.int64(nlm), np.int64(L + 1)),
                              'global_range': (nq, nlm), 'local_range': None},)}
        fn = mgr.add_gpu_process(ClProcess(kdh), lib_path=emul_lib)
        rho = rng.normal(size=(nq, nlm)) + 1j * rng.normal(size=(nq, nlm))
        assert rel_l2(fn(rho), OH.apply_direct(w[key], rho)) < 1e-12


def check_reference_sw_and_shift_sketches_on_registry(golden_mtip16, emul_lib=None):
    """The reference's 'SW' sketch (reconstruct.py:598-605) and 'shift_center' output-modifier sketch (728-734) run
    through RecipeFactory on the registry == the device shrink-wrap / the oracle's output modifier."""
    from oracle import mtip as OM
    from xframe_amd.fxs.engine import Engine
    from xframe_amd.fxs.operators import RecipeFactory, build_operators
    g = golden_mtip16
    N, L = int(g['N']), int(g['L'])
    opt = golden_settings(N, L, {'output_density_modifiers': {'shift_to_center': True}})
    data = data_from_golden(g, L)
    e = Engine(opt, data, n_batch=1, lib_path=emul_lib, fused=True)
    f = RecipeFactory({})
    ops = build_operators(e)
    f.addOperators(ops)
    e.set_density(0, g['rho0'])
    e.init_state()
    e.run('HIO', True, [0.45, 0.45])
    rho, F = e.density(0), e.reciprocal_density(0)
    sw_sketch = [
        'copy',
        ['abs_value', 'copy'],
        [(0, 1), ['fourier_transform', 'id']],
        [(0, 1), ['multiply_ft_gaussian', 'id']],
        [(0, 1), ['inverse_fourier_transform', 'id']],
        [(0, 1), ['calculate_support_mask']]]
    ops['set_shrink_wrap'](sigma=20.0, threshold=0.09)
    support = f.buildProcessFromSketch(sw_sketch).run(np.array(rho))
    support = support[0] if isinstance(support, (tuple, list)) else support
    e.shrinkwrap(20.0, 0.09, np.inf)
    assert (np.asarray(support, bool) != e.support(0)).sum() == 0
    results = {}
    shift_center = [
        [(0, 1, 1), ['copy', 'fourier_transform', 'calc_center']],
        [(0, 1, 2), [['id', np.array([], dtype=object)], ['id', np.array([], dtype=object)],
                     ['save_to_dict', np.array([results, 'neg_center_pos', 'replace'], dtype=object)]]],
        [(0, 2, 1, 2), ['negative_shift', 'negative_shift']],
        [(0, 1), ['id', 'inverse_fourier_transform']]]
    out = f.buildProcessFromSketch(shift_center).run(np.array(F), np.array(rho))
    om = OM.MTIP(opt, data)
    want = om.output_modifier((F, rho))
    assert np.allclose(results['neg_center_pos'], om.results['neg_center_pos'], rtol=1e-8)
    assert rel_l2(out[0], want[0]) < 1e-10 and rel_l2(out[1], want[1]) < 1e-9
    e.close()


def check_gpu_process_from_child_processes(emul_lib=None, n_processes=3):
    """The reference's test_GPU_in_multiprocessing contract (tests/test_framework_integration.py:640-747): a closure run in
    several child processes by Multiprocessing.process_mp_request, each of which registers the kernel and calls the GPU
    function.  Children are fresh interpreters that create their own engine before any other GPU call."""
    from xframe_amd.fxs.gpu_process import process_mp_request
    nq, nvec = 10, 5
    matrix = np.random.default_rng(5).integers(-4, 5, (nq, nq)).astype(float)

    def run_parallel(seed=0, **kwargs):
        import numpy as np
        from xframe_amd.fxs.gpu_process import ClProcess, add_gpu_process
        vects = np.random.default_rng(seed).integers(-4, 5, (nq, nvec)).astype(float)
        kd = {'kernel': 'apply_matrix', 'name': 'gpu_func',
              'functions': ({'name': 'apply_matrix', 'dtypes': (float, float, float, np.int64, np.int64),
                             'shapes': ((nq, nvec), matrix.shape, (nq, nvec), None, None, None),
                             'arg_roles': ('output', 'const_input', 'input', 'const_input', 'const_input'),
                             'const_inputs': (None, matrix, None, np.int64(nq), np.int64(nvec)),
                             'global_range': (nq, nvec), 'local_range': None},)}
        gpu_func = add_gpu_process(ClProcess(kd), lib_path=emul_lib)
        return bool((gpu_func(vects) == matrix @ vects).all())

    out = process_mp_request(run_parallel, n_processes=n_processes, seed=3)
    assert len(out) == n_processes and all(out)
