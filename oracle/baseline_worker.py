"""CPU baseline leg of bench.py (TEST / MEASUREMENT INFRASTRUCTURE, never on the product path): one oracle process per
restart, BLAS pinned to one thread, as the reference runs its CPU path (xframe/__init__.py:5-8 pins the thread pools,
reconstruct.py:141-157 forks one process per reconstruction).  Each process times HIO ft_stab steps of its own restart
for a bounded number of seconds on the invariants bench.py hands over; nothing here touches a GPU."""
import os
import time


def run(args):
    """args = (data dict or None, config id, seed, seconds, max_steps) -> (steps done, loop seconds, setup seconds)"""
    for k in ('OMP_NUM_THREADS', 'OPENBLAS_NUM_THREADS', 'MKL_NUM_THREADS'):
        os.environ[k] = '1'
    import numpy as np
    np.seterr(all='ignore')
    try:
        import threadpoolctl
        limiter = threadpoolctl.threadpool_limits(1)
    except Exception:                                    # pragma: no cover - optional dependency
        limiter = None
    from oracle import mtip as OM
    from xframe_amd.fxs import synthetic as S             # pure numpy module (settings of the BASELINE configs)
    data, cfg, seed, seconds, max_steps = args
    t0 = time.perf_counter()
    if data is None:                                     # the same synthetic invariants, made with the oracle's transforms
        from oracle.fourier import FourierPair
        from oracle.sht import SHT

        class _T:
            def __init__(s, fp):
                s.fp, s.rs, s.thetas, s.phis = fp, fp.rs, fp.sht.theta, fp.sht.phi

            def ft(s, x):
                return s.fp.ft(x)

            def forward_l(s, x):
                return s.fp.sht.forward_l(x)

            def hermitian_eig(s, mats):                      # numpy eigensolver in Engine.hermitian_eig's layout (descending, columns)
                w, v = np.linalg.eigh(np.asarray(mats))
                return w[:, ::-1].copy(), np.ascontiguousarray(v[:, :, ::-1])
        N, L = S._SIZES[cfg]
        data, _ = S.make_invariants(_T(FourierPair(SHT(L), N, S.data_cutoff(N), 2.0)), N, L)
    opt = OM.deep_update(OM.default_settings(), S.config_overrides(cfg))
    om = OM.MTIP(opt, data)
    rho0 = om.density_guess(np.random.default_rng(seed))
    state = om.create_initial_state(rho0)
    rho = state['density_pair_history'][-1][1]
    om.beta = 0.45
    t1 = time.perf_counter()
    n = 0
    while True:
        _, rho = om.step('HIO', rho, True)
        n += 1
        if time.perf_counter() - t1 > seconds or n >= max_steps:
            break
    t2 = time.perf_counter()
    del limiter
    return n, t2 - t1, t1 - t0
