"""Timing of the 2-D (polar) phasing loop on the GPU next to the oracle on one host core (a study script: imports the oracle).
    python tests/tools/bench_2d_loop.py [N M B steps]"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, '..', '..'))
sys.path.insert(0, os.path.join(HERE, '..'))
import parity_cases as PC                                                     # noqa: E402
from oracle import mtip2d as O2                                               # noqa: E402
from xframe_amd.fxs.reconstruct2d import MTIP2D                               # noqa: E402

N, M, B, steps = (int(a) for a in (sys.argv[1:5] + ['128', '64', '8', '200'][len(sys.argv) - 1:]))
g = np.load(os.path.join(HERE, '..', 'golden', 'mtip2d_N12_M6.npz'))
data, o = PC.mtip2d_scaled_problem(g, N, M)
m = MTIP2D(o, data, n_restarts=B, seeds=list(range(B)))
e = m.engine
rho = np.stack([m._initial_density(b) for b in range(B)])
sup = np.broadcast_to(m.initial_support, (B,) + m.shape)
for method, stab in (('HIO', True), ('ER', True), ('HIO', False)):
    for _ in range(5):
        e.step(method, stab, 0.45, rho, sup)
    t = time.perf_counter()
    r = rho
    for _ in range(steps):
        _, r, _, _ = e.step(method, stab, 0.45, r, sup)
    dt = time.perf_counter() - t
    print(f'device  {method:3s} ft_stab={int(stab)}  {N} x M{M}, {B} restarts per call: {dt / steps * 1e3:7.3f} ms per call, {B * steps / dt:9.0f} steps/s (host arrays in and out each call)')
t = time.perf_counter()
for _ in range(20):
    e.shrinkwrap(rho, 20.0, 0.09)
print(f'device  shrink-wrap: {(time.perf_counter() - t) / 20 * 1e3:7.3f} ms per call')
om = O2.MTIP2D(o, data)
om.real_pr.support = m.initial_support
om.beta = 0.45
om.errors = {'real': {'l2_projection_diff': []}, 'reciprocal': {}, 'main': []}
r = np.array(rho[0])
n_cpu = max(3, steps // 20)
t = time.perf_counter()
for _ in range(n_cpu):
    _, r = om.step('HIO', r, True)
dt = time.perf_counter() - t
print(f'oracle  HIO ft_stab=1  one restart, one core: {dt / n_cpu * 1e3:7.3f} ms per step, {n_cpu / dt:9.1f} steps/s')
