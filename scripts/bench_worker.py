"""End-to-end timing of the reconstruct worker (what a user of the drop-in sees): ProjectWorker.run() for BASELINE config 4 on one GPU
-- engine creation and uploads, the seeded initial densities, the whole tutorial schedule, and the result dicts back on the host
(seven 16.8 MB grids per restart over PCIe, shift_to_center, the output transforms).  bench.py times the loop with everything
resident in HBM; this is the figure with the host and PCIe in it.   usage: bench_worker.py [restarts [engines]]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from xframe_amd.fxs import reconstruct as R           # noqa: E402
from xframe_amd.fxs import settings as ST             # noqa: E402
from xframe_amd.fxs import synthetic as S             # noqa: E402
from xframe_amd.fxs.engine import Engine              # noqa: E402

n_restarts = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_workers = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = 4
N, L = S._SIZES[cfg]
t0 = time.perf_counter()
eng = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, max_q=S.data_cutoff(N))
data, _ = S.make_invariants(eng, N, L, eigh=eng)
eng.close()
t1 = time.perf_counter()
opt = ST.deep_update(ST.default_settings(), S.config_overrides(cfg))
opt = ST.deep_update(opt, {'grid': {'n_radial_points': N, 'max_order': L},
                           'projections': {'reciprocal': {'used_order_ids': np.arange(L + 1)}},
                           'multi_process': {'use': True, 'n_parallel_reconstructions': n_restarts},
                           'GPU': {'use': True, 'n_gpu_workers': n_workers}})
loops = opt['main_loop']['sub_loops']
n_steps = 0
for name in loops['order']:
    sl = loops[name]
    per = sum(sl['methods'][m]['iterations'] for m in sl['order'] if m in ('HIO', 'ER'))
    n_steps += per * sl['iterations']
w = R.ProjectWorker(opt, data, seeds=[1000 + i for i in range(n_restarts)])
t2 = time.perf_counter()
res, _ = w.run()
t3 = time.perf_counter()
finals = [float(r['error_dict']['main'][-1]) for r in res]
out_bytes = sum(v.nbytes for r in res for v in r.values() if isinstance(v, np.ndarray))
print(f'synthetic invariants (device transforms + eigensolver): {t1 - t0:.2f} s; worker construction {t2 - t1:.2f} s')
print(f'ProjectWorker.run(): {n_restarts} restarts x {n_steps} steps on {n_workers} engines: {t3 - t2:.2f} s wall = '
      f'{n_restarts * n_steps / (t3 - t2):.0f} MTIP iterations/s end to end (uploads, loop, {out_bytes / 1e6:.0f} MB of results back on the host); '
      f'final errors {min(finals):.2e} .. {max(finals):.2e}')
for g in w.results['stats'].get('groups', []):
    print('  engine group of %d restarts: engine %.2f s, initial densities / state %.2f s, loop %.2f s (%.0f it/s), result dicts %.2f s'
          % (g['restarts'], g['engine_seconds'], g['setup_seconds'], g['loop_seconds'], g['iterations_per_second'], g['output_seconds']))
