"""N > 1 path on CPU: two gloo ranks shard the restarts, each runs its batch through the worker (on the CPU
emulation build of the kernels -- there is no GPU here), rank 0 gathers the result dicts; the rotation-invariant
B_l are all-reduced.  The same code path runs with backend nccl (= RCCL) on the MI355X node."""
import os
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
EMUL_DIR = os.path.join(HERE, 'emul')

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
np.seterr(all='ignore')
import torch.distributed as dist
from helpers import data_from_golden, golden_settings
from xframe_amd.fxs import reconstruct as R, parallel as P
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
g = np.load(os.path.join({here!r}, 'golden', 'mtip_N16_L4.npz'))
N, L = int(g['N']), int(g['L'])
n_total, n_eng, n_full = {n_total}, {n_eng}, {n_full}
opt = golden_settings(N, L, {{'multi_process': {{'use': True, 'n_parallel_reconstructions': n_total}},
                              'GPU': {{'use': True, 'n_gpu_workers': n_eng}}}})
main = opt['main_loop']['sub_loops']['main']
main['methods']['HIO']['iterations'] = 2; main['methods']['ER']['iterations'] = 1; main['iterations'] = 1
w = R.ProjectWorker(opt, data_from_golden(g, L), seeds=[11 + i for i in range(n_total)], n_gather_full=n_full,
                    lib_path=os.path.join({emul!r}, 'libmtip_emul.so'))        # rank / world size from the launcher's environment
result, _ = w.run()
mine = P.shard_restarts(n_total, rank, world)
own = [result[i] for i in mine] if rank == 0 else list(result)     # rank 0 holds all restarts, others only theirs
allv = P.gather_scalars(np.array([float(len(mine)), float(rank)]))
bl_sum = sum(r['last_deg2_invariant'] for r in own)
mean_bl = P.average_invariants(bl_sum, len(own))
out = {{'rank': rank, 'n_results': int(len(result)), 'mine': mine,
       'gathered': allv.tolist(), 'bl_trace': float(np.trace(mean_bl[0]).real),
       'errs': [float(r['final_error']) for r in result], 'engines': len(w.mtip_instances),
       'kinds': [r.get('gathered', 'own') for r in result],
       'dens_norm': [float(np.linalg.norm(r['real_density'])) if 'real_density' in r else -1.0 for r in result],
       # order-sensitive checksums of every grid-sized array of a full dict: what arrived on rank 0 must be what the owner holds
       'sums': [[float(np.vdot(np.arange(1, r[k].size + 1) % 97, np.asarray(r[k], dtype=complex).ravel()).real) for k in
                 ('real_density', 'last_real_density', 'reciprocal_density', 'last_reciprocal_density', 'support_mask', 'last_support_mask',
                  'initial_density', 'last_deg2_invariant')] if 'real_density' in r else [] for r in result],
       'from_hbm': int(getattr(P, 'SENT_FROM_DEVICE', 0)),
       'sorted': [int(i) for i in w.results.get('sorted_ids', [])]}}
print('RESULT ' + json.dumps(out), flush=True)
dist.destroy_process_group()
'''


@pytest.fixture(scope='module')
def emul_lib():
    r = subprocess.run(['make', '-C', EMUL_DIR, '-j6'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_shard_restarts_round_robin():
    from xframe_amd.fxs.parallel import shard_restarts
    assert shard_restarts(64, 3, 8) == list(range(3, 64, 8))
    assert sorted(sum((shard_restarts(10, r, 4) for r in range(4)), [])) == list(range(10))
    assert shard_restarts(2, 3, 4) == []


def _run_two_ranks(tmp_path, port, **fmt):
    import json
    script = tmp_path / 'worker.py'
    script.write_text(WORKER.format(root=ROOT, here=HERE, emul=EMUL_DIR, **fmt))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE='2', MTIP_EMUL_THREADS='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith('RESULT ')][0][7:]))
    r0 = [o for o in outs if o['rank'] == 0][0]
    r1 = [o for o in outs if o['rank'] == 1][0]
    return r0, r1


def test_two_rank_gloo_worker(emul_lib, tmp_path):
    r0, r1 = _run_two_ranks(tmp_path, 29571, n_total=3, n_eng=1, n_full=8)
    assert r0['mine'] == [0, 2] and r1['mine'] == [1]
    assert r0['kinds'] == ['own', 'full', 'own'] and r0['dens_norm'][1] > 0     # restart 1 arrived with its arrays
    assert r0['sums'][1] == r1['sums'][0]            # ... bit for bit what its owner holds
    assert r1['from_hbm'] == 6 and r0['from_hbm'] == 0    # densities and masks went out of the owner's engine buffers, not its host copies
    assert r0['n_results'] == 3                      # rank 0 holds every restart after the gather
    assert r0['gathered'] == [[2.0, 0.0], [1.0, 1.0]] == r1['gathered']
    assert np.isclose(r0['bl_trace'], r1['bl_trace'], rtol=1e-12)       # all-reduced mean B_l identical on both ranks
    assert len(r0['errs']) == 3 and all(np.isfinite(r0['errs']))
    assert sorted(r0['sorted']) == [0, 1, 2]
    assert r0['errs'][r0['sorted'][0]] == min(r0['errs'])
    assert r1['n_results'] == 1


def test_two_ranks_two_engines_best_only(emul_lib, tmp_path):
    """world_size 2 x GPU.n_gpu_workers 2 (four restarts per rank on two engines each); only the best restart travels in full:
    the other remote restarts reach rank 0 as light dicts (scalars and error histories, no grid arrays)."""
    r0, r1 = _run_two_ranks(tmp_path, 29573, n_total=8, n_eng=2, n_full=1)
    assert r0['engines'] == 2 and r1['engines'] == 2
    assert r0['n_results'] == 8 and r1['n_results'] == 4
    errs = np.array(r0['errs'])
    assert np.isfinite(errs).all() and len(set(r0['errs'])) == 8
    best = int(np.argmin(errs))
    for i, kind in enumerate(r0['kinds']):
        if i % 2 == 0:
            assert kind == 'own' and r0['dens_norm'][i] > 0
        elif i == best:
            assert kind == 'full' and r0['dens_norm'][i] > 0
        else:
            assert kind == 'light' and r0['dens_norm'][i] == -1.0
    assert np.isclose(r0['bl_trace'], r1['bl_trace'], rtol=1e-12)


WORKER_2D = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {here!r})
np.seterr(all='ignore')
import torch.distributed as dist
import parity_cases as PC
from oracle import mtip as OM
from xframe_amd.fxs import reconstruct as R
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
g = np.load(os.path.join({here!r}, 'golden', 'mtip2d_N12_M6.npz'))
data, o = PC.mtip2d_problem(g)
o = OM.deep_update(o, {{'multi_process': {{'use': True, 'n_parallel_reconstructions': 3}}, 'GPU': {{'use': True, 'n_gpu_workers': 1}}}})
w = R.ProjectWorker(o, data, seeds=[5, 6, 7], lib_path=os.path.join({emul!r}, 'libmtip_emul.so'))
result, _ = w.run()
out = {{'rank': rank, 'n_results': int(len(result)), 'errs': [float(r['final_error']) for r in result],
       'kinds': [r.get('gathered', 'own') for r in result], 'shapes': [list(np.shape(r.get('real_density', []))) for r in result]}}
print('RESULT ' + json.dumps(out), flush=True)
dist.destroy_process_group()
'''


def test_two_rank_gloo_worker_2d(emul_lib, tmp_path):
    """`dimensions: 2` through the same sharding and gather: three restarts over two ranks, rank 0 ends with all of them, equal to a
    single-process run of the same seeds"""
    import json
    import parity_cases as PC
    from oracle import mtip as OM
    from xframe_amd.fxs import reconstruct as R
    script = tmp_path / 'worker2d.py'
    script.write_text(WORKER_2D.format(root=ROOT, here=HERE, emul=EMUL_DIR))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', WORLD_SIZE='2', MTIP_EMUL_THREADS='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=600)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([l for l in o.splitlines() if l.startswith('RESULT ')][0][7:]))
    r0 = [o for o in outs if o['rank'] == 0][0]
    assert r0['n_results'] == 3 and r0['kinds'] == ['own', 'full', 'own'] and r0['shapes'] == [[12, 13]] * 3
    g = np.load(os.path.join(HERE, 'golden', 'mtip2d_N12_M6.npz'))
    data, o = PC.mtip2d_problem(g)
    o = OM.deep_update(o, {'multi_process': {'use': True, 'n_parallel_reconstructions': 3}, 'GPU': {'use': True, 'n_gpu_workers': 1}})
    w = R.ProjectWorker(o, data, seeds=[5, 6, 7], rank=0, world_size=1, lib_path=os.path.join(EMUL_DIR, 'libmtip_emul.so'))
    single, _ = w.run()
    assert np.allclose(r0['errs'], [float(r['final_error']) for r in single], rtol=1e-12)
