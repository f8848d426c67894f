"""Multi-rank averaging on CPU: two gloo ranks hold different reconstructions, the reference is broadcast by its owner, each
rank aligns its own restarts on the CPU emulation build, the aligned sums are all-reduced -- and the result equals the
single-process average of all reconstructions.  The same code runs with backend nccl (= RCCL) on the MI355X node."""
import json
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
from oracle import alignment as OA
from oracle.fourier import FourierPair
from oracle.sht import SHT
from xframe_amd.fxs import average as AV, synthetic as S
from xframe_amd.fxs.engine import Engine
rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
if world > 1:
    dist.init_process_group('gloo', rank=rank, world_size=world)
N, L, n_rec = 10, 5, 4
max_q = float(np.max(S.midpoint_points(S.data_cutoff(N), N)))
fp = FourierPair(SHT(L), N, max_q, 2.0)
sht = fp.sht
rng = np.random.default_rng(7)
c = (rng.normal(size=(N, (L + 1) ** 2)) + 1j * rng.normal(size=(N, (L + 1) ** 2))) * np.exp(-(np.arange(N)[:, None] / (0.35 * N)) ** 2)
c[:, 1:4] = 0
base = sht.inverse_d(c).real
base = ((base - base.min() + 0.05) * np.exp(-(fp.rs[:, None, None] / (0.5 * fp.rs.max())) ** 4)).astype(complex)
al, be, ga = OA.euler_grid(L + 1)
recs, errs = [], []
for i in range(n_rec):
    euler = np.array([al[(3 * i) % len(al)], be[(2 * i + 1) % len(be)], ga[(5 * i) % len(ga)]]) if i != 1 else np.zeros(3)
    d = (1 + 0.2 * i) * sht.inverse_d(OA.rotate_coeff(sht.forward_d(base), euler, L))
    recs.append((d, fp.ft(d)))
    errs.append(0.01 * (1 + ((i + 3) % n_rec)))
mine = list(range(rank, n_rec, world))
e = Engine({{'grid': {{'n_radial_points': N, 'max_order': L}}}}, None, n_batch=2, max_q=max_q,
           lib_path=os.path.join({emul!r}, 'libmtip_emul.so'))
opt = {{'alignment_error_limit': 0.5, 'find_rotation': {{'r_limit_ids': [0, N]}}, 'center_reconstructions': False}}
res = AV.average_reconstructions(e, [recs[i] for i in mine], [errs[i] for i in mine], opt, dist=dist if world > 1 else None)
out = {{'rank': rank, 'n': res['n_averaged'], 'owner': res['reference_owner'],
       'avg': [float(np.linalg.norm(res['average']['real_density'])), float(np.abs(res['average']['real_density']).sum())],
       'prtf': [float(x) for x in res['resolution_metrics']['PRTF'].real[:4]]}}
print('RESULT ' + json.dumps(out), flush=True)
if world > 1:
    dist.destroy_process_group()
'''


@pytest.fixture(scope='module')
def emul_lib():
    r = subprocess.run(['make', '-C', EMUL_DIR, '-j6'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _run(tmp_path, world, port):
    script = tmp_path / f'avg_worker_{world}.py'
    script.write_text(WORKER.format(root=ROOT, here=HERE, emul=EMUL_DIR))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(world), MTIP_EMUL_THREADS='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for r in range(world)]
    outs = []
    for p in procs:
        o, e = p.communicate(timeout=900)
        assert p.returncode == 0, e[-3000:]
        outs.append(json.loads([ln for ln in o.splitlines() if ln.startswith('RESULT ')][0][7:]))
    return sorted(outs, key=lambda x: x['rank'])


def test_two_rank_average_equals_single_process(emul_lib, tmp_path):
    single = _run(tmp_path, 1, 29581)[0]
    r0, r1 = _run(tmp_path, 2, 29583)
    assert r0['owner'] == r1['owner'] == 1                      # restart 1 has the lowest error; rank 1 owns it
    assert r0['n'] == r1['n'] == single['n']
    assert np.allclose(r0['avg'], r1['avg'], rtol=1e-13)        # all-reduced: identical on both ranks
    # processing order (rank by rank, restart by restart) is 0, 2, 3 in both runs, so the reference's selection rule (the last
    # valid alignment in processing order is dropped) picks the same set and the averages must agree
    assert np.allclose(r0['avg'], single['avg'], rtol=1e-10)
    assert np.allclose(r0['prtf'], single['prtf'], rtol=1e-8)
