#!/usr/bin/env python3
"""MTIP phasing benchmark: iterations/sec at 128 q-shells x L_max = 32 (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (for N > 1 launched by torch.distributed.run; RANK / LOCAL_RANK / WORLD_SIZE from the
environment).  Every rank phases `--restarts-per-gpu` independent restarts (BASELINE config 4: 64 restarts
over 8 GPUs = 8 per GPU), device resident; restarts never communicate, so scaling is weak and there is no
data-path collective -- RCCL is only used for the timing barrier / max-over-ranks and for the end-of-run
reduce of the rotation-invariant B_l (outside the timed region, reported separately).
The restarts of a rank are split over `--streams` engines (one mtip_ctx + HIP stream each) so that the
latency-bound polar-factor kernel of one group overlaps the bandwidth-bound transforms of the others.

A "step" = one phasing step (HIO or ER sketch incl. ft_stab, reconstruct.py:576-593) of every restart of the
rank; the K timed steps walk the tutorial schedule (60 HIO, 1 SW, 40 ER, ...; tutorial.yaml:52-72), shrink
wrap updates are executed and timed but not counted.  value = N * restarts_per_gpu * K / seconds.
Inputs (synthetic invariants, initial densities) are resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# HIP maps streams onto 4 hardware queues by default (one is taken by the null stream): give every engine its own
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument('--gpus', type=int, default=1)
    p.add_argument('--steps', type=int, default=600, help='timed steps; 600 = one pass of the tutorial schedule (SURVEY 8d, configs 3/4)')
    p.add_argument('--warmup', type=int, default=20)
    p.add_argument('--restarts-per-gpu', type=int, default=8)
    p.add_argument('--streams', type=int, default=3, help='engines (HIP streams) the restarts of a rank are split over')
    p.add_argument('--engine-sizes', default='', help="restarts per engine, e.g. '4,2,2' (overrides the even split of --streams)")
    p.add_argument('--config', type=int, default=4, help='BASELINE config id (sizes): 1..5')
    p.add_argument('--no-turns', action='store_true', help='enqueue every engine on its own instead of mtip_run_group_async')
    p.add_argument('--exact', action='store_true', help='reference operator order instead of the fused step')
    p.add_argument('--cpu-seconds', type=float, default=20.0, help='budget of the CPU baseline sample')
    p.add_argument('--no-cpu-baseline', action='store_true')
    p.add_argument('--no-roofline', action='store_true')
    p.add_argument('--repeats', type=int, default=0,
                   help='timed windows (each: fresh state, W warm-up steps, K timed steps); the line reports the MEDIAN window. '
                        '0 = 5 when K <= 50 (a 20-step window lasts 13 ms: one shot is not a measurement), 3 up to 1000 steps, else 1')
    p.add_argument('--dist-backend', default='nccl', help="'gloo' + --same-device rehearses the multi-rank path on a one-GPU box")
    p.add_argument('--same-device', action='store_true', help='every rank uses GPU 0 (rehearsal only)')
    return p.parse_args()


def schedule(n_steps):
    """walk the tutorial schedule: 5 x (60 HIO, SW, 40 ER) + 1 x (SW, 100 ER), repeated, truncated."""
    out = []
    blocks = [('HIO', 60), ('SW', 1), ('ER', 40)] * 5 + [('SW', 1), ('ER', 100)]
    done = 0
    while done < n_steps:
        for kind, n in blocks:
            if done >= n_steps:
                break
            if kind == 'SW':
                if out:                       # a SW needs an error history; skip a leading one
                    out.append(('SW', 0))
                continue
            k = min(n, n_steps - done)
            out.append((kind, k))
            done += k
    return out


def algorithmic_bytes_per_step(N, L, nt, nphi, ft_stab=True):
    """SURVEY section 8 d / BASELINE.md section 4."""
    G = N * nt * nphi
    nlm = (L + 1) ** 2
    n_grid, n_coef, n_hankel = (12, 17, 3) if ft_stab else (8, 13, 2)
    return 16 * G * n_grid + G + 16 * N * nlm * n_coef + 8 * N * N * (L + 1) * n_hankel


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a fresh child (torch.distributed.run) BEFORE
    anything in this process touches HIP, wait for it and pass its exit code on (never exec from a GPU process)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={a.gpus}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd).returncode


def phase_of(warmup, steps):
    """which steps of the tutorial schedule the timed region covers (the polar factor of an HIO step is the more
    expensive one): the warm-up walks the first `warmup` steps of the schedule, the timed region then walks the
    schedule again from its beginning."""
    kinds = {'HIO': 0, 'ER': 0, 'SW': 0}
    for kind, k in schedule(steps):
        kinds[kind] += k if kind != 'SW' else 1
    return {'warmup_steps': warmup, 'timed_steps': steps, 'timed_HIO_steps': kinds['HIO'], 'timed_ER_steps': kinds['ER'],
            'timed_SW_updates': kinds['SW'],
            'schedule': 'tutorial schedule from its first step: 5 x (60 HIO, SW, 40 ER) + (SW, 100 ER), repeated'}


def cpu_baseline(a, N, L, B):
    """The oracle (numpy restatement of the reference algorithm), one process per restart with one BLAS thread each like the
    reference (xframe/__init__.py:5-8, reconstruct.py:141-157), a bounded sample of HIO ft_stab steps.  Runs BEFORE this
    process touches the GPU: the children are fresh interpreters (spawn) and build the same synthetic invariants with the
    oracle's own transforms."""
    import multiprocessing as mp
    from oracle import baseline_worker
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    n_proc = max(1, min(B, cores))
    ctx = mp.get_context('spawn')
    t_c0 = time.perf_counter()
    with ctx.Pool(n_proc) as pool:
        outs = pool.map(baseline_worker.run, [(None, a.config, 1000 + i, a.cpu_seconds, 200) for i in range(n_proc)])
    t_c1 = time.perf_counter()
    per_proc = [n / sec for n, sec, _ in outs]
    return {'value': float(sum(per_proc)), 'unit': 'MTIP iterations/s', 'cores': n_proc, 'kind': 'port',
            'per_process': float(np.mean(per_proc)), 'host_cpu_count': os.cpu_count(), 'usable_cores': cores,
            'sample': f'{n_proc} oracle processes (one restart each, 1 BLAS thread, as reconstruct.py:141-157), '
                      f'{sum(n for n, _, _ in outs)} HIO ft_stab steps in total at {N}x L{L} on the same synthetic invariants, '
                      f'{a.cpu_seconds:.0f} s of stepping per process; setup ({np.mean([st for _, _, st in outs]):.1f} s per '
                      f'process) excluded; wall {t_c1 - t_c0:.0f} s'}


def main():
    a = parse()
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(spawn_ranks(a))
    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if a.gpus != world:
        raise SystemExit(f'--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus} '
                         f'(or run `python bench.py --gpus {a.gpus}` without a launcher)')
    np.seterr(all='ignore')
    from xframe_amd.fxs import synthetic as S0
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:       # reported at N = 1 only; before any GPU call of this process
        cpu = cpu_baseline(a, *S0._SIZES[a.config], a.restarts_per_gpu)
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.same_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if a.dist_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(a.dist_backend)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    dev = torch.device('cuda', local_rank)

    from xframe_amd.fxs import hostsetup as hs
    from xframe_amd.fxs import synthetic as S
    from xframe_amd.fxs.engine import Engine
    from xframe_amd.fxs.parallel import average_invariants

    N, L = S._SIZES[a.config]
    B = a.restarts_per_gpu
    n_eng = max(1, min(a.streams, B))
    opt = S.config_overrides(a.config)
    # ---- synthetic inputs (product path: HIP transforms), identical on every rank
    t_setup = time.time()
    eng_d = Engine({'grid': {'n_radial_points': N, 'max_order': L}}, None, n_batch=1, device=local_rank,
                   max_q=S.data_cutoff(N))
    data, rho_true = S.make_invariants(eng_d, N, L, eigh=eng_d)     # simulate + extract front half on the device
    eng_d.close()
    sizes = [B // n_eng + (1 if i < B % n_eng else 0) for i in range(n_eng)]
    if a.engine_sizes:
        sizes = [int(x) for x in a.engine_sizes.split(',')]
        if sum(sizes) != B or min(sizes) < 1:
            raise SystemExit(f'--engine-sizes {a.engine_sizes} must add up to --restarts-per-gpu {B}')
        n_eng = len(sizes)
    engines = [Engine(opt, data, n_batch=nb, device=local_rank, fused=not a.exact) for nb in sizes]
    e0 = engines[0]
    rho0 = []
    gid = rank * B
    for e in engines:
        for b in range(e.B):                                    # global restart id -> seed 1000 + id
            rho0.append(hs.bump_density(e.rs, e.shape, S.PARTICLE_RADIUS, 0.3, 2, np.random.default_rng(1000 + gid),
                                        e.rsetup.integrated_intensity, e.int_wr, e.int_wt))
            gid += 1

    def fresh_state():
        """every restart back to its seeded initial density and the initial support (reconstruct.py:957-979): the timed windows of a
        run all walk the same trajectory"""
        i = 0
        for e in engines:
            e.reset_support()
            for b in range(e.B):
                e.set_density(b, rho0[i])
                i += 1
            e.init_state()
        for e in engines:
            e.synchronize()

    fresh_state()
    setup_s = time.time() - t_setup
    ramp = hs.ExponentialRamp(0.5, 0.4, -1 / 250, 500)
    limit = 6e-3
    sw_sigma = hs.LinearRamp(20, [False, 5], -2, default_start=e0.default_sigma, default_stop=e0.default_sigma)
    CHUNK = 10                                                  # steps enqueued per engine before switching
    BRACKET_CHUNKS = int(os.environ.get('BENCH_BRACKET_CHUNKS', '4'))  # bracketed steps per bracketed window

    host = {'enqueue_s': 0.0, 'profile': False, 'chunks': 0}

    # the engines' steps go through mtip_run_group_async, as ProjectWorker's restart groups do (EngineGroup): the contexts take
    # turns at the chip-filling transforms; --no-turns enqueues every engine on its own (what rounds 1-3 timed)
    from xframe_amd.fxs.engine import EngineGroup
    turns = None if (a.no_turns or len(engines) == 1) else EngineGroup(engines)

    def run_all(kind, betas):
        if turns is not None:
            turns.run(kind, True, betas)
        else:
            for e in engines:
                e.run(kind, True, betas, fetch=False)

    def run_schedule(n_steps, start_step=0):
        step = start_step
        sw_count = 0
        for kind, k in schedule(n_steps):
            if kind == 'SW':
                for e in engines:
                    e.shrinkwrap(sw_sigma(sw_count), 0.09, limit)
                sw_count += 1
                continue
            done = 0
            while done < k:
                c = min(CHUNK, k - done)
                betas = np.array([ramp.eval(step + done + i) for i in range(c)])
                th = time.perf_counter()
                # family timers (hipEvent brackets on engine 0's stream): the first step of the first BRACKET_CHUNKS chunks of a
                # bracketed window.  A bracketed step costs ~0.5 ms -- eighteen event records drain engine 0's queue between its
                # kernels and the other engines run ahead (measured: one bracketed step per 20-step window = 5 % of the window) --
                # so the brackets are a SAMPLE inside the timed region, in the first windows only (the line says which), and the
                # reported time is the median window
                if host['profile'] and c > 1 and host['chunks'] < BRACKET_CHUNKS:
                    engines[0].lib.mtip_profile(engines[0].ctx, 1)
                    run_all(kind, betas[:1])
                    engines[0].lib.mtip_profile(engines[0].ctx, 0)
                    run_all(kind, betas[1:])
                    host['chunks'] += 1
                else:
                    host['chunks'] += 1 if host['profile'] else 0
                    run_all(kind, betas)
                host['enqueue_s'] += time.perf_counter() - th
                done += c
            step += k
        return step

    def sync_all():
        for e in engines:
            e.synchronize()
        torch.cuda.synchronize(dev)

    from xframe_amd.fxs.engine import streams_side_by_side
    side_by_side = streams_side_by_side(engines)              # < n_eng: two engines share a hardware queue
    # diagnostic: K more streams of this process that have run one kernel and then sit idle during the timed region (as the
    # streams of a communication library do): does a hardware queue that is merely there cost the engines anything?
    idle_streams = []
    for _ in range(int(os.environ.get('BENCH_IDLE_STREAMS', '0'))):
        st = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(st):
            torch.zeros(16, device=dev).add_(1.0)
        idle_streams.append(st)
    torch.cuda.synchronize(dev)

    # ---- R windows of (fresh state, W warm-up steps, exactly K timed steps between barrier + synchronize); the line reports the
    #      median window, so ms_per_step x steps is a time that was measured, and lists them all
    R = a.repeats if a.repeats > 0 else (5 if a.steps <= 50 else (3 if a.steps <= 1000 else 1))
    windows = []
    enqueue = []
    n_bracketed = max(1, R // 2)                                # windows that carry family brackets: fewer than half, so the median has none
    for rep in range(R):
        if rep > 0:
            fresh_state()
        host['profile'] = False
        run_schedule(a.warmup)
        sync_all()
        if dist is not None:
            dist.barrier()
        host['enqueue_s'] = 0.0
        host['chunks'] = 0
        if not a.no_roofline:
            if rep == 0:
                e0.profile(True)                                # resets the timers; they accumulate over the bracketed windows
                e0.lib.mtip_profile(e0.ctx, 0)
            host['profile'] = rep < n_bracketed
        t0 = time.perf_counter()
        run_schedule(a.steps, start_step=a.warmup)
        sync_all()
        t1 = time.perf_counter()
        if dist is not None:
            dist.barrier()
        el = t1 - t0
        if dist is not None:
            t = torch.tensor([el], dtype=torch.float64, device=dev if a.dist_backend == 'nccl' else 'cpu')
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        windows.append(el)
        enqueue.append(host['enqueue_s'])
        if not a.no_roofline:
            engines[0].lib.mtip_profile(engines[0].ctx, 0)
    order = np.argsort(windows)
    mid = int(order[(R - 1) // 2])                              # the median window (the lower one of an even count)
    elapsed = windows[mid]
    host['enqueue_s'] = enqueue[mid]
    its = world * B * a.steps / elapsed
    ms_per_step = 1e3 * elapsed / a.steps

    # ---- end-of-run reduce of rotation-invariant summaries (outside the timed region)
    t_red = time.perf_counter()
    best_err = np.concatenate([e.best_error()[0] for e in engines])
    n_done = e0.best_error()[1]
    sweeps = e0.jacobi_sweeps()
    closing = e0.jacobi_closing_step()
    bl_sum = np.zeros((L + 1, N, N), complex)
    n_bl = 0
    for e in engines[:2]:                                       # bounded sample: the download is PCIe bound
        bl_sum += e.last_deg2_invariant(0)
        n_bl += 1
    bl_mean = average_invariants(bl_sum, n_bl, device=dev if (dist is not None and a.dist_backend == 'nccl') else None)
    reduce_s = time.perf_counter() - t_red

    # ---- roofline of the dominant kernel family: hipEvent brackets on engine 0's stream, recorded over the timed
    #      region itself (asynchronous event pairs, resolved after the final synchronisation)
    roofline = None
    fam_ms = {}
    Bp = e0.B
    e0_cus = int(torch.cuda.get_device_properties(dev).multi_processor_count)
    if not a.no_roofline:
        for fam in ('sht_fwd', 'sht_inv', 'sht_inv_modulus', 'sht_inv_real', 'sht_chain', 'sht_chain_modulus', 'sht_chain_real', 'hankel',
                    'proj', 'polar', 'real_update', 'deg2_metric'):
            ms, n = e0.profile_get(fam)
            if n:
                fam_ms[fam] = {'total_ms': ms, 'launches': int(n), 'avg_ms': ms / n}
        e0.profile(False)
        if fam_ms:
            G = N * e0.n_theta * e0.n_phi
            C = N * (L + 1) ** 2
            # algorithmic bytes per launch (SURVEY section 8 d: 16 B per grid point / coefficient moved once,
            # 1 B per mask byte), Bp restarts per launch
            alg = {'sht_fwd': (16 * G + 16 * C) * Bp,                       # grid in, coefficients out
                   'sht_inv': (16 * C + 16 * G) * Bp,                       # coefficients in, grid out
                   'sht_inv_modulus': (16 * C + 2 * 16 * G) * Bp,           # + F in  (F' = F sqrt(I'/I) epilogue)
                   'sht_inv_real': (2 * 16 * C + 2 * 16 * G + 2 * G) * Bp,  # two coefficient sets, rho in/out, 2 masks
                   # chained inverse -> forward kernels (k_sht_chain.hip): coefficients in, the grid once, coefficients out
                   'sht_chain': (2 * 16 * C + 16 * G) * Bp,                 # F out
                   'sht_chain_modulus': (2 * 16 * C + 2 * 16 * G) * Bp,     # F in, F' out
                   'sht_chain_real': (2 * 16 * C + 2 * 16 * G + G // 4) * Bp,   # rho in / out, packed masks (2 bits per point)
                   'hankel': 2 * 16 * C * Bp + 8 * N * N * (L + 1),
                   'real_update': (3 * 16 + 2) * G * Bp}
            hbm = {k: v for k, v in fam_ms.items() if k in alg}
            dom = max(hbm, key=lambda k: hbm[k]['total_ms'])
            achieved = alg[dom] / (fam_ms[dom]['avg_ms'] * 1e-3) / 1e9
            traffic = None
            try:                                                        # PMC pass of the same command, committed
                with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')) as f:
                    pmc = json.load(f)
                if pmc.get('restarts_per_launch') == Bp and pmc.get('config') == a.config:
                    traffic = pmc['hbm_bytes_per_launch'].get(dom)
            except (OSError, ValueError, KeyError):
                traffic = None
            hbm_roof = {'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': 8000.0, 'unit': 'GB/s',
                        'frac': achieved / 8000.0, 'traffic': traffic, 'avg_launch_ms': fam_ms[dom]['avg_ms'],
                        'algorithmic_bytes_per_launch': alg[dom], 'restarts_per_launch': Bp,
                        'families_GBps': {k: alg[k] / (fam_ms[k]['avg_ms'] * 1e-3) / 1e9 for k in hbm}}
            # the kernel with the largest share of the step.  The polar factor ("polar", nested in the "proj" bracket) is
            # not an HBM / MFMA kernel: one workgroup = one CU per (restart, order) matrix, bound by FP64 vector issue and
            # the latency of its dependent pivot chain; its roofline is the FP64 vector rate of the CUs it occupies.
            tops = {k: v for k, v in fam_ms.items() if k != 'proj'}
            if 'polar' in fam_ms and 'proj' in fam_ms:                          # what is left of the projection: its GEMMs
                tops['proj_gemms'] = {'total_ms': fam_ms['proj']['total_ms'] - fam_ms['polar']['total_ms']}
            top = max(tops, key=lambda k: tops[k]['total_ms'])
            if top == 'polar':
                polar_traffic = None
                try:
                    if pmc.get('restarts_per_launch') == Bp and pmc.get('config') == a.config:
                        polar_traffic = pmc['hbm_bytes_per_launch'].get('polar')
                except (NameError, KeyError, AttributeError):
                    polar_traffic = None
                swp = np.asarray(e0.jacobi_sweeps(), dtype=float)                  # (Bp, L+1): sweeps of the last call
                ns = 2 * np.arange(L + 1) + 1
                active = swp.mean(0) > 0
                real = (os.environ.get('MTIP_PROJ_REAL', '1') != '0'
                        and all(np.all(np.asarray(p).imag == 0) for p in data['data_projection_matrices']))
                if real:
                    # k_rproj, real arithmetic.  One sweep of the one-sided Jacobi = n (n - 1) / 2 column pairs, each 6 n flop for the
                    # three Gram sums and 6 flop per row for the rotation of the two X~ columns (n rows) and of the two V_r columns
                    # (n rows): 18 n per pair, 9 n^3 per sweep (deflated columns counted as present: an upper bound); the four
                    # products of an order (X~, warm start, U, apply) 2 n^2 (2 N + 2 n) flop
                    per_sweep, prod = 9.0 * ns ** 3, 2.0 * ns ** 2 * (2.0 * N + 2.0 * ns) * active
                    kname = ('polar (k_rproj: the whole reciprocal projection in real arithmetic -- four f64 MFMA products and the '
                             'one-sided Jacobi SVD in LDS, one workgroup per (restart, slot of orders))')
                    slots = e0.projection_slots() or int(active.sum())            # workgroups per restart (host-packed slots of orders)
                else:
                    # complex one-sided Jacobi: 16 n flop for the Gram sums and 24 flop per row for the rotations: 32 n^3 per sweep
                    per_sweep, prod = 32.0 * ns ** 3, 0.0 * ns
                    kname = 'polar (k_polar_jacobi_lds: complex one-sided Jacobi SVD in LDS, one workgroup per (restart, order))'
                    slots = int(active.sum())
                flops = float((swp * per_sweep[None, :] + prod[None, :]).sum())
                cus = int(min(Bp * slots, e0_cus))
                peak_cu = 4 * 32 * 2.4e9 / 1e12                                   # FP64 vector: 32 flop / clk / SIMD at 2.4 GHz
                ach = flops / (fam_ms['polar']['avg_ms'] * 1e-3) / 1e12
                crit = float(swp[:, -1].max() * per_sweep[-1] + prod[-1])
                roofline = {'bound': 'fp64_valu', 'kernel': kname,
                            'achieved': ach, 'peak': e0_cus * peak_cu, 'unit': 'TFLOP/s', 'frac': ach / (e0_cus * peak_cu), 'traffic': polar_traffic,
                            'cus_used': cus, 'cus_total': e0_cus, 'frac_of_cus_used': ach / (cus * peak_cu),
                            'avg_launch_ms': fam_ms['polar']['avg_ms'],
                            'algorithmic_flops_per_launch': flops, 'restarts_per_launch': Bp,
                            'sweeps_restart0': [int(x) for x in swp[0]],
                            'share_of_step': fam_ms['polar']['total_ms'] / sum(v['total_ms'] for v in tops.values()),
                            # the launch lasts as long as its largest matrix (l = L, one CU): that matrix against one CU's peak
                            'critical_matrix': {'n': int(ns[-1]), 'flops': crit, 'peak_one_cu': peak_cu,
                                                'frac_one_cu': crit / (fam_ms['polar']['avg_ms'] * 1e-3) / 1e12 / peak_cu},
                            'hbm_family': hbm_roof,
                            'note': 'dominant kernel by hipEvent time over the timed region on the stream of engine 0; not an HBM or '
                                    'MFMA kernel: a chain of dependent Jacobi rounds inside one CU per matrix (LDS exchange, lane sums, '
                                    'rotation parameters, barrier; in-kernel timers: profiles/r03_rproj_round_timers.txt); peak = FP64 '
                                    'vector rate of the whole chip, frac_of_cus_used = against the CUs the launch occupies; flops '
                                    'from the sweep counts of the last timed call; hbm_family = the dominant HBM-bound kernel '
                                    'family against 8 TB/s, traffic = FETCH_SIZE x 2 + WRITE_SIZE of profiles/pmc_traffic.json '
                                    'when collected at this batch size'}
            else:
                roofline = dict(hbm_roof)
                roofline['share_of_step'] = fam_ms[dom]['total_ms'] / sum(v['total_ms'] for v in tops.values())
                roofline['note'] = ('dominant kernel family (HBM bound), timed with hipEvents over the timed region on the stream '
                                    'of engine 0; traffic = FETCH_SIZE x 2 + WRITE_SIZE of profiles/pmc_traffic.json when it was '
                                    'collected at this batch size')
    step_bytes = algorithmic_bytes_per_step(N, L, e0.n_theta, e0.n_phi, True)
    whole_step = {'algorithmic_bytes_per_step_per_restart': step_bytes,
                  'achieved_GBps_per_gpu': step_bytes * B * a.steps / elapsed / 1e9,
                  'frac_of_8TBps': step_bytes * B * a.steps / elapsed / 8e12,
                  'pmc_bytes_per_step_per_restart': None}
    try:        # what a step physically moves: FETCH_SIZE x 2 + WRITE_SIZE summed over the kernels of one step (profiles/pmc_traffic.json)
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            pmc_w = json.load(f)
        if pmc_w.get('config') == a.config and pmc_w.get('hbm_bytes_per_step_per_restart'):
            pb = float(pmc_w['hbm_bytes_per_step_per_restart'])
            whole_step['pmc_bytes_per_step_per_restart'] = pb
            whole_step['pmc_GBps_per_gpu'] = pb * B * a.steps / elapsed / 1e9
            whole_step['pmc_frac_of_8TBps'] = pb * B * a.steps / elapsed / 8e12
    except (OSError, ValueError, KeyError):
        pass

    if rank == 0:
        line = {
            'metric': 'MTIP iterations/sec, 128 q-shells x L_max=32',
            'value': its, 'unit': 'MTIP iterations/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'repeats': {'windows_ms': [1e3 * w for w in windows], 'reported': 'median', 'spread_rel': (max(windows) - min(windows)) / elapsed,
                        'bracketed_windows': list(range(n_bracketed)) if not a.no_roofline else [],
                        'note': 'every window: fresh seeded state, W warm-up steps, then exactly K timed steps between barrier + '
                                'synchronize; value and ms_per_step are the median window; the hipEvent family brackets behind `roofline` '
                                'sample the first step of up to four chunks of the bracketed windows (a bracketed step costs ~0.5 ms)'},
            'dtype': 'f64 (complex128)', 'data': 'synthetic',
            'config': {'workload': f'BASELINE config {a.config}: {N} shells x L_max={L}, grid {N}x{e0.n_theta}x{e0.n_phi}, '
                                   f'{B} restarts per GPU on {n_eng} streams, tutorial schedule (HIO/SW/ER, ft_stab on), '
                                   f'{"reference-order" if a.exact else "fused"} step',
                       'restarts_per_gpu': B, 'restarts_total': B * world, 'streams_per_gpu': n_eng,
                       'streams_side_by_side': round(side_by_side, 2),
                       'engines_take_turns': turns is not None,      # mtip_run_group_async (ProjectWorker's default) / --no-turns
                       'parallelism': f'restart-sharded x{world}', 'step_mode': 'exact' if a.exact else 'fused'},
            'roofline': roofline, 'cpu_baseline': cpu, 'whole_step': whole_step, 'kernel_families_ms': fam_ms,
            'phase': phase_of(a.warmup, a.steps),
            'setup_seconds': setup_s, 'final_reduce_seconds': reduce_s,
            'host_enqueue_ms_per_step': 1e3 * host['enqueue_s'] / max(a.steps, 1),
            'best_error_rank0': [float(x) for x in best_err], 'steps_done_per_restart': int(n_done),
            'mean_B0_trace': float(np.trace(bl_mean[0]).real),
            'jacobi_sweeps_last_step_restart0': [int(x) for x in sweeps[0]],
            'jacobi_closing_step_last_step_restart0': [int(x) for x in closing[0]],   # 0 confirming sweep, 1 / 2 first / second order polar step
        }
        print(json.dumps(line))
    for e in engines:
        e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
