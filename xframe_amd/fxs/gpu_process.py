"""HIP-backed look-alike of the reference's GPU-process boundary.

Reference: ``Multiprocessing.openCL_plugin.ClProcess(kernel_dict)`` (``xframe/externalLibraries/openCL_plugin.py:302-325``),
``comm_module.add_gpu_process(cl_process) -> callable`` (``xframe/control/communicators.py:79-82``),
``Multiprocessing.get_number_of_gpus()`` (``xframe/Multiprocessing.py:892-898``).  There a kernel_dict carries
OpenCL source that is compiled at run time inside a daemon process and every call is shm -> H2D -> kernel -> D2H.

Here ``kernel_dict['kernel']`` names a *pre-compiled HIP kernel* of libmtip_hip.so (no PyOpenCL, no run-time
compile); the returned callable takes numpy arrays and returns fresh numpy arrays (caller owns its inputs,
outputs are copies -- same ownership rule as ``Multiprocessing.py:1072``).  Known kernel ids:

  'apply_weights'  the spherical Hankel step (hankel_transforms.py:660-766); const_inputs give the raw weights
  'apply_matrix'   out = matrix @ vects, the example of docs/framework/getting_started.md:319-340 and
                   tests/test_framework_integration.py:230-400
"""
import numpy as np

from . import _lib
from .engine import Engine

_KNOWN = ('apply_weights', 'apply_matrix')


def get_number_of_gpus(lib_path=None):
    return int(_lib.load(lib_path).mtip_device_count())


class ClProcess:
    def __init__(self, kernel_dict):
        self.kernel_dict = kernel_dict
        self.name = kernel_dict['name']
        self.functions = kernel_dict['functions']
        fn = self.functions[0]
        kid = kernel_dict['kernel'] if kernel_dict['kernel'] in _KNOWN else fn['name']
        if kid not in _KNOWN:
            raise ValueError(f'unknown pre-compiled kernel id {kid!r}; known: {_KNOWN}')
        self.kernel_id = kid
        # identity as in openCL_plugin.py:311-315
        self.hash = hash((kid, self.name, str(fn.get('shapes')), str([np.shape(c) for c in fn.get('const_inputs', ())])))


class _GpuProcessManager:
    def __init__(self):
        self.processes = {}
        self.engines = {}

    def _engine(self, key, **kw):
        if key not in self.engines:
            self.engines[key] = Engine(**kw)
        return self.engines[key]

    def add_gpu_process(self, cl_process, device=0, lib_path=None):
        fn = cl_process.functions[0]
        if cl_process.kernel_id == 'apply_matrix':
            matrix = None
            for role, const in zip(fn['arg_roles'], fn.get('const_inputs', ())):
                if role == 'const_input' and isinstance(const, np.ndarray) and const.ndim == 2:
                    matrix = np.asarray(const, dtype=np.float64)
            eng = self._engine(('generic', device, lib_path), settings={'grid': {'n_radial_points': 4, 'max_order': 1}},
                               data=None, max_q=1.0, device=device, lib_path=lib_path)

            def gpu_func(*arrays):
                if matrix is not None:
                    return eng.apply_matrix(matrix, arrays[0])
                return eng.apply_matrix(arrays[0], arrays[1])
            self.processes[cl_process.hash] = gpu_func
            return gpu_func
        # apply_weights: const_inputs = (None, weights (Np, Nk, L+1) complex, None, nq, nlm, nl)
        w = np.asarray(fn['const_inputs'][1], dtype=complex)
        n_p, nq, nl = w.shape
        L = nl - 1
        # w[p,k,l] = raw[l,p,k] * c_l with c_l = (-/+ i)^l * scale (hankel_transforms.py:426-452).  Dividing by
        # (-i)^l leaves a real array for either direction (for odd l the sign of the real part absorbs it),
        # which is what the HIP kernel contracts; it re-applies (-i)^l in its epilogue.
        real_raw = np.empty((nl, n_p, nq))
        for l in range(nl):
            x = w[:, :, l] / ((-1j) ** l)
            if np.abs(x.imag).max() > 1e-12 * max(np.abs(x.real).max(), 1e-300):
                raise ValueError('apply_weights: weights are not of the form real * (+-i)^l')
            real_raw[l] = x.real
        eng = Engine(settings={'grid': {'n_radial_points': nq, 'max_order': L},
                               'fourier_transform': {'type': 'midpoint' if n_p == nq else 'trapz'}},
                     data=None, max_q=1.0, device=device, lib_path=lib_path)
        eng._ck(eng.lib.mtip_set_hankel_weights(eng.ctx, _lib.ptr(np.ascontiguousarray(real_raw)), 1.0, 1.0))

        def gpu_func(rho):
            return eng.hankel(np.asarray(rho, dtype=complex), inverse=False)[0]
        self.processes[cl_process.hash] = gpu_func
        self.engines[cl_process.hash] = eng
        return gpu_func

    def restart_control_worker(self):
        """reference: respawns the GPU daemons (communicators.py:83-84); nothing to do in-process."""
        return None


comm_module = _GpuProcessManager()
add_gpu_process = comm_module.add_gpu_process


def _child_entry(payload):
    import cloudpickle
    func, kwargs = cloudpickle.loads(payload)
    return func(**kwargs)


def process_mp_request(func, n_processes=1, max_concurrent=4, **kwargs):
    """``Multiprocessing.process_mp_request`` look-alike (``xframe/Multiprocessing.py:360-437``) for the GPU-process contract of
    ``tests/test_framework_integration.py:640-747``: ``func(**kwargs)`` runs in ``n_processes`` child processes, each of which
    may call :func:`add_gpu_process`.  The reference forks its workers and talks to GPU daemons; here every child is a
    *fresh* interpreter (``spawn``: a process that has initialised HIP must never be forked) that creates its own engine, at
    most ``max_concurrent`` of them at a time.  The callable travels by value (cloudpickle), so closures work as in the
    reference.  Returns the list of results in task order."""
    import multiprocessing as mp

    import cloudpickle
    payload = cloudpickle.dumps((func, kwargs))
    ctx = mp.get_context('spawn')
    with ctx.Pool(max(1, min(int(n_processes), int(max_concurrent))), maxtasksperchild=1) as pool:
        return pool.map(_child_entry, [payload] * int(n_processes), chunksize=1)
