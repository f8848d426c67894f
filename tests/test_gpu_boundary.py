"""Drop-in boundary on the real MI355X (SURVEY section 8 b): the reference's own sketches executed on the HIP-backed operator
registry, the GPU-process look-alike (matrix @ vects exact, apply_weights kernel_dict vs the oracle), and the reference's
child-process client contract."""
import pytest

import boundary_cases as BC

pytestmark = pytest.mark.gpu


def test_reference_sketch_on_registry(golden_mtip16):
    BC.check_reference_sketch_on_registry(golden_mtip16, None)


def test_reference_sw_and_shift_sketches_on_registry(golden_mtip16):
    BC.check_reference_sw_and_shift_sketches_on_registry(golden_mtip16, None)


def test_gpu_process_boundary():
    BC.check_gpu_process_boundary(None)


def test_gpu_process_from_child_processes():
    """three fresh child processes, each with its own engine on the GPU (well below the box's process limit)"""
    BC.check_gpu_process_from_child_processes(None, n_processes=3)
