"""xframe_amd -- MI355X-native MTIP phasing engine behind xFrame's ``fxs reconstruct`` API.

Only the hot path of ``xframe/projects/fxs/reconstruct.py`` is implemented:
``xframe_amd.csrc``  hand-written HIP kernels + the C ABI (``include/mtip_hip.h``),
``xframe_amd.fxs``   the host-side mirror of the reference's worker / operator / GPU-process API.
"""
__version__ = "0.1.0"
