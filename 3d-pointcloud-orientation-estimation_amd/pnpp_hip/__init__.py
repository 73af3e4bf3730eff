"""pnpp_hip -- Python host side of the MI355X-native PointNet++ / von-Mises-KL training path.

    _lib    ctypes binding of libpnpp_hip.so (C ABI in include/pnpp_hip.h); no fallback
    ops     argument checking, workspaces, torch.autograd glue
    build   hipcc build of the shared library (gfx950)
    optim   flat-buffer Adam / gradient clipping on the device
    dist    one-process-per-GPU data parallelism (RCCL all-reduce of one flat gradient buffer)
"""
__all__ = ["ops", "build", "optim", "dist", "sampling"]
