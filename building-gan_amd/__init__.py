"""building-gan_amd: MI355X-native (gfx950) message-passing core of Building-GAN.

Host side (this package) mirrors the reference's operator interface for the hot path
(/root/reference/Model/models.py Generator / DiscriminatorVoxel, util.py, util_graph.py) and
calls hand-written HIP kernels through the C-ABI library ``csrc/libbgan_hip.so`` declared in
``include/bgan.h``.  There is no CPU fallback: using an op without the library raises.
"""
from .graph import Batch, VolumeDesignGraph, detach_batch, DEFAULT_FOLLOW_BATCH  # noqa: F401
from . import synthetic  # noqa: F401

__all__ = ["Batch", "VolumeDesignGraph", "detach_batch", "synthetic"]
