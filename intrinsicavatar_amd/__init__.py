"""intrinsicavatar_amd -- MI355X (gfx950) implementation of IntrinsicAvatar's volumetric
render_step hot path, exposed through the operator surface the reference imports:

    intrinsicavatar_amd.nerfacc       <- `nerfacc` (traverse_grids, render_weight_from_alpha, ...)
    intrinsicavatar_amd.lib_nerfacc   <- `lib.nerfacc` (ray_resampling*, pack/unpack)
    intrinsicavatar_amd.fast_snarf    <- fast-SNARF JIT modules (fuse_broyden, filter, precompute)

All compute runs in hand-written HIP kernels behind the C ABI of include/ia_amd.h
(libia_amd.so); PyTorch is used for device memory, streams and torch.distributed only.
"""
__version__ = "0.1.0"
