"""garmentnets_amd -- MI355X-native inference hot path of GarmentNets.

PointNet++ -> gridding -> 3-D UNet -> implicit WNF decoder -> marching cubes, behind the reference's module API
(ConvImplicitWNFPipeline / PointNet2NOCS), executed by hand-written HIP kernels (libgarmentnets_hip.so, C ABI in
include/garmentnets_hip.h).  No CPU fallback: importing is cheap, running needs the built library and a gfx950 GPU.
"""
from .batch import Batch  # noqa: F401

__all__ = ["Batch"]
