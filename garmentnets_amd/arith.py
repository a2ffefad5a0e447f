"""Arithmetic / launch-form selection of a model's dense stages, as an immutable value.

An ``Arith`` travels WITH the call (``ConvImplicitWNFPipeline.arith`` is the model's default; every stage method takes ``arith=``),
so no module-level switch is flipped on the hot path: predict_batch's fp32 re-run of a batch passes ``arith.strict_fp32()`` down its
own calls, two models in one process can run different arithmetic, and a host thread never sees another's mode (no
global mutable state except immutable LUTs).  ``DEFAULT`` is read from the environment ONCE at import.

    conv_mode          arithmetic of the 3x3x3 convolutions (csrc/unet_split.hip / csrc/unet.hip):
                         f16x2  (default) fp32 operands split into two fp16 planes, 3 products on the 16-bit matrix cores, fp32 accumulation;
                                measured error against fp64 is within 2x of the fp32-MFMA kernel's (tests/test_gpu_parity.py)
                         fp32   v_mfma_f32_32x32x2_f32: exact fp32 products, 1/16 of the matrix-core rate
                         bf16x3 bf16 planes, 6 products (fp32-class)
                         bf16x2 bf16 planes, 3 products: PREVIEW quality -- 1e-4 on the WNF is NOT guaranteed (observed 0.9-1.2e-4 on the
                                G=32 goldens), feature-volume error up to 5e-4
    decode_mode        arithmetic of the decoder MLPs: "f16x2" (csrc/decode_split.hip) or "fp32" (csrc/decode.hip)
    sparse_first_conv  occupancy-aware launch of the first two encoder convolutions (exact: bit-identical to the dense launch)
    affine_in_weights  (f16x2) the first two encoder convolutions -- whose input is at rest outside the occupied cells' neighbourhood -- take
                       the GroupNorm affine in per-sample weights + a bias table, so the matrix cores multiply exact zeros there: same MACs,
                       less power, more clock under the socket cap (csrc/conv_prep.hip).  fp32-class like f16x2 itself
    winograd           (f16x2) the 128-wide convolutions over one full-resolution source -- above all the first encoder convolution, 128 -> 128 at 128^3 --
                       run in Winograd F(2,3) form along x: 36 instead of 54 matrix-core tap products per output pair (csrc/unet_wino.hip); the transforms
                       are exact in the operands' zero pattern, error against fp64 stays within the f16x2 contract (tests/test_gpu_parity.py)
    winograd32         (f16x2, with winograd) the same form for the 32- / 64-wide layers at the two finest levels -- the encoder's second convolution 128 -> 32 at
                       128^3, the last decoders' convolutions -- through the 32-wide column-block kernel (csrc/unet_wino32.hip, round 6; the polyphase partial
                       of a decoder's first convolution is added in its epilogue).  Off: those layers take the direct x-strip kernel (csrc/unet_split.hip)
    polyphase_upconv   polyphase form of the decoders' first convolutions (csrc/upconv.hip)
    fold_final_conv    the decoders absorb the UNet's final 1x1x1 convolution into their first layer (conv_implicit_wnf.UNetResult)
    fused_lattice      lattice queries sampled INSIDE the decoder-MLP kernel (gn_implicit_decode_lattice_split, no sampled-row buffer in
                       HBM; bit-identical).  OFF by default: measured 16.57 vs 16.48 ms per 16 x 128^3 lattices -- the gather's LDS traffic and the
                       3-stage weight ring cost the decoder kernel what the separate sampler launch costs
    ggm_fp32           the batched Gaussian gradient magnitude accumulates its taps in fp32 instead of scipy's fp64 (gn_ggm3d_batch_ex, round 6).  OFF by
                       default: the default is scipy's arithmetic bit for bit; on, `volume_gradient_magnitude` is 1e-6-class against it (meshes unchanged)
"""
import dataclasses
import os

CONV_FP32, SPLIT_BF16X2, SPLIT_BF16X3, SPLIT_F16X2 = 0, 2, 3, 4          # 2..4 = GN_SPLIT_* of include/garmentnets_hip.h
CONV_MODE_NAMES = {"fp32": CONV_FP32, "f16x2": SPLIT_F16X2, "bf16x3": SPLIT_BF16X3, "bf16x2": SPLIT_BF16X2}
DECODE_MODES = ("f16x2", "fp32")


def _env_choice(name, default, choices):
    v = os.environ.get(name, default)
    if v not in choices:
        raise ValueError(f"{name}={v!r}: expected one of {sorted(choices)}")
    return v


def _env_flag(name, default=True):
    return os.environ.get(name, "1" if default else "0") != "0"


@dataclasses.dataclass(frozen=True)
class Arith:
    conv_mode: int = SPLIT_F16X2
    decode_mode: str = "f16x2"
    sparse_first_conv: bool = True
    affine_in_weights: bool = True
    winograd: bool = True
    winograd32: bool = True
    polyphase_upconv: bool = True
    fold_final_conv: bool = True
    fused_lattice: bool = False
    ggm_fp32: bool = False

    def __post_init__(self):
        if self.conv_mode not in CONV_MODE_NAMES.values():
            raise ValueError(f"conv_mode={self.conv_mode!r}")
        if self.decode_mode not in DECODE_MODES:
            raise ValueError(f"decode_mode={self.decode_mode!r}")
        if self.conv_mode == SPLIT_BF16X2:
            import warnings
            warnings.warn("garmentnets_amd: conv_mode bf16x2 is a PREVIEW arithmetic -- it does not guarantee the 1e-4 WNF tolerance (0.9-1.2e-4 "
                          "observed on the G=32 goldens) and a finite over-tolerance result is not caught by predict's NaN fallback; use f16x2 "
                          "(default), bf16x3 or fp32 for results held to the reference's tolerance", RuntimeWarning, stacklevel=3)

    @classmethod
    def named(cls, conv="f16x2", decode="f16x2", **kw):
        return cls(conv_mode=CONV_MODE_NAMES[conv], decode_mode=decode, **kw)

    @classmethod
    def from_env(cls):
        return cls(conv_mode=CONV_MODE_NAMES[_env_choice("GARMENTNETS_CONV_MODE", "f16x2", CONV_MODE_NAMES)],
                   decode_mode=_env_choice("GARMENTNETS_DECODE_MODE", "f16x2", DECODE_MODES),
                   sparse_first_conv=_env_flag("GARMENTNETS_SPARSE_CONV"), affine_in_weights=_env_flag("GARMENTNETS_AFFINE_IN_WEIGHTS"),
                   winograd=_env_flag("GARMENTNETS_WINOGRAD"), winograd32=_env_flag("GARMENTNETS_WINOGRAD32"), polyphase_upconv=_env_flag("GARMENTNETS_POLYPHASE"),
                   fold_final_conv=_env_flag("GARMENTNETS_FOLD_FINAL_CONV"), fused_lattice=_env_flag("GARMENTNETS_FUSED_LATTICE", False),
                   ggm_fp32=_env_flag("GARMENTNETS_GGM_FP32", False))

    def replace(self, **kw):
        return dataclasses.replace(self, **kw)

    @property
    def split(self):
        """some stage runs on fp16 / bf16 operand planes (a range violation surfaces as NaN and triggers predict's fp32 re-run)"""
        return self.conv_mode != CONV_FP32 or self.decode_mode != "fp32"

    def strict_fp32(self):
        return self.replace(conv_mode=CONV_FP32, decode_mode="fp32")

    @property
    def conv_name(self):
        return {v: k for k, v in CONV_MODE_NAMES.items()}[self.conv_mode]


DEFAULT = Arith.from_env()
