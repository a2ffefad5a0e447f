// split_conv.h -- definitions shared by the split-operand conv kernels (unet_split.hip: direct 27-tap forms; unet_wino.hip: the
// Winograd F(2,3)-along-x form of the 128-wide kernel): launch arguments, work-item order, plane split, MFMA wrapper.
#pragma once
#include "common.h"

typedef float f32x16s __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define SP_TZ 4
#define SP_TY 8
#define SP_TX 8
#define SP_HZ (SP_TZ + 2)
#define SP_HY (SP_TY + 2)
#define SP_HX (SP_TX + 2)
#define SP_HVOX (SP_HZ * SP_HY * SP_HX)
#define SP_KS 16

// border class of coordinate z on an axis of length D for reach r (see SplitArgs::kreach)
__device__ __forceinline__ int sp_axis_class(int z, int D, int r) { return z < r ? z : (z >= D - r ? 2 * r - (D - 1 - z) : r); }
// border mask of coordinate g on an axis of length n for the 3x3x3 taps (see SplitArgs::kbias)
__device__ __forceinline__ int sp_axis_mask(int g, int n) { return (g > 0 ? 1 : 0) | (g < n - 1 ? 2 : 0); }

struct SplitArgs {
    const float *src0;
    const float *src1;
    const float *a;
    const float *d;
    const uint4 *wp;   // [slice][tap][Cout/32][P][lane 64] x 16 B (8 bf16: channels 8h..8h+7 of cout 32*blk + r, lane = 32h + r), + eight zero pad steps
    float *out;
    double *osum;
    double *osq;
    int C0, C1, B, D, H, W, Cout, relu;
    int tiles_y, tiles_x;
    const float *out_scale;   // [Cout] exact powers of two undoing the pack's per-output-channel weight scales (1 for the bf16 modes)
    const float *act_inv;     // NULL or [B]: exact power of two undoing the sample's activation scale (gn_groupnorm_affine)
    // occupancy-aware launch (the layers behind a scattered volume): only the tiles listed in active_list -- entries b * tiles_per_sample + tile at
    // the kernel's tile granularity, ascending, *active_count of them (device values: occ_compact_kernel) -- are visited by this kernel; the grid is the
    // dense launch's, workgroups past the list's end return at once.  Every other tile holds nothing but border-class constants and is written by
    // conv_fill_inactive_kernel (kconst[b][class][Cout], class = (cz * n + cy) * n + cx with n = 2 kreach + 1, c = z < r ? z : (z >= D - r ? 2r - (D-1-z) : r)).
    const int *active_list;
    const int *active_count;
    const float *kconst;
    // polyphase form of a layer whose second source is nearest-upsampled (the decoders' first convolutions): the upsampled part is a
    // 2x2x2-tap convolution per output parity class on the COARSE volume (8/27 of the MACs), computed by gn_upconv_partial (upconv.hip);
    // the launch over the full-resolution source then adds partial[b][z>>1][y>>1][x>>1][((z&1)*4 + (y&1)*2 + (x&1)) * Cout + n] before
    // the ReLU.
    const float *partial;
    // affine-in-weights form (gn_conv_affine_pack, conv_prep.hip): the pack holds one weight set PER SAMPLE (wp_bstride bytes apart; 0 = one
    // set for the batch), out_scale is [B][Cout] (osc_bstride = Cout; 0 = [Cout]) and the GroupNorm shift arrives as the per-(sample, border
    // class, output channel) constant kbias[B][64][Cout] added before the ReLU (class = (mz * 4 + my) * 4 + mx, m = (has a previous voxel
    // on the axis) | (has a next one) << 1; 63 = interior).  The operand is then (x - c) s: exactly zero wherever the layer's input is at rest.
    const float *kbias;
    int64_t wp_bstride;
    int osc_bstride;
    int kreach;               // 1: the layer fed by the scattered volume (27 classes); 2: the layer behind it (125 classes: distance 0 / 1 from
                              // a face or further, per axis); class index per axis c = z < r ? z : (z >= D - r ? 2r - (D-1-z) : r), kconst
                              // [B][(2r+1)^3][Cout] ordered (cz * n + cy) * n + cx
    int chain;                // conv3d_split_wino_kernel: tiles per workgroup (set by gn_launch_conv3d_wino)
};

// Work item of this workgroup: sample b, tile (index inside the sample at the kernel's tile granularity), column block cb.
// XCD-aware order (workgroup i runs on XCD i % 8, each XCD has its own L2): every XCD walks a CONTIGUOUS range of (tile, column block) pairs,
// column blocks of one tile adjacent, tiles in z-fastest order -> the halo overlap of neighbouring tiles and the other column blocks' re-read of
// the same input hit that XCD's L2.  Bijective for any size.  Dense launch: per sample (blockIdx.y) over gridDim.x items.  Compacted launch
// (active_list): over the *active_count * ncb items of the whole batch, linear workgroup id across the grid; -> false: nothing to do.
__device__ __forceinline__ bool sp_work_item(const SplitArgs &p, int ncb, int tiles_per_sample, int &b, int &tile, int &cb) {
    unsigned nblk = gridDim.x, lin = blockIdx.x;
    if (p.active_list) {
        nblk = (unsigned)(*p.active_count) * (unsigned)ncb;
        lin = blockIdx.y * gridDim.x + blockIdx.x;
        if (lin >= nblk) return false;
    }
    const unsigned xcd = lin & 7u, jx = lin >> 3, qx = nblk >> 3, rx = nblk & 7u;
    const unsigned logical = (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + jx;
    const unsigned t = logical / (unsigned)ncb;
    cb = (int)(logical % (unsigned)ncb);
    if (p.active_list) {
        const int item = p.active_list[t];
        b = item / tiles_per_sample;
        tile = item - b * tiles_per_sample;
    } else {
        b = blockIdx.y;
        tile = (int)t;
    }
    return true;
}

typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

// (GroupNorm affine in the halo stage: y = fma(x, a, d) -- one rounding, 4 VALU issue slots per 4 channels instead of 8)
// four floats -> P bf16 planes (exact residual chain x = x1 + x2 [+ x3], xi = bf16_rn of the running residual), each plane
// packed as 4 x bf16 = uint2.  v_cvt_pk_bf16_f32 rounds to nearest even like the host-side pack of the weights.
template <int P, bool F16>
__device__ __forceinline__ void split4(float r0, float r1, float r2, float r3, uint2 (&out)[P]) {
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const f32x2v lo = {r0, r1}, hi = {r2, r3};
        if (F16) {
            const f16x2v blo = __builtin_convertvector(lo, f16x2v), bhi = __builtin_convertvector(hi, f16x2v);
            out[i].x = __builtin_bit_cast(unsigned, blo);
            out[i].y = __builtin_bit_cast(unsigned, bhi);
            if (i + 1 < P) {                       // (exact residuals, one v_fma_mix_f32 each: common.h)
                r0 = gn_resid_lo(out[i].x, r0); r1 = gn_resid_hi(out[i].x, r1); r2 = gn_resid_lo(out[i].y, r2); r3 = gn_resid_hi(out[i].y, r3);
            }
            continue;
        }
        const bf16x2v blo = __builtin_convertvector(lo, bf16x2v), bhi = __builtin_convertvector(hi, bf16x2v);
        out[i].x = __builtin_bit_cast(unsigned, blo);
        out[i].y = __builtin_bit_cast(unsigned, bhi);
        if (i + 1 < P) {
            r0 = __fsub_rn(r0, __uint_as_float(out[i].x << 16));
            r1 = __fsub_rn(r1, __uint_as_float(out[i].x & 0xffff0000u));
            r2 = __fsub_rn(r2, __uint_as_float(out[i].y << 16));
            r3 = __fsub_rn(r3, __uint_as_float(out[i].y & 0xffff0000u));
        }
    }
}

template <bool F16>
__device__ __forceinline__ f32x16s mfma16(const uint4 &a, const uint4 &b, const f32x16s &c) {
    if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// unet_wino.hip: launch of conv3d_split_wino_kernel<true> over `tiles` 4 x 8 x 8 tiles per sample (shape checks: conv3d_gcr_split_impl)
void gn_launch_conv3d_wino(const SplitArgs &p, int tiles, hipStream_t st);
// unet_wino32.hip: launch of conv3d_split_wino32_kernel<true> over `tiles8` 8 x 8 x 8 tiles per sample x Cout / 32 column blocks
bool gn_launch_conv3d_wino32(const SplitArgs &p, int tiles8, hipStream_t st);
void gn_launch_conv3d_wino32pc(const SplitArgs &p, unsigned grid, hipStream_t st);      // unet_wino32pc.hip (p.chain set by the caller)
