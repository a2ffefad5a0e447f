// decode_split.hip -- the implicit decoder MLP [128, 256, 256, OUT<=4] on the 16-bit matrix cores (gn_implicit_decode_split).
//
// Same layer semantics as implicit_decode_kernel (decode.hip; ImplicitWNFDecoder.forward, networks/conv_implicit_wnf.py:128-149):
// y = bn(relu(x W^T + b)) three times, on PRE-SAMPLED feature rows; the BatchNorm affine of a hidden layer is folded into the
// NEXT layer's weights and bias on the host (W' = W diag(s), b' = b + W t: exact algebra, evaluated in fp64).  Arithmetic as in unet_split.hip's f16x2 mode: every fp32
// operand is split into two fp16 planes (x = x1 + x2, residual <= 2^-22 |x|), three MFMA products per fp32 product
// (x1w2 + x2w1 + x1w1, v_mfma_f32_32x32x16_f16, fp32 accumulation).
//
// Range (block floating point, all scales exact powers of two): the decoder's input is the UN-normalised ReLU output of the UNet's
// last convolution, so nothing bounds it a priori.  (a) The rows are multiplied by s_x, chosen on the device per garment from the
// statistics the last conv's epilogue already produced (largest per-channel rms -> [1, 2): gn_decoder_input_scale), the biases by
// the same s_x (ReLU is positively homogeneous, so the whole chain computes s_x times the true values) and the output sum by 1/s_x.
// (b) Every hidden unit carries its own static scale: weight rows are scaled so that the row maximum is in [1, 2) and the inverse
// is folded into the NEXT layer's columns on the host, so hidden activations are split in units where they are O(1..10) whatever
// the checkpoint's weight magnitudes (heavy-tailed rows included).  What is left -- a hidden value beyond 65504 in those units --
// turns into inf - inf = NaN in the accumulators and reaches the output as NaN (gn_relu propagates it): predict_batch checks for
// that and re-runs the batch with the fp32 kernels.
//
// Structure (nothing in common with the fp32 kernel):
//  * TRANSPOSED chain, activations never leave the registers.  A wave owns 32 queries and computes H^T[units][32 q] =
//    W[units][K] . X^T[K][32 q]: the weights are the A operand, the activations the B operand.  The D fragment of a 32-unit
//    block holds, per lane (h = lane>>5, r = lane&31 = query), units 4h + {0..3, 8..11} (registers 0-7) and 4h + {16..19, 24..27}
//    (registers 8-15) -- exactly the 8 k-values per lane a B fragment of the NEXT layer needs, for two 16-deep k groups, once
//    the next layer's weight rows are packed in that k order (host side: ops.pack_decode_split).  So bias/ReLU and the fp16
//    split are applied to the accumulators in place and the result is the next layer's B operand: no LDS round trip, no barrier.
//  * The weights (384 KB of fp16 planes, shared by every query) stream through a 4-stage LDS ring filled by
//    global_load_lds_dwordx4 three stages ahead (stage = 16 KB = one pair of 32-unit blocks x 4 k-groups = 24 MFMAs per wave),
//    one raw s_barrier + counted vmcnt per stage.  A workgroup (4 waves = 128 queries) is persistent: it walks tiles
//    blockIdx.x, +gridDim.x, ... and the ring simply wraps, so only the first stages of a launch are exposed.
//  * One wave per SIMD (launch_bounds(256, 1)): X0 planes 64 + H1 planes 128 + accumulators 32 + A fragments 32 + the next
//    tile's rows 64 VGPRs.
#include "common.h"

typedef float f32x16q __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2q __attribute__((ext_vector_type(2)));

#define DS_K0 128
#define DS_N 256
#define DS_STAGE_BYTES 16384
#define DS_RING 4
#define DS_TILE 128                  // queries per workgroup pass

struct DecSplitArgs {
    const float *xin; int ldxin; long long M;
    const unsigned char *wp;         // [stages][4 k-group steps][2 blocks][2 planes][64 lanes] x 16 B; step order: layer 1 (pair, k-group), layer 2
    const float *tab;                // tab1 [8][2][16] (b1'') | tab2 [8][2][1+OUT][16] (b2'', w3'') | b3', s3, t3 [3][OUT]
    const float *xscale;             // NULL or device {s_x, 1 / s_x, unsafe, 0}: the garment's input scale (exact power of two); unsafe != 0:
                                     // this kernel does nothing (the caller's gated fp32 kernel computes the rows instead)
    float *out; int ldo;
    // lattice form (LAT): the rows are not read from xin but sampled here -- lattice point m0 + m of the (Q,Q,Q) grid of predict.py:145-147 from
    // the channel-last volume vol [D][H][W][32] (trilinear, border, align_corners: the arithmetic of decode.hip, bit for bit)
    const float *vol; int D, H, W, Q; long long m0;
    // blockIdx.y: one row set of a batch (gn_implicit_decode_split_batch: the surface queries of every garment of a batch in ONE launch) with its own rows,
    // outputs and input scale; strides 0 / grid.y 1 for the single call
    long long xin_bs, out_bs; int xscale_bs;
};

// ---- lattice sampling (LAT).  A lane owns 16 channels of its query: 8h..8h+7 and 16+8h..16+8h+7 = four float4 per corner.
struct LatQuery {
    unsigned goff;                   // byte offset of corner (x0, y0, z0), channel 8h, inside the volume (< 2^32: checked on the host)
    float wx0, wx1, wy0, wy1, wz0, wz1;
    unsigned ok1;                    // bit 0 / 1 / 2: x0+1 / y0+1 / z0+1 inside the volume (the lower corners always are, after the border clamp)
};

__device__ __forceinline__ float lat_src_index(float q, int size) {        // = src_index of decode.hip
    const float qn = __fsub_rn(__fmul_rn(2.0f, q), 1.0f);
    const float x = __fmul_rn(__fdiv_rn(__fadd_rn(qn, 1.0f), 2.0f), (float)(size - 1));
    return fminf((float)(size - 1), fmaxf(x, 0.0f));
}

__device__ __forceinline__ LatQuery lat_setup(const DecSplitArgs &p, long long m, int h) {
    const unsigned g = (unsigned)(p.m0 + m), uq = (unsigned)p.Q;          // Q^3 < 2^32 (checked on the host): 32-bit divides
    const unsigned gq = g / uq;
    const int k = (int)(g - gq * uq), i = (int)(gq / uq), j = (int)(gq - (unsigned)i * uq);
    const float sc = __fdiv_rn(1.0f, __fsub_rn((float)p.Q, 1.0f));
    const float qx = __fadd_rn(__fmul_rn((float)i, sc), -0.0f), qy = __fadd_rn(__fmul_rn((float)j, sc), -0.0f), qz = __fadd_rn(__fmul_rn((float)k, sc), -0.0f);
    const float ix = lat_src_index(qx, p.W), iy = lat_src_index(qy, p.H), iz = lat_src_index(qz, p.D);     // component 0 indexes the LAST volume axis
    const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
    const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
    LatQuery q;
    q.wx1 = __fsub_rn(ix, fx0); q.wx0 = __fsub_rn(__fadd_rn(fx0, 1.0f), ix);
    q.wy1 = __fsub_rn(iy, fy0); q.wy0 = __fsub_rn(__fadd_rn(fy0, 1.0f), iy);
    q.wz1 = __fsub_rn(iz, fz0); q.wz0 = __fsub_rn(__fadd_rn(fz0, 1.0f), iz);
    q.ok1 = (x0 + 1 < p.W ? 1u : 0u) | (y0 + 1 < p.H ? 2u : 0u) | (z0 + 1 < p.D ? 4u : 0u);
    q.goff = ((unsigned)((z0 * p.H + y0) * p.W + x0) * 32u + 8u * (unsigned)h) * 4u;
    return q;
}
__device__ __forceinline__ bool lat_ok(const LatQuery &q, int c) { return ((c & 1) == 0 || (q.ok1 & 1u)) && ((c & 2) == 0 || (q.ok1 & 2u)) && ((c & 4) == 0 || (q.ok1 & 4u)); }
__device__ __forceinline__ float lat_weight(const LatQuery &q, int c) {
    return __fmul_rn(__fmul_rn((c & 1) ? q.wx1 : q.wx0, (c & 2) ? q.wy1 : q.wy0), (c & 4) ? q.wz1 : q.wz0);
}
// byte offset of corner c (a corner outside the volume reads corner 0: its value is never used)
__device__ __forceinline__ unsigned lat_corner(const DecSplitArgs &p, const LatQuery &q, int c) {
    return lat_ok(q, c) ? q.goff + (unsigned)((((c >> 2) * p.H + ((c >> 1) & 1)) * p.W + (c & 1)) * 128) : q.goff;
}
typedef float lat_f4 __attribute__((ext_vector_type(4)));
// the four float4 of one corner, issued from inline asm (scalar base + 32-bit offset): invisible to hipcc's vmcnt bookkeeping, which would
// otherwise drain the weight-DMA ring at every use; the caller waits by hand (counted s_waitcnt) and pins the first use behind that wait
// half a corner (2 float4 = 8 of the lane's 16 channels; half 0: channels 8h.., half 1: 16 + 8h..)
__device__ __forceinline__ void lat_issue(const float *vol, unsigned off, int half, lat_f4 (&t)[2]) {
    if (half == 0) {
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(t[0]) : "v"(off), "s"(vol) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:16" : "=v"(t[1]) : "v"(off), "s"(vol) : "memory");
    } else {
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:64" : "=v"(t[0]) : "v"(off), "s"(vol) : "memory");
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:80" : "=v"(t[1]) : "v"(off), "s"(vol) : "memory");
    }
}

__device__ __forceinline__ void ds_split2(float a, float b, unsigned &p1, unsigned &p2) {
    const f32x2q v = {a, b};
    const h16x2 x1 = __builtin_convertvector(v, h16x2);
    p1 = __builtin_bit_cast(unsigned, x1);
    const f32x2q res = {gn_resid_lo(p1, a), gn_resid_hi(p1, b)};      // (exact residuals, one v_fma_mix_f32 each: common.h)
    const h16x2 x2 = __builtin_convertvector(res, h16x2);
    p2 = __builtin_bit_cast(unsigned, x2);
}

// bias + ReLU of an accumulator register quad: the adds as two v_pk_add_f32 (the same IEEE additions, half the issue slots), the ReLU as one v_maximum3_f32 each (gn_relu)
__device__ __forceinline__ void ds_bias_relu4(float a0, float a1, float a2, float a3, const float4 &bv, float &v0, float &v1, float &v2, float &v3) {
    const f32x2q s01 = (f32x2q){a0, a1} + (f32x2q){bv.x, bv.y};
    const f32x2q s23 = (f32x2q){a2, a3} + (f32x2q){bv.z, bv.w};
    v0 = gn_relu(s01.x); v1 = gn_relu(s01.y); v2 = gn_relu(s23.x); v3 = gn_relu(s23.y);
}

__device__ __forceinline__ f32x16q ds_mfma(const uint4 &a, const uint4 &b, const f32x16q &c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
}

// The weight DMA is issued from inline asm ON PURPOSE: hipcc (ROCm 7.2) guards every ds_read that follows a
// __builtin_amdgcn_global_load_lds with s_waitcnt vmcnt(0) (it cannot prove the read does not alias the DMA's LDS
// destination), which would drain the three-stage-deep ring at every k-group.  Slot reuse is made safe by hand instead: the
// counted s_waitcnt + s_barrier of the stage hand-over.  Measured (262144 rows): 0.165 ms with the asm DMA, 0.201 ms with the
// builtin, 0.545 ms for the fp32-MFMA kernel.  (m0 = LDS byte address of the wave's 1-KB destination; lane i lands at
// +16 i.  Nothing else in this kernel uses m0.)
__device__ __forceinline__ void ds_glds16(const void *g, unsigned lds_addr) {  // = gn_glds16 (common.h)
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds_addr) : "memory");
}
// the same with a wave-uniform base in SGPRs and a 32-bit per-lane byte offset: no 64-bit VALU address arithmetic per piece
__device__ __forceinline__ void ds_glds16_s(const void *sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// a wave's four 1-KB pieces of one stage in one statement: ONE scalar base and ONE m0 value, the pieces told apart by the instruction's immediate
// offset -- which the hardware adds to the global AND to the LDS address (LLVM's llvm.amdgcn.global.load.lds: "imm offset (applied to both global
// and LDS address)"), and both step by 1024 from piece to piece.  Four statements with four bases made hipcc keep 72 address pairs in SGPRs across
// the tile loop and spill them into VGPR lanes: 79 - 130 v_readlane_b32 per tile in the VALU stream of a VALU-issue-bound kernel.
__device__ __forceinline__ void ds_glds16x4_s(const void *sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %0, %1\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:2048\n\t"
                 "global_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

// s_waitcnt immediates (gfx9 encoding: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]); expcnt left at 7 (no wait)
#define DS_WAIT_VM_LGKM0(N) __builtin_amdgcn_s_waitcnt(((N) & 15) | 0x70 | (((N) >> 4) << 14))

// K0G = 16-deep k-groups of the first layer: 8 for the plain [128, 256, 256, OUT] decoder (the non-default path: its 448 live
// registers no longer fit without ~40 spilled values since the NaN-propagating ReLU, 0.34 ms per 262144 rows), 2 when the UNet's final 1x1x1
// convolution (32 -> 128, linear) has been folded into the first layer on the host (pack_decode_split(..., final_conv)): the
// decoder then reads 32-channel rows sampled from the PRE-final feature volume.
template <int OUTC, int K0G, bool LAT = false>
__global__ __launch_bounds__(256, (K0G <= 2 && OUTC == 1) ? 2 : 1) void implicit_decode_split_kernel(DecSplitArgs p) {
    static_assert(!LAT || (K0G == 2 && OUTC == 1), "the lattice form exists for the folded scalar decoder");
    // K0G = 2: the first-layer planes are 16 registers instead of 64, which leaves room for TWO workgroups per CU (2 waves per SIMD:
    // one wave's epilogue / load work overlaps the other's MFMAs by itself) with a single accumulator set and an immediate epilogue
    // (scalar output only: the 2-4 output variants need the registers of the second wave for their w3 tables)
    constexpr int NSETS = (K0G <= 2 && OUTC == 1) ? 1 : 2;
    constexpr int TAB1 = 8 * 2 * 16, TAB2 = 8 * 2 * (1 + OUTC) * 16, TABN = TAB1 + TAB2 + 3 * OUTC;
    constexpr int NS1 = 4 * K0G, NSTEPS = NS1 + 64, NSTAGE = NSTEPS / 4;   // k-group steps: layer 1 (4 pairs x K0G), layer 2 (4 x 16)
    constexpr int NRAW = 2 * K0G;                                          // float4 row loads per lane per tile
    constexpr int RAW_STAGE = NSTAGE - 8;                                  // where the next tile's rows are requested
    static_assert(NSTEPS % 4 == 0 && (K0G == 2 || K0G == 8), "stage = 4 k-group steps");
    // LAT: a 3-stage ring (two stages ahead) makes room in LDS for the gather's accumulators [4][256] float4 and its per-lane query record
    // [8][256] dwords: the kernel has no REGISTERS to spare for them (239 of 256 without the gather)
    constexpr int RING = LAT ? 3 : DS_RING;
    constexpr int TAB_BYTES = ((TABN * 4 + 15) / 16) * 16, LATB = LAT ? (16 + 8) * 1024 : 0;
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING * DS_STAGE_BYTES + TAB_BYTES + LATB];
    float *const tab = reinterpret_cast<float *>(smem + RING * DS_STAGE_BYTES);
    float4 *const lacc = reinterpret_cast<float4 *>(smem + RING * DS_STAGE_BYTES + TAB_BYTES);           // [4][256]: conflict-free b128
    unsigned *const lrec = reinterpret_cast<unsigned *>(smem + RING * DS_STAGE_BYTES + TAB_BYTES + 16 * 1024);   // [8][256]
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long ntiles = (p.M + DS_TILE - 1) / DS_TILE;

    if (p.xin) p.xin += (long long)blockIdx.y * p.xin_bs;
    p.out += (long long)blockIdx.y * p.out_bs;
    if (p.xscale) p.xscale += (long long)blockIdx.y * p.xscale_bs;
    if (p.xscale && p.xscale[2] != 0.f) return;     // wave-uniform: the whole grid leaves (gn_decoder_input_scale's verdict)
    const float sx = p.xscale ? p.xscale[0] : 1.f, inv_sx = p.xscale ? p.xscale[1] : 1.f;
    // the bias entries (all of tab1, row 0 of every tab2 block) enter in the scaled units of the chain
    for (int i = tid; i < TABN; i += 256) {
        const bool bias = i < TAB1 || (i < TAB1 + TAB2 && ((i - TAB1) / 16) % (1 + OUTC) == 0);
        tab[i] = bias ? __fmul_rn(p.tab[i], sx) : p.tab[i];
    }

    // each wave DMAs 4 of a stage's 16 fragments.  Stage s of tile n sits in ring slot (n * NSTAGE + s) % 4: `sb` carries n * NSTAGE
    const unsigned char *wsrc = p.wp + (wave * 4) * 1024;   // wave-uniform (SGPRs); the lane adds 16 * lane
    const unsigned lane16 = lane * 16;
    int sb = 0;                                     // stays 0 (and folds away) when NSTAGE % 4 == 0
    constexpr bool SB_ZERO = (NSTAGE % RING == 0);
    static_assert(!LAT || SB_ZERO, "the 3-stage ring is indexed with compile-time slots");
#define DS_SB (SB_ZERO ? 0 : sb)
#define DS_ISSUE(STAGE, SLOT)                                                                                                  \
    ds_glds16x4_s(wsrc + (size_t)(STAGE) * DS_STAGE_BYTES, lane16, lds_base + (SLOT) * DS_STAGE_BYTES + (wave * 4) * 1024);
    DS_ISSUE(0, 0) DS_ISSUE(1, 1) DS_ISSUE(2, 2)
    if (!LAT) { DS_ISSUE(3, 3) }

    // first tile's rows: lane (h, r) holds the 8 channels 16g + 8h .. + 7 of query r for g = 0..K0G-1
    float4 raw[NRAW];
    {
        long long m = (long long)blockIdx.x * DS_TILE + wave * 32 + r;
        if (m >= p.M) m = p.M - 1;
        if constexpr (LAT) {                        // first tile: sampled synchronously (plain loads; the prologue is drained below anyway)
            const LatQuery lq = lat_setup(p, m, h);
#pragma unroll
            for (int g = 0; g < NRAW; ++g) raw[g] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (lat_ok(lq, c)) {
                    const float4 *cp = reinterpret_cast<const float4 *>(reinterpret_cast<const unsigned char *>(p.vol) + lat_corner(p, lq, c));
                    const float w = lat_weight(lq, c);
                    const float4 v[4] = {cp[0], cp[1], cp[4], cp[5]};
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        raw[g].x = __fadd_rn(raw[g].x, __fmul_rn(v[g].x, w));
                        raw[g].y = __fadd_rn(raw[g].y, __fmul_rn(v[g].y, w));
                        raw[g].z = __fadd_rn(raw[g].z, __fmul_rn(v[g].z, w));
                        raw[g].w = __fadd_rn(raw[g].w, __fmul_rn(v[g].w, w));
                    }
                }
        } else {
            const float4 *row = reinterpret_cast<const float4 *>(p.xin + m * p.ldxin + 8 * h);
#pragma unroll
            for (int g = 0; g < K0G; ++g) { raw[2 * g] = row[4 * g]; raw[2 * g + 1] = row[4 * g + 1]; }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0): prologue DMAs + table stores
    __syncthreads();

    const unsigned char *const ring_rd = smem + lane * 16;
    uint4 A[4], nA[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + f * 1024);

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        // ---- X0 planes from the prefetched rows
        uint4 x0[2][K0G], h1[2][16];
#pragma unroll
        for (int g = 0; g < K0G; ++g) {
            ds_split2(__fmul_rn(raw[2 * g].x, sx), __fmul_rn(raw[2 * g].y, sx), x0[0][g].x, x0[1][g].x);
            ds_split2(__fmul_rn(raw[2 * g].z, sx), __fmul_rn(raw[2 * g].w, sx), x0[0][g].y, x0[1][g].y);
            ds_split2(__fmul_rn(raw[2 * g + 1].x, sx), __fmul_rn(raw[2 * g + 1].y, sx), x0[0][g].z, x0[1][g].z);
            ds_split2(__fmul_rn(raw[2 * g + 1].z, sx), __fmul_rn(raw[2 * g + 1].w, sx), x0[0][g].w, x0[1][g].w);
        }
        float psum[OUTC];
#pragma unroll
        for (int o = 0; o < OUTC; ++o) psum[o] = 0.f;
        // LAT: the NEXT tile's rows are gathered under this tile's MFMAs -- batch b = (corner b >> 1, channel half b & 1) is issued at the hand-over
        // of stage LAT_T0 + b and consumed at the next hand-over: 8 registers in flight; accumulators and query record live in LDS
        lat_f4 lt[2];
        constexpr int LAT_T0 = 1, LAT_NB = 16;
        static_assert(!LAT || LAT_T0 + LAT_NB == NSTAGE - 1, "the last batch is consumed at the tile's last hand-over");
        // two accumulator sets: the epilogue of block pair P-1 (VALU) is spread over the MFMAs of the steps that follow it
        f32x16q acc[NSETS][2];

        // epilogue of registers [4 qd, 4 qd + 4) of both blocks of block pair P (0-3: layer 1, 4-7: layer 2)
        auto epilogue = [&](int P, int qd) {
            const int set = P & (NSETS - 1);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                if (P < 4) {
                    const int nb = 2 * P + blk;
                    const float4 bv = *reinterpret_cast<const float4 *>(tab + (nb * 2 + h) * 16 + 4 * qd);
                    float v0, v1, v2, v3;
                    ds_bias_relu4(acc[set][blk][4 * qd + 0], acc[set][blk][4 * qd + 1], acc[set][blk][4 * qd + 2], acc[set][blk][4 * qd + 3], bv, v0, v1, v2, v3);
                    // registers 0-7 -> k-group 2nb, 8-15 -> k-group 2nb+1 of layer 2; a register quad fills half a fragment
                    const int g2 = 2 * nb + (qd >> 1);
                    if (qd & 1) {
                        ds_split2(v0, v1, h1[0][g2].z, h1[1][g2].z);
                        ds_split2(v2, v3, h1[0][g2].w, h1[1][g2].w);
                    } else {
                        ds_split2(v0, v1, h1[0][g2].x, h1[1][g2].x);
                        ds_split2(v2, v3, h1[0][g2].y, h1[1][g2].y);
                    }
                } else {
                    const int nb = 2 * (P - 4) + blk;
                    const float *tb = tab + TAB1 + ((nb * 2 + h) * (1 + OUTC)) * 16 + 4 * qd;
                    const float4 bv = *reinterpret_cast<const float4 *>(tb);
                    float v0, v1, v2, v3;
                    ds_bias_relu4(acc[set][blk][4 * qd + 0], acc[set][blk][4 * qd + 1], acc[set][blk][4 * qd + 2], acc[set][blk][4 * qd + 3], bv, v0, v1, v2, v3);
#pragma unroll
                    for (int o = 0; o < OUTC; ++o) {
                        const float4 wv = *reinterpret_cast<const float4 *>(tb + (1 + o) * 16);
                        psum[o] = fmaf(v0, wv.x, psum[o]);
                        psum[o] = fmaf(v1, wv.y, psum[o]);
                        psum[o] = fmaf(v2, wv.z, psum[o]);
                        psum[o] = fmaf(v3, wv.w, psum[o]);
                    }
                }
            }
        };

#pragma unroll
        for (int step = 0; step < NSTEPS; ++step) {
            const int t = step >> 2, kg = step & 3;                          // DMA stage, k-group slot inside it
            const bool l1 = step < NS1;
            const int P = l1 ? step / K0G : 4 + ((step - NS1) >> 4);         // block pair
            const int g = l1 ? step % K0G : ((step - NS1) & 15);             // k-group of the layer
            const int set = P & (NSETS - 1);
            if (g == 0) {
#pragma unroll
                for (int q = 0; q < 16; ++q) { acc[set][0][q] = 0.f; acc[set][1][q] = 0.f; }
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) A[f] = nA[f];
            if (kg < 3) {
#pragma unroll
                for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + ((t + DS_SB) % RING) * DS_STAGE_BYTES + ((kg + 1) * 4 + f) * 1024);
            } else {
                // ---- stage hand-over: stage t+1 has landed for everybody, stage t has been read by everybody
                // VM queue (oldest first): stage t+1, t+2, t+3 [+ the NRAW row loads issued in stage RAW_STAGE]; see the note below
                // LAT: VM queue, oldest first, with the corner loads issued BEFORE the stage's DMAs: G(t-3) DMA(t+1) G(t-2) DMA(t+2) G(t-1) DMA(t+3);
                // the corner issued one hand-over ago must have landed: everything but DMA(t+3)
                // LAT (3-stage ring, the batch's two loads issued BEFORE the stage's DMAs): G(t-2) DMA(t+1) G(t-1) DMA(t+2) -> everything but DMA(t+2)
                if (LAT) DS_WAIT_VM_LGKM0(4);
                else if (t > RAW_STAGE && t <= RAW_STAGE + 3) DS_WAIT_VM_LGKM0(8 + NRAW);
                else DS_WAIT_VM_LGKM0(8);
                __builtin_amdgcn_s_barrier();
                if constexpr (LAT) {
                    if (t == LAT_T0) {                                    // next tile's query (clamped: the last tile re-samples its own)
                        long long tn = tile + gridDim.x;
                        if (tn >= ntiles) tn = tile;
                        long long m = tn * DS_TILE + wave * 32 + r;
                        if (m >= p.M) m = p.M - 1;
                        const LatQuery lq = lat_setup(p, m, h);
                        lrec[0 * 256 + tid] = lq.goff; lrec[1 * 256 + tid] = lq.ok1;
                        lrec[2 * 256 + tid] = __float_as_uint(lq.wx0); lrec[3 * 256 + tid] = __float_as_uint(lq.wx1);
                        lrec[4 * 256 + tid] = __float_as_uint(lq.wy0); lrec[5 * 256 + tid] = __float_as_uint(lq.wy1);
                        lrec[6 * 256 + tid] = __float_as_uint(lq.wz0); lrec[7 * 256 + tid] = __float_as_uint(lq.wz1);
#pragma unroll
                        for (int gg = 0; gg < 4; ++gg) lacc[gg * 256 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    if (t >= LAT_T0 && t <= LAT_T0 + LAT_NB) {
                        __builtin_amdgcn_sched_barrier(0);               // the gather's temporaries live between these two fences only
                        const unsigned goff = lrec[tid], ok1 = lrec[256 + tid];
                        if (t > LAT_T0) {                                 // consume batch t - LAT_T0 - 1: accumulators read-modify-written in LDS
                            const int bb = t - LAT_T0 - 1, c = bb >> 1, half = bb & 1;
                            asm volatile("" : "+v"(lt[0]));               // first use of the asm loads: behind the hand-over's wait
                            asm volatile("" : "+v"(lt[1]));
                            const bool okc = ((c & 1) == 0 || (ok1 & 1u)) && ((c & 2) == 0 || (ok1 & 2u)) && ((c & 4) == 0 || (ok1 & 4u));
                            // branch-free (a divergent branch here would cut the unrolled MFMA stream into basic blocks): a corner outside the
                            // volume gets weight 0 on corner 0's (finite) data -- acc + (+-0) == acc exactly, acc is never -0
                            const float wx = __uint_as_float(lrec[(2 + (c & 1)) * 256 + tid]);
                            const float wy = __uint_as_float(lrec[(4 + ((c >> 1) & 1)) * 256 + tid]);
                            const float wz = __uint_as_float(lrec[(6 + (c >> 2)) * 256 + tid]);
                            const float w = okc ? __fmul_rn(__fmul_rn(wx, wy), wz) : 0.f;
#pragma unroll
                            for (int i2 = 0; i2 < 2; ++i2) {
                                float4 a4 = lacc[(2 * half + i2) * 256 + tid];
                                a4.x = __fadd_rn(a4.x, __fmul_rn(lt[i2].x, w));
                                a4.y = __fadd_rn(a4.y, __fmul_rn(lt[i2].y, w));
                                a4.z = __fadd_rn(a4.z, __fmul_rn(lt[i2].z, w));
                                a4.w = __fadd_rn(a4.w, __fmul_rn(lt[i2].w, w));
                                lacc[(2 * half + i2) * 256 + tid] = a4;
                            }
                        }
                        if (t < LAT_T0 + LAT_NB) {                        // issue batch t - LAT_T0 (a corner outside the volume reads corner 0: never used)
                            const int bb = t - LAT_T0, c = bb >> 1, half = bb & 1;
                            const bool okc = ((c & 1) == 0 || (ok1 & 1u)) && ((c & 2) == 0 || (ok1 & 2u)) && ((c & 4) == 0 || (ok1 & 4u));
                            const unsigned off = okc ? goff + (unsigned)((((c >> 2) * p.H + ((c >> 1) & 1)) * p.W + (c & 1)) * 128) : goff;
                            lat_issue(p.vol, off, half, lt);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                DS_ISSUE((t + RING) % NSTAGE, (t + DS_SB) % RING)
                if (!LAT && t == RAW_STAGE) {                             // next tile's rows (clamped: the last tile re-reads its own)
                    long long tn = tile + gridDim.x;
                    if (tn >= ntiles) tn = tile;
                    long long m = tn * DS_TILE + wave * 32 + r;
                    if (m >= p.M) m = p.M - 1;
                    const float4 *row = reinterpret_cast<const float4 *>(p.xin + m * p.ldxin + 8 * h);
#pragma unroll
                    for (int gg = 0; gg < K0G; ++gg) { raw[2 * gg] = row[4 * gg]; raw[2 * gg + 1] = row[4 * gg + 1]; }
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + ((t + 1 + DS_SB) % RING) * DS_STAGE_BYTES + f * 1024);
            }
            const uint4 b1 = l1 ? x0[0][g % K0G] : h1[0][g], b2 = l1 ? x0[1][g % K0G] : h1[1][g];
            // A: [blk0 w1, blk0 w2, blk1 w1, blk1 w2]; smallest terms first
            acc[set][0] = ds_mfma(A[1], b1, acc[set][0]);
            acc[set][1] = ds_mfma(A[3], b1, acc[set][1]);
            acc[set][0] = ds_mfma(A[0], b2, acc[set][0]);
            acc[set][1] = ds_mfma(A[2], b2, acc[set][1]);
            acc[set][0] = ds_mfma(A[0], b1, acc[set][0]);
            acc[set][1] = ds_mfma(A[2], b1, acc[set][1]);
            // the epilogue of an earlier pair rides along: pair Pp finished at step Lp; its 4 register quads are handled during the
            // NCH steps that follow (NCH <= K0G for layer 1, so that they are done before pair Pp + 2 reuses the accumulator set).
            // With a single accumulator set the pair's epilogue runs right after its last step instead.
            if (NSETS == 2) {
#pragma unroll
                for (int Pp = 0; Pp < 7; ++Pp) {
                    const int Lp = Pp < 4 ? (Pp + 1) * K0G - 1 : NS1 + (Pp - 3) * 16 - 1;
                    const int NCH = (Pp < 4 && K0G < 4) ? K0G : 4;
                    const int c = step - 1 - Lp;
                    if (c >= 0 && c < NCH) {
#pragma unroll
                        for (int qd = c * (4 / NCH); qd < (c + 1) * (4 / NCH); ++qd) epilogue(Pp, qd);
                    }
                }
            } else {
                const int Lp = P < 4 ? (P + 1) * K0G - 1 : NS1 + (P - 3) * 16 - 1;
                if (step == Lp && P < 7) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) epilogue(P, qd);
                }
            }
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) epilogue(7, qd);
        if constexpr (LAT) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) raw[gg] = lacc[gg * 256 + tid];      // the next tile's sampled rows
        }
        if (!SB_ZERO) sb = (sb + NSTAGE) % RING;
        // ---- output layer: the two lane halves hold disjoint unit sets of the same query
        const long long m = tile * DS_TILE + wave * 32 + r;
#pragma unroll
        for (int o = 0; o < OUTC; ++o) {
            const float s = psum[o] + __shfl_xor(psum[o], 32);
            if (h == 0 && m < p.M) {
                const float *t3 = tab + TAB1 + TAB2;
                float y = gn_relu(__fadd_rn(__fmul_rn(s, inv_sx), t3[o]));
                y = __fadd_rn(__fmul_rn(y, t3[OUTC + o]), t3[2 * OUTC + o]);
                p.out[m * p.ldo + o] = y;
            }
        }
    }
#undef DS_ISSUE
#undef DS_SB
    __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0): the wrapped-around DMAs must land before the LDS goes away
    __syncthreads();
}
// Note on the counted waits.  VM operations retire in issue order.  At the hand-over of stage t the queue holds the DMAs of
// stages t+1, t+2, t+3 (4 per wave each); stage t+1 has landed when at most 8 remain.  The NRAW row loads of the next tile are
// issued right after the DMAs of stage RAW_STAGE+4 (at the hand-over of stage RAW_STAGE): for the next three hand-overs they are
// younger than the stage being waited for, so 8 + NRAW may remain; after that they are older and the plain count applies again
// (they had four stages = 24 x 4 MFMAs to arrive).  The wrap-around (stage (t+4) % NSTAGE belongs to the NEXT tile; the last tile
// fetches four stages it never uses) keeps every count independent of the tile.

// ------------------------------------------------------------------------------------------------ [32, 512, 512, OUT]: the class-default hidden width
// ImplicitWNFDecoder's constructor default is nn_channels = (128, 512, 512, 1) (networks/conv_implicit_wnf.py:122); with the UNet's final 1x1x1
// convolution folded into the first layer (folded_pack) that is a [32, 512, 512, OUT] chain.  Same transposed register chain, same arithmetic and
// scaling as implicit_decode_split_kernel, laid out for 16 blocks of 32 hidden units per layer:
//  * one wave per SIMD (the layer-1 activations are 2 planes x 32 k-groups = 256 registers), 128 queries per workgroup pass, persistent workgroups;
//  * layer 1 (8 block pairs x 2 k-groups) fully unrolled -- its epilogue writes the layer-2 operand registers, whose indices must be static;
//    layer 2 (8 pairs x 32 k-groups) as a RUNTIME loop over pairs of block pairs (4 iterations x 64 unrolled steps = 384 MFMAs per body, the size
//    of the 256-wide kernel's whole tile): a pair's 8 stages are a multiple of the 4-stage ring, so every ring slot stays a compile-time constant;
//  * two accumulator sets: the epilogue of block pair P (bias, ReLU, output-layer partial sums) rides along the MFMAs of pair P+1;
//  * weights: 68 stages of 16 KB per tile (1.06 MB, L2-resident) through the same 4-stage LDS ring, global_load_lds three stages ahead, one counted
//    wait + raw barrier per stage.  The next tile's rows are ordinary loads issued at the start of the last loop iteration: hipcc's own vmcnt wait at
//    their first use can only be stricter than needed (VM operations retire in order), never too lenient.
template <int OUTC>
__global__ __launch_bounds__(256, 1) void implicit_decode_split512_kernel(DecSplitArgs p) {
    constexpr int NH = 512, NB = NH / 32, NPAIR = NB / 2, KG2 = NH / 16, K0G = 2;
    constexpr int TAB1 = NB * 2 * 16, TAB2 = NB * 2 * (1 + OUTC) * 16, TABN = TAB1 + TAB2 + 3 * OUTC;
    constexpr int NS1 = NPAIR * K0G, NSTAGE1 = NS1 / 4, NSTAGE = NSTAGE1 + NPAIR * KG2 / 4;      // 4 + 64 stages
    constexpr int TAB_BYTES = ((TABN * 4 + 15) / 16) * 16;
    static_assert(NSTAGE1 % DS_RING == 0 && (KG2 / 4) % DS_RING == 0, "ring slots must be compile-time constants");
    __shared__ __attribute__((aligned(16))) unsigned char smem[DS_RING * DS_STAGE_BYTES + TAB_BYTES];
    float *const tab = reinterpret_cast<float *>(smem + DS_RING * DS_STAGE_BYTES);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long ntiles = (p.M + DS_TILE - 1) / DS_TILE;

    if (p.xin) p.xin += (long long)blockIdx.y * p.xin_bs;
    p.out += (long long)blockIdx.y * p.out_bs;
    if (p.xscale) p.xscale += (long long)blockIdx.y * p.xscale_bs;
    if (p.xscale && p.xscale[2] != 0.f) return;     // wave-uniform: the whole grid leaves (gn_decoder_input_scale's verdict)
    const float sx = p.xscale ? p.xscale[0] : 1.f, inv_sx = p.xscale ? p.xscale[1] : 1.f;
    for (int i = tid; i < TABN; i += 256) {          // the bias entries enter in the scaled units of the chain
        const bool bias = i < TAB1 || (i < TAB1 + TAB2 && ((i - TAB1) / 16) % (1 + OUTC) == 0);
        tab[i] = bias ? __fmul_rn(p.tab[i], sx) : p.tab[i];
    }
    const unsigned char *wsrc = p.wp + (wave * 4) * 1024;
    const unsigned lane16 = lane * 16;
    // stage `st` (0 .. NSTAGE-1, wrapping into the next tile) -> ring slot `slot`
    auto issue = [&](int st, int slot) {
        ds_glds16x4_s(wsrc + (size_t)st * DS_STAGE_BYTES, lane16, lds_base + slot * DS_STAGE_BYTES + (wave * 4) * 1024);
    };
    issue(0, 0); issue(1, 1); issue(2, 2); issue(3, 3);

    float4 raw[4];
    {
        long long m = (long long)blockIdx.x * DS_TILE + wave * 32 + r;
        if (m >= p.M) m = p.M - 1;
        const float4 *row = reinterpret_cast<const float4 *>(p.xin + m * p.ldxin + 8 * h);
        raw[0] = row[0]; raw[1] = row[1]; raw[2] = row[4]; raw[3] = row[5];
    }
    __builtin_amdgcn_s_waitcnt(0x0070);              // vmcnt(0) lgkmcnt(0): prologue DMAs + table stores
    __syncthreads();

    const unsigned char *const ring_rd = smem + lane * 16;
    uint4 A[4], nA[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + f * 1024);

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint4 x0[2][K0G], h1[2][KG2];
#pragma unroll
        for (int g = 0; g < K0G; ++g) {
            ds_split2(__fmul_rn(raw[2 * g].x, sx), __fmul_rn(raw[2 * g].y, sx), x0[0][g].x, x0[1][g].x);
            ds_split2(__fmul_rn(raw[2 * g].z, sx), __fmul_rn(raw[2 * g].w, sx), x0[0][g].y, x0[1][g].y);
            ds_split2(__fmul_rn(raw[2 * g + 1].x, sx), __fmul_rn(raw[2 * g + 1].y, sx), x0[0][g].z, x0[1][g].z);
            ds_split2(__fmul_rn(raw[2 * g + 1].z, sx), __fmul_rn(raw[2 * g + 1].w, sx), x0[0][g].w, x0[1][g].w);
        }
        float psum[OUTC];
#pragma unroll
        for (int o = 0; o < OUTC; ++o) psum[o] = 0.f;
        f32x16q acc[2][2];

        // one k-group step: stage slot `slot` (compile-time), k-group slot kg inside it; `next_stage` = the stage to request at a hand-over (kg == 3)
        auto step = [&](int kg, int slot, int next_stage, const uint4 &b1, const uint4 &b2, f32x16q (&ac)[2]) {
#pragma unroll
            for (int f = 0; f < 4; ++f) A[f] = nA[f];
            if (kg < 3) {
#pragma unroll
                for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + slot * DS_STAGE_BYTES + ((kg + 1) * 4 + f) * 1024);
            } else {
                // hand-over: the next stage has landed for everybody (VM queue: three stages of four pieces; at most 8 may remain), this stage has
                // been read by everybody -> its slot takes the stage three ahead
                DS_WAIT_VM_LGKM0(8);
                __builtin_amdgcn_s_barrier();
                issue(next_stage, slot);
#pragma unroll
                for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + ((slot + 1) % DS_RING) * DS_STAGE_BYTES + f * 1024);
            }
            ac[0] = ds_mfma(A[1], b1, ac[0]);
            ac[1] = ds_mfma(A[3], b1, ac[1]);
            ac[0] = ds_mfma(A[0], b2, ac[0]);
            ac[1] = ds_mfma(A[2], b2, ac[1]);
            ac[0] = ds_mfma(A[0], b1, ac[0]);
            ac[1] = ds_mfma(A[2], b1, ac[1]);
        };
        // layer-1 epilogue of registers [4 qd, 4 qd + 4) of both blocks of pair P -> the layer-2 operand planes (k-groups 2 nb, 2 nb + 1)
        auto epi1 = [&](int P, int qd, f32x16q (&ac)[2]) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int nb = 2 * P + blk;
                const float4 bv = *reinterpret_cast<const float4 *>(tab + (nb * 2 + h) * 16 + 4 * qd);
                float v0, v1, v2, v3;
                ds_bias_relu4(ac[blk][4 * qd + 0], ac[blk][4 * qd + 1], ac[blk][4 * qd + 2], ac[blk][4 * qd + 3], bv, v0, v1, v2, v3);
                const int g2 = 2 * nb + (qd >> 1);
                if (qd & 1) {
                    ds_split2(v0, v1, h1[0][g2].z, h1[1][g2].z);
                    ds_split2(v2, v3, h1[0][g2].w, h1[1][g2].w);
                    // (the layer-2 operand planes are pinned into AGPRs where they are produced: MFMA B operands may be AGPRs, and without the pin hipcc
                    //  spills 34 values around the MFMA stream; with it: no scratch, 402 -> 432 TF-eq, same digests)
                    asm volatile("" : "+a"(h1[0][g2].z), "+a"(h1[1][g2].z), "+a"(h1[0][g2].w), "+a"(h1[1][g2].w));
                } else {
                    ds_split2(v0, v1, h1[0][g2].x, h1[1][g2].x);
                    ds_split2(v2, v3, h1[0][g2].y, h1[1][g2].y);
                    asm volatile("" : "+a"(h1[0][g2].x), "+a"(h1[1][g2].x), "+a"(h1[0][g2].y), "+a"(h1[1][g2].y));
                }
            }
        };
        // layer-2 epilogue of pair P2 (runtime): bias, ReLU, output-layer partial sums
        auto epi2 = [&](int P2, int qd, f32x16q (&ac)[2]) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const int nb = 2 * P2 + blk;
                const float *tb = tab + TAB1 + ((nb * 2 + h) * (1 + OUTC)) * 16 + 4 * qd;
                const float4 bv = *reinterpret_cast<const float4 *>(tb);
                float v0, v1, v2, v3;
                ds_bias_relu4(ac[blk][4 * qd + 0], ac[blk][4 * qd + 1], ac[blk][4 * qd + 2], ac[blk][4 * qd + 3], bv, v0, v1, v2, v3);
#pragma unroll
                for (int o = 0; o < OUTC; ++o) {
                    const float4 wv = *reinterpret_cast<const float4 *>(tb + (1 + o) * 16);
                    psum[o] = fmaf(v0, wv.x, psum[o]);
                    psum[o] = fmaf(v1, wv.y, psum[o]);
                    psum[o] = fmaf(v2, wv.z, psum[o]);
                    psum[o] = fmaf(v3, wv.w, psum[o]);
                }
            }
        };
        auto zero = [&](f32x16q (&ac)[2]) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { ac[0][q] = 0.f; ac[1][q] = 0.f; }
        };

        // ---- layer 1: pairs 0 .. 7, two k-groups each; pair P's epilogue rides along pair P+1's two steps (two register quads per step)
#pragma unroll
        for (int s1 = 0; s1 < NS1; ++s1) {
            const int P = s1 / K0G, g = s1 % K0G, t = s1 >> 2, kg = s1 & 3, set = P & 1;
            if (g == 0) zero(acc[set]);
            step(kg, t % DS_RING, t + DS_RING, x0[0][g], x0[1][g], acc[set]);
            if (P > 0) { epi1(P - 1, 2 * g, acc[set ^ 1]); epi1(P - 1, 2 * g + 1, acc[set ^ 1]); }
        }
        // ---- layer 2: iteration `it` = block pairs 2 it (accumulator set 0) and 2 it + 1 (set 1); stage of its step s2: NSTAGE1 + it * 16 + (s2 >> 2)
        for (int it = 0; it < NPAIR / 2; ++it) {
            if (it == NPAIR / 2 - 1) {               // next tile's rows (clamped: the last tile re-reads its own)
                long long tn = tile + gridDim.x;
                if (tn >= ntiles) tn = tile;
                long long m = tn * DS_TILE + wave * 32 + r;
                if (m >= p.M) m = p.M - 1;
                const float4 *row = reinterpret_cast<const float4 *>(p.xin + m * p.ldxin + 8 * h);
                raw[0] = row[0]; raw[1] = row[1]; raw[2] = row[4]; raw[3] = row[5];
            }
            const int tbase = NSTAGE1 + it * (2 * KG2 / 4);
#pragma unroll
            for (int s2 = 0; s2 < 2 * KG2; ++s2) {
                const int half = s2 / KG2, g = s2 % KG2, tl = s2 >> 2, kg = s2 & 3;
                if (g == 0) zero(acc[half]);
                int nxt = tbase + tl + DS_RING;      // (runtime: wraps into the next tile's first stages)
                if (nxt >= NSTAGE) nxt -= NSTAGE;
                step(kg, tl % DS_RING, nxt, h1[0][g], h1[1][g], acc[half]);
                // the previous pair's epilogue, one register quad per step over the first four steps: layer 1's last pair under the very first
                // layer-2 pair, then layer-2 pair (2 it + half - 1)
                if (g < 4) {
                    if (half == 0) {
                        if (it == 0) epi1(NPAIR - 1, g, acc[1]);
                        else epi2(2 * it - 1, g, acc[1]);
                    } else {
                        epi2(2 * it, g, acc[0]);
                    }
                }
            }
        }
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) epi2(NPAIR - 1, qd, acc[1]);
        // ---- output layer: the two lane halves hold disjoint unit sets of the same query
        const long long m = tile * DS_TILE + wave * 32 + r;
#pragma unroll
        for (int o = 0; o < OUTC; ++o) {
            const float s = psum[o] + __shfl_xor(psum[o], 32);
            if (h == 0 && m < p.M) {
                const float *t3 = tab + TAB1 + TAB2;
                float y = gn_relu(__fadd_rn(__fmul_rn(s, inv_sx), t3[o]));
                y = __fadd_rn(__fmul_rn(y, t3[OUTC + o]), t3[2 * OUTC + o]);
                p.out[m * p.ldo + o] = y;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);              // the wrapped-around DMAs must land before the LDS goes away
    __syncthreads();
}

// the garment's input scale from the per-channel sums of squares of the volume the rows are sampled from (= the statistics the last
// conv's epilogue emitted): s = 2^k with (largest channel rms) * s in [1, 2), clamped to smax (the pack's bound that keeps the scaled
// biases below 2^13).  When the clamp costs more than 2^4 the rows would be split below their natural scale (a checkpoint whose biases
// dwarf weights x activations in some unit): `unsafe` = 1 sends the garment to the fp32 kernel instead, decided on the device, no
// host synchronisation.  out[b] = {s, 1/s, unsafe, 0}
__global__ void decoder_input_scale_kernel(const double *__restrict__ sumsq, int64_t V, int B, int C, float smax, float *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double m2 = 0.0;
    for (int c = 0; c < C; ++c) { const double v = sumsq[(int64_t)b * C + c]; if (v > m2) m2 = v; }
    const float m = (float)sqrt(m2 / (double)V);
    float s = 1.f;
    if (m > 0.f && m < INFINITY) {
        int e = 0;
        (void)frexpf(m, &e);
        e = 1 - e;
        if (e > 100) e = 100;
        if (e < -100) e = -100;
        s = ldexpf(1.f, e);
    }
    const float s0 = s;
    if (s > smax) s = smax;
    out[4 * b] = s;
    out[4 * b + 1] = 1.f / s;
    out[4 * b + 2] = (s0 > 16.f * s) ? 1.f : 0.f;
    out[4 * b + 3] = 0.f;
}

extern "C" int gn_decoder_input_scale(const double *sumsq, int64_t V, int B, int C, float smax, float *out4, void *stream) {
    GN_REQUIRE(B >= 0 && C > 0 && V > 0 && smax > 0.f, "gn_decoder_input_scale: bad sizes");
    if (B == 0) return GN_OK;
    hipLaunchKernelGGL(decoder_input_scale_kernel, dim3((unsigned)gn_cdiv(B, 64)), dim3(64), 0, gn_stream(stream), sumsq, V, B, C, smax, out4);
    GN_LAUNCH_CHECK("gn_decoder_input_scale");
    return GN_OK;
}

// The lattice form (the sampler inside the decoder MLP): rows m0 .. m0+M-1 of the (Q,Q,Q) lattice of predict.py:145-147, sampled from the
// 32-channel channel-last volume by the decoder kernel itself while it multiplies the previous tile -- no sampled-row buffer in HBM.  Folded scalar
// decoder only ([32, 256, 256, 1]: the volume decoder on the UNet's pre-final volume); bit-identical to gn_trilinear_sample + gn_implicit_decode_split.
extern "C" int gn_implicit_decode_lattice_split(const float *vol, int D, int H, int W, int C0, int Q, int64_t m0, int64_t M, const void *wpack,
                                                const float *tab, const float *xscale, int N1, int N2, int OUT, float *out, int ldo, void *stream) {
    GN_REQUIRE(M >= 0 && ldo >= 1 && OUT == 1 && C0 == 32 && N1 == DS_N && N2 == DS_N, "gn_implicit_decode_lattice_split: the lattice form is packed for the [32, 256, 256, 1] decoder");
    GN_REQUIRE(D > 0 && H > 0 && W > 0 && (int64_t)D * H * W * 128 < ((int64_t)1 << 32), "gn_implicit_decode_lattice_split: the volume must be addressable with 32-bit byte offsets");
    GN_REQUIRE(Q > 1 && Q <= 1024 && m0 >= 0 && m0 + M <= (int64_t)Q * Q * Q, "gn_implicit_decode_lattice_split: bad lattice range");
    if (M == 0) return GN_OK;
    GN_REQUIRE(vol && wpack && tab && out, "gn_implicit_decode_lattice_split: null pointer");
    GN_REQUIRE(((uintptr_t)vol & 15) == 0, "gn_implicit_decode_lattice_split: the volume must be 16-byte aligned");
    DecSplitArgs p;
    p.xin = nullptr; p.ldxin = 0; p.M = M; p.wp = (const unsigned char *)wpack; p.tab = tab; p.xscale = xscale; p.out = out; p.ldo = ldo;
    p.vol = vol; p.D = D; p.H = H; p.W = W; p.Q = Q; p.m0 = m0; p.xin_bs = p.out_bs = 0; p.xscale_bs = 0;
    const int64_t ntiles = gn_cdiv(M, DS_TILE);
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);
    hipLaunchKernelGGL((implicit_decode_split_kernel<1, 2, true>), dim3(grid), dim3(256), 0, gn_stream(stream), p);
    GN_LAUNCH_CHECK("gn_implicit_decode_lattice_split");
    return GN_OK;
}

static int implicit_decode_split_impl(const float *xin, int ldxin, int64_t M, int B, const void *wpack, const float *tab, const float *xscale,
                                      int C0, int N1, int N2, int OUT, float *out, int ldo, void *stream) {
    GN_REQUIRE(M >= 0 && ldo >= OUT && OUT >= 1 && OUT <= 4 && B >= 0 && B <= 65535, "gn_implicit_decode_split: bad sizes");
    const bool wide512 = C0 == 32 && N1 == 512 && N2 == 512;
    GN_REQUIRE(wide512 || ((C0 == 128 || C0 == 32) && N1 == DS_N && N2 == DS_N),
               "gn_implicit_decode_split: only [128 | 32, 256, 256, out] and [32, 512, 512, out] decoders are packed for this kernel (got [%d,%d,%d,%d])", C0, N1, N2, OUT);
    GN_REQUIRE(ldxin >= C0 && ldxin % 4 == 0, "gn_implicit_decode_split: rows need a 16-byte aligned leading dimension");
    if (M == 0 || B == 0) return GN_OK;
    GN_REQUIRE(xin && wpack && tab && out, "gn_implicit_decode_split: null pointer");
    DecSplitArgs p;
    p.xin = xin; p.ldxin = ldxin; p.M = M; p.wp = (const unsigned char *)wpack; p.tab = tab; p.xscale = xscale; p.out = out; p.ldo = ldo;
    p.vol = nullptr; p.D = p.H = p.W = p.Q = 0; p.m0 = 0;
    p.xin_bs = B > 1 ? M * (long long)ldxin : 0; p.out_bs = B > 1 ? M * (long long)ldo : 0; p.xscale_bs = B > 1 ? 4 : 0;
    const int64_t ntiles = gn_cdiv(M, DS_TILE);
    // persistent workgroups: two per CU when they fit (K0G = 2), else one; a batch shares the slots between its row sets (grid.y)
    const int64_t slots0 = (C0 == 32 && OUT == 1) ? 512 : 256, slots = slots0 / B > 0 ? slots0 / B : 1;
    const dim3 grid((unsigned)(ntiles < slots ? ntiles : slots), (unsigned)B);
    hipStream_t st = gn_stream(stream);
    if (wide512) {
        const int64_t s512 = 256 / B > 0 ? 256 / B : 1;
        const dim3 g512((unsigned)(ntiles < s512 ? ntiles : s512), (unsigned)B);
        switch (OUT) {
            case 1: hipLaunchKernelGGL((implicit_decode_split512_kernel<1>), g512, dim3(256), 0, st, p); break;
            case 2: hipLaunchKernelGGL((implicit_decode_split512_kernel<2>), g512, dim3(256), 0, st, p); break;
            case 3: hipLaunchKernelGGL((implicit_decode_split512_kernel<3>), g512, dim3(256), 0, st, p); break;
            default: hipLaunchKernelGGL((implicit_decode_split512_kernel<4>), g512, dim3(256), 0, st, p); break;
        }
        GN_LAUNCH_CHECK("gn_implicit_decode_split");
        return GN_OK;
    }
#define DS_LAUNCH(O)                                                                                                           \
    do {                                                                                                                       \
        if (C0 == 128) hipLaunchKernelGGL((implicit_decode_split_kernel<O, 8>), grid, dim3(256), 0, st, p);                    \
        else hipLaunchKernelGGL((implicit_decode_split_kernel<O, 2>), grid, dim3(256), 0, st, p);                              \
    } while (0)
    switch (OUT) {
        case 1: DS_LAUNCH(1); break;
        case 2: DS_LAUNCH(2); break;
        case 3: DS_LAUNCH(3); break;
        default: DS_LAUNCH(4); break;
    }
#undef DS_LAUNCH
    GN_LAUNCH_CHECK("gn_implicit_decode_split");
    return GN_OK;
}

extern "C" int gn_implicit_decode_split(const float *xin, int ldxin, int64_t M, const void *wpack, const float *tab, const float *xscale,
                                        int C0, int N1, int N2, int OUT, float *out, int ldo, void *stream) {
    return implicit_decode_split_impl(xin, ldxin, M, 1, wpack, tab, xscale, C0, N1, N2, OUT, out, ldo, stream);
}

// B row sets of M rows each in ONE launch (blockIdx.y = row set): xin [B][M][ldxin], out [B][M][ldo], xscale NULL or [B][4] (row set b's input scale record
// of gn_decoder_input_scale; a set marked unsafe is left to gn_implicit_decode_batch(run_if = xscale + 2, stride 4)).  What the surface decoders of
// predict.py:184-187 need for a batch: 16 launches of 384 tiles on 256 persistent workgroups (1.5 rounds each) become one of 6144 tiles (24 rounds).
extern "C" int gn_implicit_decode_split_batch(const float *xin, int ldxin, int64_t M, int B, const void *wpack, const float *tab, const float *xscale,
                                              int C0, int N1, int N2, int OUT, float *out, int ldo, void *stream) {
    return implicit_decode_split_impl(xin, ldxin, M, B, wpack, tab, xscale, C0, N1, N2, OUT, out, ldo, stream);
}
