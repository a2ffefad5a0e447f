// upconv.hip -- gn_upconv_partial: the nearest-upsampled source of a decoder convolution, in polyphase form.
//
// Reference: /root/reference/components/unet3d.py:291,330 (x = interpolate(x, size, 'nearest'); x = cat((skip, x)); then the 'gcr'
// SingleConv).  For the upsampled channels every fine output voxel (2i + pz, 2j + py, 2k + px) only sees a 2 x 2 x 2 block of COARSE
// voxels: a fine tap d in {-1, 0, +1} lands on coarse offset {-1, 0, 0} for an even coordinate and {0, 0, +1} for an odd one, so the 27
// fine taps merge (sums of weights, done once on the host in fp64) into 8 coarse taps per output parity class -- 8/27 of the MACs,
// exact algebra.  This kernel computes that part for all 8 classes from ONE staged coarse halo and writes it to a partial buffer
// [B][D/2][H/2][W/2][8 classes][Cout]; the fine launch over the full-resolution source (gn_conv3d_gcr_split, `partial`) adds it in its
// epilogue before the ReLU.
//
// Structure: 512 threads = 8 waves = the 8 parity classes.  A workgroup owns a coarse tile TZC x 8 x 8 (TZC = 2 for 32 output channels
// per class, 1 for 64: 64 + 64 accumulator registers either way) x 32*NT output channels of every class.  Per 16-channel slice the coarse
// halo ((TZC+2) x 10 x 10 voxels: GroupNorm affine + exact split into two fp16 planes, the arithmetic of unet_split.hip) is staged
// ONCE for all classes -- 1/8 of the voxels the literal form converts -- double-buffered, slice s+1 during slice s; a class's tap
// (iz, iy, ix) in {0,1}^3 reads the halo at offset (pz + iz, py + iy, px + ix).  Each wave multiplies its own class's merged weights:
// B fragments go global -> registers (1 KB per wave per fragment, coalesced, L2-resident), one step ahead; nothing is shared between
// waves but the halo, so the only workgroup barrier is one per slice.
#include "common.h"

typedef float f32x16u __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8u __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8u __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2u __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2u __attribute__((ext_vector_type(2)));
typedef float f32x2u __attribute__((ext_vector_type(2)));

struct UpArgs {
    const float *src;             // coarse source [B][Dc][Hc][Wc][C1]
    const float *a, *d;           // [B][C1] GroupNorm affine of these channels (sample scale folded in)
    const uint4 *wp;              // [C1/16 slices][8 taps][8 classes][Cout/32][2 planes][64 lanes] x 16 B (ops.pack_upconv_weight)
    const float *out_scale;       // [8 * Cout] exact powers of two undoing the per-row weight scales
    const float *act_inv;         // NULL or [B]
    float *partial;               // [B][Dc][Hc][Wc][8 * Cout]
    int C1, B, Dc, Hc, Wc, Cout;
    int tiles_y, tiles_x;
};

template <bool F16>
__device__ __forceinline__ void up_split4(float r0, float r1, float r2, float r3, uint2 (&out)[2]) {
    const f32x2u lo = {r0, r1}, hi = {r2, r3};
    if (F16) {
        const f16x2u blo = __builtin_convertvector(lo, f16x2u), bhi = __builtin_convertvector(hi, f16x2u);
        out[0].x = __builtin_bit_cast(unsigned, blo);
        out[0].y = __builtin_bit_cast(unsigned, bhi);
        const f32x2u rlo = {gn_resid_lo(out[0].x, r0), gn_resid_hi(out[0].x, r1)}, rhi = {gn_resid_lo(out[0].y, r2), gn_resid_hi(out[0].y, r3)};   // (common.h)
        out[1].x = __builtin_bit_cast(unsigned, __builtin_convertvector(rlo, f16x2u));
        out[1].y = __builtin_bit_cast(unsigned, __builtin_convertvector(rhi, f16x2u));
    } else {
        const bf16x2u blo = __builtin_convertvector(lo, bf16x2u), bhi = __builtin_convertvector(hi, bf16x2u);
        out[0].x = __builtin_bit_cast(unsigned, blo);
        out[0].y = __builtin_bit_cast(unsigned, bhi);
        const f32x2u rlo = {__fsub_rn(r0, __uint_as_float(out[0].x << 16)), __fsub_rn(r1, __uint_as_float(out[0].x & 0xffff0000u))};
        const f32x2u rhi = {__fsub_rn(r2, __uint_as_float(out[0].y << 16)), __fsub_rn(r3, __uint_as_float(out[0].y & 0xffff0000u))};
        out[1].x = __builtin_bit_cast(unsigned, __builtin_convertvector(rlo, bf16x2u));
        out[1].y = __builtin_bit_cast(unsigned, __builtin_convertvector(rhi, bf16x2u));
    }
}

template <bool F16>
__device__ __forceinline__ f32x16u up_mfma(const uint4 &a, const uint4 &b, const f32x16u &c) {
    if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8u, a), __builtin_bit_cast(f16x8u, b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8u, a), __builtin_bit_cast(bf16x8u, b), c, 0, 0, 0);
}

// halo layout of unet_split.hip (two planes: 64 B per voxel, one 16-byte pad per row of 10 voxels: conflict-free ds_read_b128)
template <int HZ> struct UpHalo {
    static constexpr int ROWP = 10 * 64 + 16;
    static constexpr int BYTES = HZ * 10 * ROWP;
    __device__ static constexpr __forceinline__ int at(int hz, int hy, int hx) { return (hz * 10 + hy) * ROWP + hx * 64; }
};

template <int NT, bool F16>
__global__ __launch_bounds__(512, 1) void upconv_partial_kernel(UpArgs p) {
    constexpr int TZC = NT == 1 ? 2 : 1, NF = 2 * TZC, HZ = TZC + 2, HVOX = HZ * 100;
    using HL = UpHalo<HZ>;
    constexpr int NIT = (HVOX * 4 + 511) / 512;     // float4 row loads per thread per slice (4 / 3)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * HL::BYTES + 2 * 384 * 4];
    float *const adl = reinterpret_cast<float *>(smem + 2 * HL::BYTES);     // a[C1] | d[C1] of this sample
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int cls = __builtin_amdgcn_readfirstlane(tid >> 6), pz = cls >> 2, py = (cls >> 1) & 1, px = cls & 1;
    // XCD-aware order as in unet_split.hip: every XCD walks a contiguous range of (tile, column block) pairs, tiles z-fastest
    const unsigned nblk = gridDim.x, xcd = blockIdx.x & 7u, jx = blockIdx.x >> 3, qx = nblk >> 3, rx = nblk & 7u;
    const unsigned logical = (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + jx;
    const int ncb = p.Cout / (32 * NT);
    int tile = (int)(logical / (unsigned)ncb);
    const int cb = (int)(logical % (unsigned)ncb);
    const int tiles_z = (p.Dc + TZC - 1) / TZC;
    const int tz = tile % tiles_z; tile /= tiles_z;
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile;
    const int z0 = tz * TZC, y0 = ty * 8, x0 = tx * 8;
    const int b = blockIdx.y;
    const int nslices = p.C1 / 16;
    const int nblocks = p.Cout / 32;                // 32-wide column blocks per class

    f32x16u acc[NF][NT], tot[NF][NT];               // fragment f = 2 zc + t: coarse z-slice zc, y half t
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc[f][u][q] = 0.f; tot[f][u][q] = 0.f; }
    for (int i = tid; i < p.C1; i += 512) { adl[i] = p.a[(int64_t)b * p.C1 + i]; adl[384 + i] = p.d[(int64_t)b * p.C1 + i]; }

    // ---- staging of one 16-channel slice of the coarse halo (all classes share it)
    const int c4 = (tid & 3) * 4;
    float4 raw[NIT];
    unsigned inb = 0;
    auto issue_rows = [&](int sl) {
        inb = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tid + it * 512;
            const int hv = (idx < HVOX * 4 ? idx : HVOX * 4 - 1) >> 2;
            const int hx = hv % 10, hy = (hv / 10) % 10, hz = hv / 100;
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool in = idx < HVOX * 4 && gz >= 0 && gz < p.Dc && gy >= 0 && gy < p.Hc && gx >= 0 && gx < p.Wc;
            raw[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in) {
                raw[it] = *reinterpret_cast<const float4 *>(p.src + ((((int64_t)b * p.Dc + gz) * p.Hc + gy) * p.Wc + gx) * p.C1 + sl * 16 + c4);
                inb |= 1u << it;
            }
        }
    };
    auto convert_row = [&](int it, int sl, int buf) {
        const int idx = tid + it * 512;
        if (idx < HVOX * 4) {
            const int hv = idx >> 2;
            const float4 av = *reinterpret_cast<const float4 *>(adl + sl * 16 + c4);
            const float4 dv = *reinterpret_cast<const float4 *>(adl + 384 + sl * 16 + c4);
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
            if (inb & (1u << it)) {                  // zero padding comes AFTER the affine
                v0 = __fmaf_rn(raw[it].x, av.x, dv.x);
                v1 = __fmaf_rn(raw[it].y, av.y, dv.y);
                v2 = __fmaf_rn(raw[it].z, av.z, dv.z);
                v3 = __fmaf_rn(raw[it].w, av.w, dv.w);
            }
            uint2 pl[2];
            up_split4<F16>(v0, v1, v2, v3, pl);
            unsigned char *dst = smem + buf * HL::BYTES + HL::at(hv / 100, (hv / 10) % 10, hv % 10) + c4 * 2;
            *reinterpret_cast<uint2 *>(dst) = pl[0];
            *reinterpret_cast<uint2 *>(dst + 32) = pl[1];
        }
    };

    issue_rows(0);
    __syncthreads();                                // a / d table visible
#pragma unroll
    for (int it = 0; it < NIT; ++it) convert_row(it, 0, 0);
    __syncthreads();

    // this wave's B fragments: step (slice, tap) -> [class][column block cb*NT + u][plane][lane]
    const int64_t bstep = (int64_t)8 * nblocks * 2 * 64;                    // uint4 per (slice, tap) step
    const uint4 *bw = p.wp + ((int64_t)cls * nblocks + cb * NT) * 2 * 64 + lane;
    uint4 bf[NT][2], nbf[NT][2], nnbf[NT][2];       // B fragments two steps ahead: a step is 12 MFMAs (~400 cycles), an L2 hit under load is more
#pragma unroll
    for (int u = 0; u < NT; ++u) { nbf[u][0] = bw[(u * 2 + 0) * 64]; nbf[u][1] = bw[(u * 2 + 1) * 64]; }
    bw += bstep;
#pragma unroll
    for (int u = 0; u < NT; ++u) { nnbf[u][0] = bw[(u * 2 + 0) * 64]; nnbf[u][1] = bw[(u * 2 + 1) * 64]; }   // (step 1: C1 >= 16 -> at least 8 steps)
    bw += bstep;
    const int abase = HL::at(pz, (r >> 3) + py, (r & 7) + px) + 16 * h;       // class offset folded in; tap (iz, iy, ix) adds HL::at(iz, iy, ix)
    constexpr int AF1 = 4 * HL::ROWP, AFZ = 10 * HL::ROWP;                   // y half, next coarse z-slice
    const int nsteps = nslices * 8;
    int step = 0;
    uint4 af[NF][2], naf[NF][2];                    // A fragments double-buffered one tap ahead (across the slice boundary too)
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        naf[f][0] = *reinterpret_cast<const uint4 *>(smem + abase + (f >> 1) * AFZ + (f & 1) * AF1);
        naf[f][1] = *reinterpret_cast<const uint4 *>(smem + abase + (f >> 1) * AFZ + (f & 1) * AF1 + 32);
    }
    for (int s = 0; s < nslices; ++s) {
        const unsigned char *const halo = smem + (s & 1) * HL::BYTES, *const halo_n = smem + ((s + 1) & 1) * HL::BYTES;
        const bool more = s + 1 < nslices;
        if (more) issue_rows(s + 1);
#pragma unroll
        for (int tap = 0; tap < 8; ++tap, ++step) {
#pragma unroll
            for (int u = 0; u < NT; ++u) { bf[u][0] = nbf[u][0]; bf[u][1] = nbf[u][1]; nbf[u][0] = nnbf[u][0]; nbf[u][1] = nnbf[u][1]; }
#pragma unroll
            for (int f = 0; f < NF; ++f) { af[f][0] = naf[f][0]; af[f][1] = naf[f][1]; }
            if (step + 2 < nsteps) {
#pragma unroll
                for (int u = 0; u < NT; ++u) { nnbf[u][0] = bw[(u * 2 + 0) * 64]; nnbf[u][1] = bw[(u * 2 + 1) * 64]; }
                bw += bstep;
            }
            if (tap < 7) {                           // next tap of this slice (tap 7 prefetches after the slice barrier, below)
                const int toff = HL::at((tap + 1) >> 2, ((tap + 1) >> 1) & 1, (tap + 1) & 1);
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    naf[f][0] = *reinterpret_cast<const uint4 *>(halo + abase + toff + (f >> 1) * AFZ + (f & 1) * AF1);
                    naf[f][1] = *reinterpret_cast<const uint4 *>(halo + abase + toff + (f >> 1) * AFZ + (f & 1) * AF1 + 32);
                }
            }
#define UP_PROD(IA, IB)                                                                                                        \
            _Pragma("unroll") for (int u = 0; u < NT; ++u)                                                                     \
                _Pragma("unroll") for (int f = 0; f < NF; ++f) acc[f][u] = up_mfma<F16>(af[f][IA], bf[u][IB], acc[f][u]);
            UP_PROD(1, 0)
            if (more && tap >= 2 && tap < 2 + NIT) convert_row(tap - 2, s + 1, (s + 1) & 1);
            UP_PROD(0, 1) UP_PROD(0, 0)
#undef UP_PROD
        }
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int q = 0; q < 16; ++q) { tot[f][u][q] = __fadd_rn(tot[f][u][q], acc[f][u][q]); acc[f][u][q] = 0.f; }
        __syncthreads();                            // slice s+1 staged and visible; everybody is done reading slice s's buffer
        if (more) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                naf[f][0] = *reinterpret_cast<const uint4 *>(halo_n + abase + (f >> 1) * AFZ + (f & 1) * AF1);
                naf[f][1] = *reinterpret_cast<const uint4 *>(halo_n + abase + (f >> 1) * AFZ + (f & 1) * AF1 + 32);
            }
        }
    }
    // ---- epilogue: partial sums in true units (weight-row and sample scales undone), class-blocked channel order
    const int64_t pc = (int64_t)8 * p.Cout;
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int t = f & 1, gz = z0 + (f >> 1);
            const int n = cls * p.Cout + (cb * NT + u) * 32 + r;
            const float osc = p.act_inv ? __fmul_rn(p.out_scale[n], p.act_inv[b]) : p.out_scale[n];
            if (z0 + TZC <= p.Dc && y0 + 8 <= p.Hc && x0 + 8 <= p.Wc) {       // tile inside the volume (workgroup-uniform): one 64-bit
                                                                              // address per fragment, uniform offsets for its 16 voxels
                float *ob = p.partial + ((((int64_t)b * p.Dc + gz) * p.Hc + (y0 + t * 4)) * p.Wc + (x0 + 4 * h)) * pc + n;
                const int64_t rs = (int64_t)p.Wc * pc;
#pragma unroll
                for (int q = 0; q < 16; ++q) ob[(q >> 2) * rs + (q & 3) * pc] = __fmul_rn(tot[f][u][q], osc);
                continue;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * h;
                const int gy = y0 + t * 4 + (i >> 3), gx = x0 + (i & 7);
                if (gz < p.Dc && gy < p.Hc && gx < p.Wc)
                    p.partial[((((int64_t)b * p.Dc + gz) * p.Hc + gy) * p.Wc + gx) * pc + n] = __fmul_rn(tot[f][u][q], osc);
            }
        }
}

extern "C" int gn_upconv_partial(const float *src1, int C1, const float *a, const float *d, const void *wp, int mode, const float *out_scale,
                                 const float *act_inv_scale, int B, int Dc, int Hc, int Wc, int Cout, float *partial, void *stream) {
    GN_REQUIRE(B >= 0 && Dc > 0 && Hc > 0 && Wc > 0 && C1 > 0 && Cout > 0, "gn_upconv_partial: bad sizes");
    GN_REQUIRE(mode == GN_SPLIT_BF16X2 || mode == GN_SPLIT_F16X2, "gn_upconv_partial: two-plane modes only (GN_SPLIT_BF16X2 / GN_SPLIT_F16X2)");
    GN_REQUIRE(C1 % 16 == 0 && C1 <= 384 && Cout % 32 == 0, "gn_upconv_partial: channels must be multiples of 16 (in, <= 384) / 32 (out)");
    if (B == 0) return GN_OK;
    GN_REQUIRE(src1 && a && d && wp && out_scale && partial, "gn_upconv_partial: null pointer");
    UpArgs p;
    p.src = src1; p.a = a; p.d = d; p.wp = (const uint4 *)wp; p.out_scale = out_scale; p.act_inv = act_inv_scale; p.partial = partial;
    p.C1 = C1; p.B = B; p.Dc = Dc; p.Hc = Hc; p.Wc = Wc; p.Cout = Cout;
    p.tiles_y = (int)gn_cdiv(Hc, 8); p.tiles_x = (int)gn_cdiv(Wc, 8);
    hipStream_t st = gn_stream(stream);
    const bool f16 = mode == GN_SPLIT_F16X2;
    if (Cout % 64 == 0) {
        const unsigned g = (unsigned)(gn_cdiv(Dc, 1) * p.tiles_y * p.tiles_x * (Cout / 64));
        if (f16) hipLaunchKernelGGL((upconv_partial_kernel<2, true>), dim3(g, B), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((upconv_partial_kernel<2, false>), dim3(g, B), dim3(512), 0, st, p);
    } else {
        const unsigned g = (unsigned)(gn_cdiv(Dc, 2) * p.tiles_y * p.tiles_x * (Cout / 32));
        if (f16) hipLaunchKernelGGL((upconv_partial_kernel<1, true>), dim3(g, B), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((upconv_partial_kernel<1, false>), dim3(g, B), dim3(512), 0, st, p);
    }
    GN_LAUNCH_CHECK("gn_upconv_partial");
    return GN_OK;
}
