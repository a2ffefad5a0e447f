// unet.hip -- 3-D UNet building blocks on channel-last volumes [B][D][H][W][C] for gfx950.
//   gn_channel_stats + gn_groupnorm_affine : nn.GroupNorm statistics -> per-(sample,channel) affine
//   gn_conv3d_gcr  : fused GN-apply + Conv3d 3x3x3 (pad 1, no bias) + ReLU as an LDS-tiled implicit GEMM on
//                    v_mfma_f32_32x32x2_f32, with nearest-upsample + channel-concat folded into the loader
//   gn_maxpool3d_2
// Reference: /root/reference/components/unet3d.py:19-144 (create_conv / SingleConv / DoubleConv 'gcr'),
// :195-330 (Encoder / Decoder / Upsampling), :449-474 (forward).
#include "common.h"
#include <string.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------ channel stats
// grid (chunks, B).  Thread t owns channel t % C (C | 256) or channels t, t+256, ... (256 | C) and walks the
// chunk's voxels; per-thread fp32 partials over <= 64 voxels, then fp64 atomics into [B][C].
#define STATS_VOX_PER_BLOCK 512
__global__ __launch_bounds__(256) void channel_stats_kernel(const float *__restrict__ x, int64_t V, int C,
                                                            double *__restrict__ sum, double *__restrict__ sumsq) {
    const int b = blockIdx.y;
    const int64_t v0 = (int64_t)blockIdx.x * STATS_VOX_PER_BLOCK;
    int64_t v1 = v0 + STATS_VOX_PER_BLOCK;
    if (v1 > V) v1 = V;
    const float *xb = x + (int64_t)b * V * C;
    if (C <= 256) {
        const int c = threadIdx.x % C, g = threadIdx.x / C, ng = 256 / C;
        double s = 0.0, q = 0.0;
        float fs = 0.f, fq = 0.f;
        int cnt = 0;
        for (int64_t v = v0 + g; v < v1; v += ng) {
            float t = xb[v * C + c];
            fs += t;
            fq = fmaf(t, t, fq);
            if (++cnt == 32) { s += fs; q += fq; fs = fq = 0.f; cnt = 0; }
        }
        s += fs; q += fq;
        atomicAdd(&sum[(int64_t)b * C + c], s);
        atomicAdd(&sumsq[(int64_t)b * C + c], q);
    } else {
        for (int c = threadIdx.x; c < C; c += 256) {
            double s = 0.0, q = 0.0;
            for (int64_t v = v0; v < v1; ++v) {
                float t = xb[v * C + c];
                s += t;
                q += (double)t * t;
            }
            atomicAdd(&sum[(int64_t)b * C + c], s);
            atomicAdd(&sumsq[(int64_t)b * C + c], q);
        }
    }
}

extern "C" int gn_channel_stats(const float *x, int B, int64_t V, int C, double *sum, double *sumsq, void *stream) {
    GN_REQUIRE(B >= 0 && V >= 0 && C > 0, "gn_channel_stats: bad sizes");
    GN_REQUIRE((C <= 256 && 256 % C == 0) || (C % 256 == 0), "gn_channel_stats: C=%d must divide 256 or be a multiple of it", C);
    hipStream_t st = gn_stream(stream);
    GN_HIP(hipMemsetAsync(sum, 0, sizeof(double) * (size_t)B * C, st), "gn_channel_stats");
    GN_HIP(hipMemsetAsync(sumsq, 0, sizeof(double) * (size_t)B * C, st), "gn_channel_stats");
    if (B == 0 || V == 0) return GN_OK;
    hipLaunchKernelGGL(channel_stats_kernel, dim3((unsigned)gn_cdiv(V, STATS_VOX_PER_BLOCK), B), dim3(256), 0, st, x, V, C, sum, sumsq);
    GN_LAUNCH_CHECK("gn_channel_stats");
    return GN_OK;
}

// one thread per (sample, group)
__global__ void groupnorm_affine_kernel(const double *__restrict__ sum0, const double *__restrict__ sq0, int C0, int64_t V0,
                                        const double *__restrict__ sum1, const double *__restrict__ sq1, int C1, int64_t V1,
                                        int rep1, int B, int groups, float eps, const float *__restrict__ gamma,
                                        const float *__restrict__ beta, float *__restrict__ a, float *__restrict__ d) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * groups) return;
    const int b = t / groups, g = t % groups, C = C0 + C1, cpg = C / groups;
    double s = 0.0, q = 0.0;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        if (c < C0) { s += sum0[(int64_t)b * C0 + c]; q += sq0[(int64_t)b * C0 + c]; }
        else { s += rep1 * sum1[(int64_t)b * C1 + c - C0]; q += rep1 * sq1[(int64_t)b * C1 + c - C0]; }
    }
    const double n = (double)cpg * (double)V0;  // V0 == V1*rep1 voxels per channel after upsampling
    const double mean = s / n;
    double var = q / n - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float fmean = (float)mean;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
        float ga = gamma[c] * rstd;
        a[(int64_t)b * C + c] = ga;
        d[(int64_t)b * C + c] = beta[c] - fmean * ga;
    }
}

extern "C" int gn_groupnorm_affine(const double *sum0, const double *sq0, int C0, int64_t V0, const double *sum1, const double *sq1,
                                   int C1, int64_t V1, int rep1, int B, int groups, float eps, const float *gamma,
                                   const float *beta, float *a, float *d, void *stream) {
    GN_REQUIRE(B >= 0 && groups > 0 && C0 > 0 && C1 >= 0 && (C0 + C1) % groups == 0, "gn_groupnorm_affine: bad sizes");
    GN_REQUIRE(C1 == 0 || V1 * rep1 == V0, "gn_groupnorm_affine: source 1 must cover the same voxels after replication");
    if (B == 0) return GN_OK;
    int n = B * groups;
    hipLaunchKernelGGL(groupnorm_affine_kernel, dim3((unsigned)gn_cdiv(n, 64)), dim3(64), 0, gn_stream(stream), sum0, sq0, C0, V0, sum1,
                       sq1, C1, V1, rep1, B, groups, eps, gamma, beta, a, d);
    GN_LAUNCH_CHECK("gn_groupnorm_affine");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ conv3d 'gcr'
// Implicit GEMM: M = output voxels, N = Cout, K = 27 taps x Cin.
// Block = 256 threads (4 waves) -> output tile 4(z) x 8(y) x 8(x) voxels x (NT*32) output channels;
// wave w owns z-slice w: 64 voxels = two 32-row MFMA fragments (y 0-3 / 4-7, x 0-7), NT column fragments.
// K loop: for each 16-channel input slice { stage the 6x10x10 halo of that slice in LDS (GroupNorm affine applied
// on the way, zeros outside the volume, source 1 read at half resolution = nearest upsampling) ; for each of the
// 27 taps { stage W[tap][slice][:] (16 x Cout_tile) in LDS ; 8 k-steps of 2*NT MFMAs } }.
// LDS: halo 600 voxels x 17 words (odd stride: the 32 rows of a fragment hit distinct banks up to a 2-way
// overlap) = 40.8 KB, weights 16 x NT*32 words <= 8 KB.
#define CV_TZ 4
#define CV_TY 8
#define CV_TX 8
#define CV_HZ (CV_TZ + 2)
#define CV_HY (CV_TY + 2)
#define CV_HX (CV_TX + 2)
#define CV_HVOX (CV_HZ * CV_HY * CV_HX)
#define CV_KS 16
#define CV_VSTRIDE (CV_KS + 1)

struct ConvArgs {
    const float *src0;
    const float *src1;
    const float *a;
    const float *d;
    const float *wp;
    float *out;
    int C0, C1, B, D, H, W, Cout, relu;
    int tiles_y, tiles_x;
};

template <int NT>
__global__ __launch_bounds__(256) void conv3d_gcr_kernel(ConvArgs p) {
    constexpr int CT = NT * 32;
    __shared__ float halo[CV_HVOX * CV_VSTRIDE];
    __shared__ float wsm[CV_KS * CT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Cin = p.C0 + p.C1;
    int tile = blockIdx.x;
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile % p.tiles_y; tile /= p.tiles_y;
    const int tz = tile;
    const int z0 = tz * CV_TZ, y0 = ty * CV_TY, x0 = tx * CV_TX;
    const int n0 = blockIdx.y * CT;
    const int b = blockIdx.z;
    const int D1 = p.D >> 1, H1 = p.H >> 1, W1 = p.W >> 1;

    f32x16 acc[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    // per-lane A-operand base offsets inside the halo for fragment 0/1 at tap (0,0,0)
    const int fi = lane & 31;
    const int abase0 = ((wave * CV_HY + (fi >> 3)) * CV_HX + (fi & 7)) * CV_VSTRIDE + (lane >> 5);
    const int abase1 = abase0 + 4 * CV_HX * CV_VSTRIDE;
    const int bbase = (lane >> 5) * CT + (lane & 31);

    constexpr int WV = (CV_KS * CT / 4 + 255) / 256;  // float4 per thread for a weight tile
    float4 wreg[WV];

    const int nslices = Cin / CV_KS;
    for (int s = 0; s < nslices; ++s) {
        const int c0 = s * CV_KS;
        __syncthreads();  // previous slice fully consumed
        // ---- halo stage
        {
            const bool from1 = c0 >= p.C0;
            const float *src = from1 ? p.src1 : p.src0;
            const int Cs = from1 ? p.C1 : p.C0;
            const int cs = from1 ? c0 - p.C0 : c0;
            const float *ab = p.a + (int64_t)b * Cin + c0;
            const float *db = p.d + (int64_t)b * Cin + c0;
            for (int idx = tid; idx < CV_HVOX * 4; idx += 256) {
                const int hv = idx >> 2, c4 = (idx & 3) * 4;
                const int hx = hv % CV_HX, hy = (hv / CV_HX) % CV_HY, hz = hv / (CV_HX * CV_HY);
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                    int64_t off;
                    if (from1) off = ((((int64_t)b * D1 + (gz >> 1)) * H1 + (gy >> 1)) * W1 + (gx >> 1)) * Cs + cs + c4;
                    else off = ((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * Cs + cs + c4;
                    const float4 xin = *reinterpret_cast<const float4 *>(src + off);
                    const float4 av = *reinterpret_cast<const float4 *>(ab + c4);
                    const float4 dv = *reinterpret_cast<const float4 *>(db + c4);
                    v.x = __fadd_rn(__fmul_rn(xin.x, av.x), dv.x);
                    v.y = __fadd_rn(__fmul_rn(xin.y, av.y), dv.y);
                    v.z = __fadd_rn(__fmul_rn(xin.z, av.z), dv.z);
                    v.w = __fadd_rn(__fmul_rn(xin.w, av.w), dv.w);
                }
                float *h = halo + hv * CV_VSTRIDE + c4;
                h[0] = v.x; h[1] = v.y; h[2] = v.z; h[3] = v.w;
            }
        }
        // ---- first weight tile of this slice
        const float *wbase = p.wp + (int64_t)c0 * p.Cout + n0;
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            int idx = tid + i * 256;
            if (idx < CV_KS * CT / 4) {
                int k = idx / (CT / 4), n4 = (idx % (CT / 4)) * 4;
                wreg[i] = *reinterpret_cast<const float4 *>(wbase + (int64_t)k * p.Cout + n4);
            }
        }
        for (int tap = 0; tap < 27; ++tap) {
            __syncthreads();  // weight buffer free (and halo visible on the first tap)
#pragma unroll
            for (int i = 0; i < WV; ++i) {
                int idx = tid + i * 256;
                if (idx < CV_KS * CT / 4) *reinterpret_cast<float4 *>(wsm + idx * 4) = wreg[i];
            }
            __syncthreads();
            if (tap + 1 < 27) {
                const float *wn = wbase + (int64_t)(tap + 1) * Cin * p.Cout;
#pragma unroll
                for (int i = 0; i < WV; ++i) {
                    int idx = tid + i * 256;
                    if (idx < CV_KS * CT / 4) {
                        int k = idx / (CT / 4), n4 = (idx % (CT / 4)) * 4;
                        wreg[i] = *reinterpret_cast<const float4 *>(wn + (int64_t)k * p.Cout + n4);
                    }
                }
            }
            const int dz = tap / 9, dy = (tap / 3) % 3, dx = tap % 3;
            const int toff = ((dz * CV_HY + dy) * CV_HX + dx) * CV_VSTRIDE;
#pragma unroll
            for (int kk = 0; kk < CV_KS / 2; ++kk) {
                const float a0 = halo[abase0 + toff + kk * 2];
                const float a1 = halo[abase1 + toff + kk * 2];
                float bv[NT];
#pragma unroll
                for (int u = 0; u < NT; ++u) bv[u] = wsm[bbase + kk * 2 * CT + u * 32];
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[u], acc[0][u], 0, 0, 0);
                    acc[1][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[u], acc[1][u], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue
    const int gz = z0 + wave;
    if (gz < p.D) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u) {
                const int n = n0 + u * 32 + (lane & 31);
                if (n >= p.Cout) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    const int gy = y0 + t * 4 + (i >> 3), gx = x0 + (i & 7);
                    if (gy < p.H && gx < p.W) {
                        float v = acc[t][u][r];
                        if (p.relu) v = fmaxf(v, 0.f);
                        p.out[((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * p.Cout + n] = v;
                    }
                }
            }
    }
}

// tuning knob (gn_set_tunable("conv_nt4", 0|1)): use the 128-wide output-channel tile for Cout % 128 == 0.
// Measured on MI355X (profiles/r01_ab_conv_tile.txt): the 64-wide tile (96 accumulator registers, 2 waves/SIMD) runs at
// 128 TFLOP/s, the 128-wide one (304 registers, 1 wave/SIMD, spills) at 93 TFLOP/s -> default off.
static int g_conv_nt4 = 0;
extern "C" int gn_set_tunable(const char *name, int value) {
    if (!name) return GN_EINVAL;
    if (!strcmp(name, "conv_nt4")) { g_conv_nt4 = value; return GN_OK; }
    gn_set_error("gn_set_tunable: unknown tunable %s", name);
    return GN_EINVAL;
}

extern "C" int gn_conv3d_gcr(const float *src0, int C0, const float *src1, int C1, const float *a, const float *d,
                             const float *wp, int B, int D, int H, int W, int Cout, int relu, float *out, void *stream) {
    GN_REQUIRE(B >= 0 && D > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, "gn_conv3d_gcr: bad sizes");
    GN_REQUIRE(C0 % CV_KS == 0 && C1 % CV_KS == 0, "gn_conv3d_gcr: channel counts must be multiples of %d (C0=%d C1=%d)", CV_KS, C0, C1);
    GN_REQUIRE(Cout % 32 == 0, "gn_conv3d_gcr: Cout=%d must be a multiple of 32", Cout);
    GN_REQUIRE(C1 == 0 || (src1 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0), "gn_conv3d_gcr: upsampled source needs even dims");
    if (B == 0) return GN_OK;
    ConvArgs p;
    p.src0 = src0; p.src1 = src1; p.a = a; p.d = d; p.wp = wp; p.out = out;
    p.C0 = C0; p.C1 = C1; p.B = B; p.D = D; p.H = H; p.W = W; p.Cout = Cout; p.relu = relu;
    const int tz = (int)gn_cdiv(D, CV_TZ);
    p.tiles_y = (int)gn_cdiv(H, CV_TY);
    p.tiles_x = (int)gn_cdiv(W, CV_TX);
    const int tiles = tz * p.tiles_y * p.tiles_x;
    hipStream_t st = gn_stream(stream);
    if (Cout % 128 == 0 && g_conv_nt4) hipLaunchKernelGGL(conv3d_gcr_kernel<4>, dim3(tiles, Cout / 128, B), dim3(256), 0, st, p);
    else if (Cout % 64 == 0) hipLaunchKernelGGL(conv3d_gcr_kernel<2>, dim3(tiles, Cout / 64, B), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv3d_gcr_kernel<1>, dim3(tiles, Cout / 32, B), dim3(256), 0, st, p);
    GN_LAUNCH_CHECK("gn_conv3d_gcr");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ maxpool 2x2x2
__global__ __launch_bounds__(256) void maxpool3d_2_kernel(const float *__restrict__ in, int B, int D, int H, int W, int C,
                                                          float *__restrict__ out) {
    const int Do = D >> 1, Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = (int64_t)B * Do * Ho * Wo * C4;
    if (t >= total) return;
    const int c4 = (int)(t % C4); t /= C4;
    const int x = (int)(t % Wo); t /= Wo;
    const int y = (int)(t % Ho); t /= Ho;
    const int z = (int)(t % Do);
    const int b = (int)(t / Do);
    float4 m = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const float4 v = *reinterpret_cast<const float4 *>(
                    in + ((((int64_t)b * D + 2 * z + dz) * H + 2 * y + dy) * W + 2 * x + dx) * C + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
    *reinterpret_cast<float4 *>(out + ((((int64_t)b * Do + z) * Ho + y) * Wo + x) * C + c4 * 4) = m;
}

extern "C" int gn_maxpool3d_2(const float *in, int B, int D, int H, int W, int C, float *out, void *stream) {
    GN_REQUIRE(B >= 0 && D >= 2 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "gn_maxpool3d_2: bad sizes");
    const int64_t total = (int64_t)B * (D / 2) * (H / 2) * (W / 2) * (C / 4);
    if (total == 0) return GN_OK;
    hipLaunchKernelGGL(maxpool3d_2_kernel, dim3((unsigned)gn_cdiv(total, 256)), dim3(256), 0, gn_stream(stream), in, B, D, H, W, C, out);
    GN_LAUNCH_CHECK("gn_maxpool3d_2");
    return GN_OK;
}
