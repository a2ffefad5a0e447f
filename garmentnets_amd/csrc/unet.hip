// unet.hip -- 3-D UNet building blocks on channel-last volumes [B][D][H][W][C] for gfx950.
//   gn_channel_stats + gn_groupnorm_affine : nn.GroupNorm statistics -> per-(sample,channel) affine
//   gn_conv3d_gcr  : fused GN-apply + Conv3d 3x3x3 (pad 1, no bias) + ReLU as an LDS-tiled implicit GEMM on
//                    v_mfma_f32_32x32x2_f32, with nearest-upsample + channel-concat folded into the loader
//   gn_maxpool3d_2
// Reference: /root/reference/components/unet3d.py:19-144 (create_conv / SingleConv / DoubleConv 'gcr'),
// :195-330 (Encoder / Decoder / Upsampling), :449-474 (forward).
#include "common.h"

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
    GN_HIP(gn_zero_stats(sum, sumsq, (size_t)B * C, st), "gn_channel_stats");
    if (B == 0 || V == 0) return GN_OK;
    hipLaunchKernelGGL(channel_stats_kernel, dim3((unsigned)gn_cdiv(V, STATS_VOX_PER_BLOCK), B), dim3(256), 0, st, x, V, C, sum, sumsq);
    GN_LAUNCH_CHECK("gn_channel_stats");
    return GN_OK;
}

// one block per sample, one thread per group.  With `act_inv_scale` (split-operand convs): the affine of the whole sample is multiplied
// by a power of two 2^k chosen from the statistics so that the largest per-channel rms of the normalised activations
// y_c = a_c x + d_c (E[y^2] = a^2 E[x^2] + 2 a d E[x] + d^2, all known here) lands in [1, 2); act_inv_scale[b] = 2^-k undoes it exactly
// in the conv epilogue.  fp16's narrow exponent then never matters: a channel's values are bounded by rms * sqrt(V) <= 2 * 2^14.5
// < 65504 for any V < 2^29 voxels (no overflow by construction, whatever the checkpoint), and values down to 1/8 of the typical
// magnitude keep a normal second plane (residual <= 2^-22 |x|; below that the residual is absolute, 2^-25 of the sample's scale).
// Round 6: the statistics of the sample are first brought into LDS by all threads (coalesced), the per-GROUP sums are then formed by one thread per group
// in ascending channel order -- the summation order of the round-1 kernel, bit for bit -- and everything per CHANNEL (a, d, the rms) runs one thread per
// channel.  The thread-per-group form walked up to 48 channels through dependent global loads: 40 us for the 384-channel decoder layer, 12 us typical.
#define GNA_THREADS 256
#define GNA_MAXC 1024
__global__ __launch_bounds__(GNA_THREADS) void groupnorm_affine_kernel(const double *__restrict__ sum0, const double *__restrict__ sq0, int C0, int64_t V0,
                                        const double *__restrict__ sum1, const double *__restrict__ sq1, int C1, int64_t V1,
                                        int rep1, int B, int groups, float eps, const float *__restrict__ gamma,
                                        const float *__restrict__ beta, float *__restrict__ a, float *__restrict__ d,
                                        float *__restrict__ act_inv_scale) {
    __shared__ double ls[GNA_MAXC], lq[GNA_MAXC];           // per-channel sum / sum of squares (source 1 already times rep1)
    __shared__ float grstd[GNA_MAXC], gmean[GNA_MAXC];      // per group
    __shared__ float gmax[GNA_THREADS / 64];
    const int b = blockIdx.x, C = C0 + C1, cpg = C / groups, tid = threadIdx.x;
    for (int c = tid; c < C; c += GNA_THREADS) {
        if (c < C0) { ls[c] = sum0[(int64_t)b * C0 + c]; lq[c] = sq0[(int64_t)b * C0 + c]; }
        else { ls[c] = rep1 * sum1[(int64_t)b * C1 + c - C0]; lq[c] = rep1 * sq1[(int64_t)b * C1 + c - C0]; }
    }
    __syncthreads();
    for (int g = tid; g < groups; g += GNA_THREADS) {
        double s = 0.0, q = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s += ls[c]; q += lq[c]; }
        const double n = (double)cpg * (double)V0;  // V0 == V1*rep1 voxels per channel after upsampling
        const double mean = s / n;
        double var = q / n - mean * mean;
        if (var < 0) var = 0;
        grstd[g] = (float)(1.0 / sqrt(var + (double)eps));
        gmean[g] = (float)mean;
    }
    __syncthreads();
    float my_max = 0.f;
    for (int c = tid; c < C; c += GNA_THREADS) {
        const int g = c / cpg;
        const float ga = gamma[c] * grstd[g];
        const float dd = beta[c] - gmean[g] * ga;
        a[(int64_t)b * C + c] = ga;
        d[(int64_t)b * C + c] = dd;
        if (act_inv_scale) {
            const double ex = ls[c] / (double)V0, ex2 = lq[c] / (double)V0;
            const double ey2 = (double)ga * ga * ex2 + 2.0 * (double)ga * dd * ex + (double)dd * dd;
            const float rms = ey2 > 0 ? (float)sqrt(ey2) : 0.f;
            if (rms > my_max) my_max = rms;          // (a NaN statistic compares false: the scale stays finite and the NaN reaches the output)
        }
    }
    if (!act_inv_scale) return;
    for (int off = 32; off >= 1; off >>= 1) my_max = fmaxf(my_max, __shfl_xor(my_max, off));      // (a maximum: any order)
    if ((tid & 63) == 0) gmax[tid >> 6] = my_max;
    __syncthreads();
    const float m = fmaxf(fmaxf(gmax[0], gmax[1]), fmaxf(gmax[2], gmax[3]));
    int e = 0;
    float sc = 1.f, inv = 1.f;
    if (m > 0.f && m < INFINITY) {
        (void)frexpf(m, &e);                                 // m = f * 2^e, f in [0.5, 1)  ->  m * 2^(1-e) in [1, 2)
        e = 1 - e;
        if (e > 100) e = 100;
        if (e < -100) e = -100;
        sc = ldexpf(1.f, e);
        inv = ldexpf(1.f, -e);
    }
    for (int c = tid; c < C; c += GNA_THREADS) { a[(int64_t)b * C + c] *= sc; d[(int64_t)b * C + c] *= sc; }      // (each thread rescales what it wrote)
    if (tid == 0) act_inv_scale[b] = inv;
}

extern "C" int gn_groupnorm_affine(const double *sum0, const double *sq0, int C0, int64_t V0, const double *sum1, const double *sq1,
                                   int C1, int64_t V1, int rep1, int B, int groups, float eps, const float *gamma,
                                   const float *beta, float *a, float *d, float *act_inv_scale, void *stream) {
    GN_REQUIRE(B >= 0 && groups > 0 && C0 > 0 && C1 >= 0 && (C0 + C1) % groups == 0, "gn_groupnorm_affine: bad sizes");
    GN_REQUIRE(C0 + C1 <= GNA_MAXC, "gn_groupnorm_affine: at most %d channels (got %d)", GNA_MAXC, C0 + C1);
    GN_REQUIRE(C1 == 0 || V1 * rep1 == V0, "gn_groupnorm_affine: source 1 must cover the same voxels after replication");
    if (B == 0) return GN_OK;
    hipLaunchKernelGGL(groupnorm_affine_kernel, dim3((unsigned)B), dim3(GNA_THREADS), 0, gn_stream(stream), sum0, sq0, C0, V0, sum1,
                       sq1, C1, V1, rep1, B, groups, eps, gamma, beta, a, d, act_inv_scale);
    GN_LAUNCH_CHECK("gn_groupnorm_affine");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ conv3d 'gcr'
// Implicit GEMM: M = output voxels, N = Cout, K = 27 taps x Cin.
// Block = 256 threads (4 waves) -> output tile 4(z) x 8(y) x 8(x) voxels x (NT*32) output channels;
// wave w owns z-slice w: 64 voxels = two 32-row MFMA fragments (y 0-3 / 4-7, x 0-7), NT column fragments.
// K loop: for each 16-channel input slice { stage the 6x10x10 halo of that slice in LDS (GroupNorm affine applied on the
// way, zeros outside the volume, source 1 read at half resolution = nearest upsampling + concat never materialised);
// for each of the 27 taps { 8 k-steps of 2*NT MFMAs } }.
// k order inside a slice is PERMUTED (sum order is free): MFMA j takes channel j from lanes 0-31 and channel 8+j from lanes
// 32-63, so a lane reads 8 CONSECUTIVE channels of its voxel (2 x ds_read_b128) and 8 consecutive k of its output channel
// from the weight tile (packed [tap][slice][cout][16], staged as wsm[n][16+4]) -- 4x fewer LDS instructions than scalar
// fragment loads.  Pipelining: the weight tile of tap t+1 is fetched into registers during the MFMAs of tap t and written
// to the other half of a double-buffered LDS tile (one barrier per tap); the A fragments of tap t+1 are read during tap t.
// LDS: halo 600 voxels x 20 words = 48 KB, weights 2 x NT*32 x 20 words <= 10 KB  -> 2 workgroups per CU.
// Optional epilogue: per-(sample, channel) sum / sum-of-squares of the (post-ReLU) output, i.e. the GroupNorm statistics
// of the NEXT layer, reduced in-wave, across waves through LDS, then one fp64 atomic per channel per workgroup.
#define CV_TZ 4
#define CV_TY 8
#define CV_TX 8
#define CV_HZ (CV_TZ + 2)
#define CV_HY (CV_TY + 2)
#define CV_HX (CV_TX + 2)
#define CV_HVOX (CV_HZ * CV_HY * CV_HX)
#define CV_KS 16
#define CV_VSTRIDE (CV_KS + 4)

struct ConvArgs {
    const float *src0;
    const float *src1;
    const float *a;
    const float *d;
    const float *wp;
    float *out;
    double *osum;
    double *osq;
    int C0, C1, B, D, H, W, Cout, relu;
    int tiles_y, tiles_x;
};

struct AFrag { float4 lo, hi; };

template <int NT>
__global__ __launch_bounds__(256, 2) void conv3d_gcr_kernel(ConvArgs p) {
    constexpr int CT = NT * 32;
    __shared__ __attribute__((aligned(16))) float halo[CV_HVOX * CV_VSTRIDE];
    __shared__ __attribute__((aligned(16))) float wsm[2][CT * CV_VSTRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r = lane & 31;
    const int Cin = p.C0 + p.C1;
    // XCD-aware work-item order (workgroup b runs on XCD b % 8, each XCD has its own L2): every XCD walks a CONTIGUOUS
    // range of (tile, column block) pairs, column blocks of one tile adjacent, tiles in z-fastest order -> the halo overlap of
    // neighbouring tiles and the second column block's re-read of the same input hit that XCD's L2.  Bijective for any size.
    const unsigned nblk = gridDim.x, xcd = blockIdx.x & 7u, jx = blockIdx.x >> 3, qx = nblk >> 3, rx = nblk & 7u;
    const unsigned logical = (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + jx;
    const int ncb = p.Cout / CT;
    int tile = (int)(logical / (unsigned)ncb);
    const int cb = (int)(logical % (unsigned)ncb);
    // z-fastest tile order: the z halo is the fattest (2 of 6 slices), so tiles adjacent in z run back to back
    const int tiles_z = (p.D + 3) / 4;
    const int tz = tile % tiles_z; tile /= tiles_z;
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile;
    const int z0 = tz * CV_TZ, y0 = ty * CV_TY, x0 = tx * CV_TX;
    const int n0 = cb * CT;
    const int b = blockIdx.y;
    const int D1 = p.D >> 1, H1 = p.H >> 1, W1 = p.W >> 1;

    // two-level summation: `acc` chains the 27 x 16 products of ONE channel slice on the matrix cores, `tot` adds the
    // slice partials on the VALU -- shortens the fp32 rounding chain from 27*Cin to 432 terms (+ Cin/16 adds)
    f32x16 acc[2][NT], tot[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc[t][u][q] = 0.f; tot[t][u][q] = 0.f; }

    // A-operand base (float4 units) of fragment 0 at tap (0,0,0); fragment 1 is 4 halo rows further
    const int abase = (((wave * CV_HY + (r >> 3)) * CV_HX + (r & 7)) * CV_VSTRIDE + 8 * h) >> 2;
    constexpr int AF1 = (4 * CV_HX * CV_VSTRIDE) >> 2;
    const int bbase = (r * CV_VSTRIDE + 8 * h) >> 2;
    const float4 *halo4 = reinterpret_cast<const float4 *>(halo);

    // weight tile: CT x 16 floats contiguous in global ([tap][slice][cout][16]); one float4 per thread (NT=2: 256, NT=1: 128)
    constexpr int WV4 = CT * 4;   // float4 per tile
    const bool wact = tid < WV4;
    const int wn = tid >> 2, wk4 = tid & 3;
    const int nslices = Cin / CV_KS;
    const int64_t tap_stride = (int64_t)nslices * p.Cout * CV_KS;

    for (int s = 0; s < nslices; ++s) {
        const int c0 = s * CV_KS;
        // ---- halo stage (the barrier that ended the previous slice's last tap protects halo and wsm)
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
                *reinterpret_cast<float4 *>(halo + hv * CV_VSTRIDE + c4) = v;
            }
        }
        const float *wslice = p.wp + ((int64_t)s * p.Cout + n0) * CV_KS;   // tap 0 of this slice
        float4 wreg = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wact) {
            wreg = *reinterpret_cast<const float4 *>(wslice + (int64_t)tid * 4);
            *reinterpret_cast<float4 *>(&wsm[0][wn * CV_VSTRIDE + wk4 * 4]) = wreg;
        }
        __syncthreads();
        AFrag a0, a1, na0, na1;
        a0.lo = halo4[abase]; a0.hi = halo4[abase + 1];
        a1.lo = halo4[abase + AF1]; a1.hi = halo4[abase + AF1 + 1];
        na0 = a0; na1 = a1;
        for (int tap = 0; tap < 27; ++tap) {
            const int cur = tap & 1;
            const bool more = tap + 1 < 27;
            if (more && wact) wreg = *reinterpret_cast<const float4 *>(wslice + (int64_t)(tap + 1) * tap_stride + (int64_t)tid * 4);
            AFrag bf[NT];
            const float4 *w4 = reinterpret_cast<const float4 *>(wsm[cur]);
#pragma unroll
            for (int u = 0; u < NT; ++u) { bf[u].lo = w4[bbase + u * 32 * (CV_VSTRIDE >> 2)]; bf[u].hi = w4[bbase + u * 32 * (CV_VSTRIDE >> 2) + 1]; }
            if (more) {
                const int t1 = tap + 1;
                const int toff = ((((t1 / 9) * CV_HY + (t1 / 3) % 3) * CV_HX + t1 % 3) * CV_VSTRIDE) >> 2;
                na0.lo = halo4[abase + toff]; na0.hi = halo4[abase + toff + 1];
                na1.lo = halo4[abase + AF1 + toff]; na1.hi = halo4[abase + AF1 + toff + 1];
            }
            __builtin_amdgcn_sched_barrier(0);
#define CV_STEP(X)                                                                                   \
            _Pragma("unroll") for (int u = 0; u < NT; ++u) {                                         \
                acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.X, bf[u].X, acc[0][u], 0, 0, 0);  \
                acc[1][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.X, bf[u].X, acc[1][u], 0, 0, 0);  \
            }
            CV_STEP(lo.x) CV_STEP(lo.y) CV_STEP(lo.z) CV_STEP(lo.w) CV_STEP(hi.x) CV_STEP(hi.y) CV_STEP(hi.z) CV_STEP(hi.w)
#undef CV_STEP
            __builtin_amdgcn_sched_barrier(0);
            if (more && wact) *reinterpret_cast<float4 *>(&wsm[cur ^ 1][wn * CV_VSTRIDE + wk4 * 4]) = wreg;
            __syncthreads();
            a0 = na0; a1 = na1;
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int q = 0; q < 16; ++q) { tot[t][u][q] = __fadd_rn(tot[t][u][q], acc[t][u][q]); acc[t][u][q] = 0.f; }
    }
    // ---- epilogue: ReLU, coalesced stores, optional statistics of the output
    const int gz = z0 + wave;
    const bool full = z0 + CV_TZ <= p.D && y0 + CV_TY <= p.H && x0 + CV_TX <= p.W;    // tile inside the volume (workgroup-uniform)
    double ssum[NT], ssq[NT];                       // fp64 per lane: the statistics do not depend on the kernel variant (see unet_split.hip)
#pragma unroll
    for (int u = 0; u < NT; ++u) { ssum[u] = 0.0; ssq[u] = 0.0; }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int n = n0 + u * 32 + r;
            if (full) {                              // one 64-bit address per fragment, workgroup-uniform offsets for its 16 voxels (q = 4 j + k:
                                                     // y = j, x = 4 h + k) -- the generic path costs ~40 instructions per value
                float *ob = p.out + ((((int64_t)b * p.D + gz) * p.H + (y0 + t * 4)) * p.W + (x0 + 4 * h)) * p.Cout + n;
                const int64_t rs = (int64_t)p.W * p.Cout;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float v = tot[t][u][q];
                    if (p.relu) v = gn_relu(v);
                    ob[(q >> 2) * rs + (int64_t)(q & 3) * p.Cout] = v;
                    ssum[u] += (double)v;
                    ssq[u] += (double)v * (double)v;
                }
                continue;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * h;
                const int gy = y0 + t * 4 + (i >> 3), gx = x0 + (i & 7);
                if (gz < p.D && gy < p.H && gx < p.W) {
                    float v = tot[t][u][q];
                    if (p.relu) v = gn_relu(v);
                    p.out[((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * p.Cout + n] = v;
                    ssum[u] += (double)v;
                    ssq[u] += (double)v * (double)v;
                }
            }
        }
    if (p.osum) {
        double *red = reinterpret_cast<double *>(halo);   // [2][4 waves][CT]; every wave is past its last halo / weight read (barrier after the last tap)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const double s2 = ssum[u] + __shfl_xor(ssum[u], 32), q2 = ssq[u] + __shfl_xor(ssq[u], 32);
            if (h == 0) { red[wave * CT + u * 32 + r] = s2; red[4 * CT + wave * CT + u * 32 + r] = q2; }
        }
        __syncthreads();
        if (tid < CT) {
            const double s4 = red[tid] + red[CT + tid] + red[2 * CT + tid] + red[3 * CT + tid];
            const double q4 = red[4 * CT + tid] + red[5 * CT + tid] + red[6 * CT + tid] + red[7 * CT + tid];
            atomicAdd(&p.osum[(int64_t)b * p.Cout + n0 + tid], s4);
            atomicAdd(&p.osq[(int64_t)b * p.Cout + n0 + tid], q4);
        }
    }
}

extern "C" int gn_conv3d_gcr(const float *src0, int C0, const float *src1, int C1, const float *a, const float *d,
                             const float *wp, int B, int D, int H, int W, int Cout, int relu, float *out, double *out_sum,
                             double *out_sumsq, void *stream) {
    GN_REQUIRE(B >= 0 && D > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, "gn_conv3d_gcr: bad sizes");
    GN_REQUIRE(C0 % CV_KS == 0 && C1 % CV_KS == 0, "gn_conv3d_gcr: channel counts must be multiples of %d (C0=%d C1=%d)", CV_KS, C0, C1);
    GN_REQUIRE(Cout % 32 == 0, "gn_conv3d_gcr: Cout=%d must be a multiple of 32", Cout);
    GN_REQUIRE(C1 == 0 || (src1 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0), "gn_conv3d_gcr: upsampled source needs even dims");
    GN_REQUIRE((out_sum == nullptr) == (out_sumsq == nullptr), "gn_conv3d_gcr: out_sum and out_sumsq must come together");
    if (B == 0) return GN_OK;
    hipStream_t st = gn_stream(stream);
    if (out_sum) {
        GN_HIP(gn_zero_stats(out_sum, out_sumsq, (size_t)B * Cout, st), "gn_conv3d_gcr");
    }
    ConvArgs p;
    p.src0 = src0; p.src1 = src1; p.a = a; p.d = d; p.wp = wp; p.out = out; p.osum = out_sum; p.osq = out_sumsq;
    p.C0 = C0; p.C1 = C1; p.B = B; p.D = D; p.H = H; p.W = W; p.Cout = Cout; p.relu = relu;
    const int tz = (int)gn_cdiv(D, CV_TZ);
    p.tiles_y = (int)gn_cdiv(H, CV_TY);
    p.tiles_x = (int)gn_cdiv(W, CV_TX);
    const int tiles = tz * p.tiles_y * p.tiles_x;
    // small volumes / small batches: the 32-wide column tile doubles the number of workgroups (same per-CU throughput)
    const bool wide = (Cout % 64 == 0) && ((int64_t)tiles * (Cout / 64) * B >= 1024);
    if (wide) hipLaunchKernelGGL(conv3d_gcr_kernel<2>, dim3(tiles * (Cout / 64), B), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(conv3d_gcr_kernel<1>, dim3(tiles * (Cout / 32), B), dim3(256), 0, st, p);
    gn_note_kernel(wide ? "conv3d_gcr_kernel<2>" : "conv3d_gcr_kernel<1>");
    GN_LAUNCH_CHECK("gn_conv3d_gcr");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ maxpool 2x2x2
// grid (chunks of 256 pooled voxels, B).  Thread = (voxel group, 4-channel lane): float4 loads/stores; optional
// per-channel statistics of the pooled output (next GroupNorm) reduced through LDS, one fp64 atomic pair per channel.
#define MP_VPB 256
__global__ __launch_bounds__(256) void maxpool3d_2_kernel(const float *__restrict__ in, int D, int H, int W, int C,
                                                          float *__restrict__ out, double *__restrict__ osum, double *__restrict__ osq) {
    __shared__ float red[2][256][4];
    const int Do = D >> 1, Ho = H >> 1, Wo = W >> 1, C4 = C >> 2;
    const int b = blockIdx.y;
    const int64_t Vp = (int64_t)Do * Ho * Wo;
    const int64_t v0 = (int64_t)blockIdx.x * MP_VPB;
    const bool strided = (256 % C4) == 0;
    const int ngrp = strided ? 256 / C4 : 1;
    const int c4 = strided ? threadIdx.x % C4 : 0, grp = strided ? threadIdx.x / C4 : 0;
    float4 s4 = make_float4(0.f, 0.f, 0.f, 0.f), q4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto pool = [&](int64_t v, int cc) {
        const int x = (int)(v % Wo), y = (int)((v / Wo) % Ho), z = (int)(v / ((int64_t)Wo * Ho));
        float4 m = make_float4(-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f);
#pragma unroll
        for (int dz = 0; dz < 2; ++dz)
#pragma unroll
            for (int dy = 0; dy < 2; ++dy)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx) {
                    const float4 t = *reinterpret_cast<const float4 *>(
                        in + ((((int64_t)b * D + 2 * z + dz) * H + 2 * y + dy) * W + 2 * x + dx) * C + cc * 4);
                    m.x = fmaxf(m.x, t.x); m.y = fmaxf(m.y, t.y); m.z = fmaxf(m.z, t.z); m.w = fmaxf(m.w, t.w);
                }
        *reinterpret_cast<float4 *>(out + ((int64_t)b * Vp + v) * C + cc * 4) = m;
        return m;
    };
    if (strided) {
        for (int64_t v = v0 + grp; v < v0 + MP_VPB && v < Vp; v += ngrp) {
            const float4 m = pool(v, c4);
            s4.x += m.x; s4.y += m.y; s4.z += m.z; s4.w += m.w;
            q4.x = fmaf(m.x, m.x, q4.x); q4.y = fmaf(m.y, m.y, q4.y); q4.z = fmaf(m.z, m.z, q4.z); q4.w = fmaf(m.w, m.w, q4.w);
        }
    } else {
        for (int64_t i = threadIdx.x; i < (int64_t)MP_VPB * C4; i += 256) {
            const int64_t v = v0 + i / C4;
            if (v < Vp) pool(v, (int)(i % C4));
        }
    }
    if (osum == nullptr) return;   // host guarantees `strided` when statistics are requested
    red[0][threadIdx.x][0] = s4.x; red[0][threadIdx.x][1] = s4.y; red[0][threadIdx.x][2] = s4.z; red[0][threadIdx.x][3] = s4.w;
    red[1][threadIdx.x][0] = q4.x; red[1][threadIdx.x][1] = q4.y; red[1][threadIdx.x][2] = q4.z; red[1][threadIdx.x][3] = q4.w;
    __syncthreads();
    if (threadIdx.x < C) {
        const int cc = threadIdx.x >> 2, k = threadIdx.x & 3;
        double s = 0.0, q = 0.0;
        for (int g = 0; g < ngrp; ++g) { s += red[0][g * C4 + cc][k]; q += red[1][g * C4 + cc][k]; }
        atomicAdd(&osum[(int64_t)b * C + threadIdx.x], s);
        atomicAdd(&osq[(int64_t)b * C + threadIdx.x], q);
    }
}

extern "C" int gn_maxpool3d_2(const float *in, int B, int D, int H, int W, int C, float *out, double *out_sum, double *out_sumsq,
                              void *stream) {
    GN_REQUIRE(B >= 0 && D >= 2 && H >= 2 && W >= 2 && C > 0 && C % 4 == 0, "gn_maxpool3d_2: bad sizes");
    GN_REQUIRE((out_sum == nullptr) == (out_sumsq == nullptr), "gn_maxpool3d_2: out_sum and out_sumsq must come together");
    GN_REQUIRE(out_sum == nullptr || (C <= 256 && 256 % (C / 4) == 0), "gn_maxpool3d_2: statistics need C/4 to divide 256 (C=%d)", C);
    const int64_t Vp = (int64_t)(D / 2) * (H / 2) * (W / 2);
    if (B == 0 || Vp == 0) return GN_OK;
    hipStream_t st = gn_stream(stream);
    if (out_sum) {
        GN_HIP(gn_zero_stats(out_sum, out_sumsq, (size_t)B * C, st), "gn_maxpool3d_2");
    }
    hipLaunchKernelGGL(maxpool3d_2_kernel, dim3((unsigned)gn_cdiv(Vp, MP_VPB), B), dim3(256), 0, st, in, D, H, W, C, out, out_sum, out_sumsq);
    GN_LAUNCH_CHECK("gn_maxpool3d_2");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ elementwise: per-channel affine + bias + activation
// The pieces of create_conv's layer orders other than 'gcr' (components/unet3d.py:19-73: 'cr', 'crg', 'cl', 'ce', 'bcr', ...) that the fused conv
// kernels do not cover: a learnable conv bias (orders without a norm layer), LeakyReLU(0.1) / ELU, and a normalisation that FOLLOWS the
// non-linearity ('crg': its GroupNorm affine is applied here).  y = act(x * a[b][c] + d[b][c] + bias[c]) over channel-last [B][V][C]; a, d, bias
// may be NULL.  act: 0 none, 1 ReLU, 2 LeakyReLU(0.1), 3 ELU(alpha = 1).  In place when out == x.  HBM-bound, a non-default path.
__global__ __launch_bounds__(256) void affine_act_kernel(const float *__restrict__ x, int64_t n4, int64_t VC4, int C4, const float *__restrict__ a,
                                                         const float *__restrict__ d, const float *__restrict__ bias, int act, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int64_t b = i / VC4;
    const int c4 = (int)(i % C4);
    float4 v = reinterpret_cast<const float4 *>(x)[i];
    if (a) {
        const float4 av = reinterpret_cast<const float4 *>(a)[b * C4 + c4], dv = reinterpret_cast<const float4 *>(d)[b * C4 + c4];
        v.x = __fmaf_rn(v.x, av.x, dv.x); v.y = __fmaf_rn(v.y, av.y, dv.y); v.z = __fmaf_rn(v.z, av.z, dv.z); v.w = __fmaf_rn(v.w, av.w, dv.w);
    }
    if (bias) {
        const float4 bv = reinterpret_cast<const float4 *>(bias)[c4];
        v.x = __fadd_rn(v.x, bv.x); v.y = __fadd_rn(v.y, bv.y); v.z = __fadd_rn(v.z, bv.z); v.w = __fadd_rn(v.w, bv.w);
    }
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float t = r[k];
        if (act == 1) r[k] = gn_relu(t);
        else if (act == 2) r[k] = t > 0.f ? t : __fmul_rn(t, 0.1f);
        else if (act == 3) r[k] = t > 0.f ? t : expm1f(t);
    }
    reinterpret_cast<float4 *>(out)[i] = make_float4(r[0], r[1], r[2], r[3]);
}

extern "C" int gn_affine_act(const float *x, int B, int64_t V, int C, const float *a, const float *d, const float *bias, int act, float *out, void *stream) {
    GN_REQUIRE(B >= 0 && V >= 0 && C > 0 && C % 4 == 0, "gn_affine_act: C must be a multiple of 4");
    GN_REQUIRE(act >= 0 && act <= 3, "gn_affine_act: act must be 0 (none), 1 (ReLU), 2 (LeakyReLU 0.1) or 3 (ELU)");
    GN_REQUIRE((a == nullptr) == (d == nullptr), "gn_affine_act: a and d come together");
    if (B == 0 || V == 0) return GN_OK;
    GN_REQUIRE(x && out, "gn_affine_act: null pointer");
    const int64_t n4 = (int64_t)B * V * (C / 4);
    hipLaunchKernelGGL(affine_act_kernel, dim3((unsigned)gn_cdiv(n4, 256)), dim3(256), 0, gn_stream(stream), x, n4, V * (C / 4), C / 4, a, d, bias, act, out);
    GN_LAUNCH_CHECK("gn_affine_act");
    return GN_OK;
}
