// grid.hip -- VolumeFeatureAggregator: per-point cell index + aggregation features, scatter (max | mean) into a
// channel-last feature volume.  Reference: /root/reference/networks/conv_implicit_wnf.py:43-100,
// components/gridding.py:161-256 (VirtualGrid index maths, fp32, truncation toward zero).
#include "common.h"

struct GridParams {
    float lower[3], upper[3];
    int grid[3];
};

// one wavefront per point
__global__ __launch_bounds__(256) void grid_features_kernel(const float *__restrict__ feat, int ldf, int Cf,
                                                            const float *__restrict__ nocs, const float *__restrict__ sim_pos,
                                                            const float *__restrict__ conf, const int64_t *__restrict__ batch,
                                                            int64_t N, GridParams gp, int include_point, int include_conf,
                                                            float *__restrict__ out, int ldo, int32_t *__restrict__ flat_idx) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= N) return;
    float *o = out + p * ldo;
    for (int t = lane; t < Cf; t += 64) o[t] = feat[p * ldf + t];
    // gridding.py:161-186: idx_f = (p + (-lc)) * ((shape-1)/(uc-lc)); idx = clamp(trunc(idx_f), 0, shape-1)
    int cell[3];
    float corner[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        float lc = gp.lower[a], uc = gp.upper[a];
        float idx_scale = __fsub_rn((float)gp.grid[a], 1.0f);
        float scales = __fdiv_rn(idx_scale, __fsub_rn(uc, lc));
        float f = __fmul_rn(__fadd_rn(nocs[p * 3 + a], -lc), scales);
        long long i = (long long)f;  // truncation toward zero (v_cvt, saturating)
        if (i < 0) i = 0;
        if (i > gp.grid[a] - 1) i = gp.grid[a] - 1;
        cell[a] = (int)i;
        // gridding.py:230-256: corner = idx * ((uc-lc)/(shape-1)) + lc
        float inv = __fdiv_rn(__fsub_rn(uc, lc), idx_scale);
        corner[a] = __fadd_rn(__fmul_rn((float)cell[a], inv), lc);
    }
    if (lane == 0) {
        long long b = batch[p];
        flat_idx[p] = (int32_t)((((b * gp.grid[0] + cell[0]) * gp.grid[1] + cell[1]) * (long long)gp.grid[2]) + cell[2]);
    }
    int c = Cf;
    if (include_point) {
        if (lane < 3) o[c + lane] = __fsub_rn(nocs[p * 3 + lane], corner[lane]);
        else if (lane < 6) o[c + lane] = sim_pos[p * 3 + lane - 3];
        c += 6;
    }
    if (include_conf && lane < 3) o[c + lane] = conf[p * 3 + lane];
}

extern "C" int gn_grid_features(const float *feat, int ldf, int Cf, const float *nocs, const float *sim_pos, const float *conf,
                                const int64_t *batch, int64_t N, const float lower[3], const float upper[3], const int grid[3],
                                int include_point, int include_conf, float *out, int ldo, int32_t *flat_idx, void *stream) {
    const int Ctot = Cf + (include_point ? 6 : 0) + (include_conf ? 3 : 0);
    GN_REQUIRE(N >= 0 && Cf >= 0 && ldo >= Ctot, "gn_grid_features: bad sizes");
    if (N == 0) return GN_OK;
    GridParams gp;
    for (int a = 0; a < 3; ++a) { gp.lower[a] = lower[a]; gp.upper[a] = upper[a]; gp.grid[a] = grid[a]; }
    hipLaunchKernelGGL(grid_features_kernel, dim3((unsigned)gn_cdiv(N, 4)), dim3(256), 0, gn_stream(stream), feat, ldf, Cf, nocs,
                       sim_pos, conf, batch, N, gp, include_point, include_conf, out, ldo, flat_idx);
    GN_LAUNCH_CHECK("gn_grid_features");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ scatter
// order-preserving float <-> uint encoding: enc(x) is monotone in x and > 0 for every non-NaN float, so a
// zero-filled volume reads as "empty" and atomicMax on the encoding is an order-independent float max.
__device__ __forceinline__ unsigned enc_f32(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(unsigned e) {
    unsigned u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
    return __uint_as_float(u);
}

// max: order-independent by construction (atomicMax on the encoding).
__global__ __launch_bounds__(256) void scatter_max_accum_kernel(const float *__restrict__ src, int lds, const int32_t *__restrict__ flat_idx,
                                                                int64_t N, int C, float *__restrict__ vol, int32_t *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= N) return;
    const int64_t cell = flat_idx[p];
    if (lane == 0) atomicAdd(&count[cell], 1);
    float *v = vol + cell * C;
    for (int ch = lane; ch < C; ch += 64) atomicMax(reinterpret_cast<unsigned *>(v) + ch, enc_f32(src[p * lds + ch]));
}

// the first point that reaches a cell here finalises it (decode the max)
__global__ __launch_bounds__(256) void scatter_max_finalize_kernel(const int32_t *__restrict__ flat_idx, int64_t N, int C,
                                                                   float *__restrict__ vol, int32_t *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= N) return;
    const int64_t cell = flat_idx[p];
    int c = 0;
    if (lane == 0) c = atomicExch(&count[cell], 0);
    c = __shfl(c, 0);
    if (c <= 0) return;
    float *v = vol + cell * C;
    for (int ch = lane; ch < C; ch += 64) v[ch] = dec_f32(__float_as_uint(v[ch]));
}

// mean: DETERMINISTIC.  A float atomicAdd into the volume would make the sum depend on the arrival order of a cell's points (with
// seeded synthetic weights a cell collects > 1000 points; the run-to-run spread, amplified by the GroupNorm of a > 99 % empty
// volume, reached 4e-5 on the WNF).  Instead every occupied cell gets an OWNER point (first atomicCAS on the zeroed count
// workspace; which point wins only decides where the partial sums live), the cell's points add their channels into the owner's
// row of an fp64 scratch [N][C] with fp64 atomics -- sums of <= 2^13 fp32 values are exact in fp64 unless their exponents span
// more than 2^16, so the result does not depend on the order -- and the owner stores fp32(sum) / count: for one or two points
// per cell (the realistic case: 6000 points over 128^3 cells) bit-identical to a sequential fp32 sum, closer to the exact mean
// than it otherwise.
__global__ __launch_bounds__(256) void scatter_mean_owner_kernel(const int32_t *__restrict__ flat_idx, int64_t N, int32_t *__restrict__ count,
                                                                 int32_t *__restrict__ owner_of, int32_t *__restrict__ npts) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    const int64_t cell = flat_idx[p];
    const int old = atomicCAS(&count[cell], 0, (int)p + 1);
    const int owner = old == 0 ? (int)p : old - 1;
    owner_of[p] = owner;
    atomicAdd(&npts[owner], 1);
}

__global__ __launch_bounds__(256) void scatter_mean_accum_kernel(const float *__restrict__ src, int lds, int64_t N, int C,
                                                                 const int32_t *__restrict__ owner_of, double *__restrict__ acc) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= N) return;
    double *a = acc + (int64_t)owner_of[p] * C;
    for (int ch = lane; ch < C; ch += 64) atomicAdd(a + ch, (double)src[p * lds + ch]);
}

__global__ __launch_bounds__(256) void scatter_mean_finalize_kernel(const int32_t *__restrict__ flat_idx, int64_t N, int C,
                                                                    const int32_t *__restrict__ owner_of, const int32_t *__restrict__ npts,
                                                                    const double *__restrict__ acc, float *__restrict__ vol,
                                                                    int32_t *__restrict__ count) {
    const int lane = threadIdx.x & 63;
    const int64_t p = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= N || owner_of[p] != (int)p) return;
    const int64_t cell = flat_idx[p];
    const float c = (float)npts[p];
    float *v = vol + cell * C;
    for (int ch = lane; ch < C; ch += 64) v[ch] = __fdiv_rn((float)acc[p * C + ch], c);
    if (lane == 0) count[cell] = 0;                 // gn_grid_stats expects the count workspace zeroed
}

extern "C" size_t gn_grid_scatter_workspace_bytes(int64_t N, int C, int reduce) {
    if (reduce != 1 || N <= 0) return 0;
    return (size_t)N * C * sizeof(double) + 2 * (size_t)N * sizeof(int32_t);
}

extern "C" int gn_grid_scatter(const float *src, int lds, const int32_t *flat_idx, int64_t N, int C, int64_t cells, int reduce,
                               float *vol, int32_t *count_ws, void *ws, size_t ws_bytes, int vol_is_zeroed, void *stream) {
    GN_REQUIRE(N >= 0 && C > 0 && cells >= 0 && (reduce == 0 || reduce == 1), "gn_grid_scatter: bad arguments");
    GN_REQUIRE(N < (int64_t)0x7fffffff, "gn_grid_scatter: more than 2^31-2 points");
    GN_REQUIRE(ws_bytes >= gn_grid_scatter_workspace_bytes(N, C, reduce) && (ws || !gn_grid_scatter_workspace_bytes(N, C, reduce)),
               "gn_grid_scatter: workspace too small (gn_grid_scatter_workspace_bytes)");
    hipStream_t st = gn_stream(stream);
    if (!vol_is_zeroed) {       // (the caller may have zeroed both ahead of time, e.g. on a side stream next to the serial FPS kernels)
        GN_HIP(hipMemsetAsync(vol, 0, sizeof(float) * (size_t)cells * C, st), "gn_grid_scatter(memset vol)");
        GN_HIP(hipMemsetAsync(count_ws, 0, sizeof(int32_t) * (size_t)cells, st), "gn_grid_scatter(memset count)");
    }
    if (N == 0) return GN_OK;
    dim3 grid((unsigned)gn_cdiv(N, 4)), block(256);
    if (reduce == 0) {
        hipLaunchKernelGGL(scatter_max_accum_kernel, grid, block, 0, st, src, lds, flat_idx, N, C, vol, count_ws);
        hipLaunchKernelGGL(scatter_max_finalize_kernel, grid, block, 0, st, flat_idx, N, C, vol, count_ws);
    } else {
        double *acc = reinterpret_cast<double *>(ws);
        int32_t *owner_of = reinterpret_cast<int32_t *>(acc + (size_t)N * C), *npts = owner_of + N;
        GN_HIP(hipMemsetAsync(ws, 0, gn_grid_scatter_workspace_bytes(N, C, reduce), st), "gn_grid_scatter(memset ws)");
        hipLaunchKernelGGL(scatter_mean_owner_kernel, dim3((unsigned)gn_cdiv(N, 256)), block, 0, st, flat_idx, N, count_ws, owner_of, npts);
        hipLaunchKernelGGL(scatter_mean_accum_kernel, grid, block, 0, st, src, lds, N, C, owner_of, acc);
        hipLaunchKernelGGL(scatter_mean_finalize_kernel, grid, block, 0, st, flat_idx, N, C, owner_of, npts, acc, vol, count_ws);
    }
    GN_LAUNCH_CHECK("gn_grid_scatter");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ sparse statistics
// GroupNorm statistics of a scattered volume from its occupied cells: the first point that claims a cell (atomicCAS on
// the count workspace, which gn_grid_scatter leaves zeroed) contributes that cell's channel values.
// Block = 256 threads: 64 consecutive points, thread = (channel lane, point group); register partials are flushed with
// fp64 atomics when the sample index changes (points are sorted by sample) and at the end.
#define GS_PTS 64
__global__ __launch_bounds__(256) void grid_stats_kernel(const float *__restrict__ vol, const int32_t *__restrict__ flat_idx, int64_t N,
                                                         int C, int64_t cps, int32_t *__restrict__ count, double *__restrict__ sum,
                                                         double *__restrict__ sumsq) {
    __shared__ int owner[GS_PTS];
    const int64_t p0 = (int64_t)blockIdx.x * GS_PTS;
    if (threadIdx.x < GS_PTS) {
        const int64_t p = p0 + threadIdx.x;
        owner[threadIdx.x] = (p < N) ? (atomicCAS(&count[flat_idx[p]], 0, 1) == 0) : 0;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        double s = 0.0, q = 0.0;
        int64_t bcur = -1;
        for (int i = 0; i < GS_PTS; ++i) {
            if (!owner[i]) continue;
            const int64_t cell = flat_idx[p0 + i];
            const int64_t b = cell / cps;
            if (b != bcur) {
                if (bcur >= 0) { atomicAdd(&sum[bcur * C + c], s); atomicAdd(&sumsq[bcur * C + c], q); }
                s = q = 0.0;
                bcur = b;
            }
            const double v = (double)vol[cell * C + c];
            s += v;
            q += v * v;
        }
        if (bcur >= 0) { atomicAdd(&sum[bcur * C + c], s); atomicAdd(&sumsq[bcur * C + c], q); }
    }
}

extern "C" int gn_grid_stats(const float *vol, const int32_t *flat_idx, int64_t N, int C, int64_t cells_per_sample, int B,
                             int32_t *count_ws, double *sum, double *sumsq, void *stream) {
    GN_REQUIRE(N >= 0 && C > 0 && cells_per_sample > 0 && B >= 0, "gn_grid_stats: bad sizes");
    hipStream_t st = gn_stream(stream);
    GN_HIP(gn_zero_stats(sum, sumsq, (size_t)B * C, st), "gn_grid_stats");
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(grid_stats_kernel, dim3((unsigned)gn_cdiv(N, GS_PTS)), dim3(256), 0, st, vol, flat_idx, N, C, cells_per_sample,
                       count_ws, sum, sumsq);
    GN_LAUNCH_CHECK("gn_grid_stats");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ tile occupancy
// Which output tiles of a 3x3x3 convolution over the scattered volume can see an occupied cell: cell (d, h, w) reaches the outputs
// d-1..d+1 (x h x w), i.e. up to 2 x 2 x 2 tiles of (TD, TH, TW) voxels (reach 1; reach 2 = the convolution behind it).  flags[b][(ty * tiles_x + tx) * tiles_z + tz] = 1 -- the tile
// order of gn_conv3d_gcr_split's occupancy-aware launch (z fastest).  Everything else of that layer's output is a border-class constant.
__global__ __launch_bounds__(256) void tile_flags_kernel(const int32_t *__restrict__ flat_idx, int64_t N, int G0, int G1, int G2, int TD, int TH,
                                                         int TW, int tiles_z, int tiles_y, int tiles_x, int reach, unsigned char *__restrict__ flags) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= N) return;
    int64_t c = flat_idx[p];
    const int w = (int)(c % G2); c /= G2;
    const int h = (int)(c % G1); c /= G1;
    const int d = (int)(c % G0);
    const int64_t b = c / G0;
    const int z0 = max(d - reach, 0) / TD, z1 = min(d + reach, G0 - 1) / TD;
    const int y0 = max(h - reach, 0) / TH, y1 = min(h + reach, G1 - 1) / TH;
    const int x0 = max(w - reach, 0) / TW, x1 = min(w + reach, G2 - 1) / TW;
    unsigned char *f = flags + b * ((int64_t)tiles_z * tiles_y * tiles_x);
    for (int ty = y0; ty <= y1; ++ty)
        for (int tx = x0; tx <= x1; ++tx)
            for (int tz = z0; tz <= z1; ++tz) f[((int64_t)ty * tiles_x + tx) * tiles_z + tz] = 1;
}

extern "C" int gn_grid_tile_flags(const int32_t *flat_idx, int64_t N, int B, int G0, int G1, int G2, int reach, unsigned char *flags, void *stream) {
    GN_REQUIRE(N >= 0 && B >= 0 && G0 > 0 && G1 > 0 && G2 > 0 && reach >= 1 && reach <= 4, "gn_grid_tile_flags: bad sizes");
    const int TD = 4, TH = 8, TW = 8;               // the output tile of csrc/unet_split.hip (SP_TZ, SP_TY, SP_TX)
    const int tz = (int)gn_cdiv(G0, TD), ty = (int)gn_cdiv(G1, TH), tx = (int)gn_cdiv(G2, TW);
    hipStream_t st = gn_stream(stream);
    GN_HIP(hipMemsetAsync(flags, 0, (size_t)B * tz * ty * tx, st), "gn_grid_tile_flags");
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(tile_flags_kernel, dim3((unsigned)gn_cdiv(N, 256)), dim3(256), 0, st, flat_idx, N, G0, G1, G2, TD, TH, TW, tz, ty, tx, reach, flags);
    GN_LAUNCH_CHECK("gn_grid_tile_flags");
    return GN_OK;
}
