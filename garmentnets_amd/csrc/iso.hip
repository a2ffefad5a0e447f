// iso.hip -- isosurface stage on gfx950: Gaussian gradient magnitude, min/max, Lewiner marching cubes (MC33),
// nearest-voxel gather.  Replaces scipy.ndimage.gaussian_gradient_magnitude + skimage.measure.marching_cubes
// (method='lewiner') + the vertex look-ups of /root/reference/predict.py:160-181.
//
// MC33 is formulated for the GPU as classify -> scan -> emit instead of scikit-image's sequential sweep:
//   classify : one thread per cell: sign index, Lewiner case, face / interior tests -> tiling row + #triangles,
//              and the number of vertices this cell is the FIRST user of in sweep order (axis0 outer, axis2 inner):
//              an edge is new for a cell iff no lexicographically earlier cell shares it, which depends only on the
//              edge's position in the cell and on the cell touching the low boundary; the centre vertex is always new.
//              One BYTE per cell (new vertices << 4 | triangles) + the packed sum of each 1024-cell block.
//   scan     : exclusive prefix sums of (new vertices, triangles) over cells in sweep order: one small kernel over the block sums;
//              the emit workgroups (one per block) rebuild their cells' prefixes from the count bytes -- no per-cell offsets in HBM
//              => vertex ids = order of first use, faces in sweep order, exactly as the sequential algorithm.
//   emit     : owners write vertex positions + the global edge->vertex table; a dense pass over the VERTICES (one thread each) gathers
//              the normal and value over the (<=4) cells adjacent to the vertex's edge, visited in sweep order: the normal contributions
//              accumulate in the same order as the sequential algorithm, the value is the max cell span; then every cell writes its faces.
// The look-up tables (13.4 KB, Lewiner et al. 2003) are staged into LDS by the workgroups that hold a surface cell.
#include "common.h"
#include "mc33_luts.h"

#include <float.h>

// ================================================================================================ GGM
template <typename T> struct GgmW {
    T w[65];
    int radius;
    int symmetric;  // 1 symmetric, -1 antisymmetric, 0 generic
};
typedef GgmW<double> GgmWeights;
// accumulation arithmetic of a correlation: fp64 in scipy's operation order (bit-exact, the default) or the same order in fp32 (gn_ggm3d_batch_ex,
// accum_bits = 32: no scipy bit-parity, 1e-6-class against it)
__device__ __forceinline__ double gg_mul(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double gg_add(double a, double b) { return __dadd_rn(a, b); }
__device__ __forceinline__ double gg_sub(double a, double b) { return __dsub_rn(a, b); }
__device__ __forceinline__ float gg_mul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float gg_add(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float gg_sub(float a, float b) { return __fsub_rn(a, b); }
// NaN-propagating min / max (IEEE 754-2019 minimum / maximum = numpy.min / numpy.max; one v_minimum3_f32 / v_maximum3_f32 each)
__device__ __forceinline__ float gn_min_nan(float a, float b) { return __builtin_elementwise_minimum(a, b); }
__device__ __forceinline__ float gn_max_nan(float a, float b) { return __builtin_elementwise_maximum(a, b); }
// order-preserving encodings so that integer atomics implement float min / max; a NaN takes the extreme code of its side and decodes to a NaN
__device__ __forceinline__ unsigned gn_enc_min(float v) { const unsigned e = __float_as_uint(v); return v != v ? 0u : ((e & 0x80000000u) ? ~e : (e | 0x80000000u)); }
__device__ __forceinline__ unsigned gn_enc_max(float v) { const unsigned e = __float_as_uint(v); return v != v ? 0xffffffffu : ((e & 0x80000000u) ? ~e : (e | 0x80000000u)); }

// correlate1d along `axis`, edge-replicate, fp64 accumulation in scipy's operation order, fp32 store
__global__ __launch_bounds__(256) void ggm_correlate_kernel(const float *__restrict__ in, float *__restrict__ out, int n0, int n1,
                                                            int n2, int axis, GgmWeights gw, int accum) {
    // accum = 0: out = correlation.  1 / 2 / 3: the fp32 correlation value t is squared and accumulated instead (first / middle /
    // last axis term of the gradient magnitude: out = t*t | out += t*t | out = sqrt(out + t*t)) -- no pass of its own over
    // a stored t
    const int64_t tot = (int64_t)n0 * n1 * n2;
    in += (int64_t)blockIdx.y * tot;                // batched: one volume per blockIdx.y
    out += (int64_t)blockIdx.y * tot;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= tot) return;
    const int i2 = (int)(t % n2), i1 = (int)((t / n2) % n1), i0 = (int)(t / ((int64_t)n1 * n2));
    const int n = axis == 0 ? n0 : (axis == 1 ? n1 : n2);
    const int64_t st = axis == 0 ? (int64_t)n1 * n2 : (axis == 1 ? n2 : 1);
    const int i = axis == 0 ? i0 : (axis == 1 ? i1 : i2);
    const int64_t base = t - (int64_t)i * st;
    const int r = gw.radius;
    auto at = [&](int j) -> double {
        int k = i + j;
        k = k < 0 ? 0 : (k >= n ? n - 1 : k);
        return (double)in[base + (int64_t)k * st];
    };
    double acc;
    if (gw.symmetric == 1) {
        acc = __dmul_rn(at(0), gw.w[r]);
        for (int j = -r; j < 0; ++j) acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(at(j), at(-j)), gw.w[r + j]));
    } else if (gw.symmetric == -1) {
        acc = __dmul_rn(at(0), gw.w[r]);
        for (int j = -r; j < 0; ++j) acc = __dadd_rn(acc, __dmul_rn(__dsub_rn(at(j), at(-j)), gw.w[r + j]));
    } else {
        acc = 0.0;
        for (int j = -r; j <= r; ++j) acc = __dadd_rn(acc, __dmul_rn(at(j), gw.w[r + j]));
    }
    const float tv = (float)acc;
    if (accum == 0) { out[t] = tv; return; }
    const float sq = __fmul_rn(tv, tv);
    const float v = accum == 1 ? sq : __fadd_rn(out[t], sq);
    // correctly rounded fp32 sqrt (numpy.sqrt): fp64 sqrt of an fp32 value rounds to the same fp32 result
    out[t] = accum == 3 ? (float)__dsqrt_rn((double)v) : v;
}

// ---- fused form (radius <= GGM_R: sigma < 0.625, the reference's 0.5): the whole gradient magnitude of a GGM_TZ x GGM_TY x GGM_TX (4 x 8 x 32) output tile in
// ONE kernel -- input tile + halo staged in LDS with edge-replicated coordinates (so that every later pass is a plain correlation inside
// the tile: replicate-at-each-pass composes), pass 0 along axis 0 with both kernels (derivative -> chain d = 0, Gaussian -> shared by
// d = 1, 2), pass 1 along axis 1 (three chains), pass 2 along axis 2 squares and accumulates in the order d = 0, 1, 2 -- the arithmetic
// of the 8-pass form (fp64 accumulation in scipy's order, fp32 rounding between the passes, correctly rounded sqrt), bit for bit, with
// the compulsory HBM traffic only: each voxel read once (+ halo, mostly L2 hits) and written once instead of 8 round trips.
#define GGM_R 2
#ifndef GGM_TZ
#define GGM_TZ 4                     // (round 6: 8 -> 4: 28 instead of 57 KB of LDS per workgroup = 5 instead of 2 waves per SIMD; 0.72 -> 0.42 ms per 16 x 128^3)
#endif
#ifndef GGM_TY
#define GGM_TY 8
#endif
#define GGM_TX 32
// one correlation output from a register window win[0 .. 2R] (centre at R) -- the operation order of ggm_correlate_kernel; T = accumulation type
template <typename T>
__device__ __forceinline__ float ggm_corr_win(const float *win, const GgmW<T> &gw) {
    constexpr int R = GGM_R;
    const int r = gw.radius;
    T acc;
    if (gw.symmetric == 1) {
        acc = gg_mul((T)win[R], gw.w[r]);
#pragma unroll
        for (int j = -R; j < 0; ++j)
            if (j >= -r) acc = gg_add(acc, gg_mul(gg_add((T)win[R + j], (T)win[R - j]), gw.w[r + j]));
    } else if (gw.symmetric == -1) {
        acc = gg_mul((T)win[R], gw.w[r]);
#pragma unroll
        for (int j = -R; j < 0; ++j)
            if (j >= -r) acc = gg_add(acc, gg_mul(gg_sub((T)win[R + j], (T)win[R - j]), gw.w[r + j]));
    } else {
        acc = (T)0;
#pragma unroll
        for (int j = -R; j <= R; ++j)
            if (j >= -r && j <= r) acc = gg_add(acc, gg_mul((T)win[R + j], gw.w[r + j]));
    }
    return (float)acc;
}

// Every pass walks COLUMNS along its axis with the 2R+1 inputs of an output in a sliding register window: one LDS read per new input
// instead of 2R+1 per output, and the (z, y, x) decomposition of an index once per column instead of once per output.
// RANGE: the volume's (min, max) ride along -- every value staged here (tile + edge-replicated halo) IS a voxel of the volume, so the extremes of
// everything the workgroups load are the volume's; NaN-propagating; a pair per wave in range_ws, folded per volume by ggm_range_reduce_kernel.  Saves
// gn_minmax_batch's pass over the volume.
template <typename T, bool RANGE>
__global__ __launch_bounds__(256) void ggm_fused_kernel(const float *__restrict__ in, float *__restrict__ out, int n0, int n1, int n2,
                                                        GgmW<T> w0, GgmW<T> w1, float *__restrict__ range_ws) {
    // HX: row pitch of the LDS tiles, ONE float of padding: pass 2 walks along x with lane = row, and 37 * row mod 32 is a permutation
    constexpr int R = GGM_R, HZ = GGM_TZ + 2 * R, HY = GGM_TY + 2 * R, HXV = GGM_TX + 2 * R, HX = HXV + 1, WN = 2 * R + 1;
    constexpr int NA = HZ * HY * HX, NB1 = GGM_TZ * HY * HX, NC1 = GGM_TZ * GGM_TY * HX;
    constexpr int REG0 = (3 * NC1 > NA) ? 3 * NC1 : NA;                    // the three pass-1 arrays overlay the (dead) input tile
    __shared__ float lds[REG0 + 2 * NB1];
    float *const A = lds, *const B = lds + REG0, *const C = lds;
    const int64_t tot = (int64_t)n0 * n1 * n2;
    in += (int64_t)blockIdx.y * tot;
    out += (int64_t)blockIdx.y * tot;
    const int tx_n = (n2 + GGM_TX - 1) / GGM_TX, ty_n = (n1 + GGM_TY - 1) / GGM_TY;
    int t = blockIdx.x;
    const int x0 = (t % tx_n) * GGM_TX; t /= tx_n;
    const int y0 = (t % ty_n) * GGM_TY; t /= ty_n;
    const int z0 = t * GGM_TZ;
    const int tid = threadIdx.x;
    // stage the tile + halo, edge-replicated: rows of HXV consecutive x, a wave per row; ALL of a thread's loads are issued before the first
    // LDS store (36 dependent load -> store round trips per thread were most of this kernel's time)
    {
        constexpr int NROW = HZ * HY / 4;            // rows per wave
        static_assert(HZ * HY % 4 == 0, "rows split evenly over the 4 waves");
        const int hx = tid & 63, w = tid >> 6;
        int gx = x0 + hx - R;
        gx = gx < 0 ? 0 : (gx >= n2 ? n2 - 1 : gx);
        float tmp[NROW];
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
            const int row = w + 4 * i, hy = row % HY, hz = row / HY;
            int gz = z0 + hz - R, gy = y0 + hy - R;
            gz = gz < 0 ? 0 : (gz >= n0 ? n0 - 1 : gz);
            gy = gy < 0 ? 0 : (gy >= n1 ? n1 - 1 : gy);
            tmp[i] = hx < HXV ? in[((int64_t)gz * n1 + gy) * n2 + gx] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NROW; ++i)
            if (hx < HXV) A[(w + 4 * i) * HX + hx] = tmp[i];
        if constexpr (RANGE) {
            // one plain (min, max) store per wave into range_ws [volume][tile][wave]; ggm_range_reduce_kernel folds a volume's pairs afterwards.  (Atomics on the
            // volume's one record serialise in the L2 however they are thinned: every workgroup of the first generation -- at B = 1 that is all of them -- sees the
            // empty record; measured 0.95 against 0.42 ms for 16 x 128^3, A/B record section 11)
            float mn = INFINITY, mx = -INFINITY;
            if (hx < HXV) {
#pragma unroll
                for (int i = 0; i < NROW; ++i) { mn = gn_min_nan(mn, tmp[i]); mx = gn_max_nan(mx, tmp[i]); }
            }
            for (int off = 32; off >= 1; off >>= 1) {
                mn = gn_min_nan(mn, __shfl_xor(mn, off));
                mx = gn_max_nan(mx, __shfl_xor(mx, off));
            }
            if (hx == 0) {
                float2 *slot = reinterpret_cast<float2 *>(range_ws) + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + w;
                *slot = make_float2(mn, mx);
            }
        }
    }
    __syncthreads();
    // pass 0 (axis 0): B[0] = corr(A, w1), B[1] = corr(A, w0) at z = 0 .. TZ-1; a thread owns (hy, hx) columns
    for (int col = tid; col < HY * HX; col += 256) {
        float win[WN];
#pragma unroll
        for (int j = 0; j < WN - 1; ++j) win[j + 1] = A[j * HY * HX + col];
#pragma unroll
        for (int z = 0; z < GGM_TZ; ++z) {
#pragma unroll
            for (int j = 0; j < WN - 1; ++j) win[j] = win[j + 1];
            win[WN - 1] = A[(z + WN - 1) * HY * HX + col];
            B[z * HY * HX + col] = ggm_corr_win(win, w1);
            B[NB1 + z * HY * HX + col] = ggm_corr_win(win, w0);
        }
    }
    __syncthreads();
    // pass 1 (axis 1): C[0] = corr(B[0], w0) (d = 0), C[1] = corr(B[1], w1) (d = 1), C[2] = corr(B[1], w0) (d = 2); columns (z, hx)
    for (int col = tid; col < GGM_TZ * HX; col += 256) {
        const int hx = col % HX, z = col / HX;
        const float *b0 = B + z * HY * HX + hx, *b1 = b0 + NB1;
        float wa[WN], wb[WN];
#pragma unroll
        for (int j = 0; j < WN - 1; ++j) { wa[j + 1] = b0[j * HX]; wb[j + 1] = b1[j * HX]; }
#pragma unroll
        for (int y = 0; y < GGM_TY; ++y) {
#pragma unroll
            for (int j = 0; j < WN - 1; ++j) { wa[j] = wa[j + 1]; wb[j] = wb[j + 1]; }
            wa[WN - 1] = b0[(y + WN - 1) * HX];
            wb[WN - 1] = b1[(y + WN - 1) * HX];
            const int o = (z * GGM_TY + y) * HX + hx;
            const float c0 = ggm_corr_win(wa, w0), c1 = ggm_corr_win(wb, w1), c2 = ggm_corr_win(wb, w0);
            C[o] = c0;
            C[NC1 + o] = c1;
            C[2 * NC1 + o] = c2;
        }
    }
    __syncthreads();
    // pass 2 (axis 2): squares accumulated in the order d = 0, 1, 2 (fp32), correctly rounded square root.  thread = one of the TZ x TY (z, y) rows x one x chunk,
    // wave = an 8-wide x chunk (conflict-free LDS reads thanks to the odd row pitch); the results go through LDS (the dead B region) so
    // that the global stores are rows of 32 consecutive x
    float *const O = B;
    constexpr int OP = GGM_TX + 1;
    {
        constexpr int ROWS = GGM_TZ * GGM_TY, XC = GGM_TX / (256 / ROWS);     // a thread owns XC consecutive x of one (z, y) row
        static_assert(256 % ROWS == 0 && GGM_TX % (256 / ROWS) == 0 && (ROWS & (ROWS - 1)) == 0, "pass 2: 256 threads = rows x x-chunks");
        const int row = tid & (ROWS - 1), xs = (tid / ROWS) * XC;
        const float *c0 = C + row * HX + xs, *c1 = c0 + NC1, *c2 = c0 + 2 * NC1;
        float w0v[WN], w1v[WN], w2v[WN];
#pragma unroll
        for (int j = 0; j < WN - 1; ++j) { w0v[j + 1] = c0[j]; w1v[j + 1] = c1[j]; w2v[j + 1] = c2[j]; }
#pragma unroll
        for (int x = 0; x < XC; ++x) {
#pragma unroll
            for (int j = 0; j < WN - 1; ++j) { w0v[j] = w0v[j + 1]; w1v[j] = w1v[j + 1]; w2v[j] = w2v[j + 1]; }
            w0v[WN - 1] = c0[x + WN - 1]; w1v[WN - 1] = c1[x + WN - 1]; w2v[WN - 1] = c2[x + WN - 1];
            const float t0 = ggm_corr_win(w0v, w0), t1 = ggm_corr_win(w1v, w0), t2 = ggm_corr_win(w2v, w1);
            float v = __fmul_rn(t0, t0);
            v = __fadd_rn(v, __fmul_rn(t1, t1));
            v = __fadd_rn(v, __fmul_rn(t2, t2));
            O[row * OP + xs + x] = sizeof(T) == 8 ? (float)__dsqrt_rn((double)v) : __fsqrt_rn(v);
        }
    }
    __syncthreads();
    for (int i = tid; i < GGM_TZ * GGM_TY * GGM_TX; i += 256) {
        const int x = i & (GGM_TX - 1), row = i / GGM_TX, y = row % GGM_TY, z = row / GGM_TY;
        const int gz = z0 + z, gy = y0 + y, gx = x0 + x;
        if (gz < n0 && gy < n1 && gx < n2) out[((int64_t)gz * n1 + gy) * n2 + gx] = O[row * OP + x];
    }
}

static void ggm_kernel1d(double sigma, int order, int radius, GgmWeights &g) {
    // scipy _gaussian_kernel1d + the [::-1] of gaussian_filter1d
    const double sigma2 = sigma * sigma;
    const int n = 2 * radius + 1;
    double phi[65], sum = 0.0;
    for (int i = 0; i < n; ++i) { double x = (double)(i - radius); phi[i] = exp(-0.5 / sigma2 * x * x); sum += phi[i]; }
    for (int i = 0; i < n; ++i) phi[i] /= sum;
    for (int i = 0; i < n; ++i) {
        double x = (double)(i - radius);
        g.w[n - 1 - i] = order == 1 ? (-x / sigma2) * phi[i] : phi[i];
    }
    g.radius = radius;
    int sym = 1, anti = 1;
    for (int j = 1; j <= radius; ++j) {
        if (fabs(g.w[radius + j] - g.w[radius - j]) > DBL_EPSILON) sym = 0;
        if (fabs(g.w[radius + j] + g.w[radius - j]) > DBL_EPSILON) anti = 0;
    }
    g.symmetric = sym ? 1 : (anti ? -1 : 0);
}

// ================================================================================================ min / max
__global__ __launch_bounds__(256) void minmax_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ out_enc) {
    __shared__ float smn[4], smx[4];
    x += (int64_t)blockIdx.y * n;                   // batched: n elements and one (min, max) pair per blockIdx.y
    out_enc += 2 * blockIdx.y;
    float mn = INFINITY, mx = -INFINITY;             // NaN-propagating throughout (numpy.min / numpy.max, what skimage's level check sees)
    // a volume may start anywhere (a slice of an odd-sized batch): scalar head up to the first 16-byte boundary, float4 body, scalar tail
    int64_t head = (int64_t)(((16u - (unsigned)((uintptr_t)x & 15u)) & 15u) >> 2);
    if (head > n) head = n;
    const int64_t n4 = (n - head) >> 2, tail0 = head + (n4 << 2);
    const float4 *x4 = reinterpret_cast<const float4 *>(x + head);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 v = x4[i];
        mn = gn_min_nan(gn_min_nan(mn, v.x), gn_min_nan(v.y, gn_min_nan(v.z, v.w)));
        mx = gn_max_nan(gn_max_nan(mx, v.x), gn_max_nan(v.y, gn_max_nan(v.z, v.w)));
    }
    if (blockIdx.x == 0 && threadIdx.x < 8) {
        const int64_t i = threadIdx.x < 4 ? (int64_t)threadIdx.x : tail0 + (threadIdx.x - 4);
        const bool ok = threadIdx.x < 4 ? (int64_t)threadIdx.x < head : i < n;
        if (ok) {
            const float v = x[i];
            mn = gn_min_nan(mn, v);
            mx = gn_max_nan(mx, v);
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        mn = gn_min_nan(mn, __shfl_xor(mn, off));
        mx = gn_max_nan(mx, __shfl_xor(mx, off));
    }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = gn_min_nan(gn_min_nan(smn[0], smn[1]), gn_min_nan(smn[2], smn[3]));
        mx = gn_max_nan(gn_max_nan(smx[0], smx[1]), gn_max_nan(smx[2], smx[3]));
        atomicMin(&out_enc[0], gn_enc_min(mn));     // (one pair per workgroup)
        atomicMax(&out_enc[1], gn_enc_max(mx));
    }
}
__global__ void minmax_init_kernel(unsigned *o) { o += 2 * blockIdx.x; o[0] = 0xffffffffu; o[1] = 0u; }
__global__ void minmax_decode_kernel(unsigned *o) {
    o += 2 * blockIdx.x;
    for (int k = 0; k < 2; ++k) {
        unsigned e = o[k];
        unsigned u = (e & 0x80000000u) ? (e & 0x7fffffffu) : ~e;
        o[k] = u;
    }
}

extern "C" int gn_minmax_batch(const float *x, int batch, int64_t n, float *out2, void *stream) {
    GN_REQUIRE(n > 0 && batch >= 0 && batch <= 65535, "gn_minmax: empty input");
    if (batch == 0) return GN_OK;
    hipStream_t st = gn_stream(stream);
    unsigned *o = reinterpret_cast<unsigned *>(out2);
    hipLaunchKernelGGL(minmax_init_kernel, dim3(batch), dim3(1), 0, st, o);
    GN_REQUIRE(((uintptr_t)x & 3) == 0, "gn_minmax: x must be 4-byte aligned");
    int blocks = (int)(gn_cdiv(n, 1024) < 512 ? gn_cdiv(n, 1024) : 512);
    hipLaunchKernelGGL(minmax_kernel, dim3(blocks, batch), dim3(256), 0, st, x, n, o);
    hipLaunchKernelGGL(minmax_decode_kernel, dim3(batch), dim3(1), 0, st, o);
    GN_LAUNCH_CHECK("gn_minmax");
    return GN_OK;
}

extern "C" int gn_minmax(const float *x, int64_t n, float *out2, void *stream) { return gn_minmax_batch(x, 1, n, out2, stream); }

// ---- GGM entry points
// one workgroup per volume folds the npairs (min, max) pairs the fused launch's waves left in range_ws -> out2 [volume] = (min, max), NaN-propagating
__global__ __launch_bounds__(256) void ggm_range_reduce_kernel(const float2 *__restrict__ ws, int npairs, float *__restrict__ out2) {
    __shared__ float smn[4], smx[4];
    ws += (size_t)blockIdx.x * npairs;
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < npairs; i += 256) { const float2 v = ws[i]; mn = gn_min_nan(mn, v.x); mx = gn_max_nan(mx, v.y); }
    for (int off = 32; off >= 1; off >>= 1) { mn = gn_min_nan(mn, __shfl_xor(mn, off)); mx = gn_max_nan(mx, __shfl_xor(mx, off)); }
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        out2[2 * blockIdx.x] = gn_min_nan(gn_min_nan(smn[0], smn[1]), gn_min_nan(smn[2], smn[3]));
        out2[2 * blockIdx.x + 1] = gn_max_nan(gn_max_nan(smx[0], smx[1]), gn_max_nan(smx[2], smx[3]));
    }
}

static int64_t ggm_tiles(int n0, int n1, int n2) { return gn_cdiv(n0, GGM_TZ) * gn_cdiv(n1, GGM_TY) * gn_cdiv(n2, GGM_TX); }
extern "C" size_t gn_ggm3d_range_workspace_bytes(int batch, int n0, int n1, int n2) {
    return (size_t)(batch > 0 ? batch : 0) * (size_t)ggm_tiles(n0, n1, n2) * 4 * sizeof(float2);
}
static int ggm3d_batch_impl(const float *vol, int batch, int n0, int n1, int n2, double sigma, float *tmp, float *out, int accum_bits, float *range2,
                            void *range_ws, size_t range_ws_bytes, void *stream) {
    GN_REQUIRE(batch >= 0 && batch <= 65535 && n0 > 0 && n1 > 0 && n2 > 0 && sigma > 0, "gn_ggm3d: bad sizes");
    GN_REQUIRE(accum_bits == 64 || accum_bits == 32, "gn_ggm3d: accum_bits must be 64 (scipy's arithmetic, bit for bit) or 32");
    const int radius = (int)(4.0 * sigma + 0.5);
    GN_REQUIRE(radius >= 1 && radius <= 32, "gn_ggm3d: unsupported sigma");
    GN_REQUIRE(tmp != nullptr || radius <= GGM_R, "gn_ggm3d: tmp (2 volumes) is required for a kernel radius above %d", GGM_R);
    if (batch == 0) return GN_OK;
    GgmWeights w0, w1;
    ggm_kernel1d(sigma, 0, radius, w0);
    ggm_kernel1d(sigma, 1, radius, w1);
    const int64_t tot = (int64_t)n0 * n1 * n2;
    float *t1 = tmp, *t2 = tmp + (int64_t)batch * tot;
    hipStream_t st = gn_stream(stream);
    if (radius <= GGM_R) {                          // one fused launch (tmp is not touched)
        const dim3 grid((unsigned)ggm_tiles(n0, n1, n2), (unsigned)batch);
        GN_REQUIRE(range2 == nullptr || (range_ws != nullptr && range_ws_bytes >= gn_ggm3d_range_workspace_bytes(batch, n0, n1, n2)),
                   "gn_ggm3d_batch_ex: range_ws needs gn_ggm3d_range_workspace_bytes() bytes");
        float *ws = reinterpret_cast<float *>(range_ws);
        if (accum_bits == 64) {
            if (range2) hipLaunchKernelGGL((ggm_fused_kernel<double, true>), grid, dim3(256), 0, st, vol, out, n0, n1, n2, w0, w1, ws);
            else hipLaunchKernelGGL((ggm_fused_kernel<double, false>), grid, dim3(256), 0, st, vol, out, n0, n1, n2, w0, w1, ws);
        } else {
            GgmW<float> f0, f1;
            for (int i = 0; i < 65; ++i) { f0.w[i] = (float)w0.w[i]; f1.w[i] = (float)w1.w[i]; }
            f0.radius = w0.radius; f0.symmetric = w0.symmetric; f1.radius = w1.radius; f1.symmetric = w1.symmetric;
            if (range2) hipLaunchKernelGGL((ggm_fused_kernel<float, true>), grid, dim3(256), 0, st, vol, out, n0, n1, n2, f0, f1, ws);
            else hipLaunchKernelGGL((ggm_fused_kernel<float, false>), grid, dim3(256), 0, st, vol, out, n0, n1, n2, f0, f1, ws);
        }
        if (range2) hipLaunchKernelGGL(ggm_range_reduce_kernel, dim3(batch), dim3(256), 0, st, reinterpret_cast<const float2 *>(ws), (int)(grid.x * 4), range2);
        GN_LAUNCH_CHECK("gn_ggm3d");
        return GN_OK;
    }
    GN_REQUIRE(accum_bits == 64, "gn_ggm3d: the fp32 accumulation exists for the fused form only (kernel radius <= %d)", GGM_R);
    dim3 grid((unsigned)gn_cdiv(tot, 256), (unsigned)batch), block(256);
    // scipy: for axis d, correlate along axes 0, 1, 2 in turn (derivative kernel on d, Gaussian on the others, fp32 between the passes),
    // square, accumulate in the order d = 0, 1, 2, square root.  The chains of d = 1 and d = 2 both start with the Gaussian along
    // axis 0 (computed once), and every chain's last pass accumulates directly: 8 passes over the volume instead of 12.
#define GGM_PASS(SRC, DST, AXIS, W, ACC) hipLaunchKernelGGL(ggm_correlate_kernel, grid, block, 0, st, (const float *)(SRC), DST, n0, n1, n2, AXIS, W, ACC)
    GGM_PASS(vol, t1, 0, w1, 0); GGM_PASS(t1, t2, 1, w0, 0); GGM_PASS(t2, out, 2, w0, 1);     // d = 0
    GGM_PASS(vol, t1, 0, w0, 0);                                                                // shared by d = 1, 2
    GGM_PASS(t1, t2, 1, w1, 0); GGM_PASS(t2, out, 2, w0, 2);                                    // d = 1
    GGM_PASS(t1, t2, 1, w0, 0); GGM_PASS(t2, out, 2, w1, 3);                                    // d = 2
#undef GGM_PASS
    GN_LAUNCH_CHECK("gn_ggm3d");
    if (range2) return gn_minmax_batch(vol, batch, tot, range2, stream);      // the 8-pass form has no staging pass to ride on
    return GN_OK;
}

extern "C" int gn_ggm3d_batch(const float *vol, int batch, int n0, int n1, int n2, double sigma, float *tmp, float *out, void *stream) {
    return ggm3d_batch_impl(vol, batch, n0, n1, n2, sigma, tmp, out, 64, nullptr, nullptr, 0, stream);
}

extern "C" int gn_ggm3d_batch_ex(const float *vol, int batch, int n0, int n1, int n2, double sigma, float *tmp, float *out, int accum_bits, float *range2,
                                 void *range_ws, size_t range_ws_bytes, void *stream) {
    return ggm3d_batch_impl(vol, batch, n0, n1, n2, sigma, tmp, out, accum_bits, range2, range_ws, range_ws_bytes, stream);
}

extern "C" int gn_ggm3d(const float *vol, int n0, int n1, int n2, double sigma, float *tmp, float *out, void *stream) {
    return gn_ggm3d_batch(vol, 1, n0, n1, n2, sigma, tmp, out, stream);
}

// ================================================================================================ MC33
__device__ const int8_t MC_LUT_G[MC_LUT_BYTES] = MC_LUT_INITIALIZER;

__device__ __forceinline__ void mc_stage_lut(int8_t *lut) {
    const int4 *src = reinterpret_cast<const int4 *>(MC_LUT_G);
    int4 *dst = reinterpret_cast<int4 *>(lut);
    for (int i = threadIdx.x; i < MC_LUT_BYTES / 16; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

struct McDims {
    int n0, n1, n2;   // volume (axis0 = z, axis1 = y, axis2 = x)
    int c0, c1, c2;   // cells
    int64_t ncells, nvox;
    // batched launches (blockIdx.y = volume): element strides between consecutive volumes' workspaces / outputs (0 for one volume)
    int64_t s_cinfo, s_cnt, s_edge, s_bsum, s_verts, s_faces;
};

// sweep index -> (x, y, z) of a row-major (n1 x n2 inner) grid.  Grids of fewer than 2^32 elements (every lattice predict.py asks for) take two
// 32-bit unsigned divisions; the 64-bit divide / modulo triple this replaces was ~300 instructions per cell -- the whole cost of the classify pass
__device__ __forceinline__ void mc_unflatten(int64_t i, int n1, int n2, int64_t total, int &x, int &y, int &z) {
    if (total < ((int64_t)1 << 32)) {                  // (uniform)
        const unsigned u = (unsigned)i, q = u / (unsigned)n2, zz = q / (unsigned)n1;
        x = (int)(u - q * (unsigned)n2);
        y = (int)(q - zz * (unsigned)n1);
        z = (int)zz;
    } else {
        x = (int)(i % n2); y = (int)((i / n2) % n1); z = (int)(i / ((int64_t)n1 * n2));
    }
}

// corner values minus level, Lewiner numbering: v0=(0,0,0) v1=(x+1) v2=(x+1,y+1) v3=(y+1) v4..v7 at z+1
__device__ __forceinline__ void mc_load_cell(const float *__restrict__ vol, const McDims &d, int x, int y, int z, double level,
                                             double v[8]) {
    const float *p = vol + ((int64_t)z * d.n1 + y) * d.n2 + x;
    const int64_t sy = d.n2, sz = (int64_t)d.n1 * d.n2;
    v[0] = (double)p[0] - level;
    v[1] = (double)p[1] - level;
    v[2] = (double)p[sy + 1] - level;
    v[3] = (double)p[sy] - level;
    v[4] = (double)p[sz] - level;
    v[5] = (double)p[sz + 1] - level;
    v[6] = (double)p[sz + sy + 1] - level;
    v[7] = (double)p[sz + sy] - level;
}

__device__ __forceinline__ bool mc_face_test(const double *v, int face) {
    // corners (A,B,C,D) of face |f|, packed 3 bits each
    const unsigned packed[7] = {0u, 0u | (4u << 3) | (5u << 6) | (1u << 9), 1u | (5u << 3) | (6u << 6) | (2u << 9),
                                2u | (6u << 3) | (7u << 6) | (3u << 9), 3u | (7u << 3) | (4u << 6) | (0u << 9),
                                0u | (3u << 3) | (2u << 6) | (1u << 9), 4u | (7u << 3) | (6u << 6) | (5u << 9)};
    const int af = face < 0 ? -face : face;
    const unsigned pk = packed[af];
    const double A = v[pk & 7], B = v[(pk >> 3) & 7], C = v[(pk >> 6) & 7], D = v[(pk >> 9) & 7];
    const double x = __dsub_rn(__dmul_rn(A, C), __dmul_rn(B, D));
    if (x > -DBL_EPSILON && x < DBL_EPSILON) return face >= 0;
    return __dmul_rn(__dmul_rn((double)face, A), x) >= 0;
}

__device__ __forceinline__ bool mc_interior_test(const int8_t *lut, const double *v, int cas, int cfg, int sub, int s) {
    double At, Bt, Ct, Dt;
    if (cas == 4 || cas == 10) {
        const double e40 = __dsub_rn(v[4], v[0]), e62 = __dsub_rn(v[6], v[2]), e73 = __dsub_rn(v[7], v[3]), e51 = __dsub_rn(v[5], v[1]);
        const double a = __dsub_rn(__dmul_rn(e40, e62), __dmul_rn(e73, e51));
        double b = __dadd_rn(__dmul_rn(v[2], e40), __dmul_rn(v[0], e62));
        b = __dsub_rn(b, __dmul_rn(v[1], e73));
        b = __dsub_rn(b, __dmul_rn(v[3], e51));
        const double t = __ddiv_rn(-b, __dadd_rn(__dmul_rn(2.0, a), DBL_EPSILON));
        if (t < 0 || t > 1) return s > 0;
        At = __dadd_rn(v[0], __dmul_rn(e40, t));
        Bt = __dadd_rn(v[3], __dmul_rn(e73, t));
        Ct = __dadd_rn(v[2], __dmul_rn(e62, t));
        Dt = __dadd_rn(v[1], __dmul_rn(e51, t));
    } else {
        int edge = -1;
        if (cas == 6) edge = lut[MC_OFF_TEST6 + cfg * 3 + 2];
        else if (cas == 7) edge = lut[MC_OFF_TEST7 + cfg * 5 + 4];
        else if (cas == 12) edge = lut[MC_OFF_TEST12 + cfg * 4 + 3];
        else if (cas == 13) edge = lut[MC_OFF_TILING13_5_1 + (cfg * 4 + sub) * 18];
        if (edge < 0 || edge > 11) return s < 0;
        // reference edge (p,q) followed by the three edges parallel to it
        const signed char T[12][8] = {{0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
                                      {4, 5, 7, 6, 3, 2, 0, 1}, {5, 6, 4, 7, 0, 3, 1, 2}, {6, 7, 5, 4, 1, 0, 2, 3}, {7, 4, 6, 5, 2, 1, 3, 0},
                                      {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};
        const signed char *e = T[edge];
        const double t = __ddiv_rn(v[e[0]], __dadd_rn(__dsub_rn(v[e[0]], v[e[1]]), DBL_EPSILON));
        At = 0;
        Bt = __dadd_rn(v[e[2]], __dmul_rn(__dsub_rn(v[e[3]], v[e[2]]), t));
        Ct = __dadd_rn(v[e[4]], __dmul_rn(__dsub_rn(v[e[5]], v[e[4]]), t));
        Dt = __dadd_rn(v[e[6]], __dmul_rn(__dsub_rn(v[e[7]], v[e[6]]), t));
    }
    int test = 0;
    if (At >= 0) test += 1;
    if (Bt >= 0) test += 2;
    if (Ct >= 0) test += 4;
    if (Dt >= 0) test += 8;
    const double sad = __dsub_rn(__dmul_rn(At, Ct), __dmul_rn(Bt, Dt));
    switch (test) {
        case 5: return (sad < DBL_EPSILON) ? (s > 0) : false;    // pinned scikit-image behaviour (see oracle)
        case 10: return (sad >= DBL_EPSILON) ? (s > 0) : false;
        case 7: case 11: case 13: case 14: case 15: return s < 0;
        default: return s > 0;
    }
}

// -> LUT offset of the tiling row (or -1) and number of triangles
__device__ __forceinline__ int mc_resolve(const int8_t *lut, const double *v, int index, int &ntri) {
    const int cas = lut[MC_OFF_CASES + index * 2], cfg = lut[MC_OFF_CASES + index * 2 + 1];
#define R2(T, n) { ntri = (n); return MC_OFF_##T + cfg * MC_DIM1_##T; }
#define R3(T, j, n) { ntri = (n); return MC_OFF_##T + (cfg * MC_DIM1_##T + (j)) * MC_DIM2_##T; }
    int sub = 0;
    switch (cas) {
        case 1: R2(TILING1, 1)
        case 2: R2(TILING2, 2)
        case 3:
            if (mc_face_test(v, lut[MC_OFF_TEST3 + cfg])) R2(TILING3_2, 4)
            R2(TILING3_1, 2)
        case 4:
            if (mc_interior_test(lut, v, cas, cfg, 0, lut[MC_OFF_TEST4 + cfg])) R2(TILING4_1, 2)
            R2(TILING4_2, 6)
        case 5: R2(TILING5, 3)
        case 6:
            if (mc_face_test(v, lut[MC_OFF_TEST6 + cfg * 3])) R2(TILING6_2, 5)
            if (mc_interior_test(lut, v, cas, cfg, 0, lut[MC_OFF_TEST6 + cfg * 3 + 1])) R2(TILING6_1_1, 3)
            R2(TILING6_1_2, 9)
        case 7:
            for (int i = 0; i < 3; ++i)
                if (mc_face_test(v, lut[MC_OFF_TEST7 + cfg * 5 + i])) sub |= 1 << i;
            switch (sub) {
                case 0: R2(TILING7_1, 3)
                case 1: R3(TILING7_2, 0, 5)
                case 2: R3(TILING7_2, 1, 5)
                case 3: R3(TILING7_3, 0, 9)
                case 4: R3(TILING7_2, 2, 5)
                case 5: R3(TILING7_3, 1, 9)
                case 6: R3(TILING7_3, 2, 9)
                default:
                    if (mc_interior_test(lut, v, cas, cfg, 0, lut[MC_OFF_TEST7 + cfg * 5 + 3])) R2(TILING7_4_2, 9)
                    R2(TILING7_4_1, 5)
            }
        case 8: R2(TILING8, 2)
        case 9: R2(TILING9, 4)
        case 10: {
            const bool f0 = mc_face_test(v, lut[MC_OFF_TEST10 + cfg * 3]), f1 = mc_face_test(v, lut[MC_OFF_TEST10 + cfg * 3 + 1]);
            if (f0 && f1) R2(TILING10_1_1_, 4)
            if (f0) R2(TILING10_2, 8)
            if (f1) R2(TILING10_2_, 8)
            if (mc_interior_test(lut, v, cas, cfg, 0, lut[MC_OFF_TEST10 + cfg * 3 + 2])) R2(TILING10_1_1, 4)
            R2(TILING10_1_2, 8)
        }
        case 11: R2(TILING11, 4)
        case 12: {
            const bool f0 = mc_face_test(v, lut[MC_OFF_TEST12 + cfg * 4]), f1 = mc_face_test(v, lut[MC_OFF_TEST12 + cfg * 4 + 1]);
            if (f0 && f1) R2(TILING12_1_1_, 4)
            if (f0) R2(TILING12_2, 8)
            if (f1) R2(TILING12_2_, 8)
            if (mc_interior_test(lut, v, cas, cfg, 0, lut[MC_OFF_TEST12 + cfg * 4 + 2])) R2(TILING12_1_1, 4)
            R2(TILING12_1_2, 8)
        }
        case 13: {
            for (int i = 0; i < 6; ++i)
                if (mc_face_test(v, lut[MC_OFF_TEST13 + cfg * 7 + i])) sub |= 1 << i;
            const int sc = lut[MC_OFF_SUBCONFIG13 + sub];
            if (sc == 0) R2(TILING13_1, 4)
            if (sc <= 6) R3(TILING13_2, sc - 1, 6)
            if (sc <= 18) R3(TILING13_3, sc - 7, 10)
            if (sc <= 22) R3(TILING13_4, sc - 19, 12)
            if (sc <= 26) {
                const int s5 = sc - 23;
                if (mc_interior_test(lut, v, cas, cfg, s5, lut[MC_OFF_TEST13 + cfg * 7 + 6])) R3(TILING13_5_1, s5, 6)
                R3(TILING13_5_2, s5, 10)
            }
            if (sc <= 38) R3(TILING13_3_, sc - 27, 10)
            if (sc <= 44) R3(TILING13_2_, sc - 39, 6)
            if (sc == 45) R2(TILING13_1_, 4)
            break;
        }
        case 14: R2(TILING14, 4)
        default: break;
    }
#undef R2
#undef R3
    ntri = 0;
    return -1;
}

// bit e of the result is set iff local edge e (0..11) is first used by THIS cell in sweep order
__device__ __forceinline__ unsigned mc_owned_mask(int x, int y, int z) {
    const bool x0 = x == 0, y0 = y == 0, z0 = z == 0;
    unsigned m = 0;
    m |= (y0 && z0) ? 1u << 0 : 0u;   // x-edge (dy0,dz0)
    m |= z0 ? 1u << 2 : 0u;           // x-edge (dy1,dz0)
    m |= y0 ? 1u << 4 : 0u;           // x-edge (dy0,dz1)
    m |= 1u << 6;                     // x-edge (dy1,dz1)
    m |= (x0 && z0) ? 1u << 3 : 0u;   // y-edge (dx0,dz0)
    m |= z0 ? 1u << 1 : 0u;           // y-edge (dx1,dz0)
    m |= x0 ? 1u << 7 : 0u;           // y-edge (dx0,dz1)
    m |= 1u << 5;                     // y-edge (dx1,dz1)
    m |= (x0 && y0) ? 1u << 8 : 0u;   // z-edge (dx0,dy0)
    m |= y0 ? 1u << 9 : 0u;           // z-edge (dx1,dy0)
    m |= x0 ? 1u << 11 : 0u;          // z-edge (dx0,dy1)
    m |= 1u << 10;                    // z-edge (dx1,dy1)
    m |= 1u << 12;                    // centre vertex
    return m;
}

// slot of local edge e of cell (x,y,z) in the global edge->vertex table [voxel][4] (0: x-edge, 1: y-edge, 2: z-edge, 3: centre)
__device__ __forceinline__ int64_t mc_edge_slot(const McDims &d, int x, int y, int z, int e) {
    int dx = 0, dy = 0, dz = 0, j;
    if (e < 8) {
        int ee = e & 3;
        dz = e >> 2;
        if (ee == 0) j = 0;
        else if (ee == 1) { dx = 1; j = 1; }
        else if (ee == 2) { dy = 1; j = 0; }
        else j = 1;
    } else if (e < 12) {
        j = 2;
        if (e == 9) dx = 1;
        else if (e == 10) { dx = 1; dy = 1; }
        else if (e == 11) dy = 1;
    } else {
        j = 3;
    }
    return ((((int64_t)(z + dz) * d.n1 + (y + dy)) * d.n2 + (x + dx)) << 2) + j;
}

// One workgroup = one scan block of SCAN_ELEMS = 1024 cells in sweep order (4 per thread, strided by 256: coalesced).  Per cell it writes ONE
// byte (new vertices << 4 | triangles; <= 13 and <= 12) and -- only for the cells the surface crosses, a few per cent -- the tiling row
// (cinfo); the block's packed sum (nv << 32 | nt) goes straight to bsum, so the scan needs no pass of its own over the cells.
#define SCAN_ELEMS 1024
__global__ __launch_bounds__(256) void mc_classify_kernel(const float *__restrict__ vol, McDims d, double level,
                                                          int32_t *__restrict__ cinfo, unsigned char *__restrict__ cnt8,
                                                          unsigned long long *__restrict__ bsum) {
    __shared__ __attribute__((aligned(16))) int8_t lut[MC_LUT_BYTES];
    __shared__ unsigned long long wsum_c[4];
    vol += blockIdx.y * d.nvox; cinfo += blockIdx.y * d.s_cinfo; cnt8 += blockIdx.y * d.s_cnt; bsum += blockIdx.y * d.s_bsum;
    const int64_t wg0 = (int64_t)blockIdx.x * SCAN_ELEMS;
    // pass 1: the sign index of this thread's four cells only -- corner > level in double ((double)p - level > 0  <=>  (double)p > level, exactly);
    // the eight fp64 corner values are formed again, in pass 2, for the few per cent of the cells the surface crosses: 4 index registers
    // per thread across the workgroup vote instead of 64 fp64 ones
    int index[4];
    bool any = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t ci = wg0 + k * 256 + threadIdx.x;
        index[k] = 0;
        if (ci < d.ncells) {
            int x, y, z;
            mc_unflatten(ci, d.c1, d.c2, d.ncells, x, y, z);
            const float *p = vol + ((int64_t)z * d.n1 + y) * d.n2 + x;
            const int64_t sy = d.n2, sz = (int64_t)d.n1 * d.n2;
            const float c[8] = {p[0], p[1], p[sy + 1], p[sy], p[sz], p[sz + 1], p[sz + sy + 1], p[sz + sy]};     // Lewiner numbering
#pragma unroll
            for (int i = 0; i < 8; ++i) index[k] |= ((double)c[i] > level) ? (1 << i) : 0;
        }
        any |= index[k] != 0 && index[k] != 255;
    }
    // the 13.4 KB of tables are staged only by workgroups that hold a cell the surface crosses (a few per cent of them)
    if (!__syncthreads_or(any)) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int64_t ci = wg0 + k * 256 + threadIdx.x;
            if (ci < d.ncells) cnt8[ci] = 0;
        }
        if (threadIdx.x == 0) bsum[blockIdx.x] = 0;
        return;
    }
    mc_stage_lut(lut);
    unsigned long long mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t ci = wg0 + k * 256 + threadIdx.x;
        if (ci >= d.ncells) continue;
        unsigned char c8 = 0;
        if (index[k] != 0 && index[k] != 255) {
            int x, y, z;
            mc_unflatten(ci, d.c1, d.c2, d.ncells, x, y, z);
            double v[8];
            mc_load_cell(vol, d, x, y, z, level, v);
            int ntri = 0, info = -1;
            const int row = mc_resolve(lut, v, index[k], ntri);
            if (row >= 0 && ntri > 0) {
                unsigned used = 0;
                for (int q = 0; q < ntri * 3; ++q) used |= 1u << lut[row + q];
                const int nnew = __popc(used & mc_owned_mask(x, y, z));
                info = row | (ntri << 16);
                c8 = (unsigned char)((nnew << 4) | ntri);
                mine += ((unsigned long long)nnew << 32) | (unsigned)ntri;
            }
            cinfo[ci] = info;                       // (cells the surface does not cross are never looked up: their entries stay unwritten)
        }
        cnt8[ci] = c8;
    }
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63) == 0) wsum_c[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = wsum_c[0] + wsum_c[1] + wsum_c[2] + wsum_c[3];
}

// ---- exclusive scan of packed (nv<<32 | nt) counts, SCAN_ELEMS elements per block
__device__ __forceinline__ unsigned long long block_excl_scan(unsigned long long v, unsigned long long *total) {
    __shared__ unsigned long long wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        unsigned long long o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    unsigned long long base = 0, tot = 0;
    for (int w = 0; w < 4; ++w) {
        if (w < wave) base += wsum[w];
        tot += wsum[w];
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

__global__ __launch_bounds__(256) void scan_block_sums_kernel(const unsigned long long *__restrict__ in, int64_t n,
                                                              unsigned long long *__restrict__ bsum, int64_t s_in, int64_t s_bsum) {
    in += blockIdx.y * s_in; bsum += blockIdx.y * s_bsum;
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_ELEMS + threadIdx.x * 4;
    unsigned long long s = 0;
    for (int k = 0; k < 4; ++k)
        if (i0 + k < n) s += in[i0 + k];
    unsigned long long tot;
    block_excl_scan(s, &tot);
    if (threadIdx.x == 0) bsum[blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void scan_top_kernel(unsigned long long *__restrict__ bsum, int64_t nb,
                                                       unsigned long long *__restrict__ total_out, int64_t s_bsum) {
    bsum += blockIdx.y * s_bsum; total_out += blockIdx.y * s_bsum;
    __shared__ unsigned long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < nb; base += 256) {
        const int64_t i = base + threadIdx.x;
        unsigned long long v = i < nb ? bsum[i] : 0, tot;
        unsigned long long ex = block_excl_scan(v, &tot);
        const unsigned long long carry = carry_s;
        if (i < nb) bsum[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry_s;
}

__global__ __launch_bounds__(256) void scan_apply_kernel(const unsigned long long *__restrict__ in, int64_t n,
                                                         const unsigned long long *__restrict__ bsum,
                                                         unsigned long long *__restrict__ out, int64_t s_in, int64_t s_bsum) {
    in += blockIdx.y * s_in; out += blockIdx.y * s_in; bsum += blockIdx.y * s_bsum;
    const int64_t i0 = (int64_t)blockIdx.x * SCAN_ELEMS + threadIdx.x * 4;
    unsigned long long v[4], s = 0;
    for (int k = 0; k < 4; ++k) { v[k] = (i0 + k < n) ? in[i0 + k] : 0; s += v[k]; }
    unsigned long long tot;
    unsigned long long ex = block_excl_scan(s, &tot) + bsum[blockIdx.x];
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < n) out[i0 + k] = ex;
        ex += v[k];
    }
}

// inverse-|value| weights of the two end corners of local edge e; i1/i2 are BIT-ORDER corner ids (x + 2y + 4z)
__device__ __forceinline__ void mc_edge_weights(const int8_t *lut, const double *vv, int e, int &i1, int &i2, double &w1, double &w2,
                                                const int8_t *&rel) {
    rel = lut + MC_OFF_EDGEREL + e * 6;
    i1 = rel[2] * 4 + rel[1] * 2 + rel[0];
    i2 = rel[5] * 4 + rel[4] * 2 + rel[3];
    w1 = __ddiv_rn(1.0, __dadd_rn(DBL_EPSILON, fabs(vv[i1])));
    w2 = __ddiv_rn(1.0, __dadd_rn(DBL_EPSILON, fabs(vv[i2])));
}

// corner "gradients": table laid out in Lewiner corner order; scikit-image indexes it with the bit-order id for edge
// vertices (upstream quirk, reproduced) and with the Lewiner id for the centre vertex.
__device__ __forceinline__ double mc_vg(const double *v, int k, int comp) {
    switch (k * 3 + comp) {
        case 0: return __dsub_rn(v[0], v[1]); case 1: return __dsub_rn(v[0], v[3]); case 2: return __dsub_rn(v[0], v[4]);
        case 3: return __dsub_rn(v[0], v[1]); case 4: return __dsub_rn(v[1], v[2]); case 5: return __dsub_rn(v[1], v[5]);
        case 6: return __dsub_rn(v[3], v[2]); case 7: return __dsub_rn(v[1], v[2]); case 8: return __dsub_rn(v[2], v[6]);
        case 9: return __dsub_rn(v[3], v[2]); case 10: return __dsub_rn(v[0], v[3]); case 11: return __dsub_rn(v[3], v[7]);
        case 12: return __dsub_rn(v[4], v[5]); case 13: return __dsub_rn(v[4], v[7]); case 14: return __dsub_rn(v[0], v[4]);
        case 15: return __dsub_rn(v[4], v[5]); case 16: return __dsub_rn(v[5], v[6]); case 17: return __dsub_rn(v[1], v[5]);
        case 18: return __dsub_rn(v[7], v[6]); case 19: return __dsub_rn(v[5], v[6]); case 20: return __dsub_rn(v[2], v[6]);
        case 21: return __dsub_rn(v[7], v[6]); case 22: return __dsub_rn(v[4], v[7]); default: return __dsub_rn(v[3], v[7]);
    }
}

__device__ __forceinline__ void mc_bit_order(const double *v, double *vv) {
    vv[0] = v[0]; vv[1] = v[1]; vv[2] = v[3]; vv[3] = v[2]; vv[4] = v[4]; vv[5] = v[5]; vv[6] = v[7]; vv[7] = v[6];
}

// A workgroup of the emit kernels = one scan block (SCAN_ELEMS cells): it rebuilds the exclusive prefix of its cells' packed counts from
// the count bytes (4 consecutive cells per thread, block_excl_scan + the block's offset from scan_top_kernel) -- no per-cell offset
// array in HBM -- and queues the cells the surface crosses (a few per cent) in LDS with their offsets, to be worked off by consecutive
// lanes.  -> number of queued cells
__device__ __forceinline__ unsigned mc_scan_queue(const unsigned char *__restrict__ cnt8, const unsigned long long *__restrict__ bsum, const McDims &d,
                                                  int64_t wg0, unsigned short *queue, unsigned long long *qoff, unsigned *qn) {
    if (threadIdx.x == 0) *qn = 0;
    const int64_t i0 = wg0 + threadIdx.x * 4;
    unsigned c4 = 0;
    if (i0 + 3 < d.ncells) c4 = *reinterpret_cast<const unsigned *>(cnt8 + i0);       // (i0 and the per-volume base are multiples of 4)
    else
        for (int k = 0; k < 4; ++k)
            if (i0 + k < d.ncells) c4 |= (unsigned)cnt8[i0 + k] << (8 * k);
    unsigned long long v[4], ssum = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const unsigned c = (c4 >> (8 * k)) & 0xffu;
        v[k] = ((unsigned long long)(c >> 4) << 32) | (c & 15u);
        ssum += v[k];
    }
    unsigned long long tot;
    unsigned long long ex = block_excl_scan(ssum, &tot) + bsum[blockIdx.x];          // (its barriers also publish *qn = 0)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (v[k] != 0) {
            const unsigned slot = atomicAdd(qn, 1u);
            queue[slot] = (unsigned short)(threadIdx.x * 4 + k);
            qoff[slot] = ex;
        }
        ex += v[k];
    }
    __syncthreads();
    return *qn;
}

// owners: vertex ids, positions, edge table
__device__ __forceinline__ void mc_vertices_one(const float *__restrict__ vol, const McDims &d, double level, const int8_t *lut, int64_t ci, int info,
                                                unsigned long long off, const int32_t *__restrict__ cinfo, int32_t *__restrict__ edge_vid,
                                                float *__restrict__ verts, float *__restrict__ normals, float *__restrict__ values, int64_t cap_v) {
    const int row = info & 0xffff, ntri = info >> 16;
    int x, y, z;
            mc_unflatten(ci, d.c1, d.c2, d.ncells, x, y, z);
    const unsigned owned = mc_owned_mask(x, y, z);
    unsigned seen = 0;
    int64_t vid = (int64_t)(off >> 32);
    double v[8], vv[8];
    bool loaded = false;
    for (int k = 0; k < ntri * 3; ++k) {
        const int e = lut[row + k];
        if (seen & (1u << e)) continue;
        seen |= 1u << e;
        if (!(owned & (1u << e))) continue;
        if (!loaded) { mc_load_cell(vol, d, x, y, z, level, v); mc_bit_order(v, vv); loaded = true; }
        double px, py, pz;
        if (e == 12) {
            double fx = 0, fy = 0, fz = 0, ff = 0;
            const int LC[8] = {0, 1, 3, 2, 4, 5, 7, 6};  // Lewiner corner -> bit-order offsets (x | y<<1 | z<<2)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const double w = __ddiv_rn(1.0, __dadd_rn(DBL_EPSILON, fabs(v[c])));
                fx = __dadd_rn(fx, __dmul_rn((double)(LC[c] & 1), w));
                fy = __dadd_rn(fy, __dmul_rn((double)((LC[c] >> 1) & 1), w));
                fz = __dadd_rn(fz, __dmul_rn((double)(LC[c] >> 2), w));
                ff = __dadd_rn(ff, w);
            }
            px = __dadd_rn((double)x, __ddiv_rn(fx, ff));
            py = __dadd_rn((double)y, __ddiv_rn(fy, ff));
            pz = __dadd_rn((double)z, __ddiv_rn(fz, ff));
        } else {
            int i1, i2;
            double w1, w2;
            const int8_t *rel;
            mc_edge_weights(lut, vv, e, i1, i2, w1, w2, rel);
            const double fx = __dadd_rn(__dmul_rn((double)rel[0], w1), __dmul_rn((double)rel[3], w2));
            const double fy = __dadd_rn(__dmul_rn((double)rel[1], w1), __dmul_rn((double)rel[4], w2));
            const double fz = __dadd_rn(__dmul_rn((double)rel[2], w1), __dmul_rn((double)rel[5], w2));
            const double ff = __dadd_rn(w1, w2);
            px = __dadd_rn((double)x, __ddiv_rn(fx, ff));
            py = __dadd_rn((double)y, __ddiv_rn(fy, ff));
            pz = __dadd_rn((double)z, __ddiv_rn(fz, ff));
        }
        const int64_t slot = mc_edge_slot(d, x, y, z, e);
        edge_vid[slot] = (int32_t)vid;
        if (vid < cap_v) {
            verts[vid * 3 + 0] = (float)pz;  // array-axis order (axis0, axis1, axis2)
            verts[vid * 3 + 1] = (float)py;
            verts[vid * 3 + 2] = (float)px;
            // the vertex's edge slot is parked in its (not yet computed) normal row: mc_attrs_kernel, one thread per VERTEX, picks it up
            // (a gather inlined here, inside the per-cell loop, was miscompiled by hipcc 7.2 -O3: wrong edge for some vertices, caught by
            //  the bit-exact goldens; as its own dense pass over the vertices it is also perfectly balanced)
            reinterpret_cast<unsigned *>(normals)[vid * 3 + 0] = (unsigned)((unsigned long long)slot & 0xffffffffull);
            reinterpret_cast<unsigned *>(normals)[vid * 3 + 1] = (unsigned)((unsigned long long)slot >> 32);
        }
        ++vid;
    }
}

__global__ __launch_bounds__(256) void mc_vertices_kernel(const float *__restrict__ vol, McDims d, double level,
                                                          const int32_t *__restrict__ cinfo, const unsigned char *__restrict__ cnt8,
                                                          const unsigned long long *__restrict__ bsum, int32_t *__restrict__ edge_vid,
                                                          float *__restrict__ verts, float *__restrict__ normals, float *__restrict__ values, int64_t cap_v) {
    __shared__ __attribute__((aligned(16))) int8_t lut[MC_LUT_BYTES];
    __shared__ unsigned short queue[SCAN_ELEMS];
    __shared__ unsigned long long qoff[SCAN_ELEMS];
    __shared__ unsigned qn;
    vol += blockIdx.y * d.nvox; cinfo += blockIdx.y * d.s_cinfo; cnt8 += blockIdx.y * d.s_cnt; bsum += blockIdx.y * d.s_bsum;
    edge_vid += blockIdx.y * d.s_edge; verts += blockIdx.y * d.s_verts; normals += blockIdx.y * d.s_verts; values += blockIdx.y * (d.s_verts / 3);
    const int64_t wg0 = (int64_t)blockIdx.x * SCAN_ELEMS;
    const unsigned cnt = mc_scan_queue(cnt8, bsum, d, wg0, queue, qoff, &qn);
    if (cnt == 0) return;                           // (workgroup-uniform) no surface cell here: no table staging
    mc_stage_lut(lut);
    for (unsigned k = threadIdx.x; k < cnt; k += 256) {
        const int64_t ci = wg0 + queue[k];
        mc_vertices_one(vol, d, level, lut, ci, cinfo[ci], qoff[k], cinfo, edge_vid, verts, normals, values, cap_v);
    }
}

__global__ __launch_bounds__(256) void mc_faces_kernel(McDims d, const int32_t *__restrict__ cinfo, const unsigned char *__restrict__ cnt8,
                                                       const unsigned long long *__restrict__ bsum,
                                                       const int32_t *__restrict__ edge_vid, int32_t *__restrict__ faces, int64_t cap_f) {
    __shared__ __attribute__((aligned(16))) int8_t lut[MC_LUT_BYTES];
    __shared__ unsigned short queue[SCAN_ELEMS];
    __shared__ unsigned long long qoff[SCAN_ELEMS];
    __shared__ unsigned qn;
    cinfo += blockIdx.y * d.s_cinfo; cnt8 += blockIdx.y * d.s_cnt; bsum += blockIdx.y * d.s_bsum; edge_vid += blockIdx.y * d.s_edge; faces += blockIdx.y * d.s_faces;
    const int64_t wg0 = (int64_t)blockIdx.x * SCAN_ELEMS;
    const unsigned cnt = mc_scan_queue(cnt8, bsum, d, wg0, queue, qoff, &qn);
    if (cnt == 0) return;
    mc_stage_lut(lut);
    for (unsigned q = threadIdx.x; q < cnt; q += 256) {
        const int64_t ci = wg0 + queue[q];
        const int info = cinfo[ci];
        const int row = info & 0xffff, ntri = info >> 16;
        int x, y, z;
            mc_unflatten(ci, d.c1, d.c2, d.ncells, x, y, z);
        const int64_t t0 = (int64_t)(qoff[q] & 0xffffffffull);
        for (int k = 0; k < ntri * 3; ++k) {
            const int64_t f = t0 + k / 3;
            if (f < cap_f) faces[f * 3 + k % 3] = edge_vid[mc_edge_slot(d, x, y, z, lut[row + k])];
        }
    }
}

// per-vertex gather of normals / values over the adjacent cells in sweep order.  One thread per (voxel, slot).
// one (voxel, slot) entry t that holds vertex vid
__device__ __forceinline__ void mc_attr_one(const float *__restrict__ vol, const McDims &d, double level, const int32_t *__restrict__ cinfo,
                                            const int8_t *lut, int64_t t, int64_t vid, float *__restrict__ normals, float *__restrict__ values) {
    const int j = (int)(t & 3);
    const int64_t vx = t >> 2;
    int x, y, z;
    mc_unflatten(vx, d.n1, d.n2, d.nvox, x, y, z);
    // local edge id seen from the adjacent cell at offset (-a,-b) in the two transverse axes, a = fast transverse axis
    //   x-edge: transverse (y,z): e(a,b) = {0,2,4,6}[a + 2b] ; y-edge: transverse (x,z): {3,1,7,5} ; z-edge: (x,y): {8,9,11,10}
    // (packed 4 bits per entry [j][a + 2b]: a table in memory is not needed)
    const unsigned long long EL = 0xAB9857136420ull;
    float nx = 0.f, ny = 0.f, nz = 0.f, val = 0.f;
    const int ncell = (j == 3) ? 1 : 4;
    for (int q = 0; q < ncell; ++q) {
        int cx = x, cy = y, cz = z, e = 12;
        if (j < 3) {
            const int a = (q & 1) ? 0 : 1, b = (q & 2) ? 0 : 1;  // sweep order: (1,1),(0,1),(1,0),(0,0)
            if (j == 0) { cy = y - a; cz = z - b; }
            else if (j == 1) { cx = x - a; cz = z - b; }
            else { cx = x - a; cy = y - b; }
            e = (int)((EL >> (4 * (4 * j + a + 2 * b))) & 15ull);
        }
        if (cx < 0 || cy < 0 || cz < 0 || cx >= d.c2 || cy >= d.c1 || cz >= d.c0) continue;
        const int info = cinfo[((int64_t)cz * d.c1 + cy) * d.c2 + cx];
        if (info < 0) continue;
        const int row = info & 0xffff, ntri = info >> 16;
        int uses = 0;
        for (int k = 0; k < ntri * 3; ++k) uses += (lut[row + k] == e) ? 1 : 0;
        if (uses == 0) continue;
        double v[8], vv[8];
        mc_load_cell(vol, d, cx, cy, cz, level, v);
        mc_bit_order(v, vv);
        double vmin = 0.0, vmax = 0.0;
#pragma unroll
        for (int c = 0; c < 8; ++c) { vmax = v[c] > vmax ? v[c] : vmax; vmin = v[c] < vmin ? v[c] : vmin; }
        const float span = (float)__dsub_rn(vmax, vmin);
        val = span > val ? span : val;
        if (e == 12) {
            // pinned scikit-image quirk: centre gradient stored as (x <- z-sum, y <- y-sum, z <- 0)
            double gy = 0, gz = 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const double w = __ddiv_rn(1.0, __dadd_rn(DBL_EPSILON, fabs(v[c])));
                gy = __dadd_rn(gy, __dmul_rn(w, mc_vg(v, c, 1)));
                gz = __dadd_rn(gz, __dmul_rn(w, mc_vg(v, c, 2)));
            }
            const float cgx = (float)gz, cgy = (float)gy;
            for (int u = 0; u < uses; ++u) { nx = __fadd_rn(nx, cgx); ny = __fadd_rn(ny, cgy); nz = __fadd_rn(nz, 0.f); }
        } else {
            int i1, i2;
            double w1, w2;
            const int8_t *rel;
            mc_edge_weights(lut, vv, e, i1, i2, w1, w2, rel);
            const float s1 = (float)w1, s2 = (float)w2;
            const float g1x = (float)__dmul_rn(mc_vg(v, i1, 0), (double)s1), g1y = (float)__dmul_rn(mc_vg(v, i1, 1), (double)s1),
                        g1z = (float)__dmul_rn(mc_vg(v, i1, 2), (double)s1);
            const float g2x = (float)__dmul_rn(mc_vg(v, i2, 0), (double)s2), g2y = (float)__dmul_rn(mc_vg(v, i2, 1), (double)s2),
                        g2z = (float)__dmul_rn(mc_vg(v, i2, 2), (double)s2);
            for (int u = 0; u < uses; ++u) {
                nx = __fadd_rn(__fadd_rn(nx, g1x), g2x);
                ny = __fadd_rn(__fadd_rn(ny, g1y), g2y);
                nz = __fadd_rn(__fadd_rn(nz, g1z), g2z);
            }
        }
    }
    const double a = nz, b = ny, c = nx;  // array-axis order
    const double len = sqrt(__dadd_rn(__dadd_rn(__dmul_rn(a, a), __dmul_rn(b, b)), __dmul_rn(c, c)));
    float o0 = nz, o1 = ny, o2 = nx;
    if (len > 0.0) { o0 = (float)__ddiv_rn(a, len); o1 = (float)__ddiv_rn(b, len); o2 = (float)__ddiv_rn(c, len); }
    normals[vid * 3 + 0] = o0;
    normals[vid * 3 + 1] = o1;
    normals[vid * 3 + 2] = o2;
    values[vid] = val;
}

// normals / values: one thread per vertex, the vertex's edge slot read back from its normal row; the gather visits the (<= 4) cells that share the edge in
// sweep order -- they are all crossed by the surface, so classify has written their tiling rows.  (Round 6: 256 instead of 1024 vertices per workgroup -- a
// vertex is one long dependent chain of fp64 gathers, four of them back to back per thread left a single garment's 38 000 vertices on 38 workgroups: 79 us.)
#ifndef MC_ATTR_PER_WG
#define MC_ATTR_PER_WG 256
#endif
__global__ __launch_bounds__(256) void mc_attrs_kernel(const float *__restrict__ vol, McDims d, double level, const int32_t *__restrict__ cinfo,
                                                       const int64_t *__restrict__ counts_dev, float *__restrict__ normals, float *__restrict__ values,
                                                       int64_t cap_v) {
    __shared__ __attribute__((aligned(16))) int8_t lut[MC_LUT_BYTES];
    int64_t nv = counts_dev[2 * blockIdx.y];
    if (nv > cap_v) nv = cap_v;
    const int64_t v0 = (int64_t)blockIdx.x * MC_ATTR_PER_WG;
    if (v0 >= nv) return;                           // (workgroup-uniform)
    vol += blockIdx.y * d.nvox; cinfo += blockIdx.y * d.s_cinfo; normals += blockIdx.y * d.s_verts; values += blockIdx.y * (d.s_verts / 3);
    mc_stage_lut(lut);
    for (int k = 0; k < MC_ATTR_PER_WG / 256; ++k) {
        const int64_t vid = v0 + k * 256 + threadIdx.x;
        if (vid >= nv) break;
        const unsigned *w = reinterpret_cast<const unsigned *>(normals) + vid * 3;
        const int64_t t = (int64_t)((unsigned long long)w[0] | ((unsigned long long)w[1] << 32));
        mc_attr_one(vol, d, level, cinfo, lut, t, vid, normals, values);
    }
}

__global__ void mc_counts_kernel(const unsigned long long *__restrict__ total, int64_t *__restrict__ counts_dev, int64_t s_bsum) {
    total += blockIdx.x * s_bsum; counts_dev += 2 * blockIdx.x;
    counts_dev[0] = (int64_t)(*total >> 32);
    counts_dev[1] = (int64_t)(*total & 0xffffffffull);
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t mc33_ws_one(int n0, int n1, int n2) {
    const size_t ncells = (size_t)(n0 - 1) * (n1 - 1) * (n2 - 1), nvox = (size_t)n0 * n1 * n2;
    const size_t nb = (ncells + SCAN_ELEMS - 1) / SCAN_ELEMS;
    return align256(ncells * 4) + align256(ncells) + align256(nvox * 16) + align256((nb + 1) * 8);
}

extern "C" size_t gn_mc33_workspace_bytes(int n0, int n1, int n2) {
    if (n0 < 2 || n1 < 2 || n2 < 2) return 0;
    return mc33_ws_one(n0, n1, n2) + 256;
}

extern "C" size_t gn_mc33_batch_workspace_bytes(int batch, int n0, int n1, int n2) {
    if (n0 < 2 || n1 < 2 || n2 < 2 || batch < 1) return 0;
    return (size_t)batch * mc33_ws_one(n0, n1, n2) + 256;
}

// `batch` volumes of the same shape and level in one set of launches (blockIdx.y = volume): vol [batch][n0][n1][n2], verts / normals
// [batch][cap_v][3], faces [batch][cap_f][3], values [batch][cap_v], counts_dev [batch][2].  Workspace, array by array: [batch] tiling rows
// (int32 per cell, written for the cells the surface crosses only) | [batch] count bytes | [batch] edge -> vertex tables (int32 x 4 per
// voxel; written by the vertex owners, read by the faces of the cells that share the edge: never cleared, never scanned) | [batch] block
// sums.  Six launches: classify (+ block sums) -> scan of the block sums -> counts -> vertices -> normals / values (one thread per vertex) -> faces.
// Every volume's result is what gn_mc33 gives for it alone.  stage_ms (host, 4 floats) != NULL: the call brackets its stages with HIP
// events and SYNCHRONISES to fill (classify, scan + counts, vertices + attributes, faces) in milliseconds -- bench.py's hbm_members.
static int mc33_batch_impl(const float *vol, int batch, int n0, int n1, int n2, double level, void *ws, size_t ws_bytes, float *verts,
                           int32_t *faces, float *normals, float *values, int64_t cap_v, int64_t cap_f, int64_t *counts_dev, void *stream,
                           float *stage_ms) {
    GN_REQUIRE(n0 >= 2 && n1 >= 2 && n2 >= 2, "gn_mc33: input volume must be at least 2x2x2");
    GN_REQUIRE(batch >= 0 && batch <= 65535, "gn_mc33_batch: bad batch");
    if (batch == 0) return GN_OK;
    GN_REQUIRE(ws != nullptr && ws_bytes >= gn_mc33_batch_workspace_bytes(batch, n0, n1, n2), "gn_mc33: workspace too small");
    GN_REQUIRE(cap_v >= 0 && cap_f >= 0, "gn_mc33: bad capacities");
    McDims d;
    d.n0 = n0; d.n1 = n1; d.n2 = n2; d.c0 = n0 - 1; d.c1 = n1 - 1; d.c2 = n2 - 1;
    d.ncells = (int64_t)d.c0 * d.c1 * d.c2;
    d.nvox = (int64_t)n0 * n1 * n2;
    const int64_t nb = gn_cdiv(d.ncells, SCAN_ELEMS);
    d.s_cinfo = (int64_t)(align256(d.ncells * 4) / 4);
    d.s_cnt = (int64_t)align256(d.ncells);
    d.s_edge = (int64_t)(align256(d.nvox * 16) / 4);
    d.s_bsum = (int64_t)(align256((nb + 1) * 8) / 8);
    d.s_verts = cap_v * 3;
    d.s_faces = cap_f * 3;
    char *p = (char *)ws;
    int32_t *cinfo = (int32_t *)p; p += (size_t)batch * d.s_cinfo * 4;
    unsigned char *cnt8 = (unsigned char *)p; p += (size_t)batch * d.s_cnt;
    int32_t *edge_vid = (int32_t *)p; p += (size_t)batch * d.s_edge * 4;
    unsigned long long *bsum = (unsigned long long *)p;
    unsigned long long *total = bsum + nb;
    hipStream_t st = gn_stream(stream);
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    if (stage_ms)
        for (int i = 0; i < 5; ++i) GN_HIP(hipEventCreate(&ev[i]), "gn_mc33(event)");
#define MC_MARK(i) do { if (stage_ms) GN_HIP(hipEventRecord(ev[i], st), "gn_mc33(event)"); } while (0)
    const unsigned nby = (unsigned)batch;
    const dim3 blk(256), gblk((unsigned)nb, nby);
    MC_MARK(0);
    hipLaunchKernelGGL(mc_classify_kernel, gblk, blk, 0, st, vol, d, level, cinfo, cnt8, bsum);
    MC_MARK(1);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1, nby), blk, 0, st, bsum, nb, total, d.s_bsum);
    hipLaunchKernelGGL(mc_counts_kernel, dim3(nby), dim3(1), 0, st, total, counts_dev, d.s_bsum);
    MC_MARK(2);
    hipLaunchKernelGGL(mc_vertices_kernel, gblk, blk, 0, st, vol, d, level, cinfo, cnt8, bsum, edge_vid, verts, normals, values, cap_v);
    if (cap_v > 0)
        hipLaunchKernelGGL(mc_attrs_kernel, dim3((unsigned)gn_cdiv(cap_v, MC_ATTR_PER_WG), nby), blk, 0, st, vol, d, level, cinfo, counts_dev, normals, values, cap_v);
    MC_MARK(3);
    hipLaunchKernelGGL(mc_faces_kernel, gblk, blk, 0, st, d, cinfo, cnt8, bsum, edge_vid, faces, cap_f);
    MC_MARK(4);
#undef MC_MARK
    GN_LAUNCH_CHECK("gn_mc33");
    if (stage_ms) {
        GN_HIP(hipEventSynchronize(ev[4]), "gn_mc33(event)");
        for (int i = 0; i < 4; ++i) GN_HIP(hipEventElapsedTime(&stage_ms[i], ev[i], ev[i + 1]), "gn_mc33(event)");
        for (int i = 0; i < 5; ++i) (void)hipEventDestroy(ev[i]);
    }
    return GN_OK;
}

extern "C" int gn_mc33_batch(const float *vol, int batch, int n0, int n1, int n2, double level, void *ws, size_t ws_bytes, float *verts,
                             int32_t *faces, float *normals, float *values, int64_t cap_v, int64_t cap_f, int64_t *counts_dev, void *stream) {
    return mc33_batch_impl(vol, batch, n0, n1, n2, level, ws, ws_bytes, verts, faces, normals, values, cap_v, cap_f, counts_dev, stream, nullptr);
}

extern "C" int gn_mc33_batch_profiled(const float *vol, int batch, int n0, int n1, int n2, double level, void *ws, size_t ws_bytes, float *verts,
                                      int32_t *faces, float *normals, float *values, int64_t cap_v, int64_t cap_f, int64_t *counts_dev, void *stream,
                                      float *stage_ms) {
    GN_REQUIRE(stage_ms != nullptr, "gn_mc33_batch_profiled: stage_ms (host float[4]) is required");
    return mc33_batch_impl(vol, batch, n0, n1, n2, level, ws, ws_bytes, verts, faces, normals, values, cap_v, cap_f, counts_dev, stream, stage_ms);
}

extern "C" int gn_mc33(const float *vol, int n0, int n1, int n2, double level, void *ws, size_t ws_bytes, float *verts,
                       int32_t *faces, float *normals, float *values, int64_t cap_v, int64_t cap_f, int64_t *counts_dev,
                       void *stream) {
    GN_REQUIRE(ws != nullptr && ws_bytes >= gn_mc33_workspace_bytes(n0, n1, n2), "gn_mc33: workspace too small");
    return gn_mc33_batch(vol, 1, n0, n1, n2, level, ws, ws_bytes, verts, faces, normals, values, cap_v, cap_f, counts_dev, stream);
}

// ================================================================================================ vertex helpers
__global__ __launch_bounds__(256) void gather_nn_kernel(const float *__restrict__ vol, int n0, int n1, int n2,
                                                        const float *__restrict__ verts_vox, int64_t nv, double spacing,
                                                        float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nv) return;
    vol += (int64_t)blockIdx.y * n0 * n1 * n2;      // batched: volume, nv vertex rows and nv outputs per blockIdx.y
    verts_vox += (int64_t)blockIdx.y * nv * 3;
    out += (int64_t)blockIdx.y * nv;
    int id[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        // numpy: (float32 vert * float64 spacing) / spacing -> astype(uint32)
        const double scaled = __dmul_rn((double)verts_vox[i * 3 + a], spacing);
        const double q = __ddiv_rn(scaled, spacing);
        long long k = (long long)q;
        const int lim = (a == 0 ? n0 : (a == 1 ? n1 : n2)) - 1;
        k = k < 0 ? 0 : (k > lim ? lim : k);
        id[a] = (int)k;
    }
    out[i] = vol[((int64_t)id[0] * n1 + id[1]) * n2 + id[2]];
}

extern "C" int gn_gather_nn_batch(const float *vol, int batch, int n0, int n1, int n2, const float *verts_vox, int64_t nv, double spacing,
                                  float *out, void *stream) {
    GN_REQUIRE(batch >= 0 && batch <= 65535 && n0 > 0 && n1 > 0 && n2 > 0 && nv >= 0 && spacing > 0, "gn_gather_nn: bad sizes");
    if (nv == 0 || batch == 0) return GN_OK;
    hipLaunchKernelGGL(gather_nn_kernel, dim3((unsigned)gn_cdiv(nv, 256), (unsigned)batch), dim3(256), 0, gn_stream(stream), vol, n0, n1, n2,
                       verts_vox, nv, spacing, out);
    GN_LAUNCH_CHECK("gn_gather_nn");
    return GN_OK;
}

extern "C" int gn_gather_nn(const float *vol, int n0, int n1, int n2, const float *verts_vox, int64_t nv, double spacing,
                            float *out, void *stream) {
    return gn_gather_nn_batch(vol, 1, n0, n1, n2, verts_vox, nv, spacing, out, stream);
}

__global__ __launch_bounds__(256) void scale_verts_kernel(const float *__restrict__ in, int64_t n3, double spacing,
                                                          float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) out[i] = (float)__dmul_rn((double)in[i], spacing);
}

extern "C" int gn_scale_verts(const float *verts_vox, int64_t nv, double spacing, float *verts_out, void *stream) {
    GN_REQUIRE(nv >= 0, "gn_scale_verts: bad sizes");
    if (nv == 0) return GN_OK;
    hipLaunchKernelGGL(scale_verts_kernel, dim3((unsigned)gn_cdiv(nv * 3, 256)), dim3(256), 0, gn_stream(stream), verts_vox, nv * 3,
                       spacing, verts_out);
    GN_LAUNCH_CHECK("gn_scale_verts");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ mesh compaction
// delete_invalid_verts (common/marching_cubes_util.py:38-52; eval.py / predict.py's hole head): keep the faces whose three vertices
// are on the surface, keep the vertices those faces use (ascending raw index = np.unique order), renumber.  One flag word per index
// i (low 32 bits: vertex i is used, high 32 bits: face i is kept), the three-kernel exclusive scan of the MC33 stage over the packed
// pair, one scatter pass.  Deterministic; sizes come back in counts (device int64 [3]: vertices kept, faces kept, 1 if a face index
// was outside [0, V)), outputs have room for everything.
// A face index outside [0, V) (the reference's numpy indexing raises IndexError for it) never touches memory: the face is dropped and
// bit 62 of flags[0]'s neighbour word `bad` is raised; gn_mesh_compact reports it through counts[2].
__global__ __launch_bounds__(256) void compact_mark_kernel(const int32_t *__restrict__ faces, const unsigned char *__restrict__ on_surface,
                                                           int64_t V, int64_t F, unsigned long long *__restrict__ flags, unsigned *__restrict__ bad) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= F) return;
    const int32_t a = faces[3 * f], b = faces[3 * f + 1], c = faces[3 * f + 2];
    if (a < 0 || b < 0 || c < 0 || a >= V || b >= V || c >= V) { atomicOr(bad, 1u); return; }
    if (on_surface[a] && on_surface[b] && on_surface[c]) {
        atomicOr(reinterpret_cast<unsigned *>(flags + a), 1u);
        atomicOr(reinterpret_cast<unsigned *>(flags + b), 1u);
        atomicOr(reinterpret_cast<unsigned *>(flags + c), 1u);
        atomicOr(reinterpret_cast<unsigned *>(flags + f) + 1, 1u);
    }
}

__global__ __launch_bounds__(256) void compact_scatter_kernel(const unsigned char *__restrict__ verts, int vert_bytes, const int32_t *__restrict__ faces,
                                                              int64_t V, int64_t F, const unsigned long long *__restrict__ flags,
                                                              const unsigned long long *__restrict__ ex, unsigned char *__restrict__ out_verts,
                                                              int32_t *__restrict__ out_faces) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < V && (flags[i] & 1ull)) {
        const int64_t nv = (int64_t)(ex[i] & 0xffffffffull);
        for (int k = 0; k < vert_bytes; k += 4)
            *reinterpret_cast<uint32_t *>(out_verts + nv * vert_bytes + k) = *reinterpret_cast<const uint32_t *>(verts + i * vert_bytes + k);
    }
    if (i < F && (flags[i] >> 32)) {
        const int64_t nf = (int64_t)(ex[i] >> 32);
#pragma unroll
        for (int k = 0; k < 3; ++k) out_faces[3 * nf + k] = (int32_t)(ex[faces[3 * i + k]] & 0xffffffffull);
    }
}

__global__ void unpack_counts_kernel(const unsigned long long *__restrict__ total, const unsigned *__restrict__ bad, int64_t *__restrict__ counts) {
    counts[0] = (int64_t)(*total & 0xffffffffull);
    counts[1] = (int64_t)(*total >> 32);
    counts[2] = (int64_t)*bad;
}

extern "C" size_t gn_mesh_compact_workspace_bytes(int64_t V, int64_t F) {
    const size_t n = (size_t)(V > F ? V : F), nb = (n + SCAN_ELEMS - 1) / SCAN_ELEMS;
    return sizeof(unsigned long long) * (2 * n + nb + 3);
}

extern "C" int gn_mesh_compact(const void *verts, int vert_bytes, const int32_t *faces, const unsigned char *on_surface, int64_t V, int64_t F,
                               void *ws, size_t ws_bytes, void *out_verts, int32_t *out_faces, int64_t *counts, void *stream) {
    GN_REQUIRE(V >= 0 && F >= 0 && V < (1ll << 31) && F < (1ll << 31) && vert_bytes > 0 && vert_bytes % 4 == 0, "gn_mesh_compact: bad sizes");
    GN_REQUIRE(ws_bytes >= gn_mesh_compact_workspace_bytes(V, F), "gn_mesh_compact: workspace too small");
    hipStream_t st = gn_stream(stream);
    const int64_t n = V > F ? V : F;
    if (n == 0 || F == 0) {                         // (V == 0 with faces: every index is out of range -> the bad flag below)
        GN_HIP(hipMemsetAsync(counts, 0, 3 * sizeof(int64_t), st), "gn_mesh_compact");
        return GN_OK;
    }
    const int64_t nb = gn_cdiv(n, SCAN_ELEMS);
    unsigned long long *flags = reinterpret_cast<unsigned long long *>(ws), *ex = flags + n, *bsum = ex + n, *total = bsum + nb;
    unsigned *bad = reinterpret_cast<unsigned *>(total + 1);
    GN_HIP(hipMemsetAsync(flags, 0, sizeof(unsigned long long) * (size_t)n, st), "gn_mesh_compact");
    GN_HIP(hipMemsetAsync(bad, 0, sizeof(unsigned long long), st), "gn_mesh_compact");
    hipLaunchKernelGGL(compact_mark_kernel, dim3((unsigned)gn_cdiv(F, 256)), dim3(256), 0, st, faces, on_surface, V, F, flags, bad);
    hipLaunchKernelGGL(scan_block_sums_kernel, dim3((unsigned)nb), dim3(256), 0, st, flags, n, bsum, (int64_t)0, (int64_t)0);
    hipLaunchKernelGGL(scan_top_kernel, dim3(1), dim3(256), 0, st, bsum, nb, total, (int64_t)0);
    hipLaunchKernelGGL(scan_apply_kernel, dim3((unsigned)nb), dim3(256), 0, st, flags, n, bsum, ex, (int64_t)0, (int64_t)0);
    hipLaunchKernelGGL(compact_scatter_kernel, dim3((unsigned)gn_cdiv(n, 256)), dim3(256), 0, st, (const unsigned char *)verts, vert_bytes, faces, V, F,
                       flags, ex, (unsigned char *)out_verts, out_faces);
    // counts = (vertices kept, faces kept): unpack the scan total on the device
    hipLaunchKernelGGL(unpack_counts_kernel, dim3(1), dim3(1), 0, st, total, bad, counts);
    GN_LAUNCH_CHECK("gn_mesh_compact");
    return GN_OK;
}
