// points.hip -- PointNet++ point-set operators for gfx950 (wave64).
//   gn_segment_ptr, gn_fps, gn_ball_query, gn_sa_gather, gn_segment_max, gn_global_max_pool,
//   gn_knn_interpolate, gn_nocs_head
// Reference call sites: /root/reference/components/pointnet2.py:22-76, networks/conv_implicit_wnf.py:220-231.
#include <stdarg.h>
#include <limits.h>

#include "common.h"

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void gn_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char *gn_last_error(void) { return g_err; }
extern "C" int gn_version(void) { return 200; }
hipStream_t gn_stream(void *s) {
    hipStream_t st = (hipStream_t)s;
    if (st) {
        hipDevice_t dev = 0;
        int cur = 0;
        if (hipStreamGetDevice(st, &dev) == hipSuccess && hipGetDevice(&cur) == hipSuccess && cur != (int)dev) (void)hipSetDevice((int)dev);
    }
    return st;
}
static thread_local const char *g_last_kernel = "";
void gn_note_kernel(const char *name) { g_last_kernel = name; }
extern "C" const char *gn_last_kernel(void) { return g_last_kernel; }
extern "C" int gn_device_info(int *num_cu, int *lds_bytes_per_cu) {
    int dev = 0;
    hipDeviceProp_t p;
    GN_HIP(hipGetDevice(&dev), "gn_device_info");
    GN_HIP(hipGetDeviceProperties(&p, dev), "gn_device_info");
    if (num_cu) *num_cu = p.multiProcessorCount;
    if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ segment ptr
__global__ void segment_ptr_kernel(const int64_t *__restrict__ batch, int64_t n, int B, int32_t *__restrict__ ptr) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n) return;
    // ptr[b] = first i with batch[i] >= b ; element i opens segments (batch[i-1], batch[i]]
    int64_t lo = (i == 0) ? -1 : batch[i - 1];
    int64_t hi = (i == n) ? (int64_t)B : batch[i];
    if (hi > B) hi = B;
    for (int64_t b = lo + 1; b <= hi; ++b) ptr[b] = (int32_t)i;
}

extern "C" int gn_segment_ptr(const int64_t *batch, int64_t n, int B, int32_t *ptr, void *stream) {
    GN_REQUIRE(n >= 0 && B >= 0 && n < INT32_MAX, "gn_segment_ptr: bad sizes");
    int64_t blocks = gn_cdiv(n + 1, 256);
    hipLaunchKernelGGL(segment_ptr_kernel, dim3((unsigned)blocks), dim3(256), 0, gn_stream(stream), batch, n, B, ptr);
    GN_LAUNCH_CHECK("gn_segment_ptr");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ FPS
// One workgroup of THREADS (256 / 512 / 1024, see gn_fps_nested) per example.  Thread t owns points t, t+THREADS, ... (PPT of them) and keeps
// their coordinates and running min-distance in registers; a SoA copy of the positions lives in LDS so that every
// thread can fetch the newly selected point by broadcast read.  Per step: fused (min-update, local arg-max of the SLOT),
// wave arg-max on the DPP crossbar (quad_perm / row_half_mirror / row_mirror + 4 readlanes: max of the distance; the index is read
// from the one lane a ballot names, or -- exact ties -- the DPP min of the indices: ties go to the lowest index), one LDS exchange
// of the wave partials (double-buffered by step parity -> a single barrier per step), and a log2(waves)-step DPP reduce of the partials.
#define FPS_THREADS 1024
#define FPS_WAVES (FPS_THREADS / 64)

template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
#define DPP_QUAD_XOR1 0xB1        // quad_perm [1,0,3,2]
#define DPP_QUAD_XOR2 0x4E        // quad_perm [2,3,0,1]
#define DPP_ROW_HALF_MIRROR 0x141
#define DPP_ROW_MIRROR 0x140

// after these four steps every 16-lane row holds its own reduction in all of its lanes
__device__ __forceinline__ float row_max_f(float v) {
    v = fmaxf(v, __int_as_float(dpp_i<DPP_QUAD_XOR1>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<DPP_QUAD_XOR2>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<DPP_ROW_HALF_MIRROR>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<DPP_ROW_MIRROR>(__float_as_int(v))));
    return v;
}
__device__ __forceinline__ int row_min_i(int v) {
    v = min(v, dpp_i<DPP_QUAD_XOR1>(v));
    v = min(v, dpp_i<DPP_QUAD_XOR2>(v));
    v = min(v, dpp_i<DPP_ROW_HALF_MIRROR>(v));
    v = min(v, dpp_i<DPP_ROW_MIRROR>(v));
    return v;
}
// the same over N <= 16 values replicated with period N along the row: log2(N) steps
template <int N>
__device__ __forceinline__ float part_max_f(float v) {
    v = fmaxf(v, __int_as_float(dpp_i<DPP_QUAD_XOR1>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(dpp_i<DPP_QUAD_XOR2>(__float_as_int(v))));
    if (N > 4) v = fmaxf(v, __int_as_float(dpp_i<DPP_ROW_HALF_MIRROR>(__float_as_int(v))));
    if (N > 8) v = fmaxf(v, __int_as_float(dpp_i<DPP_ROW_MIRROR>(__float_as_int(v))));
    return v;
}
template <int N>
__device__ __forceinline__ int part_min_i(int v) {
    v = min(v, dpp_i<DPP_QUAD_XOR1>(v));
    v = min(v, dpp_i<DPP_QUAD_XOR2>(v));
    if (N > 4) v = min(v, dpp_i<DPP_ROW_HALF_MIRROR>(v));
    if (N > 8) v = min(v, dpp_i<DPP_ROW_MIRROR>(v));
    return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
    v = row_max_f(v);
    const int iv = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 0)), __int_as_float(__builtin_amdgcn_readlane(iv, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(iv, 32)), __int_as_float(__builtin_amdgcn_readlane(iv, 48))));
}
__device__ __forceinline__ int wave_min_i(int v) {
    v = row_min_i(v);
    return min(min(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
               min(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// One wave per SIMD for as long as the points fit the registers (24 per lane): a step costs what its instructions cost in issue slots -- the
// min-update is the same work however it is spread, but the arg-max reduction and the cross-wave exchange are paid once per wave.
// The update itself runs on point PAIRS in packed fp32 (v_pk_add_f32 / v_pk_mul_f32: the same IEEE operations, half the issue slots).
typedef float fps_f2 __attribute__((ext_vector_type(2)));
// running minimum in ONE instruction: fminf() compiles to two canonicalising v_max_f32 + v_min_f32 under the IEEE mode bit (3 instructions per point in a
// loop of ~8); v_min_f32 itself returns the non-NaN operand, i.e. what `if (d < dist) dist = d` leaves (oracle/gn_oracle.c gno_fps)
__device__ __forceinline__ float fps_min(float a, float b) {
    float o;
    asm("v_min_f32 %0, %1, %2" : "=v"(o) : "v"(a), "v"(b));
    return o;
}

template <int PPT, bool LDS_POS, int THREADS>
__global__ __launch_bounds__(THREADS) void fps_kernel(const float *__restrict__ pos, const int32_t *__restrict__ ptr,
                                                      const int32_t *__restrict__ out_ptr, const int32_t *__restrict__ start_idx,
                                                      int32_t *__restrict__ out_idx, float *__restrict__ gap_out,
                                                      const float *__restrict__ nested_gap) {
    constexpr int WAVES = THREADS / 64;
    static_assert(PPT == 1 || PPT % 2 == 0, "points are updated in pairs");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const int s = ptr[b], n = ptr[b + 1] - s;
    const int o0 = out_ptr[b], m = out_ptr[b + 1] - o0;
    if (n <= 0 || m <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // nested sampling (gn_fps_nested): the example's points ARE an earlier farthest-point sample in selection order, begun at the same point, whose
    // running maximum stayed positive (nested_gap[b] > 0) -- then this sample is that one's prefix: local indices 0, 1, ..., m-1 (proof at gn_fps_nested)
    if (nested_gap && nested_gap[b] > 0.f && m <= n) {
        for (int k = tid; k < m; k += THREADS) out_idx[o0 + k] = s + k;
        if (gap_out && tid == 0) gap_out[b] = nested_gap[b];      // this sample's running maxima are the first m - 1 of the earlier one's: >= its smallest
        return;
    }
    float gmin = 3.0e38f;                   // smallest running maximum over the steps (gap_out)
    float *pv = smem;                       // [2][WAVES] partial values
    int *pi = (int *)(smem + 2 * WAVES);    // [2][WAVES] partial indices
    float *lx = smem + 4 * WAVES;           // SoA positions (LDS_POS only)
    float *ly = lx + n, *lz = ly + n;
    const float *gp = pos + 3 * (size_t)s;

    float px[PPT], py[PPT], pz[PPT], dd[PPT];
#pragma unroll
    for (int j = 0; j < PPT; ++j) {
        int i = tid + j * THREADS;
        if (i < n) {
            px[j] = gp[3 * i]; py[j] = gp[3 * i + 1]; pz[j] = gp[3 * i + 2];
            if (LDS_POS) { lx[i] = px[j]; ly[i] = py[j]; lz[i] = pz[j]; }
            dd[j] = 3.0e38f;
        } else {
            // a slot past the example's last point: finite coordinates and a running distance of -1 -- min(-1, d) stays -1 and -1 never beats
            // the arg-max's start value, so the update needs no per-slot guard (the guards cut the unrolled update into 24 exec-masked blocks)
            px[j] = py[j] = pz[j] = 0.f;
            dd[j] = -1.f;
        }
    }
    if (LDS_POS) __syncthreads();
    int last = start_idx ? start_idx[b] : 0;   // torch_cluster random_start: the host draws the start; default first point
    if (last < 0 || last >= n) last = 0;
    if (tid == 0) out_idx[o0] = s + last;
    for (int k = 1; k < m; ++k) {
        float qx, qy, qz;
        if (LDS_POS) { qx = lx[last]; qy = ly[last]; qz = lz[last]; }
        else { qx = gp[3 * last]; qy = gp[3 * last + 1]; qz = gp[3 * last + 2]; }
        float bv = -1.f;
        int bj = 0;                                // slot of the lane's best point (point index tid + bj * THREADS)
        if (PPT == 1) {
            const float d = fps_min(dd[0], gn_sqdist3(px[0], py[0], pz[0], qx, qy, qz));
            dd[0] = d;
            bv = d > bv ? d : bv;
        } else {
            const fps_f2 q2x = {qx, qx}, q2y = {qy, qy}, q2z = {qz, qz};
#pragma unroll
            for (int j = 0; j < PPT; j += 2) {
                // d = (dx*dx + dy*dy) + dz*dz, fp32, no contraction: gn_sqdist3's operations on two points at a time
                const fps_f2 dx = (fps_f2){px[j], px[j + 1]} - q2x, dy = (fps_f2){py[j], py[j + 1]} - q2y, dz = (fps_f2){pz[j], pz[j + 1]} - q2z;
                const fps_f2 d2 = (dx * dx + dy * dy) + dz * dz;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float d = fps_min(dd[j + u], u ? d2.y : d2.x);
                    dd[j + u] = d;
                    if (d > bv) { bv = d; bj = j + u; }  // ascending slot = ascending point index: ties keep the lowest index
                }
            }
        }
        const int bi = bv >= 0.f ? tid + bj * THREADS : INT_MAX;   // (a lane whose slots are all past the end holds -1)
        const float wv = wave_max_f(bv);
        // the lanes that hold the wave's maximum: one lane in all but exact ties -> its index is read directly; ties take the DPP min
        const unsigned long long hit = __ballot(bv == wv);
        int wi;
        if (__builtin_popcountll(hit) == 1) wi = __builtin_amdgcn_readlane(bi, __builtin_ctzll(hit));
        else wi = wave_min_i(bv == wv ? bi : INT_MAX);
        const int par = (k & 1) * WAVES;
        if (lane == 0) { pv[par + wave] = wv; pi[par + wave] = wi; }
        __syncthreads();
        // lanes 0..WAVES-1 (replicated over the 16-lane DPP row) of every wave reduce the partials
        float fv = pv[par + (lane & (WAVES - 1))];
        int fi = pi[par + (lane & (WAVES - 1))];
        const float gv = part_max_f<WAVES>(fv);
        fi = part_min_i<WAVES>(fv == gv ? fi : INT_MAX);
        last = __builtin_amdgcn_readfirstlane(fi);
        gmin = fps_min(gmin, gv);
        if (tid == 0) out_idx[o0 + k] = s + last;
    }
    if (gap_out && tid == 0) gap_out[b] = gmin;
}

// Large examples (more points than the register-resident kernel's 16 per lane x 1024 lanes): the running distances live in LDS (one float per point, up to
// FPS_LDS_MAX points = 144 KB), the coordinates are re-read from global memory every step (L2-resident).  Same arithmetic, same tie rule (a thread walks its
// points in ascending index and keeps the first maximum; wave and workgroup reductions as above), so the index lists are the oracle's for any size; the price is
// ~n / 1024 dependent LDS round trips per step instead of a register update -- a correctness path for clouds the reference's fps (no limit,
// components/pointnet2.py:26) would take, not a tuned one.
#define FPS_LDS_MAX (36 * 1024)
__global__ __launch_bounds__(FPS_THREADS) void fps_lds_kernel(const float *__restrict__ pos, const int32_t *__restrict__ ptr, const int32_t *__restrict__ out_ptr,
                                                             const int32_t *__restrict__ start_idx, int32_t *__restrict__ out_idx, float *__restrict__ gap_out,
                                                             const float *__restrict__ nested_gap) {
    constexpr int WAVES = FPS_WAVES;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int b = blockIdx.x;
    const int s = ptr[b], n = ptr[b + 1] - s;
    const int o0 = out_ptr[b], m = out_ptr[b + 1] - o0;
    if (n <= 0 || m <= 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (nested_gap && nested_gap[b] > 0.f && m <= n) {
        for (int k = tid; k < m; k += FPS_THREADS) out_idx[o0 + k] = s + k;
        if (gap_out && tid == 0) gap_out[b] = nested_gap[b];
        return;
    }
    float *pv = smem;
    int *pi = (int *)(smem + 2 * WAVES);
    float *dd = smem + 4 * WAVES;
    const float *gp = pos + 3 * (size_t)s;
    for (int i = tid; i < n; i += FPS_THREADS) dd[i] = 3.0e38f;
    __syncthreads();
    float gmin = 3.0e38f;
    int last = start_idx ? start_idx[b] : 0;
    if (last < 0 || last >= n) last = 0;
    if (tid == 0) out_idx[o0] = s + last;
    for (int k = 1; k < m; ++k) {
        const float qx = gp[3 * last], qy = gp[3 * last + 1], qz = gp[3 * last + 2];
        float bv = -1.f;
        int bi = INT_MAX;
        for (int i = tid; i < n; i += FPS_THREADS) {
            const float d = fps_min(dd[i], gn_sqdist3(gp[3 * i], gp[3 * i + 1], gp[3 * i + 2], qx, qy, qz));
            dd[i] = d;
            if (d > bv) { bv = d; bi = i; }          // ascending index: ties keep the lowest
        }
        const float wv = wave_max_f(bv);
        const int wi = wave_min_i(bv == wv ? bi : INT_MAX);
        const int par = (k & 1) * WAVES;
        if (lane == 0) { pv[par + wave] = wv; pi[par + wave] = wi; }
        __syncthreads();
        float fv = pv[par + (lane & (WAVES - 1))];
        int fi = pi[par + (lane & (WAVES - 1))];
        const float gv = part_max_f<WAVES>(fv);
        fi = part_min_i<WAVES>(fv == gv ? fi : INT_MAX);
        last = __builtin_amdgcn_readfirstlane(fi);
        gmin = fps_min(gmin, gv);
        if (tid == 0) out_idx[o0 + k] = s + last;
    }
    if (gap_out && tid == 0) gap_out[b] = gmin;
}

extern "C" int gn_fps_nested(const float *pos, const int32_t *ptr, const int32_t *out_ptr, const int32_t *start_idx, int B,
                             int max_points_per_example, int32_t *out_idx, float *gap_out, const float *nested_gap, void *stream) {
    GN_REQUIRE(B >= 0 && max_points_per_example >= 0, "gn_fps: bad sizes");
    GN_REQUIRE(nested_gap == nullptr || start_idx == nullptr, "gn_fps_nested: a nested sample starts at the example's first point (start_idx must be NULL)");
    if (B == 0 || max_points_per_example == 0) return GN_OK;
    const int n = max_points_per_example;
    GN_REQUIRE(n <= FPS_LDS_MAX, "gn_fps: more than %d points per example is not supported (got %d)", FPS_LDS_MAX, n);
    if (n > 16 * FPS_THREADS) {                     // beyond the register-resident kernels: running distances in LDS (fps_lds_kernel)
        const size_t shl = sizeof(float) * (4 * FPS_WAVES + (size_t)n);
        GN_HIP(hipFuncSetAttribute((const void *)fps_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shl), "gn_fps");
        hipLaunchKernelGGL(fps_lds_kernel, dim3(B), dim3(FPS_THREADS), shl, gn_stream(stream), pos, ptr, out_ptr, start_idx, out_idx, gap_out, nested_gap);
        GN_LAUNCH_CHECK("gn_fps");
        return GN_OK;
    }
    const bool lds_pos = n <= 8192;
    // ONE wave per SIMD for as long as its points fit the registers (24 per lane): the min-update costs the same issue slots however it is spread, the
    // arg-max reduction and the exchange are paid per wave (n = 6000: 256 threads 2.13 ms, 512 threads 2.26 ms, 1024 threads 3.0 ms)
#ifndef FPS_MID_THREADS
#define FPS_MID_THREADS 512
#endif
    const int threads = n <= 24 * 256 ? 256 : (n <= 24 * FPS_MID_THREADS ? FPS_MID_THREADS : FPS_THREADS);
    size_t sh = sizeof(float) * 4 * (threads / 64) + (lds_pos ? sizeof(float) * 3 * (size_t)n : 0);
    const int ppt = (int)gn_cdiv(n, threads);
#define FPS_LAUNCH(P, L, T)                                                                                     \
    do {                                                                                                        \
        GN_HIP(hipFuncSetAttribute((const void *)fps_kernel<P, L, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh), "gn_fps"); \
        hipLaunchKernelGGL((fps_kernel<P, L, T>), dim3(B), dim3(T), sh, gn_stream(stream), pos, ptr, out_ptr, start_idx, out_idx, gap_out, nested_gap); \
    } while (0)
#define FPS_BY_PPT(L, T)                               \
    do {                                               \
        if (ppt <= 2) FPS_LAUNCH(2, L, T);             \
        else if (ppt <= 4) FPS_LAUNCH(4, L, T);        \
        else if (ppt <= 6) FPS_LAUNCH(6, L, T);        \
        else if (ppt <= 8) FPS_LAUNCH(8, L, T);        \
        else if (ppt <= 12) FPS_LAUNCH(12, L, T);      \
        else if (ppt <= 16) FPS_LAUNCH(16, L, T);      \
        else FPS_LAUNCH(24, L, T);                     \
    } while (0)
    if (threads == 256) FPS_BY_PPT(true, 256);
    else if (threads == FPS_MID_THREADS && lds_pos) FPS_BY_PPT(true, FPS_MID_THREADS);
    else if (threads == FPS_MID_THREADS) FPS_BY_PPT(false, FPS_MID_THREADS);
    else FPS_LAUNCH(16, false, FPS_THREADS);
#undef FPS_BY_PPT
#undef FPS_LAUNCH
    GN_LAUNCH_CHECK("gn_fps");
    return GN_OK;
}

extern "C" int gn_fps(const float *pos, const int32_t *ptr, const int32_t *out_ptr, const int32_t *start_idx, int B,
                      int max_points_per_example, int32_t *out_idx, void *stream) {
    return gn_fps_nested(pos, ptr, out_ptr, start_idx, B, max_points_per_example, out_idx, nullptr, nullptr, stream);
}

// ------------------------------------------------------------------------------------------------ ball query
// One wavefront per centre; the example's points are scanned 64 at a time in ascending index, the ballot of the
// in-range predicate and a popcount prefix give each hit its ordered output slot; stop at K.
__global__ __launch_bounds__(256) void ball_query_kernel(const float *__restrict__ pos, const int32_t *__restrict__ ptr,
                                                         const int32_t *__restrict__ centre_idx,
                                                         const int32_t *__restrict__ centre_ptr, int B, int M, float r2, int K,
                                                         int32_t *__restrict__ nbr, int32_t *__restrict__ cnt) {
    const int lane = threadIdx.x & 63;
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= M) return;
    // example of this centre: binary search in centre_ptr
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (centre_ptr[mid] <= c) lo = mid; else hi = mid;
    }
    const int b = lo;
    const int s = ptr[b], e = ptr[b + 1];
    const int ci = centre_idx[c];
    const float cx = pos[3 * (size_t)ci], cy = pos[3 * (size_t)ci + 1], cz = pos[3 * (size_t)ci + 2];
    int32_t *row = nbr + (size_t)c * K;
    int count = 0;
    for (int base = s; base < e && count < K; base += 64) {
        int j = base + lane;
        bool in = false;
        if (j < e) {
            float d = gn_sqdist3(pos[3 * (size_t)j], pos[3 * (size_t)j + 1], pos[3 * (size_t)j + 2], cx, cy, cz);
            in = d < r2;
        }
        unsigned long long mask = __ballot(in);
        int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (in && count + before < K) row[count + before] = j;
        count += __popcll(mask);
    }
    if (count > K) count = K;
    for (int t = count + lane; t < K; t += 64) row[t] = -1;
    if (lane == 0) cnt[c] = count;
}

extern "C" int gn_ball_query(const float *pos, const int32_t *ptr, const int32_t *centre_idx, const int32_t *centre_ptr,
                             int B, int M, float r2, int K, int32_t *nbr, int32_t *cnt, void *stream) {
    GN_REQUIRE(B >= 0 && M >= 0 && K > 0, "gn_ball_query: bad sizes");
    if (M == 0) return GN_OK;
    hipLaunchKernelGGL(ball_query_kernel, dim3((unsigned)gn_cdiv(M, 4)), dim3(256), 0, gn_stream(stream), pos, ptr, centre_idx,
                       centre_ptr, B, M, r2, K, nbr, cnt);
    GN_LAUNCH_CHECK("gn_ball_query");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ SA gather
// one wave per (centre, slot) row
__global__ __launch_bounds__(256) void sa_gather_kernel(const float *__restrict__ x, int ldx, int C, const float *__restrict__ pos,
                                                        const int32_t *__restrict__ centre_idx, const int32_t *__restrict__ nbr,
                                                        int M, int K, int self_loops, const int32_t *__restrict__ self_src,
                                                        float *__restrict__ out, int ldo, int32_t *__restrict__ slot_src) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int S = K + (self_loops ? 1 : 0);
    if (row >= (int64_t)M * S) return;
    const int c = (int)(row / S), sl = (int)(row % S);
    const int self = self_src ? self_src[c] : c;  // the point that plays "node c" on the source side (NULL: PyG's literal rule)
    int j;
    if (sl < K) {
        j = nbr[(size_t)c * K + sl];
        if (self_loops && j == self) j = -1;  // remove_self_loops on numeric equality source == target
    } else {
        j = self;  // add_self_loops(num_nodes = M): source = point c of the full cloud
    }
    float *o = out + row * (int64_t)ldo;
    if (lane == 0) slot_src[row] = j;
    if (j < 0) {
        for (int t = lane; t < C + 3; t += 64) o[t] = 0.f;
        return;
    }
    const float *xr = x ? x + (size_t)j * ldx : nullptr;
    for (int t = lane; t < C; t += 64) o[t] = xr[t];
    if (lane < 3) {
        int ci = centre_idx[c];
        o[C + lane] = __fsub_rn(pos[3 * (size_t)j + lane], pos[3 * (size_t)ci + lane]);
    }
}

extern "C" int gn_sa_gather_scoped(const float *x, int ldx, int C, const float *pos, const int32_t *centre_idx, const int32_t *nbr,
                                   int M, int K, int self_loops, const int32_t *self_src, float *out, int ldo, int32_t *slot_src,
                                   void *stream) {
    GN_REQUIRE(M >= 0 && K > 0 && C >= 0 && ldo >= C + 3, "gn_sa_gather: bad sizes");
    GN_REQUIRE(C == 0 || x != nullptr, "gn_sa_gather: x is NULL with C>0");
    if (M == 0) return GN_OK;
    int64_t rows = (int64_t)M * (K + (self_loops ? 1 : 0));
    hipLaunchKernelGGL(sa_gather_kernel, dim3((unsigned)gn_cdiv(rows, 4)), dim3(256), 0, gn_stream(stream), x, ldx, C, pos,
                       centre_idx, nbr, M, K, self_loops, self_src, out, ldo, slot_src);
    GN_LAUNCH_CHECK("gn_sa_gather");
    return GN_OK;
}

extern "C" int gn_sa_gather(const float *x, int ldx, int C, const float *pos, const int32_t *centre_idx, const int32_t *nbr,
                            int M, int K, int self_loops, float *out, int ldo, int32_t *slot_src, void *stream) {
    return gn_sa_gather_scoped(x, ldx, C, pos, centre_idx, nbr, M, K, self_loops, nullptr, out, ldo, slot_src, stream);
}

// ------------------------------------------------------------------------------------------------ segment max
__global__ __launch_bounds__(256) void segment_max_kernel(const float *__restrict__ in, int ldi, const int32_t *__restrict__ slot_src,
                                                          int M, int S, int C, float *__restrict__ out, int ldo) {
    const int c = blockIdx.x;
    for (int ch = threadIdx.x; ch < C; ch += blockDim.x) {
        float v = 0.f;
        bool any = false;
        for (int s = 0; s < S; ++s) {
            int64_t row = (int64_t)c * S + s;
            if (slot_src[row] < 0) continue;
            float t = in[row * ldi + ch];
            v = any ? fmaxf(v, t) : t;
            any = true;
        }
        out[(int64_t)c * ldo + ch] = v;  // no valid slot -> 0 (scatter-max of nothing)
    }
}

extern "C" int gn_segment_max(const float *in, int ldi, const int32_t *slot_src, int M, int S, int C, float *out, int ldo,
                              void *stream) {
    GN_REQUIRE(M >= 0 && S > 0 && C > 0, "gn_segment_max: bad sizes");
    if (M == 0) return GN_OK;
    int threads = C >= 256 ? 256 : (C >= 128 ? 128 : 64);
    hipLaunchKernelGGL(segment_max_kernel, dim3(M), dim3(threads), 0, gn_stream(stream), in, ldi, slot_src, M, S, C, out, ldo);
    GN_LAUNCH_CHECK("gn_segment_max");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ global max pool
// grid (B, channel tiles of 64); each block: 1024 threads = 16 row-groups x 64 channels, LDS combine (round 6: 4 -> 16 row groups -- a batch of one is 16
// workgroups, and 750 rows in 4 groups were 188 dependent steps per thread: 37 us; a maximum does not depend on the order it is taken in).
#define GMAX_GROUPS 16
__global__ __launch_bounds__(64 * GMAX_GROUPS) void global_max_kernel(const float *__restrict__ in, int ldi, const int32_t *__restrict__ ptr,
                                                                     int C, float *__restrict__ out, int ldo) {
    __shared__ float part[GMAX_GROUPS][64];
    const int b = blockIdx.x, ch = blockIdx.y * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int s = ptr[b], e = ptr[b + 1];
    float v = -3.4e38f;
    if (ch < C)
        for (int r = s + g; r < e; r += GMAX_GROUPS) v = fmaxf(v, in[(int64_t)r * ldi + ch]);
    part[g][threadIdx.x & 63] = v;
    __syncthreads();
    if (g == 0 && ch < C) {
#pragma unroll
        for (int k = 1; k < GMAX_GROUPS; ++k) v = fmaxf(v, part[k][threadIdx.x]);
        out[(int64_t)b * ldo + ch] = (e > s) ? v : 0.f;
    }
}

extern "C" int gn_global_max_pool(const float *in, int ldi, const int32_t *ptr, int B, int C, float *out, int ldo, void *stream) {
    GN_REQUIRE(B >= 0 && C > 0, "gn_global_max_pool: bad sizes");
    if (B == 0) return GN_OK;
    hipLaunchKernelGGL(global_max_kernel, dim3(B, (unsigned)gn_cdiv(C, 64)), dim3(64 * GMAX_GROUPS), 0, gn_stream(stream), in, ldi, ptr, C, out, ldo);
    GN_LAUNCH_CHECK("gn_global_max_pool");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ kNN interpolate
// One wavefront per query.  Each lane keeps the k best (d2, idx) of the sources it scanned (ascending, ties ->
// lower index), then k rounds of wave arg-min pop the global k best in ascending order; the weighted feature sum
// is then computed with channels spread over lanes, accumulating in that order.
#define KNN_MAXK 8
template <int KK>
__global__ __launch_bounds__(256) void knn_interp_kernel(const float *__restrict__ xs, int ldx, const float *__restrict__ ps,
                                                         const int32_t *__restrict__ ptr_s, const float *__restrict__ pq,
                                                         const int32_t *__restrict__ ptr_q, int B, int Nq, int C,
                                                         float *__restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (q >= Nq) return;
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (ptr_q[mid] <= q) lo = mid; else hi = mid;
    }
    const int s = ptr_s[lo], e = ptr_s[lo + 1];
    const float qx = pq[3 * (size_t)q], qy = pq[3 * (size_t)q + 1], qz = pq[3 * (size_t)q + 2];
    float bd[KK];
    int bi[KK];
#pragma unroll
    for (int t = 0; t < KK; ++t) { bd[t] = 3.4e38f; bi[t] = INT_MAX; }
    for (int j = s + lane; j < e; j += 64) {
        float d = gn_sqdist3(ps[3 * (size_t)j], ps[3 * (size_t)j + 1], ps[3 * (size_t)j + 2], qx, qy, qz);
        // insert (d, j) into the ascending list; j ascends within a lane so strict '<' keeps lower indices first
        if (d < bd[KK - 1]) {
            bd[KK - 1] = d; bi[KK - 1] = j;
#pragma unroll
            for (int t = KK - 1; t > 0; --t) {
                if (bd[t] < bd[t - 1]) {
                    float td = bd[t]; bd[t] = bd[t - 1]; bd[t - 1] = td;
                    int ti = bi[t]; bi[t] = bi[t - 1]; bi[t - 1] = ti;
                }
            }
        }
    }
    float wd[KK];
    int wi[KK];
    int nsel = 0;
#pragma unroll
    for (int r = 0; r < KK; ++r) {
        float v = bd[0];
        int i = bi[0];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            float ov = __shfl_xor(v, off);
            int oi = __shfl_xor(i, off);
            if (ov < v || (ov == v && oi < i)) { v = ov; i = oi; }
        }
        wd[r] = v; wi[r] = i;
        if (i != INT_MAX) nsel = r + 1;
        if (bi[0] == i && i != INT_MAX) {  // the owner pops its head
#pragma unroll
            for (int t = 0; t < KK - 1; ++t) { bd[t] = bd[t + 1]; bi[t] = bi[t + 1]; }
            bd[KK - 1] = 3.4e38f; bi[KK - 1] = INT_MAX;
        }
    }
    float w[KK];
    float wsum = 0.f;
#pragma unroll
    for (int r = 0; r < KK; ++r) {
        if (r < nsel) {
            float d = wd[r] < 1e-16f ? 1e-16f : wd[r];
            w[r] = __fdiv_rn(1.0f, d);
            wsum = __fadd_rn(wsum, w[r]);
        } else w[r] = 0.f;
    }
    for (int ch = lane; ch < C; ch += 64) {
        float acc = 0.f;
#pragma unroll
        for (int r = 0; r < KK; ++r)
            if (r < nsel) acc = __fadd_rn(acc, __fmul_rn(xs[(size_t)wi[r] * ldx + ch], w[r]));
        out[(size_t)q * ldo + ch] = __fdiv_rn(acc, wsum);
    }
}

extern "C" int gn_knn_interpolate(const float *xs, int ldx, const float *ps, const int32_t *ptr_s, const float *pq,
                                  const int32_t *ptr_q, int B, int Nq, int C, int k, float *out, int ldo, void *stream) {
    GN_REQUIRE(k >= 1 && k <= KNN_MAXK, "gn_knn_interpolate: k must be in [1,%d]", KNN_MAXK);
    GN_REQUIRE(B >= 0 && Nq >= 0 && C > 0, "gn_knn_interpolate: bad sizes");
    if (Nq == 0) return GN_OK;
    dim3 grid((unsigned)gn_cdiv(Nq, 4)), block(256);
#define KNN_LAUNCH(KK) hipLaunchKernelGGL((knn_interp_kernel<KK>), grid, block, 0, gn_stream(stream), xs, ldx, ps, ptr_s, pq, ptr_q, B, Nq, C, out, ldo)
    switch (k) {
        case 1: KNN_LAUNCH(1); break;
        case 2: KNN_LAUNCH(2); break;
        case 3: KNN_LAUNCH(3); break;
        case 4: KNN_LAUNCH(4); break;
        case 5: KNN_LAUNCH(5); break;
        case 6: KNN_LAUNCH(6); break;
        case 7: KNN_LAUNCH(7); break;
        default: KNN_LAUNCH(8); break;
    }
#undef KNN_LAUNCH
    GN_LAUNCH_CHECK("gn_knn_interpolate");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ NOCS head
// one thread per (point, axis): arg-max over the bins (first maximum), soft-max value at it, bin -> coordinate.
__global__ __launch_bounds__(256) void nocs_head_kernel(const float *__restrict__ logits, int ldl, int64_t N, int bins,
                                                        int64_t *__restrict__ bin_idx, float *__restrict__ conf,
                                                        float *__restrict__ nocs) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * 3) return;
    int64_t p = t / 3;
    int a = (int)(t % 3);
    const float *row = logits + p * ldl;
    float mx = row[a];
    int arg = 0;
    for (int k = 1; k < bins; ++k) {
        float v = row[k * 3 + a];
        if (v > mx) { mx = v; arg = k; }
    }
    float sum = 0.f;
    for (int k = 0; k < bins; ++k) sum = __fadd_rn(sum, expf(__fsub_rn(row[k * 3 + a], mx)));
    bin_idx[t] = arg;
    conf[t] = __fdiv_rn(1.0f, sum);  // exp(mx-mx)/sum
    const float scale = __fdiv_rn(1.0f, __fsub_rn((float)bins, 1.0f));  // (uc-lc)/(bins-1) in fp32, gridding.py:252
    nocs[t] = __fadd_rn(__fmul_rn((float)arg, scale), 0.0f);
}

extern "C" int gn_nocs_head(const float *logits, int ldl, int64_t N, int bins, int64_t *bin_idx, float *confidence,
                            float *pred_nocs, void *stream) {
    GN_REQUIRE(N >= 0 && bins > 0 && ldl >= bins * 3, "gn_nocs_head: bad sizes");
    if (N == 0) return GN_OK;
    hipLaunchKernelGGL(nocs_head_kernel, dim3((unsigned)gn_cdiv(N * 3, 256)), dim3(256), 0, gn_stream(stream), logits, ldl, N, bins,
                       bin_idx, confidence, pred_nocs);
    GN_LAUNCH_CHECK("gn_nocs_head");
    return GN_OK;
}

// ------------------------------------------------------------------------------------------------ nearest neighbour
// Brute-force 1-NN (the cKDTree queries of the Chamfer metrics, /root/reference/eval.py:259-263,381-385).
// Thread = one query; the reference set streams through LDS in tiles of 1024 points (SoA, broadcast reads).
// d2 = (dx*dx+dy*dy)+dz*dz in fp32, ties -> lowest index.
#define NN_TILE 1024
__global__ __launch_bounds__(256) void nearest_neighbor_kernel(const float *__restrict__ q, int64_t nq, const float *__restrict__ ref,
                                                               int64_t nr, int32_t *__restrict__ idx, float *__restrict__ d2) {
    __shared__ float sx[NN_TILE], sy[NN_TILE], sz[NN_TILE];
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    float qx = 0.f, qy = 0.f, qz = 0.f;
    if (i < nq) { qx = q[3 * i]; qy = q[3 * i + 1]; qz = q[3 * i + 2]; }
    float best = 3.4e38f;
    int bi = -1;
    for (int64_t t0 = 0; t0 < nr; t0 += NN_TILE) {
        const int n = (int)((nr - t0) < NN_TILE ? (nr - t0) : NN_TILE);
        __syncthreads();
        for (int j = threadIdx.x; j < n; j += blockDim.x) { sx[j] = ref[3 * (t0 + j)]; sy[j] = ref[3 * (t0 + j) + 1]; sz[j] = ref[3 * (t0 + j) + 2]; }
        __syncthreads();
        for (int j = 0; j < n; ++j) {
            const float d = gn_sqdist3(sx[j], sy[j], sz[j], qx, qy, qz);
            if (d < best) { best = d; bi = (int)(t0 + j); }
        }
    }
    if (i < nq) { idx[i] = bi; d2[i] = best; }
}

extern "C" int gn_nearest_neighbor(const float *query, int64_t nq, const float *ref, int64_t nr, int32_t *idx, float *d2, void *stream) {
    GN_REQUIRE(nq >= 0 && nr > 0 && nr < INT32_MAX, "gn_nearest_neighbor: bad sizes (the reference set must not be empty)");
    if (nq == 0) return GN_OK;
    hipLaunchKernelGGL(nearest_neighbor_kernel, dim3((unsigned)gn_cdiv(nq, 256)), dim3(256), 0, gn_stream(stream), query, nq, ref, nr, idx, d2);
    GN_LAUNCH_CHECK("gn_nearest_neighbor");
    return GN_OK;
}
