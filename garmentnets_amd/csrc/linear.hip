// linear.hip -- dense layer Y = bn(act(X W^T + b)) on the fp32 matrix cores of gfx950.
//
// Replaces torch.nn.Linear -> ReLU -> BatchNorm1d(eval) (components/mlp.py:9-20), the PointNet++ heads
// (networks/pointnet2_nocs.py:145-157) and the 1x1x1 final_conv of the UNet (components/unet3d.py:437,467).
//
// v_mfma_f32_32x32x2_f32: one wave computes D(32x32) += A(32x2) B(2x32); operand A: lane l holds A[l&31][l>>5],
// operand B: lane l holds B[l>>5][l&31]; D: lane l, reg r -> column l&31, row (r&3)+8*(r>>2)+4*(l>>5).
// Exact fp32 (k-ordered fma chain), 64 FLOP/clk/SIMD = the fp32 peak of the chip (157.3 TFLOP/s).
//
// Block = 256 threads = 4 waves.  X and W tiles ([rows][16 k], row stride 17 words -> conflict-free
// ds_read_b32 for 32 consecutive rows) are staged through LDS with a register prefetch of the next k-tile.
// Round 5 (tools/dev/ab_linear.py): half-height / quarter tiles when the full 128 x 128 tile would leave CUs without a workgroup
// (12000 x 256: 188 workgroups; 6000 x 128: 47) -- the same k order per output element, bit-identical results; measured and not
// shipped: 32-deep tiles (LIN_BK 32: no gain, K = 131 pads further) and a double-buffered tile with one barrier (LIN_NBUF 2: costs a
// workgroup per CU, 20 % slower).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define LIN_BK 16
#ifndef LIN_NBUF
#define LIN_NBUF 1
#endif
#define LIN_LDS_STRIDE (LIN_BK + 1)

template <int ROWS>
struct TileRegs {
    static constexpr int NV = (ROWS * (LIN_BK / 4) + 255) / 256;  // float4 per thread
    float4 v[NV];
};

// load a [ROWS][16] tile of a row-major matrix P (ld, rows limited by nrows, cols limited by K) into registers
template <int ROWS, bool ALIGNED>
__device__ __forceinline__ void tile_load(TileRegs<ROWS> &r, const float *__restrict__ P, int ld, int64_t row0, int64_t nrows,
                                          int k0, int K) {
#pragma unroll
    for (int i = 0; i < TileRegs<ROWS>::NV; ++i) {
        int idx = threadIdx.x + i * 256;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < ROWS * (LIN_BK / 4)) {
            int64_t row = row0 + idx / (LIN_BK / 4);
            int k = k0 + (idx % (LIN_BK / 4)) * 4;
            if (row < nrows && k < K) {
                const float *p = P + row * ld + k;
                if (ALIGNED) {
                    v = *reinterpret_cast<const float4 *>(p);
                    if (k + 1 >= K) v.y = 0.f;
                    if (k + 2 >= K) v.z = 0.f;
                    if (k + 3 >= K) v.w = 0.f;
                } else {
                    v.x = p[0];
                    if (k + 1 < K) v.y = p[1];
                    if (k + 2 < K) v.z = p[2];
                    if (k + 3 < K) v.w = p[3];
                }
            }
        }
        r.v[i] = v;
    }
}

template <int ROWS>
__device__ __forceinline__ void tile_store(const TileRegs<ROWS> &r, float *__restrict__ lds) {
#pragma unroll
    for (int i = 0; i < TileRegs<ROWS>::NV; ++i) {
        int idx = threadIdx.x + i * 256;
        if (idx < ROWS * (LIN_BK / 4)) {
            float *p = lds + (idx / (LIN_BK / 4)) * LIN_LDS_STRIDE + (idx % (LIN_BK / 4)) * 4;
            p[0] = r.v[i].x; p[1] = r.v[i].y; p[2] = r.v[i].z; p[3] = r.v[i].w;
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN, bool ALIGNED>
__global__ __launch_bounds__(256) void linear_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W, int ldw,
                                                     const float *__restrict__ bias, const float *__restrict__ bn_scale,
                                                     const float *__restrict__ bn_shift, int relu, int64_t M, int N, int K,
                                                     float *__restrict__ Y, int ldy) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per block");
    constexpr int BM = WAVES_M * TM * 32, BN = WAVES_N * TN * 32;
    __shared__ float As[LIN_NBUF][BM * LIN_LDS_STRIDE];
    __shared__ float Bs[LIN_NBUF][BN * LIN_LDS_STRIDE];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int64_t m0 = (int64_t)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    TileRegs<BM> ra;
    TileRegs<BN> rb;
    tile_load<BM, ALIGNED>(ra, X, ldx, m0, M, 0, K);
    tile_load<BN, ALIGNED>(rb, W, ldw, n0, N, 0, K);
    tile_store<BM>(ra, As[0]);
    tile_store<BN>(rb, Bs[0]);
    __syncthreads();
    const int nk = (K + LIN_BK - 1) / LIN_BK;
    const int arow = (wm * TM * 32 + (lane & 31)) * LIN_LDS_STRIDE + (lane >> 5);
    const int brow = (wn * TN * 32 + (lane & 31)) * LIN_LDS_STRIDE + (lane >> 5);
    for (int kt = 0; kt < nk; ++kt) {
        const float *const Ac = As[kt & (LIN_NBUF - 1)], *const Bc = Bs[kt & (LIN_NBUF - 1)];
        if (kt + 1 < nk) {
            tile_load<BM, ALIGNED>(ra, X, ldx, m0, M, (kt + 1) * LIN_BK, K);
            tile_load<BN, ALIGNED>(rb, W, ldw, n0, N, (kt + 1) * LIN_BK, K);
        }
#pragma unroll
        for (int kk = 0; kk < LIN_BK / 2; ++kk) {
            float a[TM], b[TN];
#pragma unroll
            for (int t = 0; t < TM; ++t) a[t] = Ac[arow + t * 32 * LIN_LDS_STRIDE + kk * 2];
#pragma unroll
            for (int u = 0; u < TN; ++u) b[u] = Bc[brow + u * 32 * LIN_LDS_STRIDE + kk * 2];
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[u], acc[t][u], 0, 0, 0);
        }
        if (LIN_NBUF == 1) __syncthreads();
        if (kt + 1 < nk) {
            // the OTHER buffer: everybody left it at the barrier that closed tile kt - 1
            tile_store<BM>(ra, As[(kt + 1) & (LIN_NBUF - 1)]);
            tile_store<BN>(rb, Bs[(kt + 1) & (LIN_NBUF - 1)]);
        }
        __syncthreads();
    }
    // epilogue: bias -> ReLU -> BN affine
#pragma unroll
    for (int u = 0; u < TN; ++u) {
        const int n = n0 + (wn * TN + u) * 32 + (lane & 31);
        if (n >= N) continue;
        const float bv = bias ? bias[n] : 0.f;
        const float sc = bn_scale ? bn_scale[n] : 1.f;
        const float sh = bn_shift ? bn_shift[n] : 0.f;
#pragma unroll
        for (int t = 0; t < TM; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + (wm * TM + t) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < M) {
                    float v = __fadd_rn(acc[t][u][r], bv);
                    if (relu) v = gn_relu(v);
                    if (bn_scale) v = __fadd_rn(__fmul_rn(v, sc), sh);
                    Y[m * ldy + n] = v;
                }
            }
        }
    }
}

extern "C" int gn_linear(const float *X, int ldx, const float *W, int ldw, const float *bias, const float *bn_scale,
                         const float *bn_shift, int relu, int64_t M, int N, int K, float *Y, int ldy, void *stream) {
    GN_REQUIRE(M >= 0 && N > 0 && K > 0 && ldx >= K && ldw >= K && ldy >= N, "gn_linear: bad sizes M=%lld N=%d K=%d", (long long)M, N, K);
    GN_REQUIRE((bn_scale == nullptr) == (bn_shift == nullptr), "gn_linear: bn_scale and bn_shift must come together");
    if (M == 0) return GN_OK;
    const bool aligned = (ldx % 4 == 0) && (ldw % 4 == 0) && (((uintptr_t)X | (uintptr_t)W) % 16 == 0);
    hipStream_t st = gn_stream(stream);
#define LIN_LAUNCH(WM, WN, TM, TN)                                                                                          \
    do {                                                                                                                    \
        constexpr int BM = WM * TM * 32, BN = WN * TN * 32;                                                                 \
        dim3 grid((unsigned)gn_cdiv(M, BM), (unsigned)gn_cdiv(N, BN));                                                      \
        if (aligned)                                                                                                        \
            hipLaunchKernelGGL((linear_kernel<WM, WN, TM, TN, true>), grid, dim3(256), 0, st, X, ldx, W, ldw, bias, bn_scale, bn_shift, relu, M, N, K, Y, ldy); \
        else                                                                                                                \
            hipLaunchKernelGGL((linear_kernel<WM, WN, TM, TN, false>), grid, dim3(256), 0, st, X, ldx, W, ldw, bias, bn_scale, bn_shift, relu, M, N, K, Y, ldy); \
    } while (0)
    const int64_t full = gn_cdiv(M, 128) * gn_cdiv(N, 128);   // workgroups of the 128 x 128 tile
    if (N <= 32) LIN_LAUNCH(4, 1, 2, 1);
    else if (N <= 64) LIN_LAUNCH(2, 2, 2, 1);
    else if (full >= 384) LIN_LAUNCH(2, 2, 2, 2);
    else if (2 * full >= 256) LIN_LAUNCH(2, 2, 1, 2);        // 64 x 128
    else LIN_LAUNCH(2, 2, 1, 1);                             // 64 x 64
#undef LIN_LAUNCH
    GN_LAUNCH_CHECK("gn_linear");
    return GN_OK;
}
