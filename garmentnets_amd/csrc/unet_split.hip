// unet_split.hip -- OPT-IN split-precision variant of the 'gcr' conv (gn_conv3d_gcr_split), NOT the default path.
//
// Same implicit-GEMM structure, tiling, GroupNorm-on-load, upsample/concat folding and epilogue as conv3d_gcr_kernel
// (unet.hip), but every fp32 operand is decomposed EXACTLY into P bf16 planes (x = x1 + x2 [+ x3], xi = bf16_rn of the
// running residual) and the products are formed on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16: exact bf16 x bf16
// products, fp32 accumulation, 16x the k-throughput of v_mfma_f32_32x32x2_f32):
//   P = 3 : 6 products x1w1 + x1w2 + x2w1 + x1w3 + x2w2 + x3w1   -> dropped terms <= 2^-24 relative: fp32-class products
//   P = 2 : 3 products x1w1 + x1w2 + x2w1                       -> 2^-16 relative per product
// One bf16 MFMA consumes the 16-channel slice of a tap at once: lane (h = lane>>5, r = lane&31) supplies channels 8h..8h+7
// of voxel r (A) / of output channel r (B) -- the same fragment the fp32 kernel reads, so the LDS layout is the fp32 one
// with P bf16 planes per voxel.  The result is validated against the same oracle and goldens as the fp32 path
// (tests/test_gpu_parity.py::test_conv3d_split_*); DESIGN.md section 4 reports speed and error next to the fp32 kernel.
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

struct SplitArgs {
    const float *src0;
    const float *src1;
    const float *a;
    const float *d;
    const uint4 *wp;   // [tap][slice][cout][P][16] bf16
    float *out;
    double *osum;
    double *osq;
    int C0, C1, B, D, H, W, Cout, relu;
    int tiles_y, tiles_x;
};

__device__ __forceinline__ unsigned bf16_rn_bits(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return u >> 16;
}

// x -> P bf16 planes (exact residual chain); returns the planes' 16-bit patterns
template <int P>
__device__ __forceinline__ void split_bf16(float x, unsigned (&pl)[P]) {
    float r = x;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        pl[i] = bf16_rn_bits(r);
        r = __fsub_rn(r, __uint_as_float(pl[i] << 16));
    }
}

template <int NT, int P>
__global__ __launch_bounds__(256, 2) void conv3d_split_kernel(SplitArgs p) {
    constexpr int CT = NT * 32;
    // bytes per voxel / per weight row in LDS.  P = 2: 64 + 16 pad (16-byte slots rotate with the row, conflict-free b128);
    // P = 3: 96 unpadded (2-way conflicts) so that halo + weights stay under 80 KB and two workgroups share a CU
    constexpr int VB = (P == 3) ? 96 : P * 32 + 16;
    __shared__ __attribute__((aligned(16))) unsigned char halo[SP_HVOX * VB];
    __shared__ __attribute__((aligned(16))) unsigned char wsm[2][CT * VB];
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
    const int z0 = tz * SP_TZ, y0 = ty * SP_TY, x0 = tx * SP_TX;
    const int n0 = cb * CT;
    const int b = blockIdx.y;
    const int D1 = p.D >> 1, H1 = p.H >> 1, W1 = p.W >> 1;

    f32x16s acc[2][NT], tot[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc[t][u][q] = 0.f; tot[t][u][q] = 0.f; }

    const int abase = ((wave * SP_HY + (r >> 3)) * SP_HX + (r & 7)) * VB + 16 * h;   // bytes; plane pl at +32*pl
    constexpr int AF1 = 4 * SP_HX * VB;
    const int bbase = r * VB + 16 * h;

    constexpr int WV = CT * P * 2;                        // uint4 per weight tile
    constexpr int WPT = (WV + 255) / 256;
    const int nslices = Cin / SP_KS;
    const int64_t tap_stride = (int64_t)nslices * p.Cout * P * 2;   // uint4 units

    for (int s = 0; s < nslices; ++s) {
        const int c0 = s * SP_KS;
        {   // ---- halo stage: GroupNorm affine, then exact split into P bf16 planes
            const bool from1 = c0 >= p.C0;
            const float *src = from1 ? p.src1 : p.src0;
            const int Cs = from1 ? p.C1 : p.C0;
            const int cs = from1 ? c0 - p.C0 : c0;
            const float *ab = p.a + (int64_t)b * Cin + c0;
            const float *db = p.d + (int64_t)b * Cin + c0;
            for (int idx = tid; idx < SP_HVOX * 4; idx += 256) {
                const int hv = idx >> 2, c4 = (idx & 3) * 4;
                const int hx = hv % SP_HX, hy = (hv / SP_HX) % SP_HY, hz = hv / (SP_HX * SP_HY);
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
                float v[4] = {0.f, 0.f, 0.f, 0.f};
                if (gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
                    int64_t off;
                    if (from1) off = ((((int64_t)b * D1 + (gz >> 1)) * H1 + (gy >> 1)) * W1 + (gx >> 1)) * Cs + cs + c4;
                    else off = ((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * Cs + cs + c4;
                    const float4 xin = *reinterpret_cast<const float4 *>(src + off);
                    const float4 av = *reinterpret_cast<const float4 *>(ab + c4);
                    const float4 dv = *reinterpret_cast<const float4 *>(db + c4);
                    v[0] = __fadd_rn(__fmul_rn(xin.x, av.x), dv.x);
                    v[1] = __fadd_rn(__fmul_rn(xin.y, av.y), dv.y);
                    v[2] = __fadd_rn(__fmul_rn(xin.z, av.z), dv.z);
                    v[3] = __fadd_rn(__fmul_rn(xin.w, av.w), dv.w);
                }
                unsigned pl[4][P];
#pragma unroll
                for (int e = 0; e < 4; ++e) split_bf16<P>(v[e], pl[e]);
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    uint2 w2;
                    w2.x = pl[0][i] | (pl[1][i] << 16);
                    w2.y = pl[2][i] | (pl[3][i] << 16);
                    *reinterpret_cast<uint2 *>(halo + hv * VB + i * 32 + c4 * 2) = w2;
                }
            }
        }
        const uint4 *wslice = p.wp + ((int64_t)s * p.Cout + n0) * (P * 2);
        uint4 wreg[WPT];
#pragma unroll
        for (int i = 0; i < WPT; ++i) {
            const int idx = tid + i * 256;
            if (idx < WV) {
                wreg[i] = wslice[idx];
                *reinterpret_cast<uint4 *>(&wsm[0][(idx / (P * 2)) * VB + (idx % (P * 2)) * 16]) = wreg[i];
            }
        }
        __syncthreads();
        uint4 a0[P], a1[P], na0[P], na1[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            a0[i] = *reinterpret_cast<const uint4 *>(halo + abase + i * 32);
            a1[i] = *reinterpret_cast<const uint4 *>(halo + abase + AF1 + i * 32);
            na0[i] = a0[i]; na1[i] = a1[i];
        }
        for (int tap = 0; tap < 27; ++tap) {
            const int cur = tap & 1;
            const bool more = tap + 1 < 27;
            if (more) {
#pragma unroll
                for (int i = 0; i < WPT; ++i) {
                    const int idx = tid + i * 256;
                    if (idx < WV) wreg[i] = wslice[(int64_t)(tap + 1) * tap_stride + idx];
                }
            }
            uint4 bf[NT][P];
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int i = 0; i < P; ++i) bf[u][i] = *reinterpret_cast<const uint4 *>(&wsm[cur][bbase + u * 32 * VB + i * 32]);
            if (more) {
                const int t1 = tap + 1;
                const int toff = (((t1 / 9) * SP_HY + (t1 / 3) % 3) * SP_HX + t1 % 3) * VB;
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    na0[i] = *reinterpret_cast<const uint4 *>(halo + abase + toff + i * 32);
                    na1[i] = *reinterpret_cast<const uint4 *>(halo + abase + AF1 + toff + i * 32);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#define SP_PROD(IA, IB)                                                                                                        \
            _Pragma("unroll") for (int u = 0; u < NT; ++u) {                                                                   \
                acc[0][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0[IA]), __builtin_bit_cast(bf16x8, bf[u][IB]), acc[0][u], 0, 0, 0); \
                acc[1][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[IA]), __builtin_bit_cast(bf16x8, bf[u][IB]), acc[1][u], 0, 0, 0); \
            }
            // smallest terms first
            if (P == 3) { SP_PROD(P - 1, 0) SP_PROD(1, P - 2) SP_PROD(0, P - 1) }
            SP_PROD(1, 0) SP_PROD(0, 1) SP_PROD(0, 0)
#undef SP_PROD
            __builtin_amdgcn_sched_barrier(0);
            if (more) {
#pragma unroll
                for (int i = 0; i < WPT; ++i) {
                    const int idx = tid + i * 256;
                    if (idx < WV) *reinterpret_cast<uint4 *>(&wsm[cur ^ 1][(idx / (P * 2)) * VB + (idx % (P * 2)) * 16]) = wreg[i];
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < P; ++i) { a0[i] = na0[i]; a1[i] = na1[i]; }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int q = 0; q < 16; ++q) { tot[t][u][q] = __fadd_rn(tot[t][u][q], acc[t][u][q]); acc[t][u][q] = 0.f; }
    }
    // ---- epilogue (identical to the fp32 kernel)
    const int gz = z0 + wave;
    float ssum[NT], ssq[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) { ssum[u] = 0.f; ssq[u] = 0.f; }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int n = n0 + u * 32 + r;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * h;
                const int gy = y0 + t * 4 + (i >> 3), gx = x0 + (i & 7);
                if (gz < p.D && gy < p.H && gx < p.W) {
                    float v = tot[t][u][q];
                    if (p.relu) v = fmaxf(v, 0.f);
                    p.out[((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * p.Cout + n] = v;
                    ssum[u] += v;
                    ssq[u] = fmaf(v, v, ssq[u]);
                }
            }
        }
    if (p.osum) {
        float *red = reinterpret_cast<float *>(halo);
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const float s2 = ssum[u] + __shfl_xor(ssum[u], 32), q2 = ssq[u] + __shfl_xor(ssq[u], 32);
            if (h == 0) { red[wave * CT + u * 32 + r] = s2; red[4 * CT + wave * CT + u * 32 + r] = q2; }
        }
        __syncthreads();
        if (tid < CT) {
            const double s4 = (double)red[tid] + (double)red[CT + tid] + (double)red[2 * CT + tid] + (double)red[3 * CT + tid];
            const double q4 = (double)red[4 * CT + tid] + (double)red[5 * CT + tid] + (double)red[6 * CT + tid] + (double)red[7 * CT + tid];
            atomicAdd(&p.osum[(int64_t)b * p.Cout + n0 + tid], s4);
            atomicAdd(&p.osq[(int64_t)b * p.Cout + n0 + tid], q4);
        }
    }
}

extern "C" int gn_conv3d_gcr_split(const float *src0, int C0, const float *src1, int C1, const float *a, const float *d,
                                   const void *wp_planes, int planes, int B, int D, int H, int W, int Cout, int relu, float *out,
                                   double *out_sum, double *out_sumsq, void *stream) {
    GN_REQUIRE(B >= 0 && D > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, "gn_conv3d_gcr_split: bad sizes");
    GN_REQUIRE(planes == 2 || planes == 3, "gn_conv3d_gcr_split: planes must be 2 or 3");
    GN_REQUIRE(C0 % SP_KS == 0 && C1 % SP_KS == 0 && Cout % 32 == 0, "gn_conv3d_gcr_split: channel counts must be multiples of 16 (in) / 32 (out)");
    GN_REQUIRE(C1 == 0 || (src1 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0), "gn_conv3d_gcr_split: upsampled source needs even dims");
    GN_REQUIRE((out_sum == nullptr) == (out_sumsq == nullptr), "gn_conv3d_gcr_split: out_sum and out_sumsq must come together");
    if (B == 0) return GN_OK;
    hipStream_t st = gn_stream(stream);
    if (out_sum) {
        GN_HIP(hipMemsetAsync(out_sum, 0, sizeof(double) * (size_t)B * Cout, st), "gn_conv3d_gcr_split");
        GN_HIP(hipMemsetAsync(out_sumsq, 0, sizeof(double) * (size_t)B * Cout, st), "gn_conv3d_gcr_split");
    }
    SplitArgs p;
    p.src0 = src0; p.src1 = src1; p.a = a; p.d = d; p.wp = (const uint4 *)wp_planes; p.out = out; p.osum = out_sum; p.osq = out_sumsq;
    p.C0 = C0; p.C1 = C1; p.B = B; p.D = D; p.H = H; p.W = W; p.Cout = Cout; p.relu = relu;
    const int tz = (int)gn_cdiv(D, SP_TZ);
    p.tiles_y = (int)gn_cdiv(H, SP_TY);
    p.tiles_x = (int)gn_cdiv(W, SP_TX);
    const int tiles = tz * p.tiles_y * p.tiles_x;
    const bool wide = (Cout % 64 == 0) && ((int64_t)tiles * (Cout / 64) * B >= 1024);
    if (planes == 3) {
        if (wide) hipLaunchKernelGGL((conv3d_split_kernel<2, 3>), dim3(tiles * (Cout / 64), B), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3d_split_kernel<1, 3>), dim3(tiles * (Cout / 32), B), dim3(256), 0, st, p);
    } else {
        if (wide) hipLaunchKernelGGL((conv3d_split_kernel<2, 2>), dim3(tiles * (Cout / 64), B), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3d_split_kernel<1, 2>), dim3(tiles * (Cout / 32), B), dim3(256), 0, st, p);
    }
    GN_LAUNCH_CHECK("gn_conv3d_gcr_split");
    return GN_OK;
}
