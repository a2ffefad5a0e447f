// decode.hip -- trilinear feature sampling for the implicit decoder.
// Replaces F.grid_sample(mode='bilinear', padding_mode='border', align_corners=True) on a 5-D input as called by
// ImplicitWNFDecoder.forward (/root/reference/networks/conv_implicit_wnf.py:128-149), including its axis
// convention (query component 0 indexes the LAST volume axis) and the lattice of predict.py:145-147.
// The arithmetic follows ATen's grid_sampler_3d (unnormalize -> clip -> floor, weights as products of
// differences, accumulation order tnw,tne,tsw,tse,bnw,bne,bsw,bse).
#include "common.h"

__device__ __forceinline__ float src_index(float q, int size) {
    // qn = 2q-1 ; ((qn+1)/2)*(size-1) ; clip to [0,size-1]
    float qn = __fsub_rn(__fmul_rn(2.0f, q), 1.0f);
    float x = __fmul_rn(__fdiv_rn(__fadd_rn(qn, 1.0f), 2.0f), (float)(size - 1));
    return fminf((float)(size - 1), fmaxf(x, 0.0f));
}

// one wavefront per query, channels over lanes
__global__ __launch_bounds__(256) void trilinear_kernel(const float *__restrict__ vol, int D, int H, int W, int C,
                                                        const float *__restrict__ query, int Q, int64_t m0, int64_t M,
                                                        float *__restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    float qx, qy, qz;
    if (query) {
        qx = query[m * 3]; qy = query[m * 3 + 1]; qz = query[m * 3 + 2];
    } else {
        // gridding.py:139-159: grid_idx.float() * ((uc-lc)/(Q-1)) + (-lc), unit cube
        const int64_t g = m0 + m;
        const int k = (int)(g % Q), j = (int)((g / Q) % Q), i = (int)(g / ((int64_t)Q * Q));
        const float sc = __fdiv_rn(1.0f, __fsub_rn((float)Q, 1.0f));
        qx = __fadd_rn(__fmul_rn((float)i, sc), -0.0f);
        qy = __fadd_rn(__fmul_rn((float)j, sc), -0.0f);
        qz = __fadd_rn(__fmul_rn((float)k, sc), -0.0f);
    }
    const float ix = src_index(qx, W), iy = src_index(qy, H), iz = src_index(qz, D);
    const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
    const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
    const float wx1 = __fsub_rn(ix, fx0), wx0 = __fsub_rn(__fadd_rn(fx0, 1.0f), ix);
    const float wy1 = __fsub_rn(iy, fy0), wy0 = __fsub_rn(__fadd_rn(fy0, 1.0f), iy);
    const float wz1 = __fsub_rn(iz, fz0), wz0 = __fsub_rn(__fadd_rn(fz0, 1.0f), iz);
    float wgt[8];
    int64_t off[8];
    bool ok[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;  // order tnw,tne,tsw,tse,bnw,bne,bsw,bse
        wgt[c] = __fmul_rn(__fmul_rn(dx ? wx1 : wx0, dy ? wy1 : wy0), dz ? wz1 : wz0);
        const int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
        ok[c] = xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D;
        off[c] = (((int64_t)zz * H + yy) * W + xx) * C;
    }
    for (int ch = lane; ch < C; ch += 64) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (ok[c]) acc = __fadd_rn(acc, __fmul_rn(vol[off[c] + ch], wgt[c]));
        out[m * ldo + ch] = acc;
    }
}

extern "C" int gn_trilinear_sample(const float *vol, int D, int H, int W, int C, const float *query, int Q, int64_t m0, int64_t M,
                                   float *out, int ldo, void *stream) {
    GN_REQUIRE(D > 0 && H > 0 && W > 0 && C > 0 && M >= 0 && ldo >= C, "gn_trilinear_sample: bad sizes");
    GN_REQUIRE(query != nullptr || (Q > 1 && m0 >= 0 && m0 + M <= (int64_t)Q * Q * Q), "gn_trilinear_sample: bad lattice range");
    if (M == 0) return GN_OK;
    hipLaunchKernelGGL(trilinear_kernel, dim3((unsigned)gn_cdiv(M, 4)), dim3(256), 0, gn_stream(stream), vol, D, H, W, C, query, Q, m0, M,
                       out, ldo);
    GN_LAUNCH_CHECK("gn_trilinear_sample");
    return GN_OK;
}
