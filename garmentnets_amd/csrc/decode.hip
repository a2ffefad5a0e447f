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

// one query, channels over the lanes of one wavefront
__device__ __forceinline__ void tri_query(const float *__restrict__ vol, int D, int H, int W, int C, const float *__restrict__ query,
                                          int Q, int64_t m0, int64_t m, float *__restrict__ out, int ldo, int lane) {
    float qx, qy, qz;
    if (query) {
        qx = query[m * 3]; qy = query[m * 3 + 1]; qz = query[m * 3 + 2];
    } else {
        // gridding.py:139-159: grid_idx.float() * ((uc-lc)/(Q-1)) + (-lc), unit cube
        const unsigned g = (unsigned)(m0 + m), uq = (unsigned)Q;   // Q^3 < 2^32 (checked on the host): 32-bit divides
        const unsigned gq = g / uq;
        const int k = (int)(g - gq * uq), i = (int)(gq / uq), j = (int)(gq - (unsigned)i * uq);
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
    if (lane < 0) {
        // 16-byte mode (C % 4 == 0, C <= 128): lane = -(1 + channel quad); a query occupies C/4 lanes, so a wavefront
        // serves 64/(C/4) queries at once with 16-byte loads and stores
        const int ch = (-lane - 1) * 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (ok[c]) {
                const float4 v = *reinterpret_cast<const float4 *>(vol + off[c] + ch);
                acc.x = __fadd_rn(acc.x, __fmul_rn(v.x, wgt[c]));
                acc.y = __fadd_rn(acc.y, __fmul_rn(v.y, wgt[c]));
                acc.z = __fadd_rn(acc.z, __fmul_rn(v.z, wgt[c]));
                acc.w = __fadd_rn(acc.w, __fmul_rn(v.w, wgt[c]));
            }
        *reinterpret_cast<float4 *>(out + m * ldo + ch) = acc;
        return;
    }
    if ((C & 1) == 0) {
        // two channels per lane: one 8-byte load per corner per lane, 512 contiguous bytes per wave for C = 128
        for (int ch = lane * 2; ch < C; ch += 128) {
            float2 acc = make_float2(0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (ok[c]) {
                    const float2 v = *reinterpret_cast<const float2 *>(vol + off[c] + ch);
                    acc.x = __fadd_rn(acc.x, __fmul_rn(v.x, wgt[c]));
                    acc.y = __fadd_rn(acc.y, __fmul_rn(v.y, wgt[c]));
                }
            *reinterpret_cast<float2 *>(out + m * ldo + ch) = acc;
        }
        return;
    }
    for (int ch = lane; ch < C; ch += 64) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (ok[c]) acc = __fadd_rn(acc, __fmul_rn(vol[off[c] + ch], wgt[c]));
        out[m * ldo + ch] = acc;
    }
}

// each wavefront walks TRI_QPW consecutive queries (amortises wave launch, keeps several queries' gathers in flight)
#define TRI_QPW 8
__global__ __launch_bounds__(256) void trilinear_kernel(const float *__restrict__ vol, int D, int H, int W, int C,
                                                        const float *__restrict__ query, int Q, int64_t m0, int64_t M,
                                                        float *__restrict__ out, int ldo, int64_t vol_bs, int64_t out_bs) {
    // blockIdx.y: one volume of a batch with its own M queries and M output rows (gn_trilinear_sample_batch; strides 0 for the single-volume call)
    vol += (int64_t)blockIdx.y * vol_bs;
    out += (int64_t)blockIdx.y * out_bs;
    if (query) query += (int64_t)blockIdx.y * M * 3;
    const int lane = threadIdx.x & 63;
    const int64_t mb = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * TRI_QPW;
    const int lpq = C >> 2;   // lanes per query in 16-byte mode
    if ((C & 3) == 0 && lpq <= 32 && (64 % lpq) == 0 && (ldo & 3) == 0) {
        const int qpw = 64 / lpq, sub = lane / lpq, c4 = lane % lpq;
        for (int q = sub; q < TRI_QPW; q += qpw)
            if (mb + q < M) tri_query(vol, D, H, W, C, query, Q, m0, mb + q, out, ldo, -(1 + c4));
        return;
    }
#pragma unroll 2
    for (int q = 0; q < TRI_QPW; ++q)
        if (mb + q < M) tri_query(vol, D, H, W, C, query, Q, m0, mb + q, out, ldo, lane);
}

// ------------------------------------------------------------------------------------------------ lattice sampler, brick version
// The lattice of predict.py:145-147 sampled brick by brick (TB_I x TB_J x TB_K lattice points per workgroup): when the lattice is at
// least as fine as the volume (Q >= size: the north-star's 128^3 / 128^3, the shipped 128^3 / 32^3) neighbouring queries share
// almost all of their 8 corners, so the brick's voxel bounding box (<= 6 x 6 x 6 voxels) of one 32-channel group is DMA'd into
// LDS (global_load_lds, 128-byte pieces) and every corner is an LDS read: L2 traffic falls from 8 x 512 B per query to
// ~1.8 x 512 B.  Same arithmetic and accumulation order as tri_query (bit-identical output, checked in the tests).
#define TB_I 4
#define TB_J 4
#define TB_K 4
#define TB_Q (TB_I * TB_J * TB_K)

__device__ __forceinline__ float tb_coord(int i, int Q) {
    const float sc = __fdiv_rn(1.0f, __fsub_rn((float)Q, 1.0f));
    return __fadd_rn(__fmul_rn((float)i, sc), -0.0f);
}

__global__ __launch_bounds__(256) void trilinear_brick_kernel(const float *__restrict__ vol, int D, int H, int W, int C, int Q, int i_begin,
                                                              int i_end, float *__restrict__ out, int ldo, int cap_vox) {
    extern __shared__ __attribute__((aligned(16))) unsigned char tb_smem[];
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)tb_smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // one workgroup = one brick x ONE 32-channel group (stage -> compute once; other resident workgroups hide the DMA latency)
    const int ngroups = C >> 5;
    const int bk = blockIdx.x / ngroups, cg = blockIdx.x % ngroups, bj = blockIdx.y, bi = blockIdx.z;
    const int i0 = i_begin + bi * TB_I, j0 = bj * TB_J, k0 = bk * TB_K;
    const int i1 = min(i0 + TB_I, i_end) - 1, j1 = min(j0 + TB_J, Q) - 1, k1 = min(k0 + TB_K, Q) - 1;   // last lattice point per axis
    // voxel bounding box of the brick (src_index is monotonic): query component 0 (i) indexes the LAST volume axis
    const int lx = (int)floorf(src_index(tb_coord(i0, Q), W)), hx = min((int)floorf(src_index(tb_coord(i1, Q), W)) + 1, W - 1);
    const int ly = (int)floorf(src_index(tb_coord(j0, Q), H)), hy = min((int)floorf(src_index(tb_coord(j1, Q), H)) + 1, H - 1);
    const int lz = (int)floorf(src_index(tb_coord(k0, Q), D)), hz = min((int)floorf(src_index(tb_coord(k1, Q), D)) + 1, D - 1);
    const int ex = hx - lx + 1, ey = hy - ly + 1, ez = hz - lz + 1, nvox = ex * ey * ez;
    const bool in_lds = nvox <= cap_vox;            // always true for Q >= size (host-side bound); otherwise read the corners from L2

    // ---- this thread's TB_Q/32 queries: 8 lanes per query (4 channels each within a 32-channel group)
    constexpr int NQ = TB_Q / 32;
    const int part = tid & 7;
    float wgt[NQ][8];
    int vidx[NQ];                                    // LDS voxel index of corner (x0, y0, z0)
    int64_t goff[NQ];                                // its global voxel offset (elements / C)
    unsigned okm[NQ];                                // bit c: corner c inside the volume; bit 8: the query exists
    int64_t mrow[NQ];
#pragma unroll
    for (int qn = 0; qn < NQ; ++qn) {
        const int ql = qn * 32 + (tid >> 3);
        const int kk = ql % TB_K, jj = (ql / TB_K) % TB_J, ii = ql / (TB_K * TB_J);
        const int i = i0 + ii, j = j0 + jj, k = k0 + kk;
        const bool live = i <= i1 && j <= j1 && k <= k1;
        const float ix = src_index(tb_coord(live ? i : i0, Q), W), iy = src_index(tb_coord(live ? j : j0, Q), H), iz = src_index(tb_coord(live ? k : k0, Q), D);
        const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
        const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
        const float wx1 = __fsub_rn(ix, fx0), wx0 = __fsub_rn(__fadd_rn(fx0, 1.0f), ix);
        const float wy1 = __fsub_rn(iy, fy0), wy0 = __fsub_rn(__fadd_rn(fy0, 1.0f), iy);
        const float wz1 = __fsub_rn(iz, fz0), wz0 = __fsub_rn(__fadd_rn(fz0, 1.0f), iz);
        unsigned ok = live ? 256u : 0u;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
            wgt[qn][c] = __fmul_rn(__fmul_rn(dx ? wx1 : wx0, dy ? wy1 : wy0), dz ? wz1 : wz0);
            if (x0 + dx < W && y0 + dy < H && z0 + dz < D) ok |= 1u << c;   // lower bounds hold after the border clamp
        }
        okm[qn] = ok;
        vidx[qn] = ((z0 - lz) * ey + (y0 - ly)) * ex + (x0 - lx);
        goff[qn] = ((int64_t)z0 * H + y0) * W + x0;
        mrow[qn] = ((int64_t)(i - i_begin) * Q + j) * Q + k;
    }

    {
        if (in_lds) {
            // stage [voxel][32 channels]: one DMA instruction per (z, y) row of the box = ex voxels x 8 16-byte pieces, lane = voxel * 8 +
            // piece (<= 48 active lanes; no integer divisions: waves take whole z planes)
            const int vx = lane >> 3, pc = lane & 7;
            for (int vz = wave; vz < ez; vz += 4)
                for (int vy = 0; vy < ey; ++vy) {
                    const float *g = vol + ((((int64_t)(lz + vz) * H + (ly + vy)) * W + (lx + vx)) * C + cg * 32 + pc * 4);
                    if (vx < ex) gn_glds16(g, lds_base + ((vz * ey + vy) * ex) * 128);
                }
            GN_WAIT_VM_LGKM0(0);
            __syncthreads();
            // corners straight from LDS, branch-free: a corner past the volume's upper face reads corner 0 (always inside, always finite when the
            // query's result is) with weight 0 -- acc + (+-0) == acc exactly and acc is never -0, so the sum is the guarded loop's bit for bit; one
            // basic block with 16 ds_read_b128 in flight instead of 16 exec-masked blocks around flat loads (the LDS / global pointer select)
#pragma unroll
            for (int qn = 0; qn < NQ; ++qn) {
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                const int vbase = vidx[qn] * 128 + part * 16;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
                    const bool okc = (okm[qn] >> c) & 1u;
                    const float w = okc ? wgt[qn][c] : 0.f;
                    int o = vbase + (((dz * ey + dy) * ex + dx) * 128 & -(int)okc);
                    asm volatile("" : "+v"(o));           // (opaque: hipcc otherwise turns the select back into 8 exec-masked loads)
                    const float4 v = *reinterpret_cast<const float4 *>(tb_smem + o);
                    acc.x = __fadd_rn(acc.x, __fmul_rn(v.x, w));
                    acc.y = __fadd_rn(acc.y, __fmul_rn(v.y, w));
                    acc.z = __fadd_rn(acc.z, __fmul_rn(v.z, w));
                    acc.w = __fadd_rn(acc.w, __fmul_rn(v.w, w));
                }
                if (okm[qn] & 256u) *reinterpret_cast<float4 *>(out + mrow[qn] * ldo + cg * 32 + part * 4) = acc;
            }
            return;
        }
        // (a lattice coarser than the volume: the corners come from L2)
#pragma unroll
        for (int qn = 0; qn < NQ; ++qn) {
            if (!(okm[qn] & 256u)) continue;
            float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (okm[qn] & (1u << c)) {
                    const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
                    const float4 v = *reinterpret_cast<const float4 *>(vol + (goff[qn] + ((int64_t)dz * H + dy) * W + dx) * C + cg * 32 + part * 4);
                    acc.x = __fadd_rn(acc.x, __fmul_rn(v.x, wgt[qn][c]));
                    acc.y = __fadd_rn(acc.y, __fmul_rn(v.y, wgt[qn][c]));
                    acc.z = __fadd_rn(acc.z, __fmul_rn(v.z, wgt[qn][c]));
                    acc.w = __fadd_rn(acc.w, __fmul_rn(v.w, wgt[qn][c]));
                }
            *reinterpret_cast<float4 *>(out + mrow[qn] * ldo + cg * 32 + part * 4) = acc;
        }
    }
}

extern "C" int gn_trilinear_sample(const float *vol, int D, int H, int W, int C, const float *query, int Q, int64_t m0, int64_t M,
                                   float *out, int ldo, void *stream) {
    GN_REQUIRE(D > 0 && H > 0 && W > 0 && C > 0 && M >= 0 && ldo >= C, "gn_trilinear_sample: bad sizes");
    if (M == 0) return GN_OK;   // (an empty query tensor has a NULL data pointer: not a lattice request)
    GN_REQUIRE(query != nullptr || (Q > 1 && Q <= 1024 && m0 >= 0 && m0 + M <= (int64_t)Q * Q * Q), "gn_trilinear_sample: bad lattice range");
    GN_REQUIRE((C & 1) || (ldo % 2 == 0), "gn_trilinear_sample: even channel counts need an even output leading dimension");
    if (M == 0) return GN_OK;
    // lattice chunks made of whole i-slabs, lattice at least as fine as the volume, 32-channel groups: the brick kernel
    const int64_t slab = (int64_t)Q * Q;
    if (query == nullptr && C % 32 == 0 && ldo % 4 == 0 && m0 % slab == 0 && M % slab == 0 && Q >= D && Q >= H && Q >= W) {
        const int cap_vox = (TB_I + 2) * (TB_J + 2) * (TB_K + 2);           // extent <= points + 2 per axis when the spacing is <= 1 voxel
        const size_t lds = (((size_t)cap_vox * 128 + 4095) / 4096) * 4096;
        const int i_begin = (int)(m0 / slab), i_end = (int)((m0 + M) / slab);
        hipLaunchKernelGGL(trilinear_brick_kernel, dim3((unsigned)(gn_cdiv(Q, TB_K) * (C / 32)), (unsigned)gn_cdiv(Q, TB_J), (unsigned)gn_cdiv(i_end - i_begin, TB_I)),
                           dim3(256), lds, gn_stream(stream), vol, D, H, W, C, Q, i_begin, i_end, out, ldo, cap_vox);
        GN_LAUNCH_CHECK("gn_trilinear_sample");
        return GN_OK;
    }
    hipLaunchKernelGGL(trilinear_kernel, dim3((unsigned)gn_cdiv(M, 4 * TRI_QPW)), dim3(256), 0, gn_stream(stream), vol, D, H, W, C, query, Q, m0, M,
                       out, ldo, (int64_t)0, (int64_t)0);
    GN_LAUNCH_CHECK("gn_trilinear_sample");
    return GN_OK;
}

extern "C" int gn_trilinear_sample_batch(const float *vol, int B, int64_t vol_bstride, int D, int H, int W, int C, const float *query, int64_t M, float *out,
                                         int ldo, void *stream) {
    GN_REQUIRE(B >= 0 && B <= 65535 && D > 0 && H > 0 && W > 0 && C > 0 && M >= 0 && ldo >= C, "gn_trilinear_sample_batch: bad sizes");
    GN_REQUIRE((C & 1) || (ldo % 2 == 0), "gn_trilinear_sample_batch: even channel counts need an even output leading dimension");
    if (M == 0 || B == 0) return GN_OK;
    GN_REQUIRE(query != nullptr && vol != nullptr && out != nullptr, "gn_trilinear_sample_batch: null pointer");
    hipLaunchKernelGGL(trilinear_kernel, dim3((unsigned)gn_cdiv(M, 4 * TRI_QPW), (unsigned)B), dim3(256), 0, gn_stream(stream), vol, D, H, W, C, query, 0,
                       (int64_t)0, M, out, ldo, vol_bstride, M * (int64_t)ldo);
    GN_LAUNCH_CHECK("gn_trilinear_sample_batch");
    return GN_OK;
}

// ================================================================================================ fused implicit decoder
// gn_implicit_decode: trilinear sample -> Linear/ReLU/BN (C0 -> N1) -> Linear/ReLU/BN (N1 -> N2) -> Linear/ReLU/BN (N2 -> OUT<=4)
// in ONE kernel (ImplicitWNFDecoder.forward, networks/conv_implicit_wnf.py:128-149 with MLP [128,256,256,out]).
//
// Block = 256 threads (4 waves) = 32 queries.  Activations never leave the CU: the sampled features X0 [32][C0], H1 [32][N1]
// and H2 [32][N2] live in LDS (row stride odd -> the 32 rows of an MFMA A fragment hit 32 distinct banks); two regions are
// ping-ponged (P: X0 then H2, Q: H1) = 66 KB for [128,256,256] -> 2 workgroups per CU, so one samples while the other feeds
// the matrix cores.  The weights (394 KB, L2-resident) are NOT staged through LDS: they are pre-packed k-pair-major
// Wp[k/16][n][2][8] (see dec_layer), so the B operands of 8 consecutive MFMAs are two coalesced 16-byte loads per lane,
// register double-buffered one 16-deep k-group ahead.  Each wave owns 64 output columns of a
// 256-column block (2 column fragments x 1 row fragment = 32 accumulator registers).
typedef float f32x16d __attribute__((ext_vector_type(16)));

struct DecodeArgs {
    const float *vol; int D, H, W, C0;
    const float *xin; int ldxin;          // optional pre-sampled features [M][C0] (then vol/query are unused)
    const float *query; int Q; long long m0, M;
    const float *w1p, *b1, *s1, *t1; int N1;
    const float *w2p, *b2, *s2, *t2; int N2;
    const float *w3, *b3, *s3, *t3; int OUT;
    float *out; int ldo;
    const float *run_if;                  // NULL, or a device flag: the kernel does nothing unless *run_if != 0
    long long xin_bs, out_bs; int run_if_bs;   // blockIdx.y: one row set of a batch (gn_implicit_decode_batch); 0 for the single call
};

#define DEC_TM 32
#define DEC_MAX_GRID 4096             // workgroups of a launch: 256 CUs x at most 3 resident workgroups, a few rounds of them; each walks its share of the row tiles

// one dense layer on the 32 resident rows: Y[32][N] = bn(relu(X[32][K] W^T + b)), X and Y in LDS (row stride K+4 / N+4
// floats: 16-byte aligned rows whose 16-byte slots rotate with the row -> conflict-free ds_read_b128).
// K is consumed in groups of 16 with a PERMUTED k order (the sum is order-free up to rounding): lanes 0-31 supply
// k = 16g+j, lanes 32-63 k = 16g+8+j for MFMA j = 0..7, so each lane reads 8 CONSECUTIVE floats of its row (2 x b128) per
// group, and the matching B values come from the pack Wp[g][n][h][j] = W[n][16g+8h+j]: 32 contiguous bytes per lane, 2 KB
// contiguous per wave (2 x global_load_dwordx4 per column fragment per group).  Register double-buffering: group g+1 is
// in flight while the 16 MFMAs of group g issue.
struct BFrag { float4 lo, hi; };

__device__ __forceinline__ void dec_mfma8(const float4 &a0, const float4 &a1, const BFrag &b0, const BFrag &b1, f32x16d &acc0, f32x16d &acc1) {
#define DEC_STEP(A, B0, B1)                                                   \
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B0, acc0, 0, 0, 0);        \
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(A, B1, acc1, 0, 0, 0);
    DEC_STEP(a0.x, b0.lo.x, b1.lo.x) DEC_STEP(a0.y, b0.lo.y, b1.lo.y) DEC_STEP(a0.z, b0.lo.z, b1.lo.z) DEC_STEP(a0.w, b0.lo.w, b1.lo.w)
    DEC_STEP(a1.x, b0.hi.x, b1.hi.x) DEC_STEP(a1.y, b0.hi.y, b1.hi.y) DEC_STEP(a1.z, b0.hi.z, b1.hi.z) DEC_STEP(a1.w, b0.hi.w, b1.hi.w)
#undef DEC_STEP
}

template <int CTRL>
__device__ __forceinline__ float dec_dpp(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }
// sum over the 16 lanes of a DPP row (quad xor 1, quad xor 2, row_half_mirror, row_mirror); every lane gets the sum
__device__ __forceinline__ float dec_row_sum(float v) {
    v += dec_dpp<0xB1>(v);
    v += dec_dpp<0x4E>(v);
    v += dec_dpp<0x141>(v);
    v += dec_dpp<0x140>(v);
    return v;
}

// OUTC == 0: write Y = bn(relu(X W^T + b)) to LDS.  OUTC > 0: this is the last hidden layer -- its activations never
// leave the registers: each lane multiplies its columns by the OUTC rows of w3 [OUTC][N] and the partial dot products are
// reduced over the 16-lane DPP rows into red[8 partials][32 rows][OUTC] (fixed summation order -> deterministic).
template <int OUTC>
__device__ __forceinline__ void dec_layer(const float *__restrict__ X, int ldx, int K, const float *__restrict__ Wp, const float *__restrict__ bias,
                                          const float *__restrict__ sc, const float *__restrict__ sh, int N, float *__restrict__ Y, int ldy,
                                          const float *__restrict__ w3 = nullptr, float *__restrict__ red = nullptr) {
    float psum[16][OUTC > 0 ? OUTC : 1];
    if (OUTC > 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int o = 0; o < (OUTC > 0 ? OUTC : 1); ++o) psum[q][o] = 0.f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
    const float4 *xrow = reinterpret_cast<const float4 *>(X + r * ldx + 8 * h);   // + 4*g float4 per group
    const int ng = K >> 4;                                                          // even (K % 32 == 0)
    const size_t gstride = (size_t)N * 4;                                           // float4 per k-group of the pack
    for (int nb = 0; nb < N; nb += 256) {
        const int n0 = nb + wave * 64;
        f32x16d acc0, acc1;
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
        const float4 *w0 = reinterpret_cast<const float4 *>(Wp) + ((size_t)(n0 + r) * 2 + h) * 2;   // column fragment 0
        const float4 *w1 = w0 + 32 * 2 * 2;                                                          // column fragment 1 (n + 32)
        BFrag c0, c1, d0, d1;
        c0.lo = w0[0]; c0.hi = w0[1]; c1.lo = w1[0]; c1.hi = w1[1];
        for (int g = 0; g < ng; g += 2) {
            {   // prefetch group g+1 into d*, compute group g from c*
                const float4 *p0 = w0 + (size_t)(g + 1) * gstride, *p1 = w1 + (size_t)(g + 1) * gstride;
                d0.lo = p0[0]; d0.hi = p0[1]; d1.lo = p1[0]; d1.hi = p1[1];
                const float4 a0 = xrow[4 * g], a1 = xrow[4 * g + 1];
                __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMA group (hipcc sinks it otherwise)
                dec_mfma8(a0, a1, c0, c1, acc0, acc1);
                __builtin_amdgcn_sched_barrier(0);
            }
            {   // prefetch group g+2 into c*, compute group g+1 from d*
                const int gn = (g + 2 < ng) ? g + 2 : g + 1;
                const float4 *p0 = w0 + (size_t)gn * gstride, *p1 = w1 + (size_t)gn * gstride;
                c0.lo = p0[0]; c0.hi = p0[1]; c1.lo = p1[0]; c1.hi = p1[1];
                const float4 a0 = xrow[4 * (g + 1)], a1 = xrow[4 * (g + 1) + 1];
                __builtin_amdgcn_sched_barrier(0);
                dec_mfma8(a0, a1, d0, d1, acc0, acc1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int n = n0 + u * 32 + r;
            const float bv = bias[n], scv = sc ? sc[n] : 1.f, shv = sh ? sh[n] : 0.f;
            float w3v[OUTC > 0 ? OUTC : 1];
            if (OUTC > 0) {
#pragma unroll
                for (int o = 0; o < (OUTC > 0 ? OUTC : 1); ++o) w3v[o] = w3[(size_t)o * N + n];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
                float v = __fadd_rn(u == 0 ? acc0[q] : acc1[q], bv);
                v = gn_relu(v);
                if (sc) v = __fadd_rn(__fmul_rn(v, scv), shv);
                if (OUTC > 0) {
#pragma unroll
                    for (int o = 0; o < (OUTC > 0 ? OUTC : 1); ++o) psum[q][o] = fmaf(v, w3v[o], psum[q][o]);
                } else {
                    Y[row * ldy + n] = v;
                }
            }
        }
    }
    if (OUTC > 0) {
        const int part = wave * 2 + ((lane >> 4) & 1);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
#pragma unroll
            for (int o = 0; o < (OUTC > 0 ? OUTC : 1); ++o) {
                const float sum = dec_row_sum(psum[q][o]);
                if ((lane & 15) == 0) red[(part * DEC_TM + row) * OUTC + o] = sum;
            }
        }
    }
}

template <int OUTC>
__global__ __launch_bounds__(256, 3) void implicit_decode_kernel(DecodeArgs p) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (p.run_if && p.run_if[(long long)blockIdx.y * p.run_if_bs] == 0.f) return;       // gated launch (the split-operand kernel handled these rows)
    if (p.xin) p.xin += (long long)blockIdx.y * p.xin_bs;
    p.out += (long long)blockIdx.y * p.out_bs;
    const int ldp = p.C0 + 4, ldq = p.N1 + 4;
    float *P = dsm, *Qb = dsm + DEC_TM * ldp, *red = Qb + DEC_TM * ldq;   // X0 | H1 | 8 x 32 x OUT partial dot products
    // a bounded grid walking the row tiles (round 6): a GATED launch that has nothing to do (the common case: the split-operand kernel handled the rows)
    // costs its workgroups' start-up only -- 21 845 of them for a third of a 128^3 lattice were 8.3 us per launch, 32 launches per 16-garment step
    for (long long mb = (long long)blockIdx.x * DEC_TM; mb < p.M; mb += (long long)gridDim.x * DEC_TM) {
    // ---- phase 0: X0 = pre-sampled rows (coalesced 16-byte loads) or trilinear sampling of 8 queries per wave
    if (p.xin) {
        const int c4n = p.C0 >> 2;
        for (int idx = threadIdx.x; idx < DEC_TM * c4n; idx += 256) {
            const int row = idx / c4n, c4 = idx % c4n;
            long long m = mb + row;
            if (m >= p.M) m = p.M - 1;
            *reinterpret_cast<float4 *>(P + row * ldp + c4 * 4) = *reinterpret_cast<const float4 *>(p.xin + m * p.ldxin + c4 * 4);
        }
    } else
    for (int qi = 0; qi < DEC_TM / 4; ++qi) {
        const int row = wave * (DEC_TM / 4) + qi;
        long long m = mb + row;
        if (m >= p.M) m = p.M - 1;   // tail rows recompute the last query; never stored
        float qx, qy, qz;
        if (p.query) {
            qx = p.query[m * 3]; qy = p.query[m * 3 + 1]; qz = p.query[m * 3 + 2];
        } else {
            const unsigned g = (unsigned)(p.m0 + m), uq = (unsigned)p.Q;
            const unsigned gq = g / uq;
            const int k = (int)(g - gq * uq), i = (int)(gq / uq), j = (int)(gq - (unsigned)i * uq);
            const float sc = __fdiv_rn(1.0f, __fsub_rn((float)p.Q, 1.0f));
            qx = __fadd_rn(__fmul_rn((float)i, sc), -0.0f);
            qy = __fadd_rn(__fmul_rn((float)j, sc), -0.0f);
            qz = __fadd_rn(__fmul_rn((float)k, sc), -0.0f);
        }
        const float ix = src_index(qx, p.W), iy = src_index(qy, p.H), iz = src_index(qz, p.D);
        const float fx0 = floorf(ix), fy0 = floorf(iy), fz0 = floorf(iz);
        const int x0 = (int)fx0, y0 = (int)fy0, z0 = (int)fz0;
        const float wx1 = __fsub_rn(ix, fx0), wx0 = __fsub_rn(__fadd_rn(fx0, 1.0f), ix);
        const float wy1 = __fsub_rn(iy, fy0), wy0 = __fsub_rn(__fadd_rn(fy0, 1.0f), iy);
        const float wz1 = __fsub_rn(iz, fz0), wz0 = __fsub_rn(__fadd_rn(fz0, 1.0f), iz);
        for (int ch = lane; ch < p.C0; ch += 64) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int dx = c & 1, dy = (c >> 1) & 1, dz = c >> 2;
                const int xx = x0 + dx, yy = y0 + dy, zz = z0 + dz;
                if (xx < p.W && yy < p.H && zz < p.D) {   // lower bounds hold after the border clamp
                    const float wgt = __fmul_rn(__fmul_rn(dx ? wx1 : wx0, dy ? wy1 : wy0), dz ? wz1 : wz0);
                    acc = __fadd_rn(acc, __fmul_rn(p.vol[(((long long)zz * p.H + yy) * p.W + xx) * p.C0 + ch], wgt));
                }
            }
            P[row * ldp + ch] = acc;
        }
    }
    __syncthreads();
    dec_layer<0>(P, ldp, p.C0, p.w1p, p.b1, p.s1, p.t1, p.N1, Qb, ldq);
    __syncthreads();
    // second hidden layer + the (N2 -> OUT) output layer; H2 stays in registers
    dec_layer<OUTC>(Qb, ldq, p.N1, p.w2p, p.b2, p.s2, p.t2, p.N2, nullptr, 0, p.w3, red);
    __syncthreads();
    if (threadIdx.x < DEC_TM * OUTC) {
        const int row = threadIdx.x / OUTC, o = threadIdx.x % OUTC;
        const long long m = mb + row;
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) sacc += red[(j * DEC_TM + row) * OUTC + o];
        if (m < p.M) {
            float v = gn_relu(__fadd_rn(sacc, p.b3[o]));
            if (p.s3) v = __fadd_rn(__fmul_rn(v, p.s3[o]), p.t3[o]);
            p.out[m * p.ldo + o] = v;
        }
    }
    __syncthreads();                                 // the next tile's rows overwrite P / red
    }
}

static int implicit_decode_impl(const float *vol, int D, int H, int W, int C0, const float *xin, int ldxin, const float *query, int Q, int64_t m0, int64_t M,
                                const float *w1p, const float *b1, const float *s1, const float *t1, int N1, const float *w2p,
                                const float *b2, const float *s2, const float *t2, int N2, const float *w3, const float *b3,
                                const float *s3, const float *t3, int OUT, float *out, int ldo, const float *run_if, int B, int run_if_bs, void *stream) {
    GN_REQUIRE((xin != nullptr || (vol != nullptr && D > 0 && H > 0 && W > 0)) && M >= 0 && ldo >= OUT, "gn_implicit_decode: bad sizes");
    if (M == 0) return GN_OK;
    GN_REQUIRE(xin == nullptr || (ldxin >= C0 && ldxin % 4 == 0), "gn_implicit_decode: pre-sampled rows need a 16-byte aligned leading dimension");
    GN_REQUIRE(C0 % 32 == 0 && N1 % 256 == 0 && N2 % 256 == 0 && OUT >= 1 && OUT <= 4,
               "gn_implicit_decode: unsupported layer widths [%d,%d,%d,%d] (need C0 %% 32 == 0, N1 and N2 multiples of 256, out <= 4)", C0, N1, N2, OUT);
    GN_REQUIRE(xin != nullptr || query != nullptr || (Q > 1 && Q <= 1024 && m0 >= 0 && m0 + M <= (int64_t)Q * Q * Q), "gn_implicit_decode: bad lattice range");
    GN_REQUIRE((s1 == nullptr) == (t1 == nullptr) && (s2 == nullptr) == (t2 == nullptr) && (s3 == nullptr) == (t3 == nullptr),
               "gn_implicit_decode: BN scale and shift must come together");
    if (M == 0) return GN_OK;
    DecodeArgs p;
    p.vol = vol; p.D = D; p.H = H; p.W = W; p.C0 = C0; p.xin = xin; p.ldxin = ldxin; p.query = query; p.Q = Q; p.m0 = m0; p.M = M;
    p.w1p = w1p; p.b1 = b1; p.s1 = s1; p.t1 = t1; p.N1 = N1; p.w2p = w2p; p.b2 = b2; p.s2 = s2; p.t2 = t2; p.N2 = N2;
    p.w3 = w3; p.b3 = b3; p.s3 = s3; p.t3 = t3; p.OUT = OUT; p.out = out; p.ldo = ldo; p.run_if = run_if;
    p.xin_bs = B > 1 ? M * (long long)ldxin : 0; p.out_bs = B > 1 ? M * (long long)ldo : 0; p.run_if_bs = B > 1 ? run_if_bs : 0;
    const int ldp = C0 + 4, ldq = N1 + 4;
    const size_t sh = sizeof(float) * DEC_TM * (size_t)(ldp + ldq + 8 * OUT);
    GN_REQUIRE(sh <= 160 * 1024, "gn_implicit_decode: layer widths need %zu bytes of LDS", sh);
    const int64_t tiles = gn_cdiv(M, DEC_TM), gcap = B > 1 ? (DEC_MAX_GRID / B > 0 ? DEC_MAX_GRID / B : 1) : DEC_MAX_GRID;
#define DEC_LAUNCH(O)                                                                                                                  \
    do {                                                                                                                               \
        GN_HIP(hipFuncSetAttribute((const void *)implicit_decode_kernel<O>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh), "gn_implicit_decode"); \
        hipLaunchKernelGGL(implicit_decode_kernel<O>, dim3((unsigned)(tiles < gcap ? tiles : gcap), (unsigned)B), dim3(256), sh, gn_stream(stream), p); \
    } while (0)
    switch (OUT) {
        case 1: DEC_LAUNCH(1); break;
        case 2: DEC_LAUNCH(2); break;
        case 3: DEC_LAUNCH(3); break;
        default: DEC_LAUNCH(4); break;
    }
#undef DEC_LAUNCH
    GN_LAUNCH_CHECK("gn_implicit_decode");
    return GN_OK;
}

extern "C" int gn_implicit_decode(const float *vol, int D, int H, int W, int C0, const float *xin, int ldxin, const float *query, int Q, int64_t m0, int64_t M,
                                  const float *w1p, const float *b1, const float *s1, const float *t1, int N1, const float *w2p,
                                  const float *b2, const float *s2, const float *t2, int N2, const float *w3, const float *b3,
                                  const float *s3, const float *t3, int OUT, float *out, int ldo, const float *run_if, void *stream) {
    return implicit_decode_impl(vol, D, H, W, C0, xin, ldxin, query, Q, m0, M, w1p, b1, s1, t1, N1, w2p, b2, s2, t2, N2, w3, b3, s3, t3, OUT, out, ldo, run_if, 1, 0,
                                stream);
}

// B row sets of M pre-sampled rows each through one launch (blockIdx.y = row set): xin [B][M][ldxin], out [B][M][ldo], run_if NULL or one flag per row
// set at run_if[b * run_if_stride] (the gated fp32 twin of gn_implicit_decode_split_batch: run_if = xscale + 2, stride 4)
extern "C" int gn_implicit_decode_batch(const float *xin, int ldxin, int64_t M, int B, int C0, const float *w1p, const float *b1, const float *s1, const float *t1,
                                        int N1, const float *w2p, const float *b2, const float *s2, const float *t2, int N2, const float *w3, const float *b3,
                                        const float *s3, const float *t3, int OUT, float *out, int ldo, const float *run_if, int run_if_stride, void *stream) {
    GN_REQUIRE(B >= 0 && B <= 65535 && xin != nullptr, "gn_implicit_decode_batch: bad sizes");
    if (B == 0) return GN_OK;
    return implicit_decode_impl(nullptr, 0, 0, 0, C0, xin, ldxin, nullptr, 0, 0, M, w1p, b1, s1, t1, N1, w2p, b2, s2, t2, N2, w3, b3, s3, t3, OUT, out, ldo, run_if, B,
                                run_if_stride, stream);
}
