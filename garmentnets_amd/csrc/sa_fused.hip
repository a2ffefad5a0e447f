// sa_fused.hip -- PointNet++ set abstraction in ONE kernel (gn_sa_fused): grouping gather -> 3-layer edge MLP on the matrix cores ->
// eval-BatchNorm affine -> segmented max, with no edge tensor in HBM.
//
// Reference: /root/reference/components/pointnet2.py:22-33 (SAModule.forward: PointConv(local_nn), aggr = max, add_self_loops) on the
// neighbour tables gn_ball_query produced.  The unfused chain it replaces (gn_sa_gather -> gn_linear x3 -> gn_segment_max) wrote
// [M*65][C+3] edge rows and two hidden tensors to HBM and read them back (1.5 GB per level at batch 16).
//
// Work decomposition.  A workgroup owns G consecutive centres.  Its work is a list of TILES of 32 edge rows: tile 2c+t = neighbour
// slots 32t..32t+31 of centre c (skipped when the ball holds no more than 32t points), plus one last tile with the G self-loop edges
// (PyG's bipartite add_self_loops quirk: centre i also receives POINT i of the whole cloud -- SURVEY.md 8a row 4).  The four waves
// pull tiles from an LDS counter; a tile is one wave's business from the gather to the max.
//
// Arithmetic: exact fp32 products on v_mfma_f32_32x32x2_f32 (the arithmetic of gn_linear), TRANSPOSED chain: a wave computes
// H^T[units][32 edges] = W[units][K] . X^T[K][32 edges] -- the weights are the A operand (one float per lane per MFMA, packed on the
// host so that a lane's four consecutive MFMAs read one 16-byte word), the activations the B operand.  The D fragment of a 32-unit
// block holds, in register q of lane (h, r = edge), unit 8(q>>2) + (q&3) + 4h: exactly the pair of k values (one per lane half) a B
// operand of the NEXT layer needs at step q once that layer's weight columns are packed in the same order.  So bias / ReLU / BatchNorm
// are applied to the accumulators in place and the registers ARE the next layer's operand: no LDS round trip, no barrier.  The first
// layer's operand comes straight from the gathered rows: lane half h loads the 16-byte chunks 8t+4h.. of its edge's feature row
// (features are L2-resident: every row is re-read by ~60 centres), the relative position is formed in registers.
// Segmented max: per 32-unit block of the last layer, invalid slots -> -inf, 32-lane DPP/bpermute max, one LDS float max per
// (centre, unit) (ordered-int encoding); the G x N3 tile of results is written once, coalesced, rows without any edge -> 0 (PyG).
#include "common.h"

typedef float f32x16a __attribute__((ext_vector_type(16)));

struct SaArgs {
    const float *x; int ldx;              // [N][ldx] point features (CIN channels) or NULL when CIN == 0
    const float *pos;                     // [N][3]
    const int32_t *centre_idx;            // [M] point index of every centre
    const int32_t *nbr;                   // [M][K] ball-query table, valid entries first, -1 padded
    const int32_t *cnt;                   // [M] number of valid entries
    int M, K, self_loops;
    const int32_t *self_src;              // [M] or NULL: the point that plays "node c" for the self-loop rule (NULL: point c of the whole cloud)
    const float4 *w1, *w2, *w3;           // A-fragment packs [N/32][Kblocks][4][64 lanes] x float4 (ops.pack_sa_fused)
    const float *tab;                     // per layer, 32-unit block, lane half: bias[16] | bn scale[16] | bn shift[16], register order
    float *out; int ldo;                  // [M][ldo]
};

__device__ __forceinline__ f32x16a sa_mfma(float a, float b, const f32x16a &c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

template <int CTRL>
__device__ __forceinline__ float sa_dpp(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true)); }

// max over the 32 lanes of each lane half (every lane of the half ends up with it)
__device__ __forceinline__ float sa_half_max(float v) {
    v = fmaxf(v, sa_dpp<0xB1>(v));        // quad_perm [1,0,3,2]
    v = fmaxf(v, sa_dpp<0x4E>(v));        // quad_perm [2,3,0,1]
    v = fmaxf(v, sa_dpp<0x141>(v));       // row_half_mirror
    v = fmaxf(v, sa_dpp<0x140>(v));       // row_mirror: every 16-lane row holds its max
    return fmaxf(v, __shfl_xor(v, 16));   // rows 0|1 and 2|3
}

// order-preserving float -> int (signed compare): LDS atomic max on floats of either sign
__device__ __forceinline__ int sa_enc(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float sa_dec(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// bias -> ReLU -> BatchNorm affine on one 32-unit block, in place (the op order of gn_linear's epilogue)
__device__ __forceinline__ void sa_epilogue(f32x16a &a, const float *tb) {
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const float4 b = *reinterpret_cast<const float4 *>(tb + 4 * q4);
        const float4 sc = *reinterpret_cast<const float4 *>(tb + 16 + 4 * q4);
        const float4 sh = *reinterpret_cast<const float4 *>(tb + 32 + 4 * q4);
        a[4 * q4 + 0] = __fadd_rn(__fmul_rn(gn_relu(__fadd_rn(a[4 * q4 + 0], b.x)), sc.x), sh.x);
        a[4 * q4 + 1] = __fadd_rn(__fmul_rn(gn_relu(__fadd_rn(a[4 * q4 + 1], b.y)), sc.y), sh.y);
        a[4 * q4 + 2] = __fadd_rn(__fmul_rn(gn_relu(__fadd_rn(a[4 * q4 + 2], b.z)), sc.z), sh.z);
        a[4 * q4 + 3] = __fadd_rn(__fmul_rn(gn_relu(__fadd_rn(a[4 * q4 + 3], b.w)), sc.w), sh.w);
    }
}

// one 32-unit output block: acc += W[32 units][NKB x 32 k] . B, B(kb, q) = the operand value of this lane for k-step q of block kb.
// The A fragments (one float4 per lane per four MFMAs) are double-buffered by hand, one k-block ahead, and a scheduling barrier per
// k-block keeps hipcc from hoisting the whole layer's loads to the top (which spilled 400 registers).  LASTQ: float4 chunks in the last block.
// `w` is a RUNNING pointer (the pack is laid out in traversal order, 4 KB per k-block: the four loads of a block are one base +
// immediate offsets) that the caller launders once per tile -- otherwise hipcc precomputes one 64-bit address per load as a
// loop-invariant of the tile loop (hundreds of registers, spilled).
template <int NKB, int LASTQ, typename BF>
__device__ __forceinline__ void sa_block(f32x16a &acc, const float4 *&w, BF bsrc) {
    float4 wc[4], wn[4];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) wc[qq] = w[qq * 64];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) {
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) wn[qq] = w[256 + qq * 64];
        }
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
            if (kb == NKB - 1 && qq >= LASTQ) continue;
            acc = sa_mfma(wc[qq].x, bsrc(kb, 4 * qq + 0), acc);
            acc = sa_mfma(wc[qq].y, bsrc(kb, 4 * qq + 1), acc);
            acc = sa_mfma(wc[qq].z, bsrc(kb, 4 * qq + 2), acc);
            acc = sa_mfma(wc[qq].w, bsrc(kb, 4 * qq + 3), acc);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) wc[qq] = wn[qq];
        w += 256;
    }
}

template <int CIN, int N1, int N2, int N3, int G>
__global__ __launch_bounds__(256, 2) void sa_fused_kernel(SaArgs p) {
    constexpr int K1 = CIN + 3, K1P = (K1 + 7) & ~7, KB1 = (K1P + 31) / 32;      // layer-1 input: [x_j (CIN) | pos_j - pos_i (3) | 0 pad]
    constexpr int LASTQ1 = (K1P - 32 * (KB1 - 1)) / 8;
    constexpr int NB1 = N1 / 32, NB2 = N2 / 32, NB3 = N3 / 32;
    constexpr int TAB2 = NB1 * 96, TAB3 = TAB2 + NB2 * 96, TABN = TAB3 + NB3 * 96;
    static_assert(N1 % 32 == 0 && N2 % 32 == 0 && N3 % 32 == 0 && G <= 32, "block sizes");
    __shared__ int out_lds[G * N3];
    __shared__ __attribute__((aligned(16))) float tab[TABN];
    __shared__ int tile_ctr;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int c0 = blockIdx.x * G;
    for (int i = tid; i < G * N3; i += 256) out_lds[i] = sa_enc(-INFINITY);
    for (int i = tid; i < TABN; i += 256) tab[i] = p.tab[i];
    if (tid == 0) tile_ctr = 0;
    __syncthreads();
    const int ntiles = 2 * G + (p.self_loops ? 1 : 0);

    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(&tile_ctr, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= ntiles) break;
        // ---- this lane's edge: source point j (-1 = empty slot), centre point ci, local centre cl
        int j, ci, cl;
        if (t < 2 * G) {
            cl = t >> 1;
            const int c = c0 + cl, half = t & 1;
            if (c >= p.M) continue;
            if (p.cnt[c] <= 32 * half) continue;
            const int slot = 32 * half + r;
            j = slot < p.K ? p.nbr[(size_t)c * p.K + slot] : -1;
            if (p.self_loops && j == (p.self_src ? p.self_src[c] : c)) j = -1;   // remove_self_loops: numeric equality of source and target index
            ci = p.centre_idx[c];
        } else {                                          // add_self_loops(num_nodes = M): source = point c of the full cloud
            const int c = c0 + r;
            const bool ok = r < G && c < p.M;
            cl = r;
            j = ok ? (p.self_src ? p.self_src[c] : c) : -1;
            ci = ok ? p.centre_idx[c] : 0;
        }
        const bool valid = j >= 0;
        const size_t jj = valid ? (size_t)j : 0;
        const float r0 = __fsub_rn(p.pos[3 * jj + 0], p.pos[3 * (size_t)ci + 0]);
        const float r1 = __fsub_rn(p.pos[3 * jj + 1], p.pos[3 * (size_t)ci + 1]);
        const float r2 = __fsub_rn(p.pos[3 * jj + 2], p.pos[3 * (size_t)ci + 2]);
        const float *xr = CIN > 0 ? p.x + jj * p.ldx : nullptr;

        // ---- the gathered row, all loads in flight at once: xin[16 kb + 4 qq + i] = input 32 kb + 8 qq + 4 h + i of this lane's edge
        float xin[KB1 * 16];
#pragma unroll
        for (int kb = 0; kb < KB1; ++kb)
#pragma unroll
            for (int qq = 0; qq < 4; ++qq) {
                const int kk = 32 * kb + 8 * qq;
                if (kk + 8 <= CIN) {
                    const float4 v = *reinterpret_cast<const float4 *>(xr + kk + 4 * h);
                    xin[16 * kb + 4 * qq + 0] = v.x; xin[16 * kb + 4 * qq + 1] = v.y; xin[16 * kb + 4 * qq + 2] = v.z; xin[16 * kb + 4 * qq + 3] = v.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = kk + 4 * h + i, rk = k - CIN;
                        float v = 0.f;
                        if (kk < K1P) {
                            if (k < CIN) v = xr[k];
                            else if (rk == 0) v = r0;
                            else if (rk == 1) v = r1;
                            else if (rk == 2) v = r2;
                        }
                        xin[16 * kb + 4 * qq + i] = v;
                    }
                }
            }

        const float4 *w1 = p.w1 + lane, *w2 = p.w2 + lane, *w3 = p.w3 + lane;
        asm volatile("" : "+v"(w1), "+v"(w2), "+v"(w3));           // see sa_block
        // ---- layer 1: K1P inputs -> N1 units
        f32x16a a1[NB1];
#pragma unroll
        for (int nb = 0; nb < NB1; ++nb) {
#pragma unroll
            for (int q = 0; q < 16; ++q) a1[nb][q] = 0.f;
            sa_block<KB1, LASTQ1>(a1[nb], w1, [&](int kb, int q) { return xin[16 * kb + q]; });
            sa_epilogue(a1[nb], tab + (nb * 2 + h) * 48);
        }
        // ---- layer 2: N1 -> N2, the operand is the accumulator file of layer 1
        f32x16a a2[NB2];
#pragma unroll
        for (int nb = 0; nb < NB2; ++nb) {
#pragma unroll
            for (int q = 0; q < 16; ++q) a2[nb][q] = 0.f;
            sa_block<NB1, 4>(a2[nb], w2, [&](int kb, int q) { return a1[kb][q]; });
            sa_epilogue(a2[nb], tab + TAB2 + (nb * 2 + h) * 48);
        }
        // ---- layer 3, one 32-unit block at a time: N2 -> 32 units -> masked max over the tile's edges -> LDS
        const bool self_tile = t >= 2 * G;
#pragma unroll 1
        for (int nb = 0; nb < NB3; ++nb) {
            f32x16a a3;
#pragma unroll
            for (int q = 0; q < 16; ++q) a3[q] = 0.f;
            sa_block<NB2, 4>(a3, w3, [&](int kb, int q) { return a2[kb][q]; });
            sa_epilogue(a3, tab + TAB3 + (nb * 2 + h) * 48);
            int *const o = out_lds + cl * N3 + 32 * nb + 4 * h;
            if (self_tile) {                               // every lane is its own centre
                if (valid) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) atomicMax(o + 8 * (q >> 2) + (q & 3), sa_enc(a3[q]));
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float m = sa_half_max(valid ? a3[q] : -INFINITY);
                    if (r == 0) atomicMax(o + 8 * (q >> 2) + (q & 3), sa_enc(m));
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < G * N3; i += 256) {
        const int cl = i / N3, u = i % N3, c = c0 + cl;
        if (c < p.M) {
            const float v = sa_dec(out_lds[i]);
            p.out[(size_t)c * p.ldo + u] = v == -INFINITY ? 0.f : v;          // a centre without any edge: scatter-max leaves 0
        }
    }
}

// centres per workgroup.  A row of the edge MLP does not depend on its tile and the maximum not on its order: every G gives the same bits, so G is free to follow
// the launch geometry (round 6): cost(G) = (2 G + 1) / (2 G)  [the self-loop tile's share]  /  fill of the last round of resident workgroups -- 12 000 second-level
// centres of a batch of 16 in groups of 16 were 750 workgroups on 512 slots (1.46 rounds), in groups of 8 they are 2.93; 750 centres of one garment in groups of
// 8 were 94 workgroups on 256 CUs.  Resident workgroups per CU from the runtime's occupancy query, once per instantiation.
template <int CIN, int N1, int N2, int N3, int G>
static int sa_slots() {
    static const int slots = [] {
        int per_cu = 0, dev = 0, cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)sa_fused_kernel<CIN, N1, N2, N3, G>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount;
        return per_cu * cus;
    }();
    return slots;
}
template <int CIN, int N1, int N2, int N3, int G>
static double sa_cost(int M) {
    const double n = (double)gn_cdiv(M, G), slots = (double)sa_slots<CIN, N1, N2, N3, G>();
    const double rounds = (double)gn_cdiv((int64_t)n, (int64_t)slots);
    return (2.0 * G + 1.0) / (2.0 * G) * rounds * slots / n;
}
template <int CIN, int N1, int N2, int N3>
static int sa_launch(const SaArgs &p, hipStream_t st) {
    const double c32 = sa_cost<CIN, N1, N2, N3, 32>(p.M), c16 = sa_cost<CIN, N1, N2, N3, 16>(p.M), c8 = sa_cost<CIN, N1, N2, N3, 8>(p.M),
                 c4 = sa_cost<CIN, N1, N2, N3, 4>(p.M), c2 = sa_cost<CIN, N1, N2, N3, 2>(p.M);
    double best = c32;                              // ties go to the larger group (fewer weight re-reads)
    int g = 32;
    if (c16 < best) { best = c16; g = 16; }
    if (c8 < best) { best = c8; g = 8; }
    if (c4 < best) { best = c4; g = 4; }
    if (c2 < best) { best = c2; g = 2; }
    switch (g) {
        case 32: hipLaunchKernelGGL((sa_fused_kernel<CIN, N1, N2, N3, 32>), dim3((unsigned)gn_cdiv(p.M, 32)), dim3(256), 0, st, p); break;
        case 16: hipLaunchKernelGGL((sa_fused_kernel<CIN, N1, N2, N3, 16>), dim3((unsigned)gn_cdiv(p.M, 16)), dim3(256), 0, st, p); break;
        case 8: hipLaunchKernelGGL((sa_fused_kernel<CIN, N1, N2, N3, 8>), dim3((unsigned)gn_cdiv(p.M, 8)), dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL((sa_fused_kernel<CIN, N1, N2, N3, 4>), dim3((unsigned)gn_cdiv(p.M, 4)), dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((sa_fused_kernel<CIN, N1, N2, N3, 2>), dim3((unsigned)gn_cdiv(p.M, 2)), dim3(256), 0, st, p); break;
    }
    return 0;
}

// Instantiated edge MLPs [C + 3, N1, N2, N3]: the two the GarmentNets checkpoints ship (first two rows) and the other PointNet++ set-abstraction shapes in
// common use (round 5; the kernel template is generic in all four, an instantiation costs compile time only).  Anything else: the unfused chain.
#define SA_SHAPES(X)                                                                                                           \
    X(3, 64, 64, 128) X(128, 128, 128, 256)                                                                                    \
    X(3, 32, 32, 64) X(3, 32, 64, 128) X(3, 64, 64, 64) X(3, 64, 128, 128)                                                     \
    X(64, 64, 64, 128) X(64, 64, 128, 256) X(128, 128, 128, 128) X(128, 128, 256, 256) X(0, 64, 64, 128)

extern "C" int gn_sa_fused_supported(int C, int N1, int N2, int N3) {
#define SA_MATCH(c, n1, n2, n3) if (C == c && N1 == n1 && N2 == n2 && N3 == n3) return 1;
    SA_SHAPES(SA_MATCH)
#undef SA_MATCH
    return 0;
}

extern "C" int gn_sa_fused_scoped(const float *x, int ldx, int C, const float *pos, const int32_t *centre_idx, const int32_t *nbr,
                                  const int32_t *cnt, int M, int K, int self_loops, const int32_t *self_src, const float *w1p,
                                  const float *w2p, const float *w3p, const float *tab, int N1, int N2, int N3, float *out, int ldo,
                                  void *stream) {
    GN_REQUIRE(M >= 0 && K > 0 && K <= 64 && ldo >= N3, "gn_sa_fused: bad sizes (the ball-query table holds at most 64 neighbours)");
    GN_REQUIRE(gn_sa_fused_supported(C, N1, N2, N3), "gn_sa_fused: edge MLP [%d+3,%d,%d,%d] is not instantiated (see SA_SHAPES in csrc/sa_fused.hip); use gn_sa_gather + gn_linear + gn_segment_max", C, N1, N2, N3);
    GN_REQUIRE(C == 0 || (x && ldx >= C && (C < 8 || ldx % 4 == 0)), "gn_sa_fused: feature rows need a 16-byte aligned leading dimension");
    if (M == 0) return GN_OK;
    GN_REQUIRE(pos && centre_idx && nbr && cnt && w1p && w2p && w3p && tab && out, "gn_sa_fused: null pointer");
    SaArgs p;
    p.x = x; p.ldx = ldx; p.pos = pos; p.centre_idx = centre_idx; p.nbr = nbr; p.cnt = cnt; p.M = M; p.K = K; p.self_loops = self_loops;
    p.self_src = self_src;
    p.w1 = (const float4 *)w1p; p.w2 = (const float4 *)w2p; p.w3 = (const float4 *)w3p; p.tab = tab; p.out = out; p.ldo = ldo;
    hipStream_t st = gn_stream(stream);
#define SA_DISPATCH(c, n1, n2, n3) if (C == c && N1 == n1 && N2 == n2 && N3 == n3) sa_launch<c, n1, n2, n3>(p, st);
    SA_SHAPES(SA_DISPATCH)
#undef SA_DISPATCH
    GN_LAUNCH_CHECK("gn_sa_fused");
    return GN_OK;
}

extern "C" int gn_sa_fused(const float *x, int ldx, int C, const float *pos, const int32_t *centre_idx, const int32_t *nbr, const int32_t *cnt,
                           int M, int K, int self_loops, const float *w1p, const float *w2p, const float *w3p, const float *tab, int N1, int N2,
                           int N3, float *out, int ldo, void *stream) {
    return gn_sa_fused_scoped(x, ldx, C, pos, centre_idx, nbr, cnt, M, K, self_loops, nullptr, w1p, w2p, w3p, tab, N1, N2, N3, out, ldo, stream);
}
