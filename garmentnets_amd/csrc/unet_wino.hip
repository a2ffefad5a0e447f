// unet_wino.hip -- the 128-wide split-operand 'gcr' convolution with Winograd F(2,3) along x: 36 instead of 54 tap products per output pair.
//
// Reference layer: components/unet3d.py:53-76 (GroupNorm -> Conv3d 3x3x3 pad 1 -> ReLU); the shape this exists for is the first encoder
// convolution (128 -> 128 at full resolution, components/unet3d.py:127-133): conv3d_split_wide_kernel runs it at the socket's power limit,
// so the only lever left is FEWER matrix-core products per result.
//
// Algebra.  For an output pair (x, x+1), x even, and one (dz, dy, input channel) the three dx taps g0 g1 g2 over the inputs d0..d3 = in[x-1..x+2]:
//      m0 = (d0 - d2) g0    m1 = (d1 + d2) (g0 + g1 + g2)/2    m2 = (d2 - d1) (g0 - g1 + g2)/2    m3 = (d1 - d3) g2
//      out[x] = m0 + m1 + m2        out[x+1] = m1 - m2 - m3
// Four products instead of six.  The input transform is done in fp32 while the halo is staged (BEFORE the fp16 plane split, so the planes
// still carry an exact two-term decomposition of what the matrix cores multiply), the weight transform on the pack side in fp64
// (gn_conv_affine_pack_wino per sample, ops.pack_conv_weight_split_wino for a static pack), the output transform in fp32 on the accumulators.
// Every transform is linear, so an operand that is exactly zero (the affine-in-weights form away from the cells) stays exactly zero.
//
// Structure (same skeleton as conv3d_split_wide_kernel: 512 threads = 8 waves = (z-slice zs, column group cg), tile 4 x 8 x 8 voxels x 128
// output channels, B fragments DMA'd global -> LDS in fragment order, one workgroup per CU):
//  * fragment rows are output PAIRS: a wave's 32 rows = (y 0..7, pair 0..3) of its z-slice -- one row fragment where the direct kernel has
//    two -- times the four transform positions j.  Loop order slice -> j -> dz -> dy: one accumulator pair (2 column fragments) lives through
//    the 9 (dz, dy) steps of a j (27 MFMAs each), then the output transform adds it into the even / odd totals: 64 + 32 accumulator registers
//    instead of 64 + 64.
//  * LDS halo: per 16-channel slice and j one SLOT of 6 x 10 rows x 4 pairs x 64 B (row pitch 272 B = 17 bank quads: the 16-lane service
//    groups of ds_read_b128 -- rows (y, p), y in {0,3,5,6} or {1,2,4,7} -- hit 16 distinct quads).  FIVE slots rotate: slice s uses slots
//    (4 s + j) % 5; slice s+1's j-th slot is the one slice s's j-1 vacated (j = 0: the spare), so the next slice is staged while this one
//    is multiplied with 1.25 instead of 2 halo buffers (81.6 KB; a double buffer would not fit beside the B ring).
//  * B ring: a GROUP = the three dy steps of one (j, dz) = 24 KB, three groups deep.  One hand-over barrier per group (18 MFMAs per wave)
//    instead of one per tap (12); the first step's fragments of group G+1 are read at the end of group G (landed: guaranteed at G's barrier),
//    so no LDS latency follows a barrier.
//  * staging: thread = (halo row, x half, channel quad): six x-consecutive voxels -> the two pairs' transformed values for one j at a time,
//    converted at groups 2 / 4 / 7 / 10 (as soon as the target slot is free) from row registers loaded at group 0.
#include "split_conv.h"

struct WinoLayout {
    static constexpr int VB = 64;                      // bytes per transformed voxel: 2 planes x 16 halfs
    static constexpr int ROWP = 4 * VB + 16;           // 4 pairs per halo row + one 16-byte pad
    static constexpr int HZ = SP_TZ + 2, HY = SP_TY + 2;
    static constexpr int SLOT = HZ * HY * ROWP;        // one transform position of one slice: 16320 B
    static constexpr int NSLOT = 5;
};

typedef float f32x4n __attribute__((ext_vector_type(4)));

template <bool F16>
__global__ __launch_bounds__(512, 1) void conv3d_split_wino_kernel(SplitArgs p) {
    constexpr int P = 2, NT = 2;
    using WL = WinoLayout;
    constexpr int STEPB = 2 * NT * P * 1024;        // B fragments of one step, both column groups: 8 KB
    constexpr int GB = 3 * STEPB;                   // group = the three dy steps of one (j, dz)
    constexpr int RING = 3;
    constexpr int HALO_BYTES = WL::NSLOT * WL::SLOT;
    constexpr int AD_OFF = HALO_BYTES + RING * GB;
    constexpr int ADN = 256;                        // Cin <= 256 (checked by the launcher)
    constexpr int NIT = 6;                          // row loads per thread per slice
    __shared__ __attribute__((aligned(16))) unsigned char smem[AD_OFF + 2 * ADN * 4];
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + HALO_BYTES;
    float *const adl = reinterpret_cast<float *>(smem + AD_OFF);          // a[Cin] | d[Cin] of this sample
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), zs = wave & 3, cg = wave >> 2;
    const int Cin = p.C0;
    const int ncb = p.Cout / 128;
    const int tiles_z = p.D / SP_TZ;
    int b, tile, cb;
    if (!sp_work_item(p, ncb, tiles_z * p.tiles_x * p.tiles_y, b, tile, cb)) return;      // (workgroup-uniform)
    const int tz = tile % tiles_z; tile /= tiles_z;
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile;
    const int z0 = tz * SP_TZ, y0 = ty * SP_TY, x0 = tx * SP_TX;
    const int n0 = cb * 128 + cg * 64;
    const int nslices = Cin / SP_KS;

    // acc: the running transform position's accumulators; tot[e]: outputs at even (e = 0) / odd x of the pairs.  (A second accumulator set -- the
    // output transform of position j folded in under position j+1's MFMAs -- needs 261 registers: 46 spilled values inside the MFMA stream.)
    f32x16s acc[NT], tot[2][NT];
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[u][q] = 0.f; tot[0][u][q] = 0.f; tot[1][u][q] = 0.f; }

    {
    for (int i = tid; i < Cin; i += 512) { adl[i] = p.a[(int64_t)b * Cin + i]; adl[ADN + i] = p.d[(int64_t)b * Cin + i]; }

    // B fragments: pack order [slice][step = (j * 3 + dz) * 3 + dy][Cout/32][plane][lane]; wave w fetches piece w of every step (wave-uniform
    // base in SGPRs + one constant per-lane offset register: no 64-bit VALU arithmetic per piece).  Step (G + 2, st) is issued during step
    // (G, st) -- BETWEEN that step's MFMAs, so that the ~60-100 cycles a piece costs to issue are spent in the matrix pipe's shadow and not,
    // by both waves of a SIMD at once, right behind the hand-over barrier
    const int64_t bstep = (int64_t)(p.Cout / 32) * P * 1024;
    const unsigned char *bgs = reinterpret_cast<const unsigned char *>(p.wp) + (int64_t)b * p.wp_bstride + (int64_t)cb * STEPB;   // (uniform)
    const unsigned bvoff = (unsigned)(wave * 1024 + lane * 16);
#define WN_ISSUE_PIECE(SLOTI, ST)                                                                                              \
    do {                                                                                                                       \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(bvoff), "s"(bgs),                    \
                     "s"(lds_ring + (SLOTI) * GB + (ST) * STEPB + wave * 1024) : "memory");                                    \
        bgs += bstep;                                                                                                          \
    } while (0)
#pragma unroll
    for (int i = 0; i < 6; ++i) WN_ISSUE_PIECE(i / 3, i % 3);             // groups 0 and 1

    // ---- staging.  thread = (halo row hz * 10 + hy, x half xh, channel quad): voxels x0 + 4 xh - 1 .. + 4 of that row -> pairs 2 xh, 2 xh + 1
    // lane bits: [1:0] channel quad, [3:2] row + 0 / 2 / 4 / 6, [4] row + 1, [5] x half -- the 16 lanes of a ds_write_b64 service group then cover
    // the 32 store banks once (row pitch 272 B = 4 banks mod 32: rows R, R+2, R+4, R+6 sit 8 banks apart; the two x halves of a row are
    // exactly 32 banks apart and would collide: they are in different groups).  The (tid >> 3, (tid >> 2) & 1) order was 4-way conflicted
    const int srow = (tid >> 6) * 8 + 2 * ((tid >> 2) & 3) + ((tid >> 4) & 1), xh = (tid >> 5) & 1, c4 = (tid & 3) * 4;
    // (threads 480 .. 511 have no row of their own: they repeat row 59's work -- the same values to the same addresses -- instead of
    //  branching around it: a divergent branch inside the unrolled MFMA stream cuts it into basic blocks)
    const int rr = srow < WL::HZ * WL::HY ? srow : WL::HZ * WL::HY - 1;
    const int hz = rr / WL::HY, hy = rr - hz * WL::HY;
    const int wrow = rr * WL::ROWP + (2 * xh) * WL::VB + c4 * 2;          // byte offset of (pair 2 xh, plane 0, this quad) inside a slot
    // byte offsets inside the sample (< 2^32: checked by the launcher) of voxel k = 1 (x0 + 4 xh: inside the volume whenever the row is), of
    // k = 0 and of k = 5 (the only two that can fall off the row's ends: they then re-read k = 1 and are masked); a row outside the volume
    // reads the sample's first voxels
    unsigned voff1, voff0, voff5, inb = 0;
    {
        const int gz = z0 + hz - 1, gy = y0 + hy - 1;
        const bool rowin = gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;
        const int gx1 = x0 + 4 * xh;
        const unsigned vs = (unsigned)p.C0 * 4u;
        voff1 = rowin ? ((unsigned)((gz * p.H + gy) * p.W + gx1) * (unsigned)p.C0 + (unsigned)c4) * 4u : (unsigned)c4 * 4u;
        const bool in0 = rowin && gx1 - 1 >= 0, in5 = rowin && gx1 + 4 < p.W;
        voff0 = in0 ? voff1 - vs : voff1;
        voff5 = in5 ? voff1 + 4u * vs : voff1;
        inb = rowin ? (0x1eu | (in0 ? 1u : 0u) | (in5 ? 0x20u : 0u)) : 0u;
    }
    const float *const base0 = p.src0 + (int64_t)b * p.D * p.H * p.W * p.C0;
    f32x4n raw[NIT];
    auto issue_rows = [&](int sl) {
        const unsigned cb4 = (unsigned)sl * (SP_KS * 4u), vs = (unsigned)p.C0 * 4u;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const unsigned vo = (k == 0 ? voff0 : k == 5 ? voff5 : voff1 + (unsigned)(k - 1) * vs) + cb4;
            asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw[k]) : "v"(vo), "s"(base0) : "memory");
        }
    };
    // the staging affine (zero padding comes AFTER it), in place; the loads above are invisible to hipcc's waitcnt pass: pin the first use here
    auto affine_rows = [&](int sl) {
        const float4 av = *reinterpret_cast<const float4 *>(adl + sl * SP_KS + c4);
        const float4 dv = *reinterpret_cast<const float4 *>(adl + ADN + sl * SP_KS + c4);
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            asm volatile("" : "+v"(raw[k]));
            const bool in = (inb >> k) & 1u;
            raw[k].x = in ? __fmaf_rn(raw[k].x, av.x, dv.x) : 0.f;
            raw[k].y = in ? __fmaf_rn(raw[k].y, av.y, dv.y) : 0.f;
            raw[k].z = in ? __fmaf_rn(raw[k].z, av.z, dv.z) : 0.f;
            raw[k].w = in ? __fmaf_rn(raw[k].w, av.w, dv.w) : 0.f;
        }
    };
    // transform position jp of both pairs -> slot `slot` (fp32 differences / sums, then the exact two-plane split)
    auto convert = [&](int jp, int slot) {
        f32x4n va, vb;
        if (jp == 0) { va = raw[0] - raw[2]; vb = raw[2] - raw[4]; }
        else if (jp == 1) { va = raw[1] + raw[2]; vb = raw[3] + raw[4]; }
        else if (jp == 2) { va = raw[2] - raw[1]; vb = raw[4] - raw[3]; }
        else { va = raw[1] - raw[3]; vb = raw[3] - raw[5]; }
        uint2 pa[P], pb[P];
        split4<P, F16>(va.x, va.y, va.z, va.w, pa);
        split4<P, F16>(vb.x, vb.y, vb.z, vb.w, pb);
        unsigned char *dst = smem + slot * WL::SLOT + wrow;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            *reinterpret_cast<uint2 *>(dst + i * 32) = pa[i];
            *reinterpret_cast<uint2 *>(dst + WL::VB + i * 32) = pb[i];
        }
    };

    // slice 0 synchronously into slots 0..3
    issue_rows(0);
    GN_WAIT_VM_LGKM0(0);
    __syncthreads();                                // a / d table visible; groups 0, 1 of the ring have landed
    affine_rows(0);
#pragma unroll
    for (int j = 0; j < 4; ++j) convert(j, j);
    __syncthreads();

    // A fragment of step (j, dz, dy): rows (y = r >> 2, pair = r & 3) of halo row (zs + dz, y + dy) in slot(j)
    const int abase = (zs * WL::HY + (r >> 2)) * WL::ROWP + (r & 3) * WL::VB + 16 * h;
    const unsigned char *const ring_rd = smem + HALO_BYTES + cg * (NT * P * 1024) + lane * 16;
    // two fragment register sets (A: 2 planes, B: 2 column fragments x 2 planes = 24 registers each).  Group g multiplies step 0 from set
    // g & 1 (read at the end of the previous group), step 1 from the other set (read at the hand-over), step 2 from set g & 1 again (read
    // behind step 0's MFMAs); the next group's step 0 goes into the other set behind step 1's MFMAs.  12 groups per slice: the parity is static.
    uint4 fa[2][P], fb[2][NT][P];
#define WN_READ(SET, SLOT_OFF, HROW, RING_OFF)                                                                                 \
    do {                                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < P; ++i)                                                                          \
            fa[SET][i] = *reinterpret_cast<const uint4 *>(smem + (SLOT_OFF) + abase + (HROW) * WL::ROWP + i * 32);              \
        _Pragma("unroll") for (int u = 0; u < NT; ++u)                                                                         \
            _Pragma("unroll") for (int i = 0; i < P; ++i)                                                                      \
                fb[SET][u][i] = *reinterpret_cast<const uint4 *>(ring_rd + (RING_OFF) + (u * P + i) * 1024);                    \
    } while (0)
    // smallest terms first, the two accumulators alternating; WN_PROD_A: the first product pair, WN_PROD_B: the other two
#define WN_PROD_A(SET, AC)                                                                                                     \
    do { AC[0] = mfma16<F16>(fa[SET][1], fb[SET][0][0], AC[0]); AC[1] = mfma16<F16>(fa[SET][1], fb[SET][1][0], AC[1]); } while (0)
#define WN_PROD_B(SET, AC)                                                                                                     \
    do {                                                                                                                       \
        AC[0] = mfma16<F16>(fa[SET][0], fb[SET][0][1], AC[0]); AC[1] = mfma16<F16>(fa[SET][0], fb[SET][1][1], AC[1]);           \
        AC[0] = mfma16<F16>(fa[SET][0], fb[SET][0][0], AC[0]); AC[1] = mfma16<F16>(fa[SET][0], fb[SET][1][0], AC[1]);           \
    } while (0)
    // output transform of transform position J: out[x] = m0 + m1 + m2, out[x+1] = m1 - m2 - m3
#define WN_FLUSH(J, AC)                                                                                                        \
    do {                                                                                                                       \
        _Pragma("unroll") for (int u = 0; u < NT; ++u)                                                                         \
            _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                                   \
                const float m = AC[u][q];                                                                                      \
                if ((J) <= 2) tot[0][u][q] = __fadd_rn(tot[0][u][q], m);                                                       \
                if ((J) == 1) tot[1][u][q] = __fadd_rn(tot[1][u][q], m);                                                       \
                if ((J) >= 2) tot[1][u][q] = __fsub_rn(tot[1][u][q], m);                                                       \
            }                                                                                                                  \
    } while (0)
    WN_READ(0, 0, 0, 0);

    int sbase = 0;                                  // (4 s) % 5: slot of this slice's j = 0
    for (int s = 0; s < nslices; ++s) {
        const int sn = s + 1 < nslices ? s + 1 : s;
        const int nbase = sbase == 0 ? 4 : sbase - 1;                          // (4 (s + 1)) % 5
        int slo[4], nslo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int a_ = sbase + j, n_ = nbase + j;
            slo[j] = (a_ >= WL::NSLOT ? a_ - WL::NSLOT : a_) * WL::SLOT;
            nslo[j] = n_ >= WL::NSLOT ? n_ - WL::NSLOT : n_;
        }
#pragma unroll
        for (int g = 0; g < 12; ++g) {
            const int j = g / 3, dz = g % 3, X = g & 1, Y = X ^ 1;
            // hand-over of group g.  Must have landed: steps 1, 2 of this group (issued two groups ago) and step 0 of the next one (issued
            // during the previous group's step 0; it is read at the end of this group).  VM queue, oldest first: ..., (g+1, 0), (g+1, 1),
            // (g+1, 2): the youngest two may stay in flight; at g == 1 also the NIT row loads issued at the very end of group 0.  Groups 3, 5, 8,
            // 11 follow a conversion group: their barrier also publishes this wave's halo stores (lgkmcnt(0); elsewhere the only LDS operations
            // in flight are the fragment reads issued a few instructions ago, which hipcc waits for right before the MFMA that consumes them)
            if (g == 1) GN_WAIT_VM_ONLY(2 + NIT);
            else if (g == 3 || g == 5 || g == 8 || g == 11) GN_WAIT_VM_LGKM0(2);
            else GN_WAIT_VM_ONLY(2);
            __builtin_amdgcn_s_barrier();
            WN_READ(Y, slo[j], dz * WL::HY + 1, (g % RING) * GB + STEPB);
            __builtin_amdgcn_sched_barrier(0);
            WN_PROD_A(X, acc);
            WN_ISSUE_PIECE((g + 2) % RING, 0);      // group g+2 -> the slot group g-1 vacated
            WN_PROD_B(X, acc);
            // (conversions also behind the last slice, into slots nobody reads any more: no branch inside the MFMA stream)
            if (g == 2) { affine_rows(sn); convert(0, nslo[0]); }
            if (g == 4) convert(1, nslo[1]);
            if (g == 7) convert(2, nslo[2]);
            if (g == 10) convert(3, nslo[3]);
            WN_READ(X, slo[j], dz * WL::HY + 2, (g % RING) * GB + 2 * STEPB);
            __builtin_amdgcn_sched_barrier(0);
            WN_PROD_A(Y, acc);
            WN_ISSUE_PIECE((g + 2) % RING, 1);
            WN_PROD_B(Y, acc);
            {   // first step of the next group
                const int g1 = g + 1 < 12 ? g + 1 : 0;
                const int so = g + 1 < 12 ? slo[g1 / 3] : nslo[0] * WL::SLOT;
                WN_READ(Y, so, (g1 % 3) * WL::HY, ((g + 1) % RING) * GB);
            }
            __builtin_amdgcn_sched_barrier(0);
            WN_PROD_A(X, acc);
            WN_ISSUE_PIECE((g + 2) % RING, 2);
            WN_PROD_B(X, acc);
            if (g == 0) issue_rows(sn);             // always (uniform wait counts); unused after the last slice
            __builtin_amdgcn_sched_barrier(0);
            if (dz == 2) {                          // this transform position is complete: fold it into the totals, restart the accumulators
                WN_FLUSH(j, acc);
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[u][q] = 0.f;
            }
        }
        sbase = nbase;
    }
#undef WN_ISSUE_PIECE
#undef WN_FLUSH
#undef WN_PROD_A
#undef WN_PROD_B
#undef WN_READ
    GN_WAIT_VM_LGKM0(0);
    }
    __syncthreads();                                // look-ahead DMAs landed; the epilogue reuses the LDS as scratch
    // ---- epilogue.  D fragment element q of lane (h, r): pair row i = (q & 3) + 8 (q >> 2) + 4 h = (y = 2 (q >> 2) + h, pair = q & 3), channel r
    // (b, r, h laundered: everything the epilogue derives from them -- output / bias-table addresses, scale loads -- is computed HERE; hipcc
    //  otherwise hoists it above the MFMA loop as loop invariants and spills it there)
    int be = b, re = r, he = h;
    asm volatile("" : "+s"(be));
    asm volatile("" : "+v"(re), "+v"(he));
    const int gz = z0 + zs;
    double ssum[NT], ssq[NT];                       // fp64 per lane (see conv3d_split_kernel)
#pragma unroll
    for (int u = 0; u < NT; ++u) { ssum[u] = 0.0; ssq[u] = 0.0; }
    const bool interior = z0 > 0 && z0 + SP_TZ < p.D && y0 > 0 && y0 + SP_TY < p.H && x0 > 0 && x0 + SP_TX < p.W;   // no voxel of the tile on a face
    const int64_t rs2 = 2 * (int64_t)p.W * p.Cout;
    const int mz = sp_axis_mask(gz, p.D);
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int n = n0 + u * 32 + re;
            const float osn = p.out_scale[(int64_t)be * p.osc_bstride + n];
            const float osc = p.act_inv ? __fmul_rn(osn, p.act_inv[be]) : osn;
            const float *kb = p.kbias ? p.kbias + (int64_t)be * 64 * p.Cout + n : nullptr;
            const int gy = y0 + he, gx = x0 + e;
            float *ob = p.out + ((((int64_t)be * p.D + gz) * p.H + gy) * p.W + gx) * p.Cout + n;
            float kv[16];
            if (kb) {
                if (interior) {
                    const float k63 = kb[63 * (int64_t)p.Cout];
#pragma unroll
                    for (int q = 0; q < 16; ++q) kv[q] = k63;
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        kv[q] = kb[(int64_t)((mz * 4 + sp_axis_mask(gy + 2 * (q >> 2), p.H)) * 4 + sp_axis_mask(gx + 2 * (q & 3), p.W)) * p.Cout];
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float v = __fmul_rn(tot[e][u][q], osc);
                if (kb) v = __fadd_rn(v, kv[q]);
                if (p.relu) v = gn_relu(v);
                ob[(q >> 2) * rs2 + (int64_t)(2 * (q & 3)) * p.Cout] = v;
                ssum[u] += (double)v;
                ssq[u] += (double)v * (double)v;
            }
        }
    if (p.osum) {
        double *red = reinterpret_cast<double *>(smem);                         // [sum | sq][cg][zs][64]
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const double s2 = ssum[u] + __shfl_xor(ssum[u], 32), q2 = ssq[u] + __shfl_xor(ssq[u], 32);
            if (he == 0) { red[(cg * 4 + zs) * 64 + u * 32 + re] = s2; red[512 + (cg * 4 + zs) * 64 + u * 32 + re] = q2; }
        }
        __syncthreads();
        if (tid < 128) {
            const int g = tid >> 6, c = tid & 63;
            const double *rsum = red + g * 256 + c, *rsq = red + 512 + g * 256 + c;
            const double s4 = rsum[0] + rsum[64] + rsum[128] + rsum[192];
            const double q4 = rsq[0] + rsq[64] + rsq[128] + rsq[192];
            atomicAdd(&p.osum[(int64_t)be * p.Cout + cb * 128 + tid], s4);
            atomicAdd(&p.osq[(int64_t)be * p.Cout + cb * 128 + tid], q4);
        }
    }
}

// (called by conv3d_gcr_split_impl, unet_split.hip, which owns the shape checks and the occupancy-aware list / fill launches)
void gn_launch_conv3d_wino(const SplitArgs &p, int tiles, hipStream_t st) {
    hipLaunchKernelGGL((conv3d_split_wino_kernel<true>), dim3(tiles * (p.Cout / 128), p.B), dim3(512), 0, st, p);
}
