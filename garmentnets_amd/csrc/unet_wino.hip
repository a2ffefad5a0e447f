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
//  * CHAINS: a workgroup walks SplitArgs::chain tiles (one workgroup per CU: nothing else hides a tile's prologue and epilogue, measured at
//    12.5 % of the launch).  While the LAST slice of a tile is multiplied, what it stages "for the next slice" is the next tile's slice 0 (rows
//    addressed with the next tile's offsets, zero padding with its border mask) and the weight cursor wraps to the pack's start two groups
//    before the end -- the slot rotation, the ring and the fragment look-ahead run through the tile boundary unchanged, the epilogue's stores
//    drain under the next tile's first group.  Tiles of a chain are 32 items apart (the 32 workgroups an XCD runs at a time work on 32
//    neighbouring tiles: their halo overlap meets in that XCD's L2); the epilogue statistics are merged in LDS (ds_add_f64) and leave with
//    one set of atomics per chain.  A chain that crosses into another sample (or column block) drains and restarts there.
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
    constexpr int ST_OFF = AD_OFF + 2 * ADN * 4;    // fp64 statistics of the running chain: [sum | sumsq][128]
    constexpr int EC_OFF = ST_OFF + 2 * 128 * 8;    // epilogue constants of the running (sample, column block): [out scale | interior bias][128]
    __shared__ __attribute__((aligned(16))) unsigned char smem[EC_OFF + 2 * 128 * 4];
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + HALO_BYTES;
    float *const adl = reinterpret_cast<float *>(smem + AD_OFF);          // a[Cin] | d[Cin] of this sample
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), zs = wave & 3, cg = wave >> 2;
    const int Cin = p.C0;
    const int ncb = p.Cout / 128;
    const int tiles_z = p.D / SP_TZ;
    const int tps = tiles_z * p.tiles_x * p.tiles_y;
    const int nslices = Cin / SP_KS;
    double *const stl = reinterpret_cast<double *>(smem + ST_OFF);         // [sum | sumsq][128] of this run's tiles
    float *const ecl = reinterpret_cast<float *>(smem + EC_OFF);

    // ---- work items.  item = (tile list position) * ncb + column block over the whole launch (dense: every tile of every sample, sample-major;
    // occupancy-aware: the active list).  Chains in XCD-aware order (workgroup i runs on XCD i % 8; every XCD owns a contiguous range of chains);
    // 32 consecutive chains interleave over 32 * chain consecutive items
    const unsigned n_items = (p.active_list ? (unsigned)(*p.active_count) : (unsigned)(p.B * tps)) * (unsigned)ncb;
    const unsigned span = 32u * (unsigned)p.chain;
    const unsigned nch = (n_items + span - 1u) / span * 32u;
    if (blockIdx.x >= nch) return;                                         // (workgroup-uniform)
    const unsigned chn = (blockIdx.x & 7u) * (nch >> 3) + (blockIdx.x >> 3);
    int item = (int)((chn >> 5) * span + (chn & 31u));
    const int item_end = (int)(((chn >> 5) + 1u) * span < n_items ? ((chn >> 5) + 1u) * span : n_items);
    if (item >= item_end) return;
    auto decode = [&](int it, int &b_, int &cb_, int &z0_, int &y0_, int &x0_) {
        const int t = it / ncb;
        cb_ = it - t * ncb;
        const int e = p.active_list ? p.active_list[t] : t;
        b_ = e / tps;
        int tile = e - b_ * tps;
        const int tz = tile % tiles_z; tile /= tiles_z;
        const int tx = tile % p.tiles_x;
        z0_ = tz * SP_TZ; y0_ = (tile / p.tiles_x) * SP_TY; x0_ = tx * SP_TX;
    };
    int b, cb, z0, y0, x0;
    decode(item, b, cb, z0, y0, x0);

    // acc: the running transform position's accumulators; tot[e]: outputs at even (e = 0) / odd x of the pairs.  (A second accumulator set -- the
    // output transform of position j folded in under position j+1's MFMAs -- needs 261 registers: 46 spilled values inside the MFMA stream.)
    f32x16s acc[NT], tot[2][NT];

    // B fragments: pack order [slice][step = (j * 3 + dz) * 3 + dy][Cout/32][plane][lane]; wave w fetches piece w of every step (wave-uniform
    // base in SGPRs + one constant per-lane offset register: no 64-bit VALU arithmetic per piece).  Step (G + 2, st) is issued during step
    // (G, st) -- BETWEEN that step's MFMAs, so that the ~60-100 cycles a piece costs to issue are spent in the matrix pipe's shadow and not,
    // by both waves of a SIMD at once, right behind the hand-over barrier
    const int64_t bstep = (int64_t)(p.Cout / 32) * P * 1024;
    const unsigned char *bgs = nullptr;                                    // (uniform) the weight cursor
    const unsigned bvoff = (unsigned)(wave * 1024 + lane * 16);
#define WN_ISSUE_PIECE(SLOTI, ST)                                                                                              \
    do {                                                                                                                       \
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(bvoff), "s"(bgs),                    \
                     "s"(lds_ring + (SLOTI) * GB + (ST) * STEPB + wave * 1024) : "memory");                                    \
        bgs += bstep;                                                                                                          \
    } while (0)

    // ---- staging.  thread = (halo row hz * 10 + hy, x half xh, channel quad): voxels x0 + 4 xh - 1 .. + 4 of that row -> pairs 2 xh, 2 xh + 1
    // lane bits: [1:0] channel quad, [3:2] row + 0 / 2 / 4 / 6, [4] row + 1, [5] x half -- the 16 lanes of a ds_write_b64 service group then cover
    // the 32 store banks once (row pitch 272 B = 4 banks mod 32: rows R, R+2, R+4, R+6 sit 8 banks apart; the two x halves of a row are
    // exactly 32 banks apart and would collide: they are in different groups).  The (tid >> 3, (tid >> 2) & 1) order was 4-way conflicted
    // (threads 480 .. 511 have no row of their own: they repeat row 59's work -- the same values to the same addresses -- instead of
    //  branching around it: a divergent branch inside the unrolled MFMA stream cuts it into basic blocks)
    auto stage_row = [&](int t) { const int srow = (t >> 6) * 8 + 2 * ((t >> 2) & 3) + ((t >> 4) & 1); return srow < WL::HZ * WL::HY ? srow : WL::HZ * WL::HY - 1; };
    const int c4 = (tid & 3) * 4;
    const int wrow = stage_row(tid) * WL::ROWP + (2 * ((tid >> 5) & 1)) * WL::VB + c4 * 2;   // byte offset of (pair 2 xh, plane 0, this quad) inside a slot
    // byte offsets inside the sample (< 2^32: checked by the launcher) of voxel k = 1 (x0 + 4 xh: inside the volume whenever the row is), of
    // k = 0 and of k = 5 (the only two that can fall off the row's ends: they then re-read k = 1 and are masked); a row outside the volume
    // reads the sample's first voxels.  (Everything but the results is re-derived from a laundered thread id at every call: lane constants
    // that lived from the prologue to the next tile's set_rows were 19 spilled registers.)
    unsigned voff1 = 0, voff0 = 0, voff5 = 0, inb = 0;
    auto set_rows = [&](int z0_, int y0_, int x0_) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const int rr = stage_row(t), hz = rr / WL::HY, hy = rr - hz * WL::HY, xh = (t >> 5) & 1, cq = (t & 3) * 4;
        const int gz = z0_ + hz - 1, gy = y0_ + hy - 1;
        const bool rowin = gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;
        const int gx1 = x0_ + 4 * xh;
        const unsigned vs = (unsigned)p.C0 * 4u;
        voff1 = rowin ? ((unsigned)((gz * p.H + gy) * p.W + gx1) * (unsigned)p.C0 + (unsigned)cq) * 4u : (unsigned)cq * 4u;
        const bool in0 = rowin && gx1 - 1 >= 0, in5 = rowin && gx1 + 4 < p.W;
        voff0 = in0 ? voff1 - vs : voff1;
        voff5 = in5 ? voff1 + 4u * vs : voff1;
        inb = rowin ? (0x1eu | (in0 ? 1u : 0u) | (in5 ? 0x20u : 0u)) : 0u;
    };
    const float *base0 = p.src0;
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
    float4 afa, afd;
    auto affine_load = [&](int sl) {
        afa = *reinterpret_cast<const float4 *>(adl + sl * SP_KS + c4);
        afd = *reinterpret_cast<const float4 *>(adl + ADN + sl * SP_KS + c4);
    };
    auto affine_math = [&](int k0, int k1) {
#pragma unroll
        for (int k = k0; k < k1; ++k) {
            asm volatile("" : "+v"(raw[k]));
            const bool in = (inb >> k) & 1u;
            raw[k].x = in ? __fmaf_rn(raw[k].x, afa.x, afd.x) : 0.f;
            raw[k].y = in ? __fmaf_rn(raw[k].y, afa.y, afd.y) : 0.f;
            raw[k].z = in ? __fmaf_rn(raw[k].z, afa.z, afd.z) : 0.f;
            raw[k].w = in ? __fmaf_rn(raw[k].w, afa.w, afd.w) : 0.f;
        }
    };
    auto affine_rows = [&](int sl) { affine_load(sl); affine_math(0, NIT); };
    // transform position jp of both pairs -> slot `slot` (fp32 differences / sums, then the exact two-plane split), in four stages the group
    // loop threads between its MFMAs: conv_a the sums, conv_b / conv_c the splits, conv_d the stores
    f32x4n cva, cvb;
    uint2 cpa[P], cpb[P];
    auto conv_a = [&](int jp) {
        if (jp == 0) { cva = raw[0] - raw[2]; cvb = raw[2] - raw[4]; }
        else if (jp == 1) { cva = raw[1] + raw[2]; cvb = raw[3] + raw[4]; }
        else if (jp == 2) { cva = raw[2] - raw[1]; cvb = raw[4] - raw[3]; }
        else { cva = raw[1] - raw[3]; cvb = raw[3] - raw[5]; }
    };
    auto conv_b = [&]() { split4<P, F16>(cva.x, cva.y, cva.z, cva.w, cpa); };
    auto conv_c = [&]() { split4<P, F16>(cvb.x, cvb.y, cvb.z, cvb.w, cpb); };
    auto conv_d = [&](int slot) {
        unsigned char *dst = smem + slot * WL::SLOT + wrow;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            *reinterpret_cast<uint2 *>(dst + i * 32) = cpa[i];
            *reinterpret_cast<uint2 *>(dst + WL::VB + i * 32) = cpb[i];
        }
    };
    auto convert = [&](int jp, int slot) { conv_a(jp); conv_b(); conv_c(); conv_d(slot); };

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
#define WN_M(SET, IA, U, IB, AC) AC[U] = mfma16<F16>(fa[SET][IA], fb[SET][U][IB], AC[U])
#define WN_FENCE() __builtin_amdgcn_sched_barrier(0)
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

    bool fresh = true;
    int sbase = 0;                                  // slot of the running slice's j = 0: four further per slice (mod 5), through tile boundaries
    for (;;) {                                      // the tiles of this chain
        if (fresh) {
            // ---- a run starts (the chain's first tile, or the first one of another sample / column block): everything synchronously
            GN_WAIT_VM_LGKM0(0);
            __syncthreads();                        // (restart: the previous run's look-ahead has landed, its statistics have been read)
            {   // (thread id laundered: the LDS addresses derived from it are otherwise hoisted above the tile loop and spilled there)
                int tf = threadIdx.x;
                asm volatile("" : "+v"(tf));
                if (tf < Cin) { adl[tf] = p.a[(int64_t)b * Cin + tf]; adl[ADN + tf] = p.d[(int64_t)b * Cin + tf]; }     // (Cin <= ADN < 512)
                if (tf < 256) stl[tf] = 0.0;
                if (tf < 128) {                     // the epilogue's per-channel constants (a global load there is a round trip behind the stores in the queue)
                    const float osn = p.out_scale[(int64_t)b * p.osc_bstride + cb * 128 + tf];
                    ecl[tf] = p.act_inv ? __fmul_rn(osn, p.act_inv[b]) : osn;
                    ecl[128 + tf] = p.kbias ? p.kbias[((int64_t)b * 64 + 63) * p.Cout + cb * 128 + tf] : 0.f;
                }
            }
            bgs = reinterpret_cast<const unsigned char *>(p.wp) + (int64_t)b * p.wp_bstride + (int64_t)cb * STEPB;
#pragma unroll
            for (int i = 0; i < 6; ++i) WN_ISSUE_PIECE(i / 3, i % 3);     // groups 0 and 1
            base0 = p.src0 + (int64_t)b * p.D * p.H * p.W * p.C0;
            set_rows(z0, y0, x0);
            issue_rows(0);
            GN_WAIT_VM_LGKM0(0);
            __syncthreads();                        // a / d table visible; groups 0, 1 of the ring have landed
            affine_rows(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) convert(j, j);
            __syncthreads();
            sbase = 0;
            WN_READ(0, 0, 0, 0);
            fresh = false;
        }
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc[u][q] = 0.f; tot[0][u][q] = 0.f; tot[1][u][q] = 0.f; }
        // the chain's next tile; `cont`: it continues this run (same sample and column block: same weights, same affine table, same statistics)
        const int nitem = item + 32;
        const bool more = nitem < item_end;
        int bn = b, cbn = cb, z0n = z0, y0n = y0, x0n = x0;
        if (more) decode(nitem, bn, cbn, z0n, y0n, x0n);
        const bool cont = more && bn == b && cbn == cb;

        for (int s = 0; s < nslices; ++s) {
            const bool last = s + 1 == nslices;
            // what this slice stages: the tile's next slice -- or, behind the last one, slice 0 of the next tile (not `cont`: slots nobody
            // reads any more and the pack's zero pad steps; no branch inside the MFMA stream)
            const int sn = last ? (cont ? 0 : s) : s + 1;         // (nothing follows: this slice's own rows again -- they sit in L2)
            if (last && cont) set_rows(z0n, y0n, x0n);
            const int64_t wrap = last && cont ? -(int64_t)nslices * 36 * bstep : 0;       // the weight cursor returns to the pack's start at group 10
            const int nbase = sbase == 0 ? 4 : sbase - 1;                          // (sbase + 4) % 5
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
                // in flight are the fragment reads issued a few instructions ago, which hipcc waits for right before the MFMA that consumes them).
                // Group 0 of a tile's FIRST slice has had its hand-over already: behind the prologue's barriers (a run's first tile), or at the
                // barrier in front of the previous tile's epilogue (below) -- here it would wait for that epilogue's stores to be acknowledged
                if (g == 1) GN_WAIT_VM_ONLY(2 + NIT);
                else if (g == 3 || g == 5 || g == 8 || g == 11) GN_WAIT_VM_LGKM0(2);
                else if (g > 0 || s > 0) GN_WAIT_VM_ONLY(2);
                if (g > 0 || s > 0) __builtin_amdgcn_s_barrier();
                WN_READ(Y, slo[j], dz * WL::HY + 1, (g % RING) * GB + STEPB);
                __builtin_amdgcn_sched_barrier(0);
                if (g == 10) bgs += wrap;
                // Conversion groups (2: + the staging affine): the conversion's VALU stages threaded between this step's six MFMAs by hand, fenced
                // (hipcc's own order flips between builds -- threaded, or dumped behind the MFMAs, 3 - 4 % of the kernel apart -- and
                // sched_group_barrier pipelines took on one of the four groups only)
                if (g == 2 || g == 4 || g == 7 || g == 10) {
                    const int jn = g == 2 ? 0 : g == 4 ? 1 : g == 7 ? 2 : 3;
                    if (g == 2) affine_load(sn);
                    WN_M(X, 1, 0, 0, acc);
                    WN_ISSUE_PIECE((g + 2) % RING, 0);
                    WN_M(X, 1, 1, 0, acc);
                    WN_FENCE();
                    if (g == 2) { affine_math(0, 3); WN_FENCE(); }
                    else { conv_a(jn); WN_FENCE(); }
                    WN_M(X, 0, 0, 1, acc);
                    WN_FENCE();
                    if (g == 2) { affine_math(3, NIT); WN_FENCE(); }
                    else { conv_b(); WN_FENCE(); }
                    WN_M(X, 0, 1, 1, acc);
                    WN_FENCE();
                    if (g == 2) { conv_a(jn); conv_b(); WN_FENCE(); }
                    else { conv_c(); WN_FENCE(); }
                    WN_M(X, 0, 0, 0, acc);
                    WN_FENCE();
                    if (g == 2) { conv_c(); }
                    conv_d(nslo[jn]);
                    WN_FENCE();
                    WN_M(X, 0, 1, 0, acc);
                } else {
                    WN_PROD_A(X, acc);
                    WN_ISSUE_PIECE((g + 2) % RING, 0);      // group g+2 -> the slot group g-1 vacated
                    WN_PROD_B(X, acc);
                }
                WN_READ(X, slo[j], dz * WL::HY + 2, (g % RING) * GB + 2 * STEPB);
                if (!(g == 2 || g == 4 || g == 7 || g == 10)) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                WN_PROD_A(Y, acc);
                WN_ISSUE_PIECE((g + 2) % RING, 1);
                WN_PROD_B(Y, acc);
                {   // first step of the next group
                    const int g1 = g + 1 < 12 ? g + 1 : 0;
                    const int so = g + 1 < 12 ? slo[g1 / 3] : nslo[0] * WL::SLOT;
                    WN_READ(Y, so, (g1 % 3) * WL::HY, ((g + 1) % RING) * GB);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_barrier(0);
                WN_PROD_A(X, acc);
                WN_ISSUE_PIECE((g + 2) % RING, 2);
                WN_PROD_B(X, acc);
                if (g == 0) issue_rows(sn);             // always (uniform wait counts)
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
        // the next tile's group-0 hand-over, in front of the epilogue: steps 1, 2 of its group 0 and step 0 of its group 1 have landed (the
        // youngest two pieces stay in flight), every wave is through with this tile's last ring slot
        GN_WAIT_VM_ONLY(2);
        __builtin_amdgcn_s_barrier();

        // ---- epilogue of this tile.  D fragment element q of lane (h, r): pair row i = (q & 3) + 8 (q >> 2) + 4 h = (y = 2 (q >> 2) + h, pair = q & 3),
        // channel r.  (Everything the epilogue derives from the tile / lane coordinates -- output / bias-table addresses, scale loads -- is computed
        // HERE, from laundered copies: hipcc otherwise hoists it above the slice loop as loop invariants and spills it there.)  No LDS scratch and
        // no barrier: the halo slots and the ring already hold the next tile's operands
        {
            int be = b, cbe = cb, z0e = z0, y0e = y0, x0e = x0, te = threadIdx.x;
            asm volatile("" : "+s"(be), "+s"(cbe), "+s"(z0e), "+s"(y0e), "+s"(x0e));
            asm volatile("" : "+v"(te));
            const int re = te & 31, he = (te >> 5) & 1;
            const int n0 = cbe * 128 + cg * 64;
            const int gz = z0e + zs;
            double ssum[NT], ssq[NT];                   // fp64 per lane (see conv3d_split_kernel)
            float osc[NT], k63[NT];
#pragma unroll
            for (int u = 0; u < NT; ++u) { ssum[u] = 0.0; ssq[u] = 0.0; osc[u] = ecl[cg * 64 + u * 32 + re]; k63[u] = ecl[128 + cg * 64 + u * 32 + re]; }
            const bool interior = z0e > 0 && z0e + SP_TZ < p.D && y0e > 0 && y0e + SP_TY < p.H && x0e > 0 && x0e + SP_TX < p.W;   // no voxel of the tile on a face
            const bool classes = p.kbias && !interior;
            const int mz = sp_axis_mask(gz, p.D);
            // stores: wave-uniform 64-bit base of output row pair (q >> 2) in SGPRs + a 32-bit lane offset per (e, u, q & 3): no 64-bit VALU
            // arithmetic, nothing for hipcc to hoist.  Non-temporal: the layer's output is read once, by the next launch, and is far larger than
            // L2 + MALL; with the hint the 64 stores per lane are acknowledged sooner (they sit in the in-order VM queue in front of the next
            // tile's weight pieces): 2 - 3 % of the launch
            const float *const orow = p.out + ((((int64_t)be * p.D + gz) * p.H + y0e) * p.W + x0e) * p.Cout + n0;
            const int64_t rs2 = 2 * (int64_t)p.W * p.Cout;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int gy = y0e + he, gx = x0e + e;
                float kv[NT][16];
                if (classes) {                          // a tile on a face: the per-class constants, all of this e's loads in flight together
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        const float *kb = p.kbias + (int64_t)be * 64 * p.Cout + n0 + u * 32 + re;
#pragma unroll
                        for (int q = 0; q < 16; ++q)
                            kv[u][q] = kb[(int64_t)((mz * 4 + sp_axis_mask(gy + 2 * (q >> 2), p.H)) * 4 + sp_axis_mask(gx + 2 * (q & 3), p.W)) * p.Cout];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < NT; ++u)
#pragma unroll
                        for (int q = 0; q < 16; ++q) kv[u][q] = k63[u];
                }
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    unsigned vo[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) vo[i] = (unsigned)(((he * p.W + e + 2 * i) * p.Cout + u * 32 + re) * 4);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        float v = __fmul_rn(tot[e][u][q], osc[u]);
                        if (p.kbias) v = __fadd_rn(v, kv[u][q]);
                        if (p.relu) v = gn_relu(v);
                        const float *ob = orow + (q >> 2) * rs2;       // (uniform)
                        asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(vo[q & 3]), "v"(v), "s"(ob) : "memory");
                        ssum[u] += (double)v;
                        ssq[u] += (double)v * (double)v;
                    }
                }
            }
            if (p.osum) {
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    const double s2 = ssum[u] + __shfl_xor(ssum[u], 32), q2 = ssq[u] + __shfl_xor(ssq[u], 32);
                    if (he == 0) {
                        const unsigned sa = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + ST_OFF + (cg * 64 + u * 32 + re) * 8;
                        asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:1024" ::"v"(sa), "v"(s2), "v"(q2) : "memory");
                    }
                }
            }
        }
        if (!cont) {
            // the run's statistics leave: one set of atomics per (run, channel)
            if (p.osum) {
                GN_WAIT_VM_LGKM0(63);
                __syncthreads();
                if (tid < 128) {
                    atomicAdd(&p.osum[(int64_t)b * p.Cout + cb * 128 + tid], stl[tid]);
                    atomicAdd(&p.osq[(int64_t)b * p.Cout + cb * 128 + tid], stl[128 + tid]);
                }
            }
            if (!more) break;
            fresh = true;
        }
        item = nitem; b = bn; cb = cbn; z0 = z0n; y0 = y0n; x0 = x0n;
    }
#undef WN_ISSUE_PIECE
#undef WN_M
#undef WN_FENCE
#undef WN_FLUSH
#undef WN_PROD_A
#undef WN_PROD_B
#undef WN_READ
    GN_WAIT_VM_LGKM0(0);                            // (the look-ahead DMAs of the last tile land in this workgroup's LDS: not past its end)
}

// (called by conv3d_gcr_split_impl, unet_split.hip, which owns the shape checks and the occupancy-aware list / fill launches)
void gn_launch_conv3d_wino(const SplitArgs &p0, int tiles, hipStream_t st) {
    SplitArgs p = p0;
    // chain length: 16 tiles hide 15 of 16 prologues / epilogues; shorter when that would leave fewer than ~4 chains per CU and sample.  From the SAMPLE's
    // tiles alone, never the batch size: the chain is the unit in which the fp64 epilogue statistics are grouped, and a garment's GroupNorm statistics
    // must not depend (not even in the last bit) on how many garments share its batch.  GARMENTNETS_WINO_CHAIN overrides (tests and
    // measurements: the outputs do not depend on it)
    const int64_t per_sample = (int64_t)tiles * (p.Cout / 128);
    int chain = (int)(per_sample / 64);
    chain = chain < 1 ? 1 : chain > 16 ? 16 : chain;
    // (looked up per launch, ~100 ns against a multi-millisecond kernel: tests/test_gpu_parity.py varies it inside one process)
    if (const char *e = getenv("GARMENTNETS_WINO_CHAIN")) { const int forced = atoi(e); if (forced > 0 && forced <= 4096) chain = forced; }
    const int64_t items = per_sample * p.B;                                    // (occupancy-aware: the dense bound; chains past the list's end return)
    p.chain = chain;
    const int64_t span = 32 * (int64_t)chain;
    hipLaunchKernelGGL((conv3d_split_wino_kernel<true>), dim3((unsigned)((items + span - 1) / span * 32)), dim3(512), 0, st, p);
}
