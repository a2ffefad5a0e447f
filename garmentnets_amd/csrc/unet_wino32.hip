// unet_wino32.hip -- the 32-wide column-block split-operand 'gcr' convolution with Winograd F(2,3) along x (round 6).
//
// Reference layers: components/unet3d.py:127-144 (the encoder's second convolution: 128 -> 32 at full resolution), :291,330 (the last decoder's
// two convolutions at full resolution), and the 32- / 64-wide layers one level down.  Rounds 2 - 5 ran them through conv3d_split_strip_kernel
// (unet_split.hip): the direct 27-tap form, 3 x 54 fp16 matrix-core products per fp32 output and input channel, at the socket's power limit --
// the only lever left is FEWER products per result.  This kernel is the F(2,3)-along-x form of unet_wino.hip (same algebra, same
// weight pack [Cin/16][36 steps = (j * 3 + dz) * 3 + dy][Cout/32][plane][64 lanes] x 16 B, same fp32 input transform BEFORE the exact two-plane
// split, same fp32 output transform) re-tiled for ONE 32-wide column fragment per wave:
//  * 512 threads = 8 waves, one workgroup per CU, tile 8 x 8 x 8 voxels x 32 output channels: wave w owns z-slice w; its 32 fragment rows are
//    the output PAIRS (y 0..7, pair 0..3) of that slice.  Per step (j, dz, dy): one A fragment pair, one B fragment pair, 3 MFMAs (the products
//    (1,0) (0,1) (0,0), smallest first, as everywhere).  One accumulator (16 registers) lives through the 9 steps of a transform position j and is
//    folded into the even / odd totals (32 registers): 48 accumulator registers where the x-strip kernel needs 128 -- what is left holds
//    the next slice's 10 halo voxels per thread in flight, so the staging never stops the matrix pipe.
//  * LDS halo: per 16-channel slice and transform position one SLOT of 10 x 10 rows x 4 pairs x 64 B (row pitch 272 B: unet_wino.hip's
//    conflict-free pitch), FIVE slots rotating exactly as there (slice s+1's position j goes where slice s's j-1 has been multiplied; j = 0: the
//    spare): 136 000 B.  B ring: a GROUP = the three dy steps of one (j, dz) = 6 KB, FOUR groups deep; the group three ahead is DMA'd right
//    behind a hand-over by waves 6 and 7 (three 1 KB pieces each), the two waves with next to no halo row to stage.
//  * staging: thread = (halo row, channel quad): the row's 10 x-consecutive voxels (loaded with asm loads at the start of group 0, invisible to
//    hipcc's waitcnt pass) -> GroupNorm affine -> four pairs x four transform positions, converted position by position in the groups behind
//    the one that frees the target slot.  100 rows x 4 quads = 400 of the 512 threads have a row of their own; the others repeat row 99.
//  * CHAINS of tiles per workgroup and the epilogue without LDS, as unet_wino.hip: the last slice of a tile stages slice 0 of the chain's next
//    tile, the weight cursor wraps three groups before the end, the statistics leave once per run.
//  * epilogue extras the x-strip kernel has and the 128-wide Winograd kernel does not need: the polyphase partial of a decoder's first
//    convolution (SplitArgs::partial) is added before the ReLU.
#include "split_conv.h"

struct Wino32Layout {
    static constexpr int VB = 64;                      // bytes per transformed voxel: 2 planes x 16 halfs
    static constexpr int ROWP = 4 * VB + 16;           // 4 pairs per halo row + one 16-byte pad
    static constexpr int TZ = 8, HZ = TZ + 2, HY = SP_TY + 2, ROWS = HZ * HY;
    static constexpr int SLOT = ROWS * ROWP;           // one transform position of one slice: 27 200 B
    static constexpr int NSLOT = 5;
};

typedef float f32x4m __attribute__((ext_vector_type(4)));

template <bool F16>
__global__ __launch_bounds__(512, 1) void conv3d_split_wino32_kernel(SplitArgs p) {
    constexpr int P = 2;
    using WL = Wino32Layout;
    constexpr int STEPB = P * 1024;                 // B fragments of one step: 2 KB
    constexpr int GB = 3 * STEPB;                   // group = the three dy steps of one (j, dz)
    constexpr int RING = 4;
    constexpr int HALO_BYTES = WL::NSLOT * WL::SLOT;
    constexpr int AD_OFF = HALO_BYTES + RING * GB;
    constexpr int ADN = 128;                        // Cin <= 128 (checked by the launcher)
    constexpr int NIT = 10;                         // row loads per thread per slice
    constexpr int ST_OFF = AD_OFF + 2 * ADN * 4;    // fp64 statistics of the running chain: [sum | sumsq][32]
    constexpr int EC_OFF = ST_OFF + 2 * 32 * 8;     // epilogue constants of the running (sample, column block): [out scale | interior bias][32]
    __shared__ __attribute__((aligned(16))) unsigned char smem[EC_OFF + 2 * 32 * 4];
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + HALO_BYTES;
    float *const adl = reinterpret_cast<float *>(smem + AD_OFF);          // a[Cin] | d[Cin] of this sample
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), zs = wave;
    const int Cin = p.C0;
    const int ncb = p.Cout / 32;
    const int tiles_z = p.D / WL::TZ;
    const int tps = tiles_z * p.tiles_x * p.tiles_y;
    const int nslices = Cin / SP_KS;
    double *const stl = reinterpret_cast<double *>(smem + ST_OFF);
    float *const ecl = reinterpret_cast<float *>(smem + EC_OFF);

    // ---- work items and chains: as conv3d_split_wino_kernel (unet_wino.hip), at this kernel's tile granularity (8 x 8 x 8)
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
        z0_ = tz * WL::TZ; y0_ = (tile / p.tiles_x) * SP_TY; x0_ = tx * SP_TX;
    };
    int b, cb, z0, y0, x0;
    decode(item, b, cb, z0, y0, x0);

    f32x16s acc, tot[2];

    // ---- B fragments.  Pack order [slice][step = (j * 3 + dz) * 3 + dy][Cout/32][plane][lane]; a group = three consecutive steps = six 1 KB pieces
    // (piece i = step i >> 1, plane i & 1).  Waves 6 and 7 -- the two whose threads have (next to) no halo row to stage -- fetch three pieces each;
    // waves 0 - 5 fetch none: vector-memory results return IN ORDER per wave, so a weight piece issued behind a wave's ten row loads (HBM latency)
    // cannot land before them -- with the pieces on waves of their own the weight ring never waits for the halo (profiles/r06_ab_experiments.txt)
    const int64_t bstep = (int64_t)ncb * STEPB;
    const bool dma_wave = wave >= 6;
    const int pi0 = wave == 7 ? 3 : 0;
    const unsigned char *bgs = nullptr;                                    // (uniform) the weight cursor: start of the group issued next
    const unsigned bvoff = (unsigned)(lane * 16);
#define W32_ISSUE_GROUP(SLOTI)                                                                                                 \
    do {                                                                                                                       \
        if (dma_wave) {                                                                                                        \
            _Pragma("unroll") for (int i_ = 0; i_ < 3; ++i_) {                                                                 \
                const int pi_ = pi0 + i_;                                                                                      \
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(bvoff),                     \
                             "s"(bgs + (int64_t)(pi_ >> 1) * bstep + (pi_ & 1) * 1024),                                        \
                             "s"(lds_ring + (SLOTI) * GB + (unsigned)((pi_ >> 1) * STEPB + (pi_ & 1) * 1024)) : "memory");     \
            }                                                                                                                  \
        }                                                                                                                      \
        bgs += 3 * bstep;                                                                                                      \
    } while (0)
#define W32_WAIT_LGKM0() __builtin_amdgcn_s_waitcnt(0xC07F)                 /* lgkmcnt(0) alone */
    // hand-over waits.  A DMA wave's queue, oldest first: ..., P(g+1) x 3, P(g+2) x 3 (+ its NIT row loads, issued behind P(3) in group 0): all but the
    // youngest 3 (groups 1, 2: 3 + NIT) must have landed.  A staging wave's queue holds its row loads (and the previous tile's output stores) only:
    // they must have landed where the conversions start (group 3), nowhere else
#define W32_HANDOVER_WAIT(G)                                                                                                   \
    do {                                                                                                                       \
        if (dma_wave) { if ((G) == 1 || (G) == 2) GN_WAIT_VM_LGKM0(3 + NIT); else GN_WAIT_VM_LGKM0(3); }                        \
        else { if ((G) == 3) GN_WAIT_VM_LGKM0(0); else W32_WAIT_LGKM0(); }                                                     \
    } while (0)

    // ---- staging.  thread = (halo row hz * 10 + hy, channel quad): voxels x0 - 1 .. x0 + 8 of that row -> the row's four pairs.
    // lane bits: [1:0] channel quad, [3:2] row + 0 / 2 / 4 / 6, [4] row + 1, [5] row + 8: the 16 lanes of a ds_write_b64 service group cover the 32
    // store banks once (row pitch 272 B = 4 banks mod 32: rows R, R+2, R+4, R+6 sit 8 banks apart).  Threads without a row of their own (rows
    // >= 100: part of wave 6, all of wave 7) repeat row 99's work -- the same values to the same addresses -- instead of branching around it
    auto stage_row = [&](int t) {
        const int srow = (t >> 6) * 16 + 2 * ((t >> 2) & 3) + ((t >> 4) & 1) + 8 * ((t >> 5) & 1);
        return srow < WL::ROWS ? srow : WL::ROWS - 1;
    };
    const int c4 = (tid & 3) * 4;
    const int wrow = stage_row(tid) * WL::ROWP + c4 * 2;                   // byte offset of (pair 0, plane 0, this quad) inside a slot
    // byte offsets inside the sample (< 2^32: checked by the launcher) of voxel k = 1 (x0: inside the volume whenever the row is), of k = 0 and of
    // k = 9 (the only two that can fall off the row's ends: they then re-read k = 1 and are masked); a row outside the volume reads the sample's
    // first voxels
    unsigned voff1 = 0, voff0 = 0, voff9 = 0, inb = 0;
    auto set_rows = [&](int z0_, int y0_, int x0_) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const int rr = stage_row(t), hz = rr / WL::HY, hy = rr - hz * WL::HY, cq = (t & 3) * 4;
        const int gz = z0_ + hz - 1, gy = y0_ + hy - 1;
        const bool rowin = gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;
        const unsigned vs = (unsigned)p.C0 * 4u;
        voff1 = rowin ? ((unsigned)((gz * p.H + gy) * p.W + x0_) * (unsigned)p.C0 + (unsigned)cq) * 4u : (unsigned)cq * 4u;
        const bool in0 = rowin && x0_ - 1 >= 0, in9 = rowin && x0_ + 8 < p.W;
        voff0 = in0 ? voff1 - vs : voff1;
        voff9 = in9 ? voff1 + 8u * vs : voff1;
        inb = rowin ? (0x1feu | (in0 ? 1u : 0u) | (in9 ? 0x200u : 0u)) : 0u;
    };
    const float *base0 = p.src0;
    f32x4m raw[NIT];
    auto issue_rows = [&](int sl) {
        const unsigned cb4 = (unsigned)sl * (SP_KS * 4u), vs = (unsigned)p.C0 * 4u;
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const unsigned vo = (k == 0 ? voff0 : k == NIT - 1 ? voff9 : voff1 + (unsigned)(k - 1) * vs) + cb4;
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
    // transform position jp of pairs k0 .. k1-1 -> slot `slot`: fp32 differences / sums, the exact two-plane split, the stores
    auto convert = [&](int jp, int slot, int k0, int k1) {
        unsigned char *dst = smem + slot * WL::SLOT + wrow;
#pragma unroll
        for (int k = k0; k < k1; ++k) {
            f32x4m cv;
            if (jp == 0) cv = raw[2 * k] - raw[2 * k + 2];
            else if (jp == 1) cv = raw[2 * k + 1] + raw[2 * k + 2];
            else if (jp == 2) cv = raw[2 * k + 2] - raw[2 * k + 1];
            else cv = raw[2 * k + 1] - raw[2 * k + 3];
            uint2 cp[P];
            split4<P, F16>(cv.x, cv.y, cv.z, cv.w, cp);
#pragma unroll
            for (int i = 0; i < P; ++i) *reinterpret_cast<uint2 *>(dst + k * WL::VB + i * 32) = cp[i];
        }
    };

    // A fragment of step (j, dz, dy): rows (y = r >> 2, pair = r & 3) of halo row (zs + dz, y + dy) in slot(j)
    const int abase = (zs * WL::HY + (r >> 2)) * WL::ROWP + (r & 3) * WL::VB + 16 * h;
    const unsigned char *const ring_rd = smem + HALO_BYTES + lane * 16;
    // two fragment register sets (A: 2 planes, B: 2 planes = 16 registers each), used as in unet_wino.hip: group g multiplies step 0 from set
    // g & 1 (read at the end of the previous group), step 1 from the other set (read at the hand-over), step 2 from set g & 1 again (read behind
    // step 0's MFMAs); the next group's step 0 goes into the other set behind step 1's MFMAs.  12 groups per slice: the parity is static.
    uint4 fa[2][P], fb[2][P];
#define W32_READ(SET, SLOT_OFF, HROW, RING_OFF)                                                                                \
    do {                                                                                                                       \
        _Pragma("unroll") for (int i = 0; i < P; ++i)                                                                          \
            fa[SET][i] = *reinterpret_cast<const uint4 *>(smem + (SLOT_OFF) + abase + (HROW) * WL::ROWP + i * 32);              \
        _Pragma("unroll") for (int i = 0; i < P; ++i)                                                                          \
            fb[SET][i] = *reinterpret_cast<const uint4 *>(ring_rd + (RING_OFF) + i * 1024);                                     \
    } while (0)
    // smallest terms first
#define W32_PROD(SET)                                                                                                          \
    do {                                                                                                                       \
        acc = mfma16<F16>(fa[SET][1], fb[SET][0], acc);                                                                        \
        acc = mfma16<F16>(fa[SET][0], fb[SET][1], acc);                                                                        \
        acc = mfma16<F16>(fa[SET][0], fb[SET][0], acc);                                                                        \
    } while (0)
    // output transform of transform position J: out[x] = m0 + m1 + m2, out[x+1] = m1 - m2 - m3
#define W32_FLUSH(J)                                                                                                           \
    do {                                                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                                       \
            const float m = acc[q];                                                                                            \
            if ((J) <= 2) tot[0][q] = __fadd_rn(tot[0][q], m);                                                                 \
            if ((J) == 1) tot[1][q] = __fadd_rn(tot[1][q], m);                                                                 \
            if ((J) >= 2) tot[1][q] = __fsub_rn(tot[1][q], m);                                                                 \
            acc[q] = 0.f;                                                                                                      \
        }                                                                                                                      \
    } while (0)

    bool fresh = true;
    int sbase = 0;                                  // slot of the running slice's j = 0: four further per slice (mod 5), through tile boundaries
    for (;;) {                                      // the tiles of this chain
        if (fresh) {
            // ---- a run starts (the chain's first tile, or the first one of another sample / column block): everything synchronously
            GN_WAIT_VM_LGKM0(0);
            __syncthreads();                        // (restart: the previous run's look-ahead has landed, its statistics have been read)
            {
                int tf = threadIdx.x;
                asm volatile("" : "+v"(tf));
                if (tf < Cin) { adl[tf] = p.a[(int64_t)b * Cin + tf]; adl[ADN + tf] = p.d[(int64_t)b * Cin + tf]; }     // (Cin <= ADN < 512)
                if (tf < 64) stl[tf] = 0.0;
                if (tf < 32) {
                    const float osn = p.out_scale[(int64_t)b * p.osc_bstride + cb * 32 + tf];
                    ecl[tf] = p.act_inv ? __fmul_rn(osn, p.act_inv[b]) : osn;
                    ecl[32 + tf] = p.kbias ? p.kbias[((int64_t)b * 64 + 63) * p.Cout + cb * 32 + tf] : 0.f;
                }
            }
            bgs = reinterpret_cast<const unsigned char *>(p.wp) + (int64_t)b * p.wp_bstride + (int64_t)cb * STEPB;
            W32_ISSUE_GROUP(0);
            W32_ISSUE_GROUP(1);
            W32_ISSUE_GROUP(2);
            base0 = p.src0 + (int64_t)b * p.D * p.H * p.W * p.C0;
            set_rows(z0, y0, x0);
            issue_rows(0);
            GN_WAIT_VM_LGKM0(0);
            __syncthreads();                        // a / d table visible; groups 0 - 2 of the ring have landed
            affine_load(0);
            affine_math(0, NIT);
#pragma unroll
            for (int j = 0; j < 4; ++j) convert(j, j, 0, 4);
            __syncthreads();
            sbase = 0;
            W32_READ(0, 0, 0, 0);
            fresh = false;
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[q] = 0.f; tot[0][q] = 0.f; tot[1][q] = 0.f; }
        // the chain's next tile; `cont`: it continues this run (same sample and column block: same weights, same affine table, same statistics)
        const int nitem = item + 32;
        const bool more = nitem < item_end;
        int bn = b, cbn = cb, z0n = z0, y0n = y0, x0n = x0;
        if (more) decode(nitem, bn, cbn, z0n, y0n, x0n);
        const bool cont = more && bn == b && cbn == cb;

        for (int s = 0; s < nslices; ++s) {
            const bool last = s + 1 == nslices;
            // what this slice stages: the tile's next slice -- or, behind the last one, slice 0 of the next tile (not `cont`: this slice's own rows again
            // into slots nobody reads any more; no branch inside the MFMA stream)
            const int sn = last ? (cont ? 0 : s) : s + 1;
            if (last && cont) set_rows(z0n, y0n, x0n);
            // the weight cursor returns to the pack's start three groups before the tile's end -- `cont` or not: nothing is ever read behind the pack
            const int64_t wrap = last ? -(int64_t)nslices * 36 * bstep : 0;
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
                // hand-over of group g.  Must have landed: this group's pieces (steps 1, 2 are read right behind the barrier) and the next group's
                // (its step 0 is read at the end of this group); the row loads where the conversions start (group 3: three groups after their issue).
                // lgkmcnt(0): the halo stores of the conversion groups are published by the next barrier.  Group 0 of a tile's FIRST slice has had its
                // hand-over already: behind the prologue's barriers, or at the barrier in front of the previous tile's epilogue.
                if (g > 0 || s > 0) W32_HANDOVER_WAIT(g);
                if (g > 0 || s > 0) __builtin_amdgcn_s_barrier();
                W32_READ(Y, slo[j], dz * WL::HY + 1, (g % RING) * GB + STEPB);
                if (g == 9) bgs += wrap;
                W32_ISSUE_GROUP((g + 3) % RING);                                   // group g + 3 -> the slot group g - 1 vacated
                // the next slice's conversions, position jn into the slot this slice's jn - 1 has left (jn = 0: the spare), two pairs per group
                if (g == 0) issue_rows(sn);             // always (uniform wait counts)
                if (g == 3) { affine_load(sn); affine_math(0, NIT); convert(0, nslo[0], 0, 2); }
                if (g == 4) convert(0, nslo[0], 2, 4);
                if (g == 5) convert(1, nslo[1], 0, 2);
                if (g == 6) convert(1, nslo[1], 2, 4);
                if (g == 7) convert(2, nslo[2], 0, 2);
                if (g == 8) convert(2, nslo[2], 2, 4);
                if (g == 9) convert(3, nslo[3], 0, 2);
                if (g == 10) convert(3, nslo[3], 2, 4);
                W32_PROD(X);
                W32_READ(X, slo[j], dz * WL::HY + 2, (g % RING) * GB + 2 * STEPB);
                W32_PROD(Y);
                {   // first step of the next group
                    const int g1 = g + 1 < 12 ? g + 1 : 0;
                    const int so = g + 1 < 12 ? slo[g1 / 3] : nslo[0] * WL::SLOT;
                    W32_READ(Y, so, (g1 % 3) * WL::HY, ((g + 1) % RING) * GB);
                }
                W32_PROD(X);
                if (dz == 2) W32_FLUSH(j);              // this transform position is complete: fold it into the totals, restart the accumulator
            }
            sbase = nbase;
        }
        // the next tile's group-0 hand-over, in front of the epilogue: its groups 0 and 1 have landed (the youngest piece stays in flight), every wave
        // is through with this tile's last ring slot, the last conversions are published
        W32_HANDOVER_WAIT(0);
        __builtin_amdgcn_s_barrier();

        // ---- epilogue of this tile.  D fragment element q of lane (h, r): pair row i = (q & 3) + 8 (q >> 2) + 4 h = (y = 2 (q >> 2) + h, pair = q & 3),
        // channel r.  Everything derived from the tile / lane coordinates is computed HERE from laundered copies (hipcc otherwise hoists it above the
        // slice loop and spills it there).  No LDS scratch and no barrier: the halo slots and the ring already hold the next tile's operands
        {
            int be = b, cbe = cb, z0e = z0, y0e = y0, x0e = x0, te = threadIdx.x;
            asm volatile("" : "+s"(be), "+s"(cbe), "+s"(z0e), "+s"(y0e), "+s"(x0e));
            asm volatile("" : "+v"(te));
            const int re = te & 31, he = (te >> 5) & 1;
            const int n0 = cbe * 32;
            const int gz = z0e + zs;
            double ssum = 0.0, ssq = 0.0;               // fp64 per lane (see conv3d_split_kernel)
            const float osc = ecl[re], k63 = ecl[32 + re];
            const bool interior = z0e > 0 && z0e + WL::TZ < p.D && y0e > 0 && y0e + SP_TY < p.H && x0e > 0 && x0e + SP_TX < p.W;   // no voxel of the tile on a face
            const bool classes = p.kbias && !interior;
            const int mz = sp_axis_mask(gz, p.D);
            const float *const orow = p.out + ((((int64_t)be * p.D + gz) * p.H + y0e) * p.W + x0e) * p.Cout + n0;
            const int64_t rs2 = 2 * (int64_t)p.W * p.Cout;
            // polyphase partial: [b][z >> 1][y >> 1][x >> 1][parity class (z & 1, y & 1, x & 1)][Cout]; z0, y0, x0 are multiples of 8
            const float *prow = nullptr;
            int64_t prs = 0;
            if (p.partial) {
                prs = (int64_t)(p.W >> 1) * 8 * p.Cout;
                prow = p.partial + ((((int64_t)be * (p.D >> 1) + (gz >> 1)) * (p.H >> 1) + (y0e >> 1)) * (p.W >> 1) + (x0e >> 1)) * (8 * (int64_t)p.Cout)
                       + (int64_t)(((gz & 1) * 4 + he * 2) * p.Cout) + n0 + re;
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int gy = y0e + he, gx = x0e + e;
                float kv[16], pv[16];
                if (classes) {                          // a tile on a face: the per-class constants, all of this e's loads in flight together
                    const float *kb = p.kbias + (int64_t)be * 64 * p.Cout + n0 + re;
#pragma unroll
                    for (int q = 0; q < 16; ++q)
                        kv[q] = kb[(int64_t)((mz * 4 + sp_axis_mask(gy + 2 * (q >> 2), p.H)) * 4 + sp_axis_mask(gx + 2 * (q & 3), p.W)) * p.Cout];
                } else {
#pragma unroll
                    for (int q = 0; q < 16; ++q) kv[q] = k63;
                }
                if (prow) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) pv[q] = prow[(q >> 2) * prs + (int64_t)((q & 3) * 8 + e) * p.Cout];
                }
                unsigned vo[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) vo[i] = (unsigned)(((he * p.W + e + 2 * i) * p.Cout + re) * 4);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float v = __fmul_rn(tot[e][q], osc);
                    if (p.kbias) v = __fadd_rn(v, kv[q]);
                    if (prow) v = __fadd_rn(v, pv[q]);
                    if (p.relu) v = gn_relu(v);
                    const float *ob = orow + (q >> 2) * rs2;       // (uniform)
                    asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(vo[q & 3]), "v"(v), "s"(ob) : "memory");
                    ssum += (double)v;
                    ssq += (double)v * (double)v;
                }
            }
            if (p.osum) {
                const double s2 = ssum + __shfl_xor(ssum, 32), q2 = ssq + __shfl_xor(ssq, 32);
                if (he == 0) {
                    const unsigned sa = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + ST_OFF + re * 8;
                    asm volatile("ds_add_f64 %0, %1\n\tds_add_f64 %0, %2 offset:256" ::"v"(sa), "v"(s2), "v"(q2) : "memory");
                }
            }
        }
        if (!cont) {
            // the run's statistics leave: one set of atomics per (run, channel)
            if (p.osum) {
                GN_WAIT_VM_LGKM0(63);
                __syncthreads();
                if (tid < 32) {
                    atomicAdd(&p.osum[(int64_t)b * p.Cout + cb * 32 + tid], stl[tid]);
                    atomicAdd(&p.osq[(int64_t)b * p.Cout + cb * 32 + tid], stl[32 + tid]);
                }
            }
            if (!more) break;
            fresh = true;
        }
        item = nitem; b = bn; cb = cbn; z0 = z0n; y0 = y0n; x0 = x0n;
    }
#undef W32_ISSUE_GROUP
#undef W32_HANDOVER_WAIT
#undef W32_WAIT_LGKM0
#undef W32_READ
#undef W32_PROD
#undef W32_FLUSH
    GN_WAIT_VM_LGKM0(0);                            // (the look-ahead DMAs of the last tile land in this workgroup's LDS: not past its end)
}

// (called by conv3d_gcr_split_impl, unet_split.hip, which owns the shape checks and the occupancy-aware list / fill launches)
// -> true: the wave-specialised kernel (unet_wino32pc.hip) was launched
bool gn_launch_conv3d_wino32(const SplitArgs &p0, int tiles8, hipStream_t st) {
    SplitArgs p = p0;
    // chain length from the SAMPLE's tiles only (not the batch size): a sample's statistics are then reduced in the same order whatever the batch
    const int64_t per_sample = (int64_t)tiles8 * (p.Cout / 32);
    int chain = (int)(per_sample / 64);
    chain = chain < 1 ? 1 : chain > 16 ? 16 : chain;
    // (looked up per launch, ~100 ns against a multi-millisecond kernel: tests/test_gpu_parity.py varies it inside one process)
    if (const char *e = getenv("GARMENTNETS_WINO_CHAIN")) { const int forced = atoi(e); if (forced > 0 && forced <= 4096) chain = forced; }
    p.chain = chain;
    const int64_t items = per_sample * p.B;                                    // (occupancy-aware: the dense bound; chains past the list's end return)
    const int64_t span = 32 * (int64_t)chain;
    const unsigned grid = (unsigned)((items + span - 1) / span * 32);
    // DEFAULT: the wave-specialised form of the same tile (unet_wino32pc.hip: waves 0 - 3 multiply two z-slices each, waves 4 - 7 stage and fetch) -- bit-identical
    // results, 2.5 - 3.7 % faster in the bench step (profiles/r06_ab_experiments.txt section 6).  GARMENTNETS_WINO32_PC=0 selects this file's kernel (every wave one
    // z-slice and a share of the staging): the form sections 1 - 4 of that record measured; tests/test_gpu_parity.py runs both and compares them bit for bit
    bool pc = true;
    if (const char *e = getenv("GARMENTNETS_WINO32_PC")) pc = atoi(e) != 0;
    if (pc) { gn_launch_conv3d_wino32pc(p, grid, st); return true; }
    hipLaunchKernelGGL((conv3d_split_wino32_kernel<true>), dim3(grid), dim3(512), 0, st, p);
    return false;
}
