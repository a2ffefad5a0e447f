// unet_wino32pc.hip -- the 32-wide Winograd kernel of unet_wino32.hip with the waves SPECIALISED (round 6; the default launch of gn_launch_conv3d_wino32).
//
// unet_wino32.hip gives every wave one z-slice, one row fragment and a share of the staging: 4 fragment reads per 3 MFMAs (nothing shares a B fragment) and a
// conversion block that both waves of a SIMD execute at the same point of the loop.  Here waves 0 - 3 only multiply -- wave cw owns z-slices 2 cw and 2 cw + 1, so
// that every B fragment feeds TWO row fragments (6 reads per 6 MFMAs, two independent accumulator chains) -- and waves 4 - 7 only stage the halo (two (row, quad)
// tasks per thread) and fetch the weights: on the SIMD a producer shares with a consumer its VALU work issues beside the consumer's MFMAs.  Tile, halo slots,
// ring, pack, chains, epilogue arithmetic and the order of the products per output are those of unet_wino32.hip: the results are BIT-IDENTICAL to it.
// The default (GARMENTNETS_WINO32_PC=0 selects unet_wino32.hip's kernel); measurement: profiles/r06_ab_experiments.txt section 6.
#include "split_conv.h"

struct Wino32PcLayout {
    static constexpr int VB = 64, ROWP = 4 * VB + 16, TZ = 8, HZ = TZ + 2, HY = SP_TY + 2, ROWS = HZ * HY, SLOT = ROWS * ROWP, NSLOT = 5;
};

typedef float f32x4p __attribute__((ext_vector_type(4)));

template <bool F16>
__global__ __launch_bounds__(512, 1) void conv3d_split_wino32pc_kernel(SplitArgs p) {
    constexpr int P = 2;
    using WL = Wino32PcLayout;
    constexpr int STEPB = P * 1024, GB = 3 * STEPB, RING = 4;
    constexpr int HALO_BYTES = WL::NSLOT * WL::SLOT;
    constexpr int AD_OFF = HALO_BYTES + RING * GB;
    constexpr int ADN = 128;
    constexpr int NIT = 10;
    constexpr int ST_OFF = AD_OFF + 2 * ADN * 4;
    constexpr int EC_OFF = ST_OFF + 2 * 32 * 8;
    __shared__ __attribute__((aligned(16))) unsigned char smem[EC_OFF + 2 * 32 * 4];
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + HALO_BYTES;
    float *const adl = reinterpret_cast<float *>(smem + AD_OFF);
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int cw = wave & 3;
    const int Cin = p.C0;
    const int ncb = p.Cout / 32;
    const int tiles_z = p.D / WL::TZ;
    const int tps = tiles_z * p.tiles_x * p.tiles_y;
    const int nslices = Cin / SP_KS;
    double *const stl = reinterpret_cast<double *>(smem + ST_OFF);
    float *const ecl = reinterpret_cast<float *>(smem + EC_OFF);

    const unsigned n_items = (p.active_list ? (unsigned)(*p.active_count) : (unsigned)(p.B * tps)) * (unsigned)ncb;
    const unsigned span = 32u * (unsigned)p.chain;
    const unsigned nch = (n_items + span - 1u) / span * 32u;
    if (blockIdx.x >= nch) return;
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

    f32x16s acc[2], tot[2][2];                      // [fragment], [fragment][even / odd x]

    // ---- weights: producer cw fetches piece cw of every group (step cw >> 1, plane cw & 1), producers 0 and 1 also piece cw + 4
    const int64_t bstep = (int64_t)ncb * STEPB;
    const unsigned char *bgs = nullptr;
    const unsigned bvoff = (unsigned)(lane * 16);
#define PC_PIECE(SLOTI, PI)                                                                                                    \
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(bvoff),                                   \
                 "s"(bgs + (int64_t)((PI) >> 1) * bstep + ((PI) & 1) * 1024),                                                  \
                 "s"(lds_ring + (SLOTI) * GB + (unsigned)(((PI) >> 1) * STEPB + ((PI) & 1) * 1024)) : "memory")
#define PC_ISSUE_GROUP(SLOTI)                                                                                                  \
    do {                                                                                                                       \
        PC_PIECE(SLOTI, cw);                                                                                                   \
        if (cw < 2) PC_PIECE(SLOTI, cw + 4);                                                                                   \
        bgs += 3 * bstep;                                                                                                      \
    } while (0)
    // hand-over wait of a producer.  Its queue, oldest first: ..., P(g+1), P(g+2) (npw pieces each: 2 for producers 0 and 1, else 1) and the rows, issued behind
    // P(3) in group 0: all but the youngest npw (groups 1, 2: + the rows) must have landed; lgkmcnt(0) publishes the conversions' halo stores
#define PC_PRODUCER_WAIT(G)                                                                                                    \
    do {                                                                                                                       \
        const bool rows_ = (G) == 1 || (G) == 2;                                                                               \
        if (cw < 2) { if (rows_) GN_WAIT_VM_LGKM0(2 + 2 * NIT); else GN_WAIT_VM_LGKM0(2); }                                     \
        else if (cw == 2) { if (rows_) GN_WAIT_VM_LGKM0(1 + 2 * NIT); else GN_WAIT_VM_LGKM0(1); }                               \
        else { if (rows_) GN_WAIT_VM_LGKM0(1 + NIT); else GN_WAIT_VM_LGKM0(1); }                                                \
    } while (0)

    // ---- staging (producers).  Task k of producer thread pt = tid - 256 is the (halo row, channel quad) of virtual thread pt + 256 k in unet_wino32.hip's
    // bank-conflict-free order; 400 tasks: producer 3 has no second task (rows >= 112), producer 2's second task is real for rows 96 .. 99 (the rest repeat row 99)
    auto stage_row = [&](int t) {
        const int srow = (t >> 6) * 16 + 2 * ((t >> 2) & 3) + ((t >> 4) & 1) + 8 * ((t >> 5) & 1);
        return srow < WL::ROWS ? srow : WL::ROWS - 1;
    };
    const int pt = tid & 255;
    const bool two_tasks = cw < 3;                                         // (uniform)
    const int c4 = (tid & 3) * 4;
    int wrow[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) wrow[k] = stage_row(pt + 256 * k) * WL::ROWP + c4 * 2;
    unsigned voff1[2] = {0, 0}, voff0[2] = {0, 0}, voff9[2] = {0, 0}, inb[2] = {0, 0};
    auto set_rows = [&](int z0_, int y0_, int x0_) {
        int t = threadIdx.x & 255;
        asm volatile("" : "+v"(t));
        const int cq = (t & 3) * 4;
        const unsigned vs = (unsigned)p.C0 * 4u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int rr = stage_row(t + 256 * k), hz = rr / WL::HY, hy = rr - hz * WL::HY;
            const int gz = z0_ + hz - 1, gy = y0_ + hy - 1;
            const bool rowin = gz >= 0 && gz < p.D && gy >= 0 && gy < p.H;
            voff1[k] = rowin ? ((unsigned)((gz * p.H + gy) * p.W + x0_) * (unsigned)p.C0 + (unsigned)cq) * 4u : (unsigned)cq * 4u;
            const bool in0 = rowin && x0_ - 1 >= 0, in9 = rowin && x0_ + 8 < p.W;
            voff0[k] = in0 ? voff1[k] - vs : voff1[k];
            voff9[k] = in9 ? voff1[k] + 8u * vs : voff1[k];
            inb[k] = rowin ? (0x1feu | (in0 ? 1u : 0u) | (in9 ? 0x200u : 0u)) : 0u;
        }
    };
    const float *base0 = p.src0;
    f32x4p raw[2][NIT];
    auto issue_rows = [&](int sl) {
        const unsigned cb4 = (unsigned)sl * (SP_KS * 4u), vs = (unsigned)p.C0 * 4u;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && !two_tasks) break;
#pragma unroll
            for (int v = 0; v < NIT; ++v) {
                const unsigned vo = (v == 0 ? voff0[k] : v == NIT - 1 ? voff9[k] : voff1[k] + (unsigned)(v - 1) * vs) + cb4;
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw[k][v]) : "v"(vo), "s"(base0) : "memory");
            }
        }
    };
    float4 afa, afd;
    auto affine_rows = [&](int sl) {
        afa = *reinterpret_cast<const float4 *>(adl + sl * SP_KS + c4);
        afd = *reinterpret_cast<const float4 *>(adl + ADN + sl * SP_KS + c4);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && !two_tasks) break;
#pragma unroll
            for (int v = 0; v < NIT; ++v) {
                asm volatile("" : "+v"(raw[k][v]));
                const bool in = (inb[k] >> v) & 1u;
                raw[k][v].x = in ? __fmaf_rn(raw[k][v].x, afa.x, afd.x) : 0.f;
                raw[k][v].y = in ? __fmaf_rn(raw[k][v].y, afa.y, afd.y) : 0.f;
                raw[k][v].z = in ? __fmaf_rn(raw[k][v].z, afa.z, afd.z) : 0.f;
                raw[k][v].w = in ? __fmaf_rn(raw[k][v].w, afa.w, afd.w) : 0.f;
            }
        }
    };
    auto convert = [&](int jp, int slot, int k0, int k1) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            if (k == 1 && !two_tasks) break;
            unsigned char *dst = smem + slot * WL::SLOT + wrow[k];
#pragma unroll
            for (int q = k0; q < k1; ++q) {
                f32x4p cv;
                if (jp == 0) cv = raw[k][2 * q] - raw[k][2 * q + 2];
                else if (jp == 1) cv = raw[k][2 * q + 1] + raw[k][2 * q + 2];
                else if (jp == 2) cv = raw[k][2 * q + 2] - raw[k][2 * q + 1];
                else cv = raw[k][2 * q + 1] - raw[k][2 * q + 3];
                uint2 cp[P];
                split4<P, F16>(cv.x, cv.y, cv.z, cv.w, cp);
#pragma unroll
                for (int i = 0; i < P; ++i) *reinterpret_cast<uint2 *>(dst + q * WL::VB + i * 32) = cp[i];
            }
        }
    };

    // ---- fragments (consumers): rows (y = r >> 2, pair = r & 3) of halo rows (2 cw + f + dz, y + dy) in slot(j), f = 0, 1
    const int abase = (2 * cw * WL::HY + (r >> 2)) * WL::ROWP + (r & 3) * WL::VB + 16 * h;
    const unsigned char *const ring_rd = smem + HALO_BYTES + lane * 16;
    uint4 fa[2][2][P], fb[2][P];
#define PC_READ(SET, SLOT_OFF, HROW, RING_OFF)                                                                                 \
    do {                                                                                                                       \
        _Pragma("unroll") for (int f = 0; f < 2; ++f)                                                                          \
            _Pragma("unroll") for (int i = 0; i < P; ++i)                                                                      \
                fa[SET][f][i] = *reinterpret_cast<const uint4 *>(smem + (SLOT_OFF) + abase + ((HROW) + f * WL::HY) * WL::ROWP + i * 32); \
        _Pragma("unroll") for (int i = 0; i < P; ++i)                                                                          \
            fb[SET][i] = *reinterpret_cast<const uint4 *>(ring_rd + (RING_OFF) + i * 1024);                                     \
    } while (0)
    // smallest terms first, the two fragments alternating (per output the products arrive in unet_wino32.hip's order)
#define PC_PROD(SET)                                                                                                           \
    do {                                                                                                                       \
        acc[0] = mfma16<F16>(fa[SET][0][1], fb[SET][0], acc[0]); acc[1] = mfma16<F16>(fa[SET][1][1], fb[SET][0], acc[1]);       \
        acc[0] = mfma16<F16>(fa[SET][0][0], fb[SET][1], acc[0]); acc[1] = mfma16<F16>(fa[SET][1][0], fb[SET][1], acc[1]);       \
        acc[0] = mfma16<F16>(fa[SET][0][0], fb[SET][0], acc[0]); acc[1] = mfma16<F16>(fa[SET][1][0], fb[SET][0], acc[1]);       \
    } while (0)
#define PC_FLUSH(J)                                                                                                            \
    do {                                                                                                                       \
        _Pragma("unroll") for (int f = 0; f < 2; ++f)                                                                          \
            _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                                   \
                const float m = acc[f][q];                                                                                     \
                if ((J) <= 2) tot[f][0][q] = __fadd_rn(tot[f][0][q], m);                                                       \
                if ((J) == 1) tot[f][1][q] = __fadd_rn(tot[f][1][q], m);                                                       \
                if ((J) >= 2) tot[f][1][q] = __fsub_rn(tot[f][1][q], m);                                                       \
                acc[f][q] = 0.f;                                                                                               \
            }                                                                                                                  \
    } while (0)

    // The two roles run SEPARATE copies of the chain / tile / slice / group loops (same barrier sequence): their register state -- accumulators, totals and fragments
    // here, two tasks' rows there -- then has disjoint live ranges and shares the wave's 256 registers (as one loop with role branches inside it needed 334 spills,
    // and a spilled register that an inline-asm load is still writing is garbage)
    if (consumer) {
        bool fresh = true;
        int sbase = 0;
        for (;;) {
            if (fresh) {
                GN_WAIT_VM_LGKM0(0);
                __syncthreads();
                {
                    int tf = threadIdx.x;               // (consumers are threads 0 .. 255: they fill the tables)
                    asm volatile("" : "+v"(tf));
                    if (tf < Cin) { adl[tf] = p.a[(int64_t)b * Cin + tf]; adl[ADN + tf] = p.d[(int64_t)b * Cin + tf]; }
                    if (tf < 64) stl[tf] = 0.0;
                    if (tf < 32) {
                        const float osn = p.out_scale[(int64_t)b * p.osc_bstride + cb * 32 + tf];
                        ecl[tf] = p.act_inv ? __fmul_rn(osn, p.act_inv[b]) : osn;
                        ecl[32 + tf] = p.kbias ? p.kbias[((int64_t)b * 64 + 63) * p.Cout + cb * 32 + tf] : 0.f;
                    }
                }
                GN_WAIT_VM_LGKM0(0);
                __syncthreads();
                __syncthreads();
                sbase = 0;
                PC_READ(0, 0, 0, 0);
                fresh = false;
            }
#pragma unroll
            for (int f = 0; f < 2; ++f)
#pragma unroll
                for (int q = 0; q < 16; ++q) { acc[f][q] = 0.f; tot[f][0][q] = 0.f; tot[f][1][q] = 0.f; }
            const int nitem = item + 32;
            const bool more = nitem < item_end;
            int bn = b, cbn = cb, z0n = z0, y0n = y0, x0n = x0;
            if (more) decode(nitem, bn, cbn, z0n, y0n, x0n);
            const bool cont = more && bn == b && cbn == cb;
            for (int s = 0; s < nslices; ++s) {
                const int nbase = sbase == 0 ? 4 : sbase - 1;
                int slo[4], nslo0;
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int a_ = sbase + j; slo[j] = (a_ >= WL::NSLOT ? a_ - WL::NSLOT : a_) * WL::SLOT; }
                nslo0 = nbase * WL::SLOT;
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    const int j = g / 3, dz = g % 3, X = g & 1, Y = X ^ 1;
                    if (g > 0 || s > 0) __builtin_amdgcn_s_barrier();
                    PC_READ(Y, slo[j], dz * WL::HY + 1, (g % RING) * GB + STEPB);
                    PC_PROD(X);
                    PC_READ(X, slo[j], dz * WL::HY + 2, (g % RING) * GB + 2 * STEPB);
                    PC_PROD(Y);
                    {
                        const int g1 = g + 1 < 12 ? g + 1 : 0;
                        const int so = g + 1 < 12 ? slo[g1 / 3] : nslo0;
                        PC_READ(Y, so, (g1 % 3) * WL::HY, ((g + 1) % RING) * GB);
                    }
                    PC_PROD(X);
                    if (dz == 2) PC_FLUSH(j);
                }
                sbase = nbase;
            }
            __builtin_amdgcn_s_barrier();
            // ---- epilogue: as unet_wino32.hip, for the wave's two z-slices
            {
                int be = b, cbe = cb, z0e = z0, y0e = y0, x0e = x0, te = threadIdx.x;
                asm volatile("" : "+s"(be), "+s"(cbe), "+s"(z0e), "+s"(y0e), "+s"(x0e));
                asm volatile("" : "+v"(te));
                const int re = te & 31, he = (te >> 5) & 1;
                const int n0 = cbe * 32;
                double ssum = 0.0, ssq = 0.0;
                const float osc = ecl[re], k63 = ecl[32 + re];
                const bool interior = z0e > 0 && z0e + WL::TZ < p.D && y0e > 0 && y0e + SP_TY < p.H && x0e > 0 && x0e + SP_TX < p.W;
                const bool classes = p.kbias && !interior;
                const int64_t rs2 = 2 * (int64_t)p.W * p.Cout;
#pragma unroll
                for (int f = 0; f < 2; ++f) {
                    const int gz = z0e + 2 * cw + f;
                    const int mz = sp_axis_mask(gz, p.D);
                    const float *const orow = p.out + ((((int64_t)be * p.D + gz) * p.H + y0e) * p.W + x0e) * p.Cout + n0;
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
                        if (classes) {
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
                            float v = __fmul_rn(tot[f][e][q], osc);
                            if (p.kbias) v = __fadd_rn(v, kv[q]);
                            if (prow) v = __fadd_rn(v, pv[q]);
                            if (p.relu) v = gn_relu(v);
                            const float *ob = orow + (q >> 2) * rs2;
                            asm volatile("global_store_dword %0, %1, %2 nt" ::"v"(vo[q & 3]), "v"(v), "s"(ob) : "memory");
                            ssum += (double)v;
                            ssq += (double)v * (double)v;
                        }
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
    } else {
        bool fresh = true;
        int sbase = 0;
        for (;;) {
            if (fresh) {
                GN_WAIT_VM_LGKM0(0);
                __syncthreads();
                bgs = reinterpret_cast<const unsigned char *>(p.wp) + (int64_t)b * p.wp_bstride + (int64_t)cb * STEPB;
                PC_ISSUE_GROUP(0);
                PC_ISSUE_GROUP(1);
                PC_ISSUE_GROUP(2);
                base0 = p.src0 + (int64_t)b * p.D * p.H * p.W * p.C0;
                set_rows(z0, y0, x0);
                issue_rows(0);
                GN_WAIT_VM_LGKM0(0);
                __syncthreads();                    // the consumers' tables are visible; groups 0 - 2 of the ring and the rows have landed
                affine_rows(0);
#pragma unroll
                for (int j = 0; j < 4; ++j) convert(j, j, 0, 4);
                GN_WAIT_VM_LGKM0(0);
                __syncthreads();
                sbase = 0;
                fresh = false;
            }
            const int nitem = item + 32;
            const bool more = nitem < item_end;
            int bn = b, cbn = cb, z0n = z0, y0n = y0, x0n = x0;
            if (more) decode(nitem, bn, cbn, z0n, y0n, x0n);
            const bool cont = more && bn == b && cbn == cb;
            for (int s = 0; s < nslices; ++s) {
                const bool last = s + 1 == nslices;
                const int sn = last ? (cont ? 0 : s) : s + 1;
                if (last && cont) set_rows(z0n, y0n, x0n);
                const int64_t wrap = last ? -(int64_t)nslices * 36 * bstep : 0;
                const int nbase = sbase == 0 ? 4 : sbase - 1;
                int nslo[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { const int n_ = nbase + j; nslo[j] = n_ >= WL::NSLOT ? n_ - WL::NSLOT : n_; }
#pragma unroll
                for (int g = 0; g < 12; ++g) {
                    if (g > 0 || s > 0) {
                        PC_PRODUCER_WAIT(g);
                        __builtin_amdgcn_s_barrier();
                    }
                    if (g == 9) bgs += wrap;
                    PC_ISSUE_GROUP((g + 3) % RING);
                    if (g == 0) issue_rows(sn);
                    if (g == 3) { affine_rows(sn); convert(0, nslo[0], 0, 2); }
                    if (g == 4) convert(0, nslo[0], 2, 4);
                    if (g == 5) convert(1, nslo[1], 0, 2);
                    if (g == 6) convert(1, nslo[1], 2, 4);
                    if (g == 7) convert(2, nslo[2], 0, 2);
                    if (g == 8) convert(2, nslo[2], 2, 4);
                    if (g == 9) convert(3, nslo[3], 0, 2);
                    if (g == 10) convert(3, nslo[3], 2, 4);
                }
                sbase = nbase;
            }
            PC_PRODUCER_WAIT(0);
            __builtin_amdgcn_s_barrier();
            if (!cont) {
                if (p.osum) __syncthreads();
                if (!more) break;
                fresh = true;
            }
            item = nitem; b = bn; cb = cbn; z0 = z0n; y0 = y0n; x0 = x0n;
        }
    }
#undef PC_PIECE
#undef PC_ISSUE_GROUP
#undef PC_PRODUCER_WAIT
#undef PC_READ
#undef PC_PROD
#undef PC_FLUSH
    GN_WAIT_VM_LGKM0(0);
}

void gn_launch_conv3d_wino32pc(const SplitArgs &p, unsigned grid, hipStream_t st) {
    hipLaunchKernelGGL((conv3d_split_wino32pc_kernel<true>), dim3(grid), dim3(512), 0, st, p);
}
