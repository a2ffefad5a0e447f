// unet_split.hip -- split-operand variant of the 'gcr' conv on the 16-bit matrix cores (gn_conv3d_gcr_split).  Its f16x2 mode is
// the default conv arithmetic of garmentnets_amd (arith.Arith.conv_mode); unet.hip holds the plain fp32-MFMA kernel.
//
// Same implicit-GEMM structure, tiling, GroupNorm-on-load, upsample/concat folding and epilogue as conv3d_gcr_kernel
// (unet.hip), but every fp32 operand is decomposed EXACTLY into P bf16 planes (x = x1 + x2 [+ x3], xi = bf16_rn of the
// running residual) and the products are formed on the bf16 matrix cores (v_mfma_f32_32x32x16_bf16: exact bf16 x bf16
// products, fp32 accumulation, 16x the k-throughput of v_mfma_f32_32x32x2_f32):
//   P = 3 : 6 products x1w1 + x1w2 + x2w1 + x1w3 + x2w2 + x3w1   -> dropped terms <= 2^-24 relative: fp32-class products
//   P = 2 : 3 products x1w1 + x1w2 + x2w1                       -> 2^-16 relative per product
// A third mode uses fp16 planes (11-bit significands): x = x1 + x2 leaves <= 2^-22 |x| WHEN BOTH PLANES ARE NORMAL, and the same 3
// products drop only x2w2 -- half the matrix-core work of bf16 x 3 at an error far below that of the fp32 accumulation of a 27*Cin-term
// dot product.  fp16's narrow exponent is handled by exact power-of-two scales: one per OUTPUT CHANNEL for the weights
// (row maximum in [1, 2); a weight below 2^-3 of its row's largest has a subnormal second plane: residual 2^-25 of the row maximum) and
// one per SAMPLE for the activations, chosen on the device from the GroupNorm statistics (largest per-channel rms of the normalised
// activations in [1, 2): gn_groupnorm_affine's act_inv_scale) -- a channel's values are bounded by rms * sqrt(V), so fp16 cannot
// overflow for any checkpoint, and values down to 1/8 of the sample's typical magnitude keep a normal second plane (below: residual
// 2^-25 of that magnitude, absolute).  Both scales are undone exactly in the epilogue.  Without the sample scale (act_inv == NULL, the
// raw kernel) a GroupNorm output beyond +-65504 turns into inf and shows up as inf/NaN in the output, never as a wrong finite number.
// One bf16 MFMA consumes the 16-channel slice of a tap at once: lane (h = lane>>5, r = lane&31) supplies channels 8h..8h+7
// of voxel r (A) / of output channel r (B) -- the same fragment the fp32 kernel reads, so the LDS layout is the fp32 one
// with P bf16 planes per voxel.  The result is validated against the same oracle and goldens as the fp32 path
// (tests/test_gpu_parity.py::test_conv3d_split_*); DESIGN.md section 4 reports speed and error next to the fp32 kernel.
#include "common.h"
#include "split_conv.h"

// Epilogue of one 32 x 32 D fragment when the tile lies inside the volume and is active: lane (h, r) holds channel n and the 16 voxels
// (y = j, x = 4 h + k), q = 4 j + k, of one y half.  `ob` / `pb` point at (j, k) = (0, 0) of this lane; every other (j, k) is a
// workgroup-uniform element offset from there (the generic path below spent ~40 instructions per value, five of them quarter-rate
// integer multiplies, on 64-bit addresses and bounds: 9 % of the 128 -> 128 layer, 25 % of a 32 -> 32 one).  Same arithmetic, same
// order as the generic path: value * scale (+ polyphase partial) (ReLU), channel statistics accumulated q = 0 .. 15.
// kb: NULL or this lane's column of the sample's kbias table (class stride Cout); interior tiles (workgroup-uniform) add the one class-63
// constant, tiles on a face of the volume look the class of each voxel up (gz: the fragment's plane; gy, gx: its first row / column).
__device__ __forceinline__ void sp_store_frag_full(const SplitArgs &p, const f32x16s &val, float osc, float *ob, const float *pb, double &ssum, double &ssq,
                                                   const float *kb = nullptr, bool interior = true, int gz = 0, int gy = 0, int gx = 0) {
    const int64_t rs = (int64_t)p.W * p.Cout;
    float pv[16];
    float kv[16];
    if (kb) {
        if (interior) {
            const float k63 = kb[63 * (int64_t)p.Cout];
#pragma unroll
            for (int q = 0; q < 16; ++q) kv[q] = k63;
        } else {
            const int mz = sp_axis_mask(gz, p.D);
#pragma unroll
            for (int q = 0; q < 16; ++q) kv[q] = kb[(int64_t)((mz * 4 + sp_axis_mask(gy + (q >> 2), p.H)) * 4 + sp_axis_mask(gx + (q & 3), p.W)) * p.Cout];
        }
    }
    if (pb) {
        const int64_t prs = (int64_t)(p.W >> 1) * 8 * p.Cout;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int j = q >> 2, k = q & 3;
            pv[q] = pb[(j >> 1) * prs + (int64_t)((k >> 1) * 8 + (j & 1) * 2 + (k & 1)) * p.Cout];
        }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int j = q >> 2, k = q & 3;
        float v = __fmul_rn(val[q], osc);
        if (kb) v = __fadd_rn(v, kv[q]);
        if (pb) v = __fadd_rn(v, pv[q]);
        if (p.relu) v = gn_relu(v);
        ob[j * rs + (int64_t)k * p.Cout] = v;
        ssum += (double)v;
        ssq += (double)v * (double)v;
    }
}

// LDS halo layout.  ds_read_b128 is serviced in four 16-lane groups that are NOT contiguous lane ranges ({0-3,12-15,20-27},
// {4-11,16-19,28-31} and the same +32: MI355X_MICROARCH.md, LDS table); with the fragment's row r = (y = r>>3, x = r&7) a group is
// four runs of 4 x-consecutive voxels in 4 different halo rows, and 16-byte bank quads repeat every 256 B.  Two planes (P = 2):
// voxel = 64 B unpadded (x step = 4 quads) and ONE 16-byte pad per halo ROW (row pitch 41 quads, odd) -> every group touches 16
// distinct quads: conflict-free, and the halo shrinks to 39.4 KB.  (The earlier per-voxel pad, 80 B, was 3-way conflicted on every A
// read; found by enumerating the real lane groups: tools/dev/lds_bank_check.py.)  P = 3: 96-byte voxels + the row pad = 2-way (no
// conflict-free pitch exists; the per-voxel pad was 3-way).
template <int P, int HZ = SP_HZ> struct HaloLayout {
    static constexpr int VB = P * 32;                                      // bytes per voxel
    static constexpr int ROWP = SP_HX * VB + 16;                           // bytes per halo row
    static constexpr int BYTES = HZ * SP_HY * ROWP;
    __device__ static constexpr __forceinline__ int at(int hz, int hy, int hx) { return (hz * SP_HY + hy) * ROWP + hx * VB; }
};

// MT = z-slices per wave: 1 -> tile 4 x 8 x 8; 2 -> tile 8 x 8 x 8 (wave w takes slices w and w+4).  The tall tile is for the 32-wide
// (NT = 1) layers: twice the MFMAs per barrier, per staged voxel (halo 1000 instead of 2 x 600) and per B fragment.
template <int NT, int P, bool F16, int MT>
__global__ __launch_bounds__(256, 2) void conv3d_split_kernel(SplitArgs p) {
    constexpr int CT = NT * 32;
    constexpr int TZ = SP_TZ * MT, HZ = TZ + 2, HVOX = HZ * SP_HY * SP_HX, NF = 2 * MT;   // NF: 32-row fragments per wave
    // LDS (ONE array: a second __shared__ object makes hipcc drain the LDS-DMA queue before every ds_read):
    //   halo : HaloLayout<P> (600 voxels, P planes of 16 halfs each)
    //   ring : DEPTH x (NT*P) B fragments of 1 KB in lane order, filled by global_load_lds_dwordx4 DEPTH taps ahead
    using HL = HaloLayout<P, HZ>;
    constexpr int HALO_BYTES = HL::BYTES;
    constexpr int BTAP = NT * P * 1024;
    constexpr int DEPTH = (P == 2) ? 4 : 2;         // power of two; P = 3 has no LDS to spare next to its 67 KB halo
    constexpr int CH = (NT * P + 3) / 4;            // DMA instructions per wave per tap (the same for every wave: counted waits)
    __shared__ __attribute__((aligned(16))) unsigned char smem[HALO_BYTES + DEPTH * BTAP];
    unsigned char *const halo = smem;
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Cin = p.C0 + p.C1;
    const int ncb = p.Cout / CT;
    const int tiles_z = (p.D + TZ - 1) / TZ;
    int b, tile, cb;
    if (!sp_work_item(p, ncb, tiles_z * p.tiles_x * p.tiles_y, b, tile, cb)) return;      // (workgroup-uniform)
    const int tz = tile % tiles_z; tile /= tiles_z;
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile;
    const int z0 = tz * TZ, y0 = ty * SP_TY, x0 = tx * SP_TX;
    const int n0 = cb * CT;
    const int D1 = p.D >> 1, H1 = p.H >> 1, W1 = p.W >> 1;

    f32x16s acc[NF][NT], tot[NF][NT];               // fragment f = 2 m + t: z-slice wave + 4 m, y half t
#pragma unroll
    for (int t = 0; t < NF; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc[t][u][q] = 0.f; tot[t][u][q] = 0.f; }

    {
    const int abase = HL::at(wave, r >> 3, r & 7) + 16 * h;                          // bytes; plane pl at +32*pl
    constexpr int AF1 = 4 * HL::ROWP, AFZ = 4 * SP_HY * HL::ROWP;                       // y half, second z-slice
    const int nslices = Cin / SP_KS;
    // B operands: the pack is in fragment order [slice][tap][Cout/32][plane][lane] (16 B per lane), so the NT*P fragments a
    // workgroup needs for one (slice, tap) step are NT*P contiguous KB.  They are DMA'd into the ring DEPTH steps ahead (no VGPRs;
    // every wave issues CH 1-KB pieces per step -- when NT*P is not a multiple of 4 some pieces are fetched twice, harmlessly, so
    // that the s_waitcnt counts are the same in every wave) and read back in lane order (conflict-free b128).
    const int64_t bstep = (int64_t)(p.Cout / 32) * P * 1024;                          // bytes per (slice, tap) step
    const unsigned char *bg = reinterpret_cast<const unsigned char *>(p.wp) + (int64_t)b * p.wp_bstride + (int64_t)cb * BTAP + lane * 16;
    int jf = 0;                                     // flat step index s*27 + tap of the NEXT step to fetch
#define SP_ISSUE_B()                                                                                                           \
    do {                                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < CH; ++k) {                                                                       \
            const int c = (wave + 4 * k) % (NT * P);                                                                           \
            gn_glds16(bg + c * 1024, lds_ring + (jf & (DEPTH - 1)) * BTAP + c * 1024);                                         \
        }                                                                                                                      \
        bg += bstep; ++jf;                                                                                                     \
    } while (0)
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) SP_ISSUE_B();

    // PRE (the 32-wide kernels, where the registers exist): a thread stages the same halo voxels in every slice, so their index
    // arithmetic (halo coordinates by magic division, bounds, 64-bit offsets) is done once -- it was more than half of the staging
    // VALU work.  goff0 / goff1: index of the voxel within the sample in src0 / the half-resolution src1 (-1 = outside the volume or
    // past the last item; voxel indices, not element offsets: 256^3 x 128 channels does not fit 32 bits); loff: its byte offset in
    // the LDS halo.
    constexpr bool PRE = (NT == 1 && MT == 1);
    constexpr int NITP = (HVOX * 4 + 255) / 256;
    int goff0[PRE ? NITP : 1], goff1[PRE ? NITP : 1], loff[PRE ? NITP : 1];
    if (PRE) {
#pragma unroll
        for (int it = 0; it < NITP; ++it) {
            const int idx = tid + it * 256, hv = idx >> 2, c4 = (tid & 3) * 4;
            const int hx = hv % SP_HX, hy = (hv / SP_HX) % SP_HY, hz = hv / (SP_HX * SP_HY);
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool in = idx < HVOX * 4 && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            goff0[it] = in ? (gz * p.H + gy) * p.W + gx : -1;
            goff1[it] = in ? ((gz >> 1) * H1 + (gy >> 1)) * W1 + (gx >> 1) : -1;
            loff[it] = idx < HVOX * 4 ? HL::at(hz, hy, hx) + c4 * 2 : -1;
        }
    }

    for (int s = 0; s < nslices; ++s) {
        const int c0 = s * SP_KS;
        // (every halo read of the previous slice completed before its last tap's barrier: the halo can be overwritten)
        if (PRE) {
            const bool from1 = c0 >= p.C0;
            const int c4 = (tid & 3) * 4;
            const float *base = (from1 ? p.src1 + (int64_t)b * D1 * H1 * W1 * p.C1 + (c0 - p.C0) : p.src0 + (int64_t)b * p.D * p.H * p.W * p.C0 + c0) + c4;
            const int64_t Cs = from1 ? p.C1 : p.C0;
            const float4 av = *reinterpret_cast<const float4 *>(p.a + (int64_t)b * Cin + c0 + c4);
            const float4 dv = *reinterpret_cast<const float4 *>(p.d + (int64_t)b * Cin + c0 + c4);
            float4 raw[NITP];
#pragma unroll
            for (int it = 0; it < NITP; ++it) {
                const int go = from1 ? goff1[PRE ? it : 0] : goff0[PRE ? it : 0];
                raw[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (go >= 0) raw[it] = *reinterpret_cast<const float4 *>(base + go * Cs);
            }
#pragma unroll
            for (int it = 0; it < NITP; ++it) {
                const int lo = loff[PRE ? it : 0];
                if (lo >= 0) {
                    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
                    if (goff0[PRE ? it : 0] >= 0) {                         // zero padding comes AFTER the affine
                        v0 = __fmaf_rn(raw[it].x, av.x, dv.x);
                        v1 = __fmaf_rn(raw[it].y, av.y, dv.y);
                        v2 = __fmaf_rn(raw[it].z, av.z, dv.z);
                        v3 = __fmaf_rn(raw[it].w, av.w, dv.w);
                    }
                    uint2 pl[P];
                    split4<P, F16>(v0, v1, v2, v3, pl);
#pragma unroll
                    for (int i = 0; i < P; ++i) *reinterpret_cast<uint2 *>(halo + lo + i * 32) = pl[i];
                }
            }
        } else {   // ---- halo stage: GroupNorm affine, then exact split into P bf16 planes.  All the tile's loads are issued before the
            //      first use (one exposed latency per slice), the other workgroup of the CU computes meanwhile
            const bool from1 = c0 >= p.C0;
            const float *src = from1 ? p.src1 : p.src0;
            const int Cs = from1 ? p.C1 : p.C0;
            const int cs = from1 ? c0 - p.C0 : c0;
            const float *ab = p.a + (int64_t)b * Cin + c0;
            const float *db = p.d + (int64_t)b * Cin + c0;
            constexpr int NIT = (HVOX * 4 + 255) / 256;
            const int c4 = (tid & 3) * 4;                                  // 256 % 4 == 0: the same channel quad every iteration
            const float4 av = *reinterpret_cast<const float4 *>(ab + c4);
            const float4 dv = *reinterpret_cast<const float4 *>(db + c4);
            // in batches of NB rows per thread (all of a batch's loads in flight before its first use; the tall tile takes two
            // batches: 16 rows would not fit beside its 128 accumulator registers)
            constexpr int NB = NIT <= 10 ? NIT : (NIT + 1) / 2;
            for (int it0 = 0; it0 < NIT; it0 += NB) {
            float4 raw[NB];
            bool inb[NB];
#pragma unroll
            for (int itb = 0; itb < NB; ++itb) {
                const int it = itb, idx = tid + (it0 + itb) * 256;
                const int hv = idx >> 2;
                const int hx = hv % SP_HX, hy = (hv / SP_HX) % SP_HY, hz = hv / (SP_HX * SP_HY);
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
                inb[it] = idx < HVOX * 4 && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                raw[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (inb[it]) {
                    int64_t off;
                    if (from1) off = ((((int64_t)b * D1 + (gz >> 1)) * H1 + (gy >> 1)) * W1 + (gx >> 1)) * Cs + cs + c4;
                    else off = ((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * Cs + cs + c4;
                    raw[it] = *reinterpret_cast<const float4 *>(src + off);
                }
            }
#pragma unroll
            for (int itb = 0; itb < NB; ++itb) {
                const int it = itb, idx = tid + (it0 + itb) * 256;
                if (idx < HVOX * 4) {
                    const int hv = idx >> 2;
                    float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
                    if (inb[it]) {                                          // zero padding comes AFTER the affine
                        v0 = __fmaf_rn(raw[it].x, av.x, dv.x);
                        v1 = __fmaf_rn(raw[it].y, av.y, dv.y);
                        v2 = __fmaf_rn(raw[it].z, av.z, dv.z);
                        v3 = __fmaf_rn(raw[it].w, av.w, dv.w);
                    }
                    uint2 pl[P];
                    split4<P, F16>(v0, v1, v2, v3, pl);
#pragma unroll
                    for (int i = 0; i < P; ++i) *reinterpret_cast<uint2 *>(halo + HL::at(hv / (SP_HX * SP_HY), (hv / SP_HX) % SP_HY, hv % SP_HX) + i * 32 + c4 * 2) = pl[i];
                }
            }
            }
        }
        __syncthreads();                            // halo visible (and this wave's outstanding ring DMAs have landed)
        // step j = s*27 + tap lives in ring slot j % DEPTH
        const unsigned char *const ring_rd = smem + HALO_BYTES + lane * 16;
        int jcur = s * 27;
        uint4 af[NF][P], naf[NF][P], bf[NT][P], nbf[NT][P];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int i = 0; i < P; ++i) naf[f][i] = *reinterpret_cast<const uint4 *>(halo + abase + (f >> 1) * AFZ + (f & 1) * AF1 + i * 32);
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int i = 0; i < P; ++i) nbf[u][i] = *reinterpret_cast<const uint4 *>(ring_rd + (jcur & (DEPTH - 1)) * BTAP + (u * P + i) * 1024);
#pragma unroll
        for (int tap = 0; tap < 27; ++tap, ++jcur) {
#pragma unroll
            for (int f = 0; f < NF; ++f)
#pragma unroll
                for (int i = 0; i < P; ++i) af[f][i] = naf[f][i];
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int i = 0; i < P; ++i) bf[u][i] = nbf[u][i];
            // hand-over: every wave holds step j's fragments in registers (lgkmcnt(0)) and its DMA share of step j+1 has landed
            // (VM queue, oldest first: steps j+1 .. j+DEPTH-1, CH pieces each)
            GN_WAIT_VM_LGKM0((DEPTH - 2) * CH);
            __builtin_amdgcn_s_barrier();
            if (tap + 1 < 27) {
                const int t1 = tap + 1;
                const int toff = HL::at(t1 / 9, (t1 / 3) % 3, t1 % 3);
#pragma unroll
                for (int f = 0; f < NF; ++f)
#pragma unroll
                    for (int i = 0; i < P; ++i) naf[f][i] = *reinterpret_cast<const uint4 *>(halo + abase + (f >> 1) * AFZ + (f & 1) * AF1 + toff + i * 32);
#pragma unroll
                for (int u = 0; u < NT; ++u)
#pragma unroll
                    for (int i = 0; i < P; ++i) nbf[u][i] = *reinterpret_cast<const uint4 *>(ring_rd + ((jcur + 1) & (DEPTH - 1)) * BTAP + (u * P + i) * 1024);
            }
            SP_ISSUE_B();                           // step j+DEPTH overwrites step j's slot (read by everyone before the barrier)
            __builtin_amdgcn_sched_barrier(0);
#define SP_PROD(IA, IB)                                                                                                        \
            _Pragma("unroll") for (int u = 0; u < NT; ++u)                                                                     \
                _Pragma("unroll") for (int f = 0; f < NF; ++f) acc[f][u] = mfma16<F16>(af[f][IA], bf[u][IB], acc[f][u]);
            // smallest terms first
            if (P == 3) { SP_PROD(P - 1, 0) SP_PROD(1, P - 2) SP_PROD(0, P - 1) }
            SP_PROD(1, 0) SP_PROD(0, 1) SP_PROD(0, 0)
#undef SP_PROD
            __builtin_amdgcn_sched_barrier(0);
        }
        // tap 26 prefetched nothing: step j+1's fragments are read after the next slice's staging barrier; the last tap's own
        // barrier is the point after which no wave reads this slice's halo any more
#pragma unroll
        for (int t = 0; t < NF; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int q = 0; q < 16; ++q) { tot[t][u][q] = __fadd_rn(tot[t][u][q], acc[t][u][q]); acc[t][u][q] = 0.f; }
    }
#undef SP_ISSUE_B
    GN_WAIT_VM_LGKM0(0);                            // the pad-step DMAs must land before the LDS goes away
    }
    __syncthreads();                                // the epilogue reuses the halo as scratch
    // ---- epilogue (identical to the fp32 kernel)
    // channel statistics in fp64 per lane: sums of fp32 values are then exact to ~1e-16, so the GroupNorm statistics a layer hands on do not
    // depend on which kernel variant (tile shape, fragment-to-lane mapping) produced them -- a garment's result is the same in any batch
    double ssum[NT], ssq[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) { ssum[u] = 0.0; ssq[u] = 0.0; }
    const bool full = z0 + TZ <= p.D && y0 + SP_TY <= p.H && x0 + SP_TX <= p.W;       // (workgroup-uniform)
    const bool interior = z0 > 0 && z0 + TZ < p.D && y0 > 0 && y0 + SP_TY < p.H && x0 > 0 && x0 + SP_TX < p.W;   // no voxel of the tile on a face
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int t = f & 1, gz = z0 + wave + 4 * (f >> 1);
            const int n = n0 + u * 32 + r;
            const float osn = p.out_scale[(int64_t)b * p.osc_bstride + n];
            const float osc = p.act_inv ? __fmul_rn(osn, p.act_inv[b]) : osn;
            const float *kb = p.kbias ? p.kbias + (int64_t)b * 64 * p.Cout + n : nullptr;
            if (full) {
                const int gy = y0 + t * 4, gx = x0 + 4 * h;
                float *ob = p.out + ((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * p.Cout + n;
                const float *pb = p.partial ? p.partial + ((((int64_t)b * (p.D >> 1) + (gz >> 1)) * (p.H >> 1) + (gy >> 1)) * (p.W >> 1) + (gx >> 1)) * (8 * p.Cout) + ((gz & 1) * 4 * p.Cout + n) : nullptr;
                sp_store_frag_full(p, tot[f][u], osc, ob, pb, ssum[u], ssq[u], kb, interior, gz, gy, gx);
                continue;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * h;
                const int gy = y0 + t * 4 + (i >> 3), gx = x0 + (i & 7);
                if (gz < p.D && gy < p.H && gx < p.W) {
                    float v;
                    v = __fmul_rn(tot[f][u][q], osc);
                    if (kb) v = __fadd_rn(v, kb[(int64_t)((sp_axis_mask(gz, p.D) * 4 + sp_axis_mask(gy, p.H)) * 4 + sp_axis_mask(gx, p.W)) * p.Cout]);
                    if (p.partial) v = __fadd_rn(v, p.partial[((((int64_t)b * (p.D >> 1) + (gz >> 1)) * (p.H >> 1) + (gy >> 1)) * (p.W >> 1) + (gx >> 1)) * (8 * p.Cout) + (((gz & 1) * 4 + (gy & 1) * 2 + (gx & 1)) * p.Cout + n)]);
                    if (p.relu) v = gn_relu(v);
                    p.out[((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * p.Cout + n] = v;
                    ssum[u] += (double)v;
                    ssq[u] += (double)v * (double)v;
                }
            }
        }
    if (p.osum) {
        double *red = reinterpret_cast<double *>(halo);
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const double s2 = ssum[u] + __shfl_xor(ssum[u], 32), q2 = ssq[u] + __shfl_xor(ssq[u], 32);
            if (h == 0) { red[wave * CT + u * 32 + r] = s2; red[4 * CT + wave * CT + u * 32 + r] = q2; }
        }
        __syncthreads();
        if (tid < CT) {
            const double s4 = red[tid] + red[CT + tid] + red[2 * CT + tid] + red[3 * CT + tid];
            const double q4 = red[4 * CT + tid] + red[5 * CT + tid] + red[6 * CT + tid] + red[7 * CT + tid];
            atomicAdd(&p.osum[(int64_t)b * p.Cout + n0 + tid], s4);
            atomicAdd(&p.osq[(int64_t)b * p.Cout + n0 + tid], q4);
        }
    }
}

// ------------------------------------------------------------------------------------------------ 32-wide "x-strip" variant (P = 2)
// The 32-wide layers (enc0c2 128 -> 32, dec2c1, dec2c2 at full resolution: 23 % of the step in round 2) through conv3d_split_kernel<1>
// read 6 KB of LDS fragments per 6 MFMAs -- one B fragment pair and one A fragment pair per tap serve a single 32-wide column block, so
// the LDS port is as busy as the matrix pipe (PMC: 8.5 LDS instructions per 512 MFMA-Mops, pipe 0.52 busy).  Here a wave's 32 fragment
// rows are the (z, y) positions of a 4 x 8 patch at ONE x, and the x direction lives in REGISTERS: for a (dz, dy) pair the wave loads a
// strip of 4 + 2 x-consecutive voxels once (6 x 2 planes) and the three dx taps' B fragments (3 x 2 planes), then runs the 4 x 3 x 3 = 36
// MFMAs of its four output x positions from registers -- 18 ds_read_b128 per 36 MFMAs instead of 36, one hand-over barrier per THREE taps
// instead of one per tap.  Per accumulator the products arrive in exactly the order of conv3d_split_kernel (taps ascending, planes
// (1,0) (0,1) (0,0), per-slice flush into `tot`): the results are BIT-IDENTICAL to it.
//   workgroup = 4 waves = 8 x 8 x 8 output voxels x 32 channels: wave w -> z half (w & 1), x half (w >> 1); halo 10 x 10 x 10 voxels
//   (1.95 staged voxels per output voxel instead of 2.34), two workgroups per CU (80 KB of LDS each).
//   LDS halo: voxel = 64 B (2 planes x 16 halfs), row pitch 656 B, z pitch 6784 B: with rows r = (zr, yr) the 16-lane service groups of
//   ds_read_b128 hit 16 distinct bank quads (row pitch = 9, z pitch = 8 quads mod 16; enumerated over the real lane groups).
//   B ring: two (dz, dy) groups of 6 KB; group G+1 is DMA'd right after the hand-over of group G.
struct StripLayout {
    static constexpr int VB = 64, RP = 10 * 64 + 16, ZP = 10 * RP + 224, HZ = 10, HY = 10, HX = 10, HVOX = 1000;
    static constexpr int BYTES = HZ * ZP;
    __device__ static constexpr __forceinline__ int at(int hz, int hy, int hx) { return hz * ZP + hy * RP + hx * VB; }
};

template <bool F16>
__global__ __launch_bounds__(256, 2) void conv3d_split_strip_kernel(SplitArgs p) {
    constexpr int P = 2, T = 8, NX = 4;
    using HL = StripLayout;
    constexpr int HALO_BYTES = HL::BYTES;
    constexpr int GB = 3 * P * 1024;                 // B bytes of one (dz, dy) group: 3 taps x 2 planes x 1 KB
    __shared__ __attribute__((aligned(16))) unsigned char smem[HALO_BYTES + 2 * GB];
    unsigned char *const halo = smem;
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + HALO_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int zw = 4 * (wave & 1), xw = NX * (wave >> 1);
    const int Cin = p.C0;
    const int ncb = p.Cout / 32;
    const int tiles_z = (p.D + T - 1) / T;
    int b, tile, cb;
    if (!sp_work_item(p, ncb, tiles_z * p.tiles_x * p.tiles_y, b, tile, cb)) return;      // (workgroup-uniform)
    const int tz = tile % tiles_z; tile /= tiles_z;
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile;
    const int z0 = tz * T, y0 = ty * T, x0 = tx * T;
    const int n0 = cb * 32;

    f32x16s acc[NX], tot[NX];
#pragma unroll
    for (int t = 0; t < NX; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc[t][q] = 0.f; tot[t][q] = 0.f; }

    {
    const int nslices = Cin / SP_KS, ngroups = nslices * 9;
    const int64_t bstep = (int64_t)ncb * P * 1024;                                    // bytes per (slice, tap) step of the pack
    // piece k of a group = tap 3g + (k >> 1), plane k & 1; wave w fetches pieces w and w + 4 (waves 0, 1 only)
    const unsigned char *bg0 = reinterpret_cast<const unsigned char *>(p.wp) + (int64_t)b * p.wp_bstride + (int64_t)cb * (P * 1024) + lane * 16;
    int G = 0;                                       // flat group index s * 9 + g of the group about to be multiplied
    auto issue_group = [&](int g) {
        const unsigned char *src = bg0 + (int64_t)(3 * g) * bstep;
        const unsigned dst = lds_ring + (g & 1) * GB;
        gn_glds16(src + (wave >> 1) * bstep + (wave & 1) * 1024, dst + wave * 1024);
        if (wave < 2) gn_glds16(src + 2 * bstep + wave * 1024, dst + (4 + wave) * 1024);
    };
    issue_group(0);

    const int abase = HL::at(zw + (r >> 3), r & 7, xw) + 16 * h;
    const unsigned char *const ring_rd = smem + HALO_BYTES + lane * 16;
    // staging: thread t owns channel quad c4 of voxels hv = (t >> 2) + 64 it, it = 0 .. 15
    const int c4 = (tid & 3) * 4;
    const float *const base0 = p.src0 + (int64_t)b * p.D * p.H * p.W * p.C0 + c4;
    for (int s = 0; s < nslices; ++s) {
        const int c0 = s * SP_KS;
        {
            const float4 av = *reinterpret_cast<const float4 *>(p.a + (int64_t)b * Cin + c0 + c4);
            const float4 dv = *reinterpret_cast<const float4 *>(p.d + (int64_t)b * Cin + c0 + c4);
            constexpr int NIT = (HL::HVOX + 63) / 64;                                // 16
            float4 raw[NIT];
            unsigned inb = 0;
            const int v0 = tid >> 2;
            int hx = v0 % 10, hy = v0 / 10, hz = 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
                const bool in = (v0 + 64 * it < HL::HVOX) && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
                raw[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (in) {
                    raw[it] = *reinterpret_cast<const float4 *>(base0 + (int64_t)((gz * p.H + gy) * p.W + gx) * p.C0 + c0);
                    inb |= 1u << it;
                }
                hx += 4; hy += 6;                    // + 64 voxels = (0, 6, 4) in (hz, hy, hx)
                if (hx >= 10) { hx -= 10; hy += 1; }
                if (hy >= 10) { hy -= 10; hz += 1; }
            }
            hx = v0 % 10; hy = v0 / 10; hz = 0;
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (v0 + 64 * it < HL::HVOX) {
                    float v0f = 0.f, v1f = 0.f, v2f = 0.f, v3f = 0.f;
                    if (inb & (1u << it)) {          // zero padding comes AFTER the affine
                        v0f = __fmaf_rn(raw[it].x, av.x, dv.x);
                        v1f = __fmaf_rn(raw[it].y, av.y, dv.y);
                        v2f = __fmaf_rn(raw[it].z, av.z, dv.z);
                        v3f = __fmaf_rn(raw[it].w, av.w, dv.w);
                    }
                    uint2 pl[P];
                    split4<P, F16>(v0f, v1f, v2f, v3f, pl);
                    unsigned char *dst = halo + HL::at(hz, hy, hx) + c4 * 2;
                    *reinterpret_cast<uint2 *>(dst) = pl[0];
                    *reinterpret_cast<uint2 *>(dst + 32) = pl[1];
                }
                hx += 4; hy += 6;
                if (hx >= 10) { hx -= 10; hy += 1; }
                if (hy >= 10) { hy -= 10; hz += 1; }
            }
        }
#pragma unroll
        for (int g = 0; g < 9; ++g, ++G) {
            // hand-over: this wave's share of group G's B fragments has landed (its only outstanding VM operations), its LDS reads of the
            // previous group are in registers and (g == 0) its halo stores are done -> after the barrier that holds for every wave: group G
            // is readable, slot (G + 1) & 1 is free, the halo is complete
            GN_WAIT_VM_LGKM0(0);
            __builtin_amdgcn_s_barrier();
            if (G + 1 < ngroups) issue_group(G + 1);
            uint4 bf[3][P], af[NX + 2][P];
            const unsigned char *rb = ring_rd + (G & 1) * GB;
            const int goff = HL::at(g / 3, g % 3, 0);
#pragma unroll
            for (int i = 0; i < 3; ++i) {            // the first output position's operands first
#pragma unroll
                for (int pl = 0; pl < P; ++pl) af[i][pl] = *reinterpret_cast<const uint4 *>(halo + abase + goff + i * HL::VB + pl * 32);
#pragma unroll
                for (int pl = 0; pl < P; ++pl) bf[i][pl] = *reinterpret_cast<const uint4 *>(rb + (i * P + pl) * 1024);
            }
#pragma unroll
            for (int i = 3; i < NX + 2; ++i)
#pragma unroll
                for (int pl = 0; pl < P; ++pl) af[i][pl] = *reinterpret_cast<const uint4 *>(halo + abase + goff + i * HL::VB + pl * 32);
#pragma unroll
            for (int xo = 0; xo < NX; ++xo)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {     // smallest terms first, as conv3d_split_kernel
                    acc[xo] = mfma16<F16>(af[xo + dx][1], bf[dx][0], acc[xo]);
                    acc[xo] = mfma16<F16>(af[xo + dx][0], bf[dx][1], acc[xo]);
                    acc[xo] = mfma16<F16>(af[xo + dx][0], bf[dx][0], acc[xo]);
                }
        }
#pragma unroll
        for (int t = 0; t < NX; ++t)
#pragma unroll
            for (int q = 0; q < 16; ++q) { tot[t][q] = __fadd_rn(tot[t][q], acc[t][q]); acc[t][q] = 0.f; }
        GN_WAIT_VM_LGKM0(0);
        __builtin_amdgcn_s_barrier();               // every wave is done reading this slice's halo: it can be overwritten
    }
    }
    __syncthreads();                                // the epilogue reuses the halo as scratch
    // ---- epilogue: fragment xo of this wave = rows (zr = q >> 2, yr = (q & 3) + 4 h) at x = x0 + xw + xo, lane r = channel
    double ssum = 0.0, ssq = 0.0;                   // fp64 per lane (see conv3d_split_kernel)
    const int n = n0 + r;
    const float osn = p.out_scale[(int64_t)b * p.osc_bstride + n];
            const float osc = p.act_inv ? __fmul_rn(osn, p.act_inv[b]) : osn;
    const bool full = z0 + T <= p.D && y0 + T <= p.H && x0 + T <= p.W;       // (workgroup-uniform)
    const bool interior = z0 > 0 && z0 + T < p.D && y0 > 0 && y0 + T < p.H && x0 > 0 && x0 + T < p.W;
    const float *kb = p.kbias ? p.kbias + (int64_t)b * 64 * p.Cout + n : nullptr;
    if (full) {
        const int64_t rs = (int64_t)p.W * p.Cout, zs_ = (int64_t)p.H * rs;
        const int gz0 = z0 + zw, gy0 = y0 + 4 * h;
#pragma unroll
        for (int xo = 0; xo < NX; ++xo) {
            const int gx = x0 + xw + xo;
            float *ob = p.out + ((((int64_t)b * p.D + gz0) * p.H + gy0) * p.W + gx) * p.Cout + n;
            float pv[16];
            if (p.partial) {                        // z0, y0, x0, zw, xw, 4 h are even: parity = (zr & 1, q & 1, xo & 1)
                const int64_t prs = (int64_t)(p.W >> 1) * 8 * p.Cout, pzs = (int64_t)(p.H >> 1) * prs;
                const float *pb = p.partial + ((((int64_t)b * (p.D >> 1) + (gz0 >> 1)) * (p.H >> 1) + (gy0 >> 1)) * (p.W >> 1) + (gx >> 1)) * (8 * p.Cout) + ((xo & 1) * p.Cout + n);
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int zr = q >> 2, yq = q & 3;
                    pv[q] = pb[(zr >> 1) * pzs + (yq >> 1) * prs + (int64_t)((zr & 1) * 4 + (yq & 1) * 2) * p.Cout];
                }
            }
            float kv[16];
            if (kb) {
                if (interior) {
                    const float k63 = kb[63 * (int64_t)p.Cout];
#pragma unroll
                    for (int q = 0; q < 16; ++q) kv[q] = k63;
                } else {
                    const int xmask = sp_axis_mask(gx, p.W);
#pragma unroll
                    for (int q = 0; q < 16; ++q) kv[q] = kb[(int64_t)((sp_axis_mask(gz0 + (q >> 2), p.D) * 4 + sp_axis_mask(gy0 + (q & 3), p.H)) * 4 + xmask) * p.Cout];
                }
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                float v = __fmul_rn(tot[xo][q], osc);
                if (kb) v = __fadd_rn(v, kv[q]);
                if (p.partial) v = __fadd_rn(v, pv[q]);
                if (p.relu) v = gn_relu(v);
                ob[(q >> 2) * zs_ + (q & 3) * rs] = v;
                ssum += (double)v;
                ssq += (double)v * (double)v;
            }
        }
    } else {
#pragma unroll
        for (int xo = 0; xo < NX; ++xo)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int gz = z0 + zw + (q >> 2), gy = y0 + 4 * h + (q & 3), gx = x0 + xw + xo;
                if (gz < p.D && gy < p.H && gx < p.W) {
                    float v;
                    v = __fmul_rn(tot[xo][q], osc);
                    if (kb) v = __fadd_rn(v, kb[(int64_t)((sp_axis_mask(gz, p.D) * 4 + sp_axis_mask(gy, p.H)) * 4 + sp_axis_mask(gx, p.W)) * p.Cout]);
                    if (p.partial) v = __fadd_rn(v, p.partial[((((int64_t)b * (p.D >> 1) + (gz >> 1)) * (p.H >> 1) + (gy >> 1)) * (p.W >> 1) + (gx >> 1)) * (8 * p.Cout) + (((gz & 1) * 4 + (gy & 1) * 2 + (gx & 1)) * p.Cout + n)]);
                    if (p.relu) v = gn_relu(v);
                    p.out[((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * p.Cout + n] = v;
                    ssum += (double)v;
                    ssq += (double)v * (double)v;
                }
            }
    }
    if (p.osum) {
        double *red = reinterpret_cast<double *>(halo);
        const double s2 = ssum + __shfl_xor(ssum, 32), q2 = ssq + __shfl_xor(ssq, 32);
        if (h == 0) { red[wave * 32 + r] = s2; red[128 + wave * 32 + r] = q2; }
        __syncthreads();
        if (tid < 32) {
            const double s4 = red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid];
            const double q4 = red[128 + tid] + red[160 + tid] + red[192 + tid] + red[224 + tid];
            atomicAdd(&p.osum[(int64_t)b * p.Cout + n0 + tid], s4);
            atomicAdd(&p.osq[(int64_t)b * p.Cout + n0 + tid], q4);
        }
    }
}

// ------------------------------------------------------------------------------------------------ 128-wide variant (P = 2)
// For Cout % 128 == 0 (58 % of the UNet's FLOPs: the 128 -> 128 layers at full resolution).  512 threads = 8 waves: wave =
// (z-slice zs, column group cg), the two column groups share ONE halo, so the GroupNorm-affine + plane-split staging is done once
// per 128 output channels instead of once per 64.  The halo is DOUBLE-buffered (2 x 48 KB + 32 KB ring = 126 KB, one workgroup
// per CU = 2 waves per SIMD as before): slice s+1 is staged WHILE slice s is multiplied -- its 5 row loads per thread are issued at
// tap 0 (inline asm, so that hipcc's own vmcnt bookkeeping does not drain the fragment DMAs) and converted / written one per tap
// at taps 8-12, ordered by the per-tap barriers alone.  The matrix-core stream never stops for staging.
typedef float f32x4w __attribute__((ext_vector_type(4)));

template <int P, bool F16>
__global__ __launch_bounds__(512, 1) void conv3d_split_wide_kernel(SplitArgs p) {
    static_assert(P == 2, "the wide variant is sized for the two-plane modes");
    constexpr int NT = 2;
    constexpr int TZ = SP_TZ, HZ = TZ + 2, HVOX = HZ * SP_HY * SP_HX;
    constexpr int CW = 128;                         // output channels per workgroup
    using HL = HaloLayout<P, HZ>;
    constexpr int HALO_BYTES = HL::BYTES;
    constexpr int NPIECE = 2 * NT * P;              // 1-KB B fragments per step: both column groups
    constexpr int BTAP = NPIECE * 1024;
    constexpr int DEPTH = 4, CH = 1;                // one piece per wave per step
    constexpr int NIT = (HVOX * 4 + 511) / 512;     // 5 row loads per thread per slice
    constexpr int AD_OFF = 2 * HALO_BYTES + DEPTH * BTAP;
    __shared__ __attribute__((aligned(16))) unsigned char smem[AD_OFF + 2 * 384 * 4];
    const unsigned lds_ring = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + 2 * HALO_BYTES;
    float *const adl = reinterpret_cast<float *>(smem + AD_OFF);          // a[Cin] | d[Cin] of this sample
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), zs = wave & 3, cg = wave >> 2;
    const int Cin = p.C0 + p.C1;
    const int ncb = p.Cout / CW;
    const int tiles_z = (p.D + TZ - 1) / TZ;
    int b, tile, cb;
    if (!sp_work_item(p, ncb, tiles_z * p.tiles_x * p.tiles_y, b, tile, cb)) return;      // (workgroup-uniform)
    const int tz = tile % tiles_z; tile /= tiles_z;
    const int tx = tile % p.tiles_x; tile /= p.tiles_x;
    const int ty = tile;
    const int z0 = tz * TZ, y0 = ty * SP_TY, x0 = tx * SP_TX;
    const int n0 = cb * 128 + cg * 64;
    const int zl = zs;                              // this wave's z-slice inside the tile
    const int D1 = p.D >> 1, H1 = p.H >> 1, W1 = p.W >> 1;
    const int nslices = Cin / SP_KS;

    f32x16s acc[2][NT], tot[2][NT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc[t][u][q] = 0.f; tot[t][u][q] = 0.f; }

    {
    for (int i = tid; i < Cin; i += 512) { adl[i] = p.a[(int64_t)b * Cin + i]; adl[384 + i] = p.d[(int64_t)b * Cin + i]; }

    const int64_t bstep = (int64_t)(p.Cout / 32) * P * 1024;
    const int piece = wave;
    const unsigned char *bg = reinterpret_cast<const unsigned char *>(p.wp) + (int64_t)b * p.wp_bstride + (int64_t)cb * BTAP + piece * 1024 + lane * 16;
    int jf = 0;
#define SPW_ISSUE_B()                                                                                                          \
    do {                                                                                                                       \
        gn_glds16(bg, lds_ring + (jf & (DEPTH - 1)) * BTAP + piece * 1024);                                                    \
        bg += bstep; ++jf;                                                                                                     \
    } while (0)
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i) SPW_ISSUE_B();   // step j+DEPTH-1 is issued at tap j, after its barrier

    // ---- staging of one 16-channel slice, split in two halves that can be far apart in time
    const int c4 = (tid & 3) * 4;                   // 512 % 4 == 0: the same channel quad every iteration
    f32x4w raw[NIT];
    unsigned inb = 0;                               // bit it: the voxel of iteration it lies inside the volume
    // rows of the full-resolution source: the byte offset of (voxel, channel quad) inside this sample fits 32 bits (checked on the host
    // side of the launch: D*H*W*C0*4 < 2^32), so a row costs ONE register across the MFMA loop and one add per slice
    // (global_load saddr form: 64-bit scalar base + 32-bit vector offset); rows outside the volume read offset 0 and are masked later
    unsigned voff[NIT], inb0 = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int idx = tid + it * 512;
        const int hv = (idx < HVOX * 4 ? idx : HVOX * 4 - 1) >> 2;
        const int hx = hv % SP_HX, hy = (hv / SP_HX) % SP_HY, hz = hv / (SP_HX * SP_HY);
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool in = idx < HVOX * 4 && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        voff[it] = in ? ((unsigned)((gz * p.H + gy) * p.W + gx) * (unsigned)p.C0 + (unsigned)c4) * 4u : (unsigned)c4 * 4u;
        if (in) inb0 |= 1u << it;
    }
    const float *const base0 = p.src0 + (int64_t)b * p.D * p.H * p.W * p.C0;
    auto issue_rows = [&](int sl) {
        const int c0 = sl * SP_KS;
        if (c0 < p.C0) {
            inb = inb0;
            const unsigned cb4 = (unsigned)c0 * 4u;
#pragma unroll
            for (int it = 0; it < NIT; ++it)
                asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(raw[it]) : "v"(voff[it] + cb4), "s"(base0) : "memory");
            return;
        }
        // the half-resolution (nearest-upsampled) source: only reached with Arith.polyphase_upconv off; offsets re-derived per slice
        const int cs = c0 - p.C0;
        inb = 0;
        int tl = tid;
        asm volatile("" : "+v"(tl));
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = tl + it * 512;
            const int hv = (idx < HVOX * 4 ? idx : HVOX * 4 - 1) >> 2;
            const int hx = hv % SP_HX, hy = (hv / SP_HX) % SP_HY, hz = hv / (SP_HX * SP_HY);
            const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
            const bool in = idx < HVOX * 4 && gz >= 0 && gz < p.D && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            int64_t off = (int64_t)b * D1 * H1 * W1 * p.C1;                      // a valid address when outside
            if (in) {
                off = ((((int64_t)b * D1 + (gz >> 1)) * H1 + (gy >> 1)) * W1 + (gx >> 1)) * p.C1 + cs + c4;
                inb |= 1u << it;
            }
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(raw[it]) : "v"(p.src1 + off) : "memory");
        }
    };
    auto convert_row = [&](int it, int sl, int buf) {
        int tl = tid;
        asm volatile("" : "+v"(tl));                // (as above: the halo address of the row is recomputed, not kept)
        const int idx = tl + it * 512;
        asm volatile("" : "+v"(raw[it]));           // the loads above are invisible to hipcc's waitcnt pass: pin the first use here
        if (idx < HVOX * 4) {
            const int hv = idx >> 2;
            const float4 av = *reinterpret_cast<const float4 *>(adl + sl * SP_KS + c4);
            const float4 dv = *reinterpret_cast<const float4 *>(adl + 384 + sl * SP_KS + c4);
            float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
            if (inb & (1u << it)) {                  // zero padding comes AFTER the affine
                v0 = __fmaf_rn(raw[it].x, av.x, dv.x);
                v1 = __fmaf_rn(raw[it].y, av.y, dv.y);
                v2 = __fmaf_rn(raw[it].z, av.z, dv.z);
                v3 = __fmaf_rn(raw[it].w, av.w, dv.w);
            }
            uint2 pl[P];
            split4<P, F16>(v0, v1, v2, v3, pl);
#pragma unroll
            for (int i = 0; i < P; ++i) *reinterpret_cast<uint2 *>(smem + buf * HALO_BYTES + HL::at(hv / (SP_HX * SP_HY), (hv / SP_HX) % SP_HY, hv % SP_HX) + i * 32 + c4 * 2) = pl[i];
        }
    };

    // slice 0 synchronously
    issue_rows(0);
    GN_WAIT_VM_LGKM0(0);
    __syncthreads();                                // a / d table visible
#pragma unroll
    for (int it = 0; it < NIT; ++it) convert_row(it, 0, 0);
    __syncthreads();

    const int abase = HL::at(zl, r >> 3, r & 7) + 16 * h;
    constexpr int AF1 = 4 * HL::ROWP;
    const unsigned char *const ring_rd = smem + 2 * HALO_BYTES + cg * NT * P * 1024 + lane * 16;
    int jcur = 0;
    uint4 bf[NT][P];                                // step 0's fragments (the prologue DMAs were drained with the slice-0 staging)
#pragma unroll
    for (int u = 0; u < NT; ++u)
#pragma unroll
        for (int i = 0; i < P; ++i) bf[u][i] = *reinterpret_cast<const uint4 *>(ring_rd + (u * P + i) * 1024);
    for (int s = 0; s < nslices; ++s) {
        const unsigned char *const halo = smem + (s & 1) * HALO_BYTES;
        const int sn = s + 1 < nslices ? s + 1 : s;
        uint4 a0[P], a1[P], na0[P], na1[P];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            na0[i] = *reinterpret_cast<const uint4 *>(halo + abase + i * 32);
            na1[i] = *reinterpret_cast<const uint4 *>(halo + abase + AF1 + i * 32);
        }
#pragma unroll
        for (int tap = 0; tap < 27; ++tap, ++jcur) {
#pragma unroll
            for (int i = 0; i < P; ++i) { a0[i] = na0[i]; a1[i] = na1[i]; }
            // hand-over: step j+1's fragments have landed for everybody (they are read at the END of this tap, into the registers the
            // MFMAs of this tap have just consumed: no register double buffer, no LDS latency after the barrier) and everybody is done
            // with step j-1's slot.  VM queue, oldest first: fragment steps j+1 .. j+DEPTH-2 and, for taps 1-2, the NIT row loads
            // issued at tap 0 (younger than the DMA of tap 0 = step j0+DEPTH-1): they may still be in flight there
            // (vmcnt only: the B fragments of this step were requested a few instructions ago, at the end of the previous tap -- hipcc
            //  waits for each of them right before the MFMA that consumes it; a blanket lgkmcnt(0) here would expose their LDS
            //  latency at every barrier.  The slot the DMA below overwrites, step j-1's, was consumed by MFMAs every wave has issued.
            //  Tap 26 drains the LDS queue once per slice so that the staged rows of the next slice are visible after its barriers.)
            if (tap == 26) GN_WAIT_VM_LGKM0((DEPTH - 3) * CH);
            else if (tap >= 1 && tap <= 2) GN_WAIT_VM_ONLY((DEPTH - 3) * CH + NIT);
            else GN_WAIT_VM_ONLY((DEPTH - 3) * CH);
            __builtin_amdgcn_s_barrier();
            SPW_ISSUE_B();                          // step j+DEPTH-1 -> the slot step j-1 vacated one tap ago
            if (tap + 1 < 27) {
                const int t1 = tap + 1;
                const int toff = HL::at(t1 / 9, (t1 / 3) % 3, t1 % 3);
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    na0[i] = *reinterpret_cast<const uint4 *>(halo + abase + toff + i * 32);
                    na1[i] = *reinterpret_cast<const uint4 *>(halo + abase + AF1 + toff + i * 32);
                }
            }
            if (tap == 0) issue_rows(sn);                                        // always (uniform wait counts); unused after the last slice
            __builtin_amdgcn_sched_barrier(0);
#define SPW_PROD(IA, IB)                                                                                                       \
            _Pragma("unroll") for (int u = 0; u < NT; ++u) {                                                                   \
                acc[0][u] = mfma16<F16>(a0[IA], bf[u][IB], acc[0][u]);                                                         \
                acc[1][u] = mfma16<F16>(a1[IA], bf[u][IB], acc[1][u]);                                                         \
            }
            SPW_PROD(1, 0)
            if (tap >= 8 && tap < 8 + NIT && s + 1 < nslices) convert_row(tap - 8, sn, (s + 1) & 1);
            SPW_PROD(0, 1) SPW_PROD(0, 0)
#undef SPW_PROD
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int i = 0; i < P; ++i) bf[u][i] = *reinterpret_cast<const uint4 *>(ring_rd + ((jcur + 1) & (DEPTH - 1)) * BTAP + (u * P + i) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int q = 0; q < 16; ++q) { tot[t][u][q] = __fadd_rn(tot[t][u][q], acc[t][u][q]); acc[t][u][q] = 0.f; }
    }
#undef SPW_ISSUE_B
    GN_WAIT_VM_LGKM0(0);
    }
    __syncthreads();                                // pad-step DMAs landed; the epilogue reuses the halo as scratch
    const int gz = z0 + zl;
    // channel statistics in fp64 per lane: sums of fp32 values are then exact to ~1e-16, so the GroupNorm statistics a layer hands on do not
    // depend on which kernel variant (tile shape, fragment-to-lane mapping) produced them -- a garment's result is the same in any batch
    double ssum[NT], ssq[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) { ssum[u] = 0.0; ssq[u] = 0.0; }
    const bool full = z0 + TZ <= p.D && y0 + SP_TY <= p.H && x0 + SP_TX <= p.W;       // (workgroup-uniform)
    const bool interior = z0 > 0 && z0 + TZ < p.D && y0 > 0 && y0 + SP_TY < p.H && x0 > 0 && x0 + SP_TX < p.W;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int n = n0 + u * 32 + r;
            const float osn = p.out_scale[(int64_t)b * p.osc_bstride + n];
            const float osc = p.act_inv ? __fmul_rn(osn, p.act_inv[b]) : osn;
            const float *kb = p.kbias ? p.kbias + (int64_t)b * 64 * p.Cout + n : nullptr;
            if (full) {
                const int gy = y0 + t * 4, gx = x0 + 4 * h;
                float *ob = p.out + ((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * p.Cout + n;
                const float *pb = p.partial ? p.partial + ((((int64_t)b * (p.D >> 1) + (gz >> 1)) * (p.H >> 1) + (gy >> 1)) * (p.W >> 1) + (gx >> 1)) * (8 * p.Cout) + ((gz & 1) * 4 * p.Cout + n) : nullptr;
                sp_store_frag_full(p, tot[t][u], osc, ob, pb, ssum[u], ssq[u], kb, interior, gz, gy, gx);
                continue;
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * h;
                const int gy = y0 + t * 4 + (i >> 3), gx = x0 + (i & 7);
                if (gz < p.D && gy < p.H && gx < p.W) {
                    float v;
                    v = __fmul_rn(tot[t][u][q], osc);
                    if (kb) v = __fadd_rn(v, kb[(int64_t)((sp_axis_mask(gz, p.D) * 4 + sp_axis_mask(gy, p.H)) * 4 + sp_axis_mask(gx, p.W)) * p.Cout]);
                    if (p.partial) v = __fadd_rn(v, p.partial[((((int64_t)b * (p.D >> 1) + (gz >> 1)) * (p.H >> 1) + (gy >> 1)) * (p.W >> 1) + (gx >> 1)) * (8 * p.Cout) + (((gz & 1) * 4 + (gy & 1) * 2 + (gx & 1)) * p.Cout + n)]);
                    if (p.relu) v = gn_relu(v);
                    p.out[((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * p.Cout + n] = v;
                    ssum[u] += (double)v;
                    ssq[u] += (double)v * (double)v;
                }
            }
        }
    if (p.osum) {
        double *red = reinterpret_cast<double *>(smem);                         // [sum | sq][cg][zs][64]
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const double s2 = ssum[u] + __shfl_xor(ssum[u], 32), q2 = ssq[u] + __shfl_xor(ssq[u], 32);
            if (h == 0) { red[(cg * 4 + zs) * 64 + u * 32 + r] = s2; red[512 + (cg * 4 + zs) * 64 + u * 32 + r] = q2; }
        }
        __syncthreads();
        if (tid < 128) {
            const int g = tid >> 6, c = tid & 63;
            const double *rs = red + g * 256 + c, *rq = red + 512 + g * 256 + c;
            const double s4 = rs[0] + rs[64] + rs[128] + rs[192];
            const double q4 = rq[0] + rq[64] + rq[128] + rq[192];
            atomicAdd(&p.osum[(int64_t)b * p.Cout + cb * 128 + tid], s4);
            atomicAdd(&p.osq[(int64_t)b * p.Cout + cb * 128 + tid], q4);
        }
    }
}


// ------------------------------------------------------------------------------------------------ occupancy-aware launch: list + fill
// tile_active holds one flag per 4 x 8 x 8 tile (gn_grid_tile_flags).  The kernels above visit only the ACTIVE tiles, through a compact,
// ascending list at their own tile granularity (pair = 1: 4 x 8 x 8; pair = 2: the x-strip kernel's 8 x 8 x 8 blocks = two flags along z),
// built here by ONE workgroup (B * tiles <= a few 10^5 entries: ~20 us; ballot / popcount scan, deterministic order).
__global__ __launch_bounds__(1024) void occ_compact_kernel(const unsigned char *__restrict__ flags, int B, int tz4, int tyx, int pair,
                                                           int *__restrict__ list, int *__restrict__ count) {
    const int Tz = pair == 2 ? (tz4 + 1) / 2 : tz4, Tk = tyx * Tz, total = B * Tk;
    __shared__ int wsum[16];
    __shared__ int running;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) running = 0;
    __syncthreads();
    for (int base = 0; base < total; base += 1024) {
        const int i = base + tid;
        bool act = false;
        if (i < total) {
            const int b = i / Tk, t = i - b * Tk, yx = t / Tz, z = t - yx * Tz;
            const unsigned char *f = flags + ((int64_t)b * tyx + yx) * tz4;
            act = pair == 2 ? (f[2 * z] != 0 || (2 * z + 1 < tz4 && f[2 * z + 1] != 0)) : f[z] != 0;
        }
        const unsigned long long m = __ballot(act);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (act) list[off + before] = i;
        __syncthreads();
        if (tid == 0) {
            int tsum = 0;
            for (int w = 0; w < 16; ++w) tsum += wsum[w];
            running += tsum;
        }
        __syncthreads();
    }
    if (tid == 0) *count = running;
}

// Every INACTIVE tile: nothing but the finished border-class constants a dense launch would store there (kconst), plus their share of the
// epilogue statistics.  One 256-thread workgroup per (y, x) COLUMN of tiles and sample: it walks the column's tiles (at the conv kernel's tile
// granularity), skips the active ones and accumulates the statistics of the rest in registers -- one reduction and one set of atomics per
// column instead of per tile.  A thread owns one channel quad and every (256 / quads)-th voxel: float4 stores, 512 contiguous bytes per voxel
// for 128 channels.  Tiles away from the faces hold ONE value per channel: their statistics are taken in closed form, count * v and
// count * v^2 in fp64.  count * v is exact (24-bit v, count <= 2^6 per thread); count * v^2 is NOT always (a 48-bit square times up to
// 2^6 needs 54 bits), and the fp64 atomics that merge workgroups land in whatever order the hardware schedules them.  The statistics of an
// occupancy-aware launch therefore agree with the dense launch's to fp64 rounding (~1e-16 relative), not bit for bit -- what IS bit-identical
// to the dense launch is this layer's OUTPUT (tests/test_gpu_parity.py::test_sparse_first_conv_is_bit_identical_to_dense compares the
// conv output; a following layer's fp32 GroupNorm coefficients can differ by an ulp).  HBM-bound.
__global__ __launch_bounds__(256) void conv_fill_inactive_kernel(SplitArgs p, const unsigned char *__restrict__ flags, int pair) {
    const int TZ = SP_TZ * pair, tz4 = (p.D + SP_TZ - 1) / SP_TZ, tiles_z = (p.D + TZ - 1) / TZ;
    const int tx = blockIdx.x % p.tiles_x, ty = blockIdx.x / p.tiles_x, b = blockIdx.y;
    const unsigned char *fl = flags + ((int64_t)b * p.tiles_y * p.tiles_x + (int64_t)ty * p.tiles_x + tx) * tz4;
    const int y0 = ty * SP_TY, x0 = tx * SP_TX, r = p.kreach, nc = 2 * r + 1;
    const int quads = p.Cout >> 2, tid = threadIdx.x;                           // Cout <= 1024 (checked by the caller): quads <= 256
    const int vpp = 256 / quads, quad = tid % quads, vl = tid / quads;          // voxels per pass; this thread's quad / voxel lane
    const bool yx_inner = y0 >= r && y0 + SP_TY <= p.H - r && x0 >= r && x0 + SP_TX <= p.W - r;
    const float *kc = p.kconst + (int64_t)b * (nc * nc * nc) * p.Cout + quad * 4;
    const float4 centre = *reinterpret_cast<const float4 *>(kc + (int64_t)((r * nc + r) * nc + r) * p.Cout);
    double s4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
    if (vl < vpp) {
        for (int tz = 0; tz < tiles_z; ++tz) {
            if (fl[pair * tz] || (pair == 2 && pair * tz + 1 < tz4 && fl[pair * tz + 1])) continue;      // active: the conv kernel's
            const int z0 = tz * TZ;
            const bool uniform = yx_inner && z0 >= r && z0 + TZ <= p.D - r;
            float *ob = p.out + ((((int64_t)b * p.D + z0) * p.H + y0) * p.W + x0) * p.Cout + quad * 4;
            if (uniform) {
                int cnt = 0;
                for (int v = vl; v < TZ * 64; v += vpp, ++cnt)
                    *reinterpret_cast<float4 *>(ob + (((int64_t)(v >> 6) * p.H + ((v >> 3) & 7)) * p.W + (v & 7)) * p.Cout) = centre;
                const double n = (double)cnt;
                s4[0] += n * (double)centre.x; s4[1] += n * (double)centre.y; s4[2] += n * (double)centre.z; s4[3] += n * (double)centre.w;
                q4[0] += n * ((double)centre.x * (double)centre.x); q4[1] += n * ((double)centre.y * (double)centre.y);
                q4[2] += n * ((double)centre.z * (double)centre.z); q4[3] += n * ((double)centre.w * (double)centre.w);
                continue;
            }
            for (int v = vl; v < TZ * 64; v += vpp) {
                const int gz = z0 + (v >> 6), gy = y0 + ((v >> 3) & 7), gx = x0 + (v & 7);
                if (gz >= p.D || gy >= p.H || gx >= p.W) continue;
                const int cls = (sp_axis_class(gz, p.D, r) * nc + sp_axis_class(gy, p.H, r)) * nc + sp_axis_class(gx, p.W, r);
                const float4 cv = *reinterpret_cast<const float4 *>(kc + (int64_t)cls * p.Cout);
                *reinterpret_cast<float4 *>(p.out + ((((int64_t)b * p.D + gz) * p.H + gy) * p.W + gx) * p.Cout + quad * 4) = cv;
                s4[0] += (double)cv.x; s4[1] += (double)cv.y; s4[2] += (double)cv.z; s4[3] += (double)cv.w;
                q4[0] += (double)cv.x * (double)cv.x; q4[1] += (double)cv.y * (double)cv.y; q4[2] += (double)cv.z * (double)cv.z; q4[3] += (double)cv.w * (double)cv.w;
            }
        }
    }
    if (!p.osum) return;
    __shared__ double red[256][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[tid][i] = s4[i]; red[tid][4 + i] = q4[i]; }
    __syncthreads();
    if (vl == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            double a = 0.0, c = 0.0;
            for (int k = 0; k < vpp; ++k) { a += red[tid + k * quads][i]; c += red[tid + k * quads][4 + i]; }
            if (a != 0.0 || c != 0.0) {
                atomicAdd(&p.osum[(int64_t)b * p.Cout + quad * 4 + i], a);
                atomicAdd(&p.osq[(int64_t)b * p.Cout + quad * 4 + i], c);
            }
        }
    }
}

extern "C" size_t gn_conv3d_occupancy_workspace_bytes(int B, int D, int H, int W) {
    if (B < 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    return ((size_t)B * gn_cdiv(D, SP_TZ) * gn_cdiv(H, SP_TY) * gn_cdiv(W, SP_TX) + 16) * sizeof(int);
}

static int conv3d_gcr_split_impl(const float *src0, int C0, const float *src1, int C1, const float *a, const float *d,
                                 const void *wp_planes, int mode, const float *out_scale, const float *act_inv_scale, int B, int D, int H,
                                 int W, int Cout, int relu, float *out, double *out_sum, double *out_sumsq, const unsigned char *tile_active,
                                 const float *kconst, int kreach, const float *partial, const float *kbias, int64_t wp_bstride, int osc_bstride,
                                 void *occ_ws, size_t occ_ws_bytes, void *stream, int wino = 0) {
    GN_REQUIRE(B >= 0 && D > 0 && H > 0 && W > 0 && C0 > 0 && C1 >= 0 && Cout > 0, "gn_conv3d_gcr_split: bad sizes");
    GN_REQUIRE(mode == GN_SPLIT_BF16X2 || mode == GN_SPLIT_BF16X3 || mode == GN_SPLIT_F16X2, "gn_conv3d_gcr_split: mode must be GN_SPLIT_BF16X2, _BF16X3 or _F16X2");
    GN_REQUIRE(out_scale != nullptr, "gn_conv3d_gcr_split: out_scale [Cout] is required (ones for the bf16 modes)");
    GN_REQUIRE(C0 % SP_KS == 0 && C1 % SP_KS == 0 && Cout % 32 == 0, "gn_conv3d_gcr_split: channel counts must be multiples of 16 (in) / 32 (out)");
    GN_REQUIRE(C1 == 0 || (src1 && D % 2 == 0 && H % 2 == 0 && W % 2 == 0), "gn_conv3d_gcr_split: upsampled source needs even dims");
    GN_REQUIRE((out_sum == nullptr) == (out_sumsq == nullptr), "gn_conv3d_gcr_split: out_sum and out_sumsq must come together");
    if (B == 0) return GN_OK;
    hipStream_t st = gn_stream(stream);
    if (out_sum) {
        GN_HIP(gn_zero_stats(out_sum, out_sumsq, (size_t)B * Cout, st), "gn_conv3d_gcr_split");
    }
    SplitArgs p;
    p.src0 = src0; p.src1 = src1; p.a = a; p.d = d; p.wp = (const uint4 *)wp_planes; p.out = out; p.osum = out_sum; p.osq = out_sumsq;
    p.C0 = C0; p.C1 = C1; p.B = B; p.D = D; p.H = H; p.W = W; p.Cout = Cout; p.relu = relu; p.out_scale = out_scale; p.act_inv = act_inv_scale;
    p.active_list = nullptr; p.active_count = nullptr; p.kconst = kconst; p.kreach = kreach; p.partial = partial;
    p.kbias = kbias; p.wp_bstride = wp_bstride; p.osc_bstride = osc_bstride; p.chain = 1;
    GN_REQUIRE(!partial || (D % 2 == 0 && H % 2 == 0 && W % 2 == 0), "gn_conv3d_gcr_split: a polyphase partial needs even dims");
    const int tz = (int)gn_cdiv(D, SP_TZ);
    p.tiles_y = (int)gn_cdiv(H, SP_TY);
    p.tiles_x = (int)gn_cdiv(W, SP_TX);
    const int tiles = tz * p.tiles_y * p.tiles_x;
    const int Cin_total = C0 + C1;
    const bool fits32 = (int64_t)D * H * W * C0 * 4 < ((int64_t)1 << 32);
    // 128-wide variant: two-plane modes, Cout % 128 == 0, at least two workgroups per CU's worth of work (its row loads address one sample of
    // the full-resolution source with 32-bit byte offsets)
    const bool wide128 = !wino && mode != GN_SPLIT_BF16X3 && Cout % 128 == 0 && Cin_total <= 384 && fits32 && (int64_t)tiles * (Cout / 128) * B >= 512;
    // Winograd F(2,3)-along-x form of the 128-wide kernel (unet_wino.hip): its own pack order (36 steps per slice), whole tiles only
    // ... and of the 32-wide column-block kernel (unet_wino32.hip: Cout % 128 != 0): 8 x 8 x 8 tiles, Cin <= 128, a polyphase partial allowed
    const bool wino32 = wino && Cout % 128 != 0;
    GN_REQUIRE(!wino || wino32 || (mode == GN_SPLIT_F16X2 && C1 == 0 && !partial && Cout % 128 == 0 && C0 <= 256 && fits32 && D % SP_TZ == 0 && H % SP_TY == 0 &&
                                   W % SP_TX == 0 && (int64_t)D * H * W <= ((int64_t)1 << 27)),
               "gn_conv3d_gcr_split_wino: needs one source, Cin <= 256, Cout %% 128 == 0, D %% 4 == H %% 8 == W %% 8 == 0, D*H*W <= 2^27 and "
               "D*H*W*Cin*4 < 2^32");
    GN_REQUIRE(!wino32 || (mode == GN_SPLIT_F16X2 && C1 == 0 && C0 <= 128 && fits32 && D % 8 == 0 && H % 8 == 0 && W % 8 == 0 && (int64_t)D * H * W <= ((int64_t)1 << 27)),
               "gn_conv3d_gcr_split_wino (Cout %% 128 != 0): needs one source, Cin <= 128, D %% 8 == H %% 8 == W %% 8 == 0, D*H*W <= 2^27 and D*H*W*Cin*4 < 2^32");
    // x-strip variant (below) also for the 64-wide layers with one full-resolution source: two 32-wide column blocks through the strip kernel
    // (18 fragment reads per 36 MFMAs) beat the 64-wide form of conv3d_split_kernel (8 per 12) by 4-7 % on every such shape of the UNet, the
    // halo staged twice notwithstanding (profiles/r04_ab_experiments.txt); bit-identical outputs
    const int tiles8 = (int)gn_cdiv(D, 8) * p.tiles_y * p.tiles_x;
    const bool strip = !wino && mode != GN_SPLIT_BF16X3 && !wide128 && C1 == 0 && (int64_t)tiles8 * (Cout / 32) * B >= 512;
    const bool wide = !wino && !wide128 && !strip && (Cout % 64 == 0) && ((int64_t)tiles * (Cout / 64) * B >= 1024);
    // (MT = 2, the tall 8 x 8 x 8 tile, was measured on the 32-wide layers: 327-347 TFLOP/s vs 340 for MT = 1 -- its synchronous
    //  staging phase is 29 % of the kernel -- so it is not dispatched)
#define SP_LAUNCH(P_, F16_)                                                                                                    \
    do {                                                                                                                       \
        if (wide) hipLaunchKernelGGL((conv3d_split_kernel<2, P_, F16_, 1>), dim3(tiles * (Cout / 64), B), dim3(256), 0, st, p); \
        else hipLaunchKernelGGL((conv3d_split_kernel<1, P_, F16_, 1>), dim3(tiles * (Cout / 32), B), dim3(256), 0, st, p);     \
        gn_note_kernel(wide ? "conv3d_split_kernel<2, " #P_ ", " #F16_ ", 1>" : "conv3d_split_kernel<1, " #P_ ", " #F16_ ", 1>"); \
    } while (0)
    // The variant is chosen from the SHAPE alone -- an occupancy-aware launch (tile_active) takes the kernel the dense launch of the same
    // shape takes, so the two give bit-identical outputs AND the same per-tile fp32 partials of the epilogue statistics.  (Outputs are
    // bit-identical across all variants anyway: per element the products arrive in the same order; the border-class constants of an
    // occupancy-aware layer may therefore come from a launch over a tiny volume, whichever variant that takes.)
    GN_REQUIRE((tile_active == nullptr) == (kconst == nullptr), "gn_conv3d_gcr_split: tile_active and kconst come together");
    GN_REQUIRE(!(tile_active && partial), "gn_conv3d_gcr_split: the occupancy-aware launch cannot take a polyphase partial (inactive tiles are "
                                          "filled with the border-class constants alone: the partial would be dropped there)");
    GN_REQUIRE(!tile_active || ((kreach == 1 || kreach == 2) && mode != GN_SPLIT_BF16X3 && D > 2 * kreach && H > 2 * kreach && W > 2 * kreach),
               "gn_conv3d_gcr_split: the occupancy-aware launch needs a two-plane mode, kreach 1 or 2 and dims > 2 kreach");
    if (tile_active) {
        // the active tiles as a compact list at the chosen kernel's tile granularity + the inactive tiles' constants (and their statistics)
        GN_REQUIRE(occ_ws && occ_ws_bytes >= gn_conv3d_occupancy_workspace_bytes(B, D, H, W) && ((uintptr_t)occ_ws & 3) == 0,
                   "gn_conv3d_gcr_split: the occupancy-aware launch needs gn_conv3d_occupancy_workspace_bytes(B, D, H, W) bytes of workspace");
        GN_REQUIRE(Cout % 4 == 0 && Cout <= 1024, "gn_conv3d_gcr_split: the occupancy-aware launch needs Cout <= 1024");
        const int pair = (strip || wino32) ? 2 : 1;
        int *count = (int *)occ_ws, *list = count + 16;
        hipLaunchKernelGGL(occ_compact_kernel, dim3(1), dim3(1024), 0, st, tile_active, B, tz, p.tiles_y * p.tiles_x, pair, list, count);
        hipLaunchKernelGGL(conv_fill_inactive_kernel, dim3(p.tiles_y * p.tiles_x, B), dim3(256), 0, st, p, tile_active, pair);
        p.active_list = list;
        p.active_count = count;
    }
    if (wino32) {
        const bool pc = gn_launch_conv3d_wino32(p, tiles8, st);
        gn_note_kernel(pc ? "conv3d_split_wino32pc_kernel<true>" : "conv3d_split_wino32_kernel<true>");
    } else if (wino) {
        gn_launch_conv3d_wino(p, tiles, st);
        gn_note_kernel("conv3d_split_wino_kernel<true>");
    } else if (strip) {
        if (mode == GN_SPLIT_F16X2) hipLaunchKernelGGL((conv3d_split_strip_kernel<true>), dim3(tiles8 * (Cout / 32), B), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv3d_split_strip_kernel<false>), dim3(tiles8 * (Cout / 32), B), dim3(256), 0, st, p);
        gn_note_kernel(mode == GN_SPLIT_F16X2 ? "conv3d_split_strip_kernel<true>" : "conv3d_split_strip_kernel<false>");
    } else if (wide128) {
        if (mode == GN_SPLIT_F16X2) hipLaunchKernelGGL((conv3d_split_wide_kernel<2, true>), dim3(tiles * (Cout / 128), B), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((conv3d_split_wide_kernel<2, false>), dim3(tiles * (Cout / 128), B), dim3(512), 0, st, p);
        gn_note_kernel(mode == GN_SPLIT_F16X2 ? "conv3d_split_wide_kernel<2, true>" : "conv3d_split_wide_kernel<2, false>");
    } else
    if (mode == GN_SPLIT_BF16X3) SP_LAUNCH(3, false);
    else if (mode == GN_SPLIT_BF16X2) SP_LAUNCH(2, false);
    else SP_LAUNCH(2, true);
#undef SP_LAUNCH
    GN_LAUNCH_CHECK("gn_conv3d_gcr_split");
    return GN_OK;
}

extern "C" int gn_conv3d_gcr_split(const float *src0, int C0, const float *src1, int C1, const float *a, const float *d,
                                   const void *wp_planes, int mode, const float *out_scale, const float *act_inv_scale, int B, int D, int H,
                                   int W, int Cout, int relu, float *out, double *out_sum, double *out_sumsq, const unsigned char *tile_active,
                                   const float *kconst, int kreach, const float *partial, void *occ_ws, size_t occ_ws_bytes, void *stream) {
    return conv3d_gcr_split_impl(src0, C0, src1, C1, a, d, wp_planes, mode, out_scale, act_inv_scale, B, D, H, W, Cout, relu, out, out_sum, out_sumsq,
                                 tile_active, kconst, kreach, partial, nullptr, 0, 0, occ_ws, occ_ws_bytes, stream);
}

// The affine-in-weights form: everything gn_conv_affine_pack (conv_prep.hip) prepared for this batch -- the staging affine (s, -c s), the
// per-sample fp16x2 weight packs, their [B][Cout] output scales and the [B][64][Cout] bias table.  Same kernels, same dispatch.  partial:
// as gn_conv3d_gcr_split (the polyphase form of a decoder layer whose skip connection is at rest away from the cells).
extern "C" int gn_conv3d_gcr_split_persample(const float *src, int Cin, const float *stage_a, const float *stage_d, const void *pack,
                                             const float *out_scale, const float *kbias, int B, int D, int H, int W, int Cout, int relu, float *out,
                                             double *out_sum, double *out_sumsq, const unsigned char *tile_active, const float *kconst, int kreach,
                                             const float *partial, void *occ_ws, size_t occ_ws_bytes, void *stream) {
    GN_REQUIRE(kbias != nullptr && pack != nullptr, "gn_conv3d_gcr_split_persample: pack and kbias are required");
    GN_REQUIRE(Cin > 0 && Cin % SP_KS == 0 && Cout > 0 && Cout % 32 == 0, "gn_conv3d_gcr_split_persample: channel counts must be multiples of 16 (in) / 32 (out)");
    const int64_t per_sample = (int64_t)(Cin / SP_KS) * 27 * (Cout / 32) * 2 * 1024;
    return conv3d_gcr_split_impl(src, Cin, nullptr, 0, stage_a, stage_d, pack, GN_SPLIT_F16X2, out_scale, nullptr, B, D, H, W, Cout, relu, out, out_sum,
                                 out_sumsq, tile_active, kconst, kreach, partial, kbias, per_sample, Cout, occ_ws, occ_ws_bytes, stream);
}

// Winograd F(2,3)-along-x form of the 128-wide kernel (unet_wino.hip): 36 instead of 54 tap products per output pair.  ONE entry for both
// operand forms: kbias == NULL -> the literal form (a, d = the GroupNorm affine incl. the sample's activation scale, act_inv_scale its
// inverse or NULL, pack = ops.pack_conv_weight_split_wino, out_scale [Cout]); kbias != NULL -> the affine-in-weights form (everything from
// gn_conv_affine_pack_wino: a, d = the staging affine, per-sample packs, out_scale [B][Cout], act_inv_scale NULL).
static int conv3d_wino_entry(const float *src, int Cin, const float *a, const float *d, const void *pack, const float *out_scale,
                             const float *act_inv_scale, const float *kbias, int B, int D, int H, int W, int Cout, int relu, float *out,
                             double *out_sum, double *out_sumsq, const unsigned char *tile_active, const float *kconst, int kreach,
                             const float *partial, void *occ_ws, size_t occ_ws_bytes, void *stream) {
    GN_REQUIRE(pack != nullptr && Cin > 0 && Cin % SP_KS == 0 && Cout > 0 && Cout % 32 == 0, "gn_conv3d_gcr_split_wino: pack, Cin %% 16 == 0 and Cout %% 32 == 0 are required");
    GN_REQUIRE(!(kbias && act_inv_scale), "gn_conv3d_gcr_split_wino: the affine-in-weights form carries its scales in out_scale [B][Cout]");
    GN_REQUIRE(!partial || Cout % 128 != 0, "gn_conv3d_gcr_split_wino32: a polyphase partial goes with the 32-wide column-block kernel (Cout %% 128 != 0)");
    const int64_t per_sample = kbias ? (int64_t)(Cin / SP_KS) * 36 * (Cout / 32) * 2 * 1024 : 0;
    return conv3d_gcr_split_impl(src, Cin, nullptr, 0, a, d, pack, GN_SPLIT_F16X2, out_scale, act_inv_scale, B, D, H, W, Cout, relu, out, out_sum,
                                 out_sumsq, tile_active, kconst, kreach, partial, kbias, per_sample, kbias ? Cout : 0, occ_ws, occ_ws_bytes, stream, 1);
}

extern "C" int gn_conv3d_gcr_split_wino(const float *src, int Cin, const float *a, const float *d, const void *pack, const float *out_scale,
                                        const float *act_inv_scale, const float *kbias, int B, int D, int H, int W, int Cout, int relu, float *out,
                                        double *out_sum, double *out_sumsq, const unsigned char *tile_active, const float *kconst, int kreach,
                                        void *occ_ws, size_t occ_ws_bytes, void *stream) {
    return conv3d_wino_entry(src, Cin, a, d, pack, out_scale, act_inv_scale, kbias, B, D, H, W, Cout, relu, out, out_sum, out_sumsq, tile_active, kconst,
                             kreach, nullptr, occ_ws, occ_ws_bytes, stream);
}

// The same entry with the polyphase partial of a decoder's first convolution (gn_upconv_partial, upconv.hip) added before the ReLU -- the 32- / 64-wide
// layers only (unet_wino32.hip; the 128-wide Winograd kernel has no layer that needs it)
extern "C" int gn_conv3d_gcr_split_wino_partial(const float *src, int Cin, const float *a, const float *d, const void *pack, const float *out_scale,
                                                const float *act_inv_scale, const float *kbias, int B, int D, int H, int W, int Cout, int relu, float *out,
                                                double *out_sum, double *out_sumsq, const float *partial, void *stream) {
    GN_REQUIRE(partial != nullptr, "gn_conv3d_gcr_split_wino_partial: partial is required (gn_conv3d_gcr_split_wino is the entry without one)");
    return conv3d_wino_entry(src, Cin, a, d, pack, out_scale, act_inv_scale, kbias, B, D, H, W, Cout, relu, out, out_sum, out_sumsq, nullptr, nullptr,
                             1, partial, nullptr, 0, stream);
}
