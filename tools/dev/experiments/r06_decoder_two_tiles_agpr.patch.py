# round 6: round 4's two-tiles-per-wave decoder kernel on today's helpers, its layer-2 operand planes pinned into AGPRs (PIN=true|false), the
# sched_group_barrier pattern optional (SGB=true|false).  usage: PIN=.. SGB=.. tools/dev/build_variant.sh <name> this-file
s=open('decode_split.hip').read()
marker='''// Note on the counted waits.  VM operations retire in issue order.'''
kernel=r'''// ---- two query tiles per wave (folded scalar decoder [32, 256, 256, 1]): every 1 KB weight fragment pair read from LDS feeds the MFMAs of TWO
// 32-query tiles (12 MFMAs per 4 ds_read_b128 instead of 6), one wave per SIMD with the whole 512-register file: x0 32 + h1 256 + accumulators 64
// + fragments 32 + next rows 32.  Same products in the same order per query as implicit_decode_split_kernel<1, 2>: bit-identical outputs.
template <int TT>
__global__ __launch_bounds__(256, 1) void implicit_decode_split_tt_kernel(DecSplitArgs p) {
    constexpr int OUTC = 1, K0G = 2;
    constexpr int TAB1 = 8 * 2 * 16, TAB2 = 8 * 2 * (1 + OUTC) * 16, TABN = TAB1 + TAB2 + 3 * OUTC;
    constexpr int NS1 = 4 * K0G, NSTEPS = NS1 + 64, NSTAGE = NSTEPS / 4;   // 18 stages
    constexpr int NRAW = 2 * K0G;
    constexpr int RAW_STAGE = NSTAGE - 8;
    constexpr int RING = DS_RING;
    constexpr int TILE = TT * DS_TILE;
    constexpr int TAB_BYTES = ((TABN * 4 + 15) / 16) * 16;
    __shared__ __attribute__((aligned(16))) unsigned char smem[RING * DS_STAGE_BYTES + TAB_BYTES];
    float *const tab = reinterpret_cast<float *>(smem + RING * DS_STAGE_BYTES);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, r = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long ntiles = (p.M + TILE - 1) / TILE;
    if (p.xscale && p.xscale[2] != 0.f) return;
    const float sx = p.xscale ? p.xscale[0] : 1.f, inv_sx = p.xscale ? p.xscale[1] : 1.f;
    for (int i = tid; i < TABN; i += 256) {
        const bool bias = i < TAB1 || (i < TAB1 + TAB2 && ((i - TAB1) / 16) % (1 + OUTC) == 0);
        tab[i] = bias ? __fmul_rn(p.tab[i], sx) : p.tab[i];
    }
    const unsigned char *wsrc = p.wp + (wave * 4) * 1024;
    const unsigned lane16 = lane * 16;
    int sb = 0;                                     // NSTAGE = 18 is not a multiple of the ring: the slot base advances per tile
#define DT_ISSUE(STAGE, SLOT)                                                                                                  \
    ds_glds16x4_s(wsrc + (size_t)(STAGE) * DS_STAGE_BYTES, lane16, lds_base + (SLOT) * DS_STAGE_BYTES + (wave * 4) * 1024);
    DT_ISSUE(0, 0) DT_ISSUE(1, 1) DT_ISSUE(2, 2) DT_ISSUE(3, 3)
    float4 raw[TT][NRAW];
#pragma unroll
    for (int u = 0; u < TT; ++u) {
        long long m = (long long)blockIdx.x * TILE + u * DS_TILE + wave * 32 + r;
        if (m >= p.M) m = p.M - 1;
        const float4 *row = reinterpret_cast<const float4 *>(p.xin + m * p.ldxin + 8 * h);
#pragma unroll
        for (int g = 0; g < K0G; ++g) { raw[u][2 * g] = row[4 * g]; raw[u][2 * g + 1] = row[4 * g + 1]; }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    const unsigned char *const ring_rd = smem + lane * 16;
    uint4 A[4], nA[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + f * 1024);

    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        uint4 x0[TT][2][K0G], h1[TT][2][16];
#pragma unroll
        for (int u = 0; u < TT; ++u)
#pragma unroll
            for (int g = 0; g < K0G; ++g) {
                ds_split2(__fmul_rn(raw[u][2 * g].x, sx), __fmul_rn(raw[u][2 * g].y, sx), x0[u][0][g].x, x0[u][1][g].x);
                ds_split2(__fmul_rn(raw[u][2 * g].z, sx), __fmul_rn(raw[u][2 * g].w, sx), x0[u][0][g].y, x0[u][1][g].y);
                ds_split2(__fmul_rn(raw[u][2 * g + 1].x, sx), __fmul_rn(raw[u][2 * g + 1].y, sx), x0[u][0][g].z, x0[u][1][g].z);
                ds_split2(__fmul_rn(raw[u][2 * g + 1].z, sx), __fmul_rn(raw[u][2 * g + 1].w, sx), x0[u][0][g].w, x0[u][1][g].w);
            }
        float psum[TT];
#pragma unroll
        for (int u = 0; u < TT; ++u) psum[u] = 0.f;
        f32x16q acc[TT][2];
        auto epilogue = [&](int u, int P, int qd) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                if (P < 4) {
                    const int nb = 2 * P + blk;
                    const float4 bv = *reinterpret_cast<const float4 *>(tab + (nb * 2 + h) * 16 + 4 * qd);
                    float v0, v1, v2, v3;
                    ds_bias_relu4(acc[u][blk][4 * qd + 0], acc[u][blk][4 * qd + 1], acc[u][blk][4 * qd + 2], acc[u][blk][4 * qd + 3], bv, v0, v1, v2, v3);
                    const int g2 = 2 * nb + (qd >> 1);
                    if (qd & 1) {
                        ds_split2(v0, v1, h1[u][0][g2].z, h1[u][1][g2].z);
                        ds_split2(v2, v3, h1[u][0][g2].w, h1[u][1][g2].w);
                        if (DT_PIN) asm volatile("" : "+a"(h1[u][0][g2].z), "+a"(h1[u][1][g2].z), "+a"(h1[u][0][g2].w), "+a"(h1[u][1][g2].w));
                    } else {
                        ds_split2(v0, v1, h1[u][0][g2].x, h1[u][1][g2].x);
                        ds_split2(v2, v3, h1[u][0][g2].y, h1[u][1][g2].y);
                        if (DT_PIN) asm volatile("" : "+a"(h1[u][0][g2].x), "+a"(h1[u][1][g2].x), "+a"(h1[u][0][g2].y), "+a"(h1[u][1][g2].y));
                    }
                } else {
                    const int nb = 2 * (P - 4) + blk;
                    const float *tb = tab + TAB1 + ((nb * 2 + h) * (1 + OUTC)) * 16 + 4 * qd;
                    const float4 bv = *reinterpret_cast<const float4 *>(tb);
                    float v0, v1, v2, v3;
                    ds_bias_relu4(acc[u][blk][4 * qd + 0], acc[u][blk][4 * qd + 1], acc[u][blk][4 * qd + 2], acc[u][blk][4 * qd + 3], bv, v0, v1, v2, v3);
                    const float4 wv = *reinterpret_cast<const float4 *>(tb + 16);
                    psum[u] = fmaf(v0, wv.x, psum[u]);
                    psum[u] = fmaf(v1, wv.y, psum[u]);
                    psum[u] = fmaf(v2, wv.z, psum[u]);
                    psum[u] = fmaf(v3, wv.w, psum[u]);
                }
            }
        };
#pragma unroll
        for (int step = 0; step < NSTEPS; ++step) {
            const int t = step >> 2, kg = step & 3;
            const bool l1 = step < NS1;
            const int P = l1 ? step / K0G : 4 + ((step - NS1) >> 4);
            const int g = l1 ? step % K0G : ((step - NS1) & 15);
            if (g == 0) {
#pragma unroll
                for (int u = 0; u < TT; ++u)
#pragma unroll
                    for (int q = 0; q < 16; ++q) { acc[u][0][q] = 0.f; acc[u][1][q] = 0.f; }
            }
#pragma unroll
            for (int f = 0; f < 4; ++f) A[f] = nA[f];
            if (kg < 3) {
#pragma unroll
                for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + ((t + sb) % RING) * DS_STAGE_BYTES + ((kg + 1) * 4 + f) * 1024);
            } else {
                if (t > RAW_STAGE && t <= RAW_STAGE + 3) DS_WAIT_VM_LGKM0(8 + TT * NRAW);
                else DS_WAIT_VM_LGKM0(8);
                __builtin_amdgcn_s_barrier();
                DT_ISSUE((t + RING) % NSTAGE, (t + sb) % RING)
                if (t == RAW_STAGE) {
                    long long tn = tile + gridDim.x;
                    if (tn >= ntiles) tn = tile;
#pragma unroll
                    for (int u = 0; u < TT; ++u) {
                        long long m = tn * TILE + u * DS_TILE + wave * 32 + r;
                        if (m >= p.M) m = p.M - 1;
                        const float4 *row = reinterpret_cast<const float4 *>(p.xin + m * p.ldxin + 8 * h);
#pragma unroll
                        for (int gg = 0; gg < K0G; ++gg) { raw[u][2 * gg] = row[4 * gg]; raw[u][2 * gg + 1] = row[4 * gg + 1]; }
                    }
                }
#pragma unroll
                for (int f = 0; f < 4; ++f) nA[f] = *reinterpret_cast<const uint4 *>(ring_rd + ((t + 1 + sb) % RING) * DS_STAGE_BYTES + f * 1024);
            }
            const int Lp = P < 4 ? (P + 1) * K0G - 1 : NS1 + (P - 3) * 16 - 1;
#pragma unroll
            for (int u = 0; u < TT; ++u) {
                const uint4 b1 = l1 ? x0[u][0][g % K0G] : h1[u][0][g], b2 = l1 ? x0[u][1][g % K0G] : h1[u][1][g];
                acc[u][0] = ds_mfma(A[1], b1, acc[u][0]);
                acc[u][1] = ds_mfma(A[3], b1, acc[u][1]);
                acc[u][0] = ds_mfma(A[0], b2, acc[u][0]);
                acc[u][1] = ds_mfma(A[2], b2, acc[u][1]);
                acc[u][0] = ds_mfma(A[0], b1, acc[u][0]);
                acc[u][1] = ds_mfma(A[2], b1, acc[u][1]);
                // the pair's epilogue of tile u (VALU) follows its last MFMAs directly: it runs under the other tile's MFMAs
                if (step == Lp && P < 7) {
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd) epilogue(u, P, qd);
                }
            }
            // issue pattern of the step: every MFMA followed by a share of the step's VALU / LDS work, so that the epilogue of one tile
            // and the fragment reads run in the shadow of the other tile's MFMAs (one wave per SIMD: nothing else would fill them)
            if (DT_SGB) {
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < TT; ++u)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) epilogue(u, 7, qd);
        sb = (sb + NSTAGE) % RING;
#pragma unroll
        for (int u = 0; u < TT; ++u) {
            const long long m = tile * TILE + u * DS_TILE + wave * 32 + r;
            const float s = psum[u] + __shfl_xor(psum[u], 32);
            if (h == 0 && m < p.M) {
                const float *t3 = tab + TAB1 + TAB2;
                float y = gn_relu(__fadd_rn(__fmul_rn(s, inv_sx), t3[0]));
                y = __fadd_rn(__fmul_rn(y, t3[1]), t3[2]);
                p.out[m * p.ldo] = y;
            }
        }
    }
#undef DT_ISSUE
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
}

'''
import os
kernel = kernel.replace('DT_PIN', os.environ.get('PIN', 'true')).replace('DT_SGB', os.environ.get('SGB', 'false'))
assert marker in s
s=s.replace(marker, kernel+marker,1)
old='''    switch (OUT) {
        case 1: DS_LAUNCH(1); break;'''
new='''    if (C0 == 32 && OUT == 1) {
        const int64_t nt2 = gn_cdiv(M, 2 * DS_TILE);
        hipLaunchKernelGGL((implicit_decode_split_tt_kernel<2>), dim3((unsigned)(nt2 < 256 ? nt2 : 256)), dim3(256), 0, st, p);
        GN_LAUNCH_CHECK("gn_implicit_decode_split");
        return GN_OK;
    }
    switch (OUT) {
        case 1: DS_LAUNCH(1); break;'''
assert old in s
s=s.replace(old,new,1)
open('decode_split.hip','w').write(s)
