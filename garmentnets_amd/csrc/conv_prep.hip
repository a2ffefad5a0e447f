// conv_prep.hip -- gn_conv_affine_pack: the GroupNorm affine of a 'gcr' layer folded into PER-SAMPLE weights.
//
// Reference: components/unet3d.py:66-76 (GroupNorm -> Conv3d -> ReLU).  conv(a x + d) is linear in the operand, so for any per-(sample,
// channel) offset c
//      conv_w(a x + d)[n] = sum_taps sum_ch (w a / s) * ((x - c) s)  +  sum_{taps inside the volume} sum_ch w (a c + d)
// with s an exact power of two.  The conv kernels then multiply u = (x - c) s -- which is EXACTLY ZERO wherever x == c -- by the
// per-sample weights w' = w a / s and add the per-(sample, border class, output channel) constant K in their epilogue.
// Why: the two full-resolution encoder convolutions read volumes that are constant almost everywhere (the scattered volume: zero outside
// ~0.25 % of the cells; the layer behind it: one value per channel outside the cells' neighbourhood).  With the GroupNorm shift inside
// the operand every voxel is non-zero and the matrix cores toggle at full rate; the socket sits at its power cap and the clock falls to
// ~1.75 GHz.  With zeros reaching the matrix cores the SAME instruction stream draws less and the clock rises:
// 444 -> 544 TFLOP/s-eq for the first layer (tools/dev/sparsity_burn.py).  Every tile still goes through the matrix cores.
//
// This file: the per-batch preparation (three small kernels) -- scales and staging affine, row maxima + the K table (fp64), the weight pack
// in the MFMA-fragment order of unet_split.hip ([sample][Cin/16][27][Cout/32][plane][64 lanes][8 fp16] + eight zero steps at the end).
#include "common.h"

// per (sample, channel): operand scale s = 2^k with rms(x - c) * s in [1, 2) (a channel's values are bounded by rms sqrt(V): no fp16
// overflow for V < 2^29, exactly the contract of gn_groupnorm_affine's sample scale, per channel); staging affine (s, -c s); weight factor
// a / s; constant input m = a c + d
__global__ void cprep_scales_kernel(const double *__restrict__ sum, const double *__restrict__ sumsq, double V, const float *__restrict__ a,
                                    const float *__restrict__ d, const float *__restrict__ coff, int BC, float *__restrict__ sa,
                                    float *__restrict__ sd, float *__restrict__ wfac, double *__restrict__ mconst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BC) return;
    const double c = coff ? (double)coff[i] : 0.0;
    double ms = (sumsq[i] - 2.0 * c * sum[i] + V * c * c) / V;             // mean of (x - c)^2
    if (!(ms > 0.0)) ms = 0.0;
    const float rms = (float)sqrt(ms);
    float s = 1.f;
    if (rms > 0.f && rms < INFINITY) {
        int e = 0;
        (void)frexpf(rms, &e);                      // rms = f 2^e, f in [0.5, 1)
        e = 1 - e;
        if (e > 100) e = 100;
        if (e < -100) e = -100;
        s = ldexpf(1.f, e);                         // rms * s in [1, 2)
    }
    sa[i] = s;
    sd[i] = (float)(-c * (double)s);
    wfac[i] = __fdiv_rn(a[i], s);                   // exact: s is a power of two
    mconst[i] = (double)a[i] * c + (double)d[i];
}

// per (sample, output channel): row scale (max |w a / s| over the row -> [1, 2)), its inverse for the epilogue, and the constants
// K[cls] = sum over the taps that lie inside the volume for a voxel of border class cls: per axis m = (has previous) | (has next) << 1
// -> cls = (mz * 4 + my) * 4 + mx (64 entries; 63 = interior)
// WINO: the row maximum is taken over the Winograd F(2,3)-along-x transformed weights (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2 per (kd, kh, channel)):
// that is what the fp16 planes of the Winograd pack hold (unet_wino.hip)
#define CPREP_ROW_LDS (128 * 27)
template <bool WINO>
__global__ __launch_bounds__(64) void cprep_rows_kernel(const float *__restrict__ w, const float *__restrict__ wfac, const double *__restrict__ mconst,
                                                        int Cin, int Cout, float *__restrict__ rowscale, float *__restrict__ osc, float *__restrict__ kbias) {
    const int n = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    __shared__ double tsum[27];
    __shared__ float tmax[64];
    // round 6: the output channel's weight row is staged in LDS once (coalesced) when it fits -- the 27 per-tap sums below walk it channel by channel in a fixed
    // order (fp64, order-sensitive: kept), and did so through 128 dependent global loads: 17 - 24 us per launch for a few KB of work
    __shared__ float wrow[CPREP_ROW_LDS];
    const float *wr = w + (int64_t)n * Cin * 27;
    const float *f = wfac + (int64_t)b * Cin;
    const double *m = mconst + (int64_t)b * Cin;
    __shared__ double mrow[CPREP_ROW_LDS / 27];
    if (Cin * 27 <= CPREP_ROW_LDS) {
        for (int i = lane; i < Cin * 27; i += 64) wrow[i] = wr[i];
        for (int i = lane; i < Cin; i += 64) mrow[i] = m[i];
        __syncthreads();
        wr = wrow;
        m = mrow;
    }
    float mx = 0.f;
    if (WINO) {
        for (int i = lane; i < Cin * 9; i += 64) {
            const double fc = (double)f[i / 9];
            const double g0 = (double)wr[3 * i] * fc, g1 = (double)wr[3 * i + 1] * fc, g2 = (double)wr[3 * i + 2] * fc;
            mx = fmaxf(mx, fmaxf(fmaxf(fabsf((float)g0), fabsf((float)g2)), fmaxf(fabsf((float)(0.5 * (g0 + g1 + g2))), fabsf((float)(0.5 * (g0 - g1 + g2))))));
        }
    } else
    for (int i = lane; i < Cin * 27; i += 64) mx = fmaxf(mx, fabsf(__fmul_rn(wr[i], f[i / 27])));
    tmax[lane] = mx;
    __syncthreads();
    if (lane < 27) {
        double acc = 0.0;
        for (int c = 0; c < Cin; ++c) acc += (double)wr[c * 27 + lane] * m[c];
        tsum[lane] = acc;
    }
    __syncthreads();
    if (lane == 0) {
        float mm = 0.f;
        for (int i = 0; i < 64; ++i) mm = fmaxf(mm, tmax[i]);
        float r = 1.f;
        if (mm > 0.f && mm < INFINITY) {
            int e = 0;
            (void)frexpf(mm, &e);
            e = 1 - e;
            if (e > 100) e = 100;
            if (e < -100) e = -100;
            r = ldexpf(1.f, e);
        }
        rowscale[(int64_t)b * Cout + n] = r;
        osc[(int64_t)b * Cout + n] = __fdiv_rn(1.f, r);
    }
    {
        const int cls = lane, mz = cls >> 4, my = (cls >> 2) & 3, mxm = cls & 3;
        double acc = 0.0;
        for (int kd = 0; kd < 3; ++kd)
            for (int kh = 0; kh < 3; ++kh)
                for (int kw = 0; kw < 3; ++kw) {
                    const bool ok = (kd != 0 || (mz & 1)) && (kd != 2 || (mz & 2)) && (kh != 0 || (my & 1)) && (kh != 2 || (my & 2)) &&
                                    (kw != 0 || (mxm & 1)) && (kw != 2 || (mxm & 2));
                    if (ok) acc += tsum[(kd * 3 + kh) * 3 + kw];
                }
        kbias[((int64_t)b * 64 + cls) * Cout + n] = (float)acc;
    }
}

// the pack: one thread per (sample, slice S, tap, 32-wide block, lane): 8 channels of one output channel, two fp16 planes
__global__ __launch_bounds__(256) void cprep_pack_kernel(const float *__restrict__ w, const float *__restrict__ wfac, const float *__restrict__ rowscale,
                                                         int Cin, int Cout, int64_t per_sample_u4, uint4 *__restrict__ pack) {
    const int nsl = Cin / 16, nblk = Cout / 32;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)nsl * 27 * nblk * 64;
    const int b = blockIdx.y;
    if (t >= per) return;
    const int lane = (int)(t & 63);
    int64_t q = t >> 6;
    const int blk = (int)(q % nblk); q /= nblk;
    const int tap = (int)(q % 27);
    const int S = (int)(q / 27);
    const int h = lane >> 5, r = lane & 31, n = blk * 32 + r;
    const float rs = rowscale[(int64_t)b * Cout + n];
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 p1, p2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = S * 16 + 8 * h + i;
        const float v = __fmul_rn(__fmul_rn(w[((int64_t)n * Cin + c) * 27 + tap], wfac[(int64_t)b * Cin + c]), rs);
        const _Float16 x1 = (_Float16)v;
        p1[i] = x1;
        p2[i] = (_Float16)__fsub_rn(v, (float)x1);
    }
    uint4 *dst = pack + (int64_t)b * per_sample_u4 + ((((int64_t)S * 27 + tap) * nblk + blk) * 2) * 64 + lane;
    dst[0] = __builtin_bit_cast(uint4, p1);
    dst[64] = __builtin_bit_cast(uint4, p2);
}

// the Winograd pack (unet_wino.hip): [sample][Cin/16][step = (j * 3 + kd) * 3 + kh][Cout/32][plane][64 lanes][8 fp16], j = transform position along x;
// transformed in fp64 from the fp32 products w * wfac, scaled by the row scale, split into two fp16 planes
__global__ __launch_bounds__(256) void cprep_pack_wino_kernel(const float *__restrict__ w, const float *__restrict__ wfac, const float *__restrict__ rowscale,
                                                              int Cin, int Cout, int64_t per_sample_u4, uint4 *__restrict__ pack) {
    const int nsl = Cin / 16, nblk = Cout / 32;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per = (int64_t)nsl * 36 * nblk * 64;
    const int b = blockIdx.y;
    if (t >= per) return;
    const int lane = (int)(t & 63);
    int64_t q = t >> 6;
    const int blk = (int)(q % nblk); q /= nblk;
    const int step = (int)(q % 36);
    const int S = (int)(q / 36);
    const int j = step / 9, kdh = step % 9;
    const int h = lane >> 5, r = lane & 31, n = blk * 32 + r;
    const double rs = (double)rowscale[(int64_t)b * Cout + n];
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 p1, p2;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = S * 16 + 8 * h + i;
        const float *wt = w + ((int64_t)n * Cin + c) * 27 + kdh * 3;
        const double fc = (double)wfac[(int64_t)b * Cin + c];
        const double g0 = (double)wt[0] * fc, g1 = (double)wt[1] * fc, g2 = (double)wt[2] * fc;
        const double u = j == 0 ? g0 : j == 1 ? 0.5 * (g0 + g1 + g2) : j == 2 ? 0.5 * (g0 - g1 + g2) : g2;
        const float v = (float)(u * rs);
        const _Float16 x1 = (_Float16)v;
        p1[i] = x1;
        p2[i] = (_Float16)__fsub_rn(v, (float)x1);
    }
    uint4 *dst = pack + (int64_t)b * per_sample_u4 + ((((int64_t)S * 36 + step) * nblk + blk) * 2) * 64 + lane;
    dst[0] = __builtin_bit_cast(uint4, p1);
    dst[64] = __builtin_bit_cast(uint4, p2);
}

#define CPREP_WINO_PAD 6    // zero steps behind the last sample: unet_wino.hip's fragment DMA runs two groups of three steps ahead

extern "C" size_t gn_conv_affine_pack_bytes(int B, int Cin, int Cout) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || Cin % 16 || Cout % 32) return 0;
    const size_t step = (size_t)(Cout / 32) * 2 * 1024;                    // bytes per (slice, tap) step
    return ((size_t)B * (Cin / 16) * 27 + 8) * step;                       // + eight zero steps behind the last sample (the kernels' DMA look-ahead: the x-strip kernel requests two groups of three steps ahead)
}

extern "C" size_t gn_conv_affine_pack_wino_bytes(int B, int Cin, int Cout) {
    if (B < 0 || Cin <= 0 || Cout <= 0 || Cin % 16 || Cout % 32) return 0;
    const size_t step = (size_t)(Cout / 32) * 2 * 1024;
    return ((size_t)B * (Cin / 16) * 36 + CPREP_WINO_PAD) * step;
}

static int conv_affine_pack_impl(const float *w, int Cin, int Cout, const float *a, const float *d, const double *sum, const double *sumsq,
                                 int64_t V, const float *coff, int B, void *pack, size_t pack_bytes, float *stage_a, float *stage_d,
                                 float *out_scale, float *kbias, void *ws, size_t ws_bytes, void *stream, bool wino) {
    GN_REQUIRE(B >= 0 && Cin > 0 && Cout > 0 && Cin % 16 == 0 && Cout % 32 == 0 && V > 0, "gn_conv_affine_pack: channels must be multiples of 16 (in) / 32 (out)");
    if (B == 0) return GN_OK;
    GN_REQUIRE(w && a && d && sum && sumsq && pack && stage_a && stage_d && out_scale && kbias && ws, "gn_conv_affine_pack: null pointer");
    GN_REQUIRE(pack_bytes >= (wino ? gn_conv_affine_pack_wino_bytes(B, Cin, Cout) : gn_conv_affine_pack_bytes(B, Cin, Cout)), "gn_conv_affine_pack: pack buffer too small");
    const size_t need = (size_t)B * Cin * (4 + 8) + (size_t)B * Cout * 4;
    GN_REQUIRE(ws_bytes >= need, "gn_conv_affine_pack: workspace too small (%zu < %zu)", ws_bytes, need);
    GN_REQUIRE(((uintptr_t)ws & 7) == 0 && ((uintptr_t)pack & 15) == 0, "gn_conv_affine_pack: ws must be 8-byte aligned, pack 16-byte aligned");
    hipStream_t st = gn_stream(stream);
    double *mconst = (double *)ws;                                          // [B][Cin] (8-byte aligned first)
    float *wfac = (float *)(mconst + (size_t)B * Cin);                      // [B][Cin]
    float *rowscale = wfac + (size_t)B * Cin;                               // [B][Cout]
    const int nsteps = wino ? 36 : 27;
    const size_t step = (size_t)(Cout / 32) * 2 * 1024, per_sample = (size_t)(Cin / 16) * nsteps * step;
    GN_HIP(hipMemsetAsync((char *)pack + (size_t)B * per_sample, 0, (wino ? CPREP_WINO_PAD : 8) * step, st), "gn_conv_affine_pack");
    hipLaunchKernelGGL(cprep_scales_kernel, dim3((unsigned)gn_cdiv((int64_t)B * Cin, 256)), dim3(256), 0, st, sum, sumsq, (double)V, a, d, coff, B * Cin,
                       stage_a, stage_d, wfac, mconst);
    if (wino) hipLaunchKernelGGL(cprep_rows_kernel<true>, dim3((unsigned)Cout, (unsigned)B), dim3(64), 0, st, w, wfac, mconst, Cin, Cout, rowscale, out_scale, kbias);
    else hipLaunchKernelGGL(cprep_rows_kernel<false>, dim3((unsigned)Cout, (unsigned)B), dim3(64), 0, st, w, wfac, mconst, Cin, Cout, rowscale, out_scale, kbias);
    const int64_t per = (int64_t)(Cin / 16) * nsteps * (Cout / 32) * 64;
    if (wino) hipLaunchKernelGGL(cprep_pack_wino_kernel, dim3((unsigned)gn_cdiv(per, 256), (unsigned)B), dim3(256), 0, st, w, wfac, rowscale, Cin, Cout,
                                 (int64_t)(per_sample / 16), (uint4 *)pack);
    else hipLaunchKernelGGL(cprep_pack_kernel, dim3((unsigned)gn_cdiv(per, 256), (unsigned)B), dim3(256), 0, st, w, wfac, rowscale, Cin, Cout,
                            (int64_t)(per_sample / 16), (uint4 *)pack);
    GN_LAUNCH_CHECK("gn_conv_affine_pack");
    return GN_OK;
}

extern "C" int gn_conv_affine_pack(const float *w, int Cin, int Cout, const float *a, const float *d, const double *sum, const double *sumsq,
                                   int64_t V, const float *coff, int B, void *pack, size_t pack_bytes, float *stage_a, float *stage_d,
                                   float *out_scale, float *kbias, void *ws, size_t ws_bytes, void *stream) {
    return conv_affine_pack_impl(w, Cin, Cout, a, d, sum, sumsq, V, coff, B, pack, pack_bytes, stage_a, stage_d, out_scale, kbias, ws, ws_bytes, stream, false);
}

// the same preparation for the Winograd F(2,3)-along-x kernel (gn_conv3d_gcr_split_wino): the pack holds the TRANSFORMED per-sample weights (36 steps per
// 16-channel slice, gn_conv_affine_pack_wino_bytes), the row scales are taken over them; stage_a / stage_d / kbias are what gn_conv_affine_pack gives
extern "C" int gn_conv_affine_pack_wino(const float *w, int Cin, int Cout, const float *a, const float *d, const double *sum, const double *sumsq,
                                        int64_t V, const float *coff, int B, void *pack, size_t pack_bytes, float *stage_a, float *stage_d,
                                        float *out_scale, float *kbias, void *ws, size_t ws_bytes, void *stream) {
    return conv_affine_pack_impl(w, Cin, Cout, a, d, sum, sumsq, V, coff, B, pack, pack_bytes, stage_a, stage_d, out_scale, kbias, ws, ws_bytes, stream, true);
}
