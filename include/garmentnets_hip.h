/*
 * garmentnets_hip.h -- C ABI of libgarmentnets_hip.so (MI355X / gfx950).
 *
 * The reference (real-stanford/garmentnets) is pure Python and has no FFI of its own: its inference hot
 * path reaches native code only through third-party operator packages.  Each entry point below replaces
 * ONE such operator call site (cited as reference file:line) with a hand-written HIP kernel; the Python
 * host (garmentnets_amd/) binds them with ctypes and keeps the reference's module API on top
 * (INTEGRATION.md shows the binding a maintainer of the reference would add).
 *
 * Conventions
 *   - every function returns GN_OK (0) or a negative GN_E* code; gn_last_error() gives a thread-local message.
 *   - all pointers are DEVICE pointers into caller-owned memory unless the name ends in _host.
 *   - stream is a hipStream_t passed as void* (NULL = default stream); all work is stream-ordered, no
 *     allocation, no host synchronisation, no global mutable state (LUTs are immutable).
 *   - dense float tensors are fp32, row-major with an explicit leading dimension (ld*) in elements.
 *   - volumes are CHANNEL-LAST: [B][D][H][W][C]  (the reference's NCDHW is a host-side view of this).
 *   - indices: point / vertex indices int32 on device (int64 only where the reference API exposes them).
 */
#ifndef GARMENTNETS_HIP_H
#define GARMENTNETS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GN_OK 0
#define GN_EINVAL (-1)   /* bad argument (shape / alignment / unsupported size) */
#define GN_ELAUNCH (-2)  /* HIP launch error */
#define GN_ECAP (-3)     /* caller-provided capacity too small (counts are still returned) */

const char *gn_last_error(void);
int gn_version(void);
/* name of the kernel variant the last gn_conv3d_gcr / gn_conv3d_gcr_split call of this thread launched (static string; ""
 * before the first call): measurement code labels its per-kernel timings with what really ran, not with a re-derived dispatch rule */
const char *gn_last_kernel(void);
/* number of CUs / XCDs the library sees on the current device (sanity: 256 / 8 on MI355X) */
int gn_device_info(int *num_cu, int *lds_bytes_per_cu);

/* ---------------------------------------------------------------------------------------------------------
 * Point-set operators (PointNet++).
 * ------------------------------------------------------------------------------------------------------- */

/* ptr[b] = first index i with batch[i] >= b, b = 0..B  (batch sorted ascending, int64).
 * replaces: PyG Batch bookkeeping used by every segmented op (components/pointnet2.py:26-29). */
int gn_segment_ptr(const int64_t *batch, int64_t n, int B, int32_t *ptr, void *stream);

/* Farthest point sampling.  replaces torch_cluster.fps -- components/pointnet2.py:26.
 * Per example b (points ptr[b]..ptr[b+1]) emits out_ptr[b+1]-out_ptr[b] indices (GLOBAL point index) starting
 * with local point start_idx[b] (start_idx == NULL: the example's first point = torch_cluster random_start=False; the
 * upstream default random_start=True is obtained by passing host-drawn random starts);
 * dist = (dx*dx+dy*dy)+dz*dz in fp32 without FMA; ties -> lowest index.
 * One workgroup per example (one wave per SIMD up to 24 points per lane), positions + running min-distance resident in registers / LDS. */
int gn_fps(const float *pos, const int32_t *ptr, const int32_t *out_ptr, const int32_t *start_idx, int B,
           int max_points_per_example, int32_t *out_idx, void *stream);

/* gn_fps for the CASCADE of components/pointnet2.py:26 (networks/pointnet2_nocs.py:139-141: sa2 samples the points sa1 selected).
 * gap_out    (NULL or [B] float): the smallest running maximum of the min-distance over this call's steps (3e38 if it took none).
 * nested_gap (NULL or [B] float): the gap_out of the EARLIER call whose output, in selection order, is this call's `pos` (caller's
 *            contract; start_idx must be NULL in both).  Farthest-point order is nested: with s_0..s_{m1-1} the earlier selection and
 *            D_k(i) = min_{j<k} d(i, s_j), position t of the new cloud holds E_k(t) = D_k(s_t) once positions 0..k-1 are chosen -- 0 for
 *            t < k, at most max_i D_k(i) = D_k(s_k) for t >= k -- so while that maximum is positive position k wins the arg-max and its
 *            lowest-index tie rule: an example with nested_gap[b] > 0 gets the indices ptr[b] + 0, 1, ..., m-1 without a single step,
 *            the others (duplicate-ridden clouds) are sampled as gn_fps samples them.  Same output as gn_fps, bit for bit, either way. */
int gn_fps_nested(const float *pos, const int32_t *ptr, const int32_t *out_ptr, const int32_t *start_idx, int B,
                  int max_points_per_example, int32_t *out_idx, float *gap_out, const float *nested_gap, void *stream);

/* Ball query.  replaces torch_cluster.radius(max_num_neighbors=K) -- components/pointnet2.py:28-29.
 * For every centre c (a point index centre_idx[c], example b): the first K points j of example b in ascending
 * index with d2(j,c) < r2 (strict).  nbr: [M][K] int32, -1 padded; cnt: [M].  One wavefront per centre
 * (ballot + popcount prefix keep the ascending order). */
int gn_ball_query(const float *pos, const int32_t *ptr, const int32_t *centre_idx, const int32_t *centre_ptr, int B,
                  int M, float r2, int K, int32_t *nbr, int32_t *cnt, void *stream);

/* Edge-feature gather for PointConv.  replaces PyG PointConv.message() gather -- components/pointnet2.py:31.
 * Row (c*(K+1)+s) of out = [x[j] (C floats), pos[j]-pos[centre_idx[c]] (3 floats)] for slot s of centre c, where
 * slots 0..K-1 are nbr[c][s] with the PointConv self-loop rule applied (a neighbour whose index equals the
 * centre ORDINAL c is dropped) and slot K is point c itself (add_self_loops on the bipartite graph);
 * self_loops=0 disables both.  Invalid slots are written as zeros and slot_src[c*(K+1)+s] = -1.
 * x may be NULL (C=0). */
int gn_sa_gather(const float *x, int ldx, int C, const float *pos, const int32_t *centre_idx, const int32_t *nbr,
                 int M, int K, int self_loops, float *out, int ldo, int32_t *slot_src, void *stream);
/* The same with the self-loop rule scoped by the caller: self_src[c] (int32 [M], or NULL = c: the call above) is the point that plays
 * "node c" on the source side -- dropped from centre c's neighbours, and the source of its added loop.  With self_src[c] = ptr[b] +
 * (c - centre_ptr[b]) (b = the centre's example) every example of a batch gets exactly what a batch of one would give it:
 * predict.py:62 asserts batch_size == 1, so that is the result the reference's pipeline produces for a garment. */
int gn_sa_gather_scoped(const float *x, int ldx, int C, const float *pos, const int32_t *centre_idx, const int32_t *nbr,
                        int M, int K, int self_loops, const int32_t *self_src, float *out, int ldo, int32_t *slot_src, void *stream);

/* out[c][ch] = max over valid slots s of in[c*S+s][ch].  replaces PointConv aggr='max' (scatter-max). */
int gn_segment_max(const float *in, int ldi, const int32_t *slot_src, int M, int S, int C, float *out, int ldo,
                   void *stream);

/* Set abstraction in ONE kernel: grouping gather -> 3-layer edge MLP (Linear -> ReLU -> eval-BatchNorm affine per layer, exact fp32
 * products on the matrix cores) -> segmented max, no edge tensor in HBM.  replaces PointConv(local_nn, aggr='max', add_self_loops)
 * on the radius graph -- components/pointnet2.py:20,31 (= gn_sa_gather + 3 x gn_linear + gn_segment_max, which remain for other MLP
 * shapes).  nbr/cnt: gn_ball_query's table (K <= 64, valid entries first); self_loops: PyG's bipartite quirk (centre i also receives
 * point i of the whole cloud; table entries equal to the centre's own index are dropped).  w1p/w2p/w3p/tab: garmentnets_amd.ops.
 * pack_sa_fused (A fragments [N/32][K blocks][4][64 lanes] x 16 B; per-unit bias | scale | shift in accumulator-register order).
 * Instantiated for the shipped edge MLPs [3+3,64,64,128] and [128+3,128,128,256] (gn_sa_fused_supported).  out [M][ldo]: max over
 * the centre's edges, 0 for a centre without edges. */
int gn_sa_fused_supported(int C, int N1, int N2, int N3);
int gn_sa_fused(const float *x, int ldx, int C, const float *pos, const int32_t *centre_idx, const int32_t *nbr, const int32_t *cnt,
                int M, int K, int self_loops, const float *w1p, const float *w2p, const float *w3p, const float *tab, int N1, int N2,
                int N3, float *out, int ldo, void *stream);
/* gn_sa_fused with the caller-scoped self-loop rule of gn_sa_gather_scoped (self_src: int32 [M] or NULL). */
int gn_sa_fused_scoped(const float *x, int ldx, int C, const float *pos, const int32_t *centre_idx, const int32_t *nbr, const int32_t *cnt,
                       int M, int K, int self_loops, const int32_t *self_src, const float *w1p, const float *w2p, const float *w3p,
                       const float *tab, int N1, int N2, int N3, float *out, int ldo, void *stream);

/* out[b][ch] = max over rows ptr[b]..ptr[b+1].  replaces PyG global_max_pool -- components/pointnet2.py:49. */
int gn_global_max_pool(const float *in, int ldi, const int32_t *ptr, int B, int C, float *out, int ldo, void *stream);

/* k-NN inverse-squared-distance interpolation.  replaces PyG knn_interpolate -- components/pointnet2.py:72.
 * k <= 8; neighbours ordered by ascending (d2, index); w = 1/max(d2,1e-16); out = sum(w x)/sum(w). */
int gn_knn_interpolate(const float *xs, int ldx, const float *ps, const int32_t *ptr_s, const float *pq,
                       const int32_t *ptr_q, int B, int Nq, int C, int k, float *out, int ldo, void *stream);

/* Dense layer:  Y = bn( act( X W^T + bias ) ),  X [M][K] (ldx), W [N][K] (ldw), Y [M][N] (ldy).
 * relu != 0 applies ReLU; bn_scale/bn_shift (may be NULL) apply y*scale[n]+shift[n] AFTER the ReLU
 * (components/mlp.py:9-20: Linear -> ReLU -> BatchNorm1d in eval mode).  fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * replaces torch.nn.Linear / ReLU / BatchNorm1d (ATen addmm + batch_norm) -- components/mlp.py:9-20,
 * networks/pointnet2_nocs.py:145-157, and the 1x1x1 final_conv components/unet3d.py:437,467. */
int gn_linear(const float *X, int ldx, const float *W, int ldw, const float *bias, const float *bn_scale,
              const float *bn_shift, int relu, int64_t M, int N, int K, float *Y, int ldy, void *stream);

/* NOCS head post-processing.  replaces argmax/softmax/gather + VirtualGrid.idxs_to_points --
 * networks/conv_implicit_wnf.py:220-231.  logits [N][bins*3] viewed as [N][bins][3]. */
int gn_nocs_head(const float *logits, int ldl, int64_t N, int bins, int64_t *bin_idx /*[N][3]*/, float *confidence /*[N][3]*/,
                 float *pred_nocs /*[N][3]*/, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Gridding (VolumeFeatureAggregator).
 * ------------------------------------------------------------------------------------------------------- */

/* Per-point aggregation features + target cell.  replaces VirtualGrid.get_points_grid_idxs / flatten_idxs /
 * idxs_to_points + torch.cat -- networks/conv_implicit_wnf.py:72-85, components/gridding.py:161-256.
 * out row = [feat (Cf), p - cell_corner (3), sim_pos (3), confidence (3)];  flat = ((b*G0+ix)*G1+iy)*G2+iz. */
int gn_grid_features(const float *feat, int ldf, int Cf, const float *nocs, const float *sim_pos, const float *conf,
                     const int64_t *batch, int64_t N, const float lower[3], const float upper[3], const int grid[3],
                     int include_point, int include_conf, float *out, int ldo, int32_t *flat_idx, void *stream);

/* Scatter point features into a zero-filled channel-last volume.  replaces torch_scatter.scatter(reduce) --
 * networks/conv_implicit_wnf.py:92-94.  reduce: 0 = max, 1 = mean.  vol [cells][C] must be zeroed by this
 * call (it does the memset); count_ws: [cells] int32 workspace.  Empty cells stay 0.  Both reductions are
 * deterministic (run-to-run bit-identical): max by construction, mean through order-independent fp64 partial sums
 * kept in `ws` (gn_grid_scatter_workspace_bytes(N, C, reduce) bytes; 0 / NULL for max).  vol_is_zeroed != 0: the caller has already
 * zero-filled vol and count_ws (e.g. on a side stream, overlapped with the serial FPS kernels) and the memsets are skipped. */
size_t gn_grid_scatter_workspace_bytes(int64_t N, int C, int reduce);
int gn_grid_scatter(const float *src, int lds, const int32_t *flat_idx, int64_t N, int C, int64_t cells, int reduce,
                    float *vol, int32_t *count_ws, void *ws, size_t ws_bytes, int vol_is_zeroed, void *stream);

/* Output tiles (4 x 8 x 8 voxels, the tiling of gn_conv3d_gcr_split) of a 3x3x3 convolution over the scattered volume that can see an
 * occupied cell: flags [B][tiles_y * tiles_x * tiles_z] bytes (zeroed inside), index (ty * tiles_x + tx) * tiles_z + tz.  reach = 1 for the
 * convolution that reads the scattered volume, 2 for the one behind it (its input is non-constant within one voxel of the cells). */
int gn_grid_tile_flags(const int32_t *flat_idx, int64_t N, int B, int G0, int G1, int G2, int reach, unsigned char *flags, void *stream);

/* Per-(sample, channel) sum / sum of squares of a scattered volume computed from its OCCUPIED cells only (all other
 * cells are zero): the GroupNorm statistics of the first UNet layer without reading the (mostly empty) volume.
 * cells_per_sample = G0*G1*G2; sum/sumsq [B][C] fp64 (zeroed inside).  Must run after gn_grid_scatter on the same stream. */
int gn_grid_stats(const float *vol, const int32_t *flat_idx, int64_t N, int C, int64_t cells_per_sample, int B,
                  int32_t *count_ws, double *sum, double *sumsq, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * 3-D UNet (channel-last volumes [B][D][H][W][C]).
 * ------------------------------------------------------------------------------------------------------- */

/* Per-(sample, channel) sum and sum of squares over the V voxels, fp64 accumulators [B][C] (zeroed inside).
 * First half of nn.GroupNorm -- components/unet3d.py:66. */
int gn_channel_stats(const float *x, int B, int64_t V, int C, double *sum, double *sumsq, void *stream);

/* GroupNorm statistics -> per-(sample, channel) affine  y = x*a + d  for the concatenation of up to two
 * sources (src0: C0 channels, V0 voxels; src1: C1 channels, V1 voxels each replicated rep1 times = nearest
 * upsampling).  groups over C0+C1 channels, eps, biased variance (nn.GroupNorm).  a,d: [B][C0+C1].
 * act_inv_scale (NULL, or [B]): range normalisation for the split-operand convs -- a and d of sample b are multiplied by the power
 * of two that brings the largest per-channel rms of the normalised activations into [1, 2) and act_inv_scale[b] receives its
 * inverse (gn_conv3d_gcr_split undoes it exactly in its epilogue).  fp16 planes then cannot overflow for any checkpoint. */
int gn_groupnorm_affine(const double *sum0, const double *sq0, int C0, int64_t V0, const double *sum1, const double *sq1,
                        int C1, int64_t V1, int rep1, int B, int groups, float eps, const float *gamma,
                        const float *beta, float *a, float *d, float *act_inv_scale, void *stream);

/* Fused GroupNorm-apply + Conv3d(3x3x3, pad 1, no bias) + ReLU, the 'gcr' SingleConv --
 * components/unet3d.py:53-66,19-76.  Input = channel concatenation [src0 (C0 ch, full res), src1 (C1 ch, HALF
 * res, nearest-upsampled on the fly)] (torch.cat((skip, up(x))) of components/unet3d.py:291,330); src1 may be
 * NULL.  Zero padding is applied AFTER the affine (as nn.Conv3d pads the normalised tensor).
 * wp: weights repacked [27 taps][Cin/16 slices][Cout][16] (tap = (kd*3+kh)*3+kw, channel = slice*16 + k).
 * out_sum / out_sumsq (both NULL or both [B][Cout] fp64, zeroed inside): per-(sample, channel) sum and sum of squares
 * of the OUTPUT, i.e. the GroupNorm statistics of the next layer, produced by the epilogue.
 * LDS-tiled implicit GEMM on fp32 MFMA. */
int gn_conv3d_gcr(const float *src0, int C0, const float *src1, int C1, const float *a, const float *d,
                  const float *wp, int B, int D, int H, int W, int Cout, int relu, float *out, double *out_sum,
                  double *out_sumsq, void *stream);

/* Split-operand variant of gn_conv3d_gcr on the 16-bit matrix cores (GN_SPLIT_F16X2 is what garmentnets_amd uses by default;
 * gn_conv3d_gcr is the plain fp32-MFMA kernel): identical contract, but the
 * fp32 operands are decomposed into low-precision planes (x = x1 + x2 [+ x3], exact residual chain) and multiplied on the
 * 16-bit matrix cores with fp32 accumulation:
 *   GN_SPLIT_BF16X3: 3 bf16 planes, 6 partial products, dropped terms <= 2^-24 relative (fp32-class products)
 *   GN_SPLIT_F16X2 : 2 fp16 planes, 3 partial products, operand residual and dropped term <= 2^-22 relative for operands within
 *                    2^3 of their scale (below: absolute, 2^-25 of the scale).  Scales: one power of two per OUTPUT CHANNEL for the
 *                    weights (row maximum in [1, 2)) and one per SAMPLE for the activations (gn_groupnorm_affine act_inv_scale)
 *   GN_SPLIT_BF16X2: 2 bf16 planes, 3 partial products, 2^-16 relative (fast preview quality)
 * wp_planes: weight pack in MFMA-fragment order [Cin/16][27 taps][Cout/32][planes][64 lanes] x 16 B + eight zero steps
 * (garmentnets_amd.ops.pack_conv_weight_split); out_scale [Cout]: the exact powers of two that undo the pack's per-output-channel
 * weight scales (all 1 for bf16); act_inv_scale: NULL or [B] from gn_groupnorm_affine. */
#define GN_SPLIT_BF16X2 2
#define GN_SPLIT_BF16X3 3
#define GN_SPLIT_F16X2 4
/* Occupancy-aware launch (the layer whose input is gn_grid_scatter's volume, > 99 % empty for 6000 points in 128^3 cells):
 * tile_active (NULL, or [B][tiles] bytes from gn_grid_tile_flags) marks the 4 x 8 x 8 output tiles whose halo holds an occupied cell;
 * every other tile's outputs are border-class constants kconst [B][(2r+1)^3][Cout], r = kreach (class (cz*n + cy)*n + cx with, per
 * axis, c = z for z < r, 2r - (D-1-z) for z >= D-r, r otherwise; the FINISHED values a dense launch produces there -- garmentnets_amd
 * takes them from dense launches of the same layers over a 5 x 5 x 5 all-zero volume with the same affines) and are stored without
 * touching the matrix cores.  kreach = 1: the layer fed by the scattered volume; 2: the layer behind it.  The output is bit-identical
 * to the dense launch.  Two-plane modes only.  The call builds a compact, ascending list of the active tiles in occ_ws
 * (gn_conv3d_occupancy_workspace_bytes(B, D, H, W) bytes; required with tile_active), fills the inactive tiles (constants + their share of
 * out_sum / out_sumsq) with an HBM-bound kernel and runs the convolution kernel over the listed tiles only. */
size_t gn_conv3d_occupancy_workspace_bytes(int B, int D, int H, int W);
/* `partial` (NULL, or [B][D/2][H/2][W/2][8][Cout] from gn_upconv_partial): the polyphase form of a layer whose second source is
 * nearest-upsampled -- this launch then covers the full-resolution source alone (src1 NULL) and its epilogue adds
 * partial[b][z>>1][y>>1][x>>1][(z&1)*4 + (y&1)*2 + (x&1)][n] before the ReLU. */
int gn_conv3d_gcr_split(const float *src0, int C0, const float *src1, int C1, const float *a, const float *d,
                        const void *wp_planes, int mode, const float *out_scale, const float *act_inv_scale, int B, int D, int H, int W,
                        int Cout, int relu, float *out, double *out_sum, double *out_sumsq, const unsigned char *tile_active,
                        const float *kconst, int kreach, const float *partial, void *occ_ws, size_t occ_ws_bytes, void *stream);

/* The GroupNorm affine folded into PER-SAMPLE weights (csrc/conv_prep.hip) -- for a 'gcr' layer (components/unet3d.py:66-76) whose input
 * is AT REST almost everywhere: the scattered volume of networks/conv_implicit_wnf.py:92-94 (zero outside the occupied cells) and the
 * layer behind it (one value per channel away from them).  conv(a x + d) is linear, so with c = the rest value (coff [B][Cin]; NULL =
 * zeros) and s a per-(sample, channel) power of two
 *     conv_w(a x + d)[n] = sum_taps sum_ch (w a / s) ((x - c) s)  +  sum_{taps inside the volume} sum_ch w (a c + d):
 * the operand (x - c) s is EXACTLY ZERO wherever x == c, the shift becomes the per-(sample, border class, output channel) constant kbias.
 * Same MACs through the same kernels; the matrix cores see mostly zeros and draw less power, which under the socket power cap is clock
 * (DESIGN.md 5.1).  gn_conv_affine_pack prepares, from the raw Conv3d weight w [Cout][Cin][3][3][3], the GroupNorm affine a, d [B][Cin]
 * (gn_groupnorm_affine without the sample scale) and the input's statistics sum / sumsq [B][Cin] over V voxels:
 *     stage_a, stage_d [B][Cin]  the staging affine (s, -c s), s = 2^k with rms(x - c) s in [1, 2)
 *     pack                       [B] fp16x2 weight sets (w a / s, row-scaled to [1, 2)) in the fragment order of gn_conv3d_gcr_split,
 *                                gn_conv_affine_pack_bytes(B, Cin, Cout) bytes
 *     out_scale [B][Cout]        exact powers of two undoing the row scales
 *     kbias [B][64][Cout]        class = (mz * 4 + my) * 4 + mx, m = (voxel has a previous neighbour on the axis) | (a next one) << 1
 * ws: B * Cin * 12 + B * Cout * 4 bytes.  gn_conv3d_gcr_split_persample runs the layer from them (GN_SPLIT_F16X2 arithmetic, one source;
 * tile_active / kconst / kreach / partial / occ_ws as gn_conv3d_gcr_split). */
size_t gn_conv_affine_pack_bytes(int B, int Cin, int Cout);
int gn_conv_affine_pack(const float *w, int Cin, int Cout, const float *a, const float *d, const double *sum, const double *sumsq, int64_t V,
                        const float *coff, int B, void *pack, size_t pack_bytes, float *stage_a, float *stage_d, float *out_scale,
                        float *kbias, void *ws, size_t ws_bytes, void *stream);
int gn_conv3d_gcr_split_persample(const float *src, int Cin, const float *stage_a, const float *stage_d, const void *pack,
                                  const float *out_scale, const float *kbias, int B, int D, int H, int W, int Cout, int relu, float *out,
                                  double *out_sum, double *out_sumsq, const unsigned char *tile_active, const float *kconst, int kreach,
                                  const float *partial, void *occ_ws, size_t occ_ws_bytes, void *stream);

/* Winograd F(2,3) along x for the 128-wide 'gcr' convolution (csrc/unet_wino.hip; components/unet3d.py:53-76, the shape it exists for:
 * the first encoder convolution 128 -> 128 at full resolution, components/unet3d.py:127-133): per output pair (x, x+1) and (kd, kh, channel)
 * FOUR products m0 = (d0 - d2) g0, m1 = (d1 + d2)(g0 + g1 + g2)/2, m2 = (d2 - d1)(g0 - g1 + g2)/2, m3 = (d1 - d3) g2 instead of six --
 * 36 instead of 54 matrix-core tap products per output pair, GN_SPLIT_F16X2 arithmetic on the transformed operands (input transform in
 * fp32 before the plane split, weight transform in fp64 on the pack side, output transform in fp32).  Exactly-zero operands stay exactly zero.
 * ONE entry for both operand forms:
 *   kbias == NULL: the literal form.  a, d [B][Cin] = gn_groupnorm_affine (with its act_inv_scale [B], or NULL); pack = the transformed
 *                  static weights [Cin/16][36 steps = (j * 3 + kd) * 3 + kh][Cout/32][2 planes][64 lanes] x 16 B + six zero steps
 *                  (garmentnets_amd.ops.pack_conv_weight_split_wino); out_scale [Cout].
 *   kbias != NULL: the affine-in-weights form -- a, d, pack, out_scale [B][Cout], kbias [B][64][Cout] from gn_conv_affine_pack_wino
 *                  (same contract as gn_conv_affine_pack, 36 steps per slice: gn_conv_affine_pack_wino_bytes); act_inv_scale NULL.
 * Requirements (GN_EINVAL otherwise): one source, Cin % 16 == 0, D*H*W <= 2^27, D*H*W*Cin*4 < 2^32 and
 *   Cout % 128 == 0 (csrc/unet_wino.hip: 4 x 8 x 8 tiles x 128 channels): Cin <= 256, D % 4 == H % 8 == W % 8 == 0;
 *   any other Cout % 32 == 0 (csrc/unet_wino32.hip, round 6: 8 x 8 x 8 tiles x one 32-wide column block -- the encoder's second convolution 128 -> 32 at
 *   full resolution, components/unet3d.py:127-144, and the last decoder's convolutions, :291,330): Cin <= 128, D % 8 == H % 8 == W % 8 == 0.
 * tile_active / kconst / kreach / occ_ws as gn_conv3d_gcr_split (tile granularity of the occupancy-aware list: the kernel's own).
 * gn_conv3d_gcr_split_wino_partial: the same layer with the polyphase partial of gn_upconv_partial added before the ReLU (Cout % 128 != 0 only; dense). */
size_t gn_conv_affine_pack_wino_bytes(int B, int Cin, int Cout);
int gn_conv_affine_pack_wino(const float *w, int Cin, int Cout, const float *a, const float *d, const double *sum, const double *sumsq, int64_t V,
                             const float *coff, int B, void *pack, size_t pack_bytes, float *stage_a, float *stage_d, float *out_scale,
                             float *kbias, void *ws, size_t ws_bytes, void *stream);
int gn_conv3d_gcr_split_wino(const float *src, int Cin, const float *a, const float *d, const void *pack, const float *out_scale,
                             const float *act_inv_scale, const float *kbias, int B, int D, int H, int W, int Cout, int relu, float *out,
                             double *out_sum, double *out_sumsq, const unsigned char *tile_active, const float *kconst, int kreach,
                             void *occ_ws, size_t occ_ws_bytes, void *stream);
int gn_conv3d_gcr_split_wino_partial(const float *src, int Cin, const float *a, const float *d, const void *pack, const float *out_scale,
                                     const float *act_inv_scale, const float *kbias, int B, int D, int H, int W, int Cout, int relu, float *out,
                                     double *out_sum, double *out_sumsq, const float *partial, void *stream);

/* The nearest-upsampled source of a decoder convolution in polyphase form (torch.cat((skip, interpolate(x, 'nearest'))) -> Conv3d,
 * components/unet3d.py:291,330): every fine output voxel (2i+pz, 2j+py, 2k+px) sees only a 2 x 2 x 2 block of coarse voxels, so the 27
 * fine taps merge into 8 coarse taps per output parity class (sums of weights, host side, fp64) -- 8/27 of the MACs of those channels,
 * and the coarse halo is staged once for all classes instead of once per fine voxel.  src1 [B][Dc][Hc][Wc][C1]; a, d [B][C1]: the
 * GroupNorm affine of THESE channels (contiguous slices of gn_groupnorm_affine's result); wp: merged weights in fragment order
 * [C1/16][8 taps][8 classes][Cout/32][2 planes][64 lanes] x 16 B with out_scale [8 * Cout] (garmentnets_amd.ops.pack_upconv_weight);
 * partial [B][Dc][Hc][Wc][8 * Cout] in true units, no activation.  Arithmetic: the two-plane split of gn_conv3d_gcr_split. */
int gn_upconv_partial(const float *src1, int C1, const float *a, const float *d, const void *wp, int mode, const float *out_scale,
                      const float *act_inv_scale, int B, int Dc, int Hc, int Wc, int Cout, float *partial, void *stream);

/* The pieces of create_conv's layer orders other than 'gcr' (components/unet3d.py:19-73: 'cr', 'crg', 'cl', 'ce', 'bcr', ...) that the fused conv
 * kernels do not cover -- a learnable conv bias, LeakyReLU(0.1) / ELU, a normalisation BEHIND the non-linearity:
 * y = act(x * a[b][c] + d[b][c] + bias[c]) over channel-last [B][V][C] (C % 4 == 0); a / d ([B][C], together) and bias ([C]) may be NULL;
 * act: 0 none, 1 ReLU, 2 LeakyReLU(0.1), 3 ELU(alpha 1).  In place when out == x. */
int gn_affine_act(const float *x, int B, int64_t V, int C, const float *a, const float *d, const float *bias, int act, float *out, void *stream);

/* MaxPool3d(2) -- components/unet3d.py:222.  in [B][D][H][W][C] -> out [B][D/2][H/2][W/2][C].
 * out_sum / out_sumsq: optional statistics of the pooled output (as gn_conv3d_gcr). */
int gn_maxpool3d_2(const float *in, int B, int D, int H, int W, int C, float *out, double *out_sum, double *out_sumsq,
                   void *stream);
/* ---------------------------------------------------------------------------------------------------------
 * Implicit decoder.
 * ------------------------------------------------------------------------------------------------------- */

/* Trilinear feature sampling.  replaces F.grid_sample(bilinear, border, align_corners=True) on a 5-D input --
 * networks/conv_implicit_wnf.py:135-145 (including its axis convention: query x -> LAST volume axis).
 * vol: one sample [D][H][W][C].  If query == NULL the queries are the lattice of predict.py:145-147:
 * q[i][j][k] = (i,j,k) * (1/(Q-1)), rows m0..m0+M of the flattened (Q,Q,Q) lattice. */
int gn_trilinear_sample(const float *vol, int D, int H, int W, int C, const float *query, int Q, int64_t m0, int64_t M,
                        float *out, int ldo, void *stream);
/* the query form for B volumes in one launch (round 6; the surface decoders of predict.py:184-187 for a whole batch): volume b = vol + b * vol_bstride
 * floats, its M queries query [b][M][3], its rows out [b][M][ldo]. */
int gn_trilinear_sample_batch(const float *vol, int B, int64_t vol_bstride, int D, int H, int W, int C, const float *query, int64_t M, float *out, int ldo,
                              void *stream);

/* Fused implicit decoder: trilinear sampling (as gn_trilinear_sample) + the 3-layer MLP of ImplicitWNFDecoder
 * (Linear -> ReLU -> BatchNorm1d per layer, widths [C0, N1, N2, OUT]) in one kernel -- networks/conv_implicit_wnf.py:128-149,
 * predict.py:145-157,184-187.  Activations stay in LDS; w1p / w2p are k-pair-major packs Wp[k/2][n][2] of the Linear
 * weights W[n][k]; w3 is [OUT][N2] row-major; (s*, t*) are the folded eval-BatchNorm scale/shift (NULL = no BN).
 * If xin != NULL the rows xin[M][C0] (ld ldxin) are used instead of sampling (two-kernel form: gn_trilinear_sample first).
 * Constraints: C0 % 32 == 0, N1 and N2 multiples of 256, OUT <= 4 (otherwise use gn_trilinear_sample + gn_linear).
 * run_if: NULL, or a device float: the launch does nothing unless *run_if != 0 (pairs with gn_implicit_decode_split, which skips
 * the garments gn_decoder_input_scale marks unsafe for the fp16 planes: the choice is made on the device, without a host sync). */
int gn_implicit_decode(const float *vol, int D, int H, int W, int C0, const float *xin, int ldxin, const float *query, int Q, int64_t m0, int64_t M,
                       const float *w1p, const float *b1, const float *s1, const float *t1, int N1, const float *w2p,
                       const float *b2, const float *s2, const float *t2, int N2, const float *w3, const float *b3,
                       const float *s3, const float *t3, int OUT, float *out, int ldo, const float *run_if, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Isosurface.
 * ------------------------------------------------------------------------------------------------------- */

/* Gaussian gradient magnitude (sigma, mode='nearest').  replaces scipy.ndimage.gaussian_gradient_magnitude --
 * predict.py:162-163.  Bit-compatible accumulation order (fp64 taps, fp32 stores).  A kernel radius int(4 sigma + 0.5) <= 2 (the
 * reference's sigma = 0.5) runs as ONE fused launch (tile + halo in LDS, each voxel read once and written once) and does not touch
 * tmp (may be NULL); larger radii run 8 separable passes through tmp: 2 volumes of workspace. */
int gn_ggm3d(const float *vol, int n0, int n1, int n2, double sigma, float *tmp, float *out, void *stream);

/* min / max of a float array (device result [2]); level-range check of skimage marching_cubes (measure/_marching_cubes_lewiner.py: volume.min() /
 * volume.max(): numpy's NaN-propagating reductions -- a NaN anywhere gives NaN in the record, round 6; before: fminf / fmaxf). */
int gn_minmax(const float *x, int64_t n, float *out2, void *stream);
/* the same for `batch` volumes of one shape in one set of launches (one volume per blockIdx.y): vol / out [batch][n0][n1][n2],
 * tmp [2][batch][n0][n1][n2]; x [batch][n] (n % 4 == 0), out2 [batch][2].  What predict.py:160-181 does garment by garment. */
int gn_ggm3d_batch(const float *vol, int batch, int n0, int n1, int n2, double sigma, float *tmp, float *out, void *stream);
int gn_minmax_batch(const float *x, int batch, int64_t n, float *out2, void *stream);
/* gn_ggm3d_batch with two options (round 6; predict.py:160-171 needs the volume's range next to its gradient magnitude):
 *   range2 != NULL  [batch][2]: (min, max) of every volume, NaN-propagating, computed from the values the fused launch stages anyway (its tiles and
 *                   edge-replicated halos are voxels of the volume): every wave leaves one pair in range_ws (gn_ggm3d_range_workspace_bytes() bytes, contents
 *                   scratch), a one-workgroup-per-volume launch folds them -- no pass of its own over the volume, no atomics, equal to gn_minmax_batch bit
 *                   for bit; a kernel radius above 2 (8-pass form) calls gn_minmax_batch and ignores range_ws.
 *   accum_bits      64: scipy's arithmetic (fp64 taps), bit for bit, = gn_ggm3d_batch.  32: the same operation order accumulated in fp32 (fused form
 *                   only): 1e-6-class against scipy, not bit-compatible -- an opt-in for callers that hold floats to a tolerance. */
size_t gn_ggm3d_range_workspace_bytes(int batch, int n0, int n1, int n2);
int gn_ggm3d_batch_ex(const float *vol, int batch, int n0, int n1, int n2, double sigma, float *tmp, float *out, int accum_bits, float *range2,
                      void *range_ws, size_t range_ws_bytes, void *stream);

/* Lewiner marching cubes (MC33).  replaces skimage.measure.marching_cubes(method='lewiner') -- predict.py:172-177.
 * gn_mc33_workspace_bytes: bytes of `ws` for a volume of n0*n1*n2.
 * gn_mc33: classify + scan + emit.  verts [cap_v][3] float32 (array-axis order, voxel units), faces [cap_f][3]
 * int32 ('ascent' orientation), normals [cap_v][3], values [cap_v]; counts_dev[2] = {V, F} (device int64) are
 * always written; entries beyond the capacities are dropped (caller re-runs with larger buffers). */
size_t gn_mc33_workspace_bytes(int n0, int n1, int n2);
int gn_mc33(const float *vol, int n0, int n1, int n2, double level, void *ws, size_t ws_bytes, float *verts,
            int32_t *faces, float *normals, float *values, int64_t cap_v, int64_t cap_f, int64_t *counts_dev,
            void *stream);
/* `batch` volumes of one shape at one level in one set of launches: vol [batch][n0][n1][n2], verts / normals [batch][cap_v][3], faces
 * [batch][cap_f][3], values [batch][cap_v], counts_dev [batch][2]; every volume's result is what gn_mc33 gives for it alone. */
size_t gn_mc33_batch_workspace_bytes(int batch, int n0, int n1, int n2);
int gn_mc33_batch(const float *vol, int batch, int n0, int n1, int n2, double level, void *ws, size_t ws_bytes, float *verts,
                  int32_t *faces, float *normals, float *values, int64_t cap_v, int64_t cap_f, int64_t *counts_dev, void *stream);
/* gn_mc33_batch with its four stages (classify, scan + counts, vertices + normals / values, faces) bracketed by HIP events: the call
 * SYNCHRONISES and fills stage_ms (host float[4], milliseconds).  Measurement hook of bench.py's hbm_members; same results. */
int gn_mc33_batch_profiled(const float *vol, int batch, int n0, int n1, int n2, double level, void *ws, size_t ws_bytes, float *verts,
                           int32_t *faces, float *normals, float *values, int64_t cap_v, int64_t cap_f, int64_t *counts_dev, void *stream,
                           float *stage_ms);

/* out[i] = vol[(uint32)(verts[i]/spacing)] (float64 division, truncation) -- predict.py:179-181.
 * verts_vox: float32 voxel-unit vertices as produced by gn_mc33; spacing applied in fp64 as numpy does. */
int gn_gather_nn(const float *vol, int n0, int n1, int n2, const float *verts_vox, int64_t nv, double spacing,
                 float *out, void *stream);
/* batched: vol [batch][n0][n1][n2], verts_vox [batch][nv][3] (nv rows per volume, padded), out [batch][nv] */
int gn_gather_nn_batch(const float *vol, int batch, int n0, int n1, int n2, const float *verts_vox, int64_t nv, double spacing,
                       float *out, void *stream);

/* verts_out[i] = (float)((double)verts_vox[i] * spacing): the float32 query points predict.py:184 feeds to the
 * surface decoder (mc_verts.astype(np.float32)). */
int gn_scale_verts(const float *verts_vox, int64_t nv, double spacing, float *verts_out, void *stream);

/* Mesh compaction = delete_invalid_verts (common/marching_cubes_util.py:38-52; the hole-prediction head of predict.py:202-209 and
 * eval.py:39): keep the faces whose three vertices are flagged on_surface, keep the vertices those faces use in ascending raw index
 * (np.unique order), renumber the faces.  verts: [V][3] of vert_bytes / 3 byte scalars (12 = float32, 24 = float64); faces [F][3]
 * int32; on_surface [V] bytes (0 / non-0).  out_verts / out_faces must hold V / F rows; counts (device int64[3]) = kept (V', F') and a flag:
 * 1 if some face index was outside [0, V) (numpy raises IndexError there; such a face is dropped without touching memory). */
size_t gn_mesh_compact_workspace_bytes(int64_t V, int64_t F);
int gn_mesh_compact(const void *verts, int vert_bytes, const int32_t *faces, const unsigned char *on_surface, int64_t V, int64_t F,
                    void *ws, size_t ws_bytes, void *out_verts, int32_t *out_faces, int64_t *counts, void *stream);

/* Largest connected component of a triangle mesh = the filter of the reference's hole removal (eval.py:497-503, :536-545:
 * igl.adjacency_matrix + igl.connected_components + np.argmax(cc_sizes), followed by delete_invalid_verts = gn_mesh_compact).  Lock-free
 * union-find over the mesh edges that always hooks the larger root under the smaller: a component's label is its LOWEST vertex index whatever
 * the interleaving (libigl numbers components in that order), and of several largest components the one with the lowest label wins (np.argmax
 * takes the first maximum).  faces [F][3] int32; mask [V] bytes (1 = vertex in the winning component); label [V] int32 or NULL; info = device
 * int64[4]: number of components, size of the winner, its label (-1 for V = 0), 1 if a face index was outside [0, V). */
size_t gn_mesh_largest_component_workspace_bytes(int64_t V);
int gn_mesh_largest_component(const int32_t *faces, int64_t F, int64_t V, void *ws, size_t ws_bytes, unsigned char *mask, int32_t *label,
                              int64_t *info, void *stream);

/* The decoder MLP of gn_implicit_decode for the shipped shape [128, 256, 256, OUT<=4] on the 16-bit matrix cores: fp32 operands
 * split into two fp16 planes (3 MFMA products per fp32 product, fp32 accumulation -- the f16x2 arithmetic of
 * gn_conv3d_gcr_split), activations chained through registers, weights streamed through an LDS ring
 * (csrc/decode_split.hip).  xin: pre-sampled rows [M][ldxin] (gn_trilinear_sample); wpack / tab: weight stages and epilogue
 * tables from garmentnets_amd.ops.pack_decode_split (per-hidden-unit power-of-two weight scales folded in); xscale: NULL or a
 * device record {s, 1/s, unsafe, 0} from gn_decoder_input_scale -- rows and biases are multiplied by s, the output sum by 1/s (exact:
 * ReLU is positively homogeneous), so un-normalised inputs of any magnitude are split at O(1); unsafe != 0: the launch does nothing
 * (run gn_implicit_decode with run_if = &xscale[2] right after it).  A hidden value beyond fp16's range in the
 * scaled units reaches the output as NaN (never as a wrong finite number); callers re-run such a batch with gn_implicit_decode.
 * replaces the three Linear/ReLU/BatchNorm1d blocks of ImplicitWNFDecoder.forward -- networks/conv_implicit_wnf.py:128-149. */
int gn_implicit_decode_split(const float *xin, int ldxin, int64_t M, const void *wpack, const float *tab, const float *xscale,
                             int C0, int N1, int N2, int OUT, float *out, int ldo, void *stream);
/* B row sets of M rows each in ONE launch (round 6): xin [B][M][ldxin], out [B][M][ldo], xscale NULL or [B][4] (one gn_decoder_input_scale record per row set;
 * a set marked unsafe is skipped and left to gn_implicit_decode_batch(run_if = xscale + 2, run_if_stride = 4)).  Row for row the single call's results. */
int gn_implicit_decode_split_batch(const float *xin, int ldxin, int64_t M, int B, const void *wpack, const float *tab, const float *xscale,
                                   int C0, int N1, int N2, int OUT, float *out, int ldo, void *stream);
/* gn_implicit_decode on B sets of M pre-sampled rows (the gated fp32 twin of the call above): run_if NULL or one flag per row set at run_if[b * run_if_stride]. */
int gn_implicit_decode_batch(const float *xin, int ldxin, int64_t M, int B, int C0, const float *w1p, const float *b1, const float *s1, const float *t1,
                             int N1, const float *w2p, const float *b2, const float *s2, const float *t2, int N2, const float *w3, const float *b3,
                             const float *s3, const float *t3, int OUT, float *out, int ldo, const float *run_if, int run_if_stride, void *stream);

/* gn_implicit_decode_split with the lattice sampler INSIDE the decoder kernel (reference call site
 * networks/conv_implicit_wnf.py:137-149 under the lattice loop of predict.py:145-157): rows m0 .. m0+M-1 of the (Q,Q,Q) lattice are sampled
 * (trilinear, border, align_corners: gn_trilinear_sample's arithmetic) from the channel-last volume vol [D][H][W][32] by the kernel that
 * multiplies them -- the next tile's corner gathers ride under the current tile's MFMAs; no sampled-row buffer exists.  Folded scalar decoder
 * only ([32, 256, 256, 1]); D*H*W*128 < 2^32.  Bit-identical to gn_trilinear_sample + gn_implicit_decode_split. */
int gn_implicit_decode_lattice_split(const float *vol, int D, int H, int W, int C0, int Q, int64_t m0, int64_t M, const void *wpack,
                                     const float *tab, const float *xscale, int N1, int N2, int OUT, float *out, int ldo, void *stream);

/* Input scale of gn_implicit_decode_split for B garments from the per-(sample, channel) sums of squares [B][C] (fp64) of the volume
 * the rows are sampled from (V voxels per channel; the statistics gn_conv3d_gcr* emit): s = 2^k with (largest channel rms) * s in
 * [1, 2), clamped to smax (pack_decode_split: keeps the scaled biases below 2^13).  out4: [B][4] = {s, 1/s, unsafe, 0}; unsafe = 1
 * when the clamp costs more than 2^4 (biases dwarf weights x activations: the fp16 planes would lose fp32-class accuracy). */
int gn_decoder_input_scale(const double *sumsq, int64_t V, int B, int C, float smax, float *out4, void *stream);

/* ---------------------------------------------------------------------------------------------------------
 * Widening: evaluation helpers.
 * ------------------------------------------------------------------------------------------------------- */

/* Exact 1-nearest-neighbour of every query in `ref` (brute force, fp32 squared distance, ties -> lowest index).
 * replaces the scipy cKDTree.query(k=1) calls of the Chamfer metrics -- eval.py:259-263,381-385.
 * idx [nq] int32, d2 [nq] squared distance. */
int gn_nearest_neighbor(const float *query, int64_t nq, const float *ref, int64_t nr, int32_t *idx, float *d2, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GARMENTNETS_HIP_H */
