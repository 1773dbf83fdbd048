/*
 * graspnet_hip.h -- C ABI of the MI355X (gfx950) GraspNet inference library
 * (libgraspnet_hip.so).
 *
 * This is the drop-in boundary: one entry point per function of the
 * reference's native operator modules
 *   pointnet2._ext            (pointnet2/_ext_src/src/bindings.cpp:12-27)
 *   knn_pytorch.knn_pytorch   (knn/src/vision.cpp:3-5)
 * plus the fused stage entry points used behind models/graspnet.py's
 * GraspNet.forward / pred_decode.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless said otherwise;
 *     tensors are dense, row-major, in the reference's layouts;
 *   - inputs are borrowed and never modified; outputs are fully written by
 *     the callee (no pre-zeroing needed, unlike ball_query.cpp:24-26);
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); calls
 *     only enqueue work, they never synchronise or allocate;
 *   - return 0 on success, a negative gn_status otherwise; nothing calls
 *     exit() (the reference's CUDA_CHECK_ERRORS does, cuda_utils.h:35-44);
 *   - floating-point tests that decide an index (distance / radius / height
 *     comparisons) are evaluated exactly as written in the reference kernels,
 *     left to right, one rounding per operation, no FMA contraction.
 */
#ifndef GRASPNET_HIP_H
#define GRASPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *gn_stream_t;

typedef enum gn_status {
  GN_OK = 0,
  GN_ERR_INVALID_ARG = -1,   /* null pointer, negative size, unsupported shape */
  GN_ERR_LAUNCH = -2,        /* hipGetLastError() != hipSuccess after a launch  */
  GN_ERR_NO_DEVICE = -3,
  GN_ERR_WORKSPACE = -4      /* workspace too small                              */
} gn_status;

/* Human-readable text for the last error on this thread ("" if none). */
const char *gn_last_error(void);
/* Library / build identification, e.g. "graspnet_hip 0.5 gfx950". */
const char *gn_version(void);
/* Binary interface revision: bumped whenever an existing entry point changes its parameter list or a parameter's type
 * (revision 2: `voxel_size` of gn_collision_detect / gn_collision_detect_batched became a double; revision 3: the
 * evaluation-policy entry gn_set_fp_policy, gn_stream_capture_id, voxel_status of gn_grasp_postprocess reported).  A host
 * built against this header checks  gn_abi_version() == GN_ABI_VERSION  once after loading the library. */
#define GN_ABI_VERSION 3
int gn_abi_version(void);
/* *id = 0 when `stream` is not being captured into a hipGraph, else the running capture's unique id (> 0).  For host-side
 * caches of device buffers (graspnet_amd/pointnet2_ext.py keeps the last grid / sampling call per stream): an entry enqueued
 * outside a capture must not be reused inside one (the work that fills it would be missing from the graph), and one enqueued
 * inside a capture only inside that same capture. */
int gn_stream_capture_id(gn_stream_t stream, unsigned long long *id);

/* Evaluation policy of the float expressions that DECIDE AN INDEX -- the squared distances of sampling_gpu.cu:108-109,
 * ball_query_gpu.cu:36-37, interpolate_gpu.cu:37, knn.cu:89-90 and the rotated coordinates of cylinder_query_gpu.cu:39-45:
 *   0 (default)  as written: one rounding per operation, left to right, no FMA contraction -- what a CPU build of the same
 *                expressions computes (the reference's own knn_cpu.cpp);
 *   1            contracted the way LLVM / NVVM fuse the expression tree:  a*a + b*b + c*c = fma(c, c, fma(a, a, b*b));
 *   2            contracted in the other plausible order:                  a*a + b*b + c*c = fma(c, c, fma(b, b, a*a)).
 * The reference builds its CUDA kernels with plain `nvcc -O2` (pointnet2/setup.py:24-27), i.e. --fmad=true, so ITS binary
 * contracts these expressions; which order it takes cannot be observed without nvcc.  A maintainer with a CUDA box picks
 * the policy whose indices match theirs (profiles/parity_r05.json holds how many indices each choice moves).  Process-wide;
 * the GN_FP environment variable (as_written | contracted | contracted_b) sets the initial value.  Affects
 * gn_furthest_point_sampling, gn_ball_query*, gn_cylinder_query*, gn_three_nn, gn_knn and gn_graspnet_forward. */
int gn_set_fp_policy(int policy);
int gn_get_fp_policy(void);

/* ------------------------------------------------------------------ *
 *  pointnet2._ext operators                                          *
 * ------------------------------------------------------------------ */

/* The operators whose results depend on the evaluation policy also exist under policy-suffixed names (`_fp1`, `_fp2`: the
 * copies compiled with -DGN_FP=1 / 2); the plain names dispatch to them, nothing else should call them. */

/* furthest_point_sampling(points (B,N,3) f32, nsamples) -> idx (B,M) i32
 * replaces sampling.cpp:70-91 + sampling_gpu.cu:74-234.
 * Semantics kept: idx[0]=0; points with x^2+y^2+z^2 <= 1e-3 are never
 * selected; ties resolved in the reference's block-tree order for block size
 * opt_n_threads(N).  `workspace` must hold gn_fps_workspace_bytes(B,N) bytes
 * (the reference's `temp` scratch, sampling.cpp:78-80; for clouds beyond one
 * CU's registers the region-sorted copy of csrc/fps_cell.hip); it may be NULL
 * when that function returns 0, and for N <= 24576 in any case (a call that
 * asks for tie_out then runs a kernel that does not track ties and reports 0).
 * new_xyz (B,M,3) is optional (NULL to skip): the
 * sampled coordinates, i.e. gather_points() fused in.
 * Optional, for the fused path (NULL otherwise):
 *   tie_out (B) i32   : the first iteration whose maximum was shared by points at different
 *                       locations (INT_MAX if none, 0 if the kernel variant does not track it);
 *   prefix_ok (B) i32 : caller's promise that `xyz` is the FPS order produced by a call whose
 *                       tie_out was prefix_ok: where prefix_ok[b] >= M the result is the identity
 *                       prefix (provably what the full computation returns,
 *                       models/backbone.py:110-119) and is written directly. */
size_t gn_fps_workspace_bytes(int b, int n);
/* 1 when a tie_out request does not change the kernel that serves n points, 0 when it moves the call onto the (slightly
 * slower) tie-tracking kernel: lets a caller that only MIGHT chain calls decide whether the report is worth asking for. */
int gn_fps_tie_is_free(int n);
/* For callers that chain sampling calls WITHOUT the fused path (the reference model does: backbone.py:105-121 samples the
 * 2048 sampled points again for 1024, ...): tie2[b] = tie1[b] if pts2 (B,n2,3) is bit for bit pts1[inds1[0..n2)] -- the
 * points an earlier call on pts1 (B,n1,3) picked, in its order, inds1 (B,m1>=n2), tie1 its tie_out -- else 0.  tie2 is then a
 * valid `prefix_ok` for sampling pts2 (and a valid tie1 for the next link of the chain). */
int gn_fps_prefix_check(const float *pts2, const float *pts1, const int32_t *inds1, const int32_t *tie1, int b, int n2,
                        int n1, int m1, int32_t *tie2, gn_stream_t stream);
int gn_furthest_point_sampling(const float *xyz, int b, int n, int m, void *workspace,
                               int32_t *idx, float *new_xyz, int32_t *tie_out, const int32_t *prefix_ok,
                               gn_stream_t stream);

/* gather_points(points (B,C,N) f32, idx (B,M) i32) -> (B,C,M)   sampling.cpp:20-42 */
int gn_gather_points(const float *points, const int32_t *idx, int b, int c, int n, int m,
                     float *out, gn_stream_t stream);
/* gather_points_grad(grad_out (B,C,M), idx (B,M), n) -> (B,C,N)  sampling.cpp:44-68
 * (float atomic scatter-add like the reference: sum order not reproducible) */
int gn_gather_points_grad(const float *grad_out, const int32_t *idx, int b, int c, int n, int m,
                          float *grad_points, gn_stream_t stream);

/* ball_query(new_xyz (B,M,3), xyz (B,N,3), radius, nsample) -> idx (B,M,nsample) i32
 * replaces ball_query.cpp:13-37 + ball_query_gpu.cu:14-59: first `nsample`
 * points in index order with d2 < radius^2 (strict); unfilled slots repeat the
 * first hit; a ball with no hit yields zeros. */
int gn_ball_query(const float *new_xyz, const float *xyz, int b, int n, int m, float radius,
                  int nsample, int32_t *idx, gn_stream_t stream);

/* cylinder_query(new_xyz (B,M,3), xyz (B,N,3), rot (B,M,9), radius, hmin, hmax, nsample)
 * -> idx (B,M,nsample) i32; replaces cylinder_query.cpp:10-37 + cylinder_query_gpu.cu:9-68. */
int gn_cylinder_query(const float *new_xyz, const float *xyz, const float *rot, int b, int n, int m,
                      float radius, float hmin, float hmax, int nsample, int32_t *idx,
                      gn_stream_t stream);

/* group_points(points (B,C,N), idx (B,NP,NS)) -> (B,C,NP,NS)   group_points.cpp:17-40 */
int gn_group_points(const float *points, const int32_t *idx, int b, int c, int n, int npoints,
                    int nsample, float *out, gn_stream_t stream);
/* group_points_grad(grad_out (B,C,NP,NS), idx, n) -> (B,C,N)    group_points.cpp:42-66
 * (float atomic scatter-add like the reference) */
int gn_group_points_grad(const float *grad_out, const int32_t *idx, int b, int c, int n,
                         int npoints, int nsample, float *grad_points, gn_stream_t stream);

/* three_nn(unknown (B,N,3), known (B,M,3)) -> dist2 (B,N,3) f32 SQUARED, idx (B,N,3) i32
 * replaces interpolate.cpp:19-45 + interpolate_gpu.cu:14-75. */
int gn_three_nn(const float *unknown, const float *known, int b, int n, int m, float *dist2,
                int32_t *idx, gn_stream_t stream);
/* three_interpolate(points (B,C,M), idx (B,N,3), weight (B,N,3)) -> (B,C,N)  interpolate.cpp:47-75 */
int gn_three_interpolate(const float *points, const int32_t *idx, const float *weight, int b, int c,
                         int m, int n, float *out, gn_stream_t stream);
/* three_interpolate_grad(grad_out (B,C,N), idx, weight, m) -> (B,C,M)        interpolate.cpp:77-104 */
int gn_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight, int b,
                              int c, int n, int m, float *grad_points, gn_stream_t stream);

/* DETERMINISTIC forms of the three gradient operators (csrc/grad_det.hip): the scatter-add of the reference
 * (sampling_gpu.cu:39-62, group_points_gpu.cu:48-80, interpolate_gpu.cu:121-159 use float atomicAdd, whose result
 * depends on retirement order) as a sorted-segment sum -- every destination adds its entries in ascending entry
 * order, bit-identical to a sequential loop and identical from run to run.  Same arguments as the atomic forms
 * above plus a workspace of gn_grad_workspace_bytes(b, destinations, entries per cloud) bytes
 * (entries: m for gather, npoints*nsample for group, 3*n for three_interpolate whose destinations are m). */
size_t gn_grad_workspace_bytes(int b, int n, long entries);
int gn_gather_points_grad_det(const float *grad_out, const int32_t *idx, int b, int c, int n, int m, void *workspace,
                              float *grad_points, gn_stream_t stream);
int gn_group_points_grad_det(const float *grad_out, const int32_t *idx, int b, int c, int n, int npoints, int nsample,
                             void *workspace, float *grad_points, gn_stream_t stream);
int gn_three_interpolate_grad_det(const float *grad_out, const int32_t *idx, const float *weight, int b, int c, int n,
                                  int m, void *workspace, float *grad_points, gn_stream_t stream);

/* ------------------------------------------------------------------ *
 *  knn_pytorch.knn_pytorch.knn                                       *
 * ------------------------------------------------------------------ */
/* knn(ref (B,D,NR) f32, query (B,D,NQ) f32, idx (B,K,NQ) i64 filled in place) ; 1-BASED indices,
 * distance ascending, lower ref index first among equal distances (knn.h:12-65, knn.cu:36-176,
 * knn_cpu.cpp:4-56).  scratch: B*K*NQ floats (the reference's per-call distance scratch, knn.h:33). */
int gn_knn(const float *ref, const float *query, int b, int d, int nr, int nq, int k, int64_t *idx,
           float *scratch, gn_stream_t stream);

/* ------------------------------------------------------------------ *
 *  Fused stages behind models/graspnet.py GraspNet.forward / pred_decode
 *  Activations between stages are POSITION-major ("channels last"):
 *  features (B, npoint, C).  Weights are BN-folded on the host and packed
 *  in MFMA fragment order (graspnet_amd/weights.py: pack_fragments):
 *    Wp[nt][kb][lane][s] = W[nt*32 + (lane&31)][kb*8 + 4*(lane>>5) + s]
 *  with COUT padded to a multiple of 32 (x4 waves where it has >= 4 tiles)
 *  and CIN to a multiple of 8; bias padded to the same COUT.
 * ------------------------------------------------------------------ */

/* Uniform-grid acceleration of the queries (exact same results as gn_ball_query / the cylinder queries):
 * gn_grid_build bins each cloud once into cells at least `reach` wide (workspace of
 * gn_grid_workspace_bytes(B,N) bytes: header, cell starts, cell-sorted coordinates and original indices);
 * gn_ball_query_grid then tests only the 3x3x3 cells around a centre and keeps the nsample (<= 64)
 * smallest hit indices in a sorted 64-lane register, so "first nsample hits in index order", padding and
 * zero rows are those of ball_query_gpu.cu:14-49.  `reach` must be >= radius. */
size_t gn_grid_workspace_bytes(int b, int n);
int gn_grid_build(const float *xyz, int b, int n, float reach, void *workspace, gn_stream_t stream);
int gn_ball_query_grid(const float *new_xyz, const void *grid, int b, int n, int m, float radius, int nsample,
                       int32_t *idx, gn_stream_t stream);
/* gn_cylinder_query (cylinder_query_gpu.cu:9-57, one depth) on the same grid: identical index lists; the cell window
 * covers the sphere that contains the cylinder.  nsample <= 64. */
int gn_cylinder_query_grid(const float *new_xyz, const float *rot, const void *grid, int b, int n, int m, float radius,
                           float hmin, float hmax, int nsample, int32_t *idx, gn_stream_t stream);

/* gn_cylinder_query_unique on the grid (any cell size; the kernel widens its cell window to the cylinder's
 * bounding sphere).  Same entries and masks; entries come grouped by minimum depth rather than in index
 * order (the consumer max-pools rows, so order is immaterial).  hmax_host must be ascending. */
int gn_cylinder_query_unique_grid(const float *new_xyz, const float *rot, const void *grid, int b, int n, int m,
                                  float radius, float hmin, const float *hmax_host, int ndepth, int nsample,
                                  int32_t *uidx, int32_t *ucount, gn_stream_t stream);

/* CloudCrop's four CylinderQueryAndGroup queries (models/modules.py:103-106,127-130) in one scan:
 * idx (B, M, ndepth, nsample) i32, each depth with the exact semantics of gn_cylinder_query.
 * hmax_host: HOST array of ndepth floats.  ndepth must be 4. */
int gn_cylinder_query_multi(const float *new_xyz, const float *xyz, const float *rot, int b, int n, int m,
                            float radius, float hmin, const float *hmax_host, int ndepth, int nsample,
                            int32_t *idx, gn_stream_t stream);

/* The same four queries as ONE de-duplicated list per seed: uidx (B, M, ndepth*nsample) i32 holds, in index
 * order, every point that is in at least one depth's list, packed as  k | (mask << 28)  with bit d of mask
 * <=> k is in the depth-d list; ucount (B, M) is the number of entries.  A depth without a hit (reference:
 * all-zero row = point 0 repeated) sets its bit on an entry for point 0.  The four reference lists are
 * recoverable exactly (entries with bit d, in order, then padding with the first).  n < 2^28. */
int gn_cylinder_query_unique(const float *new_xyz, const float *xyz, const float *rot, int b, int n, int m,
                             float radius, float hmin, const float *hmax_host, int ndepth, int nsample,
                             int32_t *uidx, int32_t *ucount, gn_stream_t stream);

/* Set-abstraction tail: group by idx, subtract centre, divide by radius, concat features,
 * SharedMLP(3 layers, BN folded, ReLU), max-pool over nsample
 * (pointnet2_utils.py:344-355 + pointnet2_modules.py:251-257).
 * xyz (B,N,3); feat (B,N,C) channels-last or NULL when C=0; new_xyz (B,NP,3); idx (B,NP,NS);
 * mlp = {c1,c2,c3}; wpacked/bias: HOST arrays of 3 device pointers; the first layer's input
 * columns are ordered [features(C) | xyz(3) | zero pad]; out (B,NP,c3) channels-last.
 * Supported: (C,mlp,NS) of models/backbone.py:30-64. */
int gn_sa_group_mlp_pool(const float *xyz, const float *feat, const float *new_xyz, const int32_t *idx,
                         int b, int n, int np, int ns, int c, float radius, const int *mlp,
                         const float *const *wpacked, const float *const *bias, float *out,
                         gn_stream_t stream);

/* The same tail with layer 1's feature part hoisted out of the per-neighbour work: W1f*feat_j + b1 does
 * not depend on the group, so it is computed once per point (gn_point_linear -> pfeat (B,N,mlp[0])) and
 * gathered as the accumulators' initial value; wpacked[0]/bias[0] hold only the K=8 coordinate block of
 * layer 1 ([xyz(3) | 0], zero bias).  Same k order => bit-identical to gn_sa_group_mlp_pool.
 * Supported: mlp = {128,128,256}, ns in {32,16} (SA2-SA4 of models/backbone.py:39-64). */
int gn_point_linear(const float *x, int rows, int cin, int cout, const float *wpacked, const float *bias,
                    float *out, gn_stream_t stream);
int gn_sa_hoisted_mlp_pool(const float *xyz, const float *pfeat, const float *new_xyz, const int32_t *idx,
                           int b, int n, int np, int ns, float radius, const int *mlp,
                           const float *const *wpacked, const float *const *bias, float *out,
                           gn_stream_t stream);

/* CloudCrop tail: group xyz by idx (B,M,ndepth,NS), subtract seed, rotate into the view frame,
 * SharedMLP[3,64,128,256], max-pool (pointnet2_utils.py:521-530 + models/modules.py:131-140).
 * out (B, M*ndepth, 256) channels-last, group order [seed][depth]. */
int gn_crop_group_mlp_pool(const float *xyz, const float *seed_xyz, const float *rot, const int32_t *idx,
                           int b, int n, int m, int ndepth, int ns, const int *mlp,
                           const float *const *wpacked, const float *const *bias, float *out,
                           gn_stream_t stream);

/* CloudCrop tail on the de-duplicated lists of gn_cylinder_query_unique: every (seed, point) row goes
 * through SharedMLP[3,64,128,256] once and is max-pooled into the depths named by its mask --
 * bit-identical to gn_crop_group_mlp_pool on the four full lists, with ~2.4x fewer MFMAs.
 * out (B, M*ndepth, 256). */
int gn_crop_unique_mlp_pool(const float *xyz, const float *seed_xyz, const float *rot, const int32_t *uidx,
                            const int32_t *ucount, int b, int n, int m, int ndepth, int ns, const int *mlp,
                            const float *const *wpacked, const float *const *bias, float *out,
                            gn_stream_t stream);

/* gn_crop_unique_mlp_pool with layers 2-3 on the bf16 matrix cores by exact operand splitting (each fp32 operand =
 * three bf16 terms, six products per 16 k, fp32 accumulate: csrc/mlp_split.h; error vs float64 no larger than the
 * fp32 FMA chain's).  wpacked[0]: layer 1 in the fp32 fragment layout above; wpacked[1..2]: split fragments
 *   Wq[nt][kb][plane][lane][j] (bf16) = W_plane[nt*32 + (lane&31)][kb*16 + 8*(lane>>5) + j],  plane = hi, mid, lo
 * (graspnet_amd/weights.py: pack_split).  Same rows, masks and output as gn_crop_unique_mlp_pool. */
int gn_crop_unique_mlp_pool_split(const float *xyz, const float *seed_xyz, const float *rot, const int32_t *uidx,
                                  const int32_t *ucount, int b, int n, int m, int ndepth, int ns, const int *mlp,
                                  const void *const *wpacked, const float *const *bias, float *out,
                                  gn_stream_t stream);

/* Set-abstraction tails with layers 2-3 on the bf16 matrix cores (same splitting as above; wpacked[0] fp32 fragments of
 * layer 1 -- for the hoisted form its K=8 coordinate block with zero bias --, wpacked[1..2] split fragments).
 * gn_sa_group_mlp_pool_split: SA1 only (no input features, mlp {64,64,128}, nsample 64).
 * gn_sa_hoisted_mlp_pool_split: SA2-SA4 (mlp {128,128,256}, nsample 32 or 16), pfeat from gn_point_linear. */
int gn_sa_group_mlp_pool_split(const float *xyz, const float *new_xyz, const int32_t *idx, int b, int n, int np, int ns,
                               float radius, const int *mlp, const void *const *wpacked, const float *const *bias,
                               float *out, gn_stream_t stream);
int gn_sa_hoisted_mlp_pool_split(const float *xyz, const float *pfeat, const float *new_xyz, const int32_t *idx, int b,
                                 int n, int np, int ns, float radius, const int *mlp, const void *const *wpacked,
                                 const float *const *bias, float *out, gn_stream_t stream);

/* PointnetFPModule tail (pointnet2_modules.py:395-416): inverse-distance weights from three_nn's
 * dist2/idx, three_interpolate, concat skip features, SharedMLP(2 layers).
 * known_feat (B,m,ck), skip_feat (B,n,cs) channels-last -> out (B,n,mlp[1]). */
int gn_fp_interp_mlp(const float *known_feat, const float *skip_feat, const float *dist2,
                     const int32_t *idx3, int b, int n, int m, int ck, int cs, const int *mlp,
                     const float *const *wpacked, const float *const *bias, float *out,
                     gn_stream_t stream);

/* The per-point chains below on the bf16 matrix cores (csrc/mlp_point_split.hip): same arguments and results as
 * gn_fp_interp_mlp / gn_approach_mlp / gn_heads_mlp / gn_point_linear, every layer's weights as split fragments
 * (graspnet_amd/weights.py: pack_split, K padded to a multiple of 32). */
int gn_fp_interp_mlp_split(const float *known_feat, const float *skip_feat, const float *dist2, const int32_t *idx3, int b,
                           int n, int m, int ck, int cs, const int *mlp, const void *const *wpacked,
                           const float *const *bias, float *out, gn_stream_t stream);
int gn_approach_mlp_split(const float *x, int rows, int cin, int nout, const void *const *wpacked, const float *const *bias,
                          float *out, gn_stream_t stream);
int gn_heads_mlp_split(const float *x, int rows, int cin, int nout, const void *const *wpacked, const float *const *bias,
                       float *out, gn_stream_t stream);
int gn_point_linear_split(const float *x, int rows, int cin, int cout, const void *wpacked, const float *bias, float *out,
                          gn_stream_t stream);

/* Tile shapes of the per-point chains above (per calling thread; returns the previous setting):
 *   0 (default) "latency": every launch spreads over as many CUs as it has 32-row tiles -- the shortest launch when nothing runs
 *     beside it (one batch at a time, hipGraph replay);
 *   1 "throughput": 64-row fp32 tiles for ApproachNet / feature propagation and 128-row tiles for the heads (csrc/
 *     mlp_point_wide.hip: half / a quarter of the weight bytes per row, half as many workgroups) -- fewer CU-microseconds per
 *     launch, for callers that keep several batches in flight on several streams.  Outputs are bit-identical either way. */
int gn_set_point_tile_policy(int policy);

/* ApproachNet conv1/bn1/relu, conv2/bn2/relu, conv3 (models/modules.py:55-57):
 * x (rows,256) -> out (rows, nout = 2+num_view) = [objectness | view scores], 3 <= nout <= 320; the packed layers are
 * 256 -> 256 -> 384 -> 384 whatever nout is (layer 3's K padded to 320; zeros beyond nout). */
int gn_approach_mlp(const float *x, int rows, int cin, int nout, const float *const *wpacked,
                    const float *const *bias, float *out, gn_stream_t stream);

/* OperationNet + ToleranceNet conv chains (models/modules.py:171-175, 216-220) as one
 * 256->256->256->48 chain: out (rows,48) = [score(12) | angle cls(12) | width(12) | tolerance(12)]. */
int gn_heads_mlp(const float *x, int rows, int cin, int nout, const float *const *wpacked,
                 const float *const *bias, float *out, gn_stream_t stream);

/* models/modules.py:64-76: top view = first arg-max over view scores; template / rotation lookup.
 * feat (rows, ldf) = [objectness(2) | nview scores]; views (nview,3); view_rot (nview,9). */
int gn_approach_post(const float *feat, int ldf, int rows, int nview, const float *views,
                     const float *view_rot, int64_t *top_inds, float *top_score, float *top_xyz,
                     float *top_rot, gn_stream_t stream);

/* models/graspnet.py:77-134 pred_decode for a whole batch: grasps (B,M,17) padded rows
 * [score,width,height,depth,R(9),center(3),obj_id], counts (B) valid rows per scene (rows keep seed
 * order; rows >= count are zero).  head (B,M,ndepth,4*nangle); rot_table (nview*nangle*9). */
int gn_pred_decode(const float *feat, int ldf, const float *head, const int64_t *top_inds,
                   const float *seed_xyz, const float *rot_table, int b, int m, int ndepth, int nangle,
                   float *grasps, int32_t *counts, gn_stream_t stream);

/* ------------------------------------------------------------------ *
 *  Whole path in one call (SURVEY.md 8b "fused entry"): what a C/C++ host (e.g. a ROS node) binds
 *  instead of GraspNet.forward (models/graspnet.py:70-75) + pred_decode (:77-134).  Host-side
 *  sequencing only: it enqueues the stages above on `stream` (grid queries, hoisted SA tails,
 *  de-duplicated CloudCrop); nothing allocates or synchronises, so the call is hipGraph-capturable.
 *  The network constants are those of models/backbone.py:30-64 and models/graspnet.py:31-45
 *  (2048/1024/512/256 centres, 300 views, 12 angles, 4 depths, 64 crop samples).
 * ------------------------------------------------------------------ */
typedef struct gn_mlp_weights {      /* up to 3 packed layers (device pointers), see the layout note above */
  const float *w[3];
  const float *b[3];
} gn_mlp_weights;

typedef struct gn_weights {
  gn_mlp_weights sa[4];      /* sa[0]: SA1 layers on [xyz | pad]; sa[1..3]: hoisted tails, w[0] = K=8 coordinate block */
  gn_mlp_weights sa_pre[4];  /* sa_pre[l].w[0]/b[0] (l>=1): layer 1's per-point feature part (gn_point_linear) */
  gn_mlp_weights fp[2];      /* fp1 (sa4->sa3), fp2 (sa3->sa2): 2 layers each */
  gn_mlp_weights approach;   /* 256->256->302->302, layer 3 K padded to 320 */
  gn_mlp_weights crop;       /* 3->64->128->256 */
  gn_mlp_weights heads;      /* OperationNet+ToleranceNet merged 256->256->256->48 */
  const float *views;        /* (300,3)  loss_utils.py:41-66 template */
  const float *view_rot;     /* (300,9)  rotation of each template view, in-plane angle 0 */
  const float *decode_rot;   /* (300*12,9) rotation of (-view, angle class) used by pred_decode */
  float cylinder_radius, hmin, hmax[4];
  /* matrix-core path (csrc/mlp_split.h): when use_split != 0 the grouped layers run from these instead of sa[] / crop:
   * w[0] fp32 fragments of layer 1 (hoisted levels: the K=8 coordinate block), w[1..2] split-bf16 fragments */
  int use_split;
  gn_mlp_weights sa_split[4];
  gn_mlp_weights crop_split;
  gn_mlp_weights sa_pre_split[4];  /* [l].w[0]/b[0], l >= 1 */
  gn_mlp_weights fp_split[2], approach_split, heads_split;   /* all layers split */
  int num_view;              /* approach views V (models/graspnet.py:64 num_view); 0 = the reference's 300.  1 <= V <= 318: the
                              * ApproachNet layers are always packed as 256 -> 256 -> 384 -> 384 with K of layer 3 padded to
                              * 320 (zero rows / columns beyond 2+V), views / view_rot / decode_rot hold V rows */
} gn_weights;

typedef struct gn_outputs {          /* device buffers owned by the caller; all required unless noted */
  int32_t *sa_inds[4];               /* (B,np_l) np = 2048,1024,512,256 */
  float *sa_xyz[4];                  /* (B,np_l,3) */
  float *sa_features[4];             /* (B,np_l,C_l) position-major, C = 128,256,256,256 */
  float *fp2_features;               /* (B,1024,256) */
  float *approach;                   /* (B,1024,302) = [objectness(2) | view_score(300)] */
  int64_t *top_view_inds;            /* (B,1024) */
  float *top_view_score;             /* (B,1024) */
  float *top_view_xyz;               /* (B,1024,3) */
  float *top_view_rot;               /* (B,1024,9) */
  float *vp_features;                /* (B,1024*4,256) */
  float *head;                       /* (B,1024,4,48) = [score | angle cls | width | tolerance] x 12 */
  int32_t *crop_uidx, *crop_ucount;  /* optional (B,1024,256) / (B,1024): the de-duplicated crop lists */
  float *grasps;                     /* optional (B,1024,17): NULL skips pred_decode */
  int32_t *counts;                   /* (B), with grasps */
  int32_t fork_streams;              /* 0 (default): every launch on `stream`.  Non-zero: the call may put the grid build and
                                        the deeper levels' geometry (samplers / ball queries 2-4, three_nn) on a side stream of
                                        its own, event-ordered and joined again before it returns.  Worth 0.03-0.06 ms of a
                                        2.8 ms batch inside a captured hipGraph (graspnet_amd.GraphedGraspNet sets it); on
                                        eagerly issued streams the gain depends on which hardware queues the two streams land
                                        on (measured 2.88 ... 3.8 ms against 2.95 unforked), and with several batches in
                                        flight the extra streams only compete for queues (-2 %) */
} gn_outputs;

/* Optional per-launch hook (profiling): called on the host right before (phase 0) and right after
 * (phase 1) each stage is enqueued. */
typedef void (*gn_stage_hook)(const char *stage, int phase, void *user);

size_t gn_graspnet_workspace_bytes(int b, int n);
int gn_graspnet_forward(const gn_weights *weights, const float *cloud /* (B,N,3) */, int b, int n,
                        void *workspace, const gn_outputs *out, gn_stage_hook hook, void *hook_user,
                        gn_stream_t stream);

/* ------------------------------------------------------------------ *
 *  Around the network: front end and post-processing   *
 * ------------------------------------------------------------------ */

/* utils/data_utils.py:18-43 create_point_cloud_from_depth_image for the picked pixels:
 * depth (H,W) u16; pix (npix) flat pixel indices (row*W+col) -> out (npix,3) f32
 * (float64 arithmetic like numpy, z = depth/factor_depth). */
int gn_depth_to_cloud(const uint16_t *depth, int h, int w, double fx, double fy, double cx, double cy,
                      double factor_depth, const int32_t *pix, int npix, float *out, gn_stream_t stream);

/* The same front end for a BATCH of images with the validity mask and the sub-sampling on the device
 * (dataset/graspnet_dataset.py:118-145, demo.py:449-463): nothing is built on the host.
 * gn_depth_valid_compact: depth (B,H,W) u16, mask (B,H,W) u8 or NULL -> valid_pix (B,H*W) i32: the flat indices of
 *   the pixels with mask != 0, depth > 0 and depth_min < depth/factor_depth < depth_max, in row-major order (numpy's
 *   cloud[mask] order), count (B) their number.  workspace: gn_frontend_workspace_bytes(B,H,W).
 * gn_depth_sample_cloud: out (B,n,3) f32 = the reference's cloud_masked[idxs] with idxs = `positions` (B,n) into the
 *   valid list (the caller's own np.random.choice draw: numpy's stream cannot be reproduced on a device), or, with
 *   positions NULL, drawn from `random_words` (B,n) u32: all valid pixels first when there are fewer than n (as the
 *   reference), then floor(word * count / 2^32) (with replacement).  out_pix (B,n) i32 (may be NULL): the pixel taken.
 *   An image without a valid pixel gives zeros / -1. */
size_t gn_frontend_workspace_bytes(int b, int h, int w);
int gn_depth_valid_compact(const uint16_t *depth, const uint8_t *mask, int b, int h, int w, double factor_depth,
                           double depth_min, double depth_max, void *workspace, int32_t *valid_pix, int32_t *count,
                           gn_stream_t stream);
int gn_depth_sample_cloud(const uint16_t *depth, int b, int h, int w, double fx, double fy, double cx, double cy,
                          double factor_depth, const int32_t *valid_pix, const int32_t *count, const int32_t *positions,
                          const uint32_t *random_words, int n, float *out, int32_t *out_pix, gn_stream_t stream);

/* cloud_masked itself (dataset/graspnet_dataset.py:133 / demo.py:455 `cloud[mask]`): out (B,nmax,3) f32, row i = the cloud point
 * of valid pixel i for i < min(count[b], nmax), zero rows behind; out_count (B) = min(count, nmax) (may be NULL).  This is the
 * scene the reference hands to ModelFreeCollisionDetector (test.py:92-96 return_raw_cloud, demo.py:481) -- NOT the num_point
 * points sampled for the network: feed it to gn_grasp_postprocess with n_valid = out_count. */
int gn_depth_masked_cloud(const uint16_t *depth, int b, int h, int w, double fx, double fy, double cx, double cy,
                          double factor_depth, const int32_t *valid_pix, const int32_t *count, int nmax, float *out,
                          int32_t *out_count, gn_stream_t stream);
/* The draw of dataset/graspnet_dataset.py:139-145 / demo.py:458-463 on the device: positions (B,n) into the valid list.
 * count[b] >= n: n DISTINCT positions (np.random.choice(..., replace=False)) = the first n values of a keyed pseudo-random
 * permutation of [0, count) (4-round Feistel network, cycle-walked; csrc/frontend.hip); count[b] < n: arange(count) followed
 * by n - count draws with replacement, as the reference does.  keys (B,4) u32: any words (e.g. hashes of a seed and the
 * image number); equal keys give equal draws.  numpy's own stream cannot be reproduced -- parity is on the distribution
 * (distinctness, range, arange prefix) and on the integer restatement in oracle/post_oracle.py. */
int gn_depth_draw_positions(const int32_t *count, const uint32_t *keys, int b, int n, int32_t *positions, gn_stream_t stream);

/* open3d voxel_down_sample as used by utils/collision_detector.py:32-35 (un-vendored; restated:
 * origin = min_bound - voxel/2, voxel index = floor((p-origin)/voxel), output = mean of members in
 * float64), for a whole batch of clouds in one call, no host read, DETERMINISTIC: the voxels of a cloud
 * come out in order of first appearance (ascending lowest member index) and every mean adds its members
 * in ascending index order, so two runs of one input are bit-identical (and equal to a sequential loop).
 *   points (B,n,3) f32; n_valid (B) device ints (NULL = n everywhere): points past it are padding;
 *   non-finite points belong to no voxel.
 *   out (B,n,3) f64: rows [0, out_counts[b]) are the voxel means, the rest zero; out_counts (B);
 *   status (B): 0 ok, 1 = the cloud spans more than 2^21 voxels on an axis, 2 = table overflow
 *   (the offending points are left out). */
size_t gn_voxel_batched_workspace_bytes(int b, int n);
int gn_voxel_down_sample_batched(const float *points, const int32_t *n_valid, int b, int n, double voxel_size,
                                 void *workspace /* 256-byte aligned */, double *out, int32_t *out_counts,
                                 int32_t *status, gn_stream_t stream);
/* one cloud: out (n,3) f64, out_count one device int; the first int of `workspace` is the status word. */
size_t gn_voxel_workspace_bytes(int n);
int gn_voxel_down_sample(const float *points, int n, double voxel_size, void *workspace, double *out,
                         int32_t *out_count, gn_stream_t stream);

/* utils/collision_detector.py:69-129 ModelFreeCollisionDetector.detect for padded grasp lists:
 * scene_points (np,3) f64 -- ONE scene for the whole batch, as the reference's detector object holds it
 * (np_dev, if non-NULL, is one device int that bounds it: e.g. gn_voxel_down_sample's out_count) --
 * grasps (B,mpad,17) f32 rows as written by gn_pred_decode, counts (B) valid rows (NULL = all mpad).
 * collision_mask / empty_mask (B,mpad) u8, ious (B,mpad,5) f32 = [global,left,right,bottom,shifting];
 * empty_mask and ious may be NULL. */
int gn_collision_detect(const double *scene_points, const int32_t *np_dev, int np, const float *grasps,
                        const int32_t *counts, int b, int mpad, float approach_dist, double collision_thresh,
                        double empty_thresh, double voxel_size, float finger_width, float finger_length,
                        uint8_t *collision_mask, uint8_t *empty_mask, float *ious, gn_stream_t stream);
/* the batch form of test.py:88-96 (one detector per scene there): scene b's points are rows
 * [0, np_dev[b]) of scene_points (B,np_max,3) f64 -- exactly what gn_voxel_down_sample_batched leaves --
 * np_dev (B) device ints (NULL = np_max everywhere).  One launch for the batch, no host read. */
int gn_collision_detect_batched(const double *scene_points, int np_max, const int32_t *np_dev, const float *grasps,
                                const int32_t *counts, int b, int mpad, float approach_dist, double collision_thresh,
                                double empty_thresh, double voxel_size, float finger_width, float finger_length,
                                uint8_t *collision_mask, uint8_t *empty_mask, float *ious, gn_stream_t stream);

/* graspnetAPI GraspGroup.sort_by_score() and .nms() (grasp_nms.nms_grasp; un-vendored, parity unpinned):
 * rows with drop_mask != 0 (e.g. collisions; may be NULL) are removed, the rest sorted by score
 * descending (stable) and, if do_nms, greedily suppressed when within translation_thresh (metres) AND
 * rotation_thresh (radians, geodesic angle) of a kept grasp.  out (B,mpad,17) padded, out_counts (B).
 * mpad <= 2048. */
size_t gn_grasp_nms_workspace_bytes(int b);
int gn_grasp_nms(const float *grasps, const int32_t *counts, const uint8_t *drop_mask, int b, int mpad,
                 double translation_thresh, double rotation_thresh, int do_nms, void *workspace, float *out,
                 int32_t *out_counts, gn_stream_t stream);

/* The three steps above for a whole batch in ONE call -- what test.py:88-96 / demo_pointcloud.py:228-249 run on the host
 * per scene after pred_decode: clouds (B,n,3) f32 (n_valid (B) or NULL) are voxel-down-sampled, scene b's padded grasp rows
 * (B,mpad,17) + counts (B) are checked against scene b's points (finger geometry of collision_detector.py:29-30), the
 * collisions dropped and the rest sorted by score (+ greedy NMS).  out_rows (B,mpad,17) padded with zero rows, out_counts (B);
 * collision_mask (B,mpad) u8 and voxel_status (B) may be NULL.  ~13 launches on `stream`, no host read. */
size_t gn_grasp_postprocess_workspace_bytes(int b, int n, int mpad);
int gn_grasp_postprocess(const float *clouds, const int32_t *n_valid, int b, int n, const float *grasps,
                         const int32_t *counts, int mpad, double voxel_size, float approach_dist,
                         double collision_thresh, double translation_thresh, double rotation_thresh, int do_nms,
                         void *workspace /* 256-byte aligned */, float *out_rows, int32_t *out_counts,
                         uint8_t *collision_mask, int32_t *voxel_status, gn_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GRASPNET_HIP_H */
