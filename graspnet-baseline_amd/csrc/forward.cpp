// forward.cpp -- gn_graspnet_forward: the whole inference path sequenced on the host in one call.
//
// Mirrors GraspNet.forward + pred_decode of the reference (models/graspnet.py:70-134; backbone.py:105-129;
// modules.py:41-77,109-141,169-232) as a fixed sequence of this library's own stage launches.  No device
// code here: every stage is one of the C-ABI entry points, called with the caller's stream.
#include "gn_common.h"

namespace gn {
namespace {

struct SaCfg { int npoint; float radius; int nsample; int cin; int mlp[3]; };
// models/backbone.py:30-64
constexpr SaCfg kSa[4] = {{2048, 0.04f, 64, 0, {64, 64, 128}},
                          {1024, 0.1f, 32, 128, {128, 128, 256}},
                          {512, 0.2f, 16, 256, {128, 128, 256}},
                          {256, 0.3f, 16, 256, {128, 128, 256}}};
constexpr int kSeeds = 1024, kViews = 300, kAngles = 12, kDepths = 4, kCropSamples = 64;   // graspnet.py:31-45
constexpr int kGridMinPoints = 4096;

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// carve-out of the caller's workspace; the same walk sizes it (base == nullptr)
struct Carver {
  char *base;
  size_t off = 0;
  template <typename T>
  T *take(size_t count) {
    T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
    off = align_up(off + count * sizeof(T));
    return p;
  }
};

struct Workspace {
  int32_t *tie_flag;
  void *grid;
  void *fps[4];
  int32_t *ball_idx[4];
  float *pfeat[4];
  float *d2[2];
  int32_t *i3[2];
  float *fp1;
  int32_t *uidx, *ucount;
  size_t bytes;
};

Workspace carve(void *base, int b, int n) {
  Carver c{static_cast<char *>(base)};
  Workspace w{};
  w.tie_flag = c.take<int32_t>(b);
  w.grid = n >= kGridMinPoints ? c.take<char>(gn_grid_workspace_bytes(b, n)) : nullptr;
  int lv_n = n;
  for (int l = 0; l < 4; ++l) {
    const size_t fb = gn_fps_workspace_bytes(b, lv_n);
    w.fps[l] = fb ? c.take<char>(fb) : nullptr;
    w.ball_idx[l] = c.take<int32_t>((size_t)b * kSa[l].npoint * kSa[l].nsample);
    w.pfeat[l] = kSa[l].cin ? c.take<float>((size_t)b * lv_n * kSa[l].mlp[0]) : nullptr;
    lv_n = kSa[l].npoint;
  }
  const int fp_n[2] = {kSa[2].npoint, kSa[1].npoint};
  for (int i = 0; i < 2; ++i) {
    w.d2[i] = c.take<float>((size_t)b * fp_n[i] * 3);
    w.i3[i] = c.take<int32_t>((size_t)b * fp_n[i] * 3);
  }
  w.fp1 = c.take<float>((size_t)b * kSa[2].npoint * 256);
  w.uidx = c.take<int32_t>((size_t)b * kSeeds * kDepths * kCropSamples);
  w.ucount = c.take<int32_t>((size_t)b * kSeeds);
  w.bytes = c.off;
  return w;
}

}  // namespace
}  // namespace gn

using namespace gn;

extern "C" size_t gn_graspnet_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 0;
  return carve(nullptr, b, n).bytes;
}

#define GN_STAGE(name, call)                 \
  do {                                       \
    if (hook) hook(name, 0, hook_user);      \
    int rc_ = (call);                        \
    if (hook) hook(name, 1, hook_user);      \
    if (rc_ != GN_OK) return rc_;            \
  } while (0)

extern "C" int gn_graspnet_forward(const gn_weights *wt, const float *cloud, int b, int n, void *workspace,
                                   const gn_outputs *o, gn_stage_hook hook, void *hook_user, gn_stream_t st) {
  GN_REQUIRE(b >= 0 && n > 0, "graspnet_forward: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(wt && cloud && workspace && o, "graspnet_forward: null pointer");
  GN_REQUIRE(n >= kSa[0].npoint, "graspnet_forward: the cloud must hold at least %d points", kSa[0].npoint);
  GN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "graspnet_forward: workspace must be 256-byte aligned");
  for (int l = 0; l < 4; ++l)
    GN_REQUIRE(o->sa_inds[l] && o->sa_xyz[l] && o->sa_features[l], "graspnet_forward: null SA output");
  GN_REQUIRE(o->fp2_features && o->approach && o->top_view_inds && o->top_view_score && o->top_view_xyz &&
                 o->top_view_rot && o->vp_features && o->head,
             "graspnet_forward: null output");
  GN_REQUIRE(!o->grasps == !o->counts, "graspnet_forward: grasps and counts go together");
  {   // every layer pointer of the weight set that will be used
    auto full = [](const gn_mlp_weights &m, int nl) {
      for (int i = 0; i < nl; ++i)
        if (!m.w[i] || !m.b[i]) return false;
      return true;
    };
    bool ok = wt->views && wt->view_rot && wt->decode_rot;
    if (wt->use_split) {
      ok = ok && full(wt->crop_split, 3) && full(wt->approach_split, 3) && full(wt->heads_split, 3) &&
           full(wt->fp_split[0], 2) && full(wt->fp_split[1], 2);
      for (int l = 0; l < 4; ++l) ok = ok && full(wt->sa_split[l], 3) && (l == 0 || full(wt->sa_pre_split[l], 1));
    } else {
      ok = ok && full(wt->crop, 3) && full(wt->approach, 3) && full(wt->heads, 3) && full(wt->fp[0], 2) && full(wt->fp[1], 2);
      for (int l = 0; l < 4; ++l) ok = ok && full(wt->sa[l], 3) && (l == 0 || full(wt->sa_pre[l], 1));
    }
    GN_REQUIRE(ok, "graspnet_forward: a weight or bias pointer of the %s set is null", wt->use_split ? "split" : "fp32");
  }
  const Workspace w = carve(workspace, b, n);
  int32_t *uidx = o->crop_uidx ? o->crop_uidx : w.uidx;
  int32_t *ucount = o->crop_ucount ? o->crop_ucount : w.ucount;

  // one grid over the input cloud serves SA1's ball query and CloudCrop's cylinder query
  if (w.grid) GN_STAGE("grid_build", gn_grid_build(cloud, b, n, kSa[0].radius, w.grid, st));

  // ---- 4 set-abstraction levels (backbone.py:105-121)
  static const char *const kFps[4] = {"furthest_point_sampling[1]", "furthest_point_sampling[2]",
                                      "furthest_point_sampling[3]", "furthest_point_sampling[4]"};
  static const char *const kBall[4] = {"ball_query[1]", "ball_query[2]", "ball_query[3]", "ball_query[4]"};
  static const char *const kLin[4] = {"", "point_linear[2]", "point_linear[3]", "point_linear[4]"};
  static const char *const kSaName[4] = {"sa_group_mlp_pool[1]", "sa_hoisted_mlp_pool[2]", "sa_hoisted_mlp_pool[3]",
                                         "sa_hoisted_mlp_pool[4]"};
  const float *lv_xyz = cloud, *lv_feat = nullptr;
  int lv_n = n;
  for (int l = 0; l < 4; ++l) {
    const SaCfg &c = kSa[l];
    // SA1's sampler records its first exact-tie step; deeper levels (FPS-ordered input) use it to return the
    // identity prefix at once when no tie can have changed the order
    GN_STAGE(kFps[l], gn_furthest_point_sampling(lv_xyz, b, lv_n, c.npoint, w.fps[l], o->sa_inds[l], o->sa_xyz[l],
                                                 l == 0 ? w.tie_flag : nullptr, l > 0 ? w.tie_flag : nullptr, st));
    if (l == 0 && w.grid)
      GN_STAGE("ball_query_grid[1]",
               gn_ball_query_grid(o->sa_xyz[0], w.grid, b, lv_n, c.npoint, c.radius, c.nsample, w.ball_idx[0], st));
    else
      GN_STAGE(kBall[l], gn_ball_query(o->sa_xyz[l], lv_xyz, b, lv_n, c.npoint, c.radius, c.nsample, w.ball_idx[l], st));
    const float *const wp[3] = {wt->sa[l].w[0], wt->sa[l].w[1], wt->sa[l].w[2]};
    const float *const bp[3] = {wt->sa[l].b[0], wt->sa[l].b[1], wt->sa[l].b[2]};
    const void *const wps[3] = {wt->sa_split[l].w[0], wt->sa_split[l].w[1], wt->sa_split[l].w[2]};
    const float *const bps[3] = {wt->sa_split[l].b[0], wt->sa_split[l].b[1], wt->sa_split[l].b[2]};
    if (c.cin == 0) {
      if (wt->use_split)
        GN_STAGE(kSaName[l], gn_sa_group_mlp_pool_split(lv_xyz, o->sa_xyz[l], w.ball_idx[l], b, lv_n, c.npoint, c.nsample,
                                                        c.radius, c.mlp, wps, bps, o->sa_features[l], st));
      else
        GN_STAGE(kSaName[l], gn_sa_group_mlp_pool(lv_xyz, nullptr, o->sa_xyz[l], w.ball_idx[l], b, lv_n, c.npoint,
                                                  c.nsample, 0, c.radius, c.mlp, wp, bp, o->sa_features[l], st));
    } else {
      if (wt->use_split)
        GN_STAGE(kLin[l], gn_point_linear_split(lv_feat, b * lv_n, c.cin, c.mlp[0], wt->sa_pre_split[l].w[0],
                                                wt->sa_pre_split[l].b[0], w.pfeat[l], st));
      else
        GN_STAGE(kLin[l], gn_point_linear(lv_feat, b * lv_n, c.cin, c.mlp[0], wt->sa_pre[l].w[0], wt->sa_pre[l].b[0],
                                          w.pfeat[l], st));
      if (wt->use_split)
        GN_STAGE(kSaName[l], gn_sa_hoisted_mlp_pool_split(lv_xyz, w.pfeat[l], o->sa_xyz[l], w.ball_idx[l], b, lv_n, c.npoint,
                                                          c.nsample, c.radius, c.mlp, wps, bps, o->sa_features[l], st));
      else
        GN_STAGE(kSaName[l], gn_sa_hoisted_mlp_pool(lv_xyz, w.pfeat[l], o->sa_xyz[l], w.ball_idx[l], b, lv_n, c.npoint,
                                                    c.nsample, c.radius, c.mlp, wp, bp, o->sa_features[l], st));
    }
    lv_xyz = o->sa_xyz[l];
    lv_feat = o->sa_features[l];
    lv_n = c.npoint;
  }

  // ---- 2 feature-propagation levels (backbone.py:124-125)
  const int fp_mlp[2] = {256, 256};
  {
    const float *const wp[2] = {wt->fp[0].w[0], wt->fp[0].w[1]};
    const float *const bp[2] = {wt->fp[0].b[0], wt->fp[0].b[1]};
    GN_STAGE("three_nn[fp1]", gn_three_nn(o->sa_xyz[2], o->sa_xyz[3], b, kSa[2].npoint, kSa[3].npoint, w.d2[0], w.i3[0], st));
    const void *const wps[2] = {wt->fp_split[0].w[0], wt->fp_split[0].w[1]};
    const float *const bps[2] = {wt->fp_split[0].b[0], wt->fp_split[0].b[1]};
    if (wt->use_split)
      GN_STAGE("fp_interp_mlp[fp1]", gn_fp_interp_mlp_split(o->sa_features[3], o->sa_features[2], w.d2[0], w.i3[0], b,
                                                            kSa[2].npoint, kSa[3].npoint, 256, 256, fp_mlp, wps, bps, w.fp1, st));
    else
      GN_STAGE("fp_interp_mlp[fp1]", gn_fp_interp_mlp(o->sa_features[3], o->sa_features[2], w.d2[0], w.i3[0], b,
                                                      kSa[2].npoint, kSa[3].npoint, 256, 256, fp_mlp, wp, bp, w.fp1, st));
  }
  {
    const float *const wp[2] = {wt->fp[1].w[0], wt->fp[1].w[1]};
    const float *const bp[2] = {wt->fp[1].b[0], wt->fp[1].b[1]};
    GN_STAGE("three_nn[fp2]", gn_three_nn(o->sa_xyz[1], o->sa_xyz[2], b, kSa[1].npoint, kSa[2].npoint, w.d2[1], w.i3[1], st));
    const void *const wps[2] = {wt->fp_split[1].w[0], wt->fp_split[1].w[1]};
    const float *const bps[2] = {wt->fp_split[1].b[0], wt->fp_split[1].b[1]};
    if (wt->use_split)
      GN_STAGE("fp_interp_mlp[fp2]", gn_fp_interp_mlp_split(w.fp1, o->sa_features[1], w.d2[1], w.i3[1], b, kSa[1].npoint,
                                                            kSa[2].npoint, 256, 256, fp_mlp, wps, bps, o->fp2_features, st));
    else
      GN_STAGE("fp_interp_mlp[fp2]", gn_fp_interp_mlp(w.fp1, o->sa_features[1], w.d2[1], w.i3[1], b, kSa[1].npoint,
                                                      kSa[2].npoint, 256, 256, fp_mlp, wp, bp, o->fp2_features, st));
  }
  const float *seed_xyz = o->sa_xyz[1];
  const int M = kSeeds, nv = 2 + kViews;

  // ---- ApproachNet (modules.py:41-77)
  {
    const float *const wp[3] = {wt->approach.w[0], wt->approach.w[1], wt->approach.w[2]};
    const float *const bp[3] = {wt->approach.b[0], wt->approach.b[1], wt->approach.b[2]};
    const void *const wps[3] = {wt->approach_split.w[0], wt->approach_split.w[1], wt->approach_split.w[2]};
    const float *const bps[3] = {wt->approach_split.b[0], wt->approach_split.b[1], wt->approach_split.b[2]};
    if (wt->use_split)
      GN_STAGE("approach_mlp", gn_approach_mlp_split(o->fp2_features, b * M, 256, nv, wps, bps, o->approach, st));
    else
      GN_STAGE("approach_mlp", gn_approach_mlp(o->fp2_features, b * M, 256, nv, wp, bp, o->approach, st));
    GN_STAGE("approach_post", gn_approach_post(o->approach, nv, b * M, kViews, wt->views, wt->view_rot, o->top_view_inds,
                                               o->top_view_score, o->top_view_xyz, o->top_view_rot, st));
  }

  // ---- CloudCrop (modules.py:109-141) on the de-duplicated nested-cylinder lists
  {
    const float *const wp[3] = {wt->crop.w[0], wt->crop.w[1], wt->crop.w[2]};
    const float *const bp[3] = {wt->crop.b[0], wt->crop.b[1], wt->crop.b[2]};
    const int mlp[3] = {64, 128, 256};
    if (w.grid)
      GN_STAGE("cylinder_query_unique_grid",
               gn_cylinder_query_unique_grid(seed_xyz, o->top_view_rot, w.grid, b, n, M, wt->cylinder_radius, wt->hmin,
                                             wt->hmax, kDepths, kCropSamples, uidx, ucount, st));
    else
      GN_STAGE("cylinder_query_unique",
               gn_cylinder_query_unique(seed_xyz, cloud, o->top_view_rot, b, n, M, wt->cylinder_radius, wt->hmin,
                                        wt->hmax, kDepths, kCropSamples, uidx, ucount, st));
    const void *const wps[3] = {wt->crop_split.w[0], wt->crop_split.w[1], wt->crop_split.w[2]};
    const float *const bps[3] = {wt->crop_split.b[0], wt->crop_split.b[1], wt->crop_split.b[2]};
    if (wt->use_split)
      GN_STAGE("crop_unique_mlp_pool", gn_crop_unique_mlp_pool_split(cloud, seed_xyz, o->top_view_rot, uidx, ucount, b, n, M,
                                                                     kDepths, kCropSamples, mlp, wps, bps, o->vp_features, st));
    else
      GN_STAGE("crop_unique_mlp_pool", gn_crop_unique_mlp_pool(cloud, seed_xyz, o->top_view_rot, uidx, ucount, b, n, M, kDepths,
                                                               kCropSamples, mlp, wp, bp, o->vp_features, st));
  }

  // ---- OperationNet + ToleranceNet (modules.py:169-191, 214-232)
  {
    const float *const wp[3] = {wt->heads.w[0], wt->heads.w[1], wt->heads.w[2]};
    const float *const bp[3] = {wt->heads.b[0], wt->heads.b[1], wt->heads.b[2]};
    const void *const wps[3] = {wt->heads_split.w[0], wt->heads_split.w[1], wt->heads_split.w[2]};
    const float *const bps[3] = {wt->heads_split.b[0], wt->heads_split.b[1], wt->heads_split.b[2]};
    if (wt->use_split)
      GN_STAGE("heads_mlp", gn_heads_mlp_split(o->vp_features, b * M * kDepths, 256, 4 * kAngles, wps, bps, o->head, st));
    else
      GN_STAGE("heads_mlp", gn_heads_mlp(o->vp_features, b * M * kDepths, 256, 4 * kAngles, wp, bp, o->head, st));
  }

  // ---- pred_decode (graspnet.py:77-134)
  if (o->grasps)
    GN_STAGE("pred_decode", gn_pred_decode(o->approach, nv, o->head, o->top_view_inds, seed_xyz, wt->decode_rot, b, M,
                                           kDepths, kAngles, o->grasps, o->counts, st));
  return GN_OK;
}
