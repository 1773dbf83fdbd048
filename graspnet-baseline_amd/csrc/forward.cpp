// forward.cpp -- gn_graspnet_forward: the whole inference path sequenced on the host in one call.
//
// Mirrors GraspNet.forward + pred_decode of the reference (models/graspnet.py:70-134; backbone.py:105-129;
// modules.py:41-77,109-141,169-232) as a fixed sequence of this library's own stage launches.  No device
// code here: every stage is one of the C-ABI entry points, called with the caller's stream.
#include <stdlib.h>

#include <map>
#include <mutex>
#include <utility>

#include "gn_common.h"

namespace gn {
namespace {

// ---- fork / join inside one forward ---------------------------------------------------------------------------------
// The forward is a chain of ~25 launches, but not all of it is ONE chain: the grid build needs only the cloud, and the
// geometry of the deeper levels (FPS 2-4, ball queries 2-4, both three_nn) needs only sampled coordinates, never features.
// Issued on one stream they sit in front of the MLP kernels that do not depend on them (0.05 ms of grid build in front of
// the SA1 sampler, ~0.2 ms of small-level geometry in front of SA2-4 / FP).  With a side stream per calling stream the
// geometry runs beside SA1's ball query + MLP, event-ordered (hipEventRecord / hipStreamWaitEvent only: no host sync, no
// allocation after the first call on a stream, and legal under hipGraph capture -- the side stream is pulled into the
// capture by its first wait and joins back before the call returns).  OPT-IN (gn_outputs.fork_streams, GN_FORWARD_FORK=1):
// inside a captured graph the edges are static and the gain is a steady 0.03-0.06 ms per batch; between eagerly issued
// streams it depends on the hardware queues (and command-processor pipes) the two streams happen to land on -- 2.88 ms per
// batch of 8 in the best mapping, 3.45-3.8 ms in others, against 2.95 ms unforked -- and with several batches in flight on
// several streams the extra streams only compete for queues (-2 %).  A call with a per-stage hook (profiling brackets on
// the caller's stream) never forks.
struct ForkCtx {
  hipStream_t side = nullptr;
  hipEvent_t ev[8] = {};
};
constexpr int kForkEvents = 8;

ForkCtx *fork_ctx(hipStream_t main) {
  static std::mutex mu;
  static std::map<std::pair<int, hipStream_t>, ForkCtx *> table;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  auto key = std::make_pair(dev, main);
  auto it = table.find(key);
  if (it != table.end()) return it->second;
  // first call on this stream: it may already be inside a stream capture whose mode forbids resource creation
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  ForkCtx *c = new ForkCtx();
  bool ok = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) == hipSuccess;   // (a high-priority stream: 3.8 ms)
  for (int i = 0; ok && i < kForkEvents; ++i) ok = hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming) == hipSuccess;
  (void)hipThreadExchangeStreamCaptureMode(&mode);
  if (!ok) {
    (void)hipGetLastError();
    delete c;
    c = nullptr;               // remembered: this stream runs unforked
  }
  table[key] = c;
  return c;
}

// the caller's choice (gn_outputs.fork_streams) unless GN_FORWARD_FORK=0 / 1 overrides it
bool fork_wanted(bool asked) {
  const char *e = getenv("GN_FORWARD_FORK");
  if (e && e[0] == '0') return false;
  if (e && e[0] == '1') return true;
  return asked;
}

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
  GN_REQUIRE(wt->num_view >= 0 && wt->num_view <= 318, "graspnet_forward: num_view must be 0..318 (0 = the default 300), got %d", wt->num_view);
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

  static const char *const kFps[4] = {"furthest_point_sampling[1]", "furthest_point_sampling[2]",
                                      "furthest_point_sampling[3]", "furthest_point_sampling[4]"};
  static const char *const kBall[4] = {"ball_query[1]", "ball_query[2]", "ball_query[3]", "ball_query[4]"};
  static const char *const kLin[4] = {"", "point_linear[2]", "point_linear[3]", "point_linear[4]"};
  static const char *const kSaName[4] = {"sa_group_mlp_pool[1]", "sa_hoisted_mlp_pool[2]", "sa_hoisted_mlp_pool[3]",
                                         "sa_hoisted_mlp_pool[4]"};
  const int lv_pts[4] = {n, kSa[0].npoint, kSa[1].npoint, kSa[2].npoint};      // points a level samples from
  const float *const lv_xyz_of[4] = {cloud, o->sa_xyz[0], o->sa_xyz[1], o->sa_xyz[2]};
  const float *const lv_feat_of[4] = {nullptr, o->sa_features[0], o->sa_features[1], o->sa_features[2]};

  // geometry of level l (coordinates only): the sampler and the ball query (backbone.py:105-121, pointnet2_modules.py:236-249).
  // SA1's sampler records its first exact-tie step; deeper levels (FPS-ordered input) use it to return the identity
  // prefix at once when no tie can have changed the order
  auto sample = [&](int l, gn_stream_t s) -> int {
    const SaCfg &c = kSa[l];
    GN_STAGE(kFps[l], gn_furthest_point_sampling(lv_xyz_of[l], b, lv_pts[l], c.npoint, w.fps[l], o->sa_inds[l], o->sa_xyz[l],
                                                 l == 0 ? w.tie_flag : nullptr, l > 0 ? w.tie_flag : nullptr, s));
    return GN_OK;
  };
  auto query = [&](int l, gn_stream_t s) -> int {
    const SaCfg &c = kSa[l];
    if (l == 0 && w.grid)
      GN_STAGE("ball_query_grid[1]",
               gn_ball_query_grid(o->sa_xyz[0], w.grid, b, lv_pts[0], c.npoint, c.radius, c.nsample, w.ball_idx[0], s));
    else
      GN_STAGE(kBall[l], gn_ball_query(o->sa_xyz[l], lv_xyz_of[l], b, lv_pts[l], c.npoint, c.radius, c.nsample, w.ball_idx[l], s));
    return GN_OK;
  };
  // features of level l: (per-point layer-1 part,) grouped MLP + max-pool
  auto features = [&](int l, gn_stream_t s) -> int {
    const SaCfg &c = kSa[l];
    const float *const wp[3] = {wt->sa[l].w[0], wt->sa[l].w[1], wt->sa[l].w[2]};
    const float *const bp[3] = {wt->sa[l].b[0], wt->sa[l].b[1], wt->sa[l].b[2]};
    const void *const wps[3] = {wt->sa_split[l].w[0], wt->sa_split[l].w[1], wt->sa_split[l].w[2]};
    const float *const bps[3] = {wt->sa_split[l].b[0], wt->sa_split[l].b[1], wt->sa_split[l].b[2]};
    if (c.cin == 0) {
      if (wt->use_split)
        GN_STAGE(kSaName[l], gn_sa_group_mlp_pool_split(lv_xyz_of[l], o->sa_xyz[l], w.ball_idx[l], b, lv_pts[l], c.npoint,
                                                        c.nsample, c.radius, c.mlp, wps, bps, o->sa_features[l], s));
      else
        GN_STAGE(kSaName[l], gn_sa_group_mlp_pool(lv_xyz_of[l], nullptr, o->sa_xyz[l], w.ball_idx[l], b, lv_pts[l], c.npoint,
                                                  c.nsample, 0, c.radius, c.mlp, wp, bp, o->sa_features[l], s));
    } else {
      if (wt->use_split)
        GN_STAGE(kLin[l], gn_point_linear_split(lv_feat_of[l], b * lv_pts[l], c.cin, c.mlp[0], wt->sa_pre_split[l].w[0],
                                                wt->sa_pre_split[l].b[0], w.pfeat[l], s));
      else
        GN_STAGE(kLin[l], gn_point_linear(lv_feat_of[l], b * lv_pts[l], c.cin, c.mlp[0], wt->sa_pre[l].w[0], wt->sa_pre[l].b[0],
                                          w.pfeat[l], s));
      if (wt->use_split)
        GN_STAGE(kSaName[l], gn_sa_hoisted_mlp_pool_split(lv_xyz_of[l], w.pfeat[l], o->sa_xyz[l], w.ball_idx[l], b, lv_pts[l],
                                                          c.npoint, c.nsample, c.radius, c.mlp, wps, bps, o->sa_features[l], s));
      else
        GN_STAGE(kSaName[l], gn_sa_hoisted_mlp_pool(lv_xyz_of[l], w.pfeat[l], o->sa_xyz[l], w.ball_idx[l], b, lv_pts[l], c.npoint,
                                                    c.nsample, c.radius, c.mlp, wp, bp, o->sa_features[l], s));
    }
    return GN_OK;
  };
  auto three_nn = [&](int i, gn_stream_t s) -> int {       // i = 0: fp1 (sa3 <- sa4), 1: fp2 (sa2 <- sa3)
    if (i == 0)
      GN_STAGE("three_nn[fp1]", gn_three_nn(o->sa_xyz[2], o->sa_xyz[3], b, kSa[2].npoint, kSa[3].npoint, w.d2[0], w.i3[0], s));
    else
      GN_STAGE("three_nn[fp2]", gn_three_nn(o->sa_xyz[1], o->sa_xyz[2], b, kSa[1].npoint, kSa[2].npoint, w.d2[1], w.i3[1], s));
    return GN_OK;
  };
#define GN_TRY(call)                 \
  do {                               \
    const int rc__ = (call);         \
    if (rc__ != GN_OK) return rc__;  \
  } while (0)
#define GN_HIP(call, what)                                                                      \
  do {                                                                                          \
    const hipError_t e__ = (call);                                                              \
    if (e__ != hipSuccess) {                                                                    \
      set_error("graspnet_forward: %s failed: %s", what, hipGetErrorString(e__));               \
      return GN_ERR_LAUNCH;                                                                     \
    }                                                                                           \
  } while (0)

  ForkCtx *fk = (!hook && fork_wanted(o->fork_streams != 0)) ? fork_ctx(as_stream(st)) : nullptr;
  if (fk) {
    hipStream_t main_s = as_stream(st), side_s = fk->side;
    gn_stream_t side = reinterpret_cast<gn_stream_t>(side_s);
    bool side_open = false;      // the side stream holds work of this call that the caller's stream has not joined yet
    auto forked = [&]() -> int {
    // side: grid build (needs the cloud only) beside the SA1 sampler
    GN_HIP(hipEventRecord(fk->ev[0], main_s), "event record");
    GN_HIP(hipStreamWaitEvent(side_s, fk->ev[0], 0), "stream wait");
    side_open = true;
    if (w.grid) {
      GN_TRY(gn_grid_build(cloud, b, n, kSa[0].radius, w.grid, side));
      GN_HIP(hipEventRecord(fk->ev[1], side_s), "event record");
    }
    GN_TRY(sample(0, st));
    GN_HIP(hipEventRecord(fk->ev[2], main_s), "event record");
    // side: the deeper levels' geometry behind SA1's centres -- samplers, ball queries, three_nn
    GN_HIP(hipStreamWaitEvent(side_s, fk->ev[2], 0), "stream wait");
    for (int l = 1; l < 4; ++l) {
      GN_TRY(sample(l, side));
      GN_TRY(query(l, side));
      GN_HIP(hipEventRecord(fk->ev[2 + l], side_s), "event record");
    }
    GN_TRY(three_nn(0, side));
    GN_TRY(three_nn(1, side));
    GN_HIP(hipEventRecord(fk->ev[6], side_s), "event record");
    // main: SA1's query + MLP meanwhile, then each level as its geometry arrives
    if (w.grid) GN_HIP(hipStreamWaitEvent(main_s, fk->ev[1], 0), "stream wait");
    GN_TRY(query(0, st));
    GN_TRY(features(0, st));
    for (int l = 1; l < 4; ++l) {
      GN_HIP(hipStreamWaitEvent(main_s, fk->ev[2 + l], 0), "stream wait");
      GN_TRY(features(l, st));
    }
    GN_HIP(hipStreamWaitEvent(main_s, fk->ev[6], 0), "stream wait");       // the join: nothing is left on the side stream
    side_open = false;
    return GN_OK;
    };
    const int frc = forked();
    if (frc != GN_OK) {
      // a failed stage must not leave the side stream running (it writes o->sa_* and the workspace) nor, under hipGraph
      // capture, an unjoined fork: best-effort join before the error goes back to the caller
      if (side_open && hipEventRecord(fk->ev[7], side_s) == hipSuccess) (void)hipStreamWaitEvent(main_s, fk->ev[7], 0);
      (void)hipGetLastError();
      return frc;
    }
  } else {
    // one grid over the input cloud serves SA1's ball query and CloudCrop's cylinder query
    if (w.grid) GN_STAGE("grid_build", gn_grid_build(cloud, b, n, kSa[0].radius, w.grid, st));
    // ---- 4 set-abstraction levels (backbone.py:105-121)
    for (int l = 0; l < 4; ++l) {
      GN_TRY(sample(l, st));
      GN_TRY(query(l, st));
      GN_TRY(features(l, st));
    }
    GN_TRY(three_nn(0, st));
    GN_TRY(three_nn(1, st));
  }

  // ---- 2 feature-propagation levels (backbone.py:124-125)
  const int fp_mlp[2] = {256, 256};
  {
    const float *const wp[2] = {wt->fp[0].w[0], wt->fp[0].w[1]};
    const float *const bp[2] = {wt->fp[0].b[0], wt->fp[0].b[1]};
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
  const int views = wt->num_view ? wt->num_view : kViews;
  const int M = kSeeds, nv = 2 + views;

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
    GN_STAGE("approach_post", gn_approach_post(o->approach, nv, b * M, views, wt->views, wt->view_rot, o->top_view_inds,
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
