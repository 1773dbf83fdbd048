// post.hip -- what every caller of the reference runs right before / after the network
// (SURVEY.md 8f "next" rows), on the device:
//
//   depth_to_cloud     utils/data_utils.py:18-43 create_point_cloud_from_depth_image (+ pixel pick)
//   voxel_down_sample  open3d PointCloud.voxel_down_sample as used by utils/collision_detector.py:32-35
//                      (un-vendored third party: restated from its documented behaviour -- voxel origin
//                      = min_bound - voxel/2, index = floor((p-origin)/voxel), output = mean of members;
//                      PARITY UNPINNED, output order is unspecified there and here)
//   collision_detect   utils/collision_detector.py:69-129 ModelFreeCollisionDetector.detect
//   grasp_nms          graspnetAPI GraspGroup.sort_by_score() / .nms() (grasp_nms.nms_grasp; un-vendored,
//                      PARITY UNPINNED): score-descending greedy suppression of grasps within a
//                      translation AND rotation-angle threshold of a kept one
//
// The reference does these in numpy / open3d on the host after a device->host copy of the grasps
// (test.py:88-96); its README calls the collision check the slow step (README.md:72).
#include "gn_common.h"

namespace gn {

// ------------------------------------------------------------------------------------------------
// depth -> cloud: float64 math like numpy (uint16 / float -> float64), stored as float32
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void depth_to_cloud_kernel(const uint16_t *__restrict__ depth, int w, double fx,
                                                             double fy, double cx, double cy, double scale,
                                                             const int32_t *__restrict__ pix, int npix,
                                                             float *__restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= npix) return;
  const int p = pix[i];
  const int row = p / w, col = p - row * w;
  const double z = (double)depth[p] / scale;            // data_utils.py:37
  const double x = ((double)col - cx) * z / fx;         // :38
  const double y = ((double)row - cy) * z / fy;         // :39
  out[i * 3 + 0] = (float)x;
  out[i * 3 + 1] = (float)y;
  out[i * 3 + 2] = (float)z;
}

// ------------------------------------------------------------------------------------------------
// voxel down-sample: hash grid with 64-bit keys, double sums
// workspace: [0] int counter, [1] int error flag, [2..7] double min bound (3) ... then the table
// ------------------------------------------------------------------------------------------------
struct VoxelHeader {
  int count;
  int error;
  double minb[3];
};

__global__ __launch_bounds__(1024) void voxel_minbound_kernel(const float *__restrict__ pts, int n,
                                                              VoxelHeader *__restrict__ hdr) {
  __shared__ float s[3][16];
  float m[3] = {INFINITY, INFINITY, INFINITY};
  for (int i = threadIdx.x; i < n; i += 1024)
    for (int a = 0; a < 3; ++a) m[a] = fminf(m[a], pts[i * 3 + a]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int a = 0; a < 3; ++a) {
    const float v = -wave_max(-m[a]);
    if (lane == 0) s[a][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = s[threadIdx.x][0];
    for (int wv = 1; wv < 16; ++wv) v = fminf(v, s[threadIdx.x][wv]);
    hdr->minb[threadIdx.x] = (double)v;
    if (threadIdx.x == 0) { hdr->count = 0; hdr->error = 0; }
  }
}

__device__ __forceinline__ unsigned long long hash64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}

constexpr unsigned long long VOXEL_EMPTY = ~0ULL;

__global__ __launch_bounds__(256) void voxel_insert_kernel(const float *__restrict__ pts, int n, double voxel,
                                                           VoxelHeader *__restrict__ hdr,
                                                           unsigned long long *__restrict__ keys,
                                                           double *__restrict__ sums, int *__restrict__ cnts,
                                                           unsigned table_mask) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  unsigned long long key = 0;
  double p[3];
  for (int a = 0; a < 3; ++a) {
    p[a] = (double)pts[i * 3 + a];
    const double origin = hdr->minb[a] - voxel * 0.5;
    const double ref = (p[a] - origin) / voxel;
    const long long vi = (long long)floor(ref);
    if (vi < 0 || vi >= (1LL << 21)) { atomicExch(&hdr->error, 1); return; }
    key |= (unsigned long long)vi << (21 * a);
  }
  unsigned slot = (unsigned)hash64(key) & table_mask;
  for (unsigned probe = 0; probe <= table_mask; ++probe) {
    const unsigned long long prev = atomicCAS(&keys[slot], VOXEL_EMPTY, key);
    if (prev == VOXEL_EMPTY || prev == key) {
      atomicAdd(&sums[slot * 3 + 0], p[0]);
      atomicAdd(&sums[slot * 3 + 1], p[1]);
      atomicAdd(&sums[slot * 3 + 2], p[2]);
      atomicAdd(&cnts[slot], 1);
      return;
    }
    slot = (slot + 1) & table_mask;
  }
  atomicExch(&hdr->error, 2);
}

__global__ __launch_bounds__(256) void voxel_emit_kernel(const double *__restrict__ sums, const int *__restrict__ cnts,
                                                         unsigned nslots, VoxelHeader *__restrict__ hdr,
                                                         double *__restrict__ out) {
  const unsigned s = blockIdx.x * 256 + threadIdx.x;
  if (s >= nslots) return;
  const int c = cnts[s];
  if (c > 0) {
    const int o = atomicAdd(&hdr->count, 1);
    out[o * 3 + 0] = sums[s * 3 + 0] / (double)c;
    out[o * 3 + 1] = sums[s * 3 + 1] / (double)c;
    out[o * 3 + 2] = sums[s * 3 + 2] / (double)c;
  }
}

static unsigned voxel_slots(int n) {
  unsigned s = 1024;
  while (s < 2u * (unsigned)n) s <<= 1;
  return s;
}

// ------------------------------------------------------------------------------------------------
// collision detection: one workgroup per grasp, float64 like numpy on open3d's float64 points
// ------------------------------------------------------------------------------------------------
struct CollisionParams {
  float finger_width, finger_length, voxel_size;
  float approach_dist;          // already max(approach_dist, finger_width)  (collision_detector.py:69)
  double collision_thresh, empty_thresh;
};

__global__ __launch_bounds__(256) void collision_kernel(const double *__restrict__ scene, const int *__restrict__ np_ptr,
                                                        int np_fixed, const float *__restrict__ grasps, int ldg,
                                                        const int32_t *__restrict__ counts, int mpad,
                                                        CollisionParams P, uint8_t *__restrict__ collision_mask,
                                                        uint8_t *__restrict__ empty_mask, float *__restrict__ ious) {
  __shared__ int s_cnt[6];
  const int g = blockIdx.x;             // grasp row within the scene's padded list
  const int b = blockIdx.y;
  const int m_valid = counts ? counts[b] : mpad;
  const size_t orow = (size_t)b * mpad + g;
  if (g >= m_valid) {                   // padding row
    if (threadIdx.x == 0) {
      collision_mask[orow] = 0;
      if (empty_mask) empty_mask[orow] = 0;
    }
    return;
  }
  if (threadIdx.x < 6) s_cnt[threadIdx.x] = 0;
  __syncthreads();
  const int np = np_ptr ? *np_ptr : np_fixed;
  const float *gr = grasps + orow * ldg;
  const float width = gr[1], height = gr[2], depth = gr[3];
  double R[9], T[3];
  for (int i = 0; i < 9; ++i) R[i] = (double)gr[4 + i];
  for (int i = 0; i < 3; ++i) T[i] = (double)gr[13 + i];
  // thresholds are float32 expressions in numpy (float32 arrays with python scalars), :84-99
  const double zlo = (double)(-height / 2.0f), zhi = (double)(height / 2.0f);
  const double x_fl = (double)(depth - P.finger_length);
  const double x_d = (double)depth;
  const double y3 = (double)(-(width / 2.0f + P.finger_width));
  const double y4 = (double)(-width / 2.0f);
  const double y5 = (double)(width / 2.0f + P.finger_width);
  const double y6 = (double)(width / 2.0f);
  const double x_b = (double)(depth - P.finger_length - P.finger_width);
  const double x_s = (double)(depth - P.finger_length - P.finger_width - P.approach_dist);
  int c_left = 0, c_right = 0, c_bottom = 0, c_shift = 0, c_global = 0, c_inner = 0;
  for (int i = threadIdx.x; i < np; i += 256) {
    const double dx = scene[i * 3 + 0] - T[0], dy = scene[i * 3 + 1] - T[1], dz = scene[i * 3 + 2] - T[2];
    // targets = (scene - T) @ R  (:80-81)
    const double tx = dx * R[0] + dy * R[3] + dz * R[6];
    const double ty = dx * R[1] + dy * R[4] + dz * R[7];
    const double tz = dx * R[2] + dy * R[5] + dz * R[8];
    const bool m1 = tz > zlo && tz < zhi;
    const bool m2 = tx > x_fl && tx < x_d;
    const bool m3 = ty > y3, m4 = ty < y4, m5 = ty < y5, m6 = ty > y6;
    const bool m7 = tx <= x_fl && tx > x_b;
    const bool m8 = tx <= x_b && tx > x_s;
    const bool left = m1 && m2 && m3 && m4;
    const bool right = m1 && m2 && m5 && m6;
    const bool bottom = m1 && m3 && m5 && m7;
    const bool shift = m1 && m3 && m5 && m8;
    c_left += left; c_right += right; c_bottom += bottom; c_shift += shift;
    c_global += (left || right || bottom || shift);
    c_inner += (m1 && m2 && !m4 && !m6);
  }
  atomicAdd(&s_cnt[0], c_global); atomicAdd(&s_cnt[1], c_left); atomicAdd(&s_cnt[2], c_right);
  atomicAdd(&s_cnt[3], c_bottom); atomicAdd(&s_cnt[4], c_shift); atomicAdd(&s_cnt[5], c_inner);
  __syncthreads();
  if (threadIdx.x == 0) {
    const float vs3 = (float)((double)P.voxel_size * (double)P.voxel_size * (double)P.voxel_size);  // python float **3
    // float32 volume expressions, evaluated left to right (:113-116)
    const float lr_vol = height * P.finger_length * P.finger_width / vs3;
    const float bottom_vol = height * (width + 2.0f * P.finger_width) * P.finger_width / vs3;
    const float shift_vol = height * (width + 2.0f * P.finger_width) * P.approach_dist / vs3;
    const float volume = lr_vol * 2.0f + bottom_vol + shift_vol;
    const double giou = (double)s_cnt[0] / (double)(volume + 1e-6f);     // int64 / float32 -> float64 (:119)
    collision_mask[orow] = giou > P.collision_thresh;
    if (empty_mask) {
      const float inner_vol = height * P.finger_length * width / vs3;
      empty_mask[orow] = ((double)s_cnt[5] / (double)inner_vol) < P.empty_thresh;
    }
    if (ious) {
      float *o = ious + orow * 5;
      o[0] = (float)giou;
      o[1] = (float)((double)s_cnt[1] / (double)(lr_vol + 1e-6f));
      o[2] = (float)((double)s_cnt[2] / (double)(lr_vol + 1e-6f));
      o[3] = (float)((double)s_cnt[3] / (double)(bottom_vol + 1e-6f));
      o[4] = (float)((double)s_cnt[4] / (double)(shift_vol + 1e-6f));
    }
  }
}

// ------------------------------------------------------------------------------------------------
// sort by score + greedy NMS, one workgroup per scene (M <= 2048 rows of 17 floats)
// ------------------------------------------------------------------------------------------------
constexpr int NMS_MAX = 2048;

__global__ __launch_bounds__(1024) void grasp_nms_kernel(const float *__restrict__ grasps,
                                                         const int32_t *__restrict__ counts,
                                                         const uint8_t *__restrict__ drop, int mpad, double tt2,
                                                         double cos_rt, int do_nms,
                                                         unsigned long long *__restrict__ supp /* (B, NMS_MAX, NMS_MAX/64) */,
                                                         float *__restrict__ out, int32_t *__restrict__ out_counts) {
  __shared__ unsigned long long s_key[NMS_MAX];   // (score descending, index ascending) packed
  __shared__ unsigned long long s_removed[NMS_MAX / 64];
  __shared__ int s_n;
  const int b = blockIdx.x, tid = threadIdx.x;
  const int m = counts ? counts[b] : mpad;
  grasps += (size_t)b * mpad * 17;
  out += (size_t)b * mpad * 17;
  supp += (size_t)b * NMS_MAX * (NMS_MAX / 64);
  // ---- keys: float score -> order-preserving uint, inverted for descending; rows flagged in `drop`
  //      (e.g. collisions) and padding sort to the end
  for (int i = tid; i < NMS_MAX; i += 1024) {
    unsigned long long key = ~0ULL;
    if (i < m && !(drop && drop[(size_t)b * mpad + i])) {
      unsigned u = __float_as_uint(grasps[i * 17]);
      u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // ascending float order
      key = ((unsigned long long)(~u) << 32) | (unsigned)i; // descending score, then ascending index
    }
    s_key[i] = key;
  }
  if (tid == 0) s_n = 0;
  __syncthreads();
  // ---- bitonic sort of NMS_MAX keys
  for (int k = 2; k <= NMS_MAX; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < NMS_MAX; i += 1024) {
        const int l = i ^ j;
        if (l > i) {
          const unsigned long long a = s_key[i], c = s_key[l];
          const bool up = (i & k) == 0;
          if ((a > c) == up) { s_key[i] = c; s_key[l] = a; }
        }
      }
      __syncthreads();
    }
  // number of live rows
  for (int i = tid; i < NMS_MAX; i += 1024)
    if (s_key[i] != ~0ULL && (i + 1 == NMS_MAX || s_key[i + 1] == ~0ULL)) s_n = i + 1;
  __syncthreads();
  const int n = s_n;
  const int words = (n + 63) >> 6;
  if (do_nms) {
    // ---- suppression matrix: bit j of row i  <=>  j > i (in sorted order) and j is within both thresholds of i
    for (int e = tid; e < n * words; e += 1024) {
      const int i = e / words, wj = e - i * words;
      const float *gi = grasps + (size_t)(unsigned)(s_key[i] & 0xffffffffu) * 17;
      unsigned long long bits = 0;
      for (int t = 0; t < 64; ++t) {
        const int j = wj * 64 + t;
        if (j <= i || j >= n) continue;
        const float *gj = grasps + (size_t)(unsigned)(s_key[j] & 0xffffffffu) * 17;
        const double dx = (double)gi[13] - (double)gj[13], dy = (double)gi[14] - (double)gj[14],
                     dz = (double)gi[15] - (double)gj[15];
        if (dx * dx + dy * dy + dz * dz > tt2) continue;
        double tr = 0.0;                     // trace(Ri^T Rj) = sum of elementwise products
        for (int q = 0; q < 9; ++q) tr += (double)gi[4 + q] * (double)gj[4 + q];
        double cv = (tr - 1.0) * 0.5;
        cv = cv > 1.0 ? 1.0 : (cv < -1.0 ? -1.0 : cv);
        if (cv >= cos_rt) bits |= 1ULL << t;
      }
      supp[(size_t)i * (NMS_MAX / 64) + wj] = bits;
    }
  }
  for (int i = tid; i < NMS_MAX / 64; i += 1024) s_removed[i] = 0;
  __syncthreads();
  // ---- greedy sweep in score order (one wave; lane l owns word l of the removed set, 32 words)
  if (do_nms && tid < 64) {
    unsigned long long removed = 0;
    for (int i = 0; i < n; ++i) {
      const unsigned long long rw = __shfl(removed, i >> 6);
      if (!((rw >> (i & 63)) & 1ULL)) {
        if (tid < words) removed |= supp[(size_t)i * (NMS_MAX / 64) + tid];
      }
    }
    if (tid < NMS_MAX / 64) s_removed[tid] = tid < words ? removed : 0;
  }
  __syncthreads();
  // ---- compact kept rows in sorted order
  for (int i0 = 0; i0 < n; i0 += 1024) {
    const int i = i0 + tid;
    const bool keep = i < n && !((s_removed[i >> 6] >> (i & 63)) & 1ULL);
    // order-preserving position: count kept rows before i
    int pos = 0;
    if (keep) {
      for (int w = 0; w < (i >> 6); ++w) pos += 64 - __popcll(s_removed[w]);
      const unsigned long long below = (i & 63) ? (~s_removed[i >> 6] & ((1ULL << (i & 63)) - 1ULL)) : 0ULL;
      pos += __popcll(below);
      const float *src = grasps + (size_t)(unsigned)(s_key[i] & 0xffffffffu) * 17;
      for (int q = 0; q < 17; ++q) out[(size_t)pos * 17 + q] = src[q];
    }
  }
  __syncthreads();
  if (tid == 0) {
    int kept = 0;
    for (int w = 0; w < words; ++w) {
      const int valid = min(64, n - w * 64);
      const unsigned long long vm = valid == 64 ? ~0ULL : ((1ULL << valid) - 1ULL);
      kept += __popcll(~s_removed[w] & vm);
    }
    out_counts[b] = kept;
    s_n = kept;
  }
  __syncthreads();
  for (int e = s_n * 17 + tid; e < mpad * 17; e += 1024) out[e] = 0.f;
}

}  // namespace gn

using namespace gn;

extern "C" int gn_depth_to_cloud(const uint16_t *depth, int h, int w, double fx, double fy, double cx, double cy,
                                 double factor_depth, const int32_t *pix, int npix, float *out, gn_stream_t stream) {
  GN_REQUIRE(h > 0 && w > 0 && npix >= 0 && factor_depth != 0.0 && fx != 0.0 && fy != 0.0, "depth_to_cloud: bad sizes");
  if (npix == 0) return GN_OK;
  GN_REQUIRE(depth && pix && out, "depth_to_cloud: null pointer");
  hipLaunchKernelGGL(depth_to_cloud_kernel, dim3((npix + 255) / 256), dim3(256), 0, as_stream(stream), depth, w, fx, fy,
                     cx, cy, factor_depth, pix, npix, out);
  GN_CHECK_LAUNCH("depth_to_cloud");
  return GN_OK;
}

extern "C" size_t gn_voxel_workspace_bytes(int n) {
  if (n <= 0) return 0;
  const size_t slots = voxel_slots(n);
  return 64 + slots * (8 + 24 + 4);
}

extern "C" int gn_voxel_down_sample(const float *points, int n, double voxel_size, void *workspace, double *out,
                                    int32_t *out_count, gn_stream_t stream) {
  GN_REQUIRE(n >= 0 && voxel_size > 0.0, "voxel_down_sample: bad sizes");
  GN_REQUIRE(points && workspace && out && out_count, "voxel_down_sample: null pointer");
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    if (hipMemsetAsync(out_count, 0, sizeof(int32_t), s) != hipSuccess) return GN_ERR_LAUNCH;
    return GN_OK;
  }
  const unsigned slots = voxel_slots(n);
  char *ws = (char *)workspace;
  VoxelHeader *hdr = (VoxelHeader *)ws;
  unsigned long long *keys = (unsigned long long *)(ws + 64);
  double *sums = (double *)(ws + 64 + (size_t)slots * 8);
  int *cnts = (int *)(ws + 64 + (size_t)slots * 32);
  if (hipMemsetAsync(keys, 0xff, (size_t)slots * 8, s) != hipSuccess ||
      hipMemsetAsync(sums, 0, (size_t)slots * 28, s) != hipSuccess) {
    set_error("voxel_down_sample: memset failed");
    return GN_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(voxel_minbound_kernel, dim3(1), dim3(1024), 0, s, points, n, hdr);
  hipLaunchKernelGGL(voxel_insert_kernel, dim3((n + 255) / 256), dim3(256), 0, s, points, n, voxel_size, hdr, keys, sums,
                     cnts, slots - 1);
  hipLaunchKernelGGL(voxel_emit_kernel, dim3((slots + 255) / 256), dim3(256), 0, s, sums, cnts, slots, hdr, out);
  if (hipMemcpyAsync(out_count, &hdr->count, sizeof(int32_t), hipMemcpyDeviceToDevice, s) != hipSuccess) {
    set_error("voxel_down_sample: count copy failed");
    return GN_ERR_LAUNCH;
  }
  GN_CHECK_LAUNCH("voxel_down_sample");
  return GN_OK;
}

extern "C" int gn_collision_detect(const double *scene_points, const int32_t *np_dev, int np, const float *grasps,
                                   const int32_t *counts, int b, int mpad, float approach_dist, double collision_thresh,
                                   double empty_thresh, float voxel_size, float finger_width, float finger_length,
                                   uint8_t *collision_mask, uint8_t *empty_mask, float *ious, gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && mpad >= 0 && (np_dev || np >= 0), "collision_detect: bad sizes");
  if (b == 0 || mpad == 0) return GN_OK;
  GN_REQUIRE(scene_points && grasps && collision_mask, "collision_detect: null pointer");
  GN_REQUIRE(b <= 65535, "collision_detect: batch > 65535 unsupported");
  CollisionParams P;
  P.finger_width = finger_width;
  P.finger_length = finger_length;
  P.voxel_size = voxel_size;
  P.approach_dist = approach_dist > finger_width ? approach_dist : finger_width;   // collision_detector.py:69
  P.collision_thresh = collision_thresh;
  P.empty_thresh = empty_thresh;
  hipLaunchKernelGGL(collision_kernel, dim3(mpad, b), dim3(256), 0, as_stream(stream), scene_points, np_dev, np, grasps,
                     17, counts, mpad, P, collision_mask, empty_mask, ious);
  GN_CHECK_LAUNCH("collision_detect");
  return GN_OK;
}

extern "C" size_t gn_grasp_nms_workspace_bytes(int b) {
  return b <= 0 ? 0 : (size_t)b * NMS_MAX * (NMS_MAX / 64) * sizeof(unsigned long long);
}

extern "C" int gn_grasp_nms(const float *grasps, const int32_t *counts, const uint8_t *drop_mask, int b, int mpad,
                            double translation_thresh, double rotation_thresh, int do_nms, void *workspace, float *out,
                            int32_t *out_counts, gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && mpad >= 0 && mpad <= NMS_MAX, "grasp_nms: at most %d grasps per scene", NMS_MAX);
  if (b == 0) return GN_OK;
  GN_REQUIRE(grasps && out && out_counts && (workspace || !do_nms), "grasp_nms: null pointer");
  hipLaunchKernelGGL(grasp_nms_kernel, dim3(b), dim3(1024), 0, as_stream(stream), grasps, counts, drop_mask, mpad,
                     translation_thresh * translation_thresh, cos(rotation_thresh), do_nms,
                     (unsigned long long *)workspace, out, out_counts);
  GN_CHECK_LAUNCH("grasp_nms");
  return GN_OK;
}
