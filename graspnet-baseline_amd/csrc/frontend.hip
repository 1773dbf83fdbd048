// frontend.hip -- the step in front of the network on the device: a batch of depth images + workspace
// masks in, sampled clouds out, nothing built on the host.
//
//   reference (host, numpy):  dataset/graspnet_dataset.py:118-145 and demo.py:449-463 / demo_pointcloud.py:178-184
//       cloud = create_point_cloud_from_depth_image(depth, camera)          utils/data_utils.py:18-43
//       mask  = workspace_mask & (depth > 0)   [demo.py:453 also: depth_min < z < depth_max]
//       cloud_masked = cloud[mask]                                           row-major order of the valid pixels
//       idxs = np.random.choice(len(cloud_masked), num_point, replace=False)   (>= num_point valid pixels)
//            | concat(arange(len), np.random.choice(len, num_point - len, replace=True))
//       cloud_sampled = cloud_masked[idxs]
//
//   here:  gn_depth_valid_compact   the ordered list of valid pixels per image + its length, on the device
//                                   (ballot / popcount prefix: a wave owns 1024 consecutive pixels; counts -> one scan
//                                   per image -> ordered scatter; no atomics, so the list is in numpy's order)
//          gn_depth_sample_cloud    cloud_sampled from positions into that list -- the caller's own `idxs` (numpy's random
//                                   stream cannot be reproduced; with the reference's idxs the result is the reference's,
//                                   bit for bit) -- or from a stream of random 32-bit words the caller seeds (position =
//                                   all valid pixels first when there are fewer than num_point, as the reference does, then
//                                   floor(word * count / 2^32): sampling with replacement)
// float64 pixel arithmetic like numpy's (uint16 / float -> float64), float32 out.
#include "gn_common.h"

namespace gn {

constexpr int FE_SEG = 1024;      // pixels per wave segment

struct FeValid {
  const uint16_t *depth;
  const uint8_t *mask;
  double factor, dmin, dmax;
  __device__ __forceinline__ bool operator()(size_t img_off, int p, int npix) const {
    if (p >= npix) return false;
    const uint16_t dv = depth[img_off + p];
    if (dv == 0 || (mask && mask[img_off + p] == 0)) return false;
    const double z = (double)dv / factor;           // data_utils.py:37
    return z > dmin && z < dmax;
  }
};

// one wave per segment: how many valid pixels it holds
__global__ __launch_bounds__(256) void fe_count_kernel(FeValid V, int npix, int nseg, int32_t *__restrict__ seg_cnt) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= nseg) return;
  const size_t img = (size_t)b * npix;
  int c = 0;
  for (int j = 0; j < FE_SEG / 64; ++j) c += __popcll(__ballot(V(img, seg * FE_SEG + j * 64 + lane, npix)));
  if (lane == 0) seg_cnt[(size_t)b * (nseg + 1) + seg] = c;
}

// one workgroup per image: exclusive scan of the segment counts in place; [nseg] = the image's total
__global__ __launch_bounds__(1024) void fe_scan_kernel(int nseg, int32_t *__restrict__ seg_cnt, int32_t *__restrict__ count) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  int32_t *c = seg_cnt + (size_t)blockIdx.x * (nseg + 1);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base < nseg; base += 1024) {
    const int i = base + t;
    const int v = i < nseg ? c[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    int woff = s_carry;
    for (int w = 0; w < wave; ++w) woff += s_wave[w];
    if (i < nseg) c[i] = woff + incl - v;
    __syncthreads();
    if (t == 1023) s_carry = woff + incl;
    __syncthreads();
  }
  if (t == 0) { c[nseg] = s_carry; count[blockIdx.x] = s_carry; }
}

// one wave per segment again: its valid pixels go to consecutive slots behind the segment's offset, in pixel order
__global__ __launch_bounds__(256) void fe_scatter_kernel(FeValid V, int npix, int nseg, const int32_t *__restrict__ seg_off,
                                                         int32_t *__restrict__ valid_pix) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (seg >= nseg) return;
  const size_t img = (size_t)b * npix;
  int pos = seg_off[(size_t)b * (nseg + 1) + seg];
  int32_t *out = valid_pix + img;
  for (int j = 0; j < FE_SEG / 64; ++j) {
    const int p = seg * FE_SEG + j * 64 + lane;
    const bool ok = V(img, p, npix);
    const unsigned long long m = __ballot(ok);
    if (ok) out[pos + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = p;
    pos += __popcll(m);
  }
}

__global__ __launch_bounds__(256) void fe_sample_kernel(const uint16_t *__restrict__ depth, int w, int npix, double fx, double fy,
                                                        double cx, double cy, double factor,
                                                        const int32_t *__restrict__ valid_pix, const int32_t *__restrict__ count,
                                                        const int32_t *__restrict__ positions,
                                                        const uint32_t *__restrict__ words, int n, float *__restrict__ out,
                                                        int32_t *__restrict__ out_pix) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= n) return;
  const int cnt = count[b];
  float *o = out + ((size_t)b * n + i) * 3;
  if (cnt <= 0) {                     // an image without a single valid pixel: a cloud of zeros, pixel -1
    o[0] = o[1] = o[2] = 0.f;
    if (out_pix) out_pix[(size_t)b * n + i] = -1;
    return;
  }
  int pos;
  if (positions) pos = positions[(size_t)b * n + i];
  else if (i < cnt && cnt < n) pos = i;                       // fewer valid pixels than points: all of them first
  else pos = (int)(((unsigned long long)words[(size_t)b * n + i] * (unsigned long long)cnt) >> 32);
  pos = min(max(pos, 0), cnt - 1);
  const int p = valid_pix[(size_t)b * npix + pos];
  const int row = p / w, col = p - row * w;
  const double z = (double)depth[(size_t)b * npix + p] / factor;      // data_utils.py:37
  const double x = ((double)col - cx) * z / fx;                       // :38
  const double y = ((double)row - cy) * z / fy;                       // :39
  o[0] = (float)x; o[1] = (float)y; o[2] = (float)z;
  if (out_pix) out_pix[(size_t)b * n + i] = p;
}

// cloud_masked itself (dataset/graspnet_dataset.py:133, demo.py:455: `cloud[mask]`): valid pixel i -> row i, rows past the
// image's count (or past nmax) zero -- the scene every caller hands to the collision detector (test.py:92-96 with
// return_raw_cloud=True; demo.py builds it from cloud_masked), as opposed to the num_point rows sampled for the network
__global__ __launch_bounds__(256) void fe_masked_cloud_kernel(const uint16_t *__restrict__ depth, int w, int npix, double fx, double fy,
                                                              double cx, double cy, double factor,
                                                              const int32_t *__restrict__ valid_pix, const int32_t *__restrict__ count,
                                                              int nmax, float *__restrict__ out, int32_t *__restrict__ out_count) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  const int cnt = min(count[b], nmax);
  if (i == 0 && out_count) out_count[b] = cnt;
  if (i >= nmax) return;
  float *o = out + ((size_t)b * nmax + i) * 3;
  if (i >= cnt) { o[0] = o[1] = o[2] = 0.f; return; }
  const int p = valid_pix[(size_t)b * npix + i];
  const int row = p / w, col = p - row * w;
  const double z = (double)depth[(size_t)b * npix + p] / factor;      // data_utils.py:37
  o[0] = (float)(((double)col - cx) * z / fx);                        // :38
  o[1] = (float)(((double)row - cy) * z / fy);                        // :39
  o[2] = (float)z;
}

// ---- the draw (dataset/graspnet_dataset.py:139-145, demo.py:458-463) -------------------------------------------------
//   len(cloud_masked) >= num_point:  idxs = np.random.choice(len, num_point, replace=False)
//   else:                            idxs = concat(arange(len), np.random.choice(len, num_point - len, replace=True))
// numpy's stream cannot be reproduced on a device; what CAN be kept is the distribution: WITHOUT replacement when there are
// enough valid pixels (no duplicated points reach the sampler and the groupers, as in the reference).  position i = P(i) for
// a keyed pseudo-random PERMUTATION P of [0, count): a 4-round Feistel network over the next even power of two >= count,
// cycle-walked back into range (a bijection on [0, 2^k) restricted to [0, count) by iterating until the value falls below
// count is a bijection on [0, count): num_point distinct positions, one independent hash chain per output, no sort, no
// atomics, integer arithmetic only -- oracle/post_oracle.py draw_positions restates it bit for bit).
__device__ __host__ __forceinline__ uint32_t fe_hash32(uint32_t x) {      // lowbias32
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__global__ __launch_bounds__(256) void fe_draw_kernel(const int32_t *__restrict__ count, const uint32_t *__restrict__ keys, int n,
                                                      int32_t *__restrict__ positions) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= n) return;
  const int cnt = count[b];
  const uint32_t k0 = keys[b * 4 + 0], k1 = keys[b * 4 + 1], k2 = keys[b * 4 + 2], k3 = keys[b * 4 + 3];
  int pos = 0;
  if (cnt <= 0) {
    pos = 0;
  } else if (cnt < n) {
    // fewer valid pixels than points: all of them first, then draws with replacement
    pos = i < cnt ? i : (int)(((unsigned long long)fe_hash32((uint32_t)i * 0x9e3779b9u ^ k0) * (unsigned long long)cnt) >> 32);
  } else {
    int half = 1;                                      // bits per Feistel half: 2 * half >= ceil(log2(cnt))
    while ((1ll << (2 * half)) < (long long)cnt) ++half;
    const uint32_t hm = (1u << half) - 1u;
    uint32_t x = (uint32_t)i;
    for (long long it = 0; it < (1ll << (2 * half)); ++it) {      // cycle walk: expected < 4 turns, always ends
      uint32_t l = x >> half, r = x & hm;
      uint32_t t;
      t = l ^ (fe_hash32(r ^ k0) & hm); l = r; r = t;
      t = l ^ (fe_hash32(r ^ k1) & hm); l = r; r = t;
      t = l ^ (fe_hash32(r ^ k2) & hm); l = r; r = t;
      t = l ^ (fe_hash32(r ^ k3) & hm); l = r; r = t;
      x = (l << half) | r;
      if (x < (uint32_t)cnt) break;
    }
    pos = (int)x;
  }
  positions[(size_t)b * n + i] = pos;
}

static inline int fe_nseg(long npix) { return (int)((npix + FE_SEG - 1) / FE_SEG); }

}  // namespace gn

using namespace gn;

extern "C" size_t gn_frontend_workspace_bytes(int b, int h, int w) {
  if (b <= 0 || h <= 0 || w <= 0) return 0;
  return (size_t)b * (fe_nseg((long)h * w) + 1) * sizeof(int32_t);
}

extern "C" int gn_depth_valid_compact(const uint16_t *depth, const uint8_t *mask, int b, int h, int w, double factor_depth,
                                      double depth_min, double depth_max, void *workspace, int32_t *valid_pix, int32_t *count,
                                      gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && h > 0 && w > 0 && factor_depth != 0.0, "depth_valid_compact: bad sizes");
  GN_REQUIRE((long)h * w < 0x7fffffffL && b <= 65535, "depth_valid_compact: image / batch too large");
  if (b == 0) return GN_OK;
  GN_REQUIRE(depth && workspace && valid_pix && count, "depth_valid_compact: null pointer");
  const int npix = h * w, nseg = fe_nseg(npix);
  FeValid V{depth, mask, factor_depth, depth_min, depth_max};
  int32_t *seg = static_cast<int32_t *>(workspace);
  hipStream_t s = as_stream(stream);
  const dim3 g((nseg + 3) / 4, b);
  hipLaunchKernelGGL(fe_count_kernel, g, dim3(256), 0, s, V, npix, nseg, seg);
  hipLaunchKernelGGL(fe_scan_kernel, dim3(b), dim3(1024), 0, s, nseg, seg, count);
  hipLaunchKernelGGL(fe_scatter_kernel, g, dim3(256), 0, s, V, npix, nseg, seg, valid_pix);
  GN_CHECK_LAUNCH("depth_valid_compact");
  return GN_OK;
}

extern "C" int gn_depth_sample_cloud(const uint16_t *depth, int b, int h, int w, double fx, double fy, double cx, double cy,
                                     double factor_depth, const int32_t *valid_pix, const int32_t *count,
                                     const int32_t *positions, const uint32_t *random_words, int n, float *out,
                                     int32_t *out_pix, gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && h > 0 && w > 0 && n >= 0 && factor_depth != 0.0 && fx != 0.0 && fy != 0.0, "depth_sample_cloud: bad sizes");
  GN_REQUIRE((long)h * w < 0x7fffffffL && b <= 65535, "depth_sample_cloud: image / batch too large");
  if (b == 0 || n == 0) return GN_OK;
  GN_REQUIRE(depth && valid_pix && count && out, "depth_sample_cloud: null pointer");
  GN_REQUIRE(positions || random_words, "depth_sample_cloud: positions or random words are needed");
  hipLaunchKernelGGL(fe_sample_kernel, dim3((n + 255) / 256, b), dim3(256), 0, as_stream(stream), depth, w, h * w, fx, fy, cx, cy,
                     factor_depth, valid_pix, count, positions, random_words, n, out, out_pix);
  GN_CHECK_LAUNCH("depth_sample_cloud");
  return GN_OK;
}

extern "C" int gn_depth_masked_cloud(const uint16_t *depth, int b, int h, int w, double fx, double fy, double cx, double cy,
                                     double factor_depth, const int32_t *valid_pix, const int32_t *count, int nmax, float *out,
                                     int32_t *out_count, gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && h > 0 && w > 0 && nmax >= 0 && factor_depth != 0.0 && fx != 0.0 && fy != 0.0, "depth_masked_cloud: bad sizes");
  GN_REQUIRE((long)h * w < 0x7fffffffL && b <= 65535 && nmax <= h * w, "depth_masked_cloud: image / batch too large or nmax > H*W");
  if (b == 0) return GN_OK;
  GN_REQUIRE(depth && valid_pix && count && (out || nmax == 0), "depth_masked_cloud: null pointer");
  hipLaunchKernelGGL(fe_masked_cloud_kernel, dim3((nmax > 0 ? nmax + 255 : 256) / 256, b), dim3(256), 0, as_stream(stream), depth, w, h * w,
                     fx, fy, cx, cy, factor_depth, valid_pix, count, nmax, out, out_count);
  GN_CHECK_LAUNCH("depth_masked_cloud");
  return GN_OK;
}

extern "C" int gn_depth_draw_positions(const int32_t *count, const uint32_t *keys, int b, int n, int32_t *positions, gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n >= 0 && b <= 65535, "depth_draw_positions: bad sizes");
  if (b == 0 || n == 0) return GN_OK;
  GN_REQUIRE(count && keys && positions, "depth_draw_positions: null pointer");
  hipLaunchKernelGGL(fe_draw_kernel, dim3((n + 255) / 256, b), dim3(256), 0, as_stream(stream), count, keys, n, positions);
  GN_CHECK_LAUNCH("depth_draw_positions");
  return GN_OK;
}
