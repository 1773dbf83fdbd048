// group.hip -- gather / group / 3-NN / 3-interpolate (+ their gradients) for gfx950.
//
// Replaces pointnet2/_ext_src/src/sampling_gpu.cu:13-62 (gather_points[_grad]),
// group_points_gpu.cu:13-80, interpolate_gpu.cu:14-159.
//
// These are HBM-bound copies.  The reference launches one block per batch
// element; here the (batch, channel, position) space is spread over the whole
// chip with the OUTPUT index on the lane so stores are fully coalesced, and the
// index row is read once per lane and reused for every channel.
#include "gn_common.h"

namespace gn {

// out[b,c,j] = points[b,c,idx[b,j]]   (sampling_gpu.cu:13-25)
__global__ __launch_bounds__(256) void gather_points_kernel(const float *__restrict__ points,
                                                            const int32_t *__restrict__ idx, int c,
                                                            int n, int m, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const int a = idx[(size_t)b * m + j];
  for (int l = blockIdx.y; l < c; l += gridDim.y)
    out[((size_t)b * c + l) * m + j] = points[((size_t)b * c + l) * n + a];
}

// grad_points[b,c,idx[b,j]] += grad_out[b,c,j]   (sampling_gpu.cu:39-52)
__global__ __launch_bounds__(256) void gather_points_grad_kernel(const float *__restrict__ grad_out,
                                                                 const int32_t *__restrict__ idx, int c,
                                                                 int n, int m,
                                                                 float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= m) return;
  const int a = idx[(size_t)b * m + j];
  for (int l = blockIdx.y; l < c; l += gridDim.y)
    atomicAdd(grad_points + ((size_t)b * c + l) * n + a, grad_out[((size_t)b * c + l) * m + j]);
}

// out[b,c,j,k] = points[b,c,idx[b,j,k]]   (group_points_gpu.cu:13-33); e = j*nsample + k
__global__ __launch_bounds__(256) void group_points_kernel(const float *__restrict__ points,
                                                           const int32_t *__restrict__ idx, int c, int n,
                                                           int ne, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= ne) return;
  const int a = idx[(size_t)b * ne + e];
  for (int l = blockIdx.y; l < c; l += gridDim.y)
    out[((size_t)b * c + l) * ne + e] = points[((size_t)b * c + l) * n + a];
}

// group_points_gpu.cu:48-69
__global__ __launch_bounds__(256) void group_points_grad_kernel(const float *__restrict__ grad_out,
                                                                const int32_t *__restrict__ idx, int c,
                                                                int n, int ne,
                                                                float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= ne) return;
  const int a = idx[(size_t)b * ne + e];
  for (int l = blockIdx.y; l < c; l += gridDim.y)
    atomicAdd(grad_points + ((size_t)b * c + l) * n + a, grad_out[((size_t)b * c + l) * ne + e]);
}

// interpolate_gpu.cu:14-64: one lane per unknown point, serial ascending scan of `known`
// (order-dependent strict '<' cascade, so the scan order is the reference's).
// The known set is staged through LDS in tiles so each coordinate is read from HBM once per block.
constexpr int NN_THREADS = 64;   // small blocks: the scan is a serial chain per lane, so spread lanes over many CUs
__global__ __launch_bounds__(NN_THREADS) void three_nn_kernel(const float *__restrict__ unknown,
                                                       const float *__restrict__ known, int n, int m,
                                                       float *__restrict__ dist2,
                                                       int32_t *__restrict__ idx) {
  constexpr int TILE = 1024;
  __shared__ float s_known[TILE * 3];
  const int b = blockIdx.y;
  const int j = blockIdx.x * NN_THREADS + threadIdx.x;
  unknown += (size_t)b * n * 3;
  known += (size_t)b * m * 3;
  const bool live = j < n;
  const float ux = live ? unknown[j * 3 + 0] : 0.f;
  const float uy = live ? unknown[j * 3 + 1] : 0.f;
  const float uz = live ? unknown[j * 3 + 2] : 0.f;
  // reference keeps the bests in doubles initialised to 1e40 (interpolate_gpu.cu:33); a float d
  // compares against them exactly like against +inf, and (float)1e40 == +inf on output.
  float best1 = INFINITY, best2 = INFINITY, best3 = INFINITY;
  int besti1 = 0, besti2 = 0, besti3 = 0;
  for (int k0 = 0; k0 < m; k0 += TILE) {
    const int cnt = min(TILE, m - k0);
    __syncthreads();
    for (int i = threadIdx.x; i < cnt * 3; i += NN_THREADS) s_known[i] = known[k0 * 3 + i];
    __syncthreads();
    for (int kk = 0; kk < cnt; ++kk) {
      const float x = s_known[kk * 3 + 0], y = s_known[kk * 3 + 1], z = s_known[kk * 3 + 2];
      const float d = (ux - x) * (ux - x) + (uy - y) * (uy - y) + (uz - z) * (uz - z);
      const int k = k0 + kk;
      if (d < best1) {
        best3 = best2; besti3 = besti2;
        best2 = best1; besti2 = besti1;
        best1 = d;     besti1 = k;
      } else if (d < best2) {
        best3 = best2; besti3 = besti2;
        best2 = d;     besti2 = k;
      } else if (d < best3) {
        best3 = d;     besti3 = k;
      }
    }
  }
  if (live) {
    float *od = dist2 + ((size_t)b * n + j) * 3;
    int32_t *oi = idx + ((size_t)b * n + j) * 3;
    od[0] = best1; od[1] = best2; od[2] = best3;
    oi[0] = besti1; oi[1] = besti2; oi[2] = besti3;
  }
}

// interpolate_gpu.cu:77-106: out[b,c,j] = p[c,i1]*w1 + p[c,i2]*w2 + p[c,i3]*w3 (sum order as written)
__global__ __launch_bounds__(256) void three_interpolate_kernel(const float *__restrict__ points,
                                                                const int32_t *__restrict__ idx,
                                                                const float *__restrict__ weight, int c,
                                                                int m, int n, float *__restrict__ out) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int32_t *ix = idx + ((size_t)b * n + j) * 3;
  const float *w = weight + ((size_t)b * n + j) * 3;
  const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  for (int l = blockIdx.y; l < c; l += gridDim.y) {
    const float *p = points + ((size_t)b * c + l) * m;
    out[((size_t)b * c + l) * n + j] = p[i1] * w1 + p[i2] * w2 + p[i3] * w3;
  }
}

// interpolate_gpu.cu:121-148
__global__ __launch_bounds__(256) void three_interpolate_grad_kernel(
    const float *__restrict__ grad_out, const int32_t *__restrict__ idx,
    const float *__restrict__ weight, int c, int n, int m, float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  const int32_t *ix = idx + ((size_t)b * n + j) * 3;
  const float *w = weight + ((size_t)b * n + j) * 3;
  const int i1 = ix[0], i2 = ix[1], i3 = ix[2];
  const float w1 = w[0], w2 = w[1], w3 = w[2];
  for (int l = blockIdx.y; l < c; l += gridDim.y) {
    float *g = grad_points + ((size_t)b * c + l) * m;
    const float go = grad_out[((size_t)b * c + l) * n + j];
    atomicAdd(g + i1, go * w1);
    atomicAdd(g + i2, go * w2);
    atomicAdd(g + i3, go * w3);
  }
}

static inline int chan_split(int c) { return c < 1 ? 1 : (c > 64 ? 64 : c); }

}  // namespace gn

using namespace gn;

#define GN_COMMON_CHECKS(name, ok)                                                     \
  GN_REQUIRE(ok, "%s: bad sizes", name);                                              \
  GN_REQUIRE(b <= 65535, "%s: batch > 65535 unsupported", name)

extern "C" int gn_gather_points(const float *points, const int32_t *idx, int b, int c, int n, int m,
                                float *out, gn_stream_t stream) {
  GN_COMMON_CHECKS("gather_points", b >= 0 && c >= 0 && n >= 0 && m >= 0);
  if (b == 0 || c == 0 || m == 0) return GN_OK;
  GN_REQUIRE(points && idx && out, "gather_points: null pointer");
  dim3 grid((m + 255) / 256, chan_split(c), b);
  hipLaunchKernelGGL(gather_points_kernel, grid, dim3(256), 0, as_stream(stream), points, idx, c, n, m, out);
  GN_CHECK_LAUNCH("gather_points");
  return GN_OK;
}

extern "C" int gn_gather_points_grad(const float *grad_out, const int32_t *idx, int b, int c, int n, int m,
                                     float *grad_points, gn_stream_t stream) {
  GN_COMMON_CHECKS("gather_points_grad", b >= 0 && c >= 0 && n >= 0 && m >= 0);
  if (b == 0 || c == 0 || n == 0) return GN_OK;
  GN_REQUIRE(grad_out && idx && grad_points, "gather_points_grad: null pointer");
  if (hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, as_stream(stream)) != hipSuccess) {
    set_error("gather_points_grad: memset failed");
    return GN_ERR_LAUNCH;
  }
  if (m == 0) return GN_OK;
  dim3 grid((m + 255) / 256, chan_split(c), b);
  hipLaunchKernelGGL(gather_points_grad_kernel, grid, dim3(256), 0, as_stream(stream), grad_out, idx, c, n,
                     m, grad_points);
  GN_CHECK_LAUNCH("gather_points_grad");
  return GN_OK;
}

extern "C" int gn_group_points(const float *points, const int32_t *idx, int b, int c, int n, int npoints,
                               int nsample, float *out, gn_stream_t stream) {
  GN_COMMON_CHECKS("group_points", b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
  const long ne = (long)npoints * nsample;
  if (b == 0 || c == 0 || ne == 0) return GN_OK;
  GN_REQUIRE(points && idx && out, "group_points: null pointer");
  GN_REQUIRE(ne < 0x7fffffffL, "group_points: npoints*nsample too large");
  dim3 grid((unsigned)((ne + 255) / 256), chan_split(c), b);
  hipLaunchKernelGGL(group_points_kernel, grid, dim3(256), 0, as_stream(stream), points, idx, c, n, (int)ne,
                     out);
  GN_CHECK_LAUNCH("group_points");
  return GN_OK;
}

extern "C" int gn_group_points_grad(const float *grad_out, const int32_t *idx, int b, int c, int n,
                                    int npoints, int nsample, float *grad_points, gn_stream_t stream) {
  GN_COMMON_CHECKS("group_points_grad", b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
  const long ne = (long)npoints * nsample;
  if (b == 0 || c == 0 || n == 0) return GN_OK;
  GN_REQUIRE(grad_out && idx && grad_points, "group_points_grad: null pointer");
  GN_REQUIRE(ne < 0x7fffffffL, "group_points_grad: npoints*nsample too large");
  if (hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * n, as_stream(stream)) != hipSuccess) {
    set_error("group_points_grad: memset failed");
    return GN_ERR_LAUNCH;
  }
  if (ne == 0) return GN_OK;
  dim3 grid((unsigned)((ne + 255) / 256), chan_split(c), b);
  hipLaunchKernelGGL(group_points_grad_kernel, grid, dim3(256), 0, as_stream(stream), grad_out, idx, c, n,
                     (int)ne, grad_points);
  GN_CHECK_LAUNCH("group_points_grad");
  return GN_OK;
}

extern "C" int gn_three_nn(const float *unknown, const float *known, int b, int n, int m, float *dist2,
                           int32_t *idx, gn_stream_t stream) {
  GN_COMMON_CHECKS("three_nn", b >= 0 && n >= 0 && m >= 0);
  if (b == 0 || n == 0) return GN_OK;
  GN_REQUIRE(unknown && dist2 && idx && (known || m == 0), "three_nn: null pointer");
  dim3 grid((n + NN_THREADS - 1) / NN_THREADS, b);
  hipLaunchKernelGGL(three_nn_kernel, grid, dim3(NN_THREADS), 0, as_stream(stream), unknown, known, n, m, dist2, idx);
  GN_CHECK_LAUNCH("three_nn");
  return GN_OK;
}

extern "C" int gn_three_interpolate(const float *points, const int32_t *idx, const float *weight, int b,
                                    int c, int m, int n, float *out, gn_stream_t stream) {
  GN_COMMON_CHECKS("three_interpolate", b >= 0 && c >= 0 && n >= 0 && m >= 0);
  if (b == 0 || c == 0 || n == 0) return GN_OK;
  GN_REQUIRE(points && idx && weight && out, "three_interpolate: null pointer");
  dim3 grid((n + 255) / 256, chan_split(c), b);
  hipLaunchKernelGGL(three_interpolate_kernel, grid, dim3(256), 0, as_stream(stream), points, idx, weight, c,
                     m, n, out);
  GN_CHECK_LAUNCH("three_interpolate");
  return GN_OK;
}

extern "C" int gn_three_interpolate_grad(const float *grad_out, const int32_t *idx, const float *weight,
                                         int b, int c, int n, int m, float *grad_points,
                                         gn_stream_t stream) {
  GN_COMMON_CHECKS("three_interpolate_grad", b >= 0 && c >= 0 && n >= 0 && m >= 0);
  if (b == 0 || c == 0 || m == 0) return GN_OK;
  GN_REQUIRE(grad_out && idx && weight && grad_points, "three_interpolate_grad: null pointer");
  if (hipMemsetAsync(grad_points, 0, sizeof(float) * (size_t)b * c * m, as_stream(stream)) != hipSuccess) {
    set_error("three_interpolate_grad: memset failed");
    return GN_ERR_LAUNCH;
  }
  if (n == 0) return GN_OK;
  dim3 grid((n + 255) / 256, chan_split(c), b);
  hipLaunchKernelGGL(three_interpolate_grad_kernel, grid, dim3(256), 0, as_stream(stream), grad_out, idx,
                     weight, c, n, m, grad_points);
  GN_CHECK_LAUNCH("three_interpolate_grad");
  return GN_OK;
}
