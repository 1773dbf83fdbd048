// group.hip -- gather / group / 3-NN / 3-interpolate (+ their gradients) for gfx950.
//
// Replaces pointnet2/_ext_src/src/sampling_gpu.cu:13-62 (gather_points[_grad]),
// group_points_gpu.cu:13-80, interpolate_gpu.cu:14-159.
//
// These are HBM-bound copies.  The reference launches one block per batch
// element; here the (batch, channel, position) space is spread over the whole
// chip with the OUTPUT index on the lane so stores are fully coalesced, and the
// index row is read once per lane and reused for every channel.
#include <stdlib.h>

#include "gn_common.h"

namespace gn {
namespace GN_FPNS {

using f32x4 = __attribute__((ext_vector_type(4))) float;

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

// The same gather for the write-bound shapes (group_points is named in BASELINE.json's metric): a workgroup stages CG
// channel rows of one cloud in LDS (n floats each), then every thread takes 4 consecutive output elements at a time --
// one 16-byte index load, 4 LDS gathers per channel, one 16-byte store per channel (1 KiB per wave-instruction instead
// of 256 B, and no 4-byte global gathers).  grid = (element splits, channel groups, batch); ne % 4 == 0.
template <int CG>
__global__ __launch_bounds__(256) void group_points_lds_kernel(const float *__restrict__ points,
                                                               const int32_t *__restrict__ idx, int c, int n, int ne,
                                                               int quads_per_block, float *__restrict__ out) {
  extern __shared__ float s_rows[];     // [CG][n]
  const int b = blockIdx.z, c0 = blockIdx.y * CG, tid = threadIdx.x;
  const int nc = min(CG, c - c0);
  const float *src = points + ((size_t)b * c + c0) * n;      // nc consecutive rows = one contiguous block of nc*n floats
  const int total = nc * n;
  if ((n & 3) == 0) {
    for (int i = tid * 4; i < total; i += 256 * 4) *reinterpret_cast<f32x4 *>(s_rows + i) = *reinterpret_cast<const f32x4 *>(src + i);
  } else {
    for (int i = tid; i < total; i += 256) s_rows[i] = src[i];
  }
  __syncthreads();
  const int nq = ne >> 2;
  const int q0 = blockIdx.x * quads_per_block, q1 = min(nq, q0 + quads_per_block);
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  const i32x4 *iq = reinterpret_cast<const i32x4 *>(idx + (size_t)b * ne);
  float *ob = out + ((size_t)b * c + c0) * ne;
  for (int q = q0 + tid; q < q1; q += 256) {
    const i32x4 a = iq[q];
#pragma unroll 4
    for (int l = 0; l < nc; ++l) {
      const float *r = s_rows + l * n;
      const f32x4 v{r[a[0]], r[a[1]], r[a[2]], r[a[3]]};
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4 *>(ob + (size_t)l * ne) + q);
    }
  }
}

// rows too long for LDS (SA1 / CloudCrop: C = 3, n = N): 4 consecutive elements per thread straight from global,
// 16-byte stores, two channels' gathers in flight.  ne % 4 == 0.
__global__ __launch_bounds__(256) void group_points_vec_kernel(const float *__restrict__ points,
                                                               const int32_t *__restrict__ idx, int c, int n, int ne,
                                                               float *__restrict__ out) {
  const int b = blockIdx.z;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= (ne >> 2)) return;
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  const i32x4 a = reinterpret_cast<const i32x4 *>(idx + (size_t)b * ne)[q];
  for (int l = blockIdx.y; l < c; l += gridDim.y) {
    const float *r = points + ((size_t)b * c + l) * n;
    const f32x4 v{r[a[0]], r[a[1]], r[a[2]], r[a[3]]};
    reinterpret_cast<f32x4 *>(out + ((size_t)b * c + l) * ne)[q] = v;
  }
}

// the same for C <= 4 channels (the xyz rows of SA1 / CloudCrop): ONE thread serves all channels of its 4 elements -- the
// index quad is loaded once (the per-channel form fetched it once per channel: 1.7-2.0 x the algorithmic bytes by PMC,
// profiles/r4_ops_summary.txt) and all 4 C gathers are in flight together before the first 16-byte store
template <int C>
__global__ __launch_bounds__(256) void group_points_vec_small_kernel(const float *__restrict__ points,
                                                                     const int32_t *__restrict__ idx, int n, int ne,
                                                                     float *__restrict__ out) {
  const int b = blockIdx.z;
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= (ne >> 2)) return;
  using i32x4 = __attribute__((ext_vector_type(4))) int;
  const i32x4 a = __builtin_nontemporal_load(reinterpret_cast<const i32x4 *>(idx + (size_t)b * ne) + q);
  f32x4 v[C];
#pragma unroll
  for (int l = 0; l < C; ++l) {
    const float *r = points + ((size_t)b * C + l) * n;
    v[l] = f32x4{r[a[0]], r[a[1]], r[a[2]], r[a[3]]};
  }
#pragma unroll
  for (int l = 0; l < C; ++l) __builtin_nontemporal_store(v[l], reinterpret_cast<f32x4 *>(out + ((size_t)b * C + l) * ne) + q);
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

// interpolate_gpu.cu:14-64.  The reference scans `known` serially per unknown point with a strict '<'
// cascade, i.e. it returns the three smallest (distance, index) pairs in lexicographic order.  Here one
// WAVE owns an unknown point: lane l scans known points l, l+64, ... with the same cascade (so its own
// three candidates are the lexicographically smallest of its subset), then the wave extracts the global
// three smallest by three rounds of (min distance, then min index) DPP reductions.
constexpr int NN_WAVES = 4;
__global__ __launch_bounds__(NN_WAVES * 64) void three_nn_kernel(const float *__restrict__ unknown,
                                                                 const float *__restrict__ known, int n, int m,
                                                                 float *__restrict__ dist2,
                                                                 int32_t *__restrict__ idx) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * NN_WAVES + (threadIdx.x >> 6);
  if (j >= n) return;
  unknown += (size_t)b * n * 3;
  known += (size_t)b * m * 3;
  const float ux = unknown[j * 3 + 0], uy = unknown[j * 3 + 1], uz = unknown[j * 3 + 2];
  // reference keeps the bests in doubles initialised to 1e40 (interpolate_gpu.cu:33); a float d
  // compares against them exactly like against +inf, and (float)1e40 == +inf on output.
  float b1 = INFINITY, b2 = INFINITY, b3 = INFINITY;
  int i1 = 0x7fffffff, i2 = 0x7fffffff, i3 = 0x7fffffff;
  for (int k = lane; k < m; k += 64) {
    const float x = known[k * 3 + 0], y = known[k * 3 + 1], z = known[k * 3 + 2];
    const float d = fp_sumsq3(ux - x, uy - y, uz - z);       // interpolate_gpu.cu:37
    if (d < b1) { b3 = b2; i3 = i2; b2 = b1; i2 = i1; b1 = d; i1 = k; }
    else if (d < b2) { b3 = b2; i3 = i2; b2 = d; i2 = k; }
    else if (d < b3) { b3 = d; i3 = k; }
  }
  float od[3];
  int oi[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    // smallest head distance over the wave (distances are >= 0 or +inf: negate for the max reduction)
    const float dmin = -wave_max(-b1);
    const int imin = wave_min_i(b1 == dmin ? i1 : 0x7fffffff);
    od[r] = dmin;
    oi[r] = imin == 0x7fffffff ? 0 : imin;     // fewer than r+1 finite candidates: reference leaves index 0
    if (b1 == dmin && i1 == imin) { b1 = b2; i1 = i2; b2 = b3; i2 = i3; b3 = INFINITY; i3 = 0x7fffffff; }
  }
  if (lane == 0) {
    float *o = dist2 + ((size_t)b * n + j) * 3;
    int32_t *q = idx + ((size_t)b * n + j) * 3;
    o[0] = od[0]; o[1] = od[1]; o[2] = od[2];
    q[0] = oi[0]; q[1] = oi[1]; q[2] = oi[2];
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

}  // namespace GN_FPNS
}  // namespace gn

using namespace gn;
using namespace gn::GN_FPNS;

#define GN_COMMON_CHECKS(name, ok)                                                     \
  GN_REQUIRE(ok, "%s: bad sizes", name);                                              \
  GN_REQUIRE(b <= 65535, "%s: batch > 65535 unsupported", name)

#if GN_FP == 0      // copies and blends: nothing here depends on the evaluation policy
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
  GN_REQUIRE(b <= 65535, "group_points: batch > 65535 unsupported");
  const bool vec = (ne & 3) == 0 && (reinterpret_cast<uintptr_t>(idx) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                   (reinterpret_cast<uintptr_t>(points) & 15) == 0;
  if (vec && c >= 8 && (size_t)n * sizeof(float) * 2 <= 64 * 1024) {
    // LDS-staged rows
    static const int env_cg = []() { const char *e = getenv("GN_GROUP_CG"); return e ? atoi(e) : 0; }();
    static const int env_wgs = []() { const char *e = getenv("GN_GROUP_WGS"); return e ? atoi(e) : 0; }();
    // measured at the SA2 shape (profiles/group_probe.py): 2 rows per workgroup 5.3 TB/s, 4: 5.1-5.3, 8: 4.1-4.6, 16: 3.4-4.7
    // (small LDS footprints keep many workgroups resident, so one's row staging hides under the others' stores)
    int cg = (env_cg == 2 || env_cg == 4 || env_cg == 8 || env_cg == 16) ? env_cg : 2;
    while ((size_t)cg * n * sizeof(float) > 64 * 1024) cg >>= 1;
    const int gy = (c + cg - 1) / cg;
    const int nq = (int)(ne >> 2);
    const int want = env_wgs > 0 ? env_wgs : 1024;             // ~4 workgroups per CU in the grid
    int split = 1;                                             // a power of two, so the element ranges stay even
    while (split * b * gy < want) split <<= 1;
    const int max_split = (nq + 511) / 512;                    // at least two quads per thread per workgroup
    split = split < 1 ? 1 : (split > max_split ? max_split : split);
    const int qpb = (nq + split - 1) / split;
    const size_t lds = (size_t)cg * n * sizeof(float);
    dim3 grid((unsigned)((nq + qpb - 1) / qpb), gy, b);
    auto launch = [&](auto k) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        set_error("group_points: cannot reserve %zu bytes of LDS", lds);
        return GN_ERR_LAUNCH;
      }
      hipLaunchKernelGGL(k, grid, dim3(256), lds, as_stream(stream), points, idx, c, n, (int)ne, qpb, out);
      return GN_OK;
    };
    int rc;
    if (cg == 16) rc = launch(group_points_lds_kernel<16>);
    else if (cg == 8) rc = launch(group_points_lds_kernel<8>);
    else if (cg == 4) rc = launch(group_points_lds_kernel<4>);
    else rc = launch(group_points_lds_kernel<2>);
    if (rc) return rc;
  } else if (vec && c <= 4) {
    dim3 grid((unsigned)(((ne >> 2) + 255) / 256), 1, b);
#define GN_GP_SMALL(CC) hipLaunchKernelGGL(group_points_vec_small_kernel<CC>, grid, dim3(256), 0, as_stream(stream), points, idx, n, (int)ne, out)
    if (c == 1) GN_GP_SMALL(1); else if (c == 2) GN_GP_SMALL(2); else if (c == 3) GN_GP_SMALL(3); else GN_GP_SMALL(4);
#undef GN_GP_SMALL
  } else if (vec) {
    dim3 grid((unsigned)(((ne >> 2) + 255) / 256), chan_split(c), b);
    hipLaunchKernelGGL(group_points_vec_kernel, grid, dim3(256), 0, as_stream(stream), points, idx, c, n, (int)ne, out);
  } else {
    dim3 grid((unsigned)((ne + 255) / 256), chan_split(c), b);
    hipLaunchKernelGGL(group_points_kernel, grid, dim3(256), 0, as_stream(stream), points, idx, c, n, (int)ne,
                       out);
  }
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

#endif

extern "C" int GN_FP_ENTRY(gn_three_nn)(const float *unknown, const float *known, int b, int n, int m, float *dist2,
                           int32_t *idx, gn_stream_t stream) {
  GN_FP_DISPATCH(gn_three_nn, unknown, known, b, n, m, dist2, idx, stream);
  GN_COMMON_CHECKS("three_nn", b >= 0 && n >= 0 && m >= 0);
  if (b == 0 || n == 0) return GN_OK;
  GN_REQUIRE(unknown && dist2 && idx && (known || m == 0), "three_nn: null pointer");
  dim3 grid((n + NN_WAVES - 1) / NN_WAVES, b);
  hipLaunchKernelGGL(three_nn_kernel, grid, dim3(NN_WAVES * 64), 0, as_stream(stream), unknown, known, n, m, dist2, idx);
  GN_CHECK_LAUNCH("three_nn");
  return GN_OK;
}

#if GN_FP == 0
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
#endif
