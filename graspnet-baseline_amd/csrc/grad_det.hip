// grad_det.hip -- DETERMINISTIC gradients of gather_points / group_points / three_interpolate.
//
// The reference scatters with float atomicAdd (sampling_gpu.cu:39-62, group_points_gpu.cu:48-80,
// interpolate_gpu.cu:121-159), so its sums depend on the order the hardware retires the atomics: two runs of one
// input differ in the last bits.  Here the scatter is turned into a sorted-segment sum: the index
// tensor is inverted once per call into, for every destination point, the ascending list of the entries that hit
// it, and one thread per (batch, channel, destination) adds its entries in that order -- the order a sequential
// loop over the entries (oracle/gn_oracle.c gno_*_grad) uses, so the results are bit-identical to it, every run.
//
// Inversion (workspace: counts/offsets (B, n+1), cursors (B, n), lists (B, ne), all int32):
//   count    one integer atomicAdd per entry (an integer sum has no order)
//   scan     exclusive prefix per cloud, one workgroup each
//   fill     slot = atomicAdd(cursor) -- entries land in their destination's segment in ARBITRARY order
//   order    segments of <= 64 entries: insertion sort by the lane that owns the destination; up to 4096 entries: a
//            bitonic sort of the segment by the whole wave in LDS (ball_query pads short lists by repeating their first
//            hit, so real SA index tensors hold many destinations with hundreds of entries); longer ones are rebuilt
//            with an ordered ballot scan over the cloud's entries
// Entries whose index lies outside [0, n) are dropped (the reference would write out of bounds).
#include "gn_common.h"

namespace gn {

constexpr int SORT_MAX = 64;
constexpr int WSORT_MAX = 4096;     // longest segment a wave sorts in its 16 KB of LDS

__global__ __launch_bounds__(256) void inv_count_kernel(const int32_t *__restrict__ idx, int n, int ne,
                                                        int32_t *__restrict__ cnt) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= ne) return;
  const int a = idx[(size_t)b * ne + e];
  if (a >= 0 && a < n) atomicAdd(&cnt[(size_t)b * (n + 1) + a], 1);
}

// exclusive scan of cnt[b][0..n] in place (cnt[b][n] becomes the total)
__global__ __launch_bounds__(1024) void inv_scan_kernel(int n, int32_t *__restrict__ cnt) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  int32_t *c = cnt + (size_t)blockIdx.x * (n + 1);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) s_carry = 0;
  __syncthreads();
  for (int base = 0; base <= n; base += 1024) {
    const int i = base + t;
    const int v = i <= n ? c[i] : 0;
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
    if (i <= n) c[i] = woff + incl - v;
    __syncthreads();
    if (t == 1023) s_carry = woff + incl;
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void inv_fill_kernel(const int32_t *__restrict__ idx, int n, int ne,
                                                       const int32_t *__restrict__ off, int32_t *__restrict__ cursor,
                                                       int32_t *__restrict__ list) {
  const int b = blockIdx.y;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= ne) return;
  const int a = idx[(size_t)b * ne + e];
  if (a < 0 || a >= n) return;
  const int slot = atomicAdd(&cursor[(size_t)b * n + a], 1);
  list[(size_t)b * ne + off[(size_t)b * (n + 1) + a] + slot] = e;
}

// one wave per 64 consecutive destinations.  skip (may be NULL): destinations whose sum does not depend on the order of its
// entries (post.hip: a voxel whose float64 partial sums are all exact) keep their ticket order
__global__ __launch_bounds__(256) void inv_order_kernel(const int32_t *__restrict__ idx, int n, int ne,
                                                        const int32_t *__restrict__ off, int32_t *__restrict__ list,
                                                        const uint8_t *__restrict__ skip) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int a0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (a0 >= n) return;      // wave-uniform
  const int a = a0 + lane;
  const int32_t *o = off + (size_t)b * (n + 1);
  int32_t *l = list + (size_t)b * ne;
  const int lo = a < n ? o[a] : 0, hi = a < n ? o[a + 1] : 0;
  const int s = (a < n && skip && skip[(size_t)b * n + a]) ? 0 : hi - lo;
  if (s > 1 && s <= SORT_MAX) {
    for (int i = lo + 1; i < hi; ++i) {
      const int v = l[i];
      int j = i - 1;
      while (j >= lo && l[j] > v) { l[j + 1] = l[j]; --j; }
      l[j + 1] = v;
    }
  }
}

// the long segments, in a launch of their own: its 64 KB of LDS would otherwise cap the common case above (every segment
// <= 64 entries, no LDS at all) at two workgroups per CU.  A wave without a long segment leaves at once.
__global__ __launch_bounds__(256) void inv_order_long_kernel(const int32_t *__restrict__ idx, int n, int ne,
                                                             const int32_t *__restrict__ off, int32_t *__restrict__ list,
                                                             const uint8_t *__restrict__ skip) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int a0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (a0 >= n) return;      // wave-uniform
  const int a = a0 + lane;
  const int32_t *o = off + (size_t)b * (n + 1);
  int32_t *l = list + (size_t)b * ne;
  const int lo = a < n ? o[a] : 0, hi = a < n ? o[a + 1] : 0;
  const int s = (a < n && skip && skip[(size_t)b * n + a]) ? 0 : hi - lo;
  if (!__ballot(s > SORT_MAX)) return;
  // 64 < s <= WSORT_MAX: the wave sorts the segment in LDS (s log^2 s / 64 compare-exchange steps; the ballot rebuild
  // below costs ne / 64 steps per destination whatever s is: 2 048 at SA1's 2048 x 64 entries)
  __shared__ int s_sort[4][WSORT_MAX];
  volatile int *buf = s_sort[threadIdx.x >> 6];
  unsigned long long mid = __ballot(s > SORT_MAX && s <= WSORT_MAX);
  while (mid) {
    const int src = __ffsll((long long)mid) - 1;
    mid &= mid - 1;
    const int slo = __shfl(lo, src), ss = __shfl(s, src);
    int P = 128;
    while (P < ss) P <<= 1;
    for (int i = lane; i < P; i += 64) buf[i] = i < ss ? l[slo + i] : 0x7fffffff;
    __builtin_amdgcn_wave_barrier();
    for (int k = 2; k <= P; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = lane; i < P; i += 64) {
          const int q = i ^ j;
          if (q > i) {
            const int x = buf[i], y = buf[q];
            const bool up = (i & k) == 0;
            if ((x > y) == up) { buf[i] = y; buf[q] = x; }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    for (int i = lane; i < ss; i += 64) l[slo + i] = buf[i];
    __builtin_amdgcn_wave_barrier();
  }
  unsigned long long big = __ballot(s > WSORT_MAX);
  const int32_t *ib = idx + (size_t)b * ne;
  while (big) {
    const int src = __ffsll((long long)big) - 1;
    big &= big - 1;
    const int dest = a0 + src;
    int pos = __shfl(lo, src);
    for (int e0 = 0; e0 < ne; e0 += 64) {
      const int e = e0 + lane;
      const bool hit = e < ne && ib[e] == dest;
      const unsigned long long m = __ballot(hit);
      if (hit) l[pos + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = e;
      pos += __popcll(m);
    }
  }
}

enum { GRAD_GATHER = 0, GRAD_GROUP = 1, GRAD_INTERP = 2 };

// grad_points[b,l,a] = sum over the entries e of destination a, ascending:
//   GATHER / GROUP: grad_out[b,l,e]                (sampling_gpu.cu:46-50, group_points_gpu.cu:59-66)
//   INTERP        : grad_out[b,l,e/3] * weight[b,e] (interpolate_gpu.cu:136-146)
template <int KIND>
__global__ __launch_bounds__(256) void grad_sum_kernel(const float *__restrict__ grad_out, const float *__restrict__ weight,
                                                       const int32_t *__restrict__ off, const int32_t *__restrict__ list, int c,
                                                       int n, int ne, int row, float *__restrict__ grad_points) {
  const int b = blockIdx.z;
  const int a = blockIdx.x * 256 + threadIdx.x;
  if (a >= n) return;
  const int lo = off[(size_t)b * (n + 1) + a], hi = off[(size_t)b * (n + 1) + a + 1];
  const int32_t *l = list + (size_t)b * ne;
  for (int ch = blockIdx.y; ch < c; ch += gridDim.y) {
    const float *go = grad_out + ((size_t)b * c + ch) * row;
    float acc = 0.f;
    for (int t = lo; t < hi; ++t) {
      const int e = l[t];
      if (KIND == GRAD_INTERP) acc += go[e / 3] * weight[(size_t)b * ne + e];
      else acc += go[e];
    }
    grad_points[((size_t)b * c + ch) * n + a] = acc;
  }
}

int seg_order(const int32_t *idx, int b, int n, int ne, const int32_t *off, int32_t *list, hipStream_t s, const uint8_t *skip);

struct InvWs { int32_t *off, *cursor, *list; };

static InvWs inv_carve(void *ws, int b, int n, long ne) {
  InvWs w;
  w.off = static_cast<int32_t *>(ws);
  w.cursor = w.off + (size_t)b * (n + 1);
  w.list = w.cursor + (size_t)b * n;
  (void)ne;
  return w;
}

// idx (b, ne) with values in [0, n) -> per-destination ascending entry lists
static int invert(const int32_t *idx, int b, int n, long ne, void *workspace, InvWs *out, hipStream_t s, const char *name) {
  GN_REQUIRE(workspace != nullptr, "%s: workspace required (gn_grad_workspace_bytes)", name);
  GN_REQUIRE(ne < 0x7fffffffL && b <= 65535, "%s: too many entries / batch > 65535", name);
  const InvWs w = inv_carve(workspace, b, n, ne);
  if (hipMemsetAsync(w.off, 0, sizeof(int32_t) * ((size_t)b * (n + 1) + (size_t)b * n), s) != hipSuccess) {
    set_error("%s: memset failed", name);
    return GN_ERR_LAUNCH;
  }
  if (ne > 0) {
    dim3 ge((unsigned)((ne + 255) / 256), b);
    hipLaunchKernelGGL(inv_count_kernel, ge, dim3(256), 0, s, idx, n, (int)ne, w.off);
    hipLaunchKernelGGL(inv_scan_kernel, dim3(b), dim3(1024), 0, s, n, w.off);
    hipLaunchKernelGGL(inv_fill_kernel, ge, dim3(256), 0, s, idx, n, (int)ne, w.off, w.cursor, w.list);
    if (int rc = seg_order(idx, b, n, (int)ne, w.off, w.list, s, nullptr)) return rc;
  }
  *out = w;
  return GN_OK;
}

// off (b, n+1) / list (b, ne) as the inversion leaves them after its fill pass, idx (b, ne) the destination of every entry:
// puts each destination's segment of `list` in ascending order (shared with the voxel down-sample of post.hip)
int seg_order(const int32_t *idx, int b, int n, int ne, const int32_t *off, int32_t *list, hipStream_t s, const uint8_t *skip) {
  if (b <= 0 || n <= 0 || ne <= 0) return GN_OK;
  dim3 go((unsigned)((n + 255) / 256), b);
  hipLaunchKernelGGL(inv_order_kernel, go, dim3(256), 0, s, idx, n, ne, off, list, skip);
  if (ne > SORT_MAX) hipLaunchKernelGGL(inv_order_long_kernel, go, dim3(256), 0, s, idx, n, ne, off, list, skip);
  GN_CHECK_LAUNCH("seg_order");
  return GN_OK;
}

static inline int chan_split_det(int c) { return c < 1 ? 1 : (c > 64 ? 64 : c); }

}  // namespace gn

using namespace gn;

#define GN_COMMON_CHECKS(name, ok)          \
  GN_REQUIRE(ok, "%s: bad sizes", name);    \
  GN_REQUIRE(b <= 65535, "%s: batch > 65535 unsupported", name)

extern "C" size_t gn_grad_workspace_bytes(int b, int n, long ne) {
  if (b <= 0 || n <= 0 || ne < 0) return 0;
  return sizeof(int32_t) * ((size_t)b * (n + 1) + (size_t)b * n + (size_t)b * ne);
}

extern "C" int gn_gather_points_grad_det(const float *grad_out, const int32_t *idx, int b, int c, int n, int m,
                                         void *workspace, float *grad_points, gn_stream_t stream) {
  GN_COMMON_CHECKS("gather_points_grad_det", b >= 0 && c >= 0 && n >= 0 && m >= 0);
  if (b == 0 || c == 0 || n == 0) return GN_OK;
  GN_REQUIRE(grad_out && idx && grad_points, "gather_points_grad_det: null pointer");
  InvWs w;
  if (int rc = invert(idx, b, n, m, workspace, &w, as_stream(stream), "gather_points_grad_det")) return rc;
  dim3 g((n + 255) / 256, chan_split_det(c), b);
  hipLaunchKernelGGL(grad_sum_kernel<GRAD_GATHER>, g, dim3(256), 0, as_stream(stream), grad_out, (const float *)nullptr, w.off,
                     w.list, c, n, m, m, grad_points);
  GN_CHECK_LAUNCH("gather_points_grad_det");
  return GN_OK;
}

extern "C" int gn_group_points_grad_det(const float *grad_out, const int32_t *idx, int b, int c, int n, int npoints,
                                        int nsample, void *workspace, float *grad_points, gn_stream_t stream) {
  GN_COMMON_CHECKS("group_points_grad_det", b >= 0 && c >= 0 && n >= 0 && npoints >= 0 && nsample >= 0);
  const long ne = (long)npoints * nsample;
  if (b == 0 || c == 0 || n == 0) return GN_OK;
  GN_REQUIRE(grad_out && idx && grad_points, "group_points_grad_det: null pointer");
  InvWs w;
  if (int rc = invert(idx, b, n, ne, workspace, &w, as_stream(stream), "group_points_grad_det")) return rc;
  dim3 g((n + 255) / 256, chan_split_det(c), b);
  hipLaunchKernelGGL(grad_sum_kernel<GRAD_GROUP>, g, dim3(256), 0, as_stream(stream), grad_out, (const float *)nullptr, w.off,
                     w.list, c, n, (int)ne, (int)ne, grad_points);
  GN_CHECK_LAUNCH("group_points_grad_det");
  return GN_OK;
}

extern "C" int gn_three_interpolate_grad_det(const float *grad_out, const int32_t *idx, const float *weight, int b, int c,
                                             int n, int m, void *workspace, float *grad_points, gn_stream_t stream) {
  GN_COMMON_CHECKS("three_interpolate_grad_det", b >= 0 && c >= 0 && n >= 0 && m >= 0);
  if (b == 0 || c == 0 || m == 0) return GN_OK;
  GN_REQUIRE(grad_out && idx && weight && grad_points, "three_interpolate_grad_det: null pointer");
  InvWs w;
  if (int rc = invert(idx, b, m, (long)n * 3, workspace, &w, as_stream(stream), "three_interpolate_grad_det")) return rc;
  dim3 g((m + 255) / 256, chan_split_det(c), b);
  hipLaunchKernelGGL(grad_sum_kernel<GRAD_INTERP>, g, dim3(256), 0, as_stream(stream), grad_out, weight, w.off, w.list, c, m,
                     n * 3, n, grad_points);
  GN_CHECK_LAUNCH("three_interpolate_grad_det");
  return GN_OK;
}
