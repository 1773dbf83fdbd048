// knn.hip -- brute-force k nearest neighbours for gfx950.
//
// Replaces knn/src/cuda/knn.cu:36-263 (+ knn/src/knn.h:12-65 batch loop).
// The reference materialises the full (ref_nb x query_nb) distance matrix in a
// scratch buffer and then insertion-sorts every column.  Result order is the
// reference's: distance ascending, lower reference index first among equal
// distances (strict compares, knn.cu:129-171 / knn_cpu.cpp:29) -- i.e. the k
// smallest under the strict total order (distance, index).  Distances
// accumulate over the dimension in order 0..D-1 (knn.cu:76-93), one rounding
// per operation (this file is built with -ffp-contract=off).
//
// knn_wave_kernel (k <= 8): ONE WAVE PER QUERY (or per 4 queries when there are
// enough of them to fill the chip that way): the 64 lanes split the reference
// points -- lane l takes l, l+64, ... in ascending order, coalesced -- and every
// lane keeps its own k best (distance, index) pairs as a sorted list in
// REGISTERS (a later equal distance never displaces an earlier one: ascending
// visits + strict compares = the reference's stability inside a lane); the 64
// lists are then merged by k rounds of a wave-wide minimum over the 64-bit key
// distance_bits << 32 | index (non-negative floats order like their bit
// patterns), the owner of the minimum popping its head.  At the reference's
// call sites (utils/label_generation.py:58-60,84-86: 300 x 300 and
// Np' x 1024, k = 1) that is 300 / 1024 waves instead of the 2 / 4 workgroups
// of the scan kernel below, whose k-list lived in global memory.
// knn_scan_kernel (k > 8): one lane per query, k-list in the caller's scratch.
#include <stdlib.h>

#include "gn_common.h"

namespace gn {
namespace GN_FPNS {

__global__ __launch_bounds__(256) void knn_scan_kernel(const float *__restrict__ ref,
                                                  const float *__restrict__ query, int d, int nr, int nq,
                                                  int k, int64_t *__restrict__ idx,
                                                  float *__restrict__ dist) {
  constexpr int TILE_FLOATS = 3072;
  __shared__ float s_ref[TILE_FLOATS];
  const int b = blockIdx.y;
  const int q = blockIdx.x * 256 + threadIdx.x;
  ref += (size_t)b * d * nr;
  query += (size_t)b * d * nq;
  idx += (size_t)b * k * nq;
  dist += (size_t)b * k * nq;
  const bool live = q < nq;
  const int tile = TILE_FLOATS / d;  // reference points per tile (d <= TILE_FLOATS checked on host)
  int cnt = 0;
  for (int r0 = 0; r0 < nr; r0 += tile) {
    const int tn = min(tile, nr - r0);
    __syncthreads();
    for (int i = threadIdx.x; i < tn * d; i += 256) {
      const int h = i / tn, rr = i - h * tn;
      s_ref[h * tn + rr] = ref[(size_t)h * nr + r0 + rr];
    }
    __syncthreads();
    if (!live) continue;
    for (int rr = 0; rr < tn; ++rr) {
      float dd = 0.f;
      for (int h = 0; h < d; ++h) {
        const float t = s_ref[h * tn + rr] - query[(size_t)h * nq + q];
        dd = fp_acc_sq(dd, t);                              // knn.cu:89-90
      }
      // stable insertion: after every kept element with dist <= dd
      int pos = cnt;
      while (pos > 0 && dist[(size_t)(pos - 1) * nq + q] > dd) --pos;
      if (pos >= k) continue;
      const int last = cnt < k ? cnt : k - 1;
      for (int t = last; t > pos; --t) {
        dist[(size_t)t * nq + q] = dist[(size_t)(t - 1) * nq + q];
        idx[(size_t)t * nq + q] = idx[(size_t)(t - 1) * nq + q];
      }
      dist[(size_t)pos * nq + q] = dd;
      idx[(size_t)pos * nq + q] = (int64_t)(r0 + rr) + 1;  // 1-based (knn.cu:122,144)
      if (cnt < k) ++cnt;
    }
  }
  if (live)
    for (int t = cnt; t < k; ++t) idx[(size_t)t * nq + q] = 0;  // k > nr: undefined upstream, 0 here
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)v, o), hi = __shfl_xor((unsigned)(v >> 32), o);
    const unsigned long long u = ((unsigned long long)hi << 32) | lo;
    v = u < v ? u : v;
  }
  return v;
}

constexpr unsigned long long KNN_NONE = ~0ULL;

// K: list length (>= k), QW: queries per wave
template <int K, int QW, bool D3>
__global__ __launch_bounds__(256) void knn_wave_kernel(const float *__restrict__ ref, const float *__restrict__ query, int d,
                                                       int nr, int nq, int k, int64_t *__restrict__ idx) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
  const int q0 = wave * QW;
  if (q0 >= nq) return;
  ref += (size_t)b * d * nr;
  query += (size_t)b * d * nq;
  idx += (size_t)b * k * nq;
  // sorted ascending by (distance, index); empty slots hold KNN_NONE (above every real key: a distance is never NaN-free
  // larger than 0x7f800000 = +Inf in the high word with an index below 2^32 - 1)
  unsigned long long best[QW][K];
#pragma unroll
  for (int qi = 0; qi < QW; ++qi)
#pragma unroll
    for (int t = 0; t < K; ++t) best[qi][t] = KNN_NONE;
  // U reference points per lane and pass: their D * U loads are all requested before the first distance is formed (one L2
  // round trip per pass instead of one per point: at 20 000 x 1 024, one query per wave, the loop was 313 dependent round
  // trips -- 0.118 ms for ~15 us of arithmetic); D = 3 (the reference's call sites, utils/label_generation.py:58-60,84-86) is
  // unrolled in full, any other dimension walks h with the U points side by side
  constexpr int U = QW == 1 ? 8 : 4;
  auto insert_pass = [&](int r0, const float (&dd)[U][QW]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {           // ascending r within the lane: a later equal distance never displaces an earlier one
      const int r = r0 + u * 64;
      if (r >= nr) break;
#pragma unroll
      for (int qi = 0; qi < QW; ++qi) {
        // a NaN distance sorts after every number (the reference's strict `>` never moves anything past it either)
        const unsigned bits = dd[u][qi] == dd[u][qi] ? __float_as_uint(dd[u][qi]) : 0x7fc00000u;
        unsigned long long key = ((unsigned long long)bits << 32) | (unsigned)r;
        // insertion keeping the K smallest: bubble the new key down the sorted list
#pragma unroll
        for (int t = 0; t < K; ++t) {
          const unsigned long long cur = best[qi][t];
          const bool less = key < cur;
          best[qi][t] = less ? key : cur;
          key = less ? cur : key;
        }
      }
    }
  };
  if (D3) {
    // the query's coordinates once (wave-uniform), the reference points of pass p+1 requested before pass p is evaluated
    float qv[QW][3];
#pragma unroll
    for (int qi = 0; qi < QW; ++qi)
#pragma unroll
      for (int h = 0; h < 3; ++h) qv[qi][h] = query[(size_t)h * nq + min(q0 + qi, nq - 1)];
    float cur[U][3], nxt[U][3];
    auto fetch = [&](int r0, float (&rv)[U][3]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int r = min(r0 + u * 64, nr - 1);
#pragma unroll
        for (int h = 0; h < 3; ++h) rv[u][h] = ref[(size_t)h * nr + r];
      }
    };
    if (lane < nr) fetch(lane, cur);
    for (int r0 = lane; r0 < nr; r0 += 64 * U) {
      const bool more = r0 + 64 * U < nr;
      if (more) fetch(r0 + 64 * U, nxt);
      float dd[U][QW];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
          float acc = 0.f;
#pragma unroll
          for (int h = 0; h < 3; ++h) acc = fp_acc_sq(acc, cur[u][h] - qv[qi][h]);       // knn.cu:89-90, h ascending
          dd[u][qi] = acc;
        }
      insert_pass(r0, dd);
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int h = 0; h < 3; ++h) cur[u][h] = nxt[u][h];
      }
    }
  } else {
    for (int r0 = lane; r0 < nr; r0 += 64 * U) {
      float dd[U][QW];
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) dd[u][qi] = 0.f;
      for (int h = 0; h < d; ++h) {
        float rv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) rv[u] = ref[(size_t)h * nr + min(r0 + u * 64, nr - 1)];
#pragma unroll
        for (int qi = 0; qi < QW; ++qi) {
          const float qvv = query[(size_t)h * nq + min(q0 + qi, nq - 1)];
#pragma unroll
          for (int u = 0; u < U; ++u) dd[u][qi] = fp_acc_sq(dd[u][qi], rv[u] - qvv);
        }
      }
      insert_pass(r0, dd);
    }
  }
  // merge the 64 lists: k rounds of a wave-wide minimum; its owner pops
#pragma unroll
  for (int qi = 0; qi < QW; ++qi) {
    if (q0 + qi >= nq) break;       // wave-uniform
    for (int t = 0; t < k; ++t) {
      const unsigned long long m = wave_min_u64(best[qi][0]);
      if (best[qi][0] == m && m != KNN_NONE) {
#pragma unroll
        for (int u = 0; u + 1 < K; ++u) best[qi][u] = best[qi][u + 1];
        best[qi][K - 1] = KNN_NONE;
      }
      if (lane == 0) idx[(size_t)t * nq + q0 + qi] = m == KNN_NONE ? 0 : (int64_t)(unsigned)(m & 0xffffffffu) + 1;   // 1-based
    }
  }
}

}  // namespace GN_FPNS
}  // namespace gn

extern "C" int GN_FP_ENTRY(gn_knn)(const float *ref, const float *query, int b, int d, int nr, int nq, int k,
                      int64_t *idx, float *scratch, gn_stream_t stream) {
  using namespace gn;
  using namespace gn::GN_FPNS;
  GN_FP_DISPATCH(gn_knn, ref, query, b, d, nr, nq, k, idx, scratch, stream);
  GN_REQUIRE(b >= 0 && d >= 1 && nr >= 0 && nq >= 0 && k >= 0, "knn: bad sizes");
  if (b == 0 || nq == 0 || k == 0) return GN_OK;
  GN_REQUIRE(ref && query && idx, "knn: null pointer");
  GN_REQUIRE(b <= 65535, "knn: batch > 65535 unsupported");
  const char *ve = getenv("GN_KNN");            // GN_KNN=scan: the one-lane-per-query kernel for every k (A/B, tests)
  const bool scan = k > 8 || (ve && ve[0] == 's');
  if (!scan) {
    // four queries per wave once that still leaves every CU a few workgroups (the reference points are then read once per
    // four queries); one query per wave otherwise -- 1024 queries = 256 workgroups
    const bool qw4 = (long)b * nq >= 4L * 4 * 1024;
#define GN_KNN_LAUNCH(K, QW)                                                                                          \
  do {                                                                                                                \
    if (d == 3)                                                                                                       \
      hipLaunchKernelGGL((knn_wave_kernel<K, QW, true>), dim3(((nq + QW - 1) / QW + 3) / 4, b), dim3(256), 0, as_stream(stream), \
                         ref, query, d, nr, nq, k, idx);                                                              \
    else                                                                                                              \
      hipLaunchKernelGGL((knn_wave_kernel<K, QW, false>), dim3(((nq + QW - 1) / QW + 3) / 4, b), dim3(256), 0, as_stream(stream), \
                         ref, query, d, nr, nq, k, idx);                                                              \
  } while (0)
#define GN_KNN_K(K) do { if (qw4) GN_KNN_LAUNCH(K, 4); else GN_KNN_LAUNCH(K, 1); } while (0)
    if (k == 1) GN_KNN_K(1);
    else if (k == 2) GN_KNN_K(2);
    else if (k <= 4) GN_KNN_K(4);
    else GN_KNN_K(8);
#undef GN_KNN_K
#undef GN_KNN_LAUNCH
    GN_CHECK_LAUNCH("knn");
    return GN_OK;
  }
  GN_REQUIRE(scratch, "knn: scratch required for k > 8");
  GN_REQUIRE(d <= 3072, "knn: dim > 3072 unsupported");
  dim3 grid((nq + 255) / 256, b);
  hipLaunchKernelGGL(knn_scan_kernel, grid, dim3(256), 0, as_stream(stream), ref, query, d, nr, nq, k, idx,
                     scratch);
  GN_CHECK_LAUNCH("knn");
  return GN_OK;
}
