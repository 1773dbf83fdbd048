// knn.hip -- brute-force k nearest neighbours for gfx950.
//
// Replaces knn/src/cuda/knn.cu:36-263 (+ knn/src/knn.h:12-65 batch loop).
// The reference materialises the full (ref_nb x query_nb) distance matrix in a
// scratch buffer and then insertion-sorts every column; here one lane owns one
// query, walks the reference points in ascending index order and keeps its k
// best (distance, 1-based index) pairs sorted in the caller's (K,NQ) scratch /
// idx planes -- coalesced across lanes because the query index is innermost.
// Result order is the reference's: distance ascending, lower reference index
// first among equal distances (strict compares, knn.cu:129-171 / knn_cpu.cpp:29).
// Distances accumulate over the dimension in order 0..D-1 (knn.cu:76-93).
// Reference points are staged through LDS tiles shared by the block.
#include "gn_common.h"

namespace gn {

__global__ __launch_bounds__(256) void knn_kernel(const float *__restrict__ ref,
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
        dd += t * t;
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

}  // namespace gn

extern "C" int gn_knn(const float *ref, const float *query, int b, int d, int nr, int nq, int k,
                      int64_t *idx, float *scratch, gn_stream_t stream) {
  using namespace gn;
  GN_REQUIRE(b >= 0 && d >= 1 && nr >= 0 && nq >= 0 && k >= 0, "knn: bad sizes");
  if (b == 0 || nq == 0 || k == 0) return GN_OK;
  GN_REQUIRE(ref && query && idx && scratch, "knn: null pointer");
  GN_REQUIRE(d <= 3072, "knn: dim > 3072 unsupported");
  GN_REQUIRE(b <= 65535, "knn: batch > 65535 unsupported");
  dim3 grid((nq + 255) / 256, b);
  hipLaunchKernelGGL(knn_kernel, grid, dim3(256), 0, as_stream(stream), ref, query, d, nr, nq, k, idx,
                     scratch);
  GN_CHECK_LAUNCH("knn");
  return GN_OK;
}
