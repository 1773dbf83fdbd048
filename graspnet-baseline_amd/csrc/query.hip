// query.hip -- ball query and cylinder query for gfx950.
//
// Replaces pointnet2/_ext_src/src/ball_query_gpu.cu:14-59 and
// cylinder_query_gpu.cu:9-68.  The reference gives each centre to ONE thread
// that scans the cloud serially; here ONE WAVE owns a centre: the 64 lanes test
// 64 consecutive points at once, a 64-bit ballot + prefix popcount assigns the
// output slots in index order (so "first nsample hits in index order" is kept
// exactly), and the wave leaves the scan as soon as it has nsample hits.
// The predicate is evaluated exactly as the reference writes it (this file is
// built with -ffp-contract=off).
#include "gn_common.h"

namespace gn {
namespace GN_FPNS {

constexpr int QUERY_WAVES = 4;  // waves (= centres in flight) per workgroup

// NC centres per wave share every loaded 64-point chunk (the loads and address math are amortised;
// most chunks contain no hit for a given centre, so the slot bookkeeping is rarely executed).
template <bool CYL, int NC>
__global__ __launch_bounds__(QUERY_WAVES * 64) void query_kernel(
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, const float *__restrict__ rot,
    int n, int m, float radius, float hmin, float hmax, int nsample, int32_t *__restrict__ idx) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int j0 = (blockIdx.x * QUERY_WAVES + (threadIdx.x >> 6)) * NC;
  if (j0 >= m) return;  // wave-uniform
  xyz += (size_t)b * n * 3;
  const float radius2 = radius * radius;  // ball_query_gpu.cu:27
  float cx[NC], cy[NC], cz[NC], r[NC][9];
  int cnt[NC], first[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int j = min(j0 + c, m - 1);
    const float *cp = new_xyz + ((size_t)b * m + j) * 3;
    cx[c] = cp[0]; cy[c] = cp[1]; cz[c] = cp[2];
    if (CYL) {
      const float *rp = rot + ((size_t)b * m + j) * 9;
#pragma unroll
      for (int q = 0; q < 9; ++q) r[c][q] = rp[q];
    }
    cnt[c] = (j0 + c < m) ? 0 : nsample;   // centres past the end are born finished
    first[c] = 0;
  }
  constexpr int U = 2;  // 64-point chunks per iteration: their loads are issued together
  for (int k0 = 0; k0 < n; k0 += 64 * U) {
    bool done = true;
#pragma unroll
    for (int c = 0; c < NC; ++c) done = done && cnt[c] >= nsample;
    if (done) break;
    float px[U], py[U], pz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + u * 64 + lane, n - 1);
      px[u] = xyz[k * 3 + 0]; py[u] = xyz[k * 3 + 1]; pz[u] = xyz[k * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * 64 + lane;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        bool hit;
        if (CYL) {
          // cylinder_query_gpu.cu:36-46
          const float x = px[u] - cx[c], y = py[u] - cy[c], z = pz[u] - cz[c];
          const float x_rot = fp_dot3(r[c][0], x, r[c][3], y, r[c][6], z);
          const float y_rot = fp_dot3(r[c][1], x, r[c][4], y, r[c][7], z);
          const float z_rot = fp_dot3(r[c][2], x, r[c][5], y, r[c][8], z);
          const float d2 = fp_sumsq2(y_rot, z_rot);
          hit = d2 < radius2 && x_rot > hmin && x_rot < hmax;
        } else {
          // ball_query_gpu.cu:33-38
          const float d2 = fp_sumsq3(cx[c] - px[u], cy[c] - py[u], cz[c] - pz[u]);
          hit = d2 < radius2;
        }
        hit = hit && k < n;
        const unsigned long long mask = __ballot(hit);
        if (mask && cnt[c] < nsample) {
          int32_t *row = idx + ((size_t)b * m + j0 + c) * nsample;
          if (cnt[c] == 0) first[c] = k0 + u * 64 + __ffsll((long long)mask) - 1;
          const int slot = cnt[c] + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          if (hit && slot < nsample) row[slot] = k;
          cnt[c] += __popcll(mask);
        }
      }
    }
  }
  // unfilled slots repeat the first hit (ball_query_gpu.cu:39-43); no hit -> zeros
  // (the reference's torch::zeros output, ball_query.cpp:24-26)
#pragma unroll
  for (int c = 0; c < NC; ++c)
    if (j0 + c < m && cnt[c] < nsample) {
      int32_t *row = idx + ((size_t)b * m + j0 + c) * nsample;
      for (int l = cnt[c] + lane; l < nsample; l += 64) row[l] = first[c];
    }
}

// CloudCrop runs four cylinder queries that differ only in hmax (models/modules.py:103-106).
// One scan serves all ND depths: the radial / hmin tests are shared, each depth keeps its own
// "first nsample in index order" list, padding and zero-row rule.  idx: (B, M, ND, nsample).
template <int ND>
__global__ __launch_bounds__(QUERY_WAVES * 64) void cylinder_multi_kernel(
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, const float *__restrict__ rot, int n, int m,
    float radius, float hmin, float hmax0, float hmax1, float hmax2, float hmax3, int nsample,
    int32_t *__restrict__ idx) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * QUERY_WAVES + (threadIdx.x >> 6);
  if (j >= m) return;
  xyz += (size_t)b * n * 3;
  const float *c = new_xyz + ((size_t)b * m + j) * 3;
  const float *r = rot + ((size_t)b * m + j) * 9;
  int32_t *row = idx + ((size_t)b * m + j) * ND * nsample;
  const float radius2 = radius * radius;
  const float cx = c[0], cy = c[1], cz = c[2];
  const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6], r7 = r[7], r8 = r[8];
  const float hmax[4] = {hmax0, hmax1, hmax2, hmax3};
  int cnt[ND], first[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) { cnt[d] = 0; first[d] = 0; }
  constexpr int U = 4;
  for (int k0 = 0; k0 < n; k0 += 64 * U) {
    bool done = true;
#pragma unroll
    for (int d = 0; d < ND; ++d) done = done && (cnt[d] >= nsample);
    if (done) break;
    float px[U], py[U], pz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + u * 64 + lane, n - 1);
      px[u] = xyz[k * 3 + 0]; py[u] = xyz[k * 3 + 1]; pz[u] = xyz[k * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * 64 + lane;
      const float x = px[u] - cx, y = py[u] - cy, z = pz[u] - cz;
      const float x_rot = fp_dot3(r0, x, r3, y, r6, z);
      const float y_rot = fp_dot3(r1, x, r4, y, r7, z);
      const float z_rot = fp_dot3(r2, x, r5, y, r8, z);
      const float d2 = fp_sumsq2(y_rot, z_rot);
      const bool base = d2 < radius2 && x_rot > hmin && k < n;
      if (!__ballot(base)) continue;
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const bool hit = base && x_rot < hmax[d];
        const unsigned long long mask = __ballot(hit);
        if (mask && cnt[d] < nsample) {
          if (cnt[d] == 0) first[d] = k0 + u * 64 + __ffsll((long long)mask) - 1;
          const int slot = cnt[d] + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          if (hit && slot < nsample) row[d * nsample + slot] = k;
          cnt[d] += __popcll(mask);
        }
      }
    }
  }
#pragma unroll
  for (int d = 0; d < ND; ++d)
    if (cnt[d] < nsample)
      for (int l = cnt[d] + lane; l < nsample; l += 64) row[d * nsample + l] = first[d];
}

// Same scan as cylinder_multi_kernel, but instead of ND (mostly overlapping) index lists it emits,
// per seed, the UNION of the lists -- every cloud point that belongs to at least one depth's list,
// once, in index order -- with a ND-bit membership mask packed above the index:
//     entry = k | (mask << 28),   bit d of mask <=> k is in the depth-d list of the reference.
// A depth with no hit is the reference's all-zero row, i.e. 64 copies of point 0: its bit is put on an
// entry for point 0.  Padding slots (repeats of the first hit) add no new point.  The per-point MLP
// of CloudCrop depends only on (seed, point), so the fused crop kernel evaluates each entry once and
// max-pools it into the depths named by its mask -- bit-identical to pooling the ND full lists.
// uidx: (B, M, ND*nsample) i32, ucount: (B, M) i32.
template <int ND>
__global__ __launch_bounds__(QUERY_WAVES * 64) void cylinder_unique_kernel(
    const float *__restrict__ new_xyz, const float *__restrict__ xyz, const float *__restrict__ rot, int n, int m,
    float radius, float hmin, float hmax0, float hmax1, float hmax2, float hmax3, int nsample,
    int32_t *__restrict__ uidx, int32_t *__restrict__ ucount) {
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int j = blockIdx.x * QUERY_WAVES + (threadIdx.x >> 6);
  if (j >= m) return;
  xyz += (size_t)b * n * 3;
  const float *c = new_xyz + ((size_t)b * m + j) * 3;
  const float *r = rot + ((size_t)b * m + j) * 9;
  int32_t *row = uidx + ((size_t)b * m + j) * ND * nsample;
  const float radius2 = radius * radius;
  const float cx = c[0], cy = c[1], cz = c[2];
  const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6], r7 = r[7], r8 = r[8];
  const float hmax[4] = {hmax0, hmax1, hmax2, hmax3};
  int cnt[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) cnt[d] = 0;
  int ucnt = 0;
  int first_entry_k = -1;  // index held by entry 0 (entries are in index order)
  unsigned first_entry_mask = 0;
  constexpr int U = 4;
  for (int k0 = 0; k0 < n; k0 += 64 * U) {
    bool done = true;
#pragma unroll
    for (int d = 0; d < ND; ++d) done = done && (cnt[d] >= nsample);
    if (done) break;
    float px[U], py[U], pz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = min(k0 + u * 64 + lane, n - 1);
      px[u] = xyz[k * 3 + 0]; py[u] = xyz[k * 3 + 1]; pz[u] = xyz[k * 3 + 2];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + u * 64 + lane;
      const float x = px[u] - cx, y = py[u] - cy, z = pz[u] - cz;
      const float x_rot = fp_dot3(r0, x, r3, y, r6, z);
      const float y_rot = fp_dot3(r1, x, r4, y, r7, z);
      const float z_rot = fp_dot3(r2, x, r5, y, r8, z);
      const float d2 = fp_sumsq2(y_rot, z_rot);
      const bool base = d2 < radius2 && x_rot > hmin && k < n;
      if (!__ballot(base)) continue;
      unsigned mymask = 0;
#pragma unroll
      for (int d = 0; d < ND; ++d) {
        const bool hit = base && x_rot < hmax[d];
        const unsigned long long mask = __ballot(hit);
        if (mask && cnt[d] < nsample) {
          const int slot = cnt[d] + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
          if (hit && slot < nsample) mymask |= 1u << d;
          cnt[d] += __popcll(mask);
        }
      }
      const unsigned long long um = __ballot(mymask != 0);
      if (um) {
        if (ucnt == 0) {
          const int fl = __ffsll((long long)um) - 1;
          first_entry_k = k0 + u * 64 + fl;
          first_entry_mask = (unsigned)__builtin_amdgcn_readlane((int)mymask, fl);
        }
        const int slot = ucnt + __builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
        if (mymask) row[slot] = k | (int)(mymask << 28);   // at most ND*nsample entries by construction
        ucnt += __popcll(um);
      }
    }
  }
  // depths without any hit: the reference row is all zeros = point 0 repeated
  unsigned emask = 0;
#pragma unroll
  for (int d = 0; d < ND; ++d) emask |= (cnt[d] == 0 ? 1u : 0u) << d;
  if (emask) {
    if (first_entry_k == 0) {
      if (lane == 0) row[0] = 0 | (int)((first_entry_mask | emask) << 28);
    } else {
      if (lane == 0) row[ucnt] = 0 | (int)(emask << 28);
      ucnt += 1;
    }
  }
  if (lane == 0) ucount[(size_t)b * m + j] = ucnt;
}

template <bool CYL>
static int launch_query(const float *new_xyz, const float *xyz, const float *rot, int b, int n, int m,
                        float radius, float hmin, float hmax, int nsample, int32_t *idx,
                        gn_stream_t stream, const char *name) {
  GN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "%s: bad sizes", name);
  if (b == 0 || m == 0 || nsample == 0) return GN_OK;
  GN_REQUIRE(new_xyz && xyz && idx && (!CYL || rot), "%s: null pointer", name);
  GN_REQUIRE(b <= 65535, "%s: batch > 65535 unsupported", name);
  GN_REQUIRE((long)n * 3 < 0x7fffffffL, "%s: n too large", name);
  constexpr int NC = CYL ? 2 : 4;
  // few centres (deep SA levels): one centre per wave keeps the chip busy
  if ((long)b * m >= 8192) {
    dim3 grid((m + QUERY_WAVES * NC - 1) / (QUERY_WAVES * NC), b);
    hipLaunchKernelGGL((query_kernel<CYL, NC>), grid, dim3(QUERY_WAVES * 64), 0, as_stream(stream), new_xyz, xyz, rot,
                       n, m, radius, hmin, hmax, nsample, idx);
  } else {
    dim3 grid((m + QUERY_WAVES - 1) / QUERY_WAVES, b);
    hipLaunchKernelGGL((query_kernel<CYL, 1>), grid, dim3(QUERY_WAVES * 64), 0, as_stream(stream), new_xyz, xyz, rot, n,
                       m, radius, hmin, hmax, nsample, idx);
  }
  GN_CHECK_LAUNCH(name);
  return GN_OK;
}

}  // namespace GN_FPNS
}  // namespace gn
using namespace gn::GN_FPNS;

extern "C" int GN_FP_ENTRY(gn_ball_query)(const float *new_xyz, const float *xyz, int b, int n, int m, float radius,
                             int nsample, int32_t *idx, gn_stream_t stream) {
  GN_FP_DISPATCH(gn_ball_query, new_xyz, xyz, b, n, m, radius, nsample, idx, stream);
  return launch_query<false>(new_xyz, xyz, nullptr, b, n, m, radius, 0.f, 0.f, nsample, idx, stream,
                                 "ball_query");
}

extern "C" int GN_FP_ENTRY(gn_cylinder_query)(const float *new_xyz, const float *xyz, const float *rot, int b, int n,
                                 int m, float radius, float hmin, float hmax, int nsample, int32_t *idx,
                                 gn_stream_t stream) {
  GN_FP_DISPATCH(gn_cylinder_query, new_xyz, xyz, rot, b, n, m, radius, hmin, hmax, nsample, idx, stream);
  return launch_query<true>(new_xyz, xyz, rot, b, n, m, radius, hmin, hmax, nsample, idx, stream,
                                "cylinder_query");
}

extern "C" int GN_FP_ENTRY(gn_cylinder_query_multi)(const float *new_xyz, const float *xyz, const float *rot, int b, int n, int m,
                                       float radius, float hmin, const float *hmax_host, int ndepth, int nsample,
                                       int32_t *idx, gn_stream_t stream) {
  using namespace gn;
  GN_FP_DISPATCH(gn_cylinder_query_multi, new_xyz, xyz, rot, b, n, m, radius, hmin, hmax_host, ndepth, nsample, idx, stream);
  GN_REQUIRE(b >= 0 && n >= 0 && m >= 0 && nsample >= 0, "cylinder_query_multi: bad sizes");
  if (b == 0 || m == 0 || nsample == 0) return GN_OK;
  GN_REQUIRE(new_xyz && xyz && rot && idx && hmax_host, "cylinder_query_multi: null pointer");
  GN_REQUIRE(ndepth == 4, "cylinder_query_multi: only 4 depths supported (got %d)", ndepth);
  GN_REQUIRE(b <= 65535, "cylinder_query_multi: batch > 65535 unsupported");
  GN_REQUIRE((long)n * 3 < 0x7fffffffL, "cylinder_query_multi: n too large");
  dim3 grid((m + QUERY_WAVES - 1) / QUERY_WAVES, b);
  hipLaunchKernelGGL((cylinder_multi_kernel<4>), grid, dim3(QUERY_WAVES * 64), 0, as_stream(stream), new_xyz, xyz,
                     rot, n, m, radius, hmin, hmax_host[0], hmax_host[1], hmax_host[2], hmax_host[3], nsample, idx);
  GN_CHECK_LAUNCH("cylinder_query_multi");
  return GN_OK;
}

extern "C" int GN_FP_ENTRY(gn_cylinder_query_unique)(const float *new_xyz, const float *xyz, const float *rot, int b, int n, int m,
                                        float radius, float hmin, const float *hmax_host, int ndepth, int nsample,
                                        int32_t *uidx, int32_t *ucount, gn_stream_t stream) {
  using namespace gn;
  GN_FP_DISPATCH(gn_cylinder_query_unique, new_xyz, xyz, rot, b, n, m, radius, hmin, hmax_host, ndepth, nsample, uidx, ucount, stream);
  GN_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 1, "cylinder_query_unique: bad sizes");
  if (b == 0 || m == 0) return GN_OK;
  GN_REQUIRE(new_xyz && xyz && rot && uidx && ucount && hmax_host, "cylinder_query_unique: null pointer");
  GN_REQUIRE(ndepth == 4, "cylinder_query_unique: only 4 depths supported (got %d)", ndepth);
  GN_REQUIRE(b <= 65535, "cylinder_query_unique: batch > 65535 unsupported");
  GN_REQUIRE(n < (1 << 28), "cylinder_query_unique: n must be < 2^28 (mask is packed above the index)");
  dim3 grid((m + QUERY_WAVES - 1) / QUERY_WAVES, b);
  hipLaunchKernelGGL((cylinder_unique_kernel<4>), grid, dim3(QUERY_WAVES * 64), 0, as_stream(stream), new_xyz, xyz,
                     rot, n, m, radius, hmin, hmax_host[0], hmax_host[1], hmax_host[2], hmax_host[3], nsample, uidx,
                     ucount);
  GN_CHECK_LAUNCH("cylinder_query_unique");
  return GN_OK;
}
