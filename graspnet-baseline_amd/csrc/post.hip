// post.hip -- what every caller of the reference runs right before / after the network
// (SURVEY.md 8f "next" rows), on the device:
//
//   depth_to_cloud     utils/data_utils.py:18-43 create_point_cloud_from_depth_image (+ pixel pick)
//   voxel_down_sample  open3d PointCloud.voxel_down_sample as used by utils/collision_detector.py:32-35
//                      (un-vendored third party: restated from its documented behaviour -- voxel origin
//                      = min_bound - voxel/2, index = floor((p-origin)/voxel), output = mean of members;
//                      PARITY UNPINNED upstream; open3d leaves the output order unspecified, here it is the
//                      order of first appearance and the sums are taken in ascending member order, so the
//                      result is bit-identical to the sequential numpy restatement and from run to run)
//                      -- a whole batch of clouds per call, nothing read back
//   collision_detect   utils/collision_detector.py:69-129 ModelFreeCollisionDetector.detect (one scene for all grasp
//                      lists as the reference's object holds it, or scene b's own points for grasp list b)
//   grasp_nms          graspnetAPI GraspGroup.sort_by_score() / .nms() (grasp_nms.nms_grasp; un-vendored,
//                      PARITY UNPINNED): score-descending greedy suppression of grasps within a
//                      translation AND rotation-angle threshold of a kept one
//
// The reference does these in numpy / open3d on the host after a device->host copy of the grasps
// (test.py:88-96); its README calls the collision check the slow step (README.md:72).
#include <stdlib.h>

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
// voxel down-sample, a whole batch of clouds per call, no host read, DETERMINISTIC:
//   * which voxel a point falls in, a voxel's member count and its lowest member index are integer facts (CAS on the
//     64-bit key, integer atomics) -- the same whatever order the hardware retires them in;
//   * the output ORDER is the order of first appearance (voxels sorted by their lowest member index): a prefix scan over
//     the points' "I am my voxel's first member" flags, no ticket;
//   * a voxel's mean adds its members in ASCENDING index order (the ticket-filled member lists are put in order by the
//     sorted-segment machinery of grad_det.hip), float64 -- the order numpy's sequential np.add.at uses, so the means are
//     bit-identical to oracle/post_oracle.py and to themselves from run to run (float64 atomics in arrival order, the
//     round-3 form, were neither).
// workspace per call (b clouds of n points, S = table slots per cloud):
//   keys u64 [b][S] | first u32 [b][S]            <- one 0xff memset
//   cnt i32 [b][S] | cursor i32 [b][S] | sflag u32 [b][S] | hdr [b]  <- one zero memset
//   srank i32 [b][S] | pslot i32 [b][n] | vidx i32 [b][n] | off i32 [b][n+1] | list i32 [b][n] | skip u8 [b][n]
// Round 5 -- scenes of a whole camera frame (the reference's callers hand the detector the RAW masked cloud: 3-5 x 10^5
// points, ~100 per 1 cm voxel on a table top, where round 4 measured 2.8 ms per batch of 8: 0.87 + 0.33 ms of it putting member
// lists of 60-130 entries in ascending order, 0.63 + 0.25 ms in the two one-workgroup-per-cloud passes):
//   * ORDER ONLY WHERE IT MATTERS.  A float64 sum of float32 values does not depend on the order of its terms as long as every
//     partial sum is exact, i.e. as long as  (largest exponent) + 1 + ceil(log2 members) - (smallest exponent - 23) <= 53.
//     A voxel's coordinates on an axis lie in an interval known from its index, so the exponents are bounded from the interval
//     -- except where the interval holds zero, where the smallest exponent is measured (an integer atomic max per such
//     point, sflag).  The leader of a voxel decides `exact`, and seg_order() leaves exact voxels in ticket order: on camera
//     data that is every voxel (x = (col - cx) z / fx is never closer to zero than ~1e-4), and the result is STILL bit-identical
//     to the sequential restatement by construction -- voxels that fail the test are ordered as before.
//   * the min-bound and ranking passes run on (chunks x clouds) workgroups beyond 65 536 points (two-level reduce / scan).
// ------------------------------------------------------------------------------------------------
struct VoxelHeader {
  double minb[3];
  int pad[2];
};

__device__ __forceinline__ bool finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

// one workgroup per cloud: min bound over the valid, finite points
__global__ __launch_bounds__(1024) void voxel_minbound_kernel(const float *__restrict__ pts, const int32_t *__restrict__ n_valid,
                                                              int n, VoxelHeader *__restrict__ hdr, int32_t *__restrict__ status) {
  __shared__ float s[3][16];
  const int b = blockIdx.x;
  pts += (size_t)b * n * 3;
  const int nv = n_valid ? min(max(n_valid[b], 0), n) : n;
  float m[3] = {INFINITY, INFINITY, INFINITY};
  for (int i = threadIdx.x; i < nv; i += 1024) {
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    if (finite3(x, y, z)) { m[0] = fminf(m[0], x); m[1] = fminf(m[1], y); m[2] = fminf(m[2], z); }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int a = 0; a < 3; ++a) {
    const float v = -wave_max(-m[a]);
    if (lane == 0) s[a][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = s[threadIdx.x][0];
    for (int wv = 1; wv < 16; ++wv) v = fminf(v, s[threadIdx.x][wv]);
    hdr[b].minb[threadIdx.x] = (double)v;
    if (threadIdx.x == 0) status[b] = 0;
  }
}

// beyond 65 536 points: (chunks x clouds) workgroups reduce 16 384 points each into scratch, one small pass folds them
constexpr int VOX_CHUNK = 16384;
__global__ __launch_bounds__(1024) void voxel_minbound_part_kernel(const float *__restrict__ pts, const int32_t *__restrict__ n_valid,
                                                                   int n, int nchunk, float *__restrict__ part) {
  __shared__ float s[3][16];
  const int b = blockIdx.y, ck = blockIdx.x;
  pts += (size_t)b * n * 3;
  const int nv = n_valid ? min(max(n_valid[b], 0), n) : n;
  float m[3] = {INFINITY, INFINITY, INFINITY};
  const int hi = min(nv, (ck + 1) * VOX_CHUNK);
  for (int i = ck * VOX_CHUNK + threadIdx.x; i < hi; i += 1024) {
    const float x = pts[i * 3], y = pts[i * 3 + 1], z = pts[i * 3 + 2];
    if (finite3(x, y, z)) { m[0] = fminf(m[0], x); m[1] = fminf(m[1], y); m[2] = fminf(m[2], z); }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int a = 0; a < 3; ++a) {
    const float v = -wave_max(-m[a]);
    if (lane == 0) s[a][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float v = s[threadIdx.x][0];
    for (int wv = 1; wv < 16; ++wv) v = fminf(v, s[threadIdx.x][wv]);
    part[((size_t)b * nchunk + ck) * 3 + threadIdx.x] = v;
  }
}
__global__ __launch_bounds__(64) void voxel_minbound_fold_kernel(const float *__restrict__ part, int nchunk,
                                                                 VoxelHeader *__restrict__ hdr, int32_t *__restrict__ status) {
  const int b = blockIdx.x, lane = threadIdx.x;
  for (int a = 0; a < 3; ++a) {
    float m = INFINITY;
    for (int c = lane; c < nchunk; c += 64) m = fminf(m, part[((size_t)b * nchunk + c) * 3 + a]);
    m = -wave_max(-m);
    if (lane == 0) hdr[b].minb[a] = (double)m;
  }
  if (lane == 0) status[b] = 0;
}

__device__ __forceinline__ unsigned long long hash64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdULL; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ULL; k ^= k >> 33;
  return k;
}

constexpr unsigned long long VOXEL_EMPTY = ~0ULL;

// does voxel vi's interval [origin + vi voxel, + voxel) on an axis hold zero, or come within a thousandth of a voxel of it?
// (then the smallest exponent among its members cannot be bounded from the interval and is measured instead)
__device__ __forceinline__ bool voxel_holds_zero(double origin, long long vi, double voxel) {
  const double lo = origin + (double)vi * voxel, hi = lo + voxel;
  return lo <= voxel * 1e-3 && hi >= -voxel * 1e-3;
}
__device__ __forceinline__ int f32_biased_exp(double v) {       // biased exponent of v rounded to float32, subnormals as 1
  const unsigned be = (__float_as_uint((float)fabs(v)) >> 23) & 0xffu;
  return (int)max(be, 1u);
}

// one thread per point: its voxel's table slot; the slot's member count and lowest member index
__global__ __launch_bounds__(256) void voxel_insert_kernel(const float *__restrict__ pts, const int32_t *__restrict__ n_valid,
                                                           int n, double voxel, const VoxelHeader *__restrict__ hdr,
                                                           unsigned long long *__restrict__ keys, unsigned *__restrict__ first,
                                                           int *__restrict__ cnt, unsigned *__restrict__ sflag,
                                                           int32_t *__restrict__ pslot, unsigned table_mask,
                                                           int32_t *__restrict__ status) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= n) return;
  const size_t tb = (size_t)b * (table_mask + 1);
  pslot += (size_t)b * n;
  const int nv = n_valid ? min(max(n_valid[b], 0), n) : n;
  const float *p3 = pts + ((size_t)b * n + i) * 3;
  const float fx = p3[0], fy = p3[1], fz = p3[2];
  if (i >= nv || !finite3(fx, fy, fz)) { pslot[i] = -1; return; }     // padding / non-finite points belong to no voxel
  const double p[3] = {(double)fx, (double)fy, (double)fz};
  unsigned long long key = 0;
  unsigned small = 0;      // 255 - (smallest biased float32 exponent among this point's coordinates on axes whose voxel interval holds zero)
  for (int a = 0; a < 3; ++a) {
    const double origin = hdr[b].minb[a] - voxel * 0.5;
    const long long vi = (long long)floor((p[a] - origin) / voxel);
    if (vi < 0 || vi >= (1LL << 21)) { atomicExch(&status[b], 1); pslot[i] = -1; return; }
    key |= (unsigned long long)vi << (21 * a);
    if (voxel_holds_zero(origin, vi, voxel) && p[a] != 0.0) {
      const unsigned be = (__float_as_uint((float)p[a]) >> 23) & 0xffu;
      small = max(small, 255u - max(be, 1u));
    }
  }
  // RUNS: consecutive points of a camera frame fall into the same voxel ~10 at a time (row-major pixels), so only the first
  // lane of a run of equal keys goes to the table -- one CAS, one atomicMin (its own index: the run's lowest), one atomicAdd
  // of the run's length -- and hands the slot to the rest by a shuffle: ~10x fewer atomics on the same address
  const int lane = threadIdx.x & 63;
  const unsigned long long prev_key = __shfl_up(key, 1);
  const unsigned long long act = __ballot(true);                       // lanes that reached this point (the others returned)
  const bool prev_active = lane > 0 && ((act >> (lane - 1)) & 1ull);
  const bool head = !(prev_active && prev_key == key);
  const unsigned long long heads = __ballot(head);
  const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));      // heads at or below this lane
  const int head_lane = 63 - __builtin_clzll(below);
  // the run ends before the next head above this lane, or at the last active lane
  const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
  const int next_head = above ? __builtin_ctzll(above) : 64;
  const unsigned long long run_mask = act & ((next_head == 64 ? ~0ull : ((1ull << next_head) - 1ull)) & ~((1ull << head_lane) - 1ull));
  // (every active lane in [head_lane, next_head) has this key: a lane with another key would have been a head)
  const int run_len = __popcll(run_mask);
  unsigned run_small = small;
  for (int o = 1; o < 64; o <<= 1) {                                   // max of `small` over the run (runs are short: a few steps matter)
    const unsigned v = __shfl_down(run_small, o);
    const int src = lane + o;
    if (src < next_head && src < 64 && ((act >> src) & 1ull)) run_small = max(run_small, v);
  }
  int myslot = -1;
  if (head) {
    unsigned slot = (unsigned)hash64(key) & table_mask;
    for (unsigned probe = 0; probe <= table_mask; ++probe) {
      const unsigned long long prev = atomicCAS(&keys[tb + slot], VOXEL_EMPTY, key);
      if (prev == VOXEL_EMPTY || prev == key) {
        atomicMin(&first[tb + slot], (unsigned)i);
        atomicAdd(&cnt[tb + slot], run_len);
        if (run_small) atomicMax(&sflag[tb + slot], run_small);
        myslot = (int)slot;
        break;
      }
      slot = (slot + 1) & table_mask;
    }
    if (myslot < 0) atomicExch(&status[b], 2);
  }
  pslot[i] = __shfl(myslot, head_lane);
}

// one thread per point: lc[i] = its voxel's member count if the point is the voxel's lowest member (its "leader"), else 0.
// Wide on purpose: the two dependent random reads (slot -> first / cnt) are what the ranking pass below must NOT do on its
// one workgroup per cloud (done there they were 40 serial L2 round trips per wave: 57-70 us).
// The leader also decides whether its voxel's float64 sums are order-independent (every partial sum exact; see the header of
// this section) and leaves the verdict in sflag[slot] (1 = exact: seg_order() need not order the member list).
__global__ __launch_bounds__(256) void voxel_leader_kernel(const int32_t *__restrict__ pslot, const unsigned *__restrict__ first,
                                                           const int *__restrict__ cnt, const unsigned long long *__restrict__ keys,
                                                           unsigned *__restrict__ sflag, const VoxelHeader *__restrict__ hdr,
                                                           double voxel, int n, unsigned nslots, int32_t *__restrict__ lc) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (i >= n) return;
  const int slot = pslot[(size_t)b * n + i];
  int v = 0;
  if (slot >= 0 && first[(size_t)b * nslots + slot] == (unsigned)i) {
    const size_t s = (size_t)b * nslots + slot;
    v = cnt[s];
    const unsigned long long key = keys[s];
    const unsigned measured = sflag[s];
    int lg = 0;
    while ((1 << lg) < v) ++lg;                                   // ceil(log2 members)
    bool exact = true;
    for (int a = 0; a < 3; ++a) {
      const long long vi = (long long)((key >> (21 * a)) & ((1ULL << 21) - 1));
      const double origin = hdr[b].minb[a] - voxel * 0.5;
      const double lo = origin + (double)vi * voxel, hi = lo + voxel;
      const int emax = f32_biased_exp(fmax(fabs(lo), fabs(hi))) + 1;          // members may exceed the interval by a rounding
      int emin;
      if (voxel_holds_zero(origin, vi, voxel)) emin = measured ? 255 - (int)measured : emax;     // nothing small and non-zero seen
      else emin = f32_biased_exp(fmin(fabs(lo), fabs(hi))) - 1;
      // all members are multiples of 2^(emin - 23), every partial sum is below 2^(emax + 1 + lg): exact in 53 bits?
      exact = exact && (emax + 1 + lg) - (emin - 23) <= 53;
    }
    sflag[s] = exact ? 1u : 0u;
  }
  lc[(size_t)b * n + i] = v;
}

// one workgroup per cloud: voxels ranked by their lowest member index (a prefix count of the leader flags) and, in the
// same pass, the member-list offsets in that order (a prefix sum of the leaders' member counts), both from the lc[] stream.
// Every wave owns one contiguous run of the cloud: pass 1 adds up its leaders and their counts, ONE barrier publishes the
// 16 totals, pass 2 walks the run again handing out ranks (ballot + mbcnt) and offsets (a shuffle scan) behind the wave's base.
__global__ __launch_bounds__(1024) void voxel_rank_kernel(const int32_t *__restrict__ pslot, const int32_t *__restrict__ lc,
                                                          int n, unsigned nslots, int32_t *__restrict__ srank,
                                                          int32_t *__restrict__ off, int32_t *__restrict__ out_counts) {
  __shared__ int s_f[16], s_c[16];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  pslot += (size_t)b * n;
  lc += (size_t)b * n;
  srank += (size_t)b * nslots;
  off += (size_t)b * (n + 1);
  const int run = ((n + 16 * 64 - 1) / (16 * 64)) * 64;      // points per wave, a multiple of 64
  const int lo = wave * run, hi = min(n, lo + run);
  int facc = 0, cacc = 0;
  for (int i0 = lo; i0 < hi; i0 += 256) {                    // four coalesced loads in flight per lane
    int c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int i = i0 + u * 64 + lane; c[u] = i < hi ? lc[i] : 0; }
#pragma unroll
    for (int u = 0; u < 4; ++u) { facc += c[u] > 0 ? 1 : 0; cacc += c[u]; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { facc += __shfl_xor(facc, o); cacc += __shfl_xor(cacc, o); }
  if (lane == 0) { s_f[wave] = facc; s_c[wave] = cacc; }
  __syncthreads();
  int rf = 0, rc = 0, nv = 0, total = 0;
  for (int w = 0; w < 16; ++w) {
    if (w < wave) { rf += s_f[w]; rc += s_c[w]; }
    nv += s_f[w]; total += s_c[w];
  }
  for (int i0 = lo; i0 < hi; i0 += 256) {
    int c[4], sl[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int i = i0 + u * 64 + lane;
      c[u] = i < hi ? lc[i] : 0;
      sl[u] = i < hi ? pslot[i] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool leader = c[u] > 0;
      const unsigned long long m = __ballot(leader);
      int incl = c[u];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
      }
      if (leader) {
        const int pos = rf + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        srank[sl[u]] = pos;
        off[pos] = rc + incl - c[u];
      }
      rf += __popcll(m);
      rc += __shfl(incl, 63);
    }
  }
  for (int a = nv + t; a <= n; a += 1024) off[a] = total;
  if (t == 0) out_counts[b] = nv;
}

// The same ranking beyond 65 536 points, on (chunks x clouds) workgroups: chunk totals -> one small scan per cloud -> ranks and
// offsets behind the chunk's base (a chunk = VOX_CHUNK points = 16 waves x 1024).
__global__ __launch_bounds__(1024) void voxel_rank_part_kernel(const int32_t *__restrict__ lc, int n, int nchunk,
                                                               int32_t *__restrict__ part) {
  __shared__ int s_f[16], s_c[16];
  const int b = blockIdx.y, ck = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  lc += (size_t)b * n;
  int facc = 0, cacc = 0;
  const int hi = min(n, (ck + 1) * VOX_CHUNK);
  for (int i = ck * VOX_CHUNK + t; i < hi; i += 1024) {
    const int c = lc[i];
    facc += c > 0 ? 1 : 0;
    cacc += c;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { facc += __shfl_xor(facc, o); cacc += __shfl_xor(cacc, o); }
  if (lane == 0) { s_f[wave] = facc; s_c[wave] = cacc; }
  __syncthreads();
  if (t == 0) {
    int f = 0, c = 0;
    for (int w = 0; w < 16; ++w) { f += s_f[w]; c += s_c[w]; }
    part[(size_t)b * (nchunk * 2 + 2) + ck * 2] = f;
    part[(size_t)b * (nchunk * 2 + 2) + ck * 2 + 1] = c;
  }
}
// one wave per cloud: exclusive scan of the chunk totals in place; totals -> out_counts, part[2 nchunk ..]
__global__ __launch_bounds__(64) void voxel_rank_scan_kernel(int nchunk, int32_t *__restrict__ part, int32_t *__restrict__ out_counts) {
  const int b = blockIdx.x, lane = threadIdx.x;
  int32_t *p = part + (size_t)b * (nchunk * 2 + 2);
  int bf = 0, bc = 0;
  for (int c0 = 0; c0 < nchunk; c0 += 64) {
    const int c = c0 + lane;
    const int f = c < nchunk ? p[c * 2] : 0, k = c < nchunk ? p[c * 2 + 1] : 0;
    int fi = f, ki = k;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(fi, o), v = __shfl_up(ki, o);
      if (lane >= o) { fi += u; ki += v; }
    }
    if (c < nchunk) { p[c * 2] = bf + fi - f; p[c * 2 + 1] = bc + ki - k; }
    bf += __shfl(fi, 63);
    bc += __shfl(ki, 63);
  }
  if (lane == 0) { out_counts[b] = bf; p[nchunk * 2] = bc; }
}
__global__ __launch_bounds__(1024) void voxel_rank_apply_kernel(const int32_t *__restrict__ pslot, const int32_t *__restrict__ lc,
                                                                int n, int nchunk, unsigned nslots, const int32_t *__restrict__ part,
                                                                const int32_t *__restrict__ out_counts, int32_t *__restrict__ srank,
                                                                int32_t *__restrict__ off) {
  __shared__ int s_f[16], s_c[16];
  const int b = blockIdx.y, ck = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  pslot += (size_t)b * n;
  lc += (size_t)b * n;
  srank += (size_t)b * nslots;
  off += (size_t)b * (n + 1);
  const int lo = ck * VOX_CHUNK + wave * 1024, hi = min(n, lo + 1024);        // this wave's run of the chunk
  int facc = 0, cacc = 0;
  for (int i = lo + lane; i < hi; i += 64) {
    const int c = lc[i];
    facc += c > 0 ? 1 : 0;
    cacc += c;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { facc += __shfl_xor(facc, o); cacc += __shfl_xor(cacc, o); }
  if (lane == 0) { s_f[wave] = facc; s_c[wave] = cacc; }
  __syncthreads();
  int rf = part[(size_t)b * (nchunk * 2 + 2) + ck * 2], rc = part[(size_t)b * (nchunk * 2 + 2) + ck * 2 + 1];
  for (int w = 0; w < wave; ++w) { rf += s_f[w]; rc += s_c[w]; }
  for (int i0 = lo; i0 < hi; i0 += 64) {
    const int i = i0 + lane;
    const int c = i < hi ? lc[i] : 0;
    const int sl = i < hi ? pslot[i] : -1;
    const bool leader = c > 0;
    const unsigned long long m = __ballot(leader);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (leader) {
      const int pos = rf + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      srank[sl] = pos;
      off[pos] = rc + incl - c;
    }
    rf += __popcll(m);
    rc += __shfl(incl, 63);
  }
  // off[a] = total members for a in [voxels, n]: every chunk fills the part of that tail that lies in its own index range
  const int nv = out_counts[b], total = part[(size_t)b * (nchunk * 2 + 2) + nchunk * 2];
  for (int a = ck * VOX_CHUNK + t; a < min(n + 1, (ck + 1) * VOX_CHUNK); a += 1024)
    if (a >= nv) off[a] = total;
  if (ck == nchunk - 1 && t == 0) off[n] = total;
}

// member lists: every point takes a ticket in its voxel's segment (arbitrary order; seg_order() sorts the segments)
__global__ __launch_bounds__(256) void voxel_fill_kernel(const int32_t *__restrict__ pslot, const int32_t *__restrict__ srank,
                                                         const int32_t *__restrict__ off, int *__restrict__ cursor,
                                                         const unsigned *__restrict__ first, const unsigned *__restrict__ sflag, int n,
                                                         unsigned nslots, int32_t *__restrict__ vidx,
                                                         int32_t *__restrict__ list, uint8_t *__restrict__ skip) {
  const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int slot = i < n ? pslot[(size_t)b * n + i] : -1;
  // runs of consecutive points in one voxel take their tickets together: the run's first lane adds the run's length
  const int prev_slot = __shfl_up(slot, 1);
  const bool head = slot >= 0 && !(lane > 0 && prev_slot == slot);
  const unsigned long long heads = __ballot(head), mine = __ballot(slot >= 0);
  int v = -1;
  if (slot >= 0) {
    const unsigned long long below = heads & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
    const int head_lane = 63 - __builtin_clzll(below);
    const unsigned long long above = lane == 63 ? 0ull : (heads >> (lane + 1)) << (lane + 1);
    // the run also ends at the first lane without a voxel
    const unsigned long long gaps = lane == 63 ? 0ull : ((~mine) >> (lane + 1)) << (lane + 1);
    const unsigned long long stop = above | gaps;
    const int end = stop ? __builtin_ctzll(stop) : 64;
    int base = 0;
    if (head) base = atomicAdd(&cursor[(size_t)b * nslots + slot], end - lane);
    base = __shfl(base, head_lane);
    v = srank[(size_t)b * nslots + slot];
    const int pos = off[(size_t)b * (n + 1) + v] + base + (lane - head_lane);
    list[(size_t)b * n + pos] = i;
    if (first[(size_t)b * nslots + slot] == (unsigned)i) skip[(size_t)b * n + v] = (uint8_t)sflag[(size_t)b * nslots + slot];
  }
  if (i < n) vidx[(size_t)b * n + i] = v;
}

// one thread per output row: the voxel's members added in ascending index order, float64, then the mean; rows past
// the cloud's voxel count are zero
__global__ __launch_bounds__(256) void voxel_mean_kernel(const float *__restrict__ pts, const int32_t *__restrict__ off,
                                                         const int32_t *__restrict__ list, const int32_t *__restrict__ counts,
                                                         int n, double *__restrict__ out) {
  const int a = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (a >= n) return;
  double sx = 0.0, sy = 0.0, sz = 0.0;
  if (a < counts[b]) {
    const int lo = off[(size_t)b * (n + 1) + a], hi = off[(size_t)b * (n + 1) + a + 1];
    const float *p = pts + (size_t)b * n * 3;
    const int32_t *l = list + (size_t)b * n;
    for (int t = lo; t < hi; ++t) {
      const int e = l[t];
      sx += (double)p[e * 3]; sy += (double)p[e * 3 + 1]; sz += (double)p[e * 3 + 2];
    }
    const double c = (double)(hi - lo);
    sx /= c; sy /= c; sz /= c;
  }
  double *o = out + ((size_t)b * n + a) * 3;
  o[0] = sx; o[1] = sy; o[2] = sz;
}

// the same for frame-sized clouds, one WAVE per 64 output rows: a row with few members (<= 8: a sparse scene, ~2 points per
// voxel) is summed by its own lane, sequentially in list order (ascending where the order matters: seg_order() has put those
// lists in order); a row with many members (a camera frame's table top: ~100 per voxel) is taken by the whole wave -- flagged
// exact (skip != 0: its float64 sums are the same in any order and any association) lane-strided and folded by shuffles,
// otherwise by its owner lane alone.  Rows past the cloud's voxel count are zero.
__global__ __launch_bounds__(256) void voxel_mean_wave_kernel(const float *__restrict__ pts, const int32_t *__restrict__ off,
                                                              const int32_t *__restrict__ list, const int32_t *__restrict__ counts,
                                                              const uint8_t *__restrict__ skip, int n, double *__restrict__ out) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int nv = counts[b];
  const float *p = pts + (size_t)b * n * 3;
  const int32_t *l = list + (size_t)b * n;
  const int a0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
  if (a0 >= n) return;
  const int a = a0 + lane;
  double sx = 0.0, sy = 0.0, sz = 0.0;
  int lo = 0, hi = 0;
  if (a < nv) { lo = off[(size_t)b * (n + 1) + a]; hi = off[(size_t)b * (n + 1) + a + 1]; }
  const int cnt = hi - lo;
  const bool big = cnt > 8 && skip[(size_t)b * n + a] != 0;      // order-free and long: worth the whole wave
  if (!big) {
    for (int t = lo; t < hi; ++t) {
      const int e = l[t];
      sx += (double)p[e * 3]; sy += (double)p[e * 3 + 1]; sz += (double)p[e * 3 + 2];
    }
  }
  unsigned long long todo = __ballot(big);
  while (todo) {
    const int src = __ffsll((long long)todo) - 1;
    todo &= todo - 1;
    const int slo = __shfl(lo, src), shi = __shfl(hi, src);
    double tx = 0.0, ty = 0.0, tz = 0.0;
    for (int t = slo + lane; t < shi; t += 64) {
      const int e = l[t];
      tx += (double)p[e * 3]; ty += (double)p[e * 3 + 1]; tz += (double)p[e * 3 + 2];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { tx += __shfl_xor(tx, o); ty += __shfl_xor(ty, o); tz += __shfl_xor(tz, o); }
    if (lane == src) { sx = tx; sy = ty; sz = tz; }
  }
  if (a < n) {
    if (cnt > 0) { const double c = (double)cnt; sx /= c; sy /= c; sz /= c; }
    double *o = out + ((size_t)b * n + a) * 3;
    o[0] = sx; o[1] = sy; o[2] = sz;
  }
}

static unsigned voxel_slots(int n) {
  unsigned s = 1024;
  while (s < 2u * (unsigned)n) s <<= 1;
  return s;
}

struct VoxelWs {
  unsigned long long *keys;
  unsigned *first;
  int *cnt, *cursor;
  unsigned *sflag;
  VoxelHeader *hdr;
  int32_t *srank, *pslot, *vidx, *off, *list, *part;
  uint8_t *skip;
  size_t ff_bytes, zero_bytes, bytes;
};

static VoxelWs voxel_carve(void *base, int b, int n) {
  const size_t S = voxel_slots(n), B = (size_t)b, N = (size_t)n;
  char *p = static_cast<char *>(base);
  size_t o = 0;
  auto take = [&](size_t bytes) { char *q = p ? p + o : nullptr; o += (bytes + 255) & ~(size_t)255; return q; };
  VoxelWs w;
  w.keys = (unsigned long long *)take(B * S * 8 + B * S * 4);
  w.first = p ? (unsigned *)(w.keys + B * S) : nullptr;
  w.ff_bytes = B * S * 12;
  const size_t z0 = o;
  w.cnt = (int *)take(B * S * 4);
  w.cursor = (int *)take(B * S * 4);
  w.sflag = (unsigned *)take(B * S * 4);
  w.hdr = (VoxelHeader *)take(B * sizeof(VoxelHeader));
  w.zero_bytes = o - z0;
  w.srank = (int32_t *)take(B * S * 4);
  w.pslot = (int32_t *)take(B * N * 4);
  w.vidx = (int32_t *)take(B * N * 4);
  w.off = (int32_t *)take(B * (N + 1) * 4);
  w.list = (int32_t *)take(B * N * 4);
  w.skip = (uint8_t *)take(B * N);
  w.part = (int32_t *)take(B * ((N + VOX_CHUNK - 1) / VOX_CHUNK * 3 + 2) * 4);      // chunk minima (3 floats) / chunk totals (2 ints + 2)
  w.bytes = o;
  return w;
}

// ------------------------------------------------------------------------------------------------
// collision detection: one workgroup per grasp, float64 like numpy on open3d's float64 points
// ------------------------------------------------------------------------------------------------
struct CollisionParams {
  double voxel_size;            // a python float in the reference: voxel_size ** 3 is taken in double, then meets float32 arrays
  float finger_width, finger_length;
  float approach_dist;          // already max(approach_dist, finger_width)  (collision_detector.py:69)
  double collision_thresh, empty_thresh;
};

// One workgroup = COL_GB grasps of one scene against ALL of the scene's points.
//   * the counts the reference takes are tiny (a gripper's boxes hold ~5-50 of a scene's ~10^4 voxel points), so every
//     (point, grasp) pair first meets a CONSERVATIVE float32 test against the union of the boxes in the grasp frame, widened
//     by eps (the float32 evaluation's error bound for a point that truly lies inside: its |p| <= |T| + the box extent, so
//     |t_f32 - t| <= ~(|T|_1 + 1) * 2^-22; eps is 4x that); only pairs that pass are evaluated in float64 exactly as numpy
//     does (collision_detector.py:80-129) and bump the grasp's six LDS counters.  A pair that fails is outside every box in
//     exact arithmetic too, so the counts -- integers -- are the reference's;
//   * a thread keeps COL_K points in registers per tile and walks the workgroup's grasps over them (per-grasp constants
//     from LDS, broadcast), so the scene is read once per workgroup instead of once per grasp (the round-3 kernel moved
//     2.2 GB through L2 per batch and did the float64 transform for every pair).
constexpr int COL_GB = 8, COL_K = 4;

struct ColGrasp {            // float32 pre-test constants of one grasp, 16 floats
  float r[9];                // R row-major: targets = (p - T) @ R
  float t[3];
  float xlo, xhi, yb, zb;    // union of the four boxes + the inner box, widened by eps
};

struct ColGraspD {           // the exact test's constants: float32 expressions of numpy, widened to float64 once per grasp
  double R[9], T[3];
  double zlo, zhi, x_fl, x_d, y3, y4, y5, y6, x_b, x_s;
};

// exact evaluation of one (point, grasp) pair, float64 as numpy does it (collision_detector.py:80-112)
__device__ __forceinline__ void collision_exact(const double *__restrict__ sp, const ColGraspD &G, int *cnt) {
  const double dx = sp[0] - G.T[0], dy = sp[1] - G.T[1], dz = sp[2] - G.T[2];
  // targets = (scene - T) @ R  (:80-81)
  const double tx = dx * G.R[0] + dy * G.R[3] + dz * G.R[6];
  const double ty = dx * G.R[1] + dy * G.R[4] + dz * G.R[7];
  const double tz = dx * G.R[2] + dy * G.R[5] + dz * G.R[8];
  const bool m1 = tz > G.zlo && tz < G.zhi;
  const bool m2 = tx > G.x_fl && tx < G.x_d;
  const bool m3 = ty > G.y3, m4 = ty < G.y4, m5 = ty < G.y5, m6 = ty > G.y6;
  const bool m7 = tx <= G.x_fl && tx > G.x_b;
  const bool m8 = tx <= G.x_b && tx > G.x_s;
  const bool left = m1 && m2 && m3 && m4;
  const bool right = m1 && m2 && m5 && m6;
  const bool bottom = m1 && m3 && m5 && m7;
  const bool shift = m1 && m3 && m5 && m8;
  if (left || right || bottom || shift) atomicAdd(&cnt[0], 1);
  if (left) atomicAdd(&cnt[1], 1);
  if (right) atomicAdd(&cnt[2], 1);
  if (bottom) atomicAdd(&cnt[3], 1);
  if (shift) atomicAdd(&cnt[4], 1);
  if (m1 && m2 && !m4 && !m6) atomicAdd(&cnt[5], 1);
}

__global__ __launch_bounds__(256) void collision_kernel(const double *__restrict__ scene, size_t scene_stride,
                                                        const int *__restrict__ np_ptr, int np_stride,
                                                        int np_fixed, const float *__restrict__ grasps, int ldg,
                                                        const int32_t *__restrict__ counts, int mpad,
                                                        CollisionParams P, uint8_t *__restrict__ collision_mask,
                                                        uint8_t *__restrict__ empty_mask, float *__restrict__ ious) {
  __shared__ int s_cnt[COL_GB][8];
  __shared__ ColGrasp s_g[COL_GB];
  __shared__ ColGraspD s_gd[COL_GB];
  __shared__ int s_queue[4][128];      // per wave: (point index << 3 | grasp slot) of the pairs that passed the pre-test
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int g0 = blockIdx.x * COL_GB;
  const int m_valid = counts ? min(counts[b], mpad) : mpad;
  const int ng = min(COL_GB, m_valid - g0);          // grasps of this workgroup that exist; the rest are padding rows
  if (tid < COL_GB && g0 + tid < mpad && tid >= max(ng, 0)) {
    const size_t orow = (size_t)b * mpad + g0 + tid;
    collision_mask[orow] = 0;
    if (empty_mask) empty_mask[orow] = 0;
    if (ious) for (int k = 0; k < 5; ++k) ious[orow * 5 + k] = 0.f;
  }
  if (ng <= 0) return;
  // scene b's own down-sampled cloud (scene_stride doubles apart; 0 = one scene for the whole batch) and point count
  scene += (size_t)b * scene_stride;
  const int np = np_ptr ? min(np_ptr[(size_t)b * np_stride], np_fixed) : np_fixed;
  const float *grow = grasps + ((size_t)b * mpad + g0) * ldg;
  if (tid < COL_GB * 8) s_cnt[tid >> 3][tid & 7] = 0;
  if (tid < ng) {
    const float *gr = grow + (size_t)tid * ldg;
    ColGrasp c;
    ColGraspD G;
    for (int i = 0; i < 9; ++i) { c.r[i] = gr[4 + i]; G.R[i] = (double)gr[4 + i]; }
    for (int i = 0; i < 3; ++i) { c.t[i] = gr[13 + i]; G.T[i] = (double)gr[13 + i]; }
    const float width = gr[1], height = gr[2], depth = gr[3];
    // thresholds are float32 expressions in numpy (float32 arrays with python scalars), :84-99
    G.zlo = (double)(-height / 2.0f); G.zhi = (double)(height / 2.0f);
    G.x_fl = (double)(depth - P.finger_length);
    G.x_d = (double)depth;
    G.y3 = (double)(-(width / 2.0f + P.finger_width));
    G.y4 = (double)(-width / 2.0f);
    G.y5 = (double)(width / 2.0f + P.finger_width);
    G.y6 = (double)(width / 2.0f);
    G.x_b = (double)(depth - P.finger_length - P.finger_width);
    G.x_s = (double)(depth - P.finger_length - P.finger_width - P.approach_dist);
    const float eps = 2e-6f * (fabsf(c.t[0]) + fabsf(c.t[1]) + fabsf(c.t[2]) + 1.0f);
    c.xlo = depth - P.finger_length - P.finger_width - P.approach_dist - eps;
    c.xhi = depth + eps;
    c.yb = fabsf(width) * 0.5f + P.finger_width + eps;
    c.zb = fabsf(height) * 0.5f + eps;
    s_g[tid] = c;
    s_gd[tid] = G;
  }
  __syncthreads();
  int *queue = s_queue[wv];
  int qn = 0;                        // wave-uniform: pairs waiting in the queue
  // the pairs that pass are rare (tens of a scene's ~10^4 points per grasp) and scattered over the lanes: they are queued
  // and evaluated 64 at a time, one per lane, instead of dragging the wave through the float64 path for one lane each
  auto drain = [&](int take) {
    if (lane < take) {
      const int e = queue[lane];
      collision_exact(scene + (size_t)(e >> 3) * 3, s_gd[e & 7], s_cnt[e & 7]);
    }
    const int rest = qn - take;
    const int moved = lane < rest ? queue[take + lane] : 0;
    if (lane < rest) queue[lane] = moved;     // rest < 64: a queue never holds more than 64 + 63 pairs
    qn = rest;
  };
  for (int base = 0; base < np; base += 256 * COL_K) {
    float px[COL_K], py[COL_K], pz[COL_K];
#pragma unroll
    for (int k = 0; k < COL_K; ++k) {
      const int i = base + k * 256 + tid;
      if (i < np) {
        px[k] = (float)scene[(size_t)i * 3]; py[k] = (float)scene[(size_t)i * 3 + 1]; pz[k] = (float)scene[(size_t)i * 3 + 2];
      } else {
        px[k] = py[k] = pz[k] = INFINITY;       // fails every test (Inf - T = Inf, Inf * 0 = NaN: all compares false)
      }
    }
    for (int g = 0; g < ng; ++g) {
      const ColGrasp c = s_g[g];
#pragma unroll
      for (int k = 0; k < COL_K; ++k) {
        const float dx = px[k] - c.t[0], dy = py[k] - c.t[1], dz = pz[k] - c.t[2];
        const float tx = fmaf(dz, c.r[6], fmaf(dy, c.r[3], dx * c.r[0]));
        const float ty = fmaf(dz, c.r[7], fmaf(dy, c.r[4], dx * c.r[1]));
        const float tz = fmaf(dz, c.r[8], fmaf(dy, c.r[5], dx * c.r[2]));
        const bool pass = fabsf(tz) < c.zb && fabsf(ty) < c.yb && tx > c.xlo && tx < c.xhi;
        const unsigned long long m = __ballot(pass);
        if (m) {                       // wave-uniform
          if (pass)
            queue[qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] =
                ((base + k * 256 + tid) << 3) | g;
          qn += __popcll(m);
          if (qn >= 64) drain(64);
        }
      }
    }
  }
  if (qn > 0) drain(qn);
  __syncthreads();
  if (tid < ng) {
    const float *gr = grow + (size_t)tid * ldg;
    const float width = gr[1], height = gr[2];
    const int *cnt = s_cnt[tid];
    const size_t orow = (size_t)b * mpad + g0 + tid;
    const float vs3 = (float)(P.voxel_size * P.voxel_size * P.voxel_size);  // python float ** 3, then float32 / python float
    // float32 volume expressions, evaluated left to right (:113-116)
    const float lr_vol = height * P.finger_length * P.finger_width / vs3;
    const float bottom_vol = height * (width + 2.0f * P.finger_width) * P.finger_width / vs3;
    const float shift_vol = height * (width + 2.0f * P.finger_width) * P.approach_dist / vs3;
    const float volume = lr_vol * 2.0f + bottom_vol + shift_vol;
    const double giou = (double)cnt[0] / (double)(volume + 1e-6f);     // int64 / float32 -> float64 (:119)
    collision_mask[orow] = giou > P.collision_thresh;
    if (empty_mask) {
      const float inner_vol = height * P.finger_length * width / vs3;
      empty_mask[orow] = ((double)cnt[5] / (double)inner_vol) < P.empty_thresh;
    }
    if (ious) {
      float *o = ious + orow * 5;
      o[0] = (float)giou;
      o[1] = (float)((double)cnt[1] / (double)(lr_vol + 1e-6f));
      o[2] = (float)((double)cnt[2] / (double)(lr_vol + 1e-6f));
      o[3] = (float)((double)cnt[3] / (double)(bottom_vol + 1e-6f));
      o[4] = (float)((double)cnt[4] / (double)(shift_vol + 1e-6f));
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
                                                         double cos_rt, int do_nms, int P /* power of two, mpad <= P <= NMS_MAX */,
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
  // the sort runs over P slots, the smallest power of two that holds the padded list (1024 for the model's 1024 seeds:
  // 55 compare-exchange passes over 1024 keys instead of 66 over 2048)
  for (int i = tid; i < P; i += 1024) {
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
  // ---- bitonic sort of P keys
  for (int k = 2; k <= P; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < P; i += 1024) {
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
  for (int i = tid; i < P; i += 1024)
    if (s_key[i] != ~0ULL && (i + 1 == P || s_key[i + 1] == ~0ULL)) s_n = i + 1;
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

extern "C" size_t gn_voxel_batched_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 0;
  return voxel_carve(nullptr, b, n).bytes;
}

extern "C" int gn_voxel_down_sample_batched(const float *points, const int32_t *n_valid, int b, int n, double voxel_size,
                                            void *workspace, double *out, int32_t *out_counts, int32_t *status,
                                            gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n >= 0 && voxel_size > 0.0, "voxel_down_sample: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(out_counts && status, "voxel_down_sample: null pointer");
  GN_REQUIRE(b <= 65535, "voxel_down_sample: batch > 65535 unsupported");
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    if (hipMemsetAsync(out_counts, 0, sizeof(int32_t) * (size_t)b, s) != hipSuccess ||
        hipMemsetAsync(status, 0, sizeof(int32_t) * (size_t)b, s) != hipSuccess) {
      set_error("voxel_down_sample: memset failed");
      return GN_ERR_LAUNCH;
    }
    return GN_OK;
  }
  GN_REQUIRE(points && workspace && out, "voxel_down_sample: null pointer");
  GN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "voxel_down_sample: workspace must be 256-byte aligned");
  const VoxelWs w = voxel_carve(workspace, b, n);
  const unsigned slots = voxel_slots(n);
  if (hipMemsetAsync(w.keys, 0xff, w.ff_bytes, s) != hipSuccess ||
      hipMemsetAsync(w.cnt, 0, w.zero_bytes, s) != hipSuccess) {
    set_error("voxel_down_sample: memset failed");
    return GN_ERR_LAUNCH;
  }
  const dim3 gp((n + 255) / 256, b);
  const int nchunk = (n + VOX_CHUNK - 1) / VOX_CHUNK;
  const bool wide = n > 65536;          // a whole frame's cloud: the per-cloud passes on (chunks x clouds) workgroups
  if (wide) {
    hipLaunchKernelGGL(voxel_minbound_part_kernel, dim3(nchunk, b), dim3(1024), 0, s, points, n_valid, n, nchunk, (float *)w.part);
    hipLaunchKernelGGL(voxel_minbound_fold_kernel, dim3(b), dim3(64), 0, s, (const float *)w.part, nchunk, w.hdr, status);
  } else {
    hipLaunchKernelGGL(voxel_minbound_kernel, dim3(b), dim3(1024), 0, s, points, n_valid, n, w.hdr, status);
  }
  hipLaunchKernelGGL(voxel_insert_kernel, gp, dim3(256), 0, s, points, n_valid, n, voxel_size, w.hdr, w.keys, w.first, w.cnt,
                     w.sflag, w.pslot, slots - 1, status);
  // (lc lives in w.vidx until the fill pass overwrites it with the points' voxel ranks)
  hipLaunchKernelGGL(voxel_leader_kernel, gp, dim3(256), 0, s, w.pslot, w.first, w.cnt, w.keys, w.sflag, w.hdr, voxel_size, n,
                     slots, w.vidx);
  if (wide) {
    hipLaunchKernelGGL(voxel_rank_part_kernel, dim3(nchunk, b), dim3(1024), 0, s, w.vidx, n, nchunk, w.part);
    hipLaunchKernelGGL(voxel_rank_scan_kernel, dim3(b), dim3(64), 0, s, nchunk, w.part, out_counts);
    hipLaunchKernelGGL(voxel_rank_apply_kernel, dim3(nchunk, b), dim3(1024), 0, s, w.pslot, w.vidx, n, nchunk, slots, w.part,
                       out_counts, w.srank, w.off);
  } else {
    hipLaunchKernelGGL(voxel_rank_kernel, dim3(b), dim3(1024), 0, s, w.pslot, w.vidx, n, slots, w.srank, w.off, out_counts);
  }
  hipLaunchKernelGGL(voxel_fill_kernel, gp, dim3(256), 0, s, w.pslot, w.srank, w.off, w.cursor, w.first, w.sflag, n, slots, w.vidx,
                     w.list, w.skip);
  GN_CHECK_LAUNCH("voxel_down_sample");
  const char *oe = getenv("GN_VOXEL_ORDER");          // GN_VOXEL_ORDER=all: order every member list (round 4's form; A/B, tests)
  if (int rc = seg_order(w.vidx, b, n, n, w.off, w.list, s, (oe && oe[0] == 'a') ? nullptr : w.skip)) return rc;
  if (wide && !(oe && oe[0] == 'a'))
    hipLaunchKernelGGL(voxel_mean_wave_kernel, dim3((n + 255) / 256, b), dim3(256), 0, s, points, w.off, w.list, out_counts, w.skip, n, out);
  else
    hipLaunchKernelGGL(voxel_mean_kernel, gp, dim3(256), 0, s, points, w.off, w.list, out_counts, n, out);
  GN_CHECK_LAUNCH("voxel_down_sample");
  return GN_OK;
}

// the one-cloud form: workspace = [status (256 B) | the batched workspace]
extern "C" size_t gn_voxel_workspace_bytes(int n) {
  if (n <= 0) return 0;
  return 256 + gn_voxel_batched_workspace_bytes(1, n);
}

extern "C" int gn_voxel_down_sample(const float *points, int n, double voxel_size, void *workspace, double *out,
                                    int32_t *out_count, gn_stream_t stream) {
  GN_REQUIRE(n >= 0 && voxel_size > 0.0, "voxel_down_sample: bad sizes");
  GN_REQUIRE(points && workspace && out && out_count, "voxel_down_sample: null pointer");
  return gn_voxel_down_sample_batched(points, nullptr, 1, n, voxel_size, static_cast<char *>(workspace) + 256, out, out_count,
                                      static_cast<int32_t *>(workspace), stream);
}

static int collision_launch(const double *scene_points, size_t scene_stride, const int32_t *np_dev, int np_stride, int np,
                            const float *grasps, const int32_t *counts, int b, int mpad, float approach_dist,
                            double collision_thresh, double empty_thresh, double voxel_size, float finger_width,
                            float finger_length, uint8_t *collision_mask, uint8_t *empty_mask, float *ious,
                            gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && mpad >= 0 && np >= 0, "collision_detect: bad sizes");
  GN_REQUIRE(np < (1 << 28) || np == 0x7fffffff, "collision_detect: more than 2^28 scene points unsupported");
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
  hipLaunchKernelGGL(collision_kernel, dim3((mpad + COL_GB - 1) / COL_GB, b), dim3(256), 0, as_stream(stream), scene_points, scene_stride, np_dev,
                     np_stride, np, grasps, 17, counts, mpad, P, collision_mask, empty_mask, ious);
  GN_CHECK_LAUNCH("collision_detect");
  return GN_OK;
}

extern "C" int gn_collision_detect(const double *scene_points, const int32_t *np_dev, int np, const float *grasps,
                                   const int32_t *counts, int b, int mpad, float approach_dist, double collision_thresh,
                                   double empty_thresh, double voxel_size, float finger_width, float finger_length,
                                   uint8_t *collision_mask, uint8_t *empty_mask, float *ious, gn_stream_t stream) {
  // np_dev (one device int, e.g. gn_voxel_down_sample's out_count) bounds the scene when given; np is then its capacity
  GN_REQUIRE(np_dev || np >= 0, "collision_detect: bad sizes");
  return collision_launch(scene_points, 0, np_dev, 0, np_dev && np <= 0 ? 0x7fffffff : np, grasps, counts, b, mpad,
                          approach_dist, collision_thresh, empty_thresh, voxel_size, finger_width, finger_length,
                          collision_mask, empty_mask, ious, stream);
}

extern "C" int gn_collision_detect_batched(const double *scene_points, int np_max, const int32_t *np_dev, const float *grasps,
                                           const int32_t *counts, int b, int mpad, float approach_dist,
                                           double collision_thresh, double empty_thresh, double voxel_size, float finger_width,
                                           float finger_length, uint8_t *collision_mask, uint8_t *empty_mask, float *ious,
                                           gn_stream_t stream) {
  return collision_launch(scene_points, (size_t)(np_max > 0 ? np_max : 0) * 3, np_dev, 1, np_max, grasps, counts, b, mpad,
                          approach_dist, collision_thresh, empty_thresh, voxel_size, finger_width, finger_length,
                          collision_mask, empty_mask, ious, stream);
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
  int P = 64;
  while (P < mpad) P <<= 1;
  hipLaunchKernelGGL(grasp_nms_kernel, dim3(b), dim3(1024), 0, as_stream(stream), grasps, counts, drop_mask, mpad,
                     translation_thresh * translation_thresh, cos(rotation_thresh), do_nms, P,
                     (unsigned long long *)workspace, out, out_counts);
  GN_CHECK_LAUNCH("grasp_nms");
  return GN_OK;
}

// ------------------------------------------------------------------------------------------------
// the whole post-processing of a batch in ONE call (what test.py:88-96 / demo_pointcloud.py:228-249 run on the host per
// scene after pred_decode): voxel down-sample of every scene's cloud -> collision check of scene b's grasps against scene
// b's points -> collisions dropped, sort by score (+ NMS).  One workspace, ~13 launches on the caller's stream, nothing
// read back, hipGraph-capturable.
// ------------------------------------------------------------------------------------------------
namespace gn {
struct PostWs {
  void *voxel;
  double *scene;
  int32_t *scene_counts, *status;
  uint8_t *cmask;
  void *nms;
  size_t bytes;
};
static PostWs post_carve(void *base, int b, int n, int mpad) {
  char *p = static_cast<char *>(base);
  size_t o = 0;
  auto take = [&](size_t bytes) { char *q = p ? p + o : nullptr; o += (bytes + 255) & ~(size_t)255; return q; };
  PostWs w;
  w.voxel = take(gn_voxel_batched_workspace_bytes(b, n));
  w.scene = (double *)take((size_t)b * (n > 0 ? n : 1) * 3 * sizeof(double));
  w.scene_counts = (int32_t *)take((size_t)b * sizeof(int32_t));
  w.status = (int32_t *)take((size_t)b * sizeof(int32_t));
  w.cmask = (uint8_t *)take((size_t)b * (mpad > 0 ? mpad : 1));
  w.nms = take(gn_grasp_nms_workspace_bytes(b));
  w.bytes = o;
  return w;
}
}  // namespace gn

extern "C" size_t gn_grasp_postprocess_workspace_bytes(int b, int n, int mpad) {
  if (b <= 0 || n < 0 || mpad < 0) return 0;
  return post_carve(nullptr, b, n, mpad).bytes;
}

extern "C" int gn_grasp_postprocess(const float *clouds, const int32_t *n_valid, int b, int n, const float *grasps,
                                    const int32_t *counts, int mpad, double voxel_size, float approach_dist,
                                    double collision_thresh, double translation_thresh, double rotation_thresh, int do_nms,
                                    void *workspace, float *out_rows, int32_t *out_counts, uint8_t *collision_mask,
                                    int32_t *voxel_status, gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n >= 0 && mpad >= 0 && voxel_size > 0.0, "grasp_postprocess: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(clouds && grasps && workspace && out_rows && out_counts, "grasp_postprocess: null pointer");
  GN_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255) == 0, "grasp_postprocess: workspace must be 256-byte aligned");
  const PostWs w = post_carve(workspace, b, n, mpad);
  uint8_t *cm = collision_mask ? collision_mask : w.cmask;
  int32_t *st = voxel_status ? voxel_status : w.status;
  if (int rc = gn_voxel_down_sample_batched(clouds, n_valid, b, n, voxel_size, w.voxel, w.scene, w.scene_counts, st, stream)) return rc;
  // the detector's finger geometry: utils/collision_detector.py:29-30
  if (int rc = gn_collision_detect_batched(w.scene, n, w.scene_counts, grasps, counts, b, mpad, approach_dist, collision_thresh,
                                           0.01, voxel_size, 0.01f, 0.06f, cm, nullptr, nullptr, stream))
    return rc;
  return gn_grasp_nms(grasps, counts, cm, b, mpad, translation_thresh, rotation_thresh, do_nms, w.nms, out_rows, out_counts,
                      stream);
}
