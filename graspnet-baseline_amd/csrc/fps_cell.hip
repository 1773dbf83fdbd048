// fps_cell.hip -- furthest point sampling with exact work skipping for LARGE clouds (one workgroup per cloud).
//
// Replaces pointnet2/_ext_src/src/sampling_gpu.cu:74-234 for clouds that do not fit one CU's registers
// (config[3] of BASELINE.json: 50 000 points): fps_tier_kernel streamed 600 KB of coordinates through one CU on every
// one of the 2 047 steps (4.8 us per step).  A new sample can only lower the running distance of points closer to it
// than their current value, i.e. of points in a ball whose radius shrinks like 1/sqrt(step); everything else is
// min(d, temp) == temp, bit for bit what the reference computes.  So:
//
//   * the cloud is cut into S x S regions of equal population (x slabs, then y cells inside each slab: histogram ->
//     prefix -> ticket, as fps_region_kernel does), a region being RS rows of 64 slots;
//   * all per-point state lives in an L2-resident workspace, (x, y, z, temp) as one 16-byte word per slot plus the
//     slot's arg-max key word, so a region is fetched with ONE 16-byte load per lane and row, at an address computed
//     at run time -- nothing forces the unrolled, statically indexed register file of the register kernels, and the
//     kernel serves any n up to 262 144;
//   * thread t owns region t: it keeps the region's bounding box, cached maximum, candidate (key word + coordinates)
//     and tie flag in registers, so the skip test of all 1 024 regions is one short vector sequence:
//         skip  <=>  dist^2(sample, box) * (1 - 1e-5)  >=  cached max;
//   * region (i, j) belongs to wave (i + 3j) mod 16: the 3 x 3 (5 x 5) regions around a sample land in different waves,
//     each of which then re-reduces one or two 64-point rows;
//   * the candidates' coordinates travel with them (owner lane -> wave candidate -> LDS), so the next sample's
//     coordinates come from LDS right behind the barrier instead of from L2 through the scalar cache.
//
// Tie order is the reference's shared-memory tree's: rank(k) = bitrev(k % bs) * Q + k / bs, smallest wins (fps.hip
// header); it is carried in the key word, ((0x7ffff - rank) << 4) | wave (never 0: ranks stay below 2^18), under the value in a 64-bit ds_max_u64.
// The kernel also reports the first step whose maximum was shared by DIFFERENT locations (tie_out), which lets the
// deeper levels' samplers return the identity prefix (gn_furthest_point_sampling's prefix_ok).
// Distances are evaluated exactly as sampling_gpu.cu:108-109 writes them (-ffp-contract=off).
#include <stdlib.h>

#include "fps_util.h"

namespace gn {
namespace GN_FPNS {
namespace {

#ifndef FC_ROWS_IN_FLIGHT
#define FC_ROWS_IN_FLIGHT 4     // rows of 64 slots a wave requests before it waits (8: 3.436 vs 3.441 ms at n = 50 000, no gain)
#endif
constexpr int FC_XB = 2048;   // x histogram bins
constexpr int FC_YB = 1024;   // y histogram bins per slab

using f4 = __attribute__((ext_vector_type(4))) float;

// phase probe (instrumented build only, profiles/build_probe.sh): wave-uniform s_memtime deltas per wave of cloud 0 --
// 0 skip test, 1 row fetch (issue -> data), 2 region update + reduce, 3 wave candidate, 4 LDS fold + barrier, 5 read-back +
// tie check + bookkeeping; 6 = regions processed, 7 = fetch batches
#ifdef GN_PHASE_PROBE
__device__ unsigned long long gn_fpsc_probe[16 * 8];
#define FC_PROBE_INIT() unsigned long long fp_[8] = {}; unsigned long long fl_ = __builtin_amdgcn_s_memtime()
#define FC_PROBE(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); fp_[i] += n_ - fl_; fl_ = n_; } while (0)
#define FC_PROBE_ADD(i, v) fp_[i] += (unsigned long long)(v)
#define FC_PROBE_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define FC_PROBE_FLUSH() do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) gn_fpsc_probe[(threadIdx.x >> 6) * 8 + i_] = fp_[i_]; } while (0)
#else
#define FC_PROBE_INIT()
#define FC_PROBE(i)
#define FC_PROBE_ADD(i, v)
#define FC_PROBE_WAIT()
#define FC_PROBE_FLUSH()
#endif

__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = row16_max_u32_fast(v);
  const unsigned a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const unsigned c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return max(max(a, b), max(c, d));
}
__device__ __forceinline__ float rl(float v, int src) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
}
// floor(q / d) for q * d < 2^32 with magic = ceil(2^32 / d) (d >= 2)
__device__ __forceinline__ int fdiv(int q, int d, unsigned magic) { return d == 1 ? q : (int)__umulhi((unsigned)q, magic); }

template <int S, int NWC, int RS>
__global__ __launch_bounds__(1024) void fps_cell_kernel(const float *__restrict__ xyz, int n, int m, int bs, int lg, int Q,
                                                        unsigned qmagic, char *__restrict__ ws, size_t ws_stride,
                                                        int32_t *__restrict__ idx, float *__restrict__ new_xyz,
                                                        int32_t *__restrict__ tie_out) {
  // the build runs on all 16 waves; the chain keeps NWC of them (the others end before it): every wave that takes part in
  // a step pays ~110 instructions of bookkeeping whether or not one of its regions changed, and four waves share a SIMD
  constexpr int NT = 1024, NW = 16, R = S * S, RPW = R / NWC, ROW = 64 * RS, P = R * ROW, NTC = NWC * 64;
  constexpr int UNR = FC_ROWS_IN_FLIGHT / RS;        // regions fetched together: UNR * RS rows in flight
  static_assert(S == 16 || S == 32, "16 x 16 or 32 x 32 regions");
  static_assert(RS == 1 || RS == 2 || RS == 4, "1, 2 or 4 rows of 64 slots per region");
  static_assert(RPW <= 64 && RPW * NWC == R && S % NWC == 0, "one owner lane per region");
  extern __shared__ int lds_i[];
  int *hx = lds_i;                 // [FC_XB]
  int *hy = hx + FC_XB;            // [S][FC_YB]
  int *scr = hy + S * FC_YB;       // [64]
  __shared__ unsigned long long s_key[3];
  __shared__ f4 s_cand[3][NWC];      // per step and wave: the wave candidate's coordinates
  __shared__ f4 s_reg[R];            // per region: its candidate's (x, y, z, key word), written by the lane that holds it
  __shared__ int s_tie;

  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  idx += (size_t)b * m;
  if (new_xyz) new_xyz += (size_t)b * m * 3;
  f4 *pt = reinterpret_cast<f4 *>(ws + (size_t)b * ws_stride);          // [P] (x, y, z, temp)
  unsigned *rk = reinterpret_cast<unsigned *>(pt + P);                     // [P] key word, 0 = empty slot
  unsigned char *slab_of = reinterpret_cast<unsigned char *>(rk + P);      // [n] scratch of the build
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bmask = bs - 1;
  auto rank_of = [&](int k) { return bitrev_bits(k & bmask, lg) * Q + (k >> lg); };

  // ================= build: equal-population regions -> workspace =================
  float xmn = INFINITY, xmx = -INFINITY, ymn = INFINITY, ymx = -INFINITY;
  for (int k = t; k < n; k += NT) {
    const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1];
    xmn = fminf(xmn, x); xmx = fmaxf(xmx, x); ymn = fminf(ymn, y); ymx = fmaxf(ymx, y);
  }
  float *scrf = reinterpret_cast<float *>(scr);
  {
    const float a0 = -wave_max(-xmn), a1 = wave_max(xmx), a2 = -wave_max(-ymn), a3 = wave_max(ymx);
    if (lane == 0) { scrf[wave * 4 + 0] = a0; scrf[wave * 4 + 1] = a1; scrf[wave * 4 + 2] = a2; scrf[wave * 4 + 3] = a3; }
  }
  for (int i = t; i < FC_XB + S * FC_YB; i += NT) hx[i] = 0;              // hx and hy are contiguous
  {
    const f4 empty = {0.f, 0.f, 0.f, -1.0f};                              // temp = -1: never updated, never a candidate
    for (int p = t; p < P; p += NT) { pt[p] = empty; rk[p] = 0u; }
  }
  if (t < 3) s_key[t] = 0ULL;
  if (t == 0) s_tie = 0x7fffffff;
  __syncthreads();
  xmn = scrf[0]; xmx = scrf[1]; ymn = scrf[2]; ymx = scrf[3];
  for (int w = 1; w < NW; ++w) {
    xmn = fminf(xmn, scrf[w * 4 + 0]); xmx = fmaxf(xmx, scrf[w * 4 + 1]);
    ymn = fminf(ymn, scrf[w * 4 + 2]); ymx = fmaxf(ymx, scrf[w * 4 + 3]);
  }
  __syncthreads();
  const float sx = xmx > xmn ? (float)FC_XB / (xmx - xmn) : 0.f;
  const float sy = ymx > ymn ? (float)FC_YB / (ymx - ymn) : 0.f;
  auto binx = [&](float x) { return min(FC_XB - 1, max(0, (int)((x - xmn) * sx))); };
  auto biny = [&](float y) { return min(FC_YB - 1, max(0, (int)((y - ymn) * sy))); };
  const int T2 = (n + R - 1) / R;        // points per region (<= ROW, host checks)
  const int T1 = S * T2;                 // points per x slab
  const unsigned mg1 = (unsigned)((0x100000000ull + T1 - 1) / T1), mg2 = (unsigned)((0x100000000ull + T2 - 1) / T2);

  // a wave scans one 1024-entry histogram row (exclusive prefix), 16 entries per lane, no workgroup barrier
  auto wave_scan_row = [&](int *a) {
    int v[16], loc = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) { v[e] = a[lane * 16 + e]; loc += v[e]; }
    int incl = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    int run = incl - loc;
#pragma unroll
    for (int e = 0; e < 16; ++e) { a[lane * 16 + e] = run; run += v[e]; }
  };

  // ---- x slabs: histogram -> prefix (two rows of 1024, the second offset by the first's total) -> ticket
  for (int k = t; k < n; k += NT) atomicAdd(&hx[binx(xyz[k * 3 + 0])], 1);
  __syncthreads();
  if (wave == 1) {                       // total of the first row, read before wave 0 scans it in place
    int loc = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) loc += hx[lane * 16 + e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) loc += __shfl_xor(loc, o);
    scr[33] = loc;                       // uniform: every lane writes the same value
  }
  __syncthreads();
  if (wave < 2) {
    wave_scan_row(hx + wave * 1024);
    if (wave == 1) {
      const int off = scr[33];
#pragma unroll
      for (int e = 0; e < 16; ++e) hx[1024 + lane * 16 + e] += off;
    }
  }
  __syncthreads();
  for (int k = t; k < n; k += NT) {
    const int q = atomicAdd(&hx[binx(xyz[k * 3 + 0])], 1);
    const int sl = min(fdiv(q, T1, mg1), S - 1);
    slab_of[k] = (unsigned char)sl;
    atomicAdd(&hy[sl * FC_YB + biny(xyz[k * 3 + 1])], 1);
  }
  __syncthreads();
  // ---- y cells inside each slab
  for (int sl = wave; sl < S; sl += NW) wave_scan_row(hy + sl * FC_YB);
  __syncthreads();
  for (int k = t; k < n; k += NT) {
    const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
    const int sl = slab_of[k];
    const int q2 = atomicAdd(&hy[sl * FC_YB + biny(y)], 1);
    const int cy = min(fdiv(q2, T2, mg2), S - 1);
    const int r = min(q2 - cy * T2, ROW - 1);
    const int w = (sl + 3 * cy) & (NWC - 1);                              // region (sl, cy) -> chain wave
    const int ol = (sl / NWC) * S + cy;                                   //                -> owner lane in that wave
    const int p = (w * RPW + ol) * ROW + r;
    const float mag = fp_sumsq3(x, y, z);
    const f4 v = {x, y, z, ((double)mag <= 1e-3) ? -1.0f : 1e10f};        // sampling_gpu.cu:105-106, sampling.cpp:78-80
    pt[p] = v;
    rk[p] = ((0x7ffffu - (unsigned)rank_of(k)) << 4) | (unsigned)w;
  }
  __syncthreads();                       // workspace writes of the whole workgroup are visible to its waves from here
  if (wave >= NWC) return;               // the chain's barriers count the remaining waves only

  // ================= the chain =================
  // owner-lane state of region (wave, lane), lane < RPW
  float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY, bz0 = INFINITY, bz1 = -INFINITY;
  float gmax = lane < RPW ? INFINITY : -1.0f;      // cached max of the region's running distances (inf = not computed yet)
  bool gtie = false;                                // the region's max is shared by different locations
  f4 *regw = s_reg + wave * RPW;                    // the candidates of this wave's regions (coordinates + key word)
  f4 *ptw = pt + (size_t)wave * RPW * ROW;          // this wave's regions
  const unsigned *rkw = rk + (size_t)wave * RPW * ROW;

  const float p0x = xyz[0], p0y = xyz[1], p0z = xyz[2];
  float x1 = p0x, y1 = p0y, z1 = p0z;
  if (t == 0) {
    idx[0] = 0;
    if (new_xyz) { new_xyz[0] = x1; new_xyz[1] = y1; new_xyz[2] = z1; }
  }
  unsigned long long wkey = 0ULL;                   // this wave's candidate (wave-uniform), kept while no region changes
  f4 wrec = {0.f, 0.f, 0.f, 0.f};
  int wsrc = -1;                                    // ... and the owner lane of its region
  int first_tie = 0x7fffffff;

  FC_PROBE_INIT();
  for (int j = 1; j < m; ++j) {
    const bool first = j == 1;
    FC_PROBE(5);
    // ---- which of this wave's regions can change?  (lane a tests region a)
    const float ex = fmaxf(fmaxf(bx0 - x1, x1 - bx1), 0.f);
    const float ey = fmaxf(fmaxf(by0 - y1, y1 - by1), 0.f);
    const float ez = fmaxf(fmaxf(bz0 - z1, z1 - bz1), 0.f);
    const float lb = (ex * ex + ey * ey + ez * ez) * 0.99999f;
    unsigned long long am = __ballot(lane < RPW && (first || !(lb >= gmax)));
    const bool touched = am != 0ULL;
    FC_PROBE(0);
    while (am) {
      FC_PROBE_ADD(7, 1);
      int a[UNR];
      f4 v[UNR][RS];
      unsigned kk[UNR][RS];
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        a[u] = -1;
        if (am) {
          a[u] = __ffsll((long long)am) - 1;
          am &= am - 1;
#pragma unroll
          for (int r = 0; r < RS; ++r) {
            v[u][r] = ptw[(a[u] * RS + r) * 64 + lane];
            kk[u][r] = rkw[(a[u] * RS + r) * 64 + lane];
          }
        }
      }
      FC_PROBE_WAIT();
      FC_PROBE(1);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (a[u] < 0) continue;                                           // wave-uniform
        FC_PROBE_ADD(6, 1);
        float best = -1.0f, d2s[RS];
#pragma unroll
        for (int r = 0; r < RS; ++r) {
          const float dx = v[u][r].x - x1, dy = v[u][r].y - y1, dz = v[u][r].z - z1;
          const float d = fp_sumsq3(dx, dy, dz);
          const float d2 = vmin(d, v[u][r].w);
          if (d2 != v[u][r].w) reinterpret_cast<float *>(ptw + (a[u] * RS + r) * 64 + lane)[3] = d2;
          d2s[r] = d2;
          best = vmax(best, d2);
        }
        const float wmax = wave_max_fast(best);
        bool ctie = false;
        if (wmax >= 0.0f) {
          // who holds the maximum?  (skipped points and padding sit at -1)
          unsigned long long hm[RS];
          int cnt = 0;
#pragma unroll
          for (int r = 0; r < RS; ++r) { hm[r] = __ballot(d2s[r] == wmax); cnt += __popcll(hm[r]); }
          if (cnt == 1) {
#pragma unroll
            for (int r = 0; r < RS; ++r)
              if (hm[r] && d2s[r] == wmax) {                               // the one lane (and row) that holds it
                const f4 rec = {v[u][r].x, v[u][r].y, v[u][r].z, __uint_as_float(kk[u][r])};
                regw[a[u]] = rec;
              }
          } else {
            // rare: several points of the region share the maximum -> reference order = smallest rank = largest key word;
            // copies of one location are harmless, different locations make the step a tie
            unsigned my = 0u;
            float sxx = 0.f, syy = 0.f, szz = 0.f;
            float lx0 = INFINITY, lx1 = -INFINITY, ly0 = INFINITY, ly1 = -INFINITY, lz0 = INFINITY, lz1 = -INFINITY;
#pragma unroll
            for (int r = 0; r < RS; ++r)
              if (d2s[r] == wmax) {
                if (kk[u][r] > my) { my = kk[u][r]; sxx = v[u][r].x; syy = v[u][r].y; szz = v[u][r].z; }
                lx0 = fminf(lx0, v[u][r].x); lx1 = fmaxf(lx1, v[u][r].x); ly0 = fminf(ly0, v[u][r].y);
                ly1 = fmaxf(ly1, v[u][r].y); lz0 = fminf(lz0, v[u][r].z); lz1 = fmaxf(lz1, v[u][r].z);
              }
            const unsigned ckey = wave_max_u32(my);
            ctie = (-wave_max(-lx0) != wave_max(lx1)) || (-wave_max(-ly0) != wave_max(ly1)) ||
                   (-wave_max(-lz0) != wave_max(lz1));
            if (my == ckey) {                                              // key words are unique: one lane
              const f4 rec = {sxx, syy, szz, __uint_as_float(ckey)};
              regw[a[u]] = rec;
            }
          }
        }
        float nx0 = 0.f, nx1 = 0.f, ny0 = 0.f, ny1 = 0.f, nz0 = 0.f, nz1 = 0.f;
        if (first) {
          // the region's bounding box, over the points that can ever be updated (temp >= 0: not skipped, not padding)
          float lx0 = INFINITY, lx1 = -INFINITY, ly0 = INFINITY, ly1 = -INFINITY, lz0 = INFINITY, lz1 = -INFINITY;
#pragma unroll
          for (int r = 0; r < RS; ++r)
            if (v[u][r].w >= 0.0f) {
              lx0 = fminf(lx0, v[u][r].x); lx1 = fmaxf(lx1, v[u][r].x); ly0 = fminf(ly0, v[u][r].y);
              ly1 = fmaxf(ly1, v[u][r].y); lz0 = fminf(lz0, v[u][r].z); lz1 = fmaxf(lz1, v[u][r].z);
            }
          nx0 = -wave_max(-lx0); nx1 = wave_max(lx1); ny0 = -wave_max(-ly0); ny1 = wave_max(ly1);
          nz0 = -wave_max(-lz0); nz1 = wave_max(lz1);
        }
        if (lane == a[u]) {
          gmax = wmax;
          gtie = ctie;
          if (first) { bx0 = nx0; bx1 = nx1; by0 = ny0; by1 = ny1; bz0 = nz0; bz1 = nz1; }
        }
      }
    }
    FC_PROBE(2);
    // ---- the wave's candidate: largest (max, key word) over its regions; recomputed only when a region changed
    if (touched) {
      const float wm = wave_max_fast(gmax);          // lanes without a region hold -1
      wkey = 0ULL;
      wsrc = -1;
      if (wm >= 0.0f) {
        const bool mine = gmax == wm;
        unsigned long long eq = __ballot(mine);
        if (__popcll(eq) > 1) {                       // rare: two regions of the wave share the value -> larger key word
          const unsigned mykey = mine ? __float_as_uint(regw[lane].w) : 0u;
          const unsigned kmax = wave_max_u32(mykey);
          eq = __ballot(mine && mykey == kmax);
        }
        wsrc = __ffsll((long long)eq) - 1;
        wrec = regw[wsrc];                            // one address: broadcast read
        wkey = ((unsigned long long)__float_as_uint(wm) << 32) | __float_as_uint(wrec.w);
      }
    }
    FC_PROBE(3);
    const int slot = j % 3;
    if (lane == 0) {
      s_cand[slot][wave] = wrec;
      atomicMax(&s_key[slot], wkey);
    }
    if (t == 0) s_key[(j + 1) % 3] = 0ULL;         // last read in step j-2; every wave is past barrier j-1
    __syncthreads();
    FC_PROBE(4);
    const unsigned long long gk = s_key[slot];
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gk);
    const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(gk >> 32));
    if (gk != 0ULL) {
      const f4 c = s_cand[slot][glo & (NW - 1)];
      x1 = c.x; y1 = c.y; z1 = c.z;
    } else {                                        // every candidate skipped -> the reference returns index 0
      x1 = p0x; y1 = p0y; z1 = p0z;
    }
    // ---- was this step a tie between DIFFERENT locations?  every region whose maximum equals the winning value is
    // looked at: the winner's own region says so itself (gtie, settled when it was reduced); any other one compares its
    // candidate's coordinates with the winner's (a copy of the winning location in another region is harmless)
    if (gk == 0ULL) {
      first_tie = min(first_tie, j);
    } else {
      const bool hit = lane < RPW && __float_as_uint(gmax) == ghi;
      const bool own = (int)(glo & (NW - 1)) == wave && lane == wsrc;      // the winner's region
      const bool cand = hit && (gtie || !own);
      if (__ballot(cand)) {                         // rare
        bool differs = false;
        if (cand) {
          differs = gtie;
          if (!differs) {
            const f4 rec = regw[lane];
            differs = rec.x != x1 || rec.y != y1 || rec.z != z1;
          }
        }
        if (__ballot(differs)) first_tie = min(first_tie, j);
      }
    }
    // the winner's key word now, its index at the end (rank -> index for all steps at once, off the chain)
    if (t == 0) idx[j] = gk != 0ULL ? (int)glo : -1;
  }
  FC_PROBE_FLUSH();
  if (lane == 0) atomicMin(&s_tie, first_tie);
  __syncthreads();                                  // also: thread 0's idx[] stores are visible to the workgroup
  if (tie_out && t == 0) tie_out[b] = s_tie;
  for (int j = 1 + t; j < m; j += NTC) {
    const int g = idx[j];
    int k = 0;                                      // every candidate skipped -> the reference returns index 0
    if (g != -1) {
      k = pos_to_k((int)(0x7ffffu - ((unsigned)g >> 4)), bs, lg, Q, qmagic, n);
      k = k < 0 ? 0 : k;
    }
    idx[j] = k;
    if (new_xyz) { new_xyz[j * 3 + 0] = xyz[k * 3 + 0]; new_xyz[j * 3 + 1] = xyz[k * 3 + 1]; new_xyz[j * 3 + 2] = xyz[k * 3 + 2]; }
  }
}

template <int S, int NWC, int RS>
int launch(const float *xyz, int b, int n, int m, int bs, int lg, int Q, unsigned qmagic, void *workspace, size_t stride,
           int32_t *idx, float *new_xyz, int32_t *tie_out, hipStream_t s) {
  auto kern = fps_cell_kernel<S, NWC, RS>;
  const int lds = (FC_XB + S * FC_YB + 64) * (int)sizeof(int);
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
    set_error("furthest_point_sampling: cannot reserve %d bytes of LDS", lds);
    return GN_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(kern, dim3(b), dim3(1024), lds, s, xyz, n, m, bs, lg, Q, qmagic, (char *)workspace, stride, idx, new_xyz,
                     tie_out);
  return GN_OK;
}

}  // namespace

#if defined(GN_PHASE_PROBE) && GN_FP == 0
extern "C" int gn_phase_probe_read_fps_cell(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gn_fpsc_probe), sizeof(unsigned long long) * 128) == hipSuccess ? 0 : -1;
}
#endif

int fps_cell_launch(const float *xyz, int b, int n, int m, int bs, int lg, int Q, unsigned qmagic, void *workspace,
                    int32_t *idx, float *new_xyz, int32_t *tie_out, int force_s, hipStream_t s) {
  const FpsCellCfg c = fps_cell_config(n, force_s);
  const size_t stride = fps_cell_workspace_bytes_one(n, force_s);
  // chain waves: all 16.  Measured at n = 50 000 (S = 16, 4 rows per region): 4 waves 5.0 ms, 8 waves 3.75, 16 waves 3.45 --
  // fewer waves pay less per-step bookkeeping, but a wave walks its active regions one fetch (L2 round trip) after the
  // other, and sixteen waves overlap sixteen of them
#define GN_CELL(SS, WW, RR) \
  if (c.S == SS && c.RS == RR)                                \
    return launch<SS, WW, RR>(xyz, b, n, m, bs, lg, Q, qmagic, workspace, stride, idx, new_xyz, tie_out, s)
  GN_CELL(16, 16, 1); GN_CELL(16, 16, 2); GN_CELL(16, 16, 4);
  GN_CELL(32, 16, 1); GN_CELL(32, 16, 2); GN_CELL(32, 16, 4);
#undef GN_CELL
  set_error("furthest_point_sampling: no cell kernel for n=%d", n);
  return GN_ERR_INVALID_ARG;
}

}  // namespace GN_FPNS
}  // namespace gn
