// fps.hip -- furthest point sampling for gfx950 (wave64, one workgroup per cloud).
//
// Replaces pointnet2/_ext_src/src/sampling_gpu.cu:74-234 (+ sampling.cpp:70-91).
//
// The reference runs `bs = opt_n_threads(n)` threads; thread `tid` strides over
// k = tid, tid+bs, ... and a shared-memory tree picks the block arg-max.  Its
// result is:  argmax_k temp[k]  with ties resolved by the total order
//     rank(k) = ( bitrev_log2(bs)(k % bs) ,  k / bs )            (lexicographic, smaller wins)
// (tree keeps the lower slot on ties at every level => LSB-first order of the
// thread id; within a thread strict '>' keeps the lowest k).  Instead of
// re-enacting that block we lay the cloud out IN RANK ORDER:
//     position p = bitrev(k % bs) * Q + k / bs,   Q = ceil(n / bs)
// and give thread t the contiguous positions [t*PPT, (t+1)*PPT).  "Lowest
// position among equal maxima" is then simply lowest (wave, lane, slot), which
// a ballot + find-first-set delivers for free.
//
// Register-resident variant (n*... <= 24576 positions): every thread keeps its
// PPT points (x,y,z) and their running min-distance in VGPRs for the whole
// 2047-iteration chain; one iteration = PPT distance updates per lane, a DPP
// wave arg-max, one LDS exchange + one barrier across the waves, and one
// broadcast load of the winner's coordinates.  Nothing but idx (and optionally
// the sampled coordinates) is written to HBM; the cloud is read once.
//
// Distances are evaluated exactly as sampling_gpu.cu:108-109 writes them
// (no FMA contraction; this file is compiled with -ffp-contract=off).
#include <stdlib.h>

#include "gn_common.h"
#include "fps_util.h"

namespace gn {
namespace GN_FPNS {

template <int NT, int PPT>
__global__ __launch_bounds__(NT) void fps_reg_kernel(const float *__restrict__ xyz, int n, int m,
                                                      int bs, int lg, int Q, unsigned qmagic,
                                                      int32_t *__restrict__ idx,
                                                      float *__restrict__ new_xyz,
                                                      const int32_t *__restrict__ prefix_ok) {
  // `xyz` is the FPS order of a superset whose sampling met no exact tie between distinct points
  // during its first prefix_ok[b] iterations (written by the producing call's tie_out): if that
  // covers the m samples wanted here, the result is provably the identity prefix.
  if (prefix_ok && prefix_ok[blockIdx.x] >= m && m <= n) {
    const float *src = xyz + (size_t)blockIdx.x * n * 3;
    for (int j = threadIdx.x; j < m; j += NT) idx[(size_t)blockIdx.x * m + j] = j;
    if (new_xyz)
      for (int e = threadIdx.x; e < m * 3; e += NT) new_xyz[(size_t)blockIdx.x * m * 3 + e] = src[e];
    return;
  }
  // cross-wave arg-max: every wave folds its candidate into ONE 64-bit LDS word with ds_max_u64,
  //   key = (value bits << 32) | ~position     (values are >= 0 or the -1 sentinel mapped to 0)
  // so the largest value wins and, among equal values, the lowest position = reference tie order.
  // Three rotating words: word j%3 is used by iteration j and re-armed during iteration j+2.
  __shared__ unsigned long long s_key[3];
  // Small levels (<= 4096 positions, 48 KB): the cloud also sits in LDS, and the winner's coordinates -- the one
  // dependent fetch of every iteration -- come from there instead of through the scalar cache from L2.  Measured on the
  // full SA2 / SA3 / SA4 chains (profiles/fps_small_probe.py): 598 / 454 / 413 ns per iteration with the global fetch.
  constexpr bool LDSXYZ = NT * PPT <= 4096;
  extern __shared__ float lds_xyz[];

  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  idx += (size_t)b * m;
  if (new_xyz) new_xyz += (size_t)b * m * 3;
  const int t = threadIdx.x;
  const int lane = t & 63;
  if (LDSXYZ)
    for (int e = t; e < n * 3; e += NT) lds_xyz[e] = xyz[e];

  float px[PPT], py[PPT], pz[PPT], tmp[PPT];
#pragma unroll
  for (int s = 0; s < PPT; ++s) {
    const int k = pos_to_k(t * PPT + s, bs, lg, Q, qmagic, n);
    float x = 0.f, y = 0.f, z = 0.f, t0 = -1.0f;
    if (k >= 0) {
      x = xyz[k * 3 + 0];
      y = xyz[k * 3 + 1];
      z = xyz[k * 3 + 2];
      const float mag = fp_sumsq3(x, y, z);
      // sampling_gpu.cu:105-106: `if (mag <= 1e-3) continue;` (float vs double literal).
      // A skipped point never updates temp and is never a candidate: temp = -1 does both.
      t0 = ((double)mag <= 1e-3) ? -1.0f : 1e10f;   // 1e10 = sampling.cpp:78-80
    }
    px[s] = x; py[s] = y; pz[s] = z; tmp[s] = t0;
  }

  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (t == 0) {
    idx[0] = 0;
    if (new_xyz) { new_xyz[0] = x1; new_xyz[1] = y1; new_xyz[2] = z1; }
  }
  if (t < 3) s_key[t] = 0ULL;
  __syncthreads();

  // Warm start behind a tie: the producing call met its first tie between distinct points at step T = prefix_ok[b] < m.
  // Steps below T are still the identity prefix (a unique maximum is the same point in any tie order), so samples
  // 0 .. T-1 are written at once and the running distances after them are rebuilt WITHOUT the arg-max chain -- one
  // distance update per (point, sample), no reduction, no barrier, ~4x cheaper per step -- and the chain proper starts
  // at step T.  min() is exact in any order: the state equals what T - 1 chain iterations would have left.
  int j0 = 1;
  if (LDSXYZ && prefix_ok && m <= n) {
    const int T = min(prefix_ok[blockIdx.x], m);
    if (T >= 2) {
      for (int j = 1 + t; j < T; j += NT) {
        idx[j] = j;
        if (new_xyz) { new_xyz[j * 3 + 0] = lds_xyz[j * 3 + 0]; new_xyz[j * 3 + 1] = lds_xyz[j * 3 + 1]; new_xyz[j * 3 + 2] = lds_xyz[j * 3 + 2]; }
      }
      for (int jj = 0; jj + 1 < T; ++jj) {          // sample jj's update belongs to chain iteration jj + 1
        const float sx = lds_xyz[jj * 3 + 0], sy = lds_xyz[jj * 3 + 1], sz = lds_xyz[jj * 3 + 2];
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
          const float dx = px[s] - sx, dy = py[s] - sy, dz = pz[s] - sz;
          const float d = fp_sumsq3(dx, dy, dz);
          tmp[s] = vmin(d, tmp[s]);
        }
      }
      x1 = lds_xyz[(T - 1) * 3 + 0]; y1 = lds_xyz[(T - 1) * 3 + 1]; z1 = lds_xyz[(T - 1) * 3 + 2];
      j0 = T;
    }
  }

  for (int j = j0; j < m; ++j) {
    float best = -1.0f;
    int bslot = 0;
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
      const float dx = px[s] - x1, dy = py[s] - y1, dz = pz[s] - z1;
      const float d = fp_sumsq3(dx, dy, dz);
      const float d2 = vmin(d, tmp[s]);
      tmp[s] = d2;
      const bool gt = d2 > best;
      best = gt ? d2 : best;
      bslot = gt ? s : bslot;
    }
    // ---- wave arg-max, lowest lane among equal maxima
    const float rmax = row16_max_fast(best);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 48));
    const float wmax = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));   // scalar operands: folds to s/v ops, uniform
    const unsigned long long eq = __ballot(best == wmax);
    const int src = eq ? (__ffsll((long long)eq) - 1) : 0;
    const int wpos = __builtin_amdgcn_readlane(t * PPT + bslot, src);
    const int slot = j % 3;
    if (lane == 0) {
      // wmax < 0 only when every candidate of the wave is skipped/padding: key 0 loses to anything
      const unsigned long long key =
          wmax >= 0.0f ? (((unsigned long long)__float_as_uint(wmax) << 32) | (unsigned)(~wpos)) : 0ULL;
      atomicMax(&s_key[slot], key);
      if (t == 0) s_key[(j + 1) % 3] = 0ULL;   // last read in iteration j-2; every wave is past barrier j-1
    }
    __syncthreads();
    const unsigned long long gk = s_key[slot];
    const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(gk >> 32));
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gk);
    // all candidates skipped -> best=-1,besti=0 in every reference thread -> index 0
    int k = gk != 0ULL ? pos_to_k((int)~glo, bs, lg, Q, qmagic, n) : 0;
    (void)ghi;
    k = __builtin_amdgcn_readfirstlane(k < 0 ? 0 : k);
    if (LDSXYZ) {
      x1 = lds_xyz[k * 3 + 0];
      y1 = lds_xyz[k * 3 + 1];
      z1 = lds_xyz[k * 3 + 2];
    } else {
      x1 = xyz[k * 3 + 0];
      y1 = xyz[k * 3 + 1];
      z1 = xyz[k * 3 + 2];
    }
    if (t == 0) {
      idx[j] = k;
      if (new_xyz) { new_xyz[j * 3 + 0] = x1; new_xyz[j * 3 + 1] = y1; new_xyz[j * 3 + 2] = z1; }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Spatially-sorted variant for the model's main level (12 288 < positions <= 20 480; 24 points per
// lane would spill past the 128 VGPRs a 1024-thread workgroup may use).
//
// Work per iteration is bound by VALU issue (4 waves per SIMD x PPT updates x ~12 instructions).
// After a few dozen samples most of those updates are no-ops: a point's running distance can only
// shrink if the new sample is closer than its current value.  Each of the 16 waves therefore owns
// one spatial CELL of the cloud (4 x-slabs x 4 y-slabs of equal population, built in LDS with two
// histogram / prefix / ticket passes), keeps its bounding box and its cached (max, candidate), and
// SKIPS an iteration when  dist^2(new sample, box) * (1 - 1e-5)  >=  its cached max:
// then min(d, temp) == temp for every point of the wave, exactly as the reference would compute.
// The placement of points is arbitrary with respect to the reference's tie order, so ties are
// resolved by the explicit rank  bitrev(k % bs) * Q + k / bs  (carried in the LDS atomic-max key
// across waves; inside a wave a flagged tie takes a rare slow path that scans for the minimum rank).
constexpr int FPS_BINS = 2048;

// exclusive prefix sum of a[0..2047] in LDS by 1024 threads (2 elements each); scratch: 16 ints
__device__ __forceinline__ void block_exclusive_scan_2048(int *a, int *scratch, int t) {
  const int lane = t & 63, wave = t >> 6;
  const int a0 = a[2 * t], a1 = a[2 * t + 1];
  int incl = a0 + a1;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) scratch[wave] = incl;
  __syncthreads();
  int woff = 0;
  for (int w = 0; w < wave; ++w) woff += scratch[w];
  const int excl = woff + incl - (a0 + a1);
  a[2 * t] = excl;
  a[2 * t + 1] = excl + a0;
  __syncthreads();
}

template <int PPT>
__global__ __launch_bounds__(1024) void fps_spatial_kernel(const float *__restrict__ xyz, int n, int m, int bs,
                                                           int lg, int Q, unsigned qmagic,
                                                           int32_t *__restrict__ idx, float *__restrict__ new_xyz,
                                                           int32_t *__restrict__ tie_out) {
  constexpr int NT = 1024, NW = 16, CELL = 64 * PPT;
  extern __shared__ int lds_i[];
  int *perm = lds_i;                 // [NW*CELL] position -> original index (-1 = empty)
  int *hx = perm + NW * CELL;        // [FPS_BINS]
  int *hy = hx + FPS_BINS;           // [4][FPS_BINS]
  int *scr = hy + 4 * FPS_BINS;      // [64]
  __shared__ unsigned long long s_key[3];

  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  idx += (size_t)b * m;
  if (new_xyz) new_xyz += (size_t)b * m * 3;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bmask = bs - 1;
  auto rank_of = [&](int k) { return bitrev_bits(k & bmask, lg) * Q + (k >> lg); };

  // ---- S1: x / y extent of the cloud
  float xmn = INFINITY, xmx = -INFINITY, ymn = INFINITY, ymx = -INFINITY;
  for (int i = 0; i < PPT; ++i) {
    const int k = t + i * NT;
    if (k < n) {
      const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1];
      xmn = fminf(xmn, x); xmx = fmaxf(xmx, x); ymn = fminf(ymn, y); ymx = fmaxf(ymx, y);
    }
  }
  float *scrf = reinterpret_cast<float *>(scr);
  {
    const float a0 = -wave_max(-xmn), a1 = wave_max(xmx), a2 = -wave_max(-ymn), a3 = wave_max(ymx);
    if (lane == 0) { scrf[wave * 4 + 0] = a0; scrf[wave * 4 + 1] = a1; scrf[wave * 4 + 2] = a2; scrf[wave * 4 + 3] = a3; }
  }
  for (int i = t; i < 5 * FPS_BINS; i += NT) hx[i] = 0;      // hx and hy are contiguous
  for (int i = t; i < NW * CELL; i += NT) perm[i] = -1;
  __syncthreads();
  xmn = scrf[0]; xmx = scrf[1]; ymn = scrf[2]; ymx = scrf[3];
  for (int w = 1; w < NW; ++w) {
    xmn = fminf(xmn, scrf[w * 4 + 0]); xmx = fmaxf(xmx, scrf[w * 4 + 1]);
    ymn = fminf(ymn, scrf[w * 4 + 2]); ymx = fmaxf(ymx, scrf[w * 4 + 3]);
  }
  __syncthreads();
  const float sx = xmx > xmn ? (float)FPS_BINS / (xmx - xmn) : 0.f;
  const float sy = ymx > ymn ? (float)FPS_BINS / (ymx - ymn) : 0.f;
  auto binx = [&](float x) { return min(FPS_BINS - 1, max(0, (int)((x - xmn) * sx))); };
  auto biny = [&](float y) { return min(FPS_BINS - 1, max(0, (int)((y - ymn) * sy))); };
  const int T2 = (n + 15) / 16;      // points per cell (<= CELL because 16*CELL >= n)
  const int T1 = 4 * T2;             // points per x slab

  // ---- S2: x slabs of equal population (histogram -> prefix -> ticket)
  for (int i = 0; i < PPT; ++i) {
    const int k = t + i * NT;
    if (k < n) atomicAdd(&hx[binx(xyz[k * 3 + 0])], 1);
  }
  __syncthreads();
  block_exclusive_scan_2048(hx, scr, t);
  unsigned long long slabs = 0ULL;     // 2 bits per owned point
  for (int i = 0; i < PPT; ++i) {
    const int k = t + i * NT;
    if (k < n) {
      const int q = atomicAdd(&hx[binx(xyz[k * 3 + 0])], 1);
      const int sl = min(q / T1, 3);
      slabs |= (unsigned long long)sl << (2 * i);
      atomicAdd(&hy[sl * FPS_BINS + biny(xyz[k * 3 + 1])], 1);
    }
  }
  __syncthreads();
  // ---- S3: y cells inside each slab
  for (int sl = 0; sl < 4; ++sl) block_exclusive_scan_2048(hy + sl * FPS_BINS, scr, t);
  for (int i = 0; i < PPT; ++i) {
    const int k = t + i * NT;
    if (k < n) {
      const int sl = (int)((slabs >> (2 * i)) & 3ULL);
      const int q2 = atomicAdd(&hy[sl * FPS_BINS + biny(xyz[k * 3 + 1])], 1);
      const int cy = min(q2 / T2, 3);
      const int r = q2 - cy * T2;                     // < T2 <= CELL (last cell: slab population <= T1)
      perm[(sl * 4 + cy) * CELL + min(r, CELL - 1)] = k;
    }
  }
  __syncthreads();

  // ---- load this lane's points: wave = cell, slot s of lane l = cell entry s*64 + l
  float px[PPT], py[PPT], pz[PPT], tmp[PPT];
  float bx0 = INFINITY, bx1 = -INFINITY, by0 = INFINITY, by1 = -INFINITY, bz0 = INFINITY, bz1 = -INFINITY;
#pragma unroll
  for (int s = 0; s < PPT; ++s) {
    const int k = perm[wave * CELL + s * 64 + lane];
    float x = 0.f, y = 0.f, z = 0.f, t0 = -1.0f;
    if (k >= 0) {
      x = xyz[k * 3 + 0]; y = xyz[k * 3 + 1]; z = xyz[k * 3 + 2];
      const float mag = fp_sumsq3(x, y, z);
      t0 = ((double)mag <= 1e-3) ? -1.0f : 1e10f;       // sampling_gpu.cu:105-106, sampling.cpp:78-80
      bx0 = fminf(bx0, x); bx1 = fmaxf(bx1, x); by0 = fminf(by0, y); by1 = fmaxf(by1, y);
      bz0 = fminf(bz0, z); bz1 = fmaxf(bz1, z);
    }
    px[s] = x; py[s] = y; pz[s] = z; tmp[s] = t0;
  }
  bx0 = -wave_max(-bx0); bx1 = wave_max(bx1); by0 = -wave_max(-by0); by1 = wave_max(by1);
  bz0 = -wave_max(-bz0); bz1 = wave_max(bz1);

  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (t == 0) {
    idx[0] = 0;
    if (new_xyz) { new_xyz[0] = x1; new_xyz[1] = y1; new_xyz[2] = z1; }
  }
  if (t < 3) s_key[t] = 0ULL;
  __syncthreads();

  float c_wmax = INFINITY;            // cached max of this wave's running distances (inf = not computed yet)
  unsigned long long c_key = 0ULL;
  bool c_tie = false;                 // the cached max is shared by points with DIFFERENT coordinates
  int first_tie = 0x7fffffff;         // first iteration whose global max was such a tie (or had no candidate)
  for (int j = 1; j < m; ++j) {
    // distance^2 from the new sample to this wave's box (0 inside); empty wave: box is inverted -> inf
    const float ex = fmaxf(fmaxf(bx0 - x1, x1 - bx1), 0.f);
    const float ey = fmaxf(fmaxf(by0 - y1, y1 - by1), 0.f);
    const float ez = fmaxf(fmaxf(bz0 - z1, z1 - bz1), 0.f);
    const float lb = (ex * ex + ey * ey + ez * ez) * 0.99999f;
    if (!(lb >= c_wmax)) {            // wave-uniform
      float best = -1.0f;
      int bslot = 0;
      bool tie = false;
#pragma unroll
      for (int s = 0; s < PPT; ++s) {
        const float dx = px[s] - x1, dy = py[s] - y1, dz = pz[s] - z1;
        const float d = fp_sumsq3(dx, dy, dz);
        const float d2 = vmin(d, tmp[s]);
        tmp[s] = d2;
        tie = tie || (d2 == best);
        const bool gt = d2 > best;
        best = gt ? d2 : best;
        bslot = gt ? s : bslot;
      }
      const float rmax = row16_max_fast(best);
      const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 0));
      const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 16));
      const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 32));
      const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 48));
      const float wmax = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
      const bool mine = best == wmax;
      const unsigned long long eq = __ballot(mine);
      int wrank;
      c_tie = false;
      if (wmax < 0.0f) {
        wrank = 0;
      } else if (__popcll(eq) > 1 || __ballot(mine && tie)) {
        // rare: several points of the wave share the maximum -> reference order = smallest rank.
        // Exact duplicates of one location are harmless for the prefix property (the loser drops to
        // distance 0 and is never sampled); a tie between different locations is recorded.
        int myr = 0x7fffffff;
        float lx0 = INFINITY, lx1 = -INFINITY, ly0 = INFINITY, ly1 = -INFINITY, lz0 = INFINITY, lz1 = -INFINITY;
#pragma unroll
        for (int s = 0; s < PPT; ++s)
          if (tmp[s] == wmax) {
            const int k = perm[wave * CELL + s * 64 + lane];
            if (k >= 0) {
              myr = min(myr, rank_of(k));
              lx0 = fminf(lx0, px[s]); lx1 = fmaxf(lx1, px[s]); ly0 = fminf(ly0, py[s]); ly1 = fmaxf(ly1, py[s]);
              lz0 = fminf(lz0, pz[s]); lz1 = fmaxf(lz1, pz[s]);
            }
          }
        wrank = wave_min_i(myr);
        c_tie = (-wave_max(-lx0) != wave_max(lx1)) || (-wave_max(-ly0) != wave_max(ly1)) ||
                (-wave_max(-lz0) != wave_max(lz1));
      } else {
        const int src = __ffsll((long long)eq) - 1;
        const int kc = perm[wave * CELL + bslot * 64 + lane];
        wrank = rank_of(__builtin_amdgcn_readlane(kc, src));
      }
      c_wmax = wmax;
      c_key = wmax >= 0.0f ? (((unsigned long long)__float_as_uint(wmax) << 32) | (unsigned)(~wrank)) : 0ULL;
    }
    const int slot = j % 3;
    if (lane == 0) {
      atomicMax(&s_key[slot], c_key);
      if (t == 0) s_key[(j + 1) % 3] = 0ULL;
    }
    __syncthreads();
    const unsigned long long gk = s_key[slot];
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gk);
    // this wave holds the global maximum value: a tie if it is not the (unique) winner or ties inside
    if (gk == 0ULL || ((unsigned)(c_key >> 32) == (unsigned)(gk >> 32) && c_key != 0ULL && (c_tie || c_key != gk)))
      first_tie = min(first_tie, j);
    int k = gk != 0ULL ? pos_to_k((int)~glo, bs, lg, Q, qmagic, n) : 0;   // rank -> index
    k = __builtin_amdgcn_readfirstlane(k < 0 ? 0 : k);
    x1 = xyz[k * 3 + 0]; y1 = xyz[k * 3 + 1]; z1 = xyz[k * 3 + 2];
    if (t == 0) {
      idx[j] = k;
      if (new_xyz) { new_xyz[j * 3 + 0] = x1; new_xyz[j * 3 + 1] = y1; new_xyz[j * 3 + 2] = z1; }
    }
  }
  if (tie_out) {
    // a cross-wave tie is seen by the losing wave(s) only: combine over the workgroup
    __shared__ int s_tie;
    if (t == 0) s_tie = 0x7fffffff;
    __syncthreads();
    if (lane == 0) atomicMin(&s_tie, first_tie);
    __syncthreads();
    if (t == 0) tie_out[b] = s_tie;
  }
}

// ---------------------------------------------------------------------------------------------
// Region-interleaved variant (the default for 16 384 < positions <= 20 480).
//
// fps_spatial_kernel skips whole waves, but the waves that do run still pay for all their points,
// and an iteration lasts as long as its busiest wave.  Here the cloud is cut into 64 regions of
// equal population (8 x-slabs x 8 y-cells, same histogram / prefix / ticket construction), and
// region (i,j) is given to wave (i + 3j) % 8 as that wave's slot group j.  The up-to-9 regions a new
// sample can affect therefore land in different waves, each of which updates ONE 5-slot group (and
// re-reduces only that group) while its other 7 groups keep their cached (max, rank).  Lanes 0..7
// of a wave hold the state of its 8 groups (box, max, ~rank, tie), so the skip test of all groups
// is one short vector sequence and the wave's candidate is a 16-lane DPP reduction.
// phase probe of the region kernel (instrumented build only, profiles/build_probe.sh): wave-uniform s_memtime deltas
// accumulated in SGPRs -- 0 skip test, 1 active groups, 2 wave candidate + atomic, 3 barrier, 4 key read-back + winner
// fetch, 5 tie check + bookkeeping; 6 = number of active groups
#ifdef GN_PHASE_PROBE
__device__ unsigned long long gn_fps_probe[8 * 8];
#define FPS_PROBE_INIT() unsigned long long fp_[8] = {}; unsigned long long fl_ = __builtin_amdgcn_s_memtime()
#define FPS_PROBE(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); fp_[i] += n_ - fl_; fl_ = n_; } while (0)
#define FPS_PROBE_ADD(i, v) fp_[i] += (unsigned long long)(v)
#define FPS_PROBE_USE(v) asm volatile("" :: "s"(__builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, (float)(v)))))
#define FPS_PROBE_FLUSH() do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) gn_fps_probe[(threadIdx.x >> 6) * 8 + i_] = fp_[i_]; } while (0)
#if GN_FP == 0
extern "C" int gn_phase_probe_read_fps(unsigned long long *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(gn_fps_probe), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : -1;
}
#endif
#else
#define FPS_PROBE_INIT()
#define FPS_PROBE(i)
#define FPS_PROBE_ADD(i, v)
#define FPS_PROBE_USE(v)
#define FPS_PROBE_FLUSH()
#endif

template <int PPT>
__global__ __launch_bounds__(512) void fps_region_kernel(const float *__restrict__ xyz, int n, int m, int bs, int lg,
                                                         int Q, unsigned qmagic, int32_t *__restrict__ idx,
                                                         float *__restrict__ new_xyz, int32_t *__restrict__ tie_out) {
  constexpr int NT = 512, NW = 8, G = 8, SG = PPT / G, CELL = 64 * PPT, NPT = (NW * CELL) / NT;
  static_assert(PPT % G == 0, "slots per lane must split into 8 groups");
  constexpr int YB = 1024;             // y bins per slab
  extern __shared__ int lds_i[];
  int *perm = lds_i;                   // [NW*CELL]  slot -> (0xffff - rank) << 16 | index of its point, 0 = empty:
                                       // the low word of the arg-max key as it stands, so neither the group reduce
                                       // (index -> rank) nor the winner fetch (rank -> index) computes anything
  int *hx = perm + NW * CELL;          // [FPS_BINS]
  int *hy = hx + FPS_BINS;             // [8][YB]
  int *scr = hy + 8 * YB;              // [64]
  __shared__ unsigned long long s_key[3];

  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  idx += (size_t)b * m;
  if (new_xyz) new_xyz += (size_t)b * m * 3;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bmask = bs - 1;
  auto rank_of = [&](int k) { return bitrev_bits(k & bmask, lg) * Q + (k >> lg); };

  // ---- extent.  The x / y of the thread's NPT points are read ONCE, all loads in flight together, and stay in registers
  // through the three binning passes below (the registers are free until the chain's own arrays come alive): read in
  // place, every pass was a chain of NPT dependent L2 round trips behind its LDS atomics -- 64 us of build per cloud
  float bxv[NPT], byv[NPT];
  {
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xyz), 0, n * 12, 0x00020000);
#pragma unroll
    for (int i = 0; i < NPT; ++i) {
      // the WHOLE point offset goes in voffset: the buffer range check covers voffset + the immediate only (soffset is
      // excluded from it), so a slot k = t + i*NT >= n reads 0 instead of memory behind the cloud; such slots are never used
      const int off = (t + i * NT) * 12;
      bxv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, off, 0, 0));
      byv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, off, 4, 0));
    }
  }
  float xmn = INFINITY, xmx = -INFINITY, ymn = INFINITY, ymx = -INFINITY;
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int k = t + i * NT;
    if (k < n) {
      const float x = bxv[i], y = byv[i];
      xmn = fminf(xmn, x); xmx = fmaxf(xmx, x); ymn = fminf(ymn, y); ymx = fmaxf(ymx, y);
    }
  }
  float *scrf = reinterpret_cast<float *>(scr);
  {
    const float a0 = -wave_max(-xmn), a1 = wave_max(xmx), a2 = -wave_max(-ymn), a3 = wave_max(ymx);
    if (lane == 0) { scrf[wave * 4 + 0] = a0; scrf[wave * 4 + 1] = a1; scrf[wave * 4 + 2] = a2; scrf[wave * 4 + 3] = a3; }
  }
  for (int i = t; i < FPS_BINS + 8 * YB; i += NT) hx[i] = 0;
  for (int i = t; i < NW * CELL; i += NT) perm[i] = 0;
  __syncthreads();
  xmn = scrf[0]; xmx = scrf[1]; ymn = scrf[2]; ymx = scrf[3];
  for (int w = 1; w < NW; ++w) {
    xmn = fminf(xmn, scrf[w * 4 + 0]); xmx = fmaxf(xmx, scrf[w * 4 + 1]);
    ymn = fminf(ymn, scrf[w * 4 + 2]); ymx = fmaxf(ymx, scrf[w * 4 + 3]);
  }
  __syncthreads();
  const float sx = xmx > xmn ? (float)FPS_BINS / (xmx - xmn) : 0.f;
  const float sy = ymx > ymn ? (float)YB / (ymx - ymn) : 0.f;
  auto binx = [&](float x) { return min(FPS_BINS - 1, max(0, (int)((x - xmn) * sx))); };
  auto biny = [&](float y) { return min(YB - 1, max(0, (int)((y - ymn) * sy))); };
  const int T2 = (n + 63) / 64;        // points per region (<= 64*SG because 64*64*SG >= n)
  const int T1 = 8 * T2;               // points per x slab

  // exclusive scan helpers for 512 threads
  auto scan = [&](int *a, int len) {   // len = multiple of NT
    const int per = len / NT;
    int loc = 0;
    for (int e = 0; e < per; ++e) loc += a[t * per + e];
    int incl = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) scr[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += scr[w];
    int run = woff + incl - loc;
    for (int e = 0; e < per; ++e) {
      const int v = a[t * per + e];
      a[t * per + e] = run;
      run += v;
    }
    __syncthreads();
  };

  // ---- x slabs
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int k = t + i * NT;
    if (k < n) atomicAdd(&hx[binx(bxv[i])], 1);
  }
  __syncthreads();
  scan(hx, FPS_BINS);
  unsigned long long slabs_lo = 0ULL, slabs_hi = 0ULL;   // 3 bits per owned point (NPT <= 40)
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int k = t + i * NT;
    if (k < n) {
      const int q = atomicAdd(&hx[binx(bxv[i])], 1);
      const unsigned long long sl = (unsigned long long)min(q / T1, 7);
      if (i < 21) slabs_lo |= sl << (3 * i); else slabs_hi |= sl << (3 * (i - 21));
      atomicAdd(&hy[(int)sl * YB + biny(byv[i])], 1);
    }
  }
  __syncthreads();
  for (int sl = 0; sl < 8; ++sl) scan(hy + sl * YB, YB);
#pragma unroll
  for (int i = 0; i < NPT; ++i) {
    const int k = t + i * NT;
    if (k < n) {
      const int sl = (int)((i < 21 ? slabs_lo >> (3 * i) : slabs_hi >> (3 * (i - 21))) & 7ULL);
      const int q2 = atomicAdd(&hy[sl * YB + biny(byv[i])], 1);
      const int cy = min(q2 / T2, 7);
      const int r = min(q2 - cy * T2, 64 * SG - 1);
      const int w = (sl + 3 * cy) & 7;                 // region (sl,cy) -> wave, group cy
      perm[w * CELL + (cy * SG + r / 64) * 64 + (r & 63)] = (int)(((0xffffu - (unsigned)rank_of(k)) << 16) | (unsigned)k);
    }
  }
  __syncthreads();

  // ---- this lane's points; lanes 0..7 keep the state of the wave's 8 groups
  float px[PPT], py[PPT], pz[PPT], tmp[PPT];
  float gx0 = INFINITY, gx1 = -INFINITY, gy0 = INFINITY, gy1 = -INFINITY, gz0 = INFINITY, gz1 = -INFINITY;
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float x0 = INFINITY, x1_ = -INFINITY, y0 = INFINITY, y1_ = -INFINITY, z0 = INFINITY, z1_ = -INFINITY;
#pragma unroll
    for (int q = 0; q < SG; ++q) {
      const int s = g * SG + q;
      const int pk = perm[wave * CELL + s * 64 + lane];
      const int k = pk & 0xffff;
      float x = 0.f, y = 0.f, z = 0.f, t0 = -1.0f;
      if (pk != 0) {
        x = xyz[k * 3 + 0]; y = xyz[k * 3 + 1]; z = xyz[k * 3 + 2];
        const float mag = fp_sumsq3(x, y, z);
        t0 = ((double)mag <= 1e-3) ? -1.0f : 1e10f;     // sampling_gpu.cu:105-106, sampling.cpp:78-80
        x0 = fminf(x0, x); x1_ = fmaxf(x1_, x); y0 = fminf(y0, y); y1_ = fmaxf(y1_, y);
        z0 = fminf(z0, z); z1_ = fmaxf(z1_, z);
      }
      px[s] = x; py[s] = y; pz[s] = z; tmp[s] = t0;
    }
    x0 = -wave_max(-x0); x1_ = wave_max(x1_); y0 = -wave_max(-y0); y1_ = wave_max(y1_);
    z0 = -wave_max(-z0); z1_ = wave_max(z1_);
    if (lane == g) { gx0 = x0; gx1 = x1_; gy0 = y0; gy1 = y1_; gz0 = z0; gz1 = z1_; }
  }
  float gmax = INFINITY;       // cached max of the group's running distances (inf = not computed yet)
  unsigned gkeylo = 0u;        // key word of the group's candidate: (0xffff - rank) << 16 | index
  bool gtie = false;           // the group's max is shared by different locations

  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (t == 0) {
    idx[0] = 0;
    if (new_xyz) { new_xyz[0] = x1; new_xyz[1] = y1; new_xyz[2] = z1; }
  }
  if (t < 3) s_key[t] = 0ULL;
  __syncthreads();

  int first_tie = 0x7fffffff;
  FPS_PROBE_INIT();
  for (int j = 1; j < m; ++j) {
    FPS_PROBE(5);
    // ---- which of this wave's groups can change?  (lane g tests group g)
    const float ex = fmaxf(fmaxf(gx0 - x1, x1 - gx1), 0.f);
    const float ey = fmaxf(fmaxf(gy0 - y1, y1 - gy1), 0.f);
    const float ez = fmaxf(fmaxf(gz0 - z1, z1 - gz1), 0.f);
    const float lb = (ex * ex + ey * ey + ez * ez) * 0.99999f;
    const unsigned amask = (unsigned)__ballot(lane < G && !(lb >= gmax)) & 0xffu;
    FPS_PROBE(0);
    FPS_PROBE_ADD(6, __popc(amask));
#pragma unroll
    for (int g = 0; g < G; ++g) {
      if (__builtin_expect(!((amask >> g) & 1u), 1)) continue;      // wave-uniform; a wave has 0.47 active groups per step on
                                                                     // average: the fall-through path is the skip
      float best = -1.0f;
      int bslot = 0;
      bool tie = false;
#pragma unroll
      for (int q = 0; q < SG; ++q) {
        const int s = g * SG + q;
        const float dx = px[s] - x1, dy = py[s] - y1, dz = pz[s] - z1;
        const float d = fp_sumsq3(dx, dy, dz);
        const float d2 = vmin(d, tmp[s]);
        tmp[s] = d2;
        tie = tie || (d2 == best);
        const bool gt = d2 > best;
        best = gt ? d2 : best;
        bslot = gt ? s : bslot;
      }
      // every lane asks for its own candidate's key word now: the LDS round trip runs under the DPP reduction instead of
      // after it (only the winning lane's word is used, and only on the common no-tie path)
      const int kc = perm[wave * CELL + bslot * 64 + lane];
      const float wmax = wave_max_fast(best);
      const bool mine = best == wmax;
      const unsigned long long eq = __ballot(mine);
      int wrank = 0;
      bool ctie = false;
      if (wmax >= 0.0f) {
        if (__popcll(eq) > 1 || __ballot(mine && tie)) {
          // rare: several points of the group share the maximum -> reference order = smallest rank;
          // duplicates of one location are harmless, different locations are recorded as a tie
          int myr = 0x7fffffff;
          float lx0 = INFINITY, lx1 = -INFINITY, ly0 = INFINITY, ly1 = -INFINITY, lz0 = INFINITY, lz1 = -INFINITY;
#pragma unroll
          for (int q = 0; q < SG; ++q) {
            const int s = g * SG + q;
            if (tmp[s] == wmax) {
              const int pk = perm[wave * CELL + s * 64 + lane];
              if (pk != 0) {
                myr = min(myr, (int)~(unsigned)pk);      // ~key: smallest = lowest rank
                lx0 = fminf(lx0, px[s]); lx1 = fmaxf(lx1, px[s]); ly0 = fminf(ly0, py[s]); ly1 = fmaxf(ly1, py[s]);
                lz0 = fminf(lz0, pz[s]); lz1 = fmaxf(lz1, pz[s]);
              }
            }
          }
          wrank = wave_min_i(myr);
          ctie = (-wave_max(-lx0) != wave_max(lx1)) || (-wave_max(-ly0) != wave_max(ly1)) ||
                 (-wave_max(-lz0) != wave_max(lz1));
        } else {
          const int src = __ffsll((long long)eq) - 1;
          wrank = (int)~(unsigned)__builtin_amdgcn_readlane(kc, src);
        }
      }
      if (lane == g) {
        gmax = wmax;
        gkeylo = wmax >= 0.0f ? (unsigned)(~wrank) : 0u;
        gtie = ctie;
      }
    }
    FPS_PROBE(1);
    // ---- every group folds its own candidate into the iteration's LDS word (lanes 0..7 of each wave, one ds_max_u64
    // with eight lanes on one address): the phase probe put the former per-wave pre-reduction -- two DPP chains, two
    // readfirstlanes, a ballot -- at ~440 cycles of the ~2200-cycle iteration, on every wave, every iteration.
    // fps.hip is compiled with the atomic optimizer off (csrc/Makefile): it would rebuild that reduction around the atomic.
    const int slot = j % 3;
    if (lane < G) {
      const unsigned long long key_l =
          gmax >= 0.0f ? (((unsigned long long)__float_as_uint(gmax) << 32) | gkeylo) : 0ULL;   // all skipped: loses to anything
      atomicMax(&s_key[slot], key_l);
    }
    if (t == 0) s_key[(j + 1) % 3] = 0ULL;
    FPS_PROBE(2);
    __syncthreads();
    FPS_PROBE(3);
    const unsigned long long gk = s_key[slot];
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gk);
    const unsigned ghi = __builtin_amdgcn_readfirstlane((unsigned)(gk >> 32));
    const int k = gk != 0ULL ? (int)(glo & 0xffffu) : 0;
    x1 = xyz[k * 3 + 0]; y1 = xyz[k * 3 + 1]; z1 = xyz[k * 3 + 2];
    FPS_PROBE_USE(x1 + y1 + z1);
    FPS_PROBE(4);
    // ---- was this step a tie between DIFFERENT locations?  Every group (of every wave) whose maximum equals the winning
    // value and whose candidate is not the winner itself is looked at: a tie inside the group was settled when the group
    // was reduced (gtie); otherwise its candidate's coordinates are fetched and compared with the winner's.  A duplicate
    // of the winning location in another region (the equal-population cuts are by ticket, so two copies of one point can
    // land on either side of a cut) is harmless -- either copy gives the same sample -- and is no longer reported: the
    // sampled-with-replacement `surface` clouds were flagged in 6 of 8 scenes by the per-wave test this replaces, with no
    // tie between distinct points in any of them, and lost the SA2-4 prefix shortcut (0.95 ms of chains per batch).
    if (gk == 0ULL) {
      first_tie = min(first_tie, j);
    } else {
      const bool cand = lane < G && __float_as_uint(gmax) == ghi && (gtie || gkeylo != glo);
      if (__ballot(cand)) {           // rare
        bool differs = false;
        if (cand) {
          differs = gtie;
          if (!differs) {
            const int kk = (int)(gkeylo & 0xffffu);
            differs = xyz[kk * 3 + 0] != x1 || xyz[kk * 3 + 1] != y1 || xyz[kk * 3 + 2] != z1;
          }
        }
        if (__ballot(differs)) first_tie = min(first_tie, j);
      }
    }
    if (t == 0) {
      idx[j] = k;
      if (new_xyz) { new_xyz[j * 3 + 0] = x1; new_xyz[j * 3 + 1] = y1; new_xyz[j * 3 + 2] = z1; }
    }
  }
  FPS_PROBE_FLUSH();
  if (tie_out) {
    __shared__ int s_tie;
    if (t == 0) s_tie = 0x7fffffff;
    __syncthreads();
    if (lane == 0) atomicMin(&s_tie, first_tie);
    __syncthreads();
    if (t == 0) tie_out[b] = s_tie;
  }
}

// Large clouds (24 576 < positions <= 65 536): the running min-distances still live in registers for
// the whole chain (PPT per lane); the coordinates are tiered: RS slots per lane in registers, LS
// slots in LDS (SoA [slot][thread], conflict-free), the rest streamed every iteration from a
// rank-ordered SoA copy in the workspace (lane-consecutive addresses -> fully coalesced, L2-resident).
// Thread t, slot s holds rank r = s*1024 + t, so a tie is resolved by the smallest r explicitly.
template <int PPT, int RS, int LS>
__global__ __launch_bounds__(1024) void fps_tier_kernel(const float *__restrict__ xyz, int n, int m, int bs, int lg,
                                                        int Q, unsigned qmagic, float *__restrict__ ws,
                                                        int32_t *__restrict__ idx, float *__restrict__ new_xyz) {
  constexpr int NT = 1024, GS = PPT - RS - LS, CAP = NT * PPT;
  static_assert(GS >= 0, "tiers exceed the slots");
  extern __shared__ float s_xyz[];                 // [3][LS][NT]
  __shared__ unsigned long long s_key[3];
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  idx += (size_t)b * m;
  if (new_xyz) new_xyz += (size_t)b * m * 3;
  // one buffer descriptor for this cloud's SoA copy; a slot is addressed as voffset = 4*t (one VGPR for
  // every slot) + a compile-time scalar offset, so no per-slot address registers are kept live
  float *gbase = ws + (size_t)b * 3 * CAP;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, 3 * CAP * 4, 0x00020000);
  const int t = threadIdx.x, lane = t & 63;
  // streamed slots are stored in groups of 4 per lane ([array][group][thread][4]) and read back as one
  // 16-byte buffer load per array and group
  static_assert(GS % 4 == 0, "streamed slots come in groups of 4");
  using f4 = __attribute__((ext_vector_type(4))) float;
  const int voff16 = t * 16;

  float rx[RS > 0 ? RS : 1], ry[RS > 0 ? RS : 1], rz[RS > 0 ? RS : 1], tmp[PPT];
#pragma unroll
  for (int s = 0; s < PPT; ++s) {
    const int k = pos_to_k(s * NT + t, bs, lg, Q, qmagic, n);
    float x = 0.f, y = 0.f, z = 0.f, t0 = -1.0f;
    if (k >= 0) {
      x = xyz[k * 3 + 0]; y = xyz[k * 3 + 1]; z = xyz[k * 3 + 2];
      const float mag = fp_sumsq3(x, y, z);
      t0 = ((double)mag <= 1e-3) ? -1.0f : 1e10f;
    }
    tmp[s] = t0;
    if (s < RS) { rx[s] = x; ry[s] = y; rz[s] = z; }
    else if (s < RS + LS) {
      s_xyz[(0 * LS + (s - RS)) * NT + t] = x;
      s_xyz[(1 * LS + (s - RS)) * NT + t] = y;
      s_xyz[(2 * LS + (s - RS)) * NT + t] = z;
    } else {
      // each lane re-reads only its own entries
      const int g = (s - RS - LS) >> 2, c = (s - RS - LS) & 3;
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, x), rsrc, voff16 + c * 4, (0 * CAP + g * NT * 4) * 4, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, y), rsrc, voff16 + c * 4, (1 * CAP + g * NT * 4) * 4, 0);
      __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, z), rsrc, voff16 + c * 4, (2 * CAP + g * NT * 4) * 4, 0);
    }
  }
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (t == 0) {
    idx[0] = 0;
    if (new_xyz) { new_xyz[0] = x1; new_xyz[1] = y1; new_xyz[2] = z1; }
  }
  if (t < 3) s_key[t] = 0ULL;
  __syncthreads();

  for (int j = 1; j < m; ++j) {
    float best = -1.0f;
    int bslot = 0;
    f4 vx = {0.f, 0.f, 0.f, 0.f}, vy = vx, vz = vx;
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
      float x, y, z;
      if (s < RS) { x = rx[s]; y = ry[s]; z = rz[s]; }
      else if (s < RS + LS) {
        x = s_xyz[(0 * LS + (s - RS)) * NT + t];
        y = s_xyz[(1 * LS + (s - RS)) * NT + t];
        z = s_xyz[(2 * LS + (s - RS)) * NT + t];
      } else {
        const int g = (s - RS - LS) >> 2, c = (s - RS - LS) & 3;
        if (c == 0) {
          vx = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff16, (0 * CAP + g * NT * 4) * 4, 0));
          vy = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff16, (1 * CAP + g * NT * 4) * 4, 0));
          vz = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff16, (2 * CAP + g * NT * 4) * 4, 0));
        }
        x = vx[c]; y = vy[c]; z = vz[c];
      }
      // bound the scheduler's load hoisting to one group: the loop is fully unrolled for static
      // register indexing and would otherwise prefetch every slot and spill
      if (s >= RS && ((s - RS) & 3) == 3) __builtin_amdgcn_sched_barrier(0);
      const float dx = x - x1, dy = y - y1, dz = z - z1;
      const float d = fp_sumsq3(dx, dy, dz);
      const float d2 = vmin(d, tmp[s]);
      tmp[s] = d2;
      const bool gt = d2 > best;
      best = gt ? d2 : best;
      bslot = gt ? s : bslot;
    }
    const float rmax = row16_max_fast(best);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rmax), 48));
    const float wmax = fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
    const int wrank = wave_min_i(best == wmax ? bslot * NT + t : 0x7fffffff);   // smallest rank among equal maxima
    const int slot = j % 3;
    if (lane == 0) {
      const unsigned long long key =
          wmax >= 0.0f ? (((unsigned long long)__float_as_uint(wmax) << 32) | (unsigned)(~wrank)) : 0ULL;
      atomicMax(&s_key[slot], key);
      if (t == 0) s_key[(j + 1) % 3] = 0ULL;
    }
    __syncthreads();
    const unsigned long long gk = s_key[slot];
    const unsigned glo = __builtin_amdgcn_readfirstlane((unsigned)gk);
    int k = gk != 0ULL ? pos_to_k((int)~glo, bs, lg, Q, qmagic, n) : 0;
    k = __builtin_amdgcn_readfirstlane(k < 0 ? 0 : k);
    x1 = xyz[k * 3 + 0]; y1 = xyz[k * 3 + 1]; z1 = xyz[k * 3 + 2];
    if (t == 0) {
      idx[j] = k;
      if (new_xyz) { new_xyz[j * 3 + 0] = x1; new_xyz[j * 3 + 1] = y1; new_xyz[j * 3 + 2] = z1; }
    }
  }
}

// Any-n fallback: running min-distances live in the HBM workspace (the
// reference's `temp`), the cloud is re-read (L2-resident) every iteration.
// Ties are resolved with the same rank order, compared explicitly.
__global__ __launch_bounds__(1024) void fps_global_kernel(const float *__restrict__ xyz, int n, int m,
                                                           int bs, int lg, int Q,
                                                           float *__restrict__ temp,
                                                           int32_t *__restrict__ idx,
                                                           float *__restrict__ new_xyz) {
  constexpr int NT = 1024, NW = NT / 64;
  __shared__ float s_val[2][16];
  __shared__ int s_rank[2][16];
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  temp += (size_t)b * n;
  idx += (size_t)b * m;
  if (new_xyz) new_xyz += (size_t)b * m * 3;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bmask = bs - 1;

  for (int k = t; k < n; k += NT) {
    const float x = xyz[k * 3 + 0], y = xyz[k * 3 + 1], z = xyz[k * 3 + 2];
    const float mag = fp_sumsq3(x, y, z);
    temp[k] = ((double)mag <= 1e-3) ? -1.0f : 1e10f;
  }
  float x1 = xyz[0], y1 = xyz[1], z1 = xyz[2];
  if (t == 0) {
    idx[0] = 0;
    if (new_xyz) { new_xyz[0] = x1; new_xyz[1] = y1; new_xyz[2] = z1; }
  }
  // each thread only ever touches its own temp[k] entries: no barrier needed for temp
  for (int j = 1; j < m; ++j) {
    float best = -1.0f;
    int brank = 0x7fffffff;
    for (int k = t; k < n; k += NT) {
      const float dx = xyz[k * 3 + 0] - x1, dy = xyz[k * 3 + 1] - y1, dz = xyz[k * 3 + 2] - z1;
      const float d = fp_sumsq3(dx, dy, dz);
      const float d2 = __builtin_fminf(d, temp[k]);
      temp[k] = d2;
      const int rank = bitrev_bits(k & bmask, lg) * Q + (k >> lg);
      const bool take = (d2 > best) || (d2 == best && rank < brank);
      best = take ? d2 : best;
      brank = take ? rank : brank;
    }
    const float wmax = wave_max(best);
    const int wrank = wave_min_i(best == wmax ? brank : 0x7fffffff);
    const int buf = j & 1;
    if (lane == 0) { s_val[buf][wave] = wmax; s_rank[buf][wave] = wrank; }
    __syncthreads();
    const float v = lane < NW ? s_val[buf][lane] : -2.0f;
    const int rr = lane < NW ? s_rank[buf][lane] : 0x7fffffff;
    const float gmax = __builtin_bit_cast(
        float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, row16_max(v))));
    const int grank = __builtin_amdgcn_readfirstlane(row16_min_i(v == gmax ? rr : 0x7fffffff));
    int k = 0;
    if (gmax >= 0.0f && grank != 0x7fffffff) {
      const int r = grank / Q, q = grank - r * Q;
      k = q * bs + bitrev_bits(r, lg);
    }
    x1 = xyz[k * 3 + 0];
    y1 = xyz[k * 3 + 1];
    z1 = xyz[k * 3 + 2];
    if (t == 0) {
      idx[j] = k;
      if (new_xyz) { new_xyz[j * 3 + 0] = x1; new_xyz[j * 3 + 1] = y1; new_xyz[j * 3 + 2] = z1; }
    }
  }
}

static const int kRegCapacity = 24576;  // largest NT*PPT instantiated below

// GN_FPS_CELL=<16|32>: take the cell kernel (fps_cell.hip) with that many regions per axis for clouds the register
// kernels would serve too (A/B measurements, tests); GN_FPS_CELL=0: never take it (the tier / any-n kernels run)
static int fps_cell_env() {
  const char *e = getenv("GN_FPS_CELL");       // read per call: tests flip it inside one process
  return e ? atoi(e) : -1;
}
static int fps_cell_forced(int n) {
  const int v = fps_cell_env();
  return (v == 16 || v == 32) && n >= 1024 && fps_cell_supports(n, v) ? v : 0;
}

}  // namespace GN_FPNS
}  // namespace gn
using namespace gn::GN_FPNS;

#if GN_FP == 0      // a bitwise compare and a size: one copy
namespace gn {
// Is `pts2` (n2 points per cloud) exactly the points a previous sampling call picked, in its order -- pts1[inds1[0 .. n2)]?
// Then the tie step that call reported carries over (tie2 = tie1), else the new cloud is unrelated (tie2 = 0: no promise).
// One workgroup per cloud, bitwise compare.
__global__ __launch_bounds__(256) void fps_prefix_check_kernel(const float *__restrict__ pts2, const float *__restrict__ pts1,
                                                               const int32_t *__restrict__ inds1, const int32_t *__restrict__ tie1,
                                                               int n2, int n1, int m1, int32_t *__restrict__ tie2) {
  __shared__ int s_bad;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) s_bad = 0;
  __syncthreads();
  const unsigned *p2 = reinterpret_cast<const unsigned *>(pts2) + (size_t)b * n2 * 3;
  const unsigned *p1 = reinterpret_cast<const unsigned *>(pts1) + (size_t)b * n1 * 3;
  const int32_t *ix = inds1 + (size_t)b * m1;
  bool bad = false;
  for (int i = threadIdx.x; i < n2; i += 256) {
    const int k = ix[i];
    if (k < 0 || k >= n1) { bad = true; continue; }
    bad |= p2[i * 3] != p1[k * 3] || p2[i * 3 + 1] != p1[k * 3 + 1] || p2[i * 3 + 2] != p1[k * 3 + 2];
  }
  if (bad) s_bad = 1;
  __syncthreads();
  if (threadIdx.x == 0) tie2[b] = s_bad ? 0 : tie1[b];
}
}  // namespace gn

extern "C" int gn_fps_prefix_check(const float *pts2, const float *pts1, const int32_t *inds1, const int32_t *tie1, int b, int n2,
                                   int n1, int m1, int32_t *tie2, gn_stream_t stream) {
  using namespace gn;
  GN_REQUIRE(b >= 0 && n2 >= 1 && n1 >= 1 && m1 >= n2, "fps_prefix_check: bad sizes (the earlier call must have picked >= n2 samples)");
  if (b == 0) return GN_OK;
  GN_REQUIRE(pts2 && pts1 && inds1 && tie1 && tie2, "fps_prefix_check: null pointer");
  hipLaunchKernelGGL(fps_prefix_check_kernel, dim3(b), dim3(256), 0, as_stream(stream), pts2, pts1, inds1, tie1, n2, n1, m1, tie2);
  GN_CHECK_LAUNCH("fps_prefix_check");
  return GN_OK;
}

// 1 when asking gn_furthest_point_sampling for `tie_out` does not change which kernel serves n points (the region kernel's
// band, clouds beyond the register kernels, and the small sizes whose kernel reports 0 anyway); 0 when it moves the call
// from a plain register kernel onto the cell kernel (4 096 < positions <= 24 576 outside the band: 2.54 vs 2.42 ms at 10 000)
extern "C" int gn_fps_tie_is_free(int n) {
  using namespace gn;
  if (n <= 0) return 1;
  const int bs = fps_ref_block_size(n);
  const long need = (long)bs * ((n + bs - 1) / bs);
  const char *ve = getenv("GN_FPS_VARIANT");
  const int variant = ve ? atoi(ve) : 4;
  const bool region_band = need > 1024 * 16 && need <= 1024 * 20 && (variant == 4 || variant == 3);
  return (need <= 4096 || need > kRegCapacity || region_band || fps_cell_forced(n)) ? 1 : 0;
}

extern "C" size_t gn_fps_workspace_bytes(int b, int n) {
  using namespace gn;
  if (b <= 0 || n <= 0) return 0;
  const int bs = fps_ref_block_size(n);
  const int Q = (n + bs - 1) / bs;
  const long need = (long)bs * Q;
  if (need <= 4096 && !fps_cell_forced(n)) return 0;
  const int forced = fps_cell_forced(n);
  if (need <= kRegCapacity) {
    // the region / spatial kernels of the model's own 20 000-point level never touch a workspace
    const char *ve = getenv("GN_FPS_VARIANT");
    const int variant = ve ? atoi(ve) : 4;
    const bool region_band = need > 1024 * 16 && need <= 1024 * 20 && (variant == 4 || variant == 3);
    if (region_band && !forced) return 0;
    return (size_t)b * fps_cell_workspace_bytes_one(n, forced);            // tie-tracking calls
  }
  size_t one = (size_t)n * sizeof(float);                                       // any-n kernel: the reference's `temp`
  if (need <= 1024L * 64) one = 3 * 1024 * 64 * sizeof(float);                  // rank-ordered SoA copy (tier kernel)
  if (fps_cell_supports(n) || forced) {
    const size_t cell = fps_cell_workspace_bytes_one(n, forced);            // GN_FPS_CELL=16|32 strides by its own S
    one = one > cell ? one : cell;
  }
  return (size_t)b * one;
}

#endif

extern "C" int GN_FP_ENTRY(gn_furthest_point_sampling)(const float *xyz, int b, int n, int m, void *workspace,
                                          int32_t *idx, float *new_xyz, int32_t *tie_out,
                                          const int32_t *prefix_ok, gn_stream_t stream) {
  using namespace gn;
  GN_FP_DISPATCH(gn_furthest_point_sampling, xyz, b, n, m, workspace, idx, new_xyz, tie_out, prefix_ok, stream);
  GN_REQUIRE(b >= 0 && n >= 1 && m >= 0, "furthest_point_sampling: bad sizes b=%d n=%d m=%d", b, n, m);
  if (b == 0 || m == 0) return GN_OK;
  GN_REQUIRE(xyz && idx, "furthest_point_sampling: null pointer");
  GN_REQUIRE((long)n * 3 < 0x7fffffffL, "furthest_point_sampling: n too large");
  const int bs = fps_ref_block_size(n);
  int lg = 0;
  while ((1 << lg) < bs) ++lg;
  const int Q = (n + bs - 1) / bs;
  const long need = (long)bs * Q;
  // ceil(2^32 / Q); exact floor-division for every position < 2^15 (register kernels hold <= 24576)
  const unsigned qmagic = Q == 1 ? 0xffffffffu : (unsigned)((0x100000000ull + Q - 1) / Q);
  hipStream_t s = as_stream(stream);
  bool tie_written = false;
  const char *ve = getenv("GN_FPS_VARIANT");       // read per call: tests and probes flip it inside one process
  const int variant = ve ? atoi(ve) : 4;
  // the cell kernel (fps_cell.hip) serves every cloud beyond the register kernels' 24 576 positions, and -- when the
  // caller wants the tie report (the fused path: it lets SA2-4 return the identity prefix, 0.86 ms of chains) -- also the
  // sizes the plain register kernels serve without tracking ties (measured at n = 10 000: 2.54 vs 2.42 ms); the model's own
  // 20 000-point level keeps fps_region_kernel (1.69 ms; the cell kernel takes 2.7 there)
  const int cell_s = fps_cell_forced(n);
  const bool region_band = need > 1024 * 16 && need <= 1024 * 20 && (variant == 4 || variant == 3);
  const bool cell_for_ties = tie_out != nullptr && workspace != nullptr && need > 4096 && need <= kRegCapacity && !region_band;
  if (cell_s || (fps_cell_env() != 0 && fps_cell_supports(n) && (need > kRegCapacity || cell_for_ties))) {
    GN_REQUIRE(workspace != nullptr, "furthest_point_sampling: workspace required for n=%d", n);
    const int rc = fps_cell_launch(xyz, b, n, m, bs, lg, Q, qmagic, workspace, idx, new_xyz, tie_out, cell_s, s);
    if (rc != GN_OK) return rc;
    GN_CHECK_LAUNCH("furthest_point_sampling");
    return GN_OK;
  }
#define GN_FPS_LAUNCH(NT, PPT)                                                                    \
  hipLaunchKernelGGL((fps_reg_kernel<NT, PPT>), dim3(b), dim3(NT), (NT) * (PPT) <= 4096 ? (size_t)n * 3 * sizeof(float) : 0, \
                     s, xyz, n, m, bs, lg, Q, qmagic, idx, new_xyz, prefix_ok)
  if (need <= 256 * 2) GN_FPS_LAUNCH(256, 2);
  else if (need <= 256 * 4) GN_FPS_LAUNCH(256, 4);
  // 2048 positions: 4 waves x 8 points beat 16 waves x 2 (521 vs 598 ns per iteration, profiles/README.md: the
  // iteration is a serial chain of ~130 instructions and three LDS round trips, and the barrier grows with the waves)
  else if (need <= 256 * 8) GN_FPS_LAUNCH(256, 8);
  else if (need <= 1024 * 4) GN_FPS_LAUNCH(1024, 4);
  else if (need <= 1024 * 8) GN_FPS_LAUNCH(1024, 8);
  else if (need <= 1024 * 12) GN_FPS_LAUNCH(1024, 12);
  else if (need <= 1024 * 16) GN_FPS_LAUNCH(1024, 16);
#define GN_FPS_SPATIAL(PPT)                                                                                 \
  do {                                                                                                      \
    auto kern = fps_spatial_kernel<PPT>;                                                                    \
    const int lds = (16 * 64 * PPT + 5 * FPS_BINS + 64) * (int)sizeof(int);                                 \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            lds) != hipSuccess) {                                                           \
      set_error("furthest_point_sampling: cannot reserve %d bytes of LDS", lds);                            \
      return GN_ERR_LAUNCH;                                                                                 \
    }                                                                                                       \
    hipLaunchKernelGGL(kern, dim3(b), dim3(1024), lds, s, xyz, n, m, bs, lg, Q, qmagic, idx, new_xyz,      \
                       tie_out);                                                                            \
    tie_written = true;                                                                                     \
  } while (0)
#define GN_FPS_REGION(PPT)                                                                                  \
  do {                                                                                                      \
    auto kern = fps_region_kernel<PPT>;                                                                     \
    const int lds = (8 * 64 * PPT + FPS_BINS + 8 * 1024 + 64) * (int)sizeof(int);                           \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            lds) != hipSuccess) {                                                           \
      set_error("furthest_point_sampling: cannot reserve %d bytes of LDS", lds);                            \
      return GN_ERR_LAUNCH;                                                                                 \
    }                                                                                                       \
    hipLaunchKernelGGL(kern, dim3(b), dim3(512), lds, s, xyz, n, m, bs, lg, Q, qmagic, idx, new_xyz, tie_out); \
    tie_written = true;                                                                                     \
  } while (0)
  // the region / spatial kernels (tie tracking -> SA2-4 prefix shortcut) serve 16 384 < need <= 20 480, i.e. the model's
  // 20 000-point level; up to 16 384 positions the register kernel above runs and reports "tie unknown" (memset below),
  // so deeper levels run their full chains (correct, only slower on those off-design cloud sizes)
  else if (need > 1024 * 16 && need <= 1024 * 20 && variant == 4) GN_FPS_REGION(40);
  else if (need > 1024 * 16 && need <= 1024 * 20 && variant == 3) GN_FPS_SPATIAL(20);
  else if (need <= 512 * 40 && variant == 1) GN_FPS_LAUNCH(512, 40);
  else if (need <= 256 * 80 && variant == 2) GN_FPS_LAUNCH(256, 80);
  else if (need <= 1024 * 20) GN_FPS_LAUNCH(1024, 20);
  else if (need <= 1024 * 24) GN_FPS_LAUNCH(1024, 24);
  else if (need <= 1024L * 64) {
    GN_REQUIRE(workspace != nullptr, "furthest_point_sampling: workspace required for n=%d", n);
#define GN_FPS_TIER(PPT, RS, LS)                                                                          \
  do {                                                                                                    \
    auto kern = fps_tier_kernel<PPT, RS, LS>;                                                             \
    const int lds = 3 * LS * 1024 * (int)sizeof(float);                                                   \
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                            lds) != hipSuccess) {                                                         \
      set_error("furthest_point_sampling: cannot reserve %d bytes of LDS", lds);                          \
      return GN_ERR_LAUNCH;                                                                               \
    }                                                                                                     \
    hipLaunchKernelGGL(kern, dim3(b), dim3(1024), lds, s, xyz, n, m, bs, lg, Q, qmagic, (float *)workspace, idx, \
                       new_xyz);                                                                          \
  } while (0)
    if (need <= 1024 * 32) GN_FPS_TIER(32, 15, 13);
    else if (need <= 1024 * 49) GN_FPS_TIER(49, 8, 13);
    else GN_FPS_TIER(64, 0, 12);
#undef GN_FPS_TIER
  } else {
    GN_REQUIRE(workspace != nullptr, "furthest_point_sampling: workspace required for n=%d", n);
    hipLaunchKernelGGL(fps_global_kernel, dim3(b), dim3(1024), 0, s, xyz, n, m, bs, lg, Q,
                       (float *)workspace, idx, new_xyz);
  }
#undef GN_FPS_LAUNCH
  // kernels without tie tracking report "unknown" = a tie at step 0
  if (tie_out && !tie_written && hipMemsetAsync(tie_out, 0, sizeof(int32_t) * (size_t)b, s) != hipSuccess) {
    set_error("furthest_point_sampling: memset failed");
    return GN_ERR_LAUNCH;
  }
  GN_CHECK_LAUNCH("furthest_point_sampling");
  return GN_OK;
}
