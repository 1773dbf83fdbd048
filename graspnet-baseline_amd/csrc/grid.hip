// grid.hip -- uniform-grid acceleration of the neighbour queries (ball / cylinder), exact semantics.
//
// The reference's queries return the FIRST nsample hits IN INDEX ORDER (ball_query_gpu.cu:30-47,
// cylinder_query_gpu.cu:35-54), which is why query.hip scans the whole cloud in order.  Here the
// cloud is binned once per level into a uniform grid whose cells are at least `reach` wide (reach =
// ball radius, or the radius of the sphere that contains the cylinder), so every hit of a centre lies
// in the 3x3x3 cells around it (40x fewer candidates at N=20 000, r=0.04).  Candidates are then visited
// in grid order, NOT index order, so each wave keeps the nsample smallest hit indices seen so far as
// a sorted 64-lane register (bitonic sort of each batch of 64 hits + bitonic merge, all by shuffles);
// the final register is exactly the reference's list.  The hit predicate is evaluated on the same
// float expression as the reference.
#include "gn_common.h"

namespace gn {
namespace GN_FPNS {

constexpr int GRID_MAX_DIM = 32;                      // cells per axis (x,y); z uses GRID_MAX_DIMZ
constexpr int GRID_MAX_DIMZ = 16;
constexpr int GRID_MAX_CELLS = GRID_MAX_DIM * GRID_MAX_DIM * GRID_MAX_DIMZ;   // 16384
constexpr int GRID_CS_STRIDE = GRID_MAX_CELLS + 4;     // cell_start entries per cloud (GRID_MAX_CELLS + 1 used; keeps the points 16-byte aligned)
using gf4 = __attribute__((ext_vector_type(4))) float;

// per-cloud header written by the build kernel (floats/ints, 16 words)
struct GridHeader {
  float minx, miny, minz;
  float invx, invy, invz;    // 1 / cell size per axis
  int dx, dy, dz;            // cells per axis
  int pad[7];
};

__device__ __forceinline__ int cell_coord(float v, float mn, float inv, int dim) {
  const float f = (v - mn) * inv;                 // NaN -> cell 0, +-Inf / huge -> an edge cell (compared in float:
  return !(f >= 1.f) ? 0 : (f >= (float)dim ? dim - 1 : (int)f);   // the int conversion never sees a value out of range)
}

// a centre's cell from its UNCLAMPED coordinate (it may lie outside the cloud's box); a far or non-finite centre lands
// 2^20 cells outside, where its window is empty (as its hit list is), without leaving the range of int arithmetic
__device__ __forceinline__ int centre_cell(float v, float mn, float inv) {
  const float f = floorf((v - mn) * inv);
  return !(f > -1048576.f) ? -1048576 : (f < 1048576.f ? (int)f : 1048576);
}

// one workgroup per cloud: extent -> cell histogram (LDS) -> exclusive scan -> ticket scatter.
// outputs per cloud: header, cell_start[GRID_MAX_CELLS + 1], sorted xyz (n,3), sorted original index (n)
//
// Each of the three passes over the cloud takes its points UB at a time with all UB x 3 loads in flight before the first
// LDS atomic: written one point per trip, a pass was a chain of n / 1024 dependent L2 round trips (the atomic keeps the
// compiler from hoisting the next trip's loads) -- 3 x 20 of them at n = 20 000, 51 us.  (Holding all 20 points of a
// thread in registers across the passes was tried: 60 coordinates + the unrolled scatter spill at 128 registers.)
__global__ __launch_bounds__(1024) void grid_build_kernel(const float *__restrict__ xyz, int n, float reach,
                                                          GridHeader *__restrict__ hdr, int *__restrict__ cell_start,
                                                          float *__restrict__ sxyz, int32_t *__restrict__ sidx) {
  constexpr int UB = 10;        // points per thread and trip of the histogram / scatter passes (2 trips at n = 20 000)
  constexpr int UBB = 20;       // ... of the bounds pass, which keeps nothing but six running extremes
  __shared__ int s_cnt[GRID_MAX_CELLS];
  __shared__ float s_red[16 * 6];
  __shared__ int s_scan[16];
  const int b = blockIdx.x;
  xyz += (size_t)b * n * 3;
  hdr += b;
  cell_start += (size_t)b * GRID_CS_STRIDE;
  gf4 *spt = reinterpret_cast<gf4 *>(sxyz) + (size_t)b * n;     // cell-sorted (x, y, z, original index)
  (void)sidx;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // reads past the cloud return 0 (buffer bounds check): no branch around the loads
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xyz), 0, n * 12, 0x00020000);
  auto fetch = [&](int k0, auto &x, auto &y, auto &z) {
    constexpr int NU = sizeof(x) / sizeof(float);
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int off = (k0 + u * 1024) * 12;
      x[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, off, 0, 0));
      y[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, off, 4, 0));
      z[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, off, 8, 0));
    }
  };

  // the box of the FINITE coordinates: an Inf / NaN coordinate can never be a hit (every predicate on it is false, as in
  // the reference's scan), so it must not stretch the grid either; cell_coord clamps such a point into an edge cell
  float mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (int k0 = t; k0 < n; k0 += UBB * 1024) {
    float x[UBB], y[UBB], z[UBB];
    fetch(k0, x, y, z);
#pragma unroll
    for (int u = 0; u < UBB; ++u)
      if (k0 + u * 1024 < n) {
        const float v[3] = {x[u], y[u], z[u]};
#pragma unroll
        for (int a = 0; a < 3; ++a)
          if (fabsf(v[a]) < INFINITY) {
            mn[a] = fminf(mn[a], v[a]);
            mx[a] = fmaxf(mx[a], v[a]);
          }
      }
  }
  for (int a = 0; a < 3; ++a) {
    const float lo = -wave_max(-mn[a]), hi = wave_max(mx[a]);
    if (lane == 0) { s_red[wave * 6 + a] = lo; s_red[wave * 6 + 3 + a] = hi; }
  }
  for (int i = t; i < GRID_MAX_CELLS; i += 1024) s_cnt[i] = 0;
  __syncthreads();
  for (int a = 0; a < 3; ++a) {
    mn[a] = s_red[a]; mx[a] = s_red[3 + a];
    for (int w = 1; w < 16; ++w) { mn[a] = fminf(mn[a], s_red[w * 6 + a]); mx[a] = fmaxf(mx[a], s_red[w * 6 + 3 + a]); }
  }
  // cell size per axis: at least `reach` (+0.1 % so that rounding in the coordinate -> cell map can never
  // put a point within `reach` of a centre more than one cell away), larger if the axis would need
  // more than the table holds
  const float h = reach * 1.001f;
  int dim[3];
  float inv[3];
  const int cap[3] = {GRID_MAX_DIM, GRID_MAX_DIM, GRID_MAX_DIMZ};
  for (int a = 0; a < 3; ++a) {
    if (!(mn[a] <= mx[a])) mn[a] = mx[a] = 0.f;               // no finite coordinate on this axis
    const float ext = fminf(fmaxf(mx[a] - mn[a], 0.f), 3.0e38f);
    // cells this axis would need, compared in float: (int)(ext / h) + 1 overflows for a far outlier (ext / h >= 2^31)
    const float need = ext / h;
    int d = need < (float)cap[a] ? (int)need + 1 : cap[a] + 1;
    float cs = h;
    if (d > cap[a]) { d = cap[a]; cs = fmaxf(ext / (float)cap[a] * 1.001f, h); }
    dim[a] = d;
    inv[a] = 1.0f / cs;
  }
  if (t == 0) {
    hdr->minx = mn[0]; hdr->miny = mn[1]; hdr->minz = mn[2];
    hdr->invx = inv[0]; hdr->invy = inv[1]; hdr->invz = inv[2];
    hdr->dx = dim[0]; hdr->dy = dim[1]; hdr->dz = dim[2];
  }
  auto cell_at = [&](float x, float y, float z) {
    const int cx = cell_coord(x, mn[0], inv[0], dim[0]);
    const int cy = cell_coord(y, mn[1], inv[1], dim[1]);
    const int cz = cell_coord(z, mn[2], inv[2], dim[2]);
    return (cz * dim[1] + cy) * dim[0] + cx;   // x fastest: the 3 x-neighbours of a cell are contiguous
  };
  for (int k0 = t; k0 < n; k0 += UB * 1024) {
    float x[UB], y[UB], z[UB];
    fetch(k0, x, y, z);
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (k0 + u * 1024 < n) atomicAdd(&s_cnt[cell_at(x[u], y[u], z[u])], 1);
  }
  __syncthreads();
  // exclusive scan of GRID_MAX_CELLS counters (16 per thread)
  {
    constexpr int PER = GRID_MAX_CELLS / 1024;
    int loc = 0;
    for (int e = 0; e < PER; ++e) loc += s_cnt[t * PER + e];
    int incl = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += s_scan[w];
    int run = woff + incl - loc;
    for (int e = 0; e < PER; ++e) {
      const int v = s_cnt[t * PER + e];
      s_cnt[t * PER + e] = run;
      run += v;
    }
    if (t == 1023) cell_start[GRID_MAX_CELLS] = run;
    __syncthreads();
    // the table goes out lane-consecutive (each thread's own 16 entries are 64 bytes apart from its neighbour's: written in
    // place they were 16 store instructions of 64 cache lines each)
    for (int i = t; i < GRID_MAX_CELLS; i += 1024) cell_start[i] = s_cnt[i];
    __syncthreads();
  }
  for (int k0 = t; k0 < n; k0 += UB * 1024) {
    float x[UB], y[UB], z[UB];
    fetch(k0, x, y, z);
    int pos[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) pos[u] = k0 + u * 1024 < n ? atomicAdd(&s_cnt[cell_at(x[u], y[u], z[u])], 1) : 0;
#pragma unroll
    for (int u = 0; u < UB; ++u)
      if (k0 + u * 1024 < n) {
        // ONE 16-byte store per point: the scatter is bound by store instructions x cache lines touched (four 4-byte
        // stores per point, each to 64 different lines per wave, were 35 of the kernel's 51 us)
        const gf4 v = {x[u], y[u], z[u], __int_as_float(k0 + u * 1024)};
        spt[pos[u]] = v;
      }
  }
}

// ---- 64-lane sorted-set helpers (one key per lane, ascending across lanes) --------------------------
// One compare-exchange step of the sorting networks: lane l keeps min or max of (v[l], v[l ^ partner_mask]).  The partner
// comes through DPP, fused into the v_min / v_max, for the distances a 16-lane row can reach (1, 2: quad_perm; 8: row_ror;
// 4: row_shl for the banks whose partner lies above, row_shr for the others) -- 15 of the 21 steps of sort64 and 4 of the
// 6 of a merge; only distances 16 and 32 still take the LDS crossbar (ds_bpermute + a wait for its round trip, which
// every step did before: a chain of 28 dependent LDS round trips per fold).
#define GN_DPP_MINMAX(CTRL)                                                                                        \
  asm volatile("s_nop 1\n\tv_min_i32_dpp %0, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"                        \
               "v_max_i32_dpp %1, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf"                                       \
               : "=&v"(lo), "=&v"(hi)                                                                               \
               : "v"(v))
__device__ __forceinline__ int cmpx(int v, int partner_mask, bool keep_min, int lane) {
  (void)lane;
  int lo, hi;
  if (partner_mask == 1) {
    GN_DPP_MINMAX("quad_perm:[1,0,3,2]");
  } else if (partner_mask == 2) {
    GN_DPP_MINMAX("quad_perm:[2,3,0,1]");
  } else if (partner_mask == 8) {
    GN_DPP_MINMAX("row_ror:8");
  } else if (partner_mask == 4) {
    // every lane is written by exactly one of the two masked steps.  The outputs are early-clobber registers of their
    // own: as read-write copies of v the compiler merged them with v, and the second step then read lanes the first had
    // just overwritten.
    asm volatile("s_nop 1\n\t"
                 "v_min_i32_dpp %0, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_min_i32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                 "v_max_i32_dpp %1, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                 "v_max_i32_dpp %1, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(v));
  } else {
    const int o = __shfl_xor(v, partner_mask);
    lo = min(v, o);
    hi = max(v, o);
  }
  return keep_min ? lo : hi;
}
#undef GN_DPP_MINMAX
// full bitonic sort of 64 keys
__device__ __forceinline__ int sort64(int v, int lane) {
#pragma unroll
  for (int k = 2; k <= 64; k <<= 1)
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const bool up = (lane & k) == 0 || k == 64;
      const bool lower = (lane & j) == 0;
      v = cmpx(v, j, lower == up, lane);
    }
  return v;
}
// a, b ascending -> the 64 smallest of their union, ascending
__device__ __forceinline__ int merge_low64(int a, int b, int lane) {
  const int br = __shfl(b, 63 - lane);
  int v = min(a, br);               // bitonic sequence holding the 64 smallest
#pragma unroll
  for (int j = 32; j > 0; j >>= 1) v = cmpx(v, j, (lane & j) == 0, lane);
  return v;
}

constexpr int GQ_WAVES = 4;

// The cylinder kernels' cell windows assume that `rot` is a rotation: then |p - c| = |R^T (p - c)| < sqrt(radius^2 + h^2)
// and the first column is the cylinder's unit axis.  cylinder_query_gpu.cu:39-47 takes ANY 3x3, so each seed tests
// max |R^T R - I| on the device; a seed that fails (scaled / sheared / NaN matrix) walks every cell of the grid -- the
// sorted-set bookkeeping does not care where candidates come from, so its list is still the reference's.  2e-5 keeps the
// length error (<= 3e-5 relative) inside the windows' own margins (reach x 1.0001, cells x 1.001, box x 1.002); float
// rotation matrices from batch_viewpoint_params_to_matrix sit at ~1e-7.
__device__ __forceinline__ bool rot_is_orthonormal(float r0, float r1, float r2, float r3, float r4, float r5, float r6,
                                                   float r7, float r8) {
  const float g00 = r0 * r0 + r3 * r3 + r6 * r6 - 1.f, g11 = r1 * r1 + r4 * r4 + r7 * r7 - 1.f,
              g22 = r2 * r2 + r5 * r5 + r8 * r8 - 1.f;
  const float g01 = r0 * r1 + r3 * r4 + r6 * r7, g02 = r0 * r2 + r3 * r5 + r6 * r8, g12 = r1 * r2 + r4 * r5 + r7 * r8;
  const float e = fmaxf(fmaxf(fmaxf(fabsf(g00), fabsf(g11)), fmaxf(fabsf(g22), fabsf(g01))), fmaxf(fabsf(g02), fabsf(g12)));
  return e <= 2e-5f && fabsf(g00) <= 2e-5f && fabsf(g11) <= 2e-5f && fabsf(g22) <= 2e-5f && fabsf(g01) <= 2e-5f &&
         fabsf(g02) <= 2e-5f && fabsf(g12) <= 2e-5f;      // the second form is false for NaN whatever fmaxf returns
}

// Cell window [lo, hi] per axis that contains every point of the cylinder { p : hmin < a.(p-c) < hmax, |p-c - (a.(p-c))a| < radius }:
// its axis-aligned bounding box is centre c + a(hmin+hmax)/2, half-extent |a_i|(hmax-hmin)/2 + radius*sqrt(1-a_i^2) per axis
// (taken 0.2 % + 1e-6 wider: a is a float rotation column).  cell_coord is the monotone map the build kernel bins with, so a
// point inside [lo_i, hi_i] lies in a cell inside [cell(lo_i), cell(hi_i)].  ~2.6x fewer candidate cells than the bounding
// sphere's cube for CloudCrop's cylinders.
struct CellWindow { int x0, x1, y0, y1, z0, z1; };
__device__ __forceinline__ CellWindow cylinder_window(const GridHeader &H, float cx, float cy, float cz, float ax, float ay,
                                                      float az, float radius, float hmin, float hmax) {
  const float mid = 0.5f * (hmin + hmax), hl = 0.5f * (hmax - hmin);
  auto span = [&](float c, float a, float mn, float inv, int dim, int &lo, int &hi) {
    const float half = (fabsf(a) * hl + radius * sqrtf(fmaxf(0.f, 1.f - a * a))) * 1.002f + 1e-6f;
    const float ctr = c + a * mid;
    lo = cell_coord(ctr - half, mn, inv, dim);
    hi = cell_coord(ctr + half, mn, inv, dim);
  };
  CellWindow w;
  span(cx, ax, H.minx, H.invx, H.dx, w.x0, w.x1);
  span(cy, ay, H.miny, H.invy, H.dy, w.y0, w.y1);
  span(cz, az, H.minz, H.invz, H.dz, w.z0, w.z1);
  return w;
}

// ball query (CYL = false) or single-depth cylinder query (CYL = true: rot (B,M,9), hmin < x_rot < hmax, reach =
// radius of the sphere containing the cylinder) on the grid.  One wave per centre.  nsample <= 64.
template <bool CYL>
__global__ __launch_bounds__(GQ_WAVES * 64) void ball_grid_kernel(const float *__restrict__ new_xyz,
                                                                  const float *__restrict__ rot,
                                                                  const GridHeader *__restrict__ hdr,
                                                                  const int *__restrict__ cell_start,
                                                                  const float *__restrict__ sxyz,
                                                                  const int32_t *__restrict__ sidx, int n, int m,
                                                                  float radius, float hmin, float hmax, float reach,
                                                                  int nsample, int32_t *__restrict__ idx, int nb) {
  __shared__ int s_queue[GQ_WAVES][128];
  // XCD-aware workgroup order: consecutive workgroup ids go to consecutive XCDs (8 of them, each with its own L2), so
  // cloud = id % nb keeps a cloud's workgroups -- and its cell-sorted points -- on nb-th of the XCDs (one XCD per cloud at
  // batch 8) instead of every XCD fetching every cloud (PMC: 4-5x the compulsory bytes, profiles/r4_pmc_traffic.json)
  const int b = blockIdx.x % nb;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = (blockIdx.x / nb) * GQ_WAVES + wv;
  if (j >= m) return;
  const GridHeader H = hdr[b];
  cell_start += (size_t)b * GRID_CS_STRIDE;
  const gf4 *spt = reinterpret_cast<const gf4 *>(sxyz) + (size_t)b * n;
  (void)sidx;
  const float *c = new_xyz + ((size_t)b * m + j) * 3;
  int32_t *row = idx + ((size_t)b * m + j) * nsample;
  const float cx = c[0], cy = c[1], cz = c[2];
  const float radius2 = radius * radius;   // ball_query_gpu.cu:27, cylinder_query_gpu.cu:24
  float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, r4 = 0.f, r5 = 0.f, r6 = 0.f, r7 = 0.f, r8 = 0.f;
  if (CYL) {
    const float *r = rot + ((size_t)b * m + j) * 9;
    r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3]; r4 = r[4]; r5 = r[5]; r6 = r[6]; r7 = r[7]; r8 = r[8];
  }
  // centre's cell from UNCLAMPED coordinates (a centre may lie outside the cloud's box)
  const int ccx = centre_cell(cx, H.minx, H.invx), ccy = centre_cell(cy, H.miny, H.invy), ccz = centre_cell(cz, H.minz, H.invz);
  // cell window: cells are at least 1.001 x the build reach wide; a grid built for a smaller reach just
  // widens the window
  const int rx = (int)(reach * H.invx) + 1, ry = (int)(reach * H.invy) + 1, rz = (int)(reach * H.invz) + 1;
  int x0 = max(ccx - rx, 0), x1 = min(ccx + rx, H.dx - 1);
  int y0 = max(ccy - ry, 0), y1 = min(ccy + ry, H.dy - 1), z0 = max(ccz - rz, 0), z1 = min(ccz + rz, H.dz - 1);
  if (CYL) {      // the oriented cylinder's own bounding box instead of its bounding sphere's cube
    const CellWindow w = cylinder_window(H, cx, cy, cz, r0, r3, r6, radius, hmin, hmax);
    x0 = max(x0, w.x0); x1 = min(x1, w.x1); y0 = max(y0, w.y0); y1 = min(y1, w.y1); z0 = max(z0, w.z0); z1 = min(z1, w.z1);
    if (!rot_is_orthonormal(r0, r1, r2, r3, r4, r5, r6, r7, r8)) {   // not a rotation: no window is safe, walk every cell
      x0 = 0; x1 = H.dx - 1; y0 = 0; y1 = H.dy - 1; z0 = 0; z1 = H.dz - 1;
    }
  }
  int *queue = s_queue[wv];
  int best = 0x7fffffff;    // ascending across lanes: the nsample... 64 smallest hit indices so far
  int qn = 0;               // hits waiting in the queue
  int total = 0;
  if (x0 <= x1) {
    for (int zz = z0; zz <= z1; ++zz)
      for (int yy = y0; yy <= y1; ++yy) {
        const int cb = (zz * H.dy + yy) * H.dx;
        const int p0 = cell_start[cb + x0], p1 = cell_start[cb + x1 + 1];
        for (int p = p0; p < p1; p += 64) {
          const int q = p + lane;
          bool hit = false;
          int k = 0;
          if (q < p1) {
            const gf4 pq = spt[q];
            const float px = pq.x, py = pq.y, pz = pq.z;
            if (CYL) {   // cylinder_query_gpu.cu:36-46
              const float x = px - cx, y = py - cy, z = pz - cz;
              const float x_rot = fp_dot3(r0, x, r3, y, r6, z);
              const float y_rot = fp_dot3(r1, x, r4, y, r7, z);
              const float z_rot = fp_dot3(r2, x, r5, y, r8, z);
              const float d2 = fp_sumsq2(y_rot, z_rot);
              hit = d2 < radius2 && x_rot > hmin && x_rot < hmax;
            } else {     // ball_query_gpu.cu:33-38
              const float d2 = fp_sumsq3(cx - px, cy - py, cz - pz);
              hit = d2 < radius2;
            }
            k = __float_as_int(pq.w);
          }
          const unsigned long long mask = __ballot(hit);
          if (mask) {
            const int pos = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                           __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (hit) queue[pos] = k;
            const int add = __popcll(mask);
            qn += add;
            total += add;
            if (qn >= 64) {           // fold the oldest 64 queued hits into the sorted set
              __builtin_amdgcn_wave_barrier();
              const int v = sort64(queue[lane], lane);
              best = merge_low64(best, v, lane);
              const int rest = queue[64 + lane];
              __builtin_amdgcn_wave_barrier();
              if (lane < qn - 64) queue[lane] = rest;
              qn -= 64;
            }
          }
        }
      }
  }
  if (qn > 0) {
    __builtin_amdgcn_wave_barrier();
    const int v = sort64(lane < qn ? queue[lane] : 0x7fffffff, lane);
    best = merge_low64(best, v, lane);
  }
  // first `nsample` hits in index order; unfilled slots repeat the first hit; no hit -> zeros
  const int cnt = min(total, nsample);
  const int first = total > 0 ? __builtin_amdgcn_readfirstlane(best) : 0;
  if (lane < nsample) row[lane] = lane < cnt ? best : first;
}

// CloudCrop's four nested cylinders on the grid -> the de-duplicated list of gn_cylinder_query_unique
// (same entries and masks; entries grouped by minimum depth instead of index order, which the
// consumer -- a max-pool over rows -- does not depend on).
//
// A base hit (inside the radius, above hmin) with x_rot < hmax[d] first at d = dmin belongs to the depth
// lists d >= dmin, each of which keeps its 64 smallest indices.  Per depth the wave keeps a sorted
// register of keys (index << 2 | dmin).  Let thr_d be the 64th smallest key of depth d (inf if it has
// fewer hits): thr_0 >= thr_1 >= thr_2 >= thr_3 because the candidate sets grow with d.  Hence an entry
// is in SOME list iff it is in the list of its own dmin, and it is in exactly the lists
// d = dmin .. max{d : key <= thr_d}: the union is the disjoint union over d of {e in L_d : dmin(e) = d}
// and the mask follows from three threshold compares.
template <int ND>
__global__ __launch_bounds__(GQ_WAVES * 64) void cylinder_unique_grid_kernel(
    const float *__restrict__ new_xyz, const float *__restrict__ rot, const GridHeader *__restrict__ hdr,
    const int *__restrict__ cell_start, const float *__restrict__ sxyz, const int32_t *__restrict__ sidx, int n, int m,
    float radius, float hmin, float hmax0, float hmax1, float hmax2, float hmax3, float reach, int nsample,
    int32_t *__restrict__ uidx, int32_t *__restrict__ ucount, int nb) {
  static_assert(ND == 4, "4 depths");
  __shared__ int s_queue[GQ_WAVES][ND][128];
  const int b = blockIdx.x % nb;          // XCD-aware order, see ball_grid_kernel
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int j = (blockIdx.x / nb) * GQ_WAVES + wv;
  if (j >= m) return;
  const GridHeader H = hdr[b];
  cell_start += (size_t)b * GRID_CS_STRIDE;
  const gf4 *spt = reinterpret_cast<const gf4 *>(sxyz) + (size_t)b * n;
  (void)sidx;
  const float *c = new_xyz + ((size_t)b * m + j) * 3;
  const float *r = rot + ((size_t)b * m + j) * 9;
  int32_t *row = uidx + ((size_t)b * m + j) * ND * nsample;
  const float cx = c[0], cy = c[1], cz = c[2];
  const float r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3], r4 = r[4], r5 = r[5], r6 = r[6], r7 = r[7], r8 = r[8];
  const float radius2 = radius * radius;
  const float hmax[4] = {hmax0, hmax1, hmax2, hmax3};
  const int ccx = centre_cell(cx, H.minx, H.invx), ccy = centre_cell(cy, H.miny, H.invy), ccz = centre_cell(cz, H.minz, H.invz);
  const int rx = (int)(reach * H.invx) + 1, ry = (int)(reach * H.invy) + 1, rz = (int)(reach * H.invz) + 1;
  // the bounding sphere's cube, cut down to the bounding box of the deepest (= largest) of the nested cylinders
  const CellWindow cw = cylinder_window(H, cx, cy, cz, r0, r3, r6, radius, hmin, hmax[ND - 1]);
  int x0 = max(max(ccx - rx, 0), cw.x0), x1 = min(min(ccx + rx, H.dx - 1), cw.x1);
  int y0 = max(max(ccy - ry, 0), cw.y0), y1 = min(min(ccy + ry, H.dy - 1), cw.y1);
  int z0 = max(max(ccz - rz, 0), cw.z0), z1 = min(min(ccz + rz, H.dz - 1), cw.z1);
  if (!rot_is_orthonormal(r0, r1, r2, r3, r4, r5, r6, r7, r8)) {     // not a rotation: walk every cell
    x0 = 0; x1 = H.dx - 1; y0 = 0; y1 = H.dy - 1; z0 = 0; z1 = H.dz - 1;
  }
  int best[ND], qn[ND], total[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) { best[d] = 0x7fffffff; qn[d] = 0; total[d] = 0; }

  auto fold = [&](int d) {     // oldest 64 queued keys of depth d -> sorted set
    int *q = s_queue[wv][d];
    __builtin_amdgcn_wave_barrier();
    const int v = sort64(q[lane], lane);
    best[d] = merge_low64(best[d], v, lane);
    const int rest = q[64 + lane];
    __builtin_amdgcn_wave_barrier();
    if (lane < qn[d] - 64) q[lane] = rest;
    qn[d] -= 64;
  };

  if (x0 <= x1) {
    for (int zz = z0; zz <= z1; ++zz)
      for (int yy = y0; yy <= y1; ++yy) {
        const int cb = (zz * H.dy + yy) * H.dx;
        const int p0 = cell_start[cb + x0], p1 = cell_start[cb + x1 + 1];
        for (int p = p0; p < p1; p += 64) {
          const int q = p + lane;
          bool base = false;
          float x_rot = 0.f;
          int k = 0;
          if (q < p1) {
            // cylinder_query_gpu.cu:36-46
            const gf4 pq = spt[q];
            const float x = pq.x - cx, y = pq.y - cy, z = pq.z - cz;
            x_rot = fp_dot3(r0, x, r3, y, r6, z);
            const float y_rot = fp_dot3(r1, x, r4, y, r7, z);
            const float z_rot = fp_dot3(r2, x, r5, y, r8, z);
            const float d2 = fp_sumsq2(y_rot, z_rot);
            base = d2 < radius2 && x_rot > hmin && x_rot < hmax[ND - 1];
            k = __float_as_int(pq.w);
          }
          if (!__ballot(base)) continue;
          const int dmin = x_rot < hmax[0] ? 0 : (x_rot < hmax[1] ? 1 : (x_rot < hmax[2] ? 2 : 3));
          const int key = (k << 2) | dmin;
          // a hit is queued ONCE, with the hits of its own minimum depth; depth d's list is rebuilt at the end as the 64
          // smallest of classes 0..d (each of which kept its own 64 smallest).  Queued into every list d >= dmin it was
          // sorted up to four times: ~10 sort + merge folds per seed instead of ~5.
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            const bool hit = base && dmin == d;
            const unsigned long long mask = __ballot(hit);
            if (mask) {
              const int pos = qn[d] + __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32),
                                                                __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
              if (hit) s_queue[wv][d][pos] = key;
              const int add = __popcll(mask);
              qn[d] += add;
              total[d] += add;
              if (qn[d] >= 64) fold(d);
            }
          }
        }
      }
  }
#pragma unroll
  for (int d = 0; d < ND; ++d)
    if (qn[d] > 0) {
      __builtin_amdgcn_wave_barrier();
      const int v = sort64(lane < qn[d] ? s_queue[wv][d][lane] : 0x7fffffff, lane);
      best[d] = merge_low64(best[d], v, lane);
    }
  // class sets -> depth lists: L_d = 64 smallest of (L_{d-1} u class d), hit counts accumulate
#pragma unroll
  for (int d = 1; d < ND; ++d) {
    best[d] = merge_low64(best[d - 1], best[d], lane);
    total[d] += total[d - 1];
  }
  // thresholds: the largest key each depth's list accepts
  int thr[ND], cnt[ND];
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    cnt[d] = min(total[d], nsample);
    thr[d] = total[d] >= nsample ? __builtin_amdgcn_readlane(best[d], nsample - 1) : 0x7fffffff;
  }
  // depths without any hit: the reference row is all zeros = point 0 repeated
  unsigned emask = 0;
#pragma unroll
  for (int d = 0; d < ND; ++d) emask |= (total[d] == 0 ? 1u : 0u) << d;
  int ucnt = 0;
  bool zero_in_union = false;
#pragma unroll
  for (int d = 0; d < ND; ++d) {
    const int key = best[d];
    const bool own = lane < cnt[d] && (key & 3) == d;       // entry whose minimum depth is d
    unsigned mk = 0;
#pragma unroll
    for (int e = d; e < ND; ++e) mk |= (key <= thr[e] ? 1u : 0u) << e;
    const int k = key >> 2;
    const bool is_zero = own && k == 0;
    if (is_zero) mk |= emask;
    if (__ballot(is_zero)) zero_in_union = true;
    const unsigned long long um = __ballot(own);
    const int slot = ucnt + __builtin_amdgcn_mbcnt_hi((unsigned)(um >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)um, 0u));
    if (own) row[slot] = k | (int)(mk << 28);
    ucnt += __popcll(um);
  }
  if (emask && !zero_in_union) {
    if (lane == 0) row[ucnt] = 0 | (int)(emask << 28);
    ucnt += 1;
  }
  if (lane == 0) ucount[(size_t)b * m + j] = ucnt;
}

}  // namespace GN_FPNS
}  // namespace gn

using namespace gn;
using namespace gn::GN_FPNS;

#if GN_FP == 0      // binning does not decide an index: one copy
extern "C" size_t gn_grid_workspace_bytes(int b, int n) {
  if (b <= 0 || n <= 0) return 0;
  return (size_t)b * (sizeof(GridHeader) + (size_t)GRID_CS_STRIDE * 4 + (size_t)n * 16);
}
#endif

static void grid_carve(void *ws, int b, int n, GridHeader **hdr, int **cs, float **sxyz, int32_t **sidx) {
  char *p = (char *)ws;
  *hdr = (GridHeader *)p;
  p += (size_t)b * sizeof(GridHeader);
  *cs = (int *)p;
  p += (size_t)b * GRID_CS_STRIDE * 4;
  *sxyz = (float *)p;            // (b, n) x (x, y, z, index): 16 bytes per point
  *sidx = (int32_t *)p;          // (folded into the points)
}

#if GN_FP == 0
extern "C" int gn_grid_build(const float *xyz, int b, int n, float reach, void *workspace, gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n >= 1 && reach > 0.f, "grid_build: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(xyz && workspace, "grid_build: null pointer");
  GN_REQUIRE((long)n * 3 < 0x7fffffffL, "grid_build: n too large");
  GridHeader *hdr; int *cs; float *sxyz; int32_t *sidx;
  grid_carve(workspace, b, n, &hdr, &cs, &sxyz, &sidx);
  hipLaunchKernelGGL(grid_build_kernel, dim3(b), dim3(1024), 0, as_stream(stream), xyz, n, reach, hdr, cs, sxyz, sidx);
  GN_CHECK_LAUNCH("grid_build");
  return GN_OK;
}
#endif

extern "C" int GN_FP_ENTRY(gn_ball_query_grid)(const float *new_xyz, const void *grid, int b, int n, int m, float radius,
                                  int nsample, int32_t *idx, gn_stream_t stream) {
  GN_FP_DISPATCH(gn_ball_query_grid, new_xyz, grid, b, n, m, radius, nsample, idx, stream);
  GN_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 1 && nsample <= 64, "ball_query_grid: bad sizes (nsample <= 64)");
  if (b == 0 || m == 0) return GN_OK;
  GN_REQUIRE(new_xyz && grid && idx, "ball_query_grid: null pointer");
  GN_REQUIRE(b <= 65535, "ball_query_grid: batch > 65535 unsupported");
  GridHeader *hdr; int *cs; float *sxyz; int32_t *sidx;
  grid_carve(const_cast<void *>(grid), b, n, &hdr, &cs, &sxyz, &sidx);
  GN_REQUIRE((long)((m + GQ_WAVES - 1) / GQ_WAVES) * b < 0x7fffffffL, "ball_query_grid: too many centres");
  dim3 g((unsigned)((m + GQ_WAVES - 1) / GQ_WAVES) * b);
  hipLaunchKernelGGL(ball_grid_kernel<false>, g, dim3(GQ_WAVES * 64), 0, as_stream(stream), new_xyz, (const float *)nullptr,
                     hdr, cs, sxyz, sidx, n, m, radius, 0.f, 0.f, radius, nsample, idx, b);
  GN_CHECK_LAUNCH("ball_query_grid");
  return GN_OK;
}

extern "C" int GN_FP_ENTRY(gn_cylinder_query_grid)(const float *new_xyz, const float *rot, const void *grid, int b, int n, int m,
                                      float radius, float hmin, float hmax, int nsample, int32_t *idx, gn_stream_t stream) {
  GN_FP_DISPATCH(gn_cylinder_query_grid, new_xyz, rot, grid, b, n, m, radius, hmin, hmax, nsample, idx, stream);
  GN_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 1 && nsample <= 64, "cylinder_query_grid: bad sizes (nsample <= 64)");
  if (b == 0 || m == 0) return GN_OK;
  GN_REQUIRE(new_xyz && rot && grid && idx, "cylinder_query_grid: null pointer");
  GN_REQUIRE(b <= 65535, "cylinder_query_grid: batch > 65535 unsupported");
  GridHeader *hdr; int *cs; float *sxyz; int32_t *sidx;
  grid_carve(const_cast<void *>(grid), b, n, &hdr, &cs, &sxyz, &sidx);
  // every hit lies within this distance of the centre: radial < radius, axial in (hmin, hmax)
  const float ax = fmaxf(fabsf(hmin), fabsf(hmax));
  const float reach = sqrtf(radius * radius + ax * ax) * 1.0001f;
  GN_REQUIRE((long)((m + GQ_WAVES - 1) / GQ_WAVES) * b < 0x7fffffffL, "cylinder_query_grid: too many centres");
  dim3 g((unsigned)((m + GQ_WAVES - 1) / GQ_WAVES) * b);
  hipLaunchKernelGGL(ball_grid_kernel<true>, g, dim3(GQ_WAVES * 64), 0, as_stream(stream), new_xyz, rot, hdr, cs, sxyz, sidx,
                     n, m, radius, hmin, hmax, reach, nsample, idx, b);
  GN_CHECK_LAUNCH("cylinder_query_grid");
  return GN_OK;
}

extern "C" int GN_FP_ENTRY(gn_cylinder_query_unique_grid)(const float *new_xyz, const float *rot, const void *grid, int b, int n,
                                             int m, float radius, float hmin, const float *hmax_host, int ndepth,
                                             int nsample, int32_t *uidx, int32_t *ucount, gn_stream_t stream) {
  GN_FP_DISPATCH(gn_cylinder_query_unique_grid, new_xyz, rot, grid, b, n, m, radius, hmin, hmax_host, ndepth, nsample, uidx, ucount, stream);
  GN_REQUIRE(b >= 0 && n >= 1 && m >= 0 && nsample >= 1 && nsample <= 64, "cylinder_query_unique_grid: bad sizes");
  if (b == 0 || m == 0) return GN_OK;
  GN_REQUIRE(new_xyz && rot && grid && uidx && ucount && hmax_host, "cylinder_query_unique_grid: null pointer");
  GN_REQUIRE(ndepth == 4, "cylinder_query_unique_grid: only 4 depths supported (got %d)", ndepth);
  GN_REQUIRE(b <= 65535, "cylinder_query_unique_grid: batch > 65535 unsupported");
  GN_REQUIRE(n < (1 << 28), "cylinder_query_unique_grid: n must be < 2^28");
  for (int d = 1; d < 4; ++d)
    GN_REQUIRE(hmax_host[d] >= hmax_host[d - 1], "cylinder_query_unique_grid: hmax_list must be ascending");
  GridHeader *hdr; int *cs; float *sxyz; int32_t *sidx;
  grid_carve(const_cast<void *>(grid), b, n, &hdr, &cs, &sxyz, &sidx);
  // every hit lies within this distance of the seed: radial <= radius, axial in (hmin, hmax]
  const float ax = fmaxf(fabsf(hmin), fabsf(hmax_host[3]));
  const float reach = sqrtf(radius * radius + ax * ax) * 1.0001f;
  GN_REQUIRE((long)((m + GQ_WAVES - 1) / GQ_WAVES) * b < 0x7fffffffL, "cylinder_query_unique_grid: too many seeds");
  dim3 g((unsigned)((m + GQ_WAVES - 1) / GQ_WAVES) * b);
  hipLaunchKernelGGL((cylinder_unique_grid_kernel<4>), g, dim3(GQ_WAVES * 64), 0, as_stream(stream), new_xyz, rot, hdr,
                     cs, sxyz, sidx, n, m, radius, hmin, hmax_host[0], hmax_host[1], hmax_host[2], hmax_host[3], reach,
                     nsample, uidx, ucount, b);
  GN_CHECK_LAUNCH("cylinder_query_unique_grid");
  return GN_OK;
}
