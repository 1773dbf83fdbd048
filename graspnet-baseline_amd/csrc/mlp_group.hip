// mlp_group.hip -- fused "group -> centre/normalise (or rotate) -> SharedMLP x3 -> max-pool" kernels.
//
// Set-abstraction tail of PointnetSAModuleVotes.forward (pointnet2/pointnet2_modules.py:243-257 with
// QueryAndGroup.forward, pointnet2/pointnet2_utils.py:344-355) and CloudCrop.forward
// (models/modules.py:127-140 with CylinderQueryAndGroup.forward, pointnet2_utils.py:521-530).
// The reference materialises the grouped tensor (B,3+C,np,ns), three conv/BN/ReLU activations and
// the pooled result in HBM (2.1 GB for CloudCrop at batch 8); here a persistent workgroup gathers a
// tile of TP neighbour rows straight into LDS, runs the three layers on fp32 MFMA with the weights
// resident in registers (mlp_core.h) and writes only the pooled (group, C3) rows.
#include <stdlib.h>

#include "mlp_core.h"

namespace gn {

struct Mlp3Weights {
  const float *w1, *b1, *w2, *b2, *w3, *b3;  // fragment-packed weights / padded bias per layer
};

// ---- loaders: fill X0[TP][CIN0+4] in LDS ------------------------------------------------------
// SA rows: [ features of neighbour (C) | (xyz[src]-centre)/radius (3) | 0-pad to 8 ]
// (the host permutes the first layer's input columns accordingly: the reference concatenates xyz
//  first, pointnet2_utils.py:353)
template <int C, int TP, int NS>
__device__ __forceinline__ void load_sa_tile(float *__restrict__ xs, const float *__restrict__ xyz,
                                             const float *__restrict__ feat, const float *__restrict__ cxyz,
                                             const int32_t *__restrict__ idx, float radius, int tid) {
  constexpr int XS = lds_stride(C + 8);
  if (C > 0) {
    constexpr int Q = C / 4;  // float4 per row
    for (int e = tid; e < TP * Q; e += MLP_THREADS) {
      const int row = e / Q, c4 = e - row * Q;
      const int src = idx[row];
      const f32x4 v = *reinterpret_cast<const f32x4 *>(feat + (size_t)src * C + c4 * 4);
      *reinterpret_cast<f32x4 *>(xs + row * XS + c4 * 4) = v;
    }
  }
  for (int p = tid; p < TP; p += MLP_THREADS) {
    const int src = idx[p];
    const float *c = cxyz + (p / NS) * 3;
    // grouped_xyz -= new_xyz; grouped_xyz /= radius   (pointnet2_utils.py:346-348)
    f32x4 v;
    v[0] = (xyz[src * 3 + 0] - c[0]) / radius;
    v[1] = (xyz[src * 3 + 1] - c[1]) / radius;
    v[2] = (xyz[src * 3 + 2] - c[2]) / radius;
    v[3] = 0.f;
    *reinterpret_cast<f32x4 *>(xs + p * XS + C) = v;
    *reinterpret_cast<f32x4 *>(xs + p * XS + C + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

template <int CIN0, int C1, int C2, int TP>
__host__ __device__ constexpr int group_mlp_lds_floats() {
  return cmax(TP * lds_stride(CIN0), TP * lds_stride(C2)) + TP * lds_stride(C1);
}

// ---- set abstraction ---------------------------------------------------------------------------
template <int C, int C1, int C2, int C3, int NS, int TP>
__global__ __launch_bounds__(MLP_THREADS) void sa_mlp_kernel(const float *__restrict__ xyz, int n,
                                                             const float *__restrict__ feat,
                                                             const float *__restrict__ new_xyz,
                                                             const int32_t *__restrict__ idx, int np, int ntiles,
                                                             int tpb, float radius, Mlp3Weights W,
                                                             float *__restrict__ out) {
  constexpr int CIN0 = C + 8;
  constexpr int GPT = TP / NS;  // groups per tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *bufA = smem;
  float *bufB = smem + cmax(TP * lds_stride(CIN0), TP * lds_stride(C2));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  LayerW<CIN0, C1> w1;
  LayerW<C1, C2> w2;
  LayerW<C2, C3> w3;
  w1.load(W.w1, W.b1, wave, lane);
  w2.load(W.w2, W.b2, wave, lane);
  w3.load(W.w3, W.b3, wave, lane);

  // a block owns `tpb` consecutive tiles (weights loaded once per block); the grid is larger than the
  // chip so the hardware dispatcher balances blocks over whatever CUs other streams leave free
  const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * tpb);
  for (int tile = blockIdx.x * tpb; tile < tile_end; ++tile) {
    const int g0 = tile * GPT;   // first (flat) group of the tile; np % GPT == 0 so it stays in one cloud
    const int b = g0 / np;
    load_sa_tile<C, TP, NS>(bufA, xyz + (size_t)b * n * 3, C > 0 ? feat + (size_t)b * n * C : nullptr,
                            new_xyz + (size_t)g0 * 3, idx + (size_t)g0 * NS, radius, tid);
    __syncthreads();
    StoreLds<true> e1{bufB, lds_stride(C1), C1, lane};
    run_layer<TP, CIN0, C1>(w1, bufA, wave, lane, e1);
    __syncthreads();
    StoreLds<true> e2{bufA, lds_stride(C2), C2, lane};
    run_layer<TP, C1, C2>(w2, bufB, wave, lane, e2);
    __syncthreads();
    PoolGlobal<NS, LayerW<C2, C3>::NTW> e3{out + (size_t)g0 * C3, C3, lane, {}};
    run_layer<TP, C2, C3>(w3, bufA, wave, lane, e3);
    __syncthreads();
  }
}

// ---- cloud crop --------------------------------------------------------------------------------
// MINW = waves per SIMD the register allocation must allow (2 => two workgroups per CU, so one
// workgroup's gather / LDS epilogues / barriers hide behind the other's MFMAs).
// RES12 = keep layers 1-2 resident in registers too (else they are re-read from L2 per tile; only
// the 256x128 last layer, 80 % of the FLOPs, stays resident).
template <int C1, int C2, int C3, int NS, int ND, int TP, int MINW, bool RES12>
__global__ __launch_bounds__(MLP_THREADS, MINW) void crop_mlp_kernel(const float *__restrict__ xyz, int n,
                                                                     const float *__restrict__ seed,
                                                                     const float *__restrict__ rot,
                                                                     const int32_t *__restrict__ idx, int m,
                                                                     int ntiles, int tpb, Mlp3Weights W,
                                                                     float *__restrict__ out) {
  constexpr int CIN0 = 8;
  constexpr int GPT = TP / NS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *bufA = smem;
  float *bufB = smem + cmax(TP * lds_stride(CIN0), TP * lds_stride(C2));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  LayerW<CIN0, C1> w1;
  LayerW<C1, C2> w2;
  LayerW<C2, C3> w3;
  if (RES12) {
    w1.load(W.w1, W.b1, wave, lane);
    w2.load(W.w2, W.b2, wave, lane);
  }
  w3.load(W.w3, W.b3, wave, lane);

  const int gpc = m * ND;  // groups per cloud
  const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * tpb);
  for (int tile = blockIdx.x * tpb; tile < tile_end; ++tile) {
    const int g0 = tile * GPT;
    const int b = g0 / gpc;
    const int gl = g0 - b * gpc;  // first group within the cloud; groups are [seed][depth]
    const float *seed_b = seed + (size_t)b * m * 3;
    const float *rot_b = rot + (size_t)b * m * 9;
    {
      // rows: [ (xyz[src]-seed)^T R (3) | 0-pad to 8 ]   (pointnet2_utils.py:524-530)
      constexpr int XS = lds_stride(8);
      const int32_t *ix = idx + (size_t)g0 * NS;
      const float *px = xyz + (size_t)b * n * 3;
      for (int p = tid; p < TP; p += MLP_THREADS) {
        const int src = ix[p];
        const int s = (gl + p / NS) / ND;
        const float *c = seed_b + s * 3;
        const float *r = rot_b + s * 9;
        const float x = px[src * 3 + 0] - c[0];
        const float y = px[src * 3 + 1] - c[1];
        const float z = px[src * 3 + 2] - c[2];
        f32x4 v;
        v[0] = x * r[0] + y * r[3] + z * r[6];
        v[1] = x * r[1] + y * r[4] + z * r[7];
        v[2] = x * r[2] + y * r[5] + z * r[8];
        v[3] = 0.f;
        *reinterpret_cast<f32x4 *>(bufA + p * XS) = v;
        *reinterpret_cast<f32x4 *>(bufA + p * XS + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    __syncthreads();
    StoreLds<true> e1{bufB, lds_stride(C1), C1, lane};
    if (!RES12) w1.load(W.w1, W.b1, wave, lane);
    run_layer<TP, CIN0, C1>(w1, bufA, wave, lane, e1);
    __syncthreads();
    StoreLds<true> e2{bufA, lds_stride(C2), C2, lane};
    if (!RES12) w2.load(W.w2, W.b2, wave, lane);
    run_layer<TP, C1, C2>(w2, bufB, wave, lane, e2);
    __syncthreads();
    PoolGlobal<NS, LayerW<C2, C3>::NTW> e3{out + (size_t)g0 * C3, C3, lane, {}};
    run_layer<TP, C2, C3>(w3, bufA, wave, lane, e3);
    __syncthreads();
  }
}

// ---- set abstraction level 1 (coordinates only): sa_mlp_kernel<0,...> with the next tile's rows
// (index -> xyz gather, centre, /radius) prefetched into registers under the current tile's MFMAs and a
// double-buffered 6 KB input tile.
template <int C1, int C2, int C3, int NS>
__global__ __launch_bounds__(MLP_THREADS) void sa_xyz_kernel(const float *__restrict__ xyz, int n,
                                                             const float *__restrict__ new_xyz,
                                                             const int32_t *__restrict__ idx, int np, int ntiles,
                                                             int tpb, float radius, Mlp3Weights W,
                                                             float *__restrict__ out) {
  constexpr int TP = 128, GPT = TP / NS, XS0 = lds_stride(8);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *bufY1 = smem;                          // [TP][C1+4]
  float *bufY2 = bufY1 + TP * lds_stride(C1);   // [TP][C2+4]
  float *bufX = bufY2 + TP * lds_stride(C2);    // 2 x [TP][12]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  LayerW<8, C1> w1;
  LayerW<C1, C2> w2;
  LayerW<C2, C3> w3;
  w1.load(W.w1, W.b1, wave, lane);
  w2.load(W.w2, W.b2, wave, lane);
  w3.load(W.w3, W.b3, wave, lane);
  const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * tpb);
  int tile = blockIdx.x * tpb;
  if (tile >= tile_end) return;
  auto fetch_x = [&](int tl, f32x4 &v) {
    const int g0 = tl * GPT;
    const int b = g0 / np;
    v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < TP) {
      const int src = idx[(size_t)g0 * NS + tid];
      const float *c = new_xyz + ((size_t)g0 + tid / NS) * 3;
      const float *px = xyz + (size_t)b * n * 3;
      v[0] = (px[src * 3 + 0] - c[0]) / radius;   // pointnet2_utils.py:346-348
      v[1] = (px[src * 3 + 1] - c[1]) / radius;
      v[2] = (px[src * 3 + 2] - c[2]) / radius;
    }
  };
  auto put_x = [&](int buf, const f32x4 &v) {
    if (tid < TP) {
      float *x = bufX + buf * TP * XS0 + tid * XS0;
      *reinterpret_cast<f32x4 *>(x) = v;
      *reinterpret_cast<f32x4 *>(x + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int buf = 0;
  {
    f32x4 v;
    fetch_x(tile, v);
    put_x(0, v);
  }
  __syncthreads();
  for (; tile < tile_end; ++tile) {
    const int g0 = tile * GPT;
    const bool has_next = tile + 1 < tile_end;
    f32x4 nv;
    if (has_next) fetch_x(tile + 1, nv);          // in flight under the three layers
    StoreLds<true> e1{bufY1, lds_stride(C1), C1, lane};
    run_layer<TP, 8, C1>(w1, bufX + buf * TP * XS0, wave, lane, e1);
    __syncthreads();
    StoreLds<true> e2{bufY2, lds_stride(C2), C2, lane};
    run_layer<TP, C1, C2>(w2, bufY1, wave, lane, e2);
    __syncthreads();
    PoolGlobal<NS, LayerW<C2, C3>::NTW> e3{out + (size_t)g0 * C3, C3, lane, {}};
    run_layer<TP, C2, C3>(w3, bufY2, wave, lane, e3);
    if (has_next) put_x(buf ^ 1, nv);
    __syncthreads();
    buf ^= 1;
  }
}

// ---- set abstraction with layer 1 hoisted (levels with input features) ---------------------------
// Layer 1 of a grouped MLP is  W1f * feat_j + W1x * (x_j - c_i)/r + b.  The first term does not depend
// on the group, yet the reference (and sa_mlp_kernel) recompute it for each of the ~16 groups a
// point belongs to.  gn_point_linear evaluates P_j = W1f * feat_j + b once per point; here the
// gathered P rows initialise the accumulators and only the K=8 coordinate block is multiplied per
// neighbour.  The k order (features, then coordinates) is the one sa_mlp_kernel uses, so results are
// bit-identical to it.  The next tile's P rows (16 float4 per lane) and coordinate rows are fetched
// into registers while the current tile runs layers 2-3.
template <int C1, int C2, int C3, int NS>
__global__ __launch_bounds__(MLP_THREADS) void sa_hoist_kernel(const float *__restrict__ xyz, int n,
                                                               const float *__restrict__ pfeat,
                                                               const float *__restrict__ new_xyz,
                                                               const int32_t *__restrict__ idx, int np, int ntiles,
                                                               int tpb, float radius, Mlp3Weights W,
                                                               float *__restrict__ out) {
  constexpr int TP = 128, GPT = TP / NS, XS0 = lds_stride(8), PS = lds_stride(C1), Q = C1 / 4;
  constexpr int PF = TP * Q / MLP_THREADS;   // float4 of P per lane per tile
  static_assert(TP * Q % MLP_THREADS == 0 && MLP_THREADS % Q == 0, "prefetch mapping");
  constexpr int RSTEP = MLP_THREADS / Q;     // rows between a lane's consecutive float4
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *bufP = smem;                        // [TP][C1+4]  P rows, then layer-1 output in place
  float *bufY2 = bufP + TP * PS;             // [TP][C2+4]
  float *bufX = bufY2 + TP * lds_stride(C2); // 2 x [TP][12]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int prow = tid / Q, pc4 = tid % Q;

  LayerW<8, C1> w1;
  LayerW<C1, C2> w2;
  LayerW<C2, C3> w3;
  w1.load(W.w1, W.b1, wave, lane);
  w2.load(W.w2, W.b2, wave, lane);
  w3.load(W.w3, W.b3, wave, lane);

  const int tile_end = min(ntiles, (int)(blockIdx.x + 1) * tpb);
  int tile = blockIdx.x * tpb;
  if (tile >= tile_end) return;

  auto fetch_p = [&](int tl, f32x4 (&pf)[PF]) {
    const int g0 = tl * GPT;
    const int b = g0 / np;
    const int32_t *ix = idx + (size_t)g0 * NS;
    const float *pb = pfeat + (size_t)b * n * C1 + pc4 * 4;
#pragma unroll
    for (int q = 0; q < PF; ++q) pf[q] = *reinterpret_cast<const f32x4 *>(pb + (size_t)ix[prow + q * RSTEP] * C1);
  };
  auto fetch_x = [&](int tl, f32x4 &v) {
    const int g0 = tl * GPT;
    const int b = g0 / np;
    v = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < TP) {
      const int src = idx[(size_t)g0 * NS + tid];
      const float *c = new_xyz + ((size_t)g0 + tid / NS) * 3;
      const float *px = xyz + (size_t)b * n * 3;
      v[0] = (px[src * 3 + 0] - c[0]) / radius;   // pointnet2_utils.py:346-348
      v[1] = (px[src * 3 + 1] - c[1]) / radius;
      v[2] = (px[src * 3 + 2] - c[2]) / radius;
    }
  };
  auto put_p = [&](const f32x4 (&pf)[PF]) {
#pragma unroll
    for (int q = 0; q < PF; ++q) *reinterpret_cast<f32x4 *>(bufP + (prow + q * RSTEP) * PS + pc4 * 4) = pf[q];
  };
  auto put_x = [&](int buf, const f32x4 &v) {
    if (tid < TP) {
      float *x = bufX + buf * TP * XS0 + tid * XS0;
      *reinterpret_cast<f32x4 *>(x) = v;
      *reinterpret_cast<f32x4 *>(x + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };

  int buf = 0;
  {
    f32x4 pf[PF], v;
    fetch_p(tile, pf);
    fetch_x(tile, v);
    put_p(pf);
    put_x(0, v);
  }
  __syncthreads();
  for (; tile < tile_end; ++tile) {
    const int g0 = tile * GPT;
    const bool has_next = tile + 1 < tile_end;
    StoreLds<true> e1{bufP, PS, C1, lane};
    run_layer_cinit<TP, 8, C1>(w1, bufX + buf * TP * XS0, bufP, PS, wave, lane, e1);
    __syncthreads();
    f32x4 pf[PF], nv;
    if (has_next) { fetch_p(tile + 1, pf); fetch_x(tile + 1, nv); }   // in flight under layers 2-3
    StoreLds<true> e2{bufY2, lds_stride(C2), C2, lane};
    run_layer<TP, C1, C2>(w2, bufP, wave, lane, e2);
    __syncthreads();
    PoolGlobal<NS, LayerW<C2, C3>::NTW> e3{out + (size_t)g0 * C3, C3, lane, {}};
    run_layer<TP, C2, C3>(w3, bufY2, wave, lane, e3);
    if (has_next) { put_p(pf); put_x(buf ^ 1, nv); }   // bufP is free: every wave is past layer 2
    __syncthreads();
    buf ^= 1;
  }
}

// ---- cloud crop over the UNIQUE rows of each seed (gn_cylinder_query_unique) ----------------------
// One seed = one ragged row set (<= ND*NS entries, typically ~90): rows are evaluated once and
// max-pooled into the depths named by their mask.  2.3-2.5x fewer MFMAs than pooling ND full lists,
// same bits.  A workgroup walks `spb` consecutive seeds with all weights resident; the work is a
// sequence of <=128-row chunks, and the NEXT chunk's rows (index -> xyz gather -> rotate) are
// fetched into registers while the current chunk runs through the three layers, then dropped into
// the other half of a double-buffered LDS input tile.
template <int C1, int C2, int C3, int ND, int CAP, int TP, int MINW>
__global__ __launch_bounds__(MLP_THREADS, MINW) void crop_unique_mlp_kernel(const float *__restrict__ xyz, int n,
                                                                      const float *__restrict__ seed,
                                                                      const float *__restrict__ rot,
                                                                      const int32_t *__restrict__ uidx,
                                                                      const int32_t *__restrict__ ucount, int m,
                                                                      int nseeds, int spb, Mlp3Weights W,
                                                                      float *__restrict__ out) {
  constexpr int CIN0 = 8, XS0 = lds_stride(CIN0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *bufY2 = smem;                                  // [TP][C2+4]
  float *bufY1 = bufY2 + TP * lds_stride(C2);           // [TP][C1+4]
  float *bufX = bufY1 + TP * lds_stride(C1);            // 2 x [TP][12]
  float *bufG = bufX + 2 * TP * XS0;                    // 2 x [TP][4] pooling gates
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  LayerW<CIN0, C1> w1;
  LayerW<C1, C2> w2;
  LayerW<C2, C3> w3;
  w1.load(W.w1, W.b1, wave, lane);
  w2.load(W.w2, W.b2, wave, lane);
  w3.load(W.w3, W.b3, wave, lane);
  using L3 = LayerW<C2, C3>;
  PoolMasked<ND, L3::NTW> pool{bufG, lane, {}};
  pool.reset();

  const int s_end = min(nseeds, (int)(blockIdx.x + 1) * spb);
  // chunk cursor: seed sf, first row row0, seed's entry count nu
  int sf = blockIdx.x * spb;
  if (sf >= s_end) return;
  int nu = ucount[sf], row0 = 0;

  // row p of the chunk (sf,row0): gather + rotate; returns the LDS row and its gates
  auto fetch = [&](int csf, int crow0, int cnu, int p, f32x4 &v, f32x4 &g) {
    v = f32x4{0.f, 0.f, 0.f, 0.f};
    g = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (crow0 + p < cnu) {
      const int e = uidx[(size_t)csf * CAP + crow0 + p];
      const int src = e & 0x0fffffff;
      const unsigned mk = (unsigned)e >> 28;
      const float *c = seed + (size_t)csf * 3;
      const float *r = rot + (size_t)csf * 9;
      const float *px = xyz + (size_t)(csf / m) * n * 3;
      const float x = px[src * 3 + 0] - c[0];
      const float y = px[src * 3 + 1] - c[1];
      const float z = px[src * 3 + 2] - c[2];
      v[0] = x * r[0] + y * r[3] + z * r[6];   // (p-c)^T R, pointnet2_utils.py:528-530
      v[1] = x * r[1] + y * r[4] + z * r[7];
      v[2] = x * r[2] + y * r[5] + z * r[8];
#pragma unroll
      for (int d = 0; d < ND; ++d) g[d] = ((mk >> d) & 1u) ? 0.f : -INFINITY;
    }
  };
  auto put = [&](int buf, int p, const f32x4 &v, const f32x4 &g) {
    float *x = bufX + buf * TP * XS0 + p * XS0;
    *reinterpret_cast<f32x4 *>(x) = v;
    *reinterpret_cast<f32x4 *>(x + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4 *>(bufG + buf * TP * 4 + p * 4) = g;
  };

  int buf = 0;
  {
    f32x4 v, g;
    if (tid < TP) { fetch(sf, row0, nu, tid, v, g); put(0, tid, v, g); }
  }
  __syncthreads();
  while (true) {
    const int rows = min(TP, nu - row0);
    const int mtc = (rows + 31) >> 5;
    const bool seed_done = row0 + TP >= nu;
    // next chunk
    int nsf = sf, nrow0 = row0 + TP, nnu = nu;
    if (seed_done) {
      nsf = sf + 1;
      nrow0 = 0;
      nnu = nsf < s_end ? ucount[nsf] : 0;
    }
    const bool has_next = nsf < s_end;
    f32x4 nv, ng;
    if (has_next && tid < TP) fetch(nsf, nrow0, nnu, tid, nv, ng);   // in flight during the layers below

    pool.sgate = bufG + buf * TP * 4;
    StoreLds<true> e1{bufY1, lds_stride(C1), C1, lane};
    run_layer<TP, CIN0, C1>(w1, bufX + buf * TP * XS0, wave, lane, e1, mtc);
    __syncthreads();
    StoreLds<true> e2{bufY2, lds_stride(C2), C2, lane};
    run_layer<TP, C1, C2>(w2, bufY1, wave, lane, e2, mtc);
    __syncthreads();
    run_layer<TP, C2, C3>(w3, bufY2, wave, lane, pool, mtc);
    if (seed_done) {
      pool.finish(out + (size_t)sf * ND * C3, C3, (wave % L3::WN) * L3::NTW);
      pool.reset();
    }
    if (!has_next) break;
    if (tid < TP) put(buf ^ 1, tid, nv, ng);
    __syncthreads();   // next input tile visible; Y1/Y2 free
    buf ^= 1;
    sf = nsf; row0 = nrow0; nu = nnu;
  }
}

// 8-wave flavour: two waves per SIMD, each owning half as many output channels (100 weight registers
// instead of 164), so one wave's LDS / pooling epilogue runs under the other's MFMAs.
template <int C1, int C2, int C3, int ND, int CAP, int TP>
__global__ __launch_bounds__(512) void crop_unique8_mlp_kernel(const float *__restrict__ xyz, int n,
                                                                      const float *__restrict__ seed,
                                                                      const float *__restrict__ rot,
                                                                      const int32_t *__restrict__ uidx,
                                                                      const int32_t *__restrict__ ucount, int m,
                                                                      int nseeds, int spb, Mlp3Weights W,
                                                                      float *__restrict__ out) {
  constexpr int CIN0 = 8, XS0 = lds_stride(CIN0);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *bufY2 = smem;                                  // [TP][C2+4]
  float *bufY1 = bufY2 + TP * lds_stride(C2);           // [TP][C1+4]
  float *bufX = bufY1 + TP * lds_stride(C1);            // 2 x [TP][12]
  float *bufG = bufX + 2 * TP * XS0;                    // 2 x [TP][4] pooling gates
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  LayerW<CIN0, C1, 8> w1;
  LayerW<C1, C2, 8> w2;
  LayerW<C2, C3, 8> w3;
  w1.load(W.w1, W.b1, wave, lane);
  w2.load(W.w2, W.b2, wave, lane);
  w3.load(W.w3, W.b3, wave, lane);
  using L3 = LayerW<C2, C3, 8>;
  PoolMasked<ND, L3::NTW> pool{bufG, lane, {}};
  pool.reset();

  const int s_end = min(nseeds, (int)(blockIdx.x + 1) * spb);
  // chunk cursor: seed sf, first row row0, seed's entry count nu
  int sf = blockIdx.x * spb;
  if (sf >= s_end) return;
  int nu = ucount[sf], row0 = 0;

  // row p of the chunk (sf,row0): gather + rotate; returns the LDS row and its gates
  auto fetch = [&](int csf, int crow0, int cnu, int p, f32x4 &v, f32x4 &g) {
    v = f32x4{0.f, 0.f, 0.f, 0.f};
    g = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (crow0 + p < cnu) {
      const int e = uidx[(size_t)csf * CAP + crow0 + p];
      const int src = e & 0x0fffffff;
      const unsigned mk = (unsigned)e >> 28;
      const float *c = seed + (size_t)csf * 3;
      const float *r = rot + (size_t)csf * 9;
      const float *px = xyz + (size_t)(csf / m) * n * 3;
      const float x = px[src * 3 + 0] - c[0];
      const float y = px[src * 3 + 1] - c[1];
      const float z = px[src * 3 + 2] - c[2];
      v[0] = x * r[0] + y * r[3] + z * r[6];   // (p-c)^T R, pointnet2_utils.py:528-530
      v[1] = x * r[1] + y * r[4] + z * r[7];
      v[2] = x * r[2] + y * r[5] + z * r[8];
#pragma unroll
      for (int d = 0; d < ND; ++d) g[d] = ((mk >> d) & 1u) ? 0.f : -INFINITY;
    }
  };
  auto put = [&](int buf, int p, const f32x4 &v, const f32x4 &g) {
    float *x = bufX + buf * TP * XS0 + p * XS0;
    *reinterpret_cast<f32x4 *>(x) = v;
    *reinterpret_cast<f32x4 *>(x + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4 *>(bufG + buf * TP * 4 + p * 4) = g;
  };

  int buf = 0;
  {
    f32x4 v, g;
    if (tid < TP) { fetch(sf, row0, nu, tid, v, g); put(0, tid, v, g); }
  }
  __syncthreads();
  GN_PROBE_INIT();
  while (true) {
    const int rows = min(TP, nu - row0);
    const int mtc = (rows + 31) >> 5;
    const bool seed_done = row0 + TP >= nu;
    // next chunk
    int nsf = sf, nrow0 = row0 + TP, nnu = nu;
    if (seed_done) {
      nsf = sf + 1;
      nrow0 = 0;
      nnu = nsf < s_end ? ucount[nsf] : 0;
    }
    const bool has_next = nsf < s_end;
    f32x4 nv, ng;
    if (has_next && tid < TP) fetch(nsf, nrow0, nnu, tid, nv, ng);   // in flight during the layers below
    GN_PROBE(0);

    pool.sgate = bufG + buf * TP * 4;
    StoreLds<true> e1{bufY1, lds_stride(C1), C1, lane};
    run_layer<TP, CIN0, C1, StoreLds<true>, 8>(w1, bufX + buf * TP * XS0, wave, lane, e1, mtc);
    GN_PROBE(1);
    __syncthreads();
    GN_PROBE(2);
    StoreLds<true> e2{bufY2, lds_stride(C2), C2, lane};
    run_layer<TP, C1, C2, StoreLds<true>, 8>(w2, bufY1, wave, lane, e2, mtc);
    GN_PROBE(3);
    __syncthreads();
    GN_PROBE(4);
    run_layer<TP, C2, C3, PoolMasked<ND, L3::NTW>, 8>(w3, bufY2, wave, lane, pool, mtc);
    GN_PROBE(5);
    if (seed_done) {
      pool.finish(out + (size_t)sf * ND * C3, C3, (wave % L3::WN) * L3::NTW);
      pool.reset();
    }
    GN_PROBE(6);
    if (!has_next) break;
    if (tid < TP) put(buf ^ 1, tid, nv, ng);
    GN_PROBE(7);
    __syncthreads();   // next input tile visible; Y1/Y2 free
    GN_PROBE(8);
    GN_PROBE_COUNT(9, 1);
    GN_PROBE_COUNT(10, mtc);
    buf ^= 1;
    sf = nsf; row0 = nrow0; nu = nnu;
  }
  GN_PROBE_FLUSH();
}

// tiles per block: amortise the per-block weight load over up to 8 tiles while keeping >= ~4 blocks
// per CU in the grid (dynamic balance, and room for concurrent streams)
static int tiles_per_block(int ntiles) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  int t = ntiles / (4 * cus);
  return t < 1 ? 1 : (t > 8 ? 8 : t);
}

template <class K>
static int set_lds(K kernel, size_t bytes, const char *name) {
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)bytes) != hipSuccess) {
    set_error("%s: cannot reserve %zu bytes of LDS", name, bytes);
    return GN_ERR_LAUNCH;
  }
  return GN_OK;
}

template <int C, int C1, int C2, int C3, int NS, int TP>
static int launch_sa(const float *xyz, const float *feat, const float *new_xyz, const int32_t *idx, int b, int n,
                     int np, float radius, const Mlp3Weights &W, float *out, hipStream_t s) {
  GN_REQUIRE(np % (TP / NS) == 0, "sa_group_mlp_pool: npoint must be a multiple of %d", TP / NS);
  const int ntiles = (int)((long)b * np / (TP / NS));
  constexpr size_t lds = sizeof(float) * group_mlp_lds_floats<C + 8, C1, C2, TP>();
  auto k = sa_mlp_kernel<C, C1, C2, C3, NS, TP>;
  int rc = set_lds(k, lds, "sa_group_mlp_pool");
  if (rc) return rc;
  const int tpb = tiles_per_block(ntiles);
  hipLaunchKernelGGL(k, dim3((ntiles + tpb - 1) / tpb), dim3(MLP_THREADS), lds, s, xyz, n, feat, new_xyz, idx, np,
                     ntiles, tpb, radius, W, out);
  GN_CHECK_LAUNCH("sa_group_mlp_pool");
  return GN_OK;
}

}  // namespace gn

using namespace gn;

extern "C" int gn_sa_group_mlp_pool(const float *xyz, const float *feat, const float *new_xyz, const int32_t *idx,
                                    int b, int n, int np, int ns, int c, float radius, const int *mlp,
                                    const float *const *wpacked, const float *const *bias, float *out,
                                    gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n > 0 && np > 0 && ns > 0 && c >= 0, "sa_group_mlp_pool: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(xyz && new_xyz && idx && mlp && wpacked && bias && out && (c == 0 || feat),
             "sa_group_mlp_pool: null pointer");
  GN_REQUIRE((long)n * cmax(c, 3) < 0x7fffffffL, "sa_group_mlp_pool: n*c too large");
  const Mlp3Weights W{wpacked[0], bias[0], wpacked[1], bias[1], wpacked[2], bias[2]};
  hipStream_t s = as_stream(stream);
  const int c1 = mlp[0], c2 = mlp[1], c3 = mlp[2];
  // configurations of models/backbone.py:30-64
  if (c == 0 && c1 == 64 && c2 == 64 && c3 == 128 && ns == 64) {
    constexpr int TP = 128;
    GN_REQUIRE(np % (TP / 64) == 0, "sa_group_mlp_pool: npoint must be even");
    const int ntiles = (int)((long)b * np / (TP / 64));
    constexpr size_t lds = sizeof(float) * (TP * lds_stride(64) + TP * lds_stride(64) + 2 * TP * lds_stride(8));
    auto k = sa_xyz_kernel<64, 64, 128, 64>;
    int rc = set_lds(k, lds, "sa_group_mlp_pool");
    if (rc) return rc;
    const int tpb = tiles_per_block(ntiles);
    hipLaunchKernelGGL(k, dim3((ntiles + tpb - 1) / tpb), dim3(MLP_THREADS), lds, s, xyz, n, new_xyz, idx, np, ntiles,
                       tpb, radius, W, out);
    GN_CHECK_LAUNCH("sa_group_mlp_pool");
    return GN_OK;
  }
  if (c == 128 && c1 == 128 && c2 == 128 && c3 == 256 && ns == 32)
    return launch_sa<128, 128, 128, 256, 32, 128>(xyz, feat, new_xyz, idx, b, n, np, radius, W, out, s);
  if (c == 256 && c1 == 128 && c2 == 128 && c3 == 256 && ns == 16)
    return launch_sa<256, 128, 128, 256, 16, 64>(xyz, feat, new_xyz, idx, b, n, np, radius, W, out, s);
  set_error("sa_group_mlp_pool: unsupported configuration C=%d mlp=[%d,%d,%d] nsample=%d", c, c1, c2, c3, ns);
  return GN_ERR_INVALID_ARG;
}

extern "C" int gn_crop_group_mlp_pool(const float *xyz, const float *seed_xyz, const float *rot, const int32_t *idx,
                                      int b, int n, int m, int ndepth, int ns, const int *mlp,
                                      const float *const *wpacked, const float *const *bias, float *out,
                                      gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n > 0 && m > 0, "crop_group_mlp_pool: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(xyz && seed_xyz && rot && idx && mlp && wpacked && bias && out, "crop_group_mlp_pool: null pointer");
  GN_REQUIRE((long)n * 3 < 0x7fffffffL, "crop_group_mlp_pool: n too large");
  if (!(ndepth == 4 && ns == 64 && mlp[0] == 64 && mlp[1] == 128 && mlp[2] == 256)) {
    set_error("crop_group_mlp_pool: unsupported configuration depth=%d nsample=%d mlp=[%d,%d,%d]", ndepth, ns, mlp[0],
              mlp[1], mlp[2]);
    return GN_ERR_INVALID_ARG;
  }
  constexpr int NS = 64, ND = 4;
  const Mlp3Weights W{wpacked[0], bias[0], wpacked[1], bias[1], wpacked[2], bias[2]};
  auto launch = [&](auto kernel, int tp, size_t lds) -> int {
    if ((m * ND) % (tp / NS) != 0) { set_error("crop_group_mlp_pool: bad seed count"); return GN_ERR_INVALID_ARG; }
    const int ntiles = (int)((long)b * m * ND / (tp / NS));
    int rc = set_lds(kernel, lds, "crop_group_mlp_pool");
    if (rc) return rc;
    const int tpb = tiles_per_block(ntiles);
    hipLaunchKernelGGL(kernel, dim3((ntiles + tpb - 1) / tpb), dim3(MLP_THREADS), lds, as_stream(stream), xyz, n,
                       seed_xyz, rot, idx, m, ntiles, tpb, W, out);
    return GN_OK;
  };
  int rc = launch(crop_mlp_kernel<64, 128, 256, NS, ND, 128, 1, true>, 128,
                  sizeof(float) * group_mlp_lds_floats<8, 64, 128, 128>());
  if (rc) return rc;
  GN_CHECK_LAUNCH("crop_group_mlp_pool");
  return GN_OK;
}

extern "C" int gn_crop_unique_mlp_pool(const float *xyz, const float *seed_xyz, const float *rot, const int32_t *uidx,
                                       const int32_t *ucount, int b, int n, int m, int ndepth, int ns, const int *mlp,
                                       const float *const *wpacked, const float *const *bias, float *out,
                                       gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n > 0 && m > 0, "crop_unique_mlp_pool: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(xyz && seed_xyz && rot && uidx && ucount && mlp && wpacked && bias && out,
             "crop_unique_mlp_pool: null pointer");
  GN_REQUIRE((long)n * 3 < 0x7fffffffL, "crop_unique_mlp_pool: n too large");
  if (!(ndepth == 4 && ns == 64 && mlp[0] == 64 && mlp[1] == 128 && mlp[2] == 256)) {
    set_error("crop_unique_mlp_pool: unsupported configuration depth=%d nsample=%d mlp=[%d,%d,%d]", ndepth, ns,
              mlp[0], mlp[1], mlp[2]);
    return GN_ERR_INVALID_ARG;
  }
  const Mlp3Weights W{wpacked[0], bias[0], wpacked[1], bias[1], wpacked[2], bias[2]};
  static const int variant = []() { const char *e = getenv("GN_CROP_VARIANT"); return e ? atoi(e) : 1; }();
  const int nseeds = b * m;
  int spb = nseeds / 2048;   // >= ~8 workgroups per CU in the grid; weights amortised over spb seeds
  spb = spb < 1 ? 1 : (spb > 8 ? 8 : spb);
  int rc;
  if (variant == 1) {        // default: 8 waves per workgroup (two per SIMD), 2-4 % faster end to end
    constexpr int TP = 128;
    constexpr size_t lds = sizeof(float) * (TP * lds_stride(128) + TP * lds_stride(64) + 2 * TP * lds_stride(8) + 2 * TP * 4);
    auto k = crop_unique8_mlp_kernel<64, 128, 256, 4, 256, TP>;
    if ((rc = set_lds(k, lds, "crop_unique_mlp_pool"))) return rc;
    hipLaunchKernelGGL(k, dim3((nseeds + spb - 1) / spb), dim3(512), lds, as_stream(stream), xyz, n, seed_xyz,
                       rot, uidx, ucount, m, nseeds, spb, W, out);
  } else {
    constexpr int TP = 128;
    constexpr size_t lds = sizeof(float) * (TP * lds_stride(128) + TP * lds_stride(64) + 2 * TP * lds_stride(8) + 2 * TP * 4);
    auto k = crop_unique_mlp_kernel<64, 128, 256, 4, 256, TP, 1>;
    if ((rc = set_lds(k, lds, "crop_unique_mlp_pool"))) return rc;
    hipLaunchKernelGGL(k, dim3((nseeds + spb - 1) / spb), dim3(MLP_THREADS), lds, as_stream(stream), xyz, n, seed_xyz,
                       rot, uidx, ucount, m, nseeds, spb, W, out);
  }
  GN_CHECK_LAUNCH("crop_unique_mlp_pool");
  return GN_OK;
}

// Set-abstraction tail with layer 1's feature part precomputed per point (gn_point_linear):
// pfeat (B,N,c1) = W1f * feat + b1.  wpacked[0] holds only the K=8 coordinate block of layer 1
// ([xyz(3) | 0]) with zero bias.  Bit-identical to gn_sa_group_mlp_pool.
extern "C" int gn_sa_hoisted_mlp_pool(const float *xyz, const float *pfeat, const float *new_xyz, const int32_t *idx,
                                      int b, int n, int np, int ns, float radius, const int *mlp,
                                      const float *const *wpacked, const float *const *bias, float *out,
                                      gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n > 0 && np > 0 && ns > 0, "sa_hoisted_mlp_pool: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(xyz && pfeat && new_xyz && idx && mlp && wpacked && bias && out, "sa_hoisted_mlp_pool: null pointer");
  GN_REQUIRE((long)n * 128 < 0x7fffffffL, "sa_hoisted_mlp_pool: n too large");
  if (!(mlp[0] == 128 && mlp[1] == 128 && mlp[2] == 256 && (ns == 32 || ns == 16))) {
    set_error("sa_hoisted_mlp_pool: unsupported configuration mlp=[%d,%d,%d] nsample=%d", mlp[0], mlp[1], mlp[2], ns);
    return GN_ERR_INVALID_ARG;
  }
  const Mlp3Weights W{wpacked[0], bias[0], wpacked[1], bias[1], wpacked[2], bias[2]};
  constexpr int TP = 128;
  GN_REQUIRE(np % (TP / ns) == 0, "sa_hoisted_mlp_pool: npoint must be a multiple of %d", TP / ns);
  const int ntiles = (int)((long)b * np / (TP / ns));
  constexpr size_t lds = sizeof(float) * (2 * TP * lds_stride(128) + 2 * TP * lds_stride(8));
  const int tpb = tiles_per_block(ntiles);
  hipStream_t s = as_stream(stream);
  int rc;
  if (ns == 32) {
    auto k = sa_hoist_kernel<128, 128, 256, 32>;
    if ((rc = set_lds(k, lds, "sa_hoisted_mlp_pool"))) return rc;
    hipLaunchKernelGGL(k, dim3((ntiles + tpb - 1) / tpb), dim3(MLP_THREADS), lds, s, xyz, n, pfeat, new_xyz, idx, np,
                       ntiles, tpb, radius, W, out);
  } else {
    auto k = sa_hoist_kernel<128, 128, 256, 16>;
    if ((rc = set_lds(k, lds, "sa_hoisted_mlp_pool"))) return rc;
    hipLaunchKernelGGL(k, dim3((ntiles + tpb - 1) / tpb), dim3(MLP_THREADS), lds, s, xyz, n, pfeat, new_xyz, idx, np,
                       ntiles, tpb, radius, W, out);
  }
  GN_CHECK_LAUNCH("sa_hoisted_mlp_pool");
  return GN_OK;
}

GN_PROBE_READER(gn_phase_probe_read)
