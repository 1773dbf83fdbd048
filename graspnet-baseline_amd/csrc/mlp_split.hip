// mlp_split.hip -- grouped MLP kernels with layers 2-3 on the bf16 matrix cores (split operands, mlp_split.h).
//
// gn_crop_unique_mlp_pool_split: CloudCrop's SharedMLP[3,64,128,256] + masked max-pool over the de-duplicated
// rows of each seed (models/modules.py:131-140, pointnet2_utils.py:521-530), same interface and row semantics as
// gn_crop_unique_mlp_pool; layer 1 (K = 3) stays on the fp32 MFMA.
#include <stdlib.h>

#include "mlp_split.h"

namespace gn {

struct SplitWeights {
  const float *w1, *b1;        // layer 1: fp32 fragments (mlp_core.h)
  const void *w2, *w3;         // layers 2-3: split-bf16 fragments (mlp_split.h)
  const float *b2, *b3;
};

constexpr int kCropMaxSpb = 16;   // seeds per workgroup the crop kernel's entry-count table holds (host clamps to it)

template <class K>
static int set_lds_attr(K kernel, size_t bytes, const char *name) {
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)bytes) != hipSuccess) {
    set_error("%s: cannot reserve %zu bytes of LDS", name, bytes);
    return GN_ERR_LAUNCH;
  }
  return GN_OK;
}

// One workgroup (4 waves) walks `spb` consecutive seeds; a seed's rows are taken in chunks of <= TP.
// LDS: Y2 3 planes [TP][C2+8] bf16, Y1 3 planes [TP][C1+8] bf16, X 2 x [TP][12] f32, gates 2 x [TP][4] f32.
template <int C1, int C2, int C3, int ND, int CAP, int TP, int PIPE>
__global__ __launch_bounds__(MLP_THREADS) void crop_unique_split_kernel(const float *__restrict__ xyz, int n,
                                                                        const float *__restrict__ seed,
                                                                        const float *__restrict__ rot,
                                                                        const int32_t *__restrict__ uidx,
                                                                        const int32_t *__restrict__ ucount, int m,
                                                                        int nseeds, int spb, SplitWeights W,
                                                                        float *__restrict__ out) {
  constexpr int CIN0 = 8, XS0 = lds_stride(CIN0);
  constexpr int P2 = TP * sp_stride(C2), P1 = TP * sp_stride(C1);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16 *bufY2 = reinterpret_cast<__bf16 *>(smem_raw);            // 3 x [TP][C2+8]
  __bf16 *bufY1 = bufY2 + 3 * P2;                                  // 3 x [TP][C1+8]
  float *bufX = reinterpret_cast<float *>(bufY1 + 3 * P1);         // 2 x [TP][12]
  float *bufG = bufX + 2 * TP * XS0;                               // 2 x [TP][4] pooling gates
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  LayerWS<C1, C2> w2;
  LayerWS<C2, C3> w3;
  using L2 = LayerWS<C1, C2>;
  using L3 = LayerWS<C2, C3>;
  w2.template load<true>(W.w2, wave, lane);     // 48 + 192 = 240 AGPRs of resident weights
  w3.template load<true>(W.w3, wave, lane);
  // Layer 1 (K = 3) on the vector ALUs, one (row, 8-channel) item per thread and M tile: 256 items per tile = exactly
  // one per thread, whatever the tile count (as 32x32 MFMA tiles it is 2 per M tile on 4 waves: 6 tiles on 4 waves in
  // the common 3-tile chunk, and the other two waves waited at the barrier).  The thread's 8 channels' weights come
  // out of the fp32 fragment image: W1[c][k] = Wp[((c/32)*64 + c%32)*4 + k].
  static_assert(C1 == 64, "8 threads x 8 channels cover a row");
  float w1v[8][3], b1v[8];
  {
    const int c0 = (tid & 7) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      const float *wp = W.w1 + (size_t)(((c >> 5) * 64 + (c & 31)) * 4);
      w1v[j][0] = wp[0]; w1v[j][1] = wp[1]; w1v[j][2] = wp[2];
      b1v[j] = W.b1[c];
    }
  }
  float b2t[16], b3[L3::NTW];
  load_bias_t(W.b2, wave % L2::WN, lane, b2t);
#pragma unroll
  for (int i = 0; i < L3::NTW; ++i) b3[i] = W.b3[((wave % L3::WN) * L3::NTW + i) * 32 + (lane & 31)];
  PoolMasked<ND, L3::NTW> pool{bufG, lane, {}};
  pool.reset();

  // ---- chunk walk.  The row fetch is a chain of dependent global loads (entry count -> uidx entry -> coordinates);
  // issued back to back it stalls the fetching waves for an L2 round trip per link at the top of every chunk.  So the
  // workgroup's entry counts go to LDS once, a chunk's uidx entries are requested two chunks ahead and its coordinates
  // (plus the seed's centre / rotation, whose address needs no load) one chunk ahead.
  constexpr int MAXSPB = kCropMaxSpb;
  __shared__ int s_cnt[MAXSPB];
  const int s0 = blockIdx.x * spb;
  const int s_end = min(nseeds, s0 + spb);
  if (s0 >= s_end) return;
  if (tid < MAXSPB) s_cnt[tid] = (s0 + tid < s_end) ? ucount[s0 + tid] : 0;
  __syncthreads();
  struct Cur { int sf, row0, nu; };
  auto advance = [&](const Cur &c) {
    Cur r = c;
    if (c.row0 + TP >= c.nu) {
      r.sf = c.sf + 1;
      r.row0 = 0;
      r.nu = r.sf < s_end ? s_cnt[r.sf - s0] : 0;
    } else {
      r.row0 = c.row0 + TP;
    }
    return r;
  };
  auto load_e = [&](const Cur &c) -> int {      // -1 = padding row
    return (tid < TP && c.sf < s_end && c.row0 + tid < c.nu) ? uidx[(size_t)c.sf * CAP + c.row0 + tid] : -1;
  };
  // raw row of chunk c: coordinates of its point, centre and rotation of its seed (no arithmetic yet)
  struct Raw { float px, py, pz, c0, c1, c2, r[9]; };
  auto load_raw = [&](const Cur &c, int e, Raw &w) {
    if (e != -1) {
      const float *pp = xyz + ((size_t)(c.sf / m) * n + (e & 0x0fffffff)) * 3;
      const float *cc = seed + (size_t)c.sf * 3;
      const float *rr = rot + (size_t)c.sf * 9;
      w.px = pp[0]; w.py = pp[1]; w.pz = pp[2];
      w.c0 = cc[0]; w.c1 = cc[1]; w.c2 = cc[2];
#pragma unroll
      for (int i = 0; i < 9; ++i) w.r[i] = rr[i];
    }
  };
  auto put = [&](int buf, int e, const Raw &w) {
    if (tid >= TP) return;
    f32x4 v{0.f, 0.f, 0.f, 0.f}, g{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    if (e != -1) {
      const unsigned mk = (unsigned)e >> 28;
      const float x = w.px - w.c0, y = w.py - w.c1, z = w.pz - w.c2;
      v[0] = x * w.r[0] + y * w.r[3] + z * w.r[6];   // (p-c)^T R, pointnet2_utils.py:528-530
      v[1] = x * w.r[1] + y * w.r[4] + z * w.r[7];
      v[2] = x * w.r[2] + y * w.r[5] + z * w.r[8];
#pragma unroll
      for (int d = 0; d < ND; ++d) g[d] = ((mk >> d) & 1u) ? 0.f : -INFINITY;
    }
    float *xo = bufX + buf * TP * XS0 + tid * XS0;
    *reinterpret_cast<f32x4 *>(xo) = v;
    *reinterpret_cast<f32x4 *>(xo + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4 *>(bufG + buf * TP * 4 + tid * 4) = g;
  };

  Cur cur{s0, 0, s_cnt[0]};
  Cur n1 = advance(cur), n2 = advance(n1);
  int buf = 0, ent1, ent2 = -1;
  {
    const int e0 = load_e(cur);
    ent1 = load_e(n1);
    Raw w0{};
    load_raw(cur, e0, w0);
    put(0, e0, w0);
  }
  __syncthreads();
  GN_PROBE_INIT();
  while (true) {
    const int sf = cur.sf;
    const int rows = min(TP, cur.nu - cur.row0);
    const int mtc = (rows + 31) >> 5;
    const bool seed_done = cur.row0 + TP >= cur.nu;
    const bool has_next = n1.sf < s_end;
    Raw w1{};
    if (has_next) {
      load_raw(n1, ent1, w1);         // its entries were requested a whole chunk ago
      ent2 = load_e(n2);              // entries of the chunk after next
    }
    GN_PROBE(0);

    pool.sgate = bufG + buf * TP * 4;
    {
      const float *xin = bufX + buf * TP * XS0;
      for (int mt = 0; mt < mtc; ++mt) {
        const int row = mt * 32 + (tid >> 3);
        const f32x4 xv = *reinterpret_cast<const f32x4 *>(xin + row * XS0);
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          y[j] = __builtin_fmaf(xv[2], w1v[j][2], __builtin_fmaf(xv[1], w1v[j][1], __builtin_fmaf(xv[0], w1v[j][0], b1v[j])));
        u32x2 h0, m0, l0, h1, m1, l1;
        split4<true>(y[0], y[1], y[2], y[3], h0, m0, l0);
        split4<true>(y[4], y[5], y[6], y[7], h1, m1, l1);
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        __bf16 *dst = bufY1 + row * sp_stride(C1) + (tid & 7) * 8;
        *reinterpret_cast<u32x4 *>(dst) = u32x4{h0[0], h0[1], h1[0], h1[1]};
        *reinterpret_cast<u32x4 *>(dst + P1) = u32x4{m0[0], m0[1], m1[0], m1[1]};
        *reinterpret_cast<u32x4 *>(dst + 2 * P1) = u32x4{l0[0], l0[1], l1[0], l1[1]};
      }
    }
    GN_PROBE(1);
    __syncthreads();
    GN_PROBE(2);
    StoreSplit e2{bufY2, sp_stride(C2), P2, lane};
    if constexpr (PIPE & 1)
      run_layer_split_t_pipe<TP, C1, C2>(w2, b2t, bufY1, P1, wave, lane, e2, mtc);
    else
      run_layer_split_t<TP, C1, C2>(w2, b2t, bufY1, P1, wave, lane, e2, mtc);
    GN_PROBE(3);
    // the next chunk's input tile and gates go into the other halves of their double buffers here, so that this
    // barrier also publishes them and no third barrier is needed: layer 1 of the next chunk only writes Y1 (nobody
    // reads it any more) and layer 2 of the next chunk starts behind its own barrier, i.e. after every wave has left
    // this chunk's layer 3
    if (has_next) put(buf ^ 1, ent1, w1);
    GN_PROBE(7);
    __syncthreads();
    GN_PROBE(4);
    if constexpr ((PIPE & 6) == 6)
      run_layer_split_n_pool_pipe2<TP, C2, C3, ND>(w3, b3, bufY2, P2, wave, lane, pool, mtc);
    else if constexpr (PIPE & 2)
      run_layer_split_n_pool_pipe<TP, C2, C3, ND>(w3, b3, bufY2, P2, wave, lane, pool, mtc);
    else
      run_layer_split_n<TP, C2, C3>(w3, b3, bufY2, P2, wave, lane, pool, mtc);
    GN_PROBE(5);
    if (seed_done) {
      pool.finish(out + (size_t)sf * ND * C3, C3, (wave % L3::WN) * L3::NTW);
      pool.reset();
    }
    GN_PROBE(6);
    if (!has_next) break;
    GN_PROBE_COUNT(9, 1);
    GN_PROBE_COUNT(10, mtc);
    buf ^= 1;
    cur = n1; n1 = n2; n2 = advance(n2);
    ent1 = ent2;
  }
  GN_PROBE_FLUSH();
}

// ---- CloudCrop on PACKED rows (the default).  crop_unique_split_kernel above pads every seed's rows to whole 32-row M
// tiles (85 unique rows per seed on the headline clouds -> 102.6 evaluated: 17 % of the MFMAs are padding) and starts a
// new chunk at every seed (2.3 of the 3 M tiles a chunk can hold).  Here a workgroup's `spb` seeds form ONE row stream
// -- seed after seed, no padding -- cut into full TP-row chunks.  A 32-row M tile may then hold rows of two (rarely more)
// seeds: the staging threads write TWO gate tables per row, GA (the row's gates if it belongs to the tile's first seed,
// else -inf) and GB (the same for the tile's second seed), so a tile is pooled with GA under the next tile's MFMAs
// exactly like a one-seed tile, and a second seed costs one more pooling pass with GB; tiles with three or more seeds
// (seeds with a handful of rows) take a generic path.  Layer 1 of chunk c+1 (vector ALUs) rides on the layer-3 MFMAs of
// chunk c.  Every row goes through the same MFMAs in the same k order and max() is exact, so the result is bit-identical
// to the padded kernel (tests/test_split_gpu.py).
// RIDE: layer 1 of the next chunk under this chunk's layer-3 MFMAs (else on its own at the top of its chunk)
template <int C1, int C2, int C3, int ND, int CAP, int TP, bool RIDE, int NWV = 4>
__global__ __launch_bounds__(NWV * 64) void crop_packed_split_kernel(const float *__restrict__ xyz, int n,
                                                                        const float *__restrict__ seed,
                                                                        const float *__restrict__ rot,
                                                                        const int32_t *__restrict__ uidx,
                                                                        const int32_t *__restrict__ ucount, int m,
                                                                        int nseeds, int spb, SplitWeights W,
                                                                        float *__restrict__ out, int nb) {
  constexpr int CIN0 = 8, XS0 = lds_stride(CIN0), MT = TP / 32;
  static_assert(MT == 3, "the layer-3 walk below is written out for three M tiles per chunk");
  constexpr int P2 = TP * sp_stride(C2), P1 = TP * sp_stride(C1);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16 *bufY2 = reinterpret_cast<__bf16 *>(smem_raw);            // 3 x [TP][C2+8]
  __bf16 *bufY1 = bufY2 + 3 * P2;                                  // 3 x [TP][C1+8]
  float *bufX = reinterpret_cast<float *>(bufY1 + 3 * P1);         // 2 x [TP+1][12]: rotated coordinates (+1 dump row)
  float *bufGA = bufX + 2 * (TP + 1) * XS0;                        // 3 x [TP][4] gates towards the tile's first seed (chunk % 3)
  float *bufGB = bufGA + 3 * TP * 4;                               // 3 x [TP][4] ... towards its second seed
  int *bufS = reinterpret_cast<int *>(bufGB + 3 * TP * 4);         // 2 x [TP] seed slot | depth mask << 8 (-1: no row)
  float *bufW1 = reinterpret_cast<float *>(bufS + 2 * TP);         // [C1][4] layer 1: w0 w1 w2 bias
  __bf16 *dumpY1 = reinterpret_cast<__bf16 *>(bufW1 + C1 * 4);     // [8 x 8] bf16 x 3 planes: sink of an idle layer-1 slice
  __shared__ int s_off[kCropMaxSpb + 1];                           // stream offset of each seed slot; [slots..] = total
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  using L2 = LayerWS<C1, C2, NWV>;
  using L3 = LayerWS<C2, C3, NWV>;
  static_assert(L3::WM == 1, "every wave walks all M tiles of layer 3");
  static_assert(C1 == 64, "8 threads x 8 channels cover a row");
  static_assert(NWV == 4 || !RIDE, "layer 1 rides on the layer-3 units of the 4-wave form only");
  constexpr int L1ROWS = NWV * 64 / 8;       // rows one pass of layer 1 covers (8 threads per row)
  constexpr int NTW = L3::NTW;
  L2 w2;
  L3 w3;
  w2.template load<true>(W.w2, wave, lane);
  w3.template load<true>(W.w3, wave, lane);
  if (tid < C1) {     // layer 1 (K = 3, on the vector ALUs) out of the fp32 fragment image: W1[c][k] = Wp[((c/32)*64 + c%32)*4 + k]
    const float *wp = W.w1 + (size_t)(((tid >> 5) * 64 + (tid & 31)) * 4);
    *reinterpret_cast<f32x4 *>(bufW1 + tid * 4) = f32x4{wp[0], wp[1], wp[2], W.b1[tid]};
  }
  float b2t[16], b3[NTW];
  load_bias_t(W.b2, wave % L2::WN, lane, b2t);
#pragma unroll
  for (int i = 0; i < NTW; ++i) b3[i] = W.b3[((wave % L3::WN) * NTW + i) * 32 + (lane & 31)];
  PoolMasked<ND, NTW> pool{bufGA, lane, {}};
  pool.reset();

  // XCD-aware workgroup order (nb > 0: the grid holds nb clouds of gridDim.x / nb workgroups each): consecutive workgroup
  // ids go to consecutive XCDs, so cloud = id % nb keeps a cloud's workgroups -- and its coordinates -- in one XCD's L2
  const int wg = nb > 0 ? (int)(blockIdx.x % nb) * (int)(gridDim.x / nb) + (int)(blockIdx.x / nb) : (int)blockIdx.x;
  const int s0 = wg * spb;
  const int nsd = min(nseeds, s0 + spb) - s0;       // seed slots of this workgroup (<= kCropMaxSpb)
  if (nsd <= 0) return;
  if (tid <= kCropMaxSpb) s_off[tid] = (tid >= 1 && tid <= nsd) ? ucount[s0 + tid - 1] : 0;
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 1; k <= kCropMaxSpb; ++k) { run += s_off[k]; s_off[k] = run; }    // slots past nsd repeat the total
  }
  __syncthreads();
  const int R = s_off[nsd];                          // rows of the stream
  const int nchunks = (R + TP - 1) / TP;
  auto slot_of = [&](int p) {                        // seed slot of stream position p < R
    int sl = 0;
#pragma unroll
    for (int k = 1; k < kCropMaxSpb; ++k) sl += (p >= s_off[k]) ? 1 : 0;
    return sl;
  };
  // entry (and its seed slot) of this thread's row in chunk c; -1 = no row
  auto load_e = [&](int c, int &slot) -> int {
    const int p = c * TP + tid;
    slot = 0;
    if (tid >= TP || c >= nchunks || p >= R) return -1;
    slot = slot_of(p);
    return uidx[(size_t)(s0 + slot) * CAP + (p - s_off[slot])];
  };
  struct Raw { float px, py, pz, c0, c1, c2, r[9]; };
  auto load_raw = [&](int e, int slot, Raw &w) {
    if (e != -1) {
      const int sf = s0 + slot;
      const float *pp = xyz + ((size_t)(sf / m) * n + (e & 0x0fffffff)) * 3;
      const float *cc = seed + (size_t)sf * 3;
      const float *rr = rot + (size_t)sf * 9;
      w.px = pp[0]; w.py = pp[1]; w.pz = pp[2];
      w.c0 = cc[0]; w.c1 = cc[1]; w.c2 = cc[2];
#pragma unroll
      for (int i = 0; i < 9; ++i) w.r[i] = rr[i];
    }
  };
  // stage row `tid` of chunk c into buffer `buf`
  auto put = [&](int buf, int c, int e, int slot, const Raw &w) {
    if (tid >= TP) return;
    const f32x4 ninf{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 v{0.f, 0.f, 0.f, 0.f}, ga = ninf, gb = ninf;
    int sv = -1;
    if (e != -1) {
      const unsigned mk = (unsigned)e >> 28;
      const float x = w.px - w.c0, y = w.py - w.c1, z = w.pz - w.c2;
      v[0] = x * w.r[0] + y * w.r[3] + z * w.r[6];   // (p-c)^T R, pointnet2_utils.py:528-530
      v[1] = x * w.r[1] + y * w.r[4] + z * w.r[7];
      v[2] = x * w.r[2] + y * w.r[5] + z * w.r[8];
      f32x4 g;
#pragma unroll
      for (int d = 0; d < ND; ++d) g[d] = ((mk >> d) & 1u) ? 0.f : -INFINITY;
      const int a_tile = slot_of(c * TP + (tid & ~31));      // seed of the tile's first row
      if (slot == a_tile) ga = g;
      if (slot == a_tile + 1) gb = g;
      // slot | depth mask << 8 | (last row of its seed) << 16
      sv = slot | (int)(mk << 8) | ((c * TP + tid + 1 == s_off[slot + 1]) ? 0x10000 : 0);
    }
    float *xo = bufX + buf * (TP + 1) * XS0 + tid * XS0;
    *reinterpret_cast<f32x4 *>(xo) = v;
    *reinterpret_cast<f32x4 *>(xo + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4 *>(bufGA + (c % 3) * TP * 4 + tid * 4) = ga;
    *reinterpret_cast<f32x4 *>(bufGB + (c % 3) * TP * 4 + tid * 4) = gb;
    bufS[buf * TP + tid] = sv;
  };

  // ---- layer 1 of one M tile, one (row, 8-channel) item per thread, cut into six slices so that it can ride on the k-blocks
  // of a layer-3 unit (slice s at k-block s): 0 coordinates, 1-2 the 24 FMAs, 3-4 relu + split, 5 the three 16-byte stores
  // (every LDS read of a slice is consumed by a LATER slice: a wait inside a slice would stall the MFMAs behind it)
  f32x4 l1x, l1wv[4];
  float l1y[8];
  u32x2 l1h0, l1m0, l1l0, l1h1, l1m1, l1l1;
  const float *l1w = bufW1 + (tid & 7) * 8 * 4;
  auto l1_slice = [&](int sl, const float *xrow, __bf16 *dst, int plane) {
    if (sl == 1 || sl == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        l1y[4 * (sl - 1) + j] = __builtin_fmaf(l1x[2], l1wv[j][2], __builtin_fmaf(l1x[1], l1wv[j][1], __builtin_fmaf(l1x[0], l1wv[j][0], l1wv[j][3])));
    }
    if (sl == 0) l1x = *reinterpret_cast<const f32x4 *>(xrow);
    if (sl == 0 || sl == 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) l1wv[j] = *reinterpret_cast<const f32x4 *>(l1w + (4 * sl + j) * 4);
    }
    if (sl == 3) split4<true>(l1y[0], l1y[1], l1y[2], l1y[3], l1h0, l1m0, l1l0);
    if (sl == 4) split4<true>(l1y[4], l1y[5], l1y[6], l1y[7], l1h1, l1m1, l1l1);
    if (sl == 5) {
      using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
      *reinterpret_cast<u32x4 *>(dst) = u32x4{l1h0[0], l1h0[1], l1h1[0], l1h1[1]};
      *reinterpret_cast<u32x4 *>(dst + plane) = u32x4{l1m0[0], l1m0[1], l1m1[0], l1m1[1]};
      *reinterpret_cast<u32x4 *>(dst + 2 * plane) = u32x4{l1l0[0], l1l0[1], l1l1[0], l1l1[1]};
    }
  };
  // source row / destination of tile mt of the chunk staged in buffer b; inactive -> the dump row / sink (no branch)
  auto l1_src = [&](int b, int mt, bool active) {
    return bufX + b * (TP + 1) * XS0 + (active ? mt * 32 + (tid >> 3) : TP) * XS0;
  };
  auto l1_dst = [&](int mt, bool active) {
    return active ? bufY1 + (mt * 32 + (tid >> 3)) * sp_stride(C1) + (tid & 7) * 8 : dumpY1 + (tid & 63) * 8;
  };
  // layer 1 of a whole chunk on its own: passes of L1ROWS rows (a thread whose row lies past the chunk's tiles sits out)
  auto l1_chunk = [&](int b, int nrows) {
    for (int r0 = 0; r0 < nrows; r0 += L1ROWS) {
      const int row = r0 + (tid >> 3);
      if (row < nrows) {
        const float *src = bufX + b * (TP + 1) * XS0 + row * XS0;
        __bf16 *dst = bufY1 + row * sp_stride(C1) + (tid & 7) * 8;
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) l1_slice(sl, src, dst, P1);
      }
    }
  };

  int buf = 0, ent1, ent2 = -1, sl1 = 0, sl2 = 0;
  {
    int sl0;
    const int e0 = load_e(0, sl0);
    ent1 = load_e(1, sl1);
    Raw w0{};
    load_raw(e0, sl0, w0);
    put(0, 0, e0, sl0, w0);
    if (tid < XS0) bufX[TP * XS0 + tid] = bufX[(TP + 1) * XS0 + TP * XS0 + tid] = 0.f;     // the two dump rows
  }
  __syncthreads();
  {   // layer 1 of the first chunk on its own
    const int mtc0 = (min(TP, R) + 31) >> 5;
    l1_chunk(0, mtc0 * 32);
  }
  const int nt0 = (wave % L3::WN) * NTW;
  GN_PROBE_INIT();
  for (int c = 0; c < nchunks; ++c) {
    const int rows = min(TP, R - c * TP);
    const int mtc = (rows + 31) >> 5;
    const bool has_next = c + 1 < nchunks;
    const int mtc_next = has_next ? (min(TP, R - (c + 1) * TP) + 31) >> 5 : 0;
    Raw w1{};
    if (has_next) {
      load_raw(ent1, sl1, w1);         // its entries were requested a whole chunk ago
      ent2 = load_e(c + 2, sl2);       // entries of the chunk after next
    }
    // seeds of every tile of the chunk: the slot words of its first and last row, requested here so that the LDS round
    // trip hides under layer 2 (consumed after it)
    int rwa[MT], rwz[MT];
    {
      const int *sch_ = bufS + buf * TP;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        rwa[mt] = rwz[mt] = 0;
        if (mt < mtc) {
          rwa[mt] = sch_[mt * 32];
          rwz[mt] = sch_[min(rows, mt * 32 + 32) - 1];
        }
      }
    }
    GN_PROBE(0);
    if constexpr (!RIDE) {
      if (c > 0) l1_chunk(buf, mtc * 32);
    }
    GN_PROBE(1);
    __syncthreads();                   // layer 1 of this chunk (done under the previous chunk's layer 3) is published
    GN_PROBE(2);
    StoreSplit e2{bufY2, sp_stride(C2), P2, lane};
    run_layer_split_t_pipe<TP, C1, C2, NWV>(w2, b2t, bufY1, P1, wave, lane, e2, mtc);
    GN_PROBE(3);
    if (has_next) put(buf ^ 1, c + 1, ent1, sl1, w1);     // published by the barrier below
    GN_PROBE(7);
    __syncthreads();
    GN_PROBE(4);

    // ---- layer 3 + pooling, layer 1 of the next chunk riding along.  Tile t is pooled (gates GA: rows of its first seed)
    // under the MFMAs of tile t+1; what cannot ride -- the chunk's last tile, a second seed's pass with GB -- is pooled on
    // its own with all its gates fetched up front (carrying the last tile's accumulators across the next chunk's layer 2
    // to let it ride too costs more registers than the kernel has: measured slower).
    const float *gAc = bufGA + (c % 3) * TP * 4 + (4 * (lane >> 5)) * 4;
    const float *gBc = bufGB + (c % 3) * TP * 4 + (4 * (lane >> 5)) * 4;
    const int *sch = bufS + buf * TP;
    const int p0 = c * TP;
    const __bf16 *xbase = bufY2 + (lane & 31) * sp_stride(C2) + 8 * (lane >> 5);
    auto xr = [&](int mt) { return xbase + mt * 32 * sp_stride(C2); };
    // first / last seed of each tile, and whether the last one ends inside it (= the tile's last row is that seed's last)
    int ta[MT], tz[MT];
    bool tend[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int wa = __builtin_amdgcn_readfirstlane(rwa[mt]), wz = __builtin_amdgcn_readfirstlane(rwz[mt]);
      ta[mt] = wa & 0xff;
      tz[mt] = wz & 0xff;
      tend[mt] = (wz & 0x10000) != 0;
    }
    auto finish_seed = [&](int q) {
      GN_PROBE(5);
      pool.finish(out + (size_t)(s0 + q) * ND * C3, C3, nt0);
      pool.reset();
      GN_PROBE(13);
    };
    // the pooling pass with GA of tile mt (seeds a..z, z <= a+1) has completed: close the tile
    auto close_tile = [&](const f32x16 (&src)[NTW], int mt) {
      if (tz[mt] == ta[mt]) {
        if (tend[mt]) finish_seed(ta[mt]);
      } else {            // the first seed ends inside the tile, the second gets its own pass
        finish_seed(ta[mt]);
        GN_PROBE(5);
        pool_tile<ND, NTW>(pool, src, gBc + mt * 32 * 4);
        GN_PROBE(12);
        if (tend[mt]) finish_seed(tz[mt]);
      }
    };
    // three or more seeds in one tile (seeds of a handful of rows): pooled on the spot, one pass per seed
    auto pool_many = [&](const f32x16 (&src)[NTW], int mt) {
      for (int q = ta[mt]; q <= tz[mt]; ++q) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int rw = mt * 32 + acc_row(2 * j, lane);
          const int s0w = sch[rw], s1w = sch[rw + 1];
          f32x4 g0, g1;
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            g0[d] = ((s0w & 0xff) == q && s0w >= 0 && ((s0w >> (8 + d)) & 1)) ? 0.f : -INFINITY;
            g1[d] = ((s1w & 0xff) == q && s1w >= 0 && ((s1w >> (8 + d)) & 1)) ? 0.f : -INFINITY;
          }
#pragma unroll
          for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int d = 0; d < ND; ++d)
              pool.hold[d][i] = fmaxf(pool.hold[d][i], fmaxf(src[i][2 * j] + g0[d], src[i][2 * j + 1] + g1[d]));
        }
        if (q < tz[mt] || s_off[q + 1] <= p0 + mt * 32 + 32) finish_seed(q);
      }
    };
    f32x16 accA[NTW], accB[NTW];
    // one unit: MFMAs of tile MTI into DST; PEND: tile MTI-1 (in PRV) is pooled with GA under them; layer 1 of the next
    // chunk's tile MTI rides along
#define GN_CROP_UNIT(DST, PRV, MTI, PEND)                                                                                \
    do {                                                                                                                 \
      const bool l1a_ = (MTI) < mtc_next;                                                                                \
      const float *l1s_ = l1_src(buf ^ 1, (MTI), l1a_);                                                                  \
      __bf16 *l1d_ = l1_dst((MTI), l1a_);                                                                                \
      auto hook_ = [&](int kb) { if (kb < 6) l1_slice(kb, l1s_, l1d_, l1a_ ? P1 : 64 * 8); };                            \
      if constexpr (RIDE) {                                                                                              \
        if (PEND) split_n_unit<1, C2, C3, ND, NWV>(w3, b3, xr(MTI), P2, DST, PRV, pool, gAc + ((MTI) - 1) * 32 * 4, pool, gAc, hook_); \
        else split_n_unit<0, C2, C3, ND, NWV>(w3, b3, xr(MTI), P2, DST, DST, pool, gAc, pool, gAc, hook_);                     \
      } else {                                                                                                           \
        if (PEND) split_n_unit<1, C2, C3, ND, NWV>(w3, b3, xr(MTI), P2, DST, PRV, pool, gAc + ((MTI) - 1) * 32 * 4, pool, gAc); \
        else split_n_unit<0, C2, C3, ND, NWV>(w3, b3, xr(MTI), P2, DST, DST, pool, gAc, pool, gAc);                       \
      }                                                                                                                  \
    } while (0)
    GN_PROBE(14);
    GN_CROP_UNIT(accA, accA, 0, false);
    GN_PROBE(6);
    bool pend = tz[0] <= ta[0] + 1;
    if (!pend) pool_many(accA, 0);
    if (mtc > 1) {
      GN_PROBE(5);
      if (pend) { GN_CROP_UNIT(accB, accA, 1, true); GN_PROBE(8); close_tile(accA, 0); }
      else { GN_CROP_UNIT(accB, accB, 1, false); GN_PROBE(6); }
      pend = tz[1] <= ta[1] + 1;
      if (!pend) pool_many(accB, 1);
      if (mtc > 2) {
        GN_PROBE(5);
        if (pend) { GN_CROP_UNIT(accA, accB, 2, true); GN_PROBE(8); close_tile(accB, 1); }
        else { GN_CROP_UNIT(accA, accA, 2, false); GN_PROBE(6); }
        if (tz[2] <= ta[2] + 1) { GN_PROBE(5); pool_tile<ND, NTW>(pool, accA, gAc + 2 * 32 * 4); GN_PROBE(12); close_tile(accA, 2); }
        else pool_many(accA, 2);
      } else if (pend) {
        pool_tile<ND, NTW>(pool, accB, gAc + 1 * 32 * 4);
        close_tile(accB, 1);
      }
    } else if (pend) {
      pool_tile<ND, NTW>(pool, accA, gAc);
      close_tile(accA, 0);
    }
#undef GN_CROP_UNIT
    // a next chunk with more tiles than this one had units cannot occur (only the last chunk is short)
    GN_PROBE(5);
    GN_PROBE_COUNT(9, 1);
    GN_PROBE_COUNT(10, mtc);
    buf ^= 1;
    ent1 = ent2;
    sl1 = sl2;
  }
  GN_PROBE_FLUSH();
}

// ---- CloudCrop on packed rows, EIGHT waves: two per SIMD, the two wave groups on DIFFERENT JOBS -------------------------
// (round-3 VERDICT item 4)  The 4-wave kernel above runs ONE wave per SIMD (496 registers: 240 of
// resident weights) and its phases are serial inside that wave: the matrix core idles through every epilogue, pooling
// tail, staging step and barrier (MFMA-busy 44 %).  Two waves per SIMD only help if, whenever one of them is NOT issuing
// MFMAs, the other one is -- eight waves walking the same phases between the same barriers do not (measured: 0.33-0.35 ms
// against 0.36; profiles/README.md).  So the two groups are given different jobs inside a slot of 32 rows (one M tile):
//   group A (waves 0-3: one per SIMD):  layer 3 of chunk c (n-tiles 0-3)  ->  layer 1 of chunk c+2  ->  staging of chunk c+3
//   group B (waves 4-7)              :  layer 2 of chunk c+1 (its four n-tiles)  ->  layer 3 of chunk c (n-tiles 4-7)
// and ONE workgroup barrier per slot: Y1 (layer 1's output), Y2 (layer 2's) and the staged coordinates are all
// double-buffered, so every hazard lies between slots, never inside one, and the groups drift freely within a slot.
//   * every wave keeps ONE of layer 3's eight n-tiles resident (96 AGPRs of its 256 registers); layer 2's fragments (a
//     group-B wave's n-tile: 12 KB) are streamed from L2, requested in front of the slot barrier they are used behind;
//   * a chunk is ONE 32-row tile -- the double buffers of a 64-row chunk need 176 KB of LDS -- so nothing rides inside a
//     wave: a tile's pooling, close-outs and epilogues run under the OTHER wave's MFMAs.
// Rows, k order, products and max() are those of the 4-wave kernel: bit-identical output (tests/test_split_gpu.py).
template <int C1, int C2, int C3, int ND, int CAP>
__global__ __launch_bounds__(512) void crop8_split_kernel(const float *__restrict__ xyz, int n, const float *__restrict__ seed,
                                                          const float *__restrict__ rot, const int32_t *__restrict__ uidx,
                                                          const int32_t *__restrict__ ucount, int m, int nseeds, int spb,
                                                          SplitWeights W, float *__restrict__ out, int nb) {
  constexpr int TP = 32, NWV = 8, CIN0 = 8, XS0 = lds_stride(CIN0), RING = 4;
  static_assert(C1 == 64 && C2 == 128 && C3 == 256, "written for CloudCrop's 64-128-256 layers");
  constexpr int P2 = TP * sp_stride(C2), P1 = TP * sp_stride(C1);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16 *bufY2 = reinterpret_cast<__bf16 *>(smem_raw);            // 2 x 3 x [TP][C2+8]   (chunk & 1)
  __bf16 *bufY1 = bufY2 + 2 * 3 * P2;                              // 2 x 3 x [TP][C1+8]   (chunk & 1)
  float *bufX = reinterpret_cast<float *>(bufY1 + 2 * 3 * P1);     // 2 x [TP][12]: rotated coordinates (chunk & 1)
  float *bufGA = bufX + 2 * TP * XS0;                              // RING x [TP][4] gates towards the tile's first seed
  float *bufGB = bufGA + RING * TP * 4;                            // RING x [TP][4] ... towards its second seed
  int *bufS = reinterpret_cast<int *>(bufGB + RING * TP * 4);      // RING x [TP] seed slot | depth mask << 8 (-1: no row)
  float *bufW1 = reinterpret_cast<float *>(bufS + RING * TP);      // [C1][4] layer 1: w0 w1 w2 bias
  float *bufB2 = bufW1 + C1 * 4;                                   // [C2] layer-2 bias
  __shared__ int s_off[kCropMaxSpb + 1];
  __shared__ float s_seed[kCropMaxSpb][12];                        // per seed slot: centre (3) + rotation (9)
  // the wave index as a SCALAR: the group branches below must be uniform branches
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool groupA = wave < 4;                                    // SIMD s holds waves s (group A) and s + 4 (group B)

  using L2 = LayerWS<C1, C2, 4>;        // the four group-B waves: n-tile wave - 4
  using L3 = LayerWS<C2, C3, NWV>;      // n-tile wave
  static_assert(L2::NTW == 1 && L3::NTW == 1 && L2::KB == 4 && L3::KB == 8 && L2::WM == 1 && L3::WM == 1, "tiling");
  L3 w3;
  w3.template load<true>(W.w3, wave, lane);
  const int wn2 = wave & 3;
  const char *w2base = static_cast<const char *>(W.w2);
  auto w2_at = [&](int kb, int p) {
    return reinterpret_cast<const bf16x8 *>(w2base + ((((size_t)(wn2 * L2::KB + kb)) * 3 + p) * 64 + lane) * 16);
  };
  if (tid < C1) {
    const float *wp = W.w1 + (size_t)(((tid >> 5) * 64 + (tid & 31)) * 4);
    *reinterpret_cast<f32x4 *>(bufW1 + tid * 4) = f32x4{wp[0], wp[1], wp[2], W.b1[tid]};
  }
  if (tid >= 64 && tid < 64 + C2) bufB2[tid - 64] = W.b2[tid - 64];
  float b3[1];
  b3[0] = W.b3[wave * 32 + (lane & 31)];
  PoolMasked<ND, 1> pool{bufGA, lane, {}};
  pool.reset();

  const int wg = nb > 0 ? (int)(blockIdx.x % nb) * (int)(gridDim.x / nb) + (int)(blockIdx.x / nb) : (int)blockIdx.x;   // XCD-aware
  const int s0 = wg * spb;
  const int nsd = min(nseeds, s0 + spb) - s0;
  if (nsd <= 0) return;
  if (tid <= kCropMaxSpb) s_off[tid] = (tid >= 1 && tid <= nsd) ? ucount[s0 + tid - 1] : 0;
  if (tid >= 64 && tid < 64 + kCropMaxSpb * 12) {
    const int sl = (tid - 64) / 12, j = (tid - 64) % 12;
    if (sl < nsd) s_seed[sl][j] = j < 3 ? seed[(size_t)(s0 + sl) * 3 + j] : rot[(size_t)(s0 + sl) * 9 + j - 3];
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int k = 1; k <= kCropMaxSpb; ++k) { run += s_off[k]; s_off[k] = run; }
  }
  __syncthreads();
  const int R = s_off[nsd];
  const int nchunks = (R + TP - 1) / TP;
  auto slot_of = [&](int p) {
    int sl = 0;
#pragma unroll
    for (int k = 1; k < kCropMaxSpb; ++k) sl += (p >= s_off[k]) ? 1 : 0;
    return sl;
  };
  GN_PROBE_INIT();
  // ---- staging (threads 0 .. TP-1, group A): row `tid` of a chunk
  const bool stager = tid < TP;
  auto load_e = [&](int c, int &slot) -> int {
    const int p = c * TP + tid;
    slot = 0;
    if (!stager || c >= nchunks || p >= R) return -1;
    slot = slot_of(p);
    return uidx[(size_t)(s0 + slot) * CAP + (p - s_off[slot])];
  };
  struct Raw { float px, py, pz; };
  auto load_raw = [&](int e, int slot, Raw &w) {
    if (e != -1) {
      const float *pp = xyz + ((size_t)((s0 + slot) / m) * n + (e & 0x0fffffff)) * 3;
      w.px = pp[0]; w.py = pp[1]; w.pz = pp[2];
    }
  };
  auto put = [&](int c, int e, int slot, const Raw &w) {
    if (!stager) return;
    const f32x4 ninf{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    f32x4 v{0.f, 0.f, 0.f, 0.f}, ga = ninf, gb = ninf;
    int sv = -1;
    if (e != -1) {
      const unsigned mk = (unsigned)e >> 28;
      const float *sd = s_seed[slot];
      const float x = w.px - sd[0], y = w.py - sd[1], z = w.pz - sd[2];
      v[0] = x * sd[3] + y * sd[6] + z * sd[9];      // (p-c)^T R, pointnet2_utils.py:528-530
      v[1] = x * sd[4] + y * sd[7] + z * sd[10];
      v[2] = x * sd[5] + y * sd[8] + z * sd[11];
      f32x4 g;
#pragma unroll
      for (int d = 0; d < ND; ++d) g[d] = ((mk >> d) & 1u) ? 0.f : -INFINITY;
      const int a_tile = slot_of(c * TP);
      if (slot == a_tile) ga = g;
      if (slot == a_tile + 1) gb = g;
      sv = slot | (int)(mk << 8) | ((c * TP + tid + 1 == s_off[slot + 1]) ? 0x10000 : 0);
    }
    float *xo = bufX + (c & 1) * TP * XS0 + tid * XS0;
    *reinterpret_cast<f32x4 *>(xo) = v;
    *reinterpret_cast<f32x4 *>(xo + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<f32x4 *>(bufGA + (c % RING) * TP * 4 + tid * 4) = ga;
    *reinterpret_cast<f32x4 *>(bufGB + (c % RING) * TP * 4 + tid * 4) = gb;
    bufS[(c % RING) * TP + tid] = sv;
  };
  // ---- layer 1 (K = 3, vector ALUs) by group A's 256 threads: one (row, 8 channels) item per thread, into Y1[c & 1]
  const float *l1w = bufW1 + (tid & 7) * 8 * 4;
  auto l1_chunk = [&](int c) {
    const int row = tid >> 3;
    const f32x4 x = *reinterpret_cast<const f32x4 *>(bufX + (c & 1) * TP * XS0 + row * XS0);
    float y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const f32x4 wv = *reinterpret_cast<const f32x4 *>(l1w + j * 4);
      y[j] = __builtin_fmaf(x[2], wv[2], __builtin_fmaf(x[1], wv[1], __builtin_fmaf(x[0], wv[0], wv[3])));
    }
    u32x2 h0, m0, l0, h1, m1, l1;
    split4<true>(y[0], y[1], y[2], y[3], h0, m0, l0);
    split4<true>(y[4], y[5], y[6], y[7], h1, m1, l1);
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    __bf16 *dst = bufY1 + (c & 1) * 3 * P1 + row * sp_stride(C1) + (tid & 7) * 8;
    *reinterpret_cast<u32x4 *>(dst) = u32x4{h0[0], h0[1], h1[0], h1[1]};
    *reinterpret_cast<u32x4 *>(dst + P1) = u32x4{m0[0], m0[1], m1[0], m1[1]};
    *reinterpret_cast<u32x4 *>(dst + 2 * P1) = u32x4{l0[0], l0[1], l1[0], l1[1]};
  };
  // ---- layer 2 of chunk c (group B: T orientation, rows = channels of n-tile wn2): Y1[c & 1] -> Y2[c & 1]
  auto layer2 = [&](int c, const bf16x8 (&w2h)[L2::KB], const bf16x8 (&w2m)[L2::KB], const bf16x8 (&w2l)[L2::KB]) {
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 bv = *reinterpret_cast<const f32x4 *>(bufB2 + wn2 * 32 + 8 * g + 4 * (lane >> 5));
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * g + j] = bv[j];
    }
    const __bf16 *xr = bufY1 + (c & 1) * 3 * P1 + (lane & 31) * sp_stride(C1) + 8 * (lane >> 5);
    bf16x8 a[L2::KB][3];
#pragma unroll
    for (int kb = 0; kb < L2::KB; ++kb)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[kb][p] = *reinterpret_cast<const bf16x8 *>(xr + p * P1 + kb * 16);
#pragma unroll
    for (int kb = 0; kb < L2::KB; ++kb) {       // mfma6<true>'s order: (x2,w0) (x1,w1) (x0,w2) (x1,w0) (x0,w1) (x0,w0)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2h[kb], a[kb][2], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2m[kb], a[kb][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2l[kb], a[kb][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2h[kb], a[kb][1], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2m[kb], a[kb][0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2h[kb], a[kb][0], acc, 0, 0, 0);
    }
    GN_PROBE_SUB(11);                   // MFMAs of the unit (incl. the wait for its operands)
    StoreSplit e2{bufY2 + (c & 1) * 3 * P2, sp_stride(C2), P2, lane};
    e2(0, wn2, acc);
    GN_PROBE_SUB(12);                   // relu + three-way split + stores
  };
  // ---- layer 3 + masked pooling of chunk c, this wave's n-tile (N orientation), from Y2[c & 1]
  auto layer3 = [&](int c) {
    const int rows = min(TP, R - c * TP);
    const float *gAc = bufGA + (c % RING) * TP * 4 + (4 * (lane >> 5)) * 4;
    const float *gBc = bufGB + (c % RING) * TP * 4 + (4 * (lane >> 5)) * 4;
    const int *sch = bufS + (c % RING) * TP;
    const int p0 = c * TP;
    const int wa = __builtin_amdgcn_readfirstlane(sch[0]);
    const int wz = __builtin_amdgcn_readfirstlane(sch[rows - 1]);
    const int ta = wa & 0xff, tz = wz & 0xff;
    const bool tend = (wz & 0x10000) != 0;
    auto finish_seed = [&](int q) {
      pool.finish(out + (size_t)(s0 + q) * ND * C3, C3, wave);
      pool.reset();
    };
    auto pool_lean = [&](const f32x16 &src, const float *gb) {      // gates read a slice ahead
      f32x4 g[2][2];
      auto gread = [&](int j) {
        const int r = 2 * j, rb = (r & 3) + 8 * (r >> 2);
        g[j & 1][0] = *reinterpret_cast<const f32x4 *>(gb + rb * 4);
        g[j & 1][1] = *reinterpret_cast<const f32x4 *>(gb + (rb + 1) * 4);
      };
      gread(0);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j + 1 < 8) gread(j + 1);
#pragma unroll
        for (int d = 0; d < ND; ++d)
          pool.hold[d][0] = fmaxf(pool.hold[d][0], fmaxf(src[2 * j] + g[j & 1][0][d], src[2 * j + 1] + g[j & 1][1][d]));
      }
    };
    const __bf16 *xr = bufY2 + (c & 1) * 3 * P2 + (lane & 31) * sp_stride(C2) + 8 * (lane >> 5);
    f32x16 acc[1];
    GN_PROBE_SUB(15);
    split_n_unit<0, C2, C3, ND, NWV>(w3, b3, xr, P2, acc, acc, pool, gAc, pool, gAc);
    GN_PROBE_SUB(13);                   // the tile's 48 MFMAs
    if (tz <= ta + 1) {
      pool_lean(acc[0], gAc);
      if (tz == ta) {
        if (tend) finish_seed(ta);
      } else {                          // the first seed ends inside the tile, the second gets its own pass
        finish_seed(ta);
        pool_lean(acc[0], gBc);
        if (tend) finish_seed(tz);
      }
    } else {                            // three or more seeds in one tile: one pass per seed, gates from the slot words
      for (int q = ta; q <= tz; ++q) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int rw = acc_row(2 * j, lane);
          const int s0w = sch[rw], s1w = sch[rw + 1];
#pragma unroll
          for (int d = 0; d < ND; ++d) {
            const float g0 = ((s0w & 0xff) == q && s0w >= 0 && ((s0w >> (8 + d)) & 1)) ? 0.f : -INFINITY;
            const float g1 = ((s1w & 0xff) == q && s1w >= 0 && ((s1w >> (8 + d)) & 1)) ? 0.f : -INFINITY;
            pool.hold[d][0] = fmaxf(pool.hold[d][0], fmaxf(acc[0][2 * j] + g0, acc[0][2 * j + 1] + g1));
          }
        }
        if (q < tz || s_off[q + 1] <= p0 + 32) finish_seed(q);
      }
    }
  };

  // The two groups run SEPARATE loops with the same number of workgroup barriers (s_barrier counts arrivals, not program
  // counters): written as one loop with group branches inside, the register allocator kept group B's streamed layer-2
  // fragments alive across group A's layer 3 (80 spilled registers).
  if (groupA) {
    int ent_n, sl_n = 0;        // entry / seed slot of this thread's row in the NEXT chunk to stage
    {
      int sl0;
      const int e0 = load_e(0, sl0);
      Raw w0{};
      load_raw(e0, sl0, w0);
      put(0, e0, sl0, w0);
      ent_n = load_e(1, sl_n);
    }
    for (int c = -2; c < nchunks; ++c) {
      Raw wr{};
      int ent_nn = -1, sl_nn = 0;
      if (c + 3 < nchunks) {
        load_raw(ent_n, sl_n, wr);          // row of chunk c+3 (its entry was requested a slot ago)
        ent_nn = load_e(c + 4, sl_nn);
      }
      GN_PROBE(0);
      __syncthreads();      // the slot barrier: Y2[c&1] (layer 2 of c), Y1[(c+1)&1] (layer 1 of c+1), the rows of c+2 are published
      GN_PROBE(1);
      if (c >= 0) layer3(c);
      GN_PROBE(2);
      if (c + 2 < nchunks) l1_chunk(c + 2);
      GN_PROBE(4);
      if (c + 3 < nchunks) put(c + 3, ent_n, sl_n, wr);
      ent_n = ent_nn; sl_n = sl_nn;
      GN_PROBE(5);
      GN_PROBE_COUNT(9, 1);
    }
  } else {
    for (int c = -2; c < nchunks; ++c) {
      GN_PROBE(0);
      if (c + 1 >= 0 && c + 1 < nchunks) {
        // layer 2's fragments: requested in front of the slot barrier, consumed right behind it (loads, barrier and use under
        // ONE branch: with the barrier outside it the fragments spilled)
        bf16x8 w2h[L2::KB], w2m[L2::KB], w2l[L2::KB];
#pragma unroll
        for (int kb = 0; kb < L2::KB; ++kb) { w2h[kb] = *w2_at(kb, 0); w2m[kb] = *w2_at(kb, 1); w2l[kb] = *w2_at(kb, 2); }
        __syncthreads();
        GN_PROBE(1);
        layer2(c + 1, w2h, w2m, w2l);
      } else {
        __syncthreads();
        GN_PROBE(1);
      }
      GN_PROBE(2);
      if (c >= 0) layer3(c);
      GN_PROBE(6);
      GN_PROBE_COUNT(9, 1);
    }
  }
  GN_PROBE_FLUSH();
}

// a neighbour's and its centre's coordinates as loaded, before (p - c) / radius
struct RawRow { float p0, p1, p2, c0, c1, c2; };

// ---- set abstraction, SA1 (no input features): SharedMLP[3,64,64,128] + max over 64 neighbours
// (pointnet2_utils.py:344-355 + pointnet2_modules.py:251-257).  Layer 1 (K = 3) on the fp32 MFMA, layers 2-3 split.
template <int C1, int C2, int C3, int NS, int TP, int MINW>
__global__ __launch_bounds__(MLP_THREADS, MINW) void sa_xyz_split_kernel(const float *__restrict__ xyz, int n,
                                                                   const float *__restrict__ new_xyz,
                                                                   const int32_t *__restrict__ idx, int np, int ntiles,
                                                                   int tpb, float radius, SplitWeights W,
                                                                   float *__restrict__ out, int nb) {
  constexpr int GPT = TP / NS, XS0 = lds_stride(8);
  static_assert(GPT >= 1, "a tile holds whole groups");
  constexpr int P1 = TP * sp_stride(C1), P2 = TP * sp_stride(C2);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16 *bufY1 = reinterpret_cast<__bf16 *>(smem_raw);     // 3 x [TP][C1+8]
  __bf16 *bufY2 = bufY1 + 3 * P1;                           // 3 x [TP][C2+8]
  float *bufX = reinterpret_cast<float *>(bufY2 + 3 * P2);  // 2 x [TP][12]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  using L1 = LayerW<8, C1>;
  using L2 = LayerWS<C1, C2>;
  using L3 = LayerWS<C2, C3>;
  L1 w1;
  L2 w2;
  L3 w3;
  w1.load(W.w1, W.b1, wave, lane);
  w2.load(W.w2, wave, lane);
  w3.load(W.w3, wave, lane);
  float b1t[16], b2t[16], b3[L3::NTW];
  load_bias_t(W.b1, wave % L1::WN, lane, b1t);
  load_bias_t(W.b2, wave % L2::WN, lane, b2t);
#pragma unroll
  for (int i = 0; i < L3::NTW; ++i) b3[i] = W.b3[((wave % L3::WN) * L3::NTW + i) * 32 + (lane & 31)];
  // XCD-aware workgroup order (nb > 0: nb clouds of gridDim.x / nb workgroups each): cloud = workgroup id % nb, so a cloud's
  // coordinates / per-point rows stay in one XCD's L2 instead of being fetched by all eight
  const int wgx = nb > 0 ? (int)(blockIdx.x % nb) * (int)(gridDim.x / nb) + (int)(blockIdx.x / nb) : (int)blockIdx.x;
  const int tile_end = min(ntiles, (wgx + 1) * tpb);
  int tile = wgx * tpb;
  if (tile >= tile_end) return;
  // the neighbour index of the tile after next is loaded while the next tile's coordinates are gathered (see the
  // hoisted kernel below)
  auto fetch_i = [&](int tl) -> int { return idx[(size_t)tl * GPT * NS + min(tid, TP - 1)]; };   // unconditional: see fetch_x
  // fetch_x only ISSUES the six loads of a row (neighbour and centre coordinates); the subtraction and the division by the
  // radius happen in put_x, a layer later.  Done at the fetch they made wave 0 -- the only wave whose rows are used -- wait
  // out the loads' L2 round trip in front of a barrier while the other three waves waited for it (phase probe of the SA2
  // kernel: 2 320 against ~700 cycles in that phase per 64-row tile; 0.1465 -> 0.1399 ms uninstrumented).
  // Every thread loads (threads beyond the tile's rows: row 0 and the last group's centre, never used): a branch around
  // the loads needs a merge value for the others, i.e. writes into registers that are still load destinations, and such
  // a write waits for ALL outstanding memory operations; and with 64 neighbours per group the centre is the same for a
  // whole wave -- a scalar load -- which inside the branch was waited for on the spot to be copied into vector registers.
  auto fetch_x = [&](int tl, int src, RawRow &r) {
    const int g0 = tl * GPT;
    const int b = g0 / np;
    const float *c = new_xyz + ((size_t)g0 + min(tid / NS, GPT - 1)) * 3;
    const float *px = xyz + (size_t)b * n * 3;
    r.p0 = px[src * 3 + 0]; r.p1 = px[src * 3 + 1]; r.p2 = px[src * 3 + 2];
    r.c0 = c[0]; r.c1 = c[1]; r.c2 = c[2];
  };
  auto put_x = [&](int buf, const RawRow &r) {
    if (tid < TP) {
      float *x = bufX + buf * TP * XS0 + tid * XS0;
      // pointnet2_utils.py:346-348
      *reinterpret_cast<f32x4 *>(x) = f32x4{(r.p0 - r.c0) / radius, (r.p1 - r.c1) / radius, (r.p2 - r.c2) / radius, 0.f};
      *reinterpret_cast<f32x4 *>(x + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int buf = 0;
  int sxn;
  RawRow nv;                 // rows of the next tile, fetched while the previous one was in layer 3
  {
    RawRow v;
    sxn = fetch_i(tile);
    fetch_x(tile, sxn, v);
    if (tile + 1 < tile_end) {
      sxn = fetch_i(tile + 1);
      fetch_x(tile + 1, sxn, nv);
      if (tile + 2 < tile_end) sxn = fetch_i(tile + 2);
    }
    put_x(0, v);
  }
  __syncthreads();
  GN_PROBE_INIT();
  for (; tile < tile_end; ++tile) {
    const int g0 = tile * GPT;
    const bool has_next = tile + 1 < tile_end;
    GN_PROBE(0);
    StoreSplit e1{bufY1, sp_stride(C1), P1, lane};
    run_layer_f32_t<TP, C1>(w1, b1t, bufX + buf * TP * XS0, wave, lane, e1);
    GN_PROBE(1);
    __syncthreads();
    GN_PROBE(2);
    StoreSplit e2{bufY2, sp_stride(C2), P2, lane};
    run_layer_split_t_pipe<TP, C1, C2>(w2, b2t, bufY1, P1, wave, lane, e2);
    if (has_next) put_x(buf ^ 1, nv);   // published by the barrier below: no third barrier per tile (see the crop kernel)
    GN_PROBE(3);
    __syncthreads();
    // The rows of the tile AFTER NEXT are requested here, in front of layer 3 (they are put into LDS behind the next
    // tile's layer 2).  Their addresses depend on an index loaded a tile ago; the wait for it is a vmcnt(0) -- loads and
    // stores share the counter and the compiler cannot count across the loop edge -- so it also waits for everything
    // issued since: at the loop top that included the pooled rows stored a few instructions earlier; here the last
    // stores are two layers old.  (With the unconditional loads of fetch_x: 0.1576 -> 0.1544 ms at the SA1 shape.)
    if (tile + 2 < tile_end) {
      fetch_x(tile + 2, sxn, nv);
      if (tile + 3 < tile_end) sxn = fetch_i(tile + 3);
    }
    GN_PROBE(4);
    PoolGlobal<NS, L3::NTW> e3{out + (size_t)g0 * C3, C3, lane, {}};
    run_layer_split_n<TP, C2, C3>(w3, b3, bufY2, P2, wave, lane, e3);
    GN_PROBE(5);
    GN_PROBE_COUNT(9, 1);
    buf ^= 1;
  }
  GN_PROBE_FLUSH();
}

// ---- set abstraction with layer 1's per-point part hoisted (gn_point_linear), SA2-SA4: the gathered P rows are
// the fp32 accumulators' initial value (each lane fetches exactly the 4 x float4 it needs per M tile, no LDS staging),
// the K = 8 coordinate block runs on the fp32 MFMA, layers 2-3 split.  TP = 64 (three bf16 planes of 128 channels).
template <int C1, int C2, int C3, int NS, int NWV>
__global__ __launch_bounds__(NWV * 64) void sa_hoist_split_kernel(const float *__restrict__ xyz, int n,
                                                                     const float *__restrict__ pfeat,
                                                                     const float *__restrict__ new_xyz,
                                                                     const int32_t *__restrict__ idx, int np, int ntiles,
                                                                     int tpb, float radius, SplitWeights W,
                                                                     float *__restrict__ out, int nb) {
  constexpr int TP = 64, MT = TP / 32, GPT = TP / NS, XS0 = lds_stride(8);
  constexpr int P1 = TP * sp_stride(C1), P2 = TP * sp_stride(C2);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16 *bufY1 = reinterpret_cast<__bf16 *>(smem_raw);     // 3 x [TP][C1+8]
  __bf16 *bufY2 = bufY1 + 3 * P1;                           // 3 x [TP][C2+8]
  float *bufX = reinterpret_cast<float *>(bufY2 + 3 * P2);  // 2 x [TP][12]
  float *bufW1 = bufX + 2 * TP * XS0;                       // [C1][4]: layer 1's coordinate block w0 w1 w2 0
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  using L2 = LayerWS<C1, C2, NWV>;
  using L3 = LayerWS<C2, C3, NWV>;
  static_assert(NWV == 4 && C1 == 128 && L3::WM == 1, "layer 1: one 32-channel n-tile per wave over all M tiles");
  constexpr int MTW = MT;
  L2 w2;
  L3 w3;
  // 96 + 192 registers of resident split weights: all of layer 3 and five of layer 2's eight k-blocks live in AGPRs (252)
  w2.template load<true, 5>(W.w2, wave, lane);
  w3.template load<true>(W.w3, wave, lane);
  if (tid < C1) {      // out of the fp32 fragment image: W1x[c][k] = Wp[((c/32)*64 + c%32)*4 + k], k < 3
    const float *wp = W.w1 + (size_t)(((tid >> 5) * 64 + (tid & 31)) * 4);
    *reinterpret_cast<f32x4 *>(bufW1 + tid * 4) = f32x4{wp[0], wp[1], wp[2], 0.f};
  }
  float b2t[16], b3[L3::NTW];
  load_bias_t(W.b2, wave % L2::WN, lane, b2t);
#pragma unroll
  for (int i = 0; i < L3::NTW; ++i) b3[i] = W.b3[((wave % L3::WN) * L3::NTW + i) * 32 + (lane & 31)];
  // XCD-aware workgroup order (nb > 0: nb clouds of gridDim.x / nb workgroups each): cloud = workgroup id % nb, so a cloud's
  // coordinates / per-point rows stay in one XCD's L2 instead of being fetched by all eight
  const int wgx = nb > 0 ? (int)(blockIdx.x % nb) * (int)(gridDim.x / nb) + (int)(blockIdx.x / nb) : (int)blockIdx.x;
  const int tile_end = min(ntiles, (wgx + 1) * tpb);
  int tile = wgx * tpb;
  if (tile >= tile_end) return;

  // The neighbour indices are fetched one tile ahead of the rows they address (two tiles ahead of their use): a row
  // gather issued right behind its index load would stall the wave for a full L2 round trip in front of layer 2.
  // this lane's accumulator-initial rows of tile tl: position mt*32 + (lane&31), channels wn*32 + 8g + 4(lane>>5) ..+3
  auto fetch_i = [&](int tl, int (&ix)[MTW], int &sx) {
    const int32_t *ixb = idx + (size_t)tl * GPT * NS;
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi) ix[mi] = ixb[mi * 32 + (lane & 31)];
    sx = ixb[min(tid, TP - 1)];
  };
  auto fetch_p = [&](int tl, const int (&ix)[MTW], f32x4 (&pf)[MTW][4]) {
    const int b = (tl * GPT) / np;
    const float *pb = pfeat + (size_t)b * n * C1 + wave * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi) {
      const float *row = pb + (size_t)ix[mi] * C1;
#pragma unroll
      for (int g = 0; g < 4; ++g) pf[mi][g] = *reinterpret_cast<const f32x4 *>(row + 8 * g);
    }
  };
  // fetch_x only ISSUES the six loads of a row (neighbour and centre coordinates); the subtraction and the division by the
  // radius happen in put_x, a layer later.  Done at the fetch they made wave 0 -- the only wave whose rows are used -- wait
  // out the loads' L2 round trip in front of a barrier while the other three waves waited for it (phase probe of the SA2
  // kernel: 2 320 against ~700 cycles in that phase per 64-row tile; 0.1465 -> 0.1399 ms uninstrumented).
  // Every thread loads (threads beyond the tile's rows: row 0 and the last group's centre, never used): a branch around
  // the loads needs a merge value for the others, i.e. writes into registers that are still load destinations, and such
  // a write waits for ALL outstanding memory operations; and with 64 neighbours per group the centre is the same for a
  // whole wave -- a scalar load -- which inside the branch was waited for on the spot to be copied into vector registers.
  auto fetch_x = [&](int tl, int src, RawRow &r) {
    const int g0 = tl * GPT;
    const int b = g0 / np;
    const float *c = new_xyz + ((size_t)g0 + min(tid / NS, GPT - 1)) * 3;
    const float *px = xyz + (size_t)b * n * 3;
    r.p0 = px[src * 3 + 0]; r.p1 = px[src * 3 + 1]; r.p2 = px[src * 3 + 2];
    r.c0 = c[0]; r.c1 = c[1]; r.c2 = c[2];
  };
  auto put_x = [&](int buf, const RawRow &r) {
    if (tid < TP) {
      float *x = bufX + buf * TP * XS0 + tid * XS0;
      // pointnet2_utils.py:346-348
      *reinterpret_cast<f32x4 *>(x) = f32x4{(r.p0 - r.c0) / radius, (r.p1 - r.c1) / radius, (r.p2 - r.c2) / radius, 0.f};
      *reinterpret_cast<f32x4 *>(x + 4) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  // ---- layer 1 on the vector ALUs (K = 3 on top of the gathered per-point rows), cut into 8 slices per M tile so that the
  // NEXT tile's layer 1 rides on the k-blocks of this tile's layer 3: slices 0-3 the FMAs of channel quad g (weights from
  // LDS), slices 4-7 relu + split + store of quad g.  acc rows of a lane: channels wave*32 + 8g + 4*(lane>>5) + j.
  // (every LDS read of a slice is consumed by the NEXT slice: a wait inside a slice would stall the MFMAs behind it)
  f32x4 l1x, l1wv[4];
  float l1y[4][4];
  const float *l1w = bufW1 + (wave * 32 + 4 * (lane >> 5)) * 4;
  auto l1_slice = [&](int sl, int mi, const f32x4 (&pfr)[MTW][4], const float *xtile) {
    if (sl >= 1 && sl <= 4) {         // FMAs of quad sl-1 with the weights requested by the previous slice
      const int g = sl - 1;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        l1y[g][j] = __builtin_fmaf(l1x[2], l1wv[j][2], __builtin_fmaf(l1x[1], l1wv[j][1], __builtin_fmaf(l1x[0], l1wv[j][0], pfr[mi][g][j])));
    }
    if (sl == 0) l1x = *reinterpret_cast<const f32x4 *>(xtile + (mi * 32 + (lane & 31)) * XS0);
    if (sl < 4) {                     // weights of quad sl for the next slice
#pragma unroll
      for (int j = 0; j < 4; ++j) l1wv[j] = *reinterpret_cast<const f32x4 *>(l1w + (8 * sl + j) * 4);
    }
    if (sl >= 4) {                    // relu + split + store of quad sl-4
      const int g = sl - 4;
      u32x2 hi, mid, lo;
      split4<true>(l1y[g][0], l1y[g][1], l1y[g][2], l1y[g][3], hi, mid, lo);
      __bf16 *row = bufY1 + (mi * 32 + (lane & 31)) * sp_stride(C1) + wave * 32 + 4 * (lane >> 5) + 8 * g;
      *reinterpret_cast<u32x2 *>(row) = hi;
      *reinterpret_cast<u32x2 *>(row + P1) = mid;
      *reinterpret_cast<u32x2 *>(row + 2 * P1) = lo;
    }
  };

  int buf = 0;
  f32x4 pf[MTW][4];
  int ixn[MTW], sxn;
  {
    RawRow v;
    fetch_i(tile, ixn, sxn);
    fetch_p(tile, ixn, pf);
    fetch_x(tile, sxn, v);
    if (tile + 1 < tile_end) fetch_i(tile + 1, ixn, sxn);
    put_x(0, v);
  }
  __syncthreads();
  // layer 1 of the first tile on its own; then the rows of the second tile are requested
#pragma unroll
  for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
    for (int sl = 0; sl < 8; ++sl) l1_slice(sl, mi, pf, bufX);
  GN_PROBE_INIT();
  for (; tile < tile_end; ++tile) {
    const int g0 = tile * GPT;
    const bool has_next = tile + 1 < tile_end;
    GN_PROBE(0);
    RawRow nv;
    GN_PROBE(1);
    __syncthreads();                      // layer 1 of this tile (done under the previous tile's layer 3) is published
    // rows of the next tile (their indices arrived during the previous tile), then the indices after: requested behind the
    // barrier rather than at the loop top, where the vmcnt(0) in front of their address arithmetic also waited for the
    // write acknowledgement of the pooled rows stored a few instructions earlier (see the SA1 kernel)
    if (has_next) {
      fetch_p(tile + 1, ixn, pf);
      fetch_x(tile + 1, sxn, nv);
      if (tile + 2 < tile_end) fetch_i(tile + 2, ixn, sxn);
    }
    GN_PROBE(2);
    StoreSplit e2{bufY2, sp_stride(C2), P2, lane};
    run_layer_split_t_pipe<TP, C1, C2, NWV>(w2, b2t, bufY1, P1, wave, lane, e2);
    if (has_next) put_x(buf ^ 1, nv);   // published by the barrier below
    GN_PROBE(3);
    __syncthreads();
    GN_PROBE(4);
    PoolGlobal<NS, L3::NTW> e3{out + (size_t)g0 * C3, C3, lane, {}};
    if (has_next) {
      const float *xnext = bufX + (buf ^ 1) * TP * XS0;
      auto hook = [&](int mt, int kb) { l1_slice(kb, mt, pf, xnext); };
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) split_n_tile<C2, C3, PoolGlobal<NS, L3::NTW>, NWV>(w3, b3, bufY2, P2, wave, lane, e3, mt, hook);
    } else {
      run_layer_split_n<TP, C2, C3, PoolGlobal<NS, L3::NTW>, NWV>(w3, b3, bufY2, P2, wave, lane, e3);
    }
    GN_PROBE(5);
    GN_PROBE_COUNT(9, 1);
    GN_PROBE_COUNT(10, MT);
    buf ^= 1;
  }
  GN_PROBE_FLUSH();
}

}  // namespace gn

using namespace gn;

// wgs_per_cu: workgroups per CU the grid aims at.  SA1 (two workgroups resident per CU, small weights) is best at 4; the
// hoisted SA2-4 kernels (one workgroup per CU, 288 registers of resident weights = 295 KB loaded by every workgroup) at 1:
// measured in isolation at batch 8 (profiles/sa_wgs_sweep.sh) SA2 0.138 -> 0.128 ms, SA3 0.0525 -> 0.0394, SA4 0.0282 -> 0.0242
// going from 4 to 1 (SA1 0.152 -> 0.185 the other way); GN_SA_WGS_PER_CU overrides both
static int split_tiles_per_block(int ntiles, int wgs_per_cu) {
  static int cus = 0;
  if (!cus) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) cus = p.multiProcessorCount;
    if (cus <= 0) cus = 256;
  }
  static const int div_env = []() { const char *e = getenv("GN_SA_WGS_PER_CU"); return e ? atoi(e) : 0; }();
  int tpb = ntiles / (cus * (div_env > 0 ? div_env : wgs_per_cu));
  return tpb < 1 ? 1 : (tpb > 64 ? 64 : tpb);
}

extern "C" int gn_sa_group_mlp_pool_split(const float *xyz, const float *new_xyz, const int32_t *idx, int b, int n, int np,
                                          int ns, float radius, const int *mlp, const void *const *wpacked,
                                          const float *const *bias, float *out, gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n > 0 && np > 0 && ns > 0, "sa_group_mlp_pool_split: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(xyz && new_xyz && idx && mlp && wpacked && bias && out, "sa_group_mlp_pool_split: null pointer");
  GN_REQUIRE((long)n * 3 < 0x7fffffffL, "sa_group_mlp_pool_split: n too large");
  if (!(mlp[0] == 64 && mlp[1] == 64 && mlp[2] == 128 && ns == 64)) {
    set_error("sa_group_mlp_pool_split: unsupported configuration mlp=[%d,%d,%d] nsample=%d", mlp[0], mlp[1], mlp[2], ns);
    return GN_ERR_INVALID_ARG;
  }
  const SplitWeights W{static_cast<const float *>(wpacked[0]), bias[0], wpacked[1], wpacked[2], bias[1], bias[2]};
  static const int variant = []() { const char *e = getenv("GN_SA1_VARIANT"); return e ? atoi(e) : 1; }();
  int rc = GN_OK;
  auto launch = [&](auto k, int tp) {
    const int ntiles = (int)((long)b * np / (tp / ns));
    const size_t lds = 2 * (size_t)(3 * tp * sp_stride(64) + 3 * tp * sp_stride(64)) + sizeof(float) * (2 * tp * lds_stride(8));
    if ((rc = set_lds_attr(k, lds, "sa_group_mlp_pool_split"))) return;
    const int tpb = split_tiles_per_block(ntiles, 4);
    const int per_cloud = ntiles / b;                              // tiles of one cloud
    const int nbx = (b > 1 && per_cloud % tpb == 0) ? b : 0;       // every cloud a whole number of workgroups -> XCD-aware order
    hipLaunchKernelGGL(k, dim3((ntiles + tpb - 1) / tpb), dim3(MLP_THREADS), lds, as_stream(stream), xyz, n, new_xyz, idx,
                       np, ntiles, tpb, radius, W, out, nbx);
  };
  if (variant == 1) {          // one group per tile, two workgroups per CU: one's epilogues and barriers under the other's MFMAs
    launch(sa_xyz_split_kernel<64, 64, 128, 64, 64, 2>, 64);
  } else {
    GN_REQUIRE(np % 2 == 0, "sa_group_mlp_pool_split: npoint must be even");
    launch(sa_xyz_split_kernel<64, 64, 128, 64, 128, 1>, 128);
  }
  if (rc) return rc;
  GN_CHECK_LAUNCH("sa_group_mlp_pool_split");
  return GN_OK;
}

extern "C" int gn_sa_hoisted_mlp_pool_split(const float *xyz, const float *pfeat, const float *new_xyz, const int32_t *idx,
                                            int b, int n, int np, int ns, float radius, const int *mlp,
                                            const void *const *wpacked, const float *const *bias, float *out,
                                            gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n > 0 && np > 0 && ns > 0, "sa_hoisted_mlp_pool_split: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(xyz && pfeat && new_xyz && idx && mlp && wpacked && bias && out, "sa_hoisted_mlp_pool_split: null pointer");
  GN_REQUIRE((long)n * 128 < 0x7fffffffL, "sa_hoisted_mlp_pool_split: n too large");
  if (!(mlp[0] == 128 && mlp[1] == 128 && mlp[2] == 256 && (ns == 32 || ns == 16))) {
    set_error("sa_hoisted_mlp_pool_split: unsupported configuration mlp=[%d,%d,%d] nsample=%d", mlp[0], mlp[1], mlp[2], ns);
    return GN_ERR_INVALID_ARG;
  }
  constexpr int TP = 64;
  GN_REQUIRE(np % (TP / ns) == 0, "sa_hoisted_mlp_pool_split: npoint must be a multiple of %d", TP / ns);
  const SplitWeights W{static_cast<const float *>(wpacked[0]), bias[0], wpacked[1], wpacked[2], bias[1], bias[2]};
  const int ntiles = (int)((long)b * np / (TP / ns));
  constexpr size_t lds = 2 * (3 * TP * sp_stride(128) + 3 * TP * sp_stride(128)) + sizeof(float) * (2 * TP * lds_stride(8) + 128 * 4);
  const int tpb = split_tiles_per_block(ntiles, 1);
  int rc;
  auto launch = [&](auto k, int threads) {
    if ((rc = set_lds_attr(k, lds, "sa_hoisted_mlp_pool_split"))) return;
    const int per_cloud = ntiles / b;
    const int nbx = (b > 1 && per_cloud * b == ntiles && per_cloud % tpb == 0) ? b : 0;
    hipLaunchKernelGGL(k, dim3((ntiles + tpb - 1) / tpb), dim3(threads), lds, as_stream(stream), xyz, n, pfeat, new_xyz, idx,
                       np, ntiles, tpb, radius, W, out, nbx);
  };
  rc = GN_OK;
  // (an 8-wave flavour, two waves per SIMD with half the weights each, needs ~270 registers per lane and spills)
  if (ns == 32) launch(sa_hoist_split_kernel<128, 128, 256, 32, 4>, 256);
  else launch(sa_hoist_split_kernel<128, 128, 256, 16, 4>, 256);
  if (rc) return rc;
  GN_CHECK_LAUNCH("sa_hoisted_mlp_pool_split");
  return GN_OK;
}

extern "C" int gn_crop_unique_mlp_pool_split(const float *xyz, const float *seed_xyz, const float *rot, const int32_t *uidx,
                                             const int32_t *ucount, int b, int n, int m, int ndepth, int ns, const int *mlp,
                                             const void *const *wpacked, const float *const *bias, float *out,
                                             gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n > 0 && m > 0, "crop_unique_mlp_pool_split: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(xyz && seed_xyz && rot && uidx && ucount && mlp && wpacked && bias && out,
             "crop_unique_mlp_pool_split: null pointer");
  GN_REQUIRE((long)n * 3 < 0x7fffffffL, "crop_unique_mlp_pool_split: n too large");
  if (!(ndepth == 4 && ns == 64 && mlp[0] == 64 && mlp[1] == 128 && mlp[2] == 256)) {
    set_error("crop_unique_mlp_pool_split: unsupported configuration depth=%d nsample=%d mlp=[%d,%d,%d]", ndepth, ns,
              mlp[0], mlp[1], mlp[2]);
    return GN_ERR_INVALID_ARG;
  }
  const SplitWeights W{static_cast<const float *>(wpacked[0]), bias[0], wpacked[1], wpacked[2], bias[1], bias[2]};
  const int nseeds = b * m;
  static const int spb_env = []() { const char *e = getenv("GN_CROP_SPB"); return e ? atoi(e) : 0; }();
  int spb = nseeds / 512;    // seeds per workgroup (the 250 KB of resident weights are loaded once per workgroup):
  spb = spb < 1 ? 1 : (spb > 16 ? 16 : spb);   // two workgroups per CU in the grid at B=8 (16: 6380, 4: 6290 scenes/s; 32 seeds with a 32-slot table: 0.362 ms, no gain)
  if (spb_env > 0) spb = spb_env > kCropMaxSpb ? kCropMaxSpb : spb_env;   // the kernel's s_cnt[] holds kCropMaxSpb seeds
  constexpr int TP = 96;
  constexpr size_t lds = 2 * (3 * TP * sp_stride(128) + 3 * TP * sp_stride(64)) + sizeof(float) * (2 * TP * lds_stride(8) + 2 * TP * 4);
  static const int pipe = []() { const char *e = getenv("GN_SPLIT_PIPE"); return e ? atoi(e) : 7; }();
  int rc = GN_OK;
  auto launch = [&](auto k) {
    if ((rc = set_lds_attr(k, lds, "crop_unique_mlp_pool_split"))) return;
    hipLaunchKernelGGL(k, dim3((nseeds + spb - 1) / spb), dim3(MLP_THREADS), lds, as_stream(stream), xyz, n, seed_xyz, rot,
                       uidx, ucount, m, nseeds, spb, W, out);
  };
  const char *pe = getenv("GN_CROP_PACKED");      // read per call: tests A/B the two kernels in one process
  const int packed = pe ? atoi(pe) : 1;
  constexpr size_t lds_packed = 2 * (3 * TP * sp_stride(128) + 3 * TP * sp_stride(64)) +
                                sizeof(float) * (2 * (TP + 1) * lds_stride(8) + 2 * 3 * TP * 4 + 2 * TP + 64 * 4) + 2 * 3 * 64 * 8;
  if (packed) {
    static const int ride = []() { const char *e = getenv("GN_CROP_RIDE"); return e ? atoi(e) : 1; }();
    // every cloud a whole number of workgroups -> XCD-aware order (cloud = workgroup id % b)
    const int nbx = (m % spb == 0 && b > 1) ? b : 0;
    auto go = [&](auto k) {
      if ((rc = set_lds_attr(k, lds_packed, "crop_unique_mlp_pool_split"))) return;
      hipLaunchKernelGGL(k, dim3((nseeds + spb - 1) / spb), dim3(MLP_THREADS), lds_packed, as_stream(stream), xyz, n, seed_xyz,
                         rot, uidx, ucount, m, nseeds, spb, W, out, nbx);
    };
    // GN_CROP_WAVES=8: the 8-wave form -- two waves per SIMD, every wave ONE n-tile of layer 3 and one of layer 2 resident
    // (144 of its 256 registers), one wave's epilogues / pooling / staging under its SIMD partner's MFMAs
    const char *we = getenv("GN_CROP_WAVES");
    const int waves = we ? atoi(we) : 4;
    if (waves == 8) {
      constexpr int TP8 = 32, RING = 4;
      constexpr size_t lds8 = 2 * (2 * 3 * TP8 * sp_stride(128) + 2 * 3 * TP8 * sp_stride(64)) +
                              sizeof(float) * (2 * TP8 * lds_stride(8) + 2 * RING * TP8 * 4 + RING * TP8 + 64 * 4 + 128);
      auto k8 = crop8_split_kernel<64, 128, 256, 4, 256>;
      if ((rc = set_lds_attr(k8, lds8, "crop_unique_mlp_pool_split"))) return rc;
      hipLaunchKernelGGL(k8, dim3((nseeds + spb - 1) / spb), dim3(512), lds8, as_stream(stream), xyz, n, seed_xyz, rot, uidx,
                         ucount, m, nseeds, spb, W, out, nbx);
    }
    else if (ride) go(crop_packed_split_kernel<64, 128, 256, 4, 256, TP, true>);
    else go(crop_packed_split_kernel<64, 128, 256, 4, 256, TP, false>);
    if (rc) return rc;
  } else if (pipe == 0) launch(crop_unique_split_kernel<64, 128, 256, 4, 256, TP, 0>);
  else if (pipe == 1) launch(crop_unique_split_kernel<64, 128, 256, 4, 256, TP, 1>);
  else if (pipe == 2) launch(crop_unique_split_kernel<64, 128, 256, 4, 256, TP, 2>);
  else if (pipe == 3) launch(crop_unique_split_kernel<64, 128, 256, 4, 256, TP, 3>);
  else launch(crop_unique_split_kernel<64, 128, 256, 4, 256, TP, 7>);
  if (rc) return rc;
  GN_CHECK_LAUNCH("crop_unique_mlp_pool_split");
  return GN_OK;
}

GN_PROBE_READER(gn_phase_probe_read_split)
