// mlp_point_wide.hip -- the per-point chains of mlp_point_split.hip with MORE ROWS PER STREAMED WEIGHT BYTE.
//
// mlp_point_split.hip keeps a tile's activations in LDS as three bf16 planes (6 bytes per element), which caps a tile of
// 256 channels at 64 rows: every 64-row tile streams a layer's 393 KB of split weights from L2 through the CU's 64 B/clk
// vector-memory path -- 6 144 clk against 12 288 clk of MFMAs, half the matrix-core time on the critical memory path
// (round-4 VERDICT weak item 2).  Here the tile is kept as FP32 (4 bytes per element: 128 rows x 256 channels = 130 KB)
// and a lane splits its 8-value operand fragment into the three bf16 terms when it LOADS it (mlp_split.h split4: the same
// round-to-nearest cuts the storing epilogue of the plane kernels makes, so every product, every accumulation order and
// therefore every output bit is the same).  A 128-row tile streams the weights once per 128 rows: 6 144 clk of weight
// stream against 24 576 clk of MFMAs.  The price is the split in the MFMA loop -- 52 VALU instructions per fragment, 12
// MFMAs per fragment with two n-tiles per wave: 4.3 per MFMA gap -- laid out so that the fragment of unit u+1 (an M tile
// x one weight chunk) is split under the MFMAs of unit u.  The storing epilogue shrinks to bias + ReLU + one 16-byte LDS
// store per register quad.
//
// Accumulators: 4 M tiles x 2 n-tiles x (hi, lo) = 256 registers per lane.  This file is compiled WITHOUT
// -amdgpu-mfma-vgpr-form (csrc/Makefile): the accumulators live in AGPRs, the streamed weight chunks and the fragments in
// VGPRs; the epilogue's v_accvgpr_read per element is ~1 % of a layer here.
//
// Reference math: models/modules.py:171-175,216-220 (OperationNet / ToleranceNet conv chains).
#include "mlp_point_stream.h"

namespace gn {

__host__ __device__ constexpr int wide_stride(int c) { return c + 4; }   // floats per LDS row: (c+4)/4 odd -> conflict-free b128

// rows of a plain channels-last tensor -> the fp32 tile; rows beyond nrows are zero
template <int C0, int TP, int XSF>
__device__ __forceinline__ void load_plain_f32(float *__restrict__ ys, const float *__restrict__ x, int row0, int nrows, int tid) {
  constexpr int Q = C0 / 4, PER = TP * Q / MLP_THREADS, BATCH = PER % 16 == 0 ? 16 : (PER % 8 == 0 ? 8 : (PER % 4 == 0 ? 4 : 1));
  static_assert(TP * Q % MLP_THREADS == 0, "tile must divide over the threads");
  using gf4 = const __attribute__((address_space(1))) f32x4;
  const __attribute__((address_space(1))) float *gx = (const __attribute__((address_space(1))) float *)x;
#pragma unroll 1
  for (int e0 = 0; e0 < PER; e0 += BATCH) {
    f32x4 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int e = tid + (e0 + u) * MLP_THREADS;
      const int r = e / Q, c4 = e - r * Q;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row0 + r < nrows) v[u] = *(gf4 *)(gx + (size_t)(row0 + r) * C0 + c4 * 4);
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int e = tid + (e0 + u) * MLP_THREADS;
      const int r = e / Q, c4 = e - r * Q;
      *reinterpret_cast<f32x4 *>(ys + r * XSF + c4 * 4) = v[u];
    }
  }
}

// The three bf16 planes of one operand fragment (8 consecutive k of one position) cut from its fp32 values: exactly what
// mlp_split.h StoreSplit / mlp_point_split.hip store_split4 leave in the plane tiles (split2: round-to-nearest cuts).
// Written as 12 STAGES (4 value pairs x 3 levels) so that the caller can put a stage or two behind each MFMA of the unit
// in flight: sched_group_barrier pipelines left half of these regions as 12 MFMAs back to back followed by the whole
// split (the greedy group solver), so the order is fixed in the source and pinned with a sched_barrier per MFMA.
struct FragSplit {
  f32x2 x[4], r[4];
  unsigned hi[4], mid[4], lo[4];
  __device__ __forceinline__ void set(const f32x4 &lo4, const f32x4 &hi4) {
    x[0] = f32x2{lo4[0], lo4[1]}; x[1] = f32x2{lo4[2], lo4[3]};
    x[2] = f32x2{hi4[0], hi4[1]}; x[3] = f32x2{hi4[2], hi4[3]};
  }
  template <int ST>
  __device__ __forceinline__ void stage() {       // ST = level * 4 + pair: dependent levels of a pair lie 4 stages apart
    constexpr int p = ST & 3, level = ST >> 2;
    if constexpr (level == 0) {
      hi[p] = cvt_pk_bf16(x[p]);
    } else if constexpr (level == 1) {
      r[p] = x[p] - widen_bf16x2(hi[p]);
      mid[p] = cvt_pk_bf16(r[p]);
    } else {
      lo[p] = cvt_pk_bf16(r[p] - widen_bf16x2(mid[p]));
    }
  }
  __device__ __forceinline__ void get(bf16x8 (&out)[3]) const {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    out[0] = __builtin_bit_cast(bf16x8, u32x4{hi[0], hi[1], hi[2], hi[3]});
    out[1] = __builtin_bit_cast(bf16x8, u32x4{mid[0], mid[1], mid[2], mid[3]});
    out[2] = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], lo[2], lo[3]});
  }
};
__device__ __forceinline__ void split_frag(const f32x4 &lo4, const f32x4 &hi4, bf16x8 (&out)[3]) {
  FragSplit f;
  f.set(lo4, hi4);
  f.stage<0>(); f.stage<1>(); f.stage<2>(); f.stage<3>(); f.stage<4>(); f.stage<5>(); f.stage<6>(); f.stage<7>();
  f.stage<8>(); f.stage<9>(); f.stage<10>(); f.stage<11>();
  f.get(out);
}
// stages [S0, S1) of a split
template <int S0, int S1>
__device__ __forceinline__ void split_stages(FragSplit &f) {
  if constexpr (S0 < S1) {
    f.template stage<S0>();
    split_stages<S0 + 1, S1>(f);
  }
}

// the t-th of the six products of one 16-k block into the two accumulators of mlp_split.h mfma6_2acc (same order per
// accumulator: the five small products into `lo` first, x0*w0 into `hi` last), weights on the A side
template <int T>
__device__ __forceinline__ void mfma6_2acc_step(const bf16x8 (&x)[3], const bf16x8 (&w)[3], f32x16 &hi, f32x16 &lo) {
  constexpr int XI[5] = {2, 1, 0, 1, 0};
  constexpr int WI[5] = {0, 1, 2, 0, 1};
  if constexpr (T < 5) lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[WI[T]], x[XI[T]], lo, 0, 0, 0);
  else hi = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], hi, 0, 0, 0);
}

// one 16-k block of a unit: slot s = t * NTW + i holds MFMA (n-tile i, product t) and split stages [s * 12 / NS, (s+1) * 12 / NS)
template <int NTW, int S, int KBC>
__device__ __forceinline__ void unit_kblock_slots(const bf16x8 (&x)[3], const bf16x8 (&w)[NTW][KBC][3], int kb, f32x16 (&hi)[NTW],
                                                  f32x16 (&lo)[NTW], FragSplit &fs, bool do_split) {
  constexpr int NS = 6 * NTW;
  if constexpr (S < NS) {
    constexpr int t = S / NTW, i = S % NTW;
    mfma6_2acc_step<t>(x, w[i][kb], hi[i], lo[i]);
    if (do_split) split_stages<S * 12 / NS, (S + 1) * 12 / NS>(fs);
    __builtin_amdgcn_sched_barrier(0);
    unit_kblock_slots<NTW, S + 1, KBC>(x, w, kb, hi, lo, fs, do_split);
  }
}
template <int NTW, int KBC>
__device__ __forceinline__ void unit_kblock(const bf16x8 (&x)[3], const bf16x8 (&w)[NTW][KBC][3], int kb, f32x16 (&hi)[NTW],
                                            f32x16 (&lo)[NTW], FragSplit &fs, bool do_split) {
  unit_kblock_slots<NTW, 0, KBC>(x, w, kb, hi, lo, fs, do_split);
}

// One streamed layer, T orientation (A = weights, B = activations), activations read as fp32 and split at fragment load.
// A UNIT is (weight chunk cc, M tile mi): KBC fragments of this lane's position, NTW * KBC * 6 MFMAs.  The raw fp32 of unit
// u+1 is requested from LDS before unit u's MFMAs and split among them.
// GRP > 0 (first layer of a tile): the tile's input arrives in column groups of GRP weight chunks; stage(g) -- called
// between the last unit of group g-1 and the first of group g -- publishes group g in LDS (store + workgroup barrier) and
// requests group g+1, so only the first group's fetch is exposed.
template <int TP, int CINP, int COUTP, int XSF, int CHREG, bool BLOCK2, bool HAVE_FIRST, int GRP, class Retire, class Epi, class Pre,
          class Stage>
__device__ __forceinline__ void run_layer_stream_f32_t(const void *__restrict__ wq, const float *__restrict__ bias,
                                                       const float *__restrict__ xs, int wave, int lane,
                                                       StreamChunk<TP, CINP, COUTP, CHREG> &first, Retire &&retire, Epi &&epi,
                                                       Pre &&pre, Stage &&stage) {
  using C = StreamCfg<TP, CINP, COUTP, CHREG>;
  constexpr int WN = C::WN, NTW = C::NTW, MTW = C::MTW, KBC = C::KBC, NCH = C::NCH;
  static_assert(!BLOCK2 || (WN % 2 == 0 && NCH % 4 == 0), "halves must fall on wave and (even) chunk boundaries");
  static_assert(GRP == 0 || (!BLOCK2 && NCH % GRP == 0), "column groups are whole chunks of a dense layer");
  const int wm = wave / WN, wn = wave % WN;
  asm volatile("" : "+s"(wq), "+s"(bias));   // keep this layer's own loads below the barrier that precedes it
  constexpr int NCC = BLOCK2 ? NCH / 2 : NCH, NUNITS = NCC * MTW;
  if (C::NACTIVE < 4 && wave >= C::NACTIVE) {
    if constexpr (GRP > 0) {
#pragma unroll
      for (int g = 1; g < NCC / GRP; ++g) stage(g);      // the idle waves keep the barrier count and store their share
    }
    pre();
    retire();
    return;
  }
  f32x16 acc[MTW][NTW], accl[MTW][NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        acc[mi][i][r] = 0.f;
        accl[mi][i][r] = 0.f;
      }
  StreamChunk<TP, CINP, COUTP, CHREG> other;
  const int c0 = BLOCK2 ? (wn >= WN / 2 ? NCH / 2 : 0) : 0;
  if constexpr (!HAVE_FIRST) first.load(wq, c0, wave, lane);
  // this lane's fragment base: position mt*32 + (lane&31), k = 8*(lane>>5) within a 16-k block
  const float *xbase = xs + (wm * MTW * 32 + (lane & 31)) * XSF + 8 * (lane >> 5);
  auto raw_read = [&](int u, f32x4 (&raw)[KBC][2]) {
    const int cc = u / MTW, mi = u % MTW;
    const float *xr = xbase + mi * 32 * XSF + (c0 + cc) * KBC * 16;
#pragma unroll
    for (int kb = 0; kb < KBC; ++kb) {
      raw[kb][0] = *reinterpret_cast<const f32x4 *>(xr + kb * 16);
      raw[kb][1] = *reinterpret_cast<const f32x4 *>(xr + kb * 16 + 4);
    }
  };
  f32x4 raw[KBC][2];
  bf16x8 cur[KBC][3], nxt[KBC][3];
  raw_read(0, raw);
#pragma unroll
  for (int kb = 0; kb < KBC; ++kb) split_frag(raw[kb][0], raw[kb][1], cur[kb]);
#pragma unroll
  for (int u = 0; u < NUNITS; ++u) {
    const int cc = u / MTW, mi = u % MTW;   // compile time (full unroll)
    // the next unit opens a new column group: its columns are published by stage() behind this unit
    const bool group_edge = GRP > 0 && mi == MTW - 1 && (cc + 1) % (GRP > 0 ? GRP : 1) == 0 && cc + 1 < NCC;
    const bool ahead = u + 1 < NUNITS && !group_edge;
    if (mi == 0) {
      if (cc + 1 < NCC) {
        if (cc & 1) first.load(wq, c0 + cc + 1, wave, lane); else other.load(wq, c0 + cc + 1, wave, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (ahead) raw_read(u + 1, raw);
    __builtin_amdgcn_sched_barrier(0);
    // per 16-k block: NTW * 6 MFMAs (t-major: consecutive MFMAs go to different accumulators), behind each of them its share
    // of the 12 split stages of ONE fragment of the next unit; a sched_barrier per MFMA pins that order
#pragma unroll
    for (int kb = 0; kb < KBC; ++kb) {
      FragSplit fs;
      if (ahead) fs.set(raw[kb][0], raw[kb][1]);
      unit_kblock<NTW>(cur[kb], (cc & 1) ? other.v : first.v, kb, acc[mi], accl[mi], fs, ahead);
      if (ahead) fs.get(nxt[kb]);
    }
    if (ahead) {
#pragma unroll
      for (int kb = 0; kb < KBC; ++kb)
#pragma unroll
        for (int p = 0; p < 3; ++p) cur[kb][p] = nxt[kb][p];
    }
    if constexpr (GRP > 0) {
      if (group_edge) {
        stage((cc + 1) / GRP);
        raw_read(u + 1, raw);
#pragma unroll
        for (int kb = 0; kb < KBC; ++kb) split_frag(raw[kb][0], raw[kb][1], cur[kb]);
      }
    }
  }
  f32x4 bt[NTW][4];
#pragma unroll
  for (int i = 0; i < NTW; ++i) load_bias_t4(bias, wn * NTW + i, lane, bt[i]);
  pre();
  // the sums are register work: they go in front of the barrier that retires the input, only the stores behind it
#pragma unroll
  for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][i][r] = (acc[mi][i][r] + accl[mi][i][r]) + bt[i][r >> 2][r & 3];
  retire();
#pragma unroll
  for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
    for (int i = 0; i < NTW; ++i) epi(wm * MTW + mi, wn * NTW + i, acc[mi][i]);
}

// [ three_interpolate(known, idx, w) (CK) | skip (CS) ] as fp32 rows,  w = (1/(sqrt(d2)+1e-8)) / sum
// (pointnet2_utils.py:138-140, pointnet2_modules.py:395-399; interpolate_gpu.cu:103-104 sum order) -- the arithmetic of
// mlp_point_split.hip load_fp_split, stored unsplit
template <int CK, int CS, int TP, int XSF>
__device__ __forceinline__ void load_fp_f32(float *__restrict__ ys, const PointSplitArgs &a, int row0, int tid) {
  constexpr int QK = CK / 4, QS = CS / 4, PERK = TP * QK / MLP_THREADS, PERS = TP * QS / MLP_THREADS, BATCH = 4;
  static_assert(TP * QK % MLP_THREADS == 0 && TP * QS % MLP_THREADS == 0 && PERK % BATCH == 0 && PERS % BATCH == 0,
                "tile must divide over the threads in batches of 4");
#pragma unroll 1
  for (int e0 = 0; e0 < PERK; e0 += BATCH) {     // all gathers of a batch in flight before the first blend
    f32x4 f0[BATCH], f1[BATCH], f2[BATCH];
    float w0[BATCH], w1[BATCH], w2[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int e = tid + (e0 + u) * MLP_THREADS;
      const int r = e / QK, c4 = e - r * QK;
      const int row = row0 + r;
      f0[u] = f1[u] = f2[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      w0[u] = w1[u] = w2[u] = 0.f;
      if (row < a.nrows) {
        const int b = row / a.n;
        const float *d2 = a.dist2 + (size_t)row * 3;
        const int32_t *ix = a.idx3 + (size_t)row * 3;
        const float r0 = 1.0f / (sqrtf(d2[0]) + 1e-8f);
        const float r1 = 1.0f / (sqrtf(d2[1]) + 1e-8f);
        const float r2 = 1.0f / (sqrtf(d2[2]) + 1e-8f);
        const float norm = r0 + r1 + r2;
        w0[u] = r0 / norm; w1[u] = r1 / norm; w2[u] = r2 / norm;
        const float *kb = a.known + (size_t)b * a.m * CK + c4 * 4;
        f0[u] = *reinterpret_cast<const f32x4 *>(kb + (size_t)ix[0] * CK);
        f1[u] = *reinterpret_cast<const f32x4 *>(kb + (size_t)ix[1] * CK);
        f2[u] = *reinterpret_cast<const f32x4 *>(kb + (size_t)ix[2] * CK);
      }
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int e = tid + (e0 + u) * MLP_THREADS;
      const int r = e / QK, c4 = e - r * QK;
      *reinterpret_cast<f32x4 *>(ys + r * XSF + c4 * 4) = f0[u] * w0[u] + f1[u] * w1[u] + f2[u] * w2[u];
    }
  }
#pragma unroll 1
  for (int e0 = 0; e0 < PERS; e0 += BATCH) {
    f32x4 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int e = tid + (e0 + u) * MLP_THREADS;
      const int r = e / QS, c4 = e - r * QS;
      const int row = row0 + r;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row < a.nrows) v[u] = *reinterpret_cast<const f32x4 *>(a.skip + (size_t)row * CS + c4 * 4);
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int e = tid + (e0 + u) * MLP_THREADS;
      const int r = e / QS, c4 = e - r * QS;
      *reinterpret_cast<f32x4 *>(ys + r * XSF + CK + c4 * 4) = v[u];
    }
  }
}

// one column group (GC columns of every row of the tile) of a plain channels-last tensor in flight in registers
template <int C0, int TP, int GC>
struct GroupStage {
  static constexpr int Q = GC / 4, PER = TP * Q / MLP_THREADS;
  static_assert(TP * Q % MLP_THREADS == 0 && GC % 4 == 0 && C0 % GC == 0, "a column group must divide over the threads");
  f32x4 v[PER];
  __device__ __forceinline__ void request(const float *__restrict__ x, int row0, int nrows, int g, int tid) {
    using gf4 = const __attribute__((address_space(1))) f32x4;
    const __attribute__((address_space(1))) float *gx = (const __attribute__((address_space(1))) float *)x;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * MLP_THREADS;
      const int r = e / Q, c4 = e - r * Q;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row0 + r < nrows) v[u] = *(gf4 *)(gx + (size_t)(row0 + r) * C0 + g * GC + c4 * 4);
    }
  }
  template <int XSF>
  __device__ __forceinline__ void publish(float *__restrict__ ys, int g, int tid) const {
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + u * MLP_THREADS;
      const int r = e / Q, c4 = e - r * Q;
      *reinterpret_cast<f32x4 *>(ys + r * XSF + g * GC + c4 * 4) = v[u];
    }
  }
};

// a workgroup barrier that orders LDS traffic only: __syncthreads() also drains every outstanding global request
// (s_waitcnt vmcnt(0)) -- here the next weight chunk and the next input group, requested on purpose ahead of the barrier
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

enum { WLOAD_PLAIN = 0, WLOAD_FP = 1 };

// NL layers C0 -> C1 (relu) [-> C2 (relu)] -> last (to global, RELU_OUT); widths as in point_mlp_split_kernel.  CHREG:
// registers one streamed weight chunk may take per wave (96; 48 where a wave holds three n-tiles of a 384-wide layer)
template <int LOADER, int NL, int C0, int C1P, int C1L, int C2P, int C2L, int C3P, int TP, bool RELU_OUT, bool BD2, int CHREG>
__global__ __launch_bounds__(MLP_THREADS, 1) void point_mlp_wide_kernel(PointSplitArgs a, int ntiles) {
  static_assert(NL == 2 || NL == 3, "two or three layers");
  constexpr int CMAX = cmax(C0, cmax(C1L, NL > 2 ? C2L : 0));
  constexpr int XSF = wide_stride(CMAX);
  // the plain loader brings the tile in column groups of two layer-1 weight chunks, group g+1 under the MFMAs of group g
  using C1 = StreamCfg<TP, C0, C1P, CHREG>;
#ifndef GN_WIDE_GRP128
#define GN_WIDE_GRP128 1       // chunks per column group on 128-row tiles (16 staging registers per chunk and thread)
#endif
  constexpr int GRP = (LOADER == WLOAD_PLAIN && C1::NCH >= 4) ? (TP >= 128 ? GN_WIDE_GRP128 : 2) : 0;
  constexpr int GC = (GRP > 0 ? GRP : 1) * C1::KBC * 16;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *buf = reinterpret_cast<float *>(smem_raw);    // [TP][CMAX+4] fp32
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto barrier = []() { __syncthreads(); };
  auto none = []() {};
  auto nostage = [](int) {};
  GN_PROBE_INIT();
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    GN_PROBE(7);
    const int row0 = tile * TP;
    const int nrows = min(TP, a.nrows - row0);
    StreamChunk<TP, C0, C1P, CHREG> ch1;
    stream_prefetch<TP, C0, C1P, CHREG, false>(ch1, a.w[0], wave, lane);      // lands under the row fetch
    GroupStage<C0, TP, GC> gs;
    if constexpr (LOADER == WLOAD_FP) {
      load_fp_f32<C0 / 2, C0 / 2, TP, XSF>(buf, a, row0, tid);
    } else if constexpr (GRP > 0) {
      gs.request(a.x, row0, a.nrows, 0, tid);
      gs.template publish<XSF>(buf, 0, tid);
      gs.request(a.x, row0, a.nrows, 1, tid);
    } else {
      load_plain_f32<C0, TP, XSF>(buf, a.x, row0, a.nrows, tid);
    }
    GN_PROBE(0);
    __syncthreads();
    GN_PROBE(1);
    auto stage1 = [&](int g) {       // group g was requested a group ago: publish it, request the one after it
      gs.template publish<XSF>(buf, g, tid);
      if ((g + 1) * GC < C0) gs.request(a.x, row0, a.nrows, g + 1, tid);
      lds_barrier();
    };
    auto to_lds = [&](int ncols) {
      return [&, ncols](int mt, int nt, const f32x16 &acc) {
        if (nt * 32 < ncols) {
          float *row = buf + (mt * 32 + (lane & 31)) * XSF + nt * 32 + 4 * (lane >> 5);
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4 *>(row + 8 * g) = f32x4{relu_raw(acc[4 * g]), relu_raw(acc[4 * g + 1]), relu_raw(acc[4 * g + 2]),
                                                             relu_raw(acc[4 * g + 3])};
        }
      };
    };
    const bool vec_out = (a.ldo & 3) == 0;       // rows of the output are 16-byte aligned
    auto to_global = [&](int mt, int nt, const f32x16 &acc) {
      const int row = mt * 32 + (lane & 31);
      if (row < nrows) {
        float *o = a.out + (size_t)(row0 + row) * a.ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int ch = nt * 32 + 8 * g + 4 * (lane >> 5);
          f32x4 v{acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
          if (RELU_OUT) v = f32x4{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
          if (vec_out && ch + 3 < a.ncols_out) {
            *reinterpret_cast<f32x4 *>(o + ch) = v;
          } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (ch + j < a.ncols_out) o[ch + j] = v[j];
          }
        }
      }
    };
    if constexpr (NL == 2) {
      StreamChunk<TP, C1L, C2P, CHREG> ch2;
      run_layer_stream_f32_t<TP, C0, C1P, XSF, CHREG, false, true, GRP>(
          a.w[0], a.b[0], buf, wave, lane, ch1, barrier, to_lds(C1L),
          [&]() { GN_PROBE_SUB(10); stream_prefetch<TP, C1L, C2P, CHREG, BD2>(ch2, a.w[1], wave, lane); }, stage1);
      GN_PROBE(2);
      __syncthreads();
      GN_PROBE(3);
      run_layer_stream_f32_t<TP, C1L, C2P, XSF, CHREG, BD2, true, 0>(a.w[1], a.b[1], buf, wave, lane, ch2, none, to_global,
                                                                     [&]() { GN_PROBE_SUB(11); }, nostage);
      GN_PROBE(4);
    } else {
      // a 64-wide last layer on a tile of four M tiles: one wave per M tile, both n-tiles (a split fragment feeds 12 MFMAs)
      constexpr int CH3 = (C3P == 64 && TP == 128) ? 1000 + 48 : CHREG;
      StreamChunk<TP, C1L, C2P, CHREG> ch2;
      StreamChunk<TP, C2L, C3P, CH3> ch3;
      run_layer_stream_f32_t<TP, C0, C1P, XSF, CHREG, false, true, GRP>(
          a.w[0], a.b[0], buf, wave, lane, ch1, barrier, to_lds(C1L),
          [&]() { GN_PROBE_SUB(10); stream_prefetch<TP, C1L, C2P, CHREG, BD2>(ch2, a.w[1], wave, lane); }, stage1);
      GN_PROBE(2);
      __syncthreads();
      GN_PROBE(3);
      run_layer_stream_f32_t<TP, C1L, C2P, XSF, CHREG, BD2, true, 0>(
          a.w[1], a.b[1], buf, wave, lane, ch2, barrier, to_lds(C2L),
          [&]() { GN_PROBE_SUB(11); stream_prefetch<TP, C2L, C3P, CH3, false>(ch3, a.w[2], wave, lane); }, nostage);
      GN_PROBE(4);
      __syncthreads();
      GN_PROBE(5);
      run_layer_stream_f32_t<TP, C2L, C3P, XSF, CH3, false, true, 0>(a.w[2], a.b[2], buf, wave, lane, ch3, none, to_global,
                                                                     [&]() { GN_PROBE_SUB(12); }, nostage);
      GN_PROBE(6);
    }
    __syncthreads();   // the next tile's loader overwrites the buffer
    GN_PROBE_COUNT(9, 1);
  }
  GN_PROBE_FLUSH();
}

template <int LOADER, int NL, int C0, int C1P, int C1L, int C2P, int C2L, int C3P, int TP, bool RELU_OUT, bool BD2 = false, int CHREG = 96>
static int launch_point_wide(const PointSplitArgs &a, hipStream_t s, const char *name) {
  constexpr int CMAX = cmax(C0, cmax(C1L, NL > 2 ? C2L : 0));
  constexpr size_t lds = (size_t)4 * TP * wide_stride(CMAX);
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = point_mlp_wide_kernel<LOADER, NL, C0, C1P, C1L, C2P, C2L, C3P, TP, RELU_OUT, BD2, CHREG>;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
    set_error("%s: cannot reserve %zu bytes of LDS", name, lds);
    return GN_ERR_LAUNCH;
  }
  const int ntiles = (a.nrows + TP - 1) / TP;
  hipLaunchKernelGGL(k, dim3(ntiles), dim3(MLP_THREADS), lds, s, a, ntiles);
  GN_CHECK_LAUNCH(name);
  return GN_OK;
}

// the heads chain (OperationNet + ToleranceNet merged, block-diagonal layer 2) on 128-row tiles; called by
// gn_heads_mlp_split (mlp_point_split.hip) when the launch has enough rows to give every CU a tile
int heads_mlp_wide(const PointSplitArgs &a, hipStream_t s) {
  return launch_point_wide<WLOAD_PLAIN, 3, 256, 256, 256, 256, 256, 64, 128, false, true>(a, s, "heads_mlp_split");
}
// ApproachNet (256 -> 256 -> 384 -> 384, layer 3's K padded to 320) and the feature-propagation chains (512 -> 256 -> 256) on
// 64-row fp32 tiles: half the weight bytes per row of the 32-row plane tiles, at half as many workgroups -- the shapes for
// SEVERAL batches in flight (what a launch leaves of the chip other batches' kernels fill)
int approach_mlp_wide(const PointSplitArgs &a, hipStream_t s) {
  return launch_point_wide<WLOAD_PLAIN, 3, 256, 256, 256, 384, 320, 384, 64, false, false, 48>(a, s, "approach_mlp_split");
}
int fp_mlp_wide(const PointSplitArgs &a, hipStream_t s) {
  return launch_point_wide<WLOAD_FP, 2, 512, 256, 256, 256, 0, 0, 64, true>(a, s, "fp_interp_mlp_split");
}

}  // namespace gn

GN_PROBE_READER(gn_phase_probe_read_point_wide)
