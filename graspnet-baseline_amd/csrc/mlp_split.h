// mlp_split.h -- fp32 layers on the bf16 matrix cores by operand splitting ("bf16x3 operands, 6 products").
//
// Why.  v_mfma_f32_32x32x2_f32 runs on the SIMD's FP32 vector ALUs: 32 MAC/cycle/SIMD, and every other VALU
// instruction of the kernel competes with it (profiles/probe/mfma_valu_overlap.hip).  The bf16 form
// v_mfma_f32_32x32x16_bf16 runs on the matrix core at 512 MAC/cycle/SIMD with VALU work overlapped.  Each fp32
// operand is therefore split exactly into three bf16 terms  x = x0 + x1 + x2  (x0 = bf16(x), x1 = bf16(x - x0),
// x2 = bf16(x - x0 - x1): 3 x 8 significand bits), and
//     x * w  ~=  x0w0 + (x0w1 + x1w0) + (x0w2 + x1w1 + x2w0)        (6 bf16 MFMAs per 16 k, fp32 accumulate)
// The dropped terms x1w2 + x2w1 + x2w2 are <= 2^-23 |x||w|, the size of one fp32 rounding; every bf16 product is
// exact in fp32 and a 16-k block is accumulated inside the MFMA, so the result is at least as close to the
// float64 value as a k-ordered fp32 FMA chain (tests/test_split_gpu.py measures both against float64).
// 6 x 32 cycles per 16 k instead of 8 x 64: 2.7x the fp32 MFMA rate, and the epilogues hide under the MFMAs.
//
// Layouts.  Activations live in LDS as three bf16 planes, position-major: plane p, row pos, k contiguous, row
// stride C+8 elements ((C+8)/8 odd: conflict-free ds_read_b128).  A lane's operand fragment for k-block kb is the
// 8 consecutive k at kb*16 + 8*(lane>>5) of row/column (lane&31) -- the same for the A and the B operand, so one
// packed weight image serves both orientations:
//     Wq[nt][kb][plane][lane][j] = W_plane[nt*32 + (lane&31)][kb*16 + 8*(lane>>5) + j]
//   N orientation (A = activations, B = weights): D columns = channels, rows = positions -> pooling in-lane
//   T orientation (A = weights, B = activations): D columns = positions, rows = channels -> a lane holds 4
//     consecutive channels per register quad, i.e. one 8-byte store per plane into the next layer's tile
#pragma once
#include "mlp_core.h"

namespace gn {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;

__host__ __device__ constexpr int sp_stride(int c) { return c + 8; }   // bf16 elements per LDS row

template <int CINP, int COUTP, int NWV = 4>
struct LayerWS {
  static_assert(CINP % 16 == 0 && COUTP % 32 == 0, "pad CIN to 16 and COUT to 32");
  static constexpr int NT = COUTP / 32;
  static constexpr int WN = wn_for(COUTP, NWV);
  static constexpr int WM = NWV / WN;
  static_assert(NT % WN == 0, "N tiles must divide over the waves");
  static constexpr int NTW = NT / WN;
  static constexpr int KB = CINP / 16;
  bf16x8 b[NTW][KB][3];

  // PIN_AGPR: the resident fragments are only ever MFMA operands, and the MFMA reads A/B straight from AGPRs.  Left to
  // itself hipcc keeps what does not fit the 256 VGPRs as AGPR *spill* copies and brings each fragment back with four
  // v_accvgpr_read_b32 before every use (the CloudCrop kernel carried 1.3 of them per MFMA, on the VALU the epilogues
  // compete for); an empty asm with an "a" constraint makes the AGPR the value's home, and the accumulators stay in VGPRs
  // where the pooling / splitting epilogues read them without moves.  At most 256 AGPRs' worth may be pinned per wave.
  // PIN_KB: k-blocks (from 0) whose fragments are pinned; KB pins the whole layer, 0 none.
  template <bool PIN_AGPR = false, int PIN_KB = KB>
  __device__ __forceinline__ void load(const void *__restrict__ wp, int wave, int lane) {
    const int wn = wave % WN;
    const char *base = static_cast<const char *>(wp);
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          b[i][kb][p] = *reinterpret_cast<const bf16x8 *>(base + ((((size_t)((wn * NTW + i) * KB + kb)) * 3 + p) * 64 + lane) * 16);
    // pins only after every load has been issued: an asm operand must be in its register, i.e. each pin waits for its load
    if constexpr (PIN_AGPR) {
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int kb = 0; kb < (PIN_KB < KB ? PIN_KB : KB); ++kb)
#pragma unroll
          for (int p = 0; p < 3; ++p) asm volatile("" : "+a"(b[i][kb][p]));
    }
  }
};

// the six products of one 16-k block, small terms first; AW = true puts the weights on the A side (T orientation)
template <bool AW>
__device__ __forceinline__ f32x16 mfma6(const bf16x8 (&x)[3], const bf16x8 (&w)[3], f32x16 acc) {
  constexpr int XI[6] = {2, 1, 0, 1, 0, 0};
  constexpr int WI[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
  for (int t = 0; t < 6; ++t)
    acc = AW ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[WI[t]], x[XI[t]], acc, 0, 0, 0)
             : __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[XI[t]], w[WI[t]], acc, 0, 0, 0);
  return acc;
}

// The same six products into TWO accumulators: `hi` takes x0*w0 only -- one fp32 rounding per 16-k block -- and `lo` the
// five small products (magnitude <= 2^-8 of hi's, so its own roundings are 2^-8 smaller still); the caller adds lo to hi
// once, after the last k-block.  With one accumulator every one of the six MFMAs rounds at the accumulator's full
// magnitude: 6*K/16 roundings, which is what made the K = 256..512 per-point chains 1.3-2x less accurate than the
// torch-CPU oracle against float64 (profiles/parity_r02.json, stage_isolated_vs_f64); this form has K/16 + 1.
template <bool AW>
__device__ __forceinline__ void mfma6_2acc(const bf16x8 (&x)[3], const bf16x8 (&w)[3], f32x16 &hi, f32x16 &lo) {
  constexpr int XI[5] = {2, 1, 0, 1, 0};
  constexpr int WI[5] = {0, 1, 2, 0, 1};
#pragma unroll
  for (int t = 0; t < 5; ++t)
    lo = AW ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[WI[t]], x[XI[t]], lo, 0, 0, 0)
            : __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[XI[t]], w[WI[t]], lo, 0, 0, 0);
  hi = AW ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[0], x[0], hi, 0, 0, 0)
          : __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[0], w[0], hi, 0, 0, 0);
}

// exact three-way split of four fp32 values (ReLU first if asked) into three packed bf16x4 (two dwords each).
// Written pair-wise on purpose: one v_cvt_pk_bf16_f32 per pair and level, the bf16 pair widened back with a shift and a
// mask, the residual by one v_pk_add_f32 -- 13 VALU per pair instead of the ~27 hipcc emits for the scalar form (it
// converts every value twice and canonicalises before each fmaxf).  No NaNs can occur here.
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

// ReLU as v_med3_f32(x, 0, +inf): fmaxf() makes hipcc canonicalise its operand first (a second v_max per value),
// and inline asm must not touch MFMA results (the hazard recognizer does not see inside it: the two-workgroup SA1
// kernel keeps its accumulators in VGPRs and read them too early that way).
__device__ __forceinline__ float relu_raw(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_inff()); }
__device__ __forceinline__ unsigned cvt_pk_bf16(f32x2 v) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ f32x2 widen_bf16x2(unsigned u) {
  return f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
}
__device__ __forceinline__ void split2(f32x2 x, unsigned &hi, unsigned &mid, unsigned &lo) {
  hi = cvt_pk_bf16(x);
  const f32x2 r = x - widen_bf16x2(hi);
  mid = cvt_pk_bf16(r);
  const f32x2 r2 = r - widen_bf16x2(mid);
  lo = cvt_pk_bf16(r2);
}
template <bool RELU>
__device__ __forceinline__ void split4(float v0, float v1, float v2, float v3, u32x2 &hi, u32x2 &mid, u32x2 &lo) {
  if (RELU) { v0 = relu_raw(v0); v1 = relu_raw(v1); v2 = relu_raw(v2); v3 = relu_raw(v3); }
  unsigned h0, m0, l0, h1, m1, l1;
  split2(f32x2{v0, v1}, h0, m0, l0);
  split2(f32x2{v2, v3}, h1, m1, l1);
  hi = u32x2{h0, h1};
  mid = u32x2{m0, m1};
  lo = u32x2{l0, l1};
}

// T-orientation epilogue: ReLU, split, store the tile into the next layer's three planes
struct StoreSplit {
  __bf16 *ys;     // plane 0
  int stride;     // elements per row
  int plane;      // elements per plane
  int lane;
  __device__ __forceinline__ void operator()(int mt, int nt, const f32x16 &acc) {
    __bf16 *row = ys + (mt * 32 + (lane & 31)) * stride + nt * 32 + 4 * (lane >> 5);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      u32x2 hi, mid, lo;
      split4<true>(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3], hi, mid, lo);
      *reinterpret_cast<u32x2 *>(row + 8 * g) = hi;
      *reinterpret_cast<u32x2 *>(row + 8 * g + plane) = mid;
      *reinterpret_cast<u32x2 *>(row + 8 * g + 2 * plane) = lo;
    }
  }
};

// bias of the 16 accumulator rows of n-tile nt in T orientation (rows = channels)
__device__ __forceinline__ void load_bias_t(const float *__restrict__ bp, int nt, int lane, float (&bt)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) bt[r] = bp[nt * 32 + acc_row(r, lane)];
}

// ---- T orientation: Y[pos][cout] planes <- relu(W * X + b), X read from three bf16 planes ------------------
template <int TP, int CINP, int COUTP, int NWV = 4>
__device__ __forceinline__ void run_layer_split_t(const LayerWS<CINP, COUTP, NWV> &w, const float (&bias_t)[16],
                                                  const __bf16 *__restrict__ xs, int xplane, int wave, int lane,
                                                  StoreSplit &epi, int mt_count = TP / 32) {
  using LW = LayerWS<CINP, COUTP, NWV>;
  static_assert(LW::NTW == 1, "T-orientation layers keep one n-tile per wave (16 bias registers each)");
  constexpr int XS = sp_stride(CINP), KB = LW::KB;
  const int wm = wave / LW::WN, wn = wave % LW::WN;
  for (int mt = wm; mt < mt_count; mt += LW::WM) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias_t[r];
    const __bf16 *xr = xs + (mt * 32 + (lane & 31)) * XS + 8 * (lane >> 5);
    bf16x8 a[KB][3];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[kb][p] = *reinterpret_cast<const bf16x8 *>(xr + p * xplane + kb * 16);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) acc = mfma6<true>(a[kb], w.b[0][kb], acc);
    if constexpr (KB > 2) {   // [6 reads] ([6 MFMAs][3 reads])* : operands two k-blocks ahead of their MFMAs
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int kb = 0; kb < KB - 2; ++kb) {
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
    }
    epi(mt, wn, acc);
  }
}

// one 16-k block of a software-pipelined layer: 6 x (1 MFMA, NV VALU) scheduling groups (see the pipelined forms below)
template <int NV>
__device__ __forceinline__ void interleave_step(int nreads, int nwrites) {
  (void)nreads; (void)nwrites;
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
  }
}

struct NoHook2 { __device__ __forceinline__ void operator()(int, int) const {} };
// one M tile: acc[pos][cout] of all of this wave's n-tiles, handed to the epilogue; hook(mt, kb) runs once per k-block
// between the MFMAs (independent VALU / LDS work riding along)
template <int CINP, int COUTP, class Epi, int NWV = 4, class Hook = NoHook2>
__device__ __forceinline__ void split_n_tile(const LayerWS<CINP, COUTP, NWV> &w, const float (&bias)[LayerWS<CINP, COUTP, NWV>::NTW],
                                             const __bf16 *__restrict__ xs, int xplane, int wave, int lane, Epi &epi, int mt,
                                             Hook &&hook = Hook{}) {
  using LW = LayerWS<CINP, COUTP, NWV>;
  constexpr int XS = sp_stride(CINP), KB = LW::KB, NTW = LW::NTW;
  const int wn = wave % LW::WN;
  f32x16 acc[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = bias[i];
  const __bf16 *xr = xs + (mt * 32 + (lane & 31)) * XS + 8 * (lane >> 5);
  // operands two k-blocks ahead of their MFMAs; a sched_barrier per k-block keeps the window at three blocks
  // (hipcc otherwise hoists all KB*3 fragment reads: 96 registers at K = 128)
  bf16x8 a[KB][3];
#pragma unroll
  for (int kb = 0; kb < (KB < 2 ? KB : 2); ++kb)
#pragma unroll
    for (int p = 0; p < 3; ++p) a[kb][p] = *reinterpret_cast<const bf16x8 *>(xr + p * xplane + kb * 16);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 2 < KB) {
#pragma unroll
      for (int p = 0; p < 3; ++p) a[kb + 2][p] = *reinterpret_cast<const bf16x8 *>(xr + p * xplane + (kb + 2) * 16);
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) acc[i] = mfma6<false>(a[kb], w.b[i][kb], acc[i]);
    hook(mt, kb);
    if constexpr (!__is_same(__remove_cvref(Hook), NoHook2)) {
#pragma unroll
      for (int i = 0; i < NTW; ++i) interleave_step<3>(0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int i = 0; i < NTW; ++i) epi(mt, wn * NTW + i, i, acc[i]);
}

// ---- N orientation: acc[pos][cout] = X * W^T + b handed to a pooling epilogue (PoolMasked / PoolGlobal) -------
template <int TP, int CINP, int COUTP, class Epi, int NWV = 4>
__device__ __forceinline__ void run_layer_split_n(const LayerWS<CINP, COUTP, NWV> &w, const float (&bias)[LayerWS<CINP, COUTP, NWV>::NTW],
                                                  const __bf16 *__restrict__ xs, int xplane, int wave, int lane, Epi &epi,
                                                  int mt_count = TP / 32) {
  using LW = LayerWS<CINP, COUTP, NWV>;
  const int wm = wave / LW::WN;
  for (int mt = wm; mt < mt_count; mt += LW::WM) split_n_tile<CINP, COUTP, Epi, NWV>(w, bias, xs, xplane, wave, lane, epi, mt);
}

// ---- software-pipelined forms.  The bf16 MFMA runs on the matrix core and VALU / LDS instructions of the same wave
// issue in its shadow (profiles/probe/mfma_bf16_overlap.hip: <= 4 fillers per MFMA are nearly free), so the epilogue of
// output tile u is spread over the MFMAs of tile u+1.  The source is written step by step (one 16-k block = 6 MFMAs
// per step) with a sched_barrier closing each step, and sched_group_barriers asking for 1 MFMA : NV VALU inside it.

// T orientation, one n-tile per wave: split + store of this wave's M tile t under the MFMAs of its M tile t+1
// (a quarter of the tile, 4 channels x this lane's position, behind every KB/4-th k-block)
// VAR (profiles/probe/split_layer_bench.hip only): 1 = no LDS stores, 2 = no split arithmetic (stores of raw bits)
template <int TP, int CINP, int COUTP, int NWV = 4, int VAR = 0>
__device__ __forceinline__ void run_layer_split_t_pipe(const LayerWS<CINP, COUTP, NWV> &w, const float (&bias_t)[16],
                                                       const __bf16 *__restrict__ xs, int xplane, int wave, int lane,
                                                       StoreSplit &epi, int mt_count = TP / 32) {
  using LW = LayerWS<CINP, COUTP, NWV>;
  static_assert(LW::NTW == 1 && LW::KB % 4 == 0, "one n-tile per wave, K a multiple of 64");
  constexpr int XS = sp_stride(CINP), KB = LW::KB, WM = LW::WM, MAXU = (TP / 32 + WM - 1) / WM, QS = KB / 4;
  const int wm = wave / LW::WN, wn = wave % LW::WN;
  if (wm >= mt_count) return;
  f32x16 acc[2];
  auto mma = [&](f32x16 &dst, int mt, auto &&fill) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r] = bias_t[r];
    const __bf16 *xr = xs + (mt * 32 + (lane & 31)) * XS + 8 * (lane >> 5);
    bf16x8 a[KB][3];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[kb][p] = *reinterpret_cast<const bf16x8 *>(xr + p * xplane + kb * 16);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      if (kb + 2 < KB) {
#pragma unroll
        for (int p = 0; p < 3; ++p) a[kb + 2][p] = *reinterpret_cast<const bf16x8 *>(xr + p * xplane + (kb + 2) * 16);
      }
      dst = mfma6<true>(a[kb], w.b[0][kb], dst);
      if (kb % QS == 0) fill(kb / QS);
      __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
      interleave_step<6>(3, 3);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  auto store_quarter = [&](const f32x16 &src, int mt, int g) {
    __bf16 *row = epi.ys + (mt * 32 + (lane & 31)) * epi.stride + wn * 32 + 4 * (lane >> 5) + 8 * g;
    u32x2 hi, mid, lo;
    if constexpr (VAR & 2) {
      hi = u32x2{__builtin_bit_cast(unsigned, src[4 * g]), __builtin_bit_cast(unsigned, src[4 * g + 1])};
      mid = u32x2{__builtin_bit_cast(unsigned, src[4 * g + 2]), __builtin_bit_cast(unsigned, src[4 * g + 3])};
      lo = hi;
    } else {
      split4<true>(src[4 * g], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3], hi, mid, lo);
    }
    if constexpr (VAR & 1) {
      if (hi[0] == 0x12345678u && mid[1] == 0x9abcdefu && lo[0] == 0x31415926u) *reinterpret_cast<u32x2 *>(row) = hi;   // keeps the values live
    } else {
      *reinterpret_cast<u32x2 *>(row) = hi;
      *reinterpret_cast<u32x2 *>(row + epi.plane) = mid;
      *reinterpret_cast<u32x2 *>(row + 2 * epi.plane) = lo;
    }
  };
  mma(acc[0], wm, [](int) {});
#pragma unroll
  for (int u = 1; u < MAXU; ++u) {
    const int mt = wm + u * WM, pmt = wm + (u - 1) * WM;
    if (mt < mt_count) {
      mma(acc[u & 1], mt, [&](int q) { store_quarter(acc[(u - 1) & 1], pmt, q); });
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) store_quarter(acc[(u - 1) & 1], pmt, g);
      return;
    }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) store_quarter(acc[(MAXU - 1) & 1], wm + (MAXU - 1) * WM, g);
}

// N orientation with the masked-pool epilogue: unit = (M tile, n-tile of this wave); pooling of unit u (8 slices of two
// accumulator rows, gates read one step ahead) under the MFMAs of unit u+1.  Same arithmetic as run_layer_split_n.
// (A stricter variant -- one pool depth pinned behind each MFMA, operands prefetched across units -- reaches the same
// 47 cycles/MFMA in profiles/probe/split_layer_bench.hip and spills in the full kernel: the VALU budget per MFMA gap is
// ~4 instructions and the pool needs ~4 plus accumulator moves.)
template <int TP, int CINP, int COUTP, int ND, int NWV = 4>
__device__ __forceinline__ void run_layer_split_n_pool_pipe(const LayerWS<CINP, COUTP, NWV> &w,
                                                            const float (&bias)[LayerWS<CINP, COUTP, NWV>::NTW],
                                                            const __bf16 *__restrict__ xs, int xplane, int wave, int lane,
                                                            PoolMasked<ND, LayerWS<CINP, COUTP, NWV>::NTW> &pool, int mt_count) {
  using LW = LayerWS<CINP, COUTP, NWV>;
  static_assert(LW::WM == 1 && LW::KB == 8, "every wave walks all M tiles; 8 pool slices ride on 8 k-blocks");
  constexpr int XS = sp_stride(CINP), KB = LW::KB, NTW = LW::NTW, MAXU = (TP / 32) * NTW;
  const int nunits = mt_count * NTW;
  if (nunits == 0) return;
  f32x16 acc[2];
  const __bf16 *xbase = xs + (lane & 31) * XS + 8 * (lane >> 5);
  const float *gbase = pool.sgate + (4 * (lane >> 5)) * 4;
  f32x4 g[8][2];
  auto gate_read = [&](int mt, int j) {
    const int r = 2 * j;
    const int rb = (r & 3) + 8 * (r >> 2);
    g[j][0] = *reinterpret_cast<const f32x4 *>(gbase + (mt * 32 + rb) * 4);
    g[j][1] = *reinterpret_cast<const f32x4 *>(gbase + (mt * 32 + rb + 1) * 4);
  };
  auto pool_slice = [&](const f32x16 &src, int i, int j) {
#pragma unroll
    for (int d = 0; d < ND; ++d)
      pool.hold[d][i] = fmaxf(pool.hold[d][i], fmaxf(src[2 * j] + g[j][0][d], src[2 * j + 1] + g[j][1][d]));
  };
  auto mma = [&](f32x16 &dst, int mt, int i, auto &&pre, auto &&fill) {
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[r] = bias[i];
    const __bf16 *xr = xbase + mt * 32 * XS;
    bf16x8 a[KB][3];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int p = 0; p < 3; ++p) a[kb][p] = *reinterpret_cast<const bf16x8 *>(xr + p * xplane + kb * 16);
    pre();
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      if (kb + 2 < KB) {
#pragma unroll
        for (int p = 0; p < 3; ++p) a[kb + 2][p] = *reinterpret_cast<const bf16x8 *>(xr + p * xplane + (kb + 2) * 16);
      }
      fill(kb, 0);
      dst = mfma6<false>(a[kb], w.b[i][kb], dst);
      fill(kb, 1);
      __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
      interleave_step<3>(5, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  mma(acc[0], 0, 0, []() {}, [](int, int) {});
#pragma unroll
  for (int u = 1; u < MAXU; ++u) {
    const int pmt = (u - 1) / NTW, pi = (u - 1) % NTW;
    if (u < nunits) {
      mma(acc[u & 1], u / NTW, u % NTW, [&]() { gate_read(pmt, 0); },
          [&](int kb, int phase) {
            if (phase == 0) {
              if (kb + 1 < 8) gate_read(pmt, kb + 1);
            } else {
              pool_slice(acc[(u - 1) & 1], pi, kb);
            }
          });
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) gate_read(pmt, j);
#pragma unroll
      for (int j = 0; j < 8; ++j) pool_slice(acc[(u - 1) & 1], pi, j);
      return;
    }
  }
  {
    constexpr int pmt = (MAXU - 1) / NTW, pi = (MAXU - 1) % NTW;
#pragma unroll
    for (int j = 0; j < 8; ++j) gate_read(pmt, j);
#pragma unroll
    for (int j = 0; j < 8; ++j) pool_slice(acc[(MAXU - 1) & 1], pi, j);
  }
}

// ---- N orientation, unit = one M tile x ALL of this wave's n-tiles --------------------------------------------------
// The M tile's activation fragments (3 KB per k-block) and its gates are read from LDS once and feed NTW x 6 MFMAs,
// where the per-(M tile, n-tile) form above reads them once per n-tile (in this orientation every wave reads the whole
// activation tile).  With POOLED, the masked pooling of the PREVIOUS M tile (accumulators `prev`, gates at `gprev`) rides
// on this tile's 8 k-blocks: NTW x 8 slices of two accumulator rows each, gates read one k-block ahead.
//   xr    : this lane's fragment base of the tile = xs + (mt*32 + (lane&31))*XS + 8*(lane>>5)
//   gprev : this lane's gate base of the previous tile = sgate + (pmt*32 + 4*(lane>>5))*4
// VAR (profiles/probe/split_layer_bench.hip only): 1 = no gate reads, 2 = gate reads but no pooling arithmetic.
template <int ND, int NTW>
__device__ __forceinline__ void pool_gate_read(f32x4 (&g)[2][2], const float *__restrict__ gbase, int j) {
  const int r = 2 * j;
  const int rb = (r & 3) + 8 * (r >> 2);
  g[j & 1][0] = *reinterpret_cast<const f32x4 *>(gbase + rb * 4);
  g[j & 1][1] = *reinterpret_cast<const f32x4 *>(gbase + (rb + 1) * 4);
}
template <int ND, int NTW>
__device__ __forceinline__ void pool_slice(PoolMasked<ND, NTW> &pool, const f32x16 (&src)[NTW], const f32x4 (&g)[2][2], int j) {
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int d = 0; d < ND; ++d)
      pool.hold[d][i] = fmaxf(pool.hold[d][i], fmaxf(src[i][2 * j] + g[j & 1][0][d], src[i][2 * j + 1] + g[j & 1][1][d]));
}
// the pooling of one M tile on its own (nothing to ride on): ALL its gates are requested before the first slice, so the
// pass pays one LDS round trip instead of one per slice (it ran ~2x longer with the gates read a slice ahead)
template <int ND, int NTW>
__device__ __forceinline__ void pool_tile(PoolMasked<ND, NTW> &pool, const f32x16 (&src)[NTW], const float *__restrict__ gbase) {
  f32x4 g[8][2];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = 2 * j;
    const int rb = (r & 3) + 8 * (r >> 2);
    g[j][0] = *reinterpret_cast<const f32x4 *>(gbase + rb * 4);
    g[j][1] = *reinterpret_cast<const f32x4 *>(gbase + (rb + 1) * 4);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j)
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int d = 0; d < ND; ++d)
        pool.hold[d][i] = fmaxf(pool.hold[d][i], fmaxf(src[i][2 * j] + g[j][0][d], src[i][2 * j + 1] + g[j][1][d]));
}
struct NoHook { __device__ __forceinline__ void operator()(int) const {} };
// hook(kb) runs once per k-block between the MFMAs (independent VALU / LDS work riding along, e.g. the next chunk's layer 1)
// NPOOL: 0 = MFMAs only; 1 = the previous tile `prev` is pooled with the gates at gprev into `pool` under them; 2 = it is
// pooled twice, with gprev into `pool` and with gprevB into `poolB` (a tile holding rows of two pooling groups)
template <int NPOOL, int CINP, int COUTP, int ND, int NWV = 4, int VAR = 0, class Hook = NoHook>
__device__ __forceinline__ void split_n_unit(const LayerWS<CINP, COUTP, NWV> &w, const float (&bias)[LayerWS<CINP, COUTP, NWV>::NTW],
                                             const __bf16 *__restrict__ xr, int xplane,
                                             f32x16 (&dst)[LayerWS<CINP, COUTP, NWV>::NTW],
                                             const f32x16 (&prev)[LayerWS<CINP, COUTP, NWV>::NTW],
                                             PoolMasked<ND, LayerWS<CINP, COUTP, NWV>::NTW> &pool, const float *__restrict__ gprev,
                                             PoolMasked<ND, LayerWS<CINP, COUTP, NWV>::NTW> &poolB, const float *__restrict__ gprevB,
                                             Hook &&hook = Hook{}) {
  using LW = LayerWS<CINP, COUTP, NWV>;
  static_assert(LW::KB == 8, "8 pool slices ride on 8 k-blocks");
  constexpr int KB = LW::KB, NTW = LW::NTW;
  constexpr bool HOOKED = !__is_same(__remove_cvref(Hook), NoHook);
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[i][r] = bias[i];
  bf16x8 a[KB][3];
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int p = 0; p < 3; ++p) a[kb][p] = *reinterpret_cast<const bf16x8 *>(xr + p * xplane + kb * 16);
  f32x4 g[2][2], gB[2][2];
  auto gread = [&](int j) {
    if constexpr (VAR & 1) g[j & 1][0] = g[j & 1][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    else pool_gate_read<ND, NTW>(g, gprev, j);
    if constexpr (NPOOL > 1) pool_gate_read<ND, NTW>(gB, gprevB, j);
  };
  if constexpr (NPOOL > 0) gread(0);
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    if (kb + 2 < KB) {
#pragma unroll
      for (int p = 0; p < 3; ++p) a[kb + 2][p] = *reinterpret_cast<const bf16x8 *>(xr + p * xplane + (kb + 2) * 16);
    }
    if constexpr (NPOOL > 0) { if (kb + 1 < 8) gread(kb + 1); }
#pragma unroll
    for (int i = 0; i < NTW; ++i) dst[i] = mfma6<false>(a[kb], w.b[i][kb], dst[i]);
    if constexpr (NPOOL > 0) {
      if constexpr (VAR & 2) pool.hold[0][0] = fmaxf(pool.hold[0][0], prev[0][2 * kb] + g[kb & 1][0][0] + g[kb & 1][1][3]);
      else pool_slice<ND, NTW>(pool, prev, g, kb);
      if constexpr (NPOOL > 1) pool_slice<ND, NTW>(poolB, prev, gB, kb);
    }
    hook(kb);
    if constexpr (HOOKED || NPOOL > 1) {   // more riders per MFMA: ask for 4-5 VALU per gap and leave the LDS operations free
#pragma unroll
      for (int i = 0; i < NTW; ++i) interleave_step<(NPOOL > 1 ? 5 : 4)>(5, 0);
    } else {
      __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
#pragma unroll
      for (int i = 0; i < NTW; ++i) interleave_step<3>(5, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// A whole chunk of up to TP/32 M tiles of ONE pooling group: tile t's pooling under tile t+1's MFMAs, the last tile's on
// its own (CloudCrop layer 3 moved 320 KB through LDS per 32 rows in the per-n-tile form, 160 KB in this one).
template <int TP, int CINP, int COUTP, int ND, int NWV = 4, int VAR = 0>
__device__ __forceinline__ void run_layer_split_n_pool_pipe2(const LayerWS<CINP, COUTP, NWV> &w,
                                                             const float (&bias)[LayerWS<CINP, COUTP, NWV>::NTW],
                                                             const __bf16 *__restrict__ xs, int xplane, int wave, int lane,
                                                             PoolMasked<ND, LayerWS<CINP, COUTP, NWV>::NTW> &pool, int mt_count) {
  using LW = LayerWS<CINP, COUTP, NWV>;
  static_assert(LW::WM == 1, "every wave walks all M tiles");
  constexpr int XS = sp_stride(CINP), NTW = LW::NTW, MAXU = TP / 32;
  static_assert(MAXU <= 4, "written out for up to four M tiles per chunk");
  if (mt_count == 0) return;
  f32x16 accA[NTW], accB[NTW];     // two named buffers (an array of buffers indexed by parity ends up in scratch)
  const __bf16 *xbase = xs + (lane & 31) * XS + 8 * (lane >> 5);
  const float *gbase = pool.sgate + (4 * (lane >> 5)) * 4;
  auto xr = [&](int mt) { return xbase + mt * 32 * XS; };
  auto gr = [&](int mt) { return gbase + mt * 32 * 4; };
  split_n_unit<0, CINP, COUTP, ND, NWV, VAR>(w, bias, xr(0), xplane, accA, accA, pool, gbase, pool, gbase);
  if constexpr (MAXU > 1) {
    if (1 < mt_count) split_n_unit<1, CINP, COUTP, ND, NWV, VAR>(w, bias, xr(1), xplane, accB, accA, pool, gr(0), pool, gbase);
    else { pool_tile<ND, NTW>(pool, accA, gr(0)); return; }
  }
  if constexpr (MAXU > 2) {
    if (2 < mt_count) split_n_unit<1, CINP, COUTP, ND, NWV, VAR>(w, bias, xr(2), xplane, accA, accB, pool, gr(1), pool, gbase);
    else { pool_tile<ND, NTW>(pool, accB, gr(1)); return; }
  }
  if constexpr (MAXU > 3) {
    if (3 < mt_count) split_n_unit<1, CINP, COUTP, ND, NWV, VAR>(w, bias, xr(3), xplane, accB, accA, pool, gr(2), pool, gbase);
    else { pool_tile<ND, NTW>(pool, accA, gr(2)); return; }
  }
  if constexpr (((MAXU - 1) & 1) == 0) pool_tile<ND, NTW>(pool, accA, gr(MAXU - 1));
  else pool_tile<ND, NTW>(pool, accB, gr(MAXU - 1));
}

// ---- first layer (K = 3 coordinates, padded to 8) stays on the fp32 MFMA, T orientation, split epilogue -------
template <int TP, int COUTP, int NWV = 4>
__device__ __forceinline__ void run_layer_f32_t(const LayerW<8, COUTP, NWV> &w, const float (&bias_t)[16],
                                                const float *__restrict__ xs, int wave, int lane, StoreSplit &epi,
                                                int mt_count = TP / 32) {
  using LW = LayerW<8, COUTP, NWV>;
  static_assert(LW::NTW == 1, "one n-tile per wave");
  constexpr int XS = lds_stride(8);
  const int wm = wave / LW::WN, wn = wave % LW::WN;
  for (int mt = wm; mt < mt_count; mt += LW::WM) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = bias_t[r];
    const f32x4 a = *reinterpret_cast<const f32x4 *>(xs + (mt * 32 + (lane & 31)) * XS + 4 * (lane >> 5));
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.b[0][0][s], a[s], acc, 0, 0, 0);
    epi(mt, wn, acc);
  }
}

// the same with the accumulators starting from per-position rows gathered by the caller (the hoisted per-point part of
// a set-abstraction layer 1, mlp_group.hip sa_hoist_kernel): c0[mi] holds, for this wave's mi-th M tile (wm + mi*WM),
// the 16 accumulator rows (channels nt*32 + acc_row(r)) of this lane's position.
template <int TP, int COUTP, int MTW, int NWV = 4>
__device__ __forceinline__ void run_layer_f32_t_cinit(const LayerW<8, COUTP, NWV> &w, const f32x4 (&c0)[MTW][4],
                                                      const float *__restrict__ xs, int wave, int lane, StoreSplit &epi) {
  using LW = LayerW<8, COUTP, NWV>;
  static_assert(LW::NTW == 1 && MTW * LW::WM == TP / 32, "one n-tile per wave, M tiles divide over the wave rows");
  constexpr int XS = lds_stride(8);
  const int wm = wave / LW::WN, wn = wave % LW::WN;
#pragma unroll
  for (int mi = 0; mi < MTW; ++mi) {
    const int mt = wm + mi * LW::WM;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * g + j] = c0[mi][g][j];
    const f32x4 a = *reinterpret_cast<const f32x4 *>(xs + (mt * 32 + (lane & 31)) * XS + 4 * (lane >> 5));
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.b[0][0][s], a[s], acc, 0, 0, 0);
    epi(mt, wn, acc);
  }
}

}  // namespace gn
