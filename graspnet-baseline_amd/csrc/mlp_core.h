// mlp_core.h -- fp32 MFMA building blocks for the fused "gather -> shared MLP -> pool" kernels.
//
// Replaces, in fused form, the reference's SharedMLP stacks
// (pointnet2/pytorch_utils.py:11-120: conv1x1 -> BatchNorm(eval) -> ReLU, executed there as
// separate cuDNN / elementwise kernels with every intermediate written to HBM).
//
// Orientation.  Activations are kept POSITION-major ("channels last"): a tile is X[TP][C] in LDS.
// One layer is  Y[TP][COUT] = act( X[TP][CIN] * W^T + bias )  with BN folded into W / bias on the
// host.  It is computed with v_mfma_f32_32x32x2_f32 (exact fp32, CDNA4):
//     A operand = activations  (row = position  = lane&31, k = lane>>5)
//     B operand = weights      (col = out chan  = lane&31, k = lane>>5)
//     C/D       = 32 positions x 32 channels:  col(chan) = lane&31,
//                 row(pos) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
// so the max-pool over a group's positions is an in-lane max over accumulator registers (plus one
// lane^32 exchange), and the next layer's A operand is read back from LDS as contiguous float4s.
//
// k mapping.  A lane reads 4 consecutive channels X[pos][kb*8 + 4*(lane>>5) + s], s=0..3, with one
// ds_read_b128 and feeds MFMA step s with element s; the host packs the weights in exactly that
// fragment order (graspnet_amd/weights.py: pack_fragments) so a B fragment is one coalesced
// 16-byte-per-lane load:   Wp[nt][kb][lane][s] = W[nt*32 + (lane&31)][kb*8 + 4*(lane>>5) + s].
//
// Weights stay STATIONARY in registers (gfx950: 512 unified VGPR/AGPR per lane at one wave per
// SIMD): a workgroup of 4 waves is persistent over many position tiles, each wave owns a slice of
// the output channels of every layer and streams position tiles from LDS past its weights, so the
// inner loop touches HBM/L2 for nothing and LDS only for the A operand (one b128 read per 4*NTW
// MFMAs).
//
// LDS row stride is C+4 floats: (C+4)/4 is odd, which makes the 16-lane groups of ds_read_b128
// (MI355X guide, LDS table) hit 16 distinct 16-byte slots -> conflict-free A reads.
#pragma once
#include "gn_common.h"

namespace gn {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int MLP_THREADS = 256;  // 4 waves, one per SIMD

__host__ __device__ constexpr int lds_stride(int c) { return c + 4; }
__host__ __device__ constexpr int wn_for(int coutp, int nwv = 4) { return (coutp / 32) >= nwv ? nwv : (coutp / 32); }
__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

// ---- one layer's weight slice of this wave, resident in registers
template <int CINP, int COUTP, int NWV = 4>
struct LayerW {
  static_assert(CINP % 8 == 0 && COUTP % 32 == 0, "pad CIN to 8 and COUT to 32");
  static constexpr int NT = COUTP / 32;
  static constexpr int WN = wn_for(COUTP, NWV);
  static constexpr int WM = NWV / WN;
  static_assert(NT % WN == 0, "N tiles must divide over the waves");
  static constexpr int NTW = NT / WN;
  static constexpr int KB = CINP / 8;
  f32x4 b[NTW][KB];
  float bias[NTW];

  __device__ __forceinline__ void load(const float *__restrict__ wp, const float *__restrict__ bp, int wave,
                                       int lane) {
    const int wn = wave % WN;
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const int nt = wn * NTW + i;
#pragma unroll
      for (int kb = 0; kb < KB; ++kb)
        b[i][kb] = *reinterpret_cast<const f32x4 *>(wp + ((size_t)(nt * KB + kb) * 64 + lane) * 4);
      bias[i] = bp[nt * 32 + (lane & 31)];
    }
  }
};

__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- epilogues ------------------------------------------------------------------------------
// write act(acc) to an LDS activation tile Y[TP][ys]; only channels < ncols exist in Y
template <bool RELU>
struct StoreLds {
  float *ys;
  int stride;
  int ncols;
  int lane;
  __device__ __forceinline__ void begin() {}
  __device__ __forceinline__ void operator()(int mt, int nt, int /*i*/, const f32x16 &acc) {
    const int col = nt * 32 + (lane & 31);
    if (col < ncols) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float v = RELU ? fmaxf(acc[r], 0.f) : acc[r];
        ys[(mt * 32 + acc_row(r, lane)) * stride + col] = v;
      }
    }
  }
};

// write act(acc) to global, row-major out[row0 + pos][ldo], channels < ncols, rows < nrows
template <bool RELU>
struct StoreGlobal {
  float *out;   // already offset to the tile's first row
  int ldo;
  int ncols;
  int nrows;    // valid rows in this tile
  int lane;
  __device__ __forceinline__ void begin() {}
  __device__ __forceinline__ void operator()(int mt, int nt, int /*i*/, const f32x16 &acc) {
    const int col = nt * 32 + (lane & 31);
    if (col < ncols) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + acc_row(r, lane);
        const float v = RELU ? fmaxf(acc[r], 0.f) : acc[r];
        if (row < nrows) out[(size_t)row * ldo + col] = v;
      }
    }
  }
};

// max of a value with its partner in the other 32-lane half (the two halves of an accumulator column):
// one v_permlane32_swap (gfx950: lanes 32-63 of the first operand swap with lanes 0-31 of the second) instead
// of a trip through the LDS crossbar.  Inline asm: with ROCm 7.2 the builtin given two copies of one value
// returned the same register twice (profiles/probe/permlane_test.hip).  s_nop 1 = the VALU-write -> permlane
// read hazard.
__device__ __forceinline__ float xhalf_max(float v, int /*lane*/) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

// ReLU + max over each group of NS consecutive positions -> out[group][ldo] (channels-last)
// (F.max_pool2d(kernel=[1,nsample]) of pointnet2_modules.py:255 / modules.py:137)
template <int NS, int NTW>
// probe builds only (profiles/build_probe.sh PROBE_DEFS=-DGN_DBG_NOSTORE): drop the pooled rows' stores (v == 12345.f never holds)
#ifdef GN_DBG_NOSTORE
#define GN_DBG_STORE_OK (acc[0] == 12345.f)
#else
#define GN_DBG_STORE_OK true
#endif
struct PoolGlobal {
  float *out;  // offset to the tile's first group
  int ldo;
  int lane;
  float hold[NTW];
  __device__ __forceinline__ void begin() {}
  __device__ __forceinline__ void operator()(int mt, int nt, int i, const f32x16 &acc) {
    const int col = nt * 32 + (lane & 31);
    if (NS == 16) {
      float v0 = acc[0], v1 = acc[8];
#pragma unroll
      for (int r = 1; r < 8; ++r) { v0 = fmaxf(v0, acc[r]); v1 = fmaxf(v1, acc[8 + r]); }
      v0 = xhalf_max(v0, lane);
      v1 = xhalf_max(v1, lane);
      if (GN_DBG_STORE_OK && lane < 32) {
        out[(size_t)(mt * 2 + 0) * ldo + col] = fmaxf(v0, 0.f);
        out[(size_t)(mt * 2 + 1) * ldo + col] = fmaxf(v1, 0.f);
      }
    } else {
      float v = acc[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[r]);
      v = xhalf_max(v, lane);
      if (NS == 32) {
        if (GN_DBG_STORE_OK && lane < 32) out[(size_t)mt * ldo + col] = fmaxf(v, 0.f);
      } else {  // NS == 64: two consecutive M tiles form a group (requires WM == 1)
        static_assert(NS == 16 || NS == 32 || NS == 64, "unsupported nsample");
        if ((mt & 1) == 0) {
          hold[i] = v;
        } else {
          v = fmaxf(v, hold[i]);
          if (GN_DBG_STORE_OK && lane < 32) out[(size_t)(mt >> 1) * ldo + col] = fmaxf(v, 0.f);
        }
      }
    }
  }
};

// ReLU + max over the rows of a ragged row set, each row contributing to the depths named by its
// ND-bit mask.  The loader stores the mask as ND additive gates per row (0 = member, -inf = not):
// acc + gate is exact for members and -inf otherwise, so pooling is one add + one max per
// (row, depth) with no bit tests.  hold[][] persists over M tiles / chunks; finish() writes the
// ReLU'd maxima.  Tile-padding rows carry all -inf gates.
template <int ND, int NTW>
struct PoolMasked {
  static_assert(ND == 4, "gates are stored as one float4 per row");
  const float *sgate;   // LDS, [row][4]
  int lane;
  float hold[ND][NTW];
  __device__ __forceinline__ void reset() {
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int i = 0; i < NTW; ++i) hold[d][i] = -INFINITY;
  }
  __device__ __forceinline__ void operator()(int mt, int /*nt*/, int i, const f32x16 &acc) {
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const f32x4 g0 = *reinterpret_cast<const f32x4 *>(sgate + (mt * 32 + acc_row(r, lane)) * 4);
      const f32x4 g1 = *reinterpret_cast<const f32x4 *>(sgate + (mt * 32 + acc_row(r + 1, lane)) * 4);
#pragma unroll
      for (int d = 0; d < ND; ++d) hold[d][i] = fmaxf(hold[d][i], fmaxf(acc[r] + g0[d], acc[r + 1] + g1[d]));
    }
  }
  // out points at the seed's first depth row; depth rows are `ldo` floats apart
  __device__ __forceinline__ void finish(float *out, int ldo, int nt0) {
    float v[ND][NTW];       // all cross-half maxima first, then ONE predicated block of stores (not a branch per value)
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
      for (int i = 0; i < NTW; ++i) v[d][i] = fmaxf(xhalf_max(hold[d][i], lane), 0.f);
    if (lane < 32) {
#pragma unroll
      for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int i = 0; i < NTW; ++i) out[(size_t)d * ldo + (nt0 + i) * 32 + lane] = v[d][i];
    }
  }
};

// ---- one layer over a TP-position LDS tile, weights resident --------------------------------
template <int TP, int CINP, int COUTP, class Epi, int NWV = 4>
__device__ __forceinline__ void run_layer(const LayerW<CINP, COUTP, NWV> &w, const float *__restrict__ xs, int wave,
                                          int lane, Epi &epi, int mt_count = TP / 32) {
  using LW = LayerW<CINP, COUTP, NWV>;
  constexpr int XS = lds_stride(CINP);
  const int wm = wave / LW::WN;
  const int wn = wave % LW::WN;
  for (int mt = wm; mt < mt_count; mt += LW::WM) {
    f32x16 acc[LW::NTW];
#pragma unroll
    for (int i = 0; i < LW::NTW; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = w.bias[i];
    const float *xr = xs + (mt * 32 + (lane & 31)) * XS + 4 * (lane >> 5);
    // All A fragments are read in program order, then consumed; the sched_group_barrier ladder
    // below pins the emitted order to [2 reads] ([4*NTW MFMAs][1 read])* so every ds_read_b128 is in
    // flight two k-blocks ahead of its MFMAs (hipcc otherwise reads right before use and waits).
    f32x4 a[LW::KB];
#pragma unroll
    for (int kb = 0; kb < LW::KB; ++kb) a[kb] = *reinterpret_cast<const f32x4 *>(xr + kb * 8);
#pragma unroll
    for (int kb = 0; kb < LW::KB; ++kb)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < LW::NTW; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kb][s], w.b[i][kb][s], acc[i], 0, 0, 0);
    if constexpr (LW::KB > 2) {
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                 // DS read x2
#pragma unroll
      for (int kb = 0; kb < LW::KB - 2; ++kb) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * LW::NTW, 0);     // MFMAs of k-block kb
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // read k-block kb+2
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 8 * LW::NTW, 0);
    }
#pragma unroll
    for (int i = 0; i < LW::NTW; ++i) epi(mt, wn * LW::NTW + i, i, acc[i]);
  }
}

// ---- like run_layer, but the accumulators start from an LDS tile C0[TP][cstride] instead of the
// bias (the hoisted, per-point part of a layer), and the result may overwrite that tile in place
// (each 32x32 tile is read and written by the same wave only).
template <int TP, int CINP, int COUTP, class Epi>
__device__ __forceinline__ void run_layer_cinit(const LayerW<CINP, COUTP> &w, const float *__restrict__ xs,
                                                const float *__restrict__ c0, int cstride, int wave, int lane,
                                                Epi &epi, int mt_count = TP / 32) {
  using LW = LayerW<CINP, COUTP>;
  constexpr int XS = lds_stride(CINP);
  const int wm = wave / LW::WN;
  const int wn = wave % LW::WN;
  for (int mt = wm; mt < mt_count; mt += LW::WM) {
    f32x16 acc[LW::NTW];
#pragma unroll
    for (int i = 0; i < LW::NTW; ++i) {
      const int col = (wn * LW::NTW + i) * 32 + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = c0[(mt * 32 + acc_row(r, lane)) * cstride + col];
    }
    const float *xr = xs + (mt * 32 + (lane & 31)) * XS + 4 * (lane >> 5);
#pragma unroll
    for (int kb = 0; kb < LW::KB; ++kb) {
      const f32x4 a = *reinterpret_cast<const f32x4 *>(xr + kb * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < LW::NTW; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], w.b[i][kb][s], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < LW::NTW; ++i) epi(mt, wn * LW::NTW + i, i, acc[i]);
  }
}

// ---- one layer, weights streamed from L2 in K chunks (for layers too wide to keep resident) ---
// accumulators for all of this wave's tiles persist across the K chunks.
template <int TP, int CINP, int COUTP, int KC, class Epi, int XS = lds_stride(CINP)>
__device__ __forceinline__ void run_layer_stream(const float *__restrict__ wp, const float *__restrict__ bp,
                                                 const float *__restrict__ xs, int wave, int lane, Epi &epi) {
  static_assert(CINP % KC == 0 && KC % 8 == 0, "bad K chunk");
  constexpr int NT = COUTP / 32;
  constexpr int WN = wn_for(COUTP);
  constexpr int WM = 4 / WN;
  static_assert(NT % WN == 0, "N tiles must divide over the waves");
  constexpr int NTW = NT / WN;
  constexpr int MT = TP / 32;
  static_assert(MT % WM == 0, "M tiles must divide over the waves");
  constexpr int MTW = MT / WM;
  constexpr int KB = CINP / 8, KBC = KC / 8;
  const int wm = wave / WN, wn = wave % WN;
  // the weights are read-only and noalias, so LLVM would hoist their loads above the workgroup barrier
  // that precedes this layer (and above the previous layers): launder the pointers here to pin them
  asm volatile("" : "+s"(wp), "+s"(bp));
  f32x16 acc[MTW][NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) {
    const float bv = bp[(wn * NTW + i) * 32 + (lane & 31)];
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][i][r] = bv;
  }
  // Weight chunks are double-buffered in registers: chunk c+1 is requested from L2 before the MFMAs of
  // chunk c are issued, and sched_barriers pin that order (the chunk loop is fully unrolled so both
  // buffers are statically indexed; without the fences hipcc hoists every chunk's loads and spills).
  constexpr int NCH = KB / KBC;
  f32x4 bbuf[2][NTW][KBC];
  auto load_chunk = [&](int c, f32x4 (&dst)[NTW][KBC]) {
#pragma unroll
    for (int i = 0; i < NTW; ++i)
#pragma unroll
      for (int kb = 0; kb < KBC; ++kb)
        dst[i][kb] = *reinterpret_cast<const f32x4 *>(wp + ((size_t)((wn * NTW + i) * KB + c * KBC + kb) * 64 + lane) * 4);
  };
  load_chunk(0, bbuf[0]);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (c + 1 < NCH) load_chunk(c + 1, bbuf[(c + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    const int kc = c * KBC;
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi) {
      const int mt = wm * MTW + mi;
      const float *xr = xs + (mt * 32 + (lane & 31)) * XS + 4 * (lane >> 5) + kc * 8;
      f32x4 a[KBC];
#pragma unroll
      for (int kb = 0; kb < KBC; ++kb) a[kb] = *reinterpret_cast<const f32x4 *>(xr + kb * 8);
#pragma unroll
      for (int kb = 0; kb < KBC; ++kb)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int i = 0; i < NTW; ++i)
            acc[mi][i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[kb][s], bbuf[c & 1][i][kb][s], acc[mi][i], 0, 0, 0);
      if constexpr (KBC > 2) {   // same read-ahead ladder as run_layer
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
        for (int kb = 0; kb < KBC - 2; ++kb) {
          __builtin_amdgcn_sched_group_barrier(0x008, 4 * NTW, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 8 * NTW, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
    for (int i = 0; i < NTW; ++i) epi(wm * MTW + mi, wn * NTW + i, i, acc[mi][i]);
}

// Phase probe (profiles/build_probe.sh, -DGN_PHASE_PROBE): per-wave cycle sums between marks, added into a global
// table read by profiles/crop_phase_probe.py.  Compiled out of the product library.
#ifdef GN_PHASE_PROBE
static __device__ unsigned long long gn_phase_probe[8 * 16];   // one table per translation unit
#define GN_PROBE_INIT() unsigned long long pt_[16] = {}; unsigned long long pl_ = __builtin_amdgcn_s_memtime()
#define GN_PROBE(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); pt_[i] += n_ - pl_; pl_ = n_; } while (0)
#define GN_PROBE_COUNT(i, v) pt_[i] += (unsigned long long)(v)
// a finer mark INSIDE a phase: slot i gets the time since the previous mark of either kind, and the enclosing phase's own
// GN_PROBE later gets only what is left (sub-marks are taken out of it)
#define GN_PROBE_SUB(i) GN_PROBE(i)
#define GN_PROBE_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&gn_phase_probe[(threadIdx.x >> 6) * 16 + i_], pt_[i_]); } while (0)
#define GN_PROBE_READER(fn)                                                                                         \
  extern "C" int fn(unsigned long long *host, int reset) {                                                          \
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(gn::gn_phase_probe), sizeof(unsigned long long) * 128) != hipSuccess) return -1; \
    if (reset) {                                                                                                    \
      unsigned long long z[128] = {};                                                                               \
      if (hipMemcpyToSymbol(HIP_SYMBOL(gn::gn_phase_probe), z, sizeof(z)) != hipSuccess) return -1;                 \
    }                                                                                                               \
    return 0;                                                                                                       \
  }
#else
#define GN_PROBE_READER(fn)
#define GN_PROBE_INIT()
#define GN_PROBE(i)
#define GN_PROBE_COUNT(i, v)
#define GN_PROBE_SUB(i)
#define GN_PROBE_FLUSH()
#endif

}  // namespace gn
