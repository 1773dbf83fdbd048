// mlp_point_stream.h -- what the per-point chain kernels share (mlp_point_split.hip: activations as three bf16 planes in LDS,
// 32- / 64-row tiles; mlp_point_wide.hip: activations as fp32 in LDS split at fragment load, 128-row tiles): the argument
// block, the shape of a streamed layer and one chunk of a wave's split weight fragments in registers.
#pragma once
#include "mlp_split.h"

namespace gn {

struct PointSplitArgs {
  const void *w[3];
  const float *b[3];
  const float *x;       // plain loader: (rows, C0) channels-last
  const float *known;   // fp loader: (B, m, CK) features to interpolate
  const float *skip;    //            (B, n, CS) skip features
  const float *dist2;   //            (B, n, 3)
  const int32_t *idx3;  //            (B, n, 3)
  int n, m;
  float *out;           // (rows, ldo)
  int ldo, ncols_out;
  int nrows;
};

// One streamed layer, T orientation.  MFMA phase: every (M tile, n-tile) of this wave accumulates over all K with the
// weight chunks double-buffered; then `retire()` (a workgroup barrier when the output overwrites the input) and the
// epilogue epi(mt, nt, acc) per tile.
// bias of the 16 accumulator rows of n-tile nt (T orientation: rows = channels 8g + 4*(lane>>5) + j): four 16-byte loads,
// issued when the layer starts and consumed by its epilogue, so their L2 round trip hides under the layer's MFMAs
__device__ __forceinline__ void load_bias_t4(const float *__restrict__ bp, int nt, int lane, f32x4 (&bt)[4]) {
  using gf4 = const __attribute__((address_space(1))) f32x4;     // global, not flat: see StreamChunk::load
  const __attribute__((address_space(1))) float *gb = (const __attribute__((address_space(1))) float *)bp;
#pragma unroll
  for (int g = 0; g < 4; ++g) bt[g] = *(gf4 *)(gb + nt * 32 + 8 * g + 4 * (lane >> 5));
}

// shape of a streamed layer for this kernel family.  CHREG: registers one weight chunk may take per wave (96 with one
// workgroup per CU, 48 with two -- a wave then has 256 registers, 64-96 of them accumulators)
// CHREG >= 1000 carries a wave layout with it: CHREG = 1000 * WN + registers forces WN waves along the output channels
// (mlp_point_wide.hip gives the 64-wide last layer of the heads WN = 1: a wave then takes ONE M tile and both n-tiles, so a
// fragment it splits feeds 12 MFMAs instead of 6)
template <int TP, int CINP, int COUTP, int CHREG_ = 96>
struct StreamCfg {
  static_assert(CINP % 32 == 0 && COUTP % 32 == 0, "K in chunks of at least 32");
  static constexpr int CHREG = CHREG_ % 1000, WNO = CHREG_ / 1000;
  static constexpr int NT = COUTP / 32, MT = TP / 32, WN = WNO > 0 ? WNO : wn_for(COUTP);
  static constexpr int WM = (4 / WN < MT) ? 4 / WN : MT;      // a 32-row tile of a 64-wide layer occupies two waves only
  static constexpr int NACTIVE = WN * WM, NTW = NT / WN;
  static_assert(NT % WN == 0 && MT % WM == 0, "tiles must divide over the waves");
  // k-blocks per weight chunk: at least ~48 MFMAs behind every chunk load (an L2 round trip) where the registers allow,
  // so small layers take longer chunks
  static constexpr int MTW = MT / WM, KB = CINP / 16;
  static constexpr int KBC0 = (MTW * NTW >= 4) ? 2 : (MTW * NTW >= 2 ? 4 : 8);
  static constexpr int KBC1 = (NTW * KBC0 * 12 <= CHREG) ? KBC0 : (NTW * (KBC0 / 2) * 12 <= CHREG || KBC0 / 2 <= 1 ? cmax(KBC0 / 2, 1)
                                                                                                                   : cmax(KBC0 / 4, 1));
  static constexpr int KBC = (KB % KBC1 == 0) ? KBC1 : (KBC1 > 2 ? 2 : 1);
  static constexpr int NCH = KB / KBC;
  static_assert(KB % KBC == 0, "K must be a multiple of the chunk");
};

// one chunk of this wave's split weight fragments in registers
template <int TP, int CINP, int COUTP, int CHREG>
struct StreamChunk {
  using C = StreamCfg<TP, CINP, COUTP, CHREG>;
  bf16x8 v[C::NTW][C::KBC][3];
  // The weights are addressed as GLOBAL memory explicitly.  Behind the `asm` that pins a layer's pointer below its barrier
  // the compiler no longer knows the address space and emitted flat_load -- which counts on lgkmcnt as well as vmcnt, so
  // the first s_waitcnt lgkmcnt(0) for an LDS operand of the CURRENT chunk also waited for the NEXT chunk's weights: the
  // double buffering never overlapped anything (58 / 75 / 91 cycles per MFMA in the heads' layers against 34 bare).
  __device__ __forceinline__ void load(const void *__restrict__ wq, int c, int wave, int lane) {
    const int wn = wave % C::WN;
    using gchar = const __attribute__((address_space(1))) char;
    using gfrag = const __attribute__((address_space(1))) bf16x8;
    gchar *wbase = (gchar *)wq;
#pragma unroll
    for (int i = 0; i < C::NTW; ++i)
#pragma unroll
      for (int kb = 0; kb < C::KBC; ++kb)
#pragma unroll
        for (int p = 0; p < 3; ++p)
          v[i][kb][p] = *(gfrag *)(wbase + ((((size_t)((wn * C::NTW + i) * C::KB + c * C::KBC + kb)) * 3 + p) * 64 + lane) * 16);
  }
  // called where the chunk is consumed (its data is needed there anyway; at the load an asm operand would force the
  // wait and undo the double buffering): from here on the fragments live in AGPRs and the MFMAs read them from there
  // (mlp_split.h LayerWS::load), so the VGPRs stay free for the two accumulators per tile
  __device__ __forceinline__ void pin() {
#pragma unroll
    for (int i = 0; i < C::NTW; ++i)
#pragma unroll
      for (int kb = 0; kb < C::KBC; ++kb)
#pragma unroll
        for (int p = 0; p < 3; ++p) asm volatile("" : "+a"(v[i][kb][p]));
  }
};

// Every layer used to start with an L2 round trip for its first weight chunk (~2.5 k cycles with nothing else to do: the
// phase probe shows 58 / 75 / 91 cycles per MFMA in the heads' three layers against 34 bare).  The first chunk of layer
// L+1 is now requested by layer L just before its epilogue -- its own chunk buffers are dead by then, so the registers are
// there -- and lands under the epilogue and the barrier behind it; layer 1's is requested at the tile top, under the row
// fetch.  (Requesting it any earlier, while layer L still streams, keeps three layers' buffers live and spills.)
// BLOCK2: the layer is two independent halves side by side (OperationNet | ToleranceNet, modules.py:171-175,216-220, packed
// as ONE block-diagonal matrix): output channels of the first half read only the first half of K, the others only the
// second, and the other K chunks of a wave's weight slice are all zeros -- they are not streamed and not multiplied.
// first chunk of a (BLOCK2) layer for this wave
template <int TP, int CINP, int COUTP, int CHREG, bool BLOCK2>
__device__ __forceinline__ int stream_first_chunk(int wave) {
  using C = StreamCfg<TP, CINP, COUTP, CHREG>;
  return BLOCK2 ? ((wave % C::WN) >= C::WN / 2 ? C::NCH / 2 : 0) : 0;
}
// request the first chunk of a layer (no-op for the waves the layer leaves idle)
template <int TP, int CINP, int COUTP, int CHREG, bool BLOCK2>
__device__ __forceinline__ void stream_prefetch(StreamChunk<TP, CINP, COUTP, CHREG> &ch, const void *__restrict__ wq, int wave, int lane) {
  using C = StreamCfg<TP, CINP, COUTP, CHREG>;
  if (C::NACTIVE < 4 && wave >= C::NACTIVE) return;
  ch.load(wq, stream_first_chunk<TP, CINP, COUTP, CHREG, BLOCK2>(wave), wave, lane);
}

// mlp_point_wide.hip: the chains on fp32 tiles split at fragment load -- the heads on 128-row tiles (weights streamed once
// per 128 rows), ApproachNet and the feature-propagation chains on 64-row tiles
int heads_mlp_wide(const PointSplitArgs &a, hipStream_t s);
int approach_mlp_wide(const PointSplitArgs &a, hipStream_t s);
int fp_mlp_wide(const PointSplitArgs &a, hipStream_t s);

// Which tile shapes the per-point chains take (gn_set_point_tile_policy; GN_POINT_TILES=latency|throughput overrides):
//   0 latency     every launch spreads over as many CUs as it has 32-row tiles (ApproachNet, FP) -- the shortest launch when
//                 nothing else runs beside it (one batch at a time, hipGraph replay);
//   1 throughput  the fewest weight bytes per row: 64-row fp32 tiles for ApproachNet / FP (half as many workgroups, each CU
//                 streams the weights once per 64 rows), 128-row tiles for the heads whatever the row count -- fewer CU-
//                 microseconds per launch, for callers that keep several batches in flight (PipelinedGraspNet, bench.py).
int point_tile_policy();

}  // namespace gn
