// mlp_point_split.hip -- the per-point MLP chains of mlp_point.hip (feature propagation, ApproachNet, OperationNet +
// ToleranceNet, the hoisted per-point linear map) on the bf16 matrix cores by operand splitting (mlp_split.h).
//
// A tile of TP positions lives in LDS as three bf16 planes [TP][CMAX+8]; ONE buffer serves every layer: a layer's MFMAs
// consume the whole tile (weights streamed from L2 as split fragments, 32 k per chunk, double-buffered in registers),
// a barrier retires the input, and the epilogue writes relu(y) -- split again -- over it.  All layers run in the T
// orientation (A = weights, B = activations): a lane ends up with 4 consecutive channels of its position per register
// quad, i.e. one 8-byte LDS store per plane, or 4 consecutive floats of an output row.
#include "mlp_point_stream.h"

namespace gn {

__device__ __forceinline__ void store_split4(__bf16 *row, int plane, const f32x4 &v) {
  u32x2 hi, mid, lo;
  split4<false>(v[0], v[1], v[2], v[3], hi, mid, lo);
  *reinterpret_cast<u32x2 *>(row) = hi;
  *reinterpret_cast<u32x2 *>(row + plane) = mid;
  *reinterpret_cast<u32x2 *>(row + 2 * plane) = lo;
}

// rows of a plain channels-last tensor -> planes; rows beyond nrows are zero.  The loads of a batch of up to 16 float4 per
// thread are all issued before the first one is split (one memory round trip per batch instead of one per float4).
template <int C0, int TP, int XS>
__device__ __forceinline__ void load_plain_split(__bf16 *__restrict__ ys, int plane, const float *__restrict__ x, int row0,
                                                 int nrows, int tid) {
  constexpr int Q = C0 / 4, PER = TP * Q / MLP_THREADS, BATCH = PER % 16 == 0 ? 16 : (PER % 8 == 0 ? 8 : (PER % 4 == 0 ? 4 : 1));
  static_assert(TP * Q % MLP_THREADS == 0, "tile must divide over the threads");
#pragma unroll 1
  for (int e0 = 0; e0 < PER; e0 += BATCH) {
    f32x4 v[BATCH];
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int e = tid + (e0 + u) * MLP_THREADS;
      const int r = e / Q, c4 = e - r * Q;
      v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (row0 + r < nrows) v[u] = *reinterpret_cast<const f32x4 *>(x + (size_t)(row0 + r) * C0 + c4 * 4);
    }
#pragma unroll
    for (int u = 0; u < BATCH; ++u) {
      const int e = tid + (e0 + u) * MLP_THREADS;
      const int r = e / Q, c4 = e - r * Q;
      store_split4(ys + r * XS + c4 * 4, plane, v[u]);
    }
  }
}

// [ three_interpolate(known, idx, w) (CK) | skip (CS) ],  w = (1/(sqrt(d2)+1e-8)) / sum
// (pointnet2_utils.py:138-140, pointnet2_modules.py:395-399; interpolate_gpu.cu:103-104 sum order)
template <int CK, int CS, int TP, int XS>
__device__ __forceinline__ void load_fp_split(__bf16 *__restrict__ ys, int plane, const PointSplitArgs &a, int row0, int tid) {
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
      const f32x4 v = f0[u] * w0[u] + f1[u] * w1[u] + f2[u] * w2[u];
      store_split4(ys + r * XS + c4 * 4, plane, v);
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
      store_split4(ys + r * XS + CK + c4 * 4, plane, v[u]);
    }
  }
}

// `first` arrives with the layer's first chunk already requested (stream_prefetch); `pre()` requests the next layer's
template <int TP, int CINP, int COUTP, int XS, int CHREG, bool BLOCK2, bool HAVE_FIRST, class Retire, class Epi, class Pre>
__device__ __forceinline__ void run_layer_stream_split_t(const void *__restrict__ wq, const float *__restrict__ bias,
                                                         const __bf16 *__restrict__ xs, int xplane, int wave, int lane,
                                                         StreamChunk<TP, CINP, COUTP, CHREG> &first, Retire &&retire, Epi &&epi,
                                                         Pre &&pre) {
  using C = StreamCfg<TP, CINP, COUTP, CHREG>;
  constexpr int WN = C::WN, NTW = C::NTW, MTW = C::MTW, KBC = C::KBC, NCH = C::NCH;
  static_assert(!BLOCK2 || (WN % 2 == 0 && NCH % 4 == 0), "halves must fall on wave and (even) chunk boundaries");
  const int wm = wave / WN, wn = wave % WN;
  asm volatile("" : "+s"(wq), "+s"(bias));   // keep this layer's own loads below the barrier that precedes it
  if (C::NACTIVE < 4 && wave >= C::NACTIVE) {   // fewer tiles than waves: the others only keep the barrier count
    pre();
    retire();
    return;
  }
  // two accumulators per tile (mlp_split.h mfma6_2acc): the large products round once per 16-k block, the small ones
  // apart; the bias joins them in the epilogue (the accumulators start from zero)
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
  const int c0 = BLOCK2 ? (wn >= WN / 2 ? NCH / 2 : 0) : 0, c1 = BLOCK2 ? c0 + NCH / 2 : NCH;   // c0 even: buffer parity = c & 1
  if constexpr (!HAVE_FIRST) first.load(wq, c0, wave, lane);
  f32x4 bt[NTW][4];
#pragma unroll
  for (int i = 0; i < NTW; ++i) load_bias_t4(bias, wn * NTW + i, lane, bt[i]);
  // Unrolled in full: with the two-step loop the compiler could not count the loads in flight across the loop edge and
  // waited for vmcnt(0) before every second chunk -- i.e. for the chunk it had just requested -- so half of the
  // double-buffered loads overlapped nothing.  Straight-line code gets the exact vmcnt(#loads of the next chunk).
  constexpr int NCC = BLOCK2 ? NCH / 2 : NCH;
  (void)c1;
#pragma unroll
  for (int cc = 0; cc < NCC; ++cc) {
    const int c = c0 + cc;                 // c0 is even: buffer parity = cc & 1 (compile time)
    if (cc + 1 < NCC) {
      if (cc & 1) first.load(wq, c + 1, wave, lane); else other.load(wq, c + 1, wave, lane);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (cc & 1) other.pin(); else first.pin();
#pragma unroll
    for (int mi = 0; mi < MTW; ++mi) {
      const int mt = wm * MTW + mi;
      const __bf16 *xr = xs + (mt * 32 + (lane & 31)) * XS + 8 * (lane >> 5) + c * KBC * 16;
      bf16x8 a[KBC][3];
#pragma unroll
      for (int kb = 0; kb < KBC; ++kb)
#pragma unroll
        for (int p = 0; p < 3; ++p) a[kb][p] = *reinterpret_cast<const bf16x8 *>(xr + p * xplane + kb * 16);
#pragma unroll
      for (int kb = 0; kb < KBC; ++kb)
#pragma unroll
        for (int i = 0; i < NTW; ++i)
          if (cc & 1) mfma6_2acc<true>(a[kb], other.v[i][kb], acc[mi][i], accl[mi][i]);
          else mfma6_2acc<true>(a[kb], first.v[i][kb], acc[mi][i], accl[mi][i]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  // the bias waits in AGPRs next to the weight chunks (its only use is here): with two workgroups per CU a wave has
  // 128 + 128 registers and the accumulators, the activation fragments and the addresses fill the VGPR half
#pragma unroll
  for (int i = 0; i < NTW; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) asm volatile("" : "+a"(bt[i][g]));
  pre();
  retire();
#pragma unroll
  for (int mi = 0; mi < MTW; ++mi)
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][i][r] = (acc[mi][i][r] + accl[mi][i][r]) + bt[i][r >> 2][r & 3];
      epi(wm * MTW + mi, wn * NTW + i, acc[mi][i]);
    }
}

enum { PLOAD_PLAIN = 0, PLOAD_FP = 1 };

// NL layers: C0 -> C1 (relu) [-> C2 (relu) [-> C3]]; the last layer goes to global (RELU_OUT), the others back to LDS.
// CxP: padded output widths (multiples of 32 dividing over the waves); CxL: channels kept for the next layer (% 32).
// WPC workgroups per CU: 2 needs TP = 32 and K <= 256 (50 KB of LDS, <= 256 registers per lane); the stalls of one workgroup
// (row fetch, the first weight chunk of a layer, epilogues, barriers -- 60 % of a lone workgroup's time) then lie under
// the other's MFMAs
template <int LOADER, int NL, int C0, int C1P, int C1L, int C2P, int C2L, int C3P, int TP, bool RELU_OUT, int WPC = 1, bool BD2 = false>
__global__ __launch_bounds__(MLP_THREADS, WPC) void point_mlp_split_kernel(PointSplitArgs a, int ntiles) {
  constexpr int CHREG = WPC >= 2 ? 48 : 96;
  // next layer's first chunk requested ahead (see run_layer_stream_split_t): not in the two-workgroup shapes (128 + 128
  // registers: a chunk held across the row fetch or an epilogue spills ~200 registers) nor where a layer is 384 wide
  constexpr bool PF = WPC == 1 && C2P <= 256 && C3P <= 256;
  constexpr int CMAX = cmax(C0, cmax(NL > 1 ? C1L : 0, NL > 2 ? C2L : 0));
  constexpr int XS = sp_stride(CMAX), PL = TP * XS;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16 *buf = reinterpret_cast<__bf16 *>(smem_raw);    // 3 x [TP][CMAX+8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto barrier = []() { __syncthreads(); };
  auto none = []() {};
  GN_PROBE_INIT();
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    GN_PROBE(7);
    const int row0 = tile * TP;
    const int nrows = min(TP, a.nrows - row0);
    StreamChunk<TP, C0, C1P, CHREG> ch1;
    if constexpr (PF) stream_prefetch<TP, C0, C1P, CHREG, false>(ch1, a.w[0], wave, lane);      // lands under the row fetch
    if constexpr (LOADER == PLOAD_FP) load_fp_split<C0 / 2, C0 / 2, TP, XS>(buf, PL, a, row0, tid);
    else load_plain_split<C0, TP, XS>(buf, PL, a.x, row0, a.nrows, tid);
    GN_PROBE(0);
    __syncthreads();
    GN_PROBE(1);
    auto to_lds = [&](int ncols) {
      return [&, ncols](int mt, int nt, const f32x16 &acc) {
        if (nt * 32 < ncols) {     // channels beyond the kept width are never read
          StoreSplit e{buf, XS, PL, lane};
          e(mt, nt, acc);
        }
      };
    };
    auto to_global = [&](int mt, int nt, const f32x16 &acc) {
      const int row = mt * 32 + (lane & 31);
      if (row < nrows) {
        float *o = a.out + (size_t)(row0 + row) * a.ldo;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int ch = nt * 32 + 8 * g + 4 * (lane >> 5) + j;
            const float v = RELU_OUT ? fmaxf(acc[4 * g + j], 0.f) : acc[4 * g + j];
            if (ch < a.ncols_out) o[ch] = v;
          }
      }
    };
    auto nopre = []() {};
    if constexpr (NL == 1) {
      run_layer_stream_split_t<TP, C0, C1P, XS, CHREG, false, PF>(a.w[0], a.b[0], buf, PL, wave, lane, ch1, none, to_global, nopre);
    } else {
      if constexpr (NL == 2) {
        StreamChunk<TP, C1L, C2P, CHREG> ch2;
        run_layer_stream_split_t<TP, C0, C1P, XS, CHREG, false, PF>(a.w[0], a.b[0], buf, PL, wave, lane, ch1, barrier, to_lds(C1L),
                                                               [&]() { if constexpr (PF) stream_prefetch<TP, C1L, C2P, CHREG, BD2>(ch2, a.w[1], wave, lane); });
        GN_PROBE(2);
        __syncthreads();
        GN_PROBE(3);
        run_layer_stream_split_t<TP, C1L, C2P, XS, CHREG, BD2, PF>(a.w[1], a.b[1], buf, PL, wave, lane, ch2, none, to_global, nopre);
      } else {
        StreamChunk<TP, C1L, C2P, CHREG> ch2;
        StreamChunk<TP, C2L, C3P, CHREG> ch3;
        run_layer_stream_split_t<TP, C0, C1P, XS, CHREG, false, PF>(a.w[0], a.b[0], buf, PL, wave, lane, ch1, barrier, to_lds(C1L),
                                                               [&]() { if constexpr (PF) stream_prefetch<TP, C1L, C2P, CHREG, BD2>(ch2, a.w[1], wave, lane); });
        GN_PROBE(2);
        __syncthreads();
        GN_PROBE(3);
        run_layer_stream_split_t<TP, C1L, C2P, XS, CHREG, BD2, PF>(a.w[1], a.b[1], buf, PL, wave, lane, ch2, barrier, to_lds(C2L),
                                                             [&]() { if constexpr (PF) stream_prefetch<TP, C2L, C3P, CHREG, false>(ch3, a.w[2], wave, lane); });
        GN_PROBE(4);
        __syncthreads();
        GN_PROBE(5);
        run_layer_stream_split_t<TP, C2L, C3P, XS, CHREG, false, PF>(a.w[2], a.b[2], buf, PL, wave, lane, ch3, none, to_global, nopre);
        GN_PROBE(6);
      }
    }
    __syncthreads();   // the next tile's loader overwrites the buffer
    GN_PROBE_COUNT(9, 1);
  }
  GN_PROBE_FLUSH();
}

template <int LOADER, int NL, int C0, int C1P, int C1L, int C2P, int C2L, int C3P, int TP, bool RELU_OUT, int WPC = 1, bool BD2 = false>
static int launch_point_split(const PointSplitArgs &a, hipStream_t s, const char *name) {
  constexpr int CMAX = cmax(C0, cmax(NL > 1 ? C1L : 0, NL > 2 ? C2L : 0));
  constexpr size_t lds = 2 * (size_t)3 * TP * sp_stride(CMAX);
  static_assert(lds * WPC <= 160 * 1024, "LDS budget");
  auto k = point_mlp_split_kernel<LOADER, NL, C0, C1P, C1L, C2P, C2L, C3P, TP, RELU_OUT, WPC, BD2>;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
      hipSuccess) {
    set_error("%s: cannot reserve %zu bytes of LDS", name, lds);
    return GN_ERR_LAUNCH;
  }
  const int ntiles = (a.nrows + TP - 1) / TP;
  hipLaunchKernelGGL(k, dim3(ntiles), dim3(MLP_THREADS), lds, s, a, ntiles);
  GN_CHECK_LAUNCH(name);
  return GN_OK;
}

}  // namespace gn

using namespace gn;

// 32-row tiles, two workgroups per CU (WPC 2) against 64-row tiles, one per CU: measured in isolation at batch 8
// (profiles/point_chain_bench.py) the heads chain gets SLOWER (69.9 -> 82.2 us: 1024 tiles each streaming the 885 KB of
// split weights = twice the L2 -> CU traffic, and at 32 rows per weight fragment the vector memory path, 64 B/clk per CU,
// is as long as the MFMAs) while ApproachNet gains a little (52.9 -> 49.6 us: its 128 64-row tiles fill half the chip,
// 256 32-row tiles all of it).  Default: 2 for ApproachNet only.  GN_POINT_WPC=1 / 2 forces one shape for both (read
// per call so that a test can A/B them).
namespace gn {
static thread_local int g_point_tile_policy = 0;
int point_tile_policy() {
  const char *e = getenv("GN_POINT_TILES");      // read per call: tests and profiles A/B the shapes in one process
  if (e) return (e[0] == 't' || e[0] == '1') ? 1 : 0;
  return g_point_tile_policy;
}
}  // namespace gn

extern "C" int gn_set_point_tile_policy(int policy) {
  const int prev = gn::g_point_tile_policy;
  gn::g_point_tile_policy = policy ? 1 : 0;
  return prev;
}

// GN_POINT_WIDE=0 / 1 forces the plane / fp32-tile form of every chain (tests); otherwise the policy decides
static int wide_forced() {
  const char *we = getenv("GN_POINT_WIDE");
  return we ? (atoi(we) != 0 ? 1 : 0) : -1;
}

static int point_wpc(int dflt) {
  const char *e = getenv("GN_POINT_WPC");
  return e ? atoi(e) : dflt;
}

extern "C" int gn_fp_interp_mlp_split(const float *known_feat, const float *skip_feat, const float *dist2, const int32_t *idx3,
                                      int b, int n, int m, int ck, int cs, const int *mlp, const void *const *wpacked,
                                      const float *const *bias, float *out, gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n > 0 && m > 0, "fp_interp_mlp_split: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(known_feat && skip_feat && dist2 && idx3 && mlp && wpacked && bias && out, "fp_interp_mlp_split: null pointer");
  if (!(ck == 256 && cs == 256 && mlp[0] == 256 && mlp[1] == 256)) {
    set_error("fp_interp_mlp_split: unsupported configuration ck=%d cs=%d mlp=[%d,%d]", ck, cs, mlp[0], mlp[1]);
    return GN_ERR_INVALID_ARG;
  }
  PointSplitArgs a{};
  a.w[0] = wpacked[0]; a.b[0] = bias[0]; a.w[1] = wpacked[1]; a.b[1] = bias[1];
  a.known = known_feat; a.skip = skip_feat; a.dist2 = dist2; a.idx3 = idx3; a.n = n; a.m = m;
  a.out = out; a.ldo = 256; a.ncols_out = 256; a.nrows = b * n;
  {
    const int wf = wide_forced();
    if (wf == 1 || (wf < 0 && point_tile_policy() == 1)) return fp_mlp_wide(a, as_stream(stream));
  }
  return launch_point_split<PLOAD_FP, 2, 512, 256, 256, 256, 0, 0, 32, true>(a, as_stream(stream), "fp_interp_mlp_split");
}

extern "C" int gn_approach_mlp_split(const float *x, int rows, int cin, int nout, const void *const *wpacked,
                                     const float *const *bias, float *out, gn_stream_t stream) {
  GN_REQUIRE(rows >= 0, "approach_mlp_split: bad sizes");
  if (rows == 0) return GN_OK;
  GN_REQUIRE(x && wpacked && bias && out, "approach_mlp_split: null pointer");
  if (!(cin == 256 && nout >= 3 && nout <= 320)) {
    set_error("approach_mlp_split: unsupported configuration cin=%d nout=%d (need 256 and 3 <= 2+num_view <= 320)", cin, nout);
    return GN_ERR_INVALID_ARG;
  }
  PointSplitArgs a{};
  for (int i = 0; i < 3; ++i) { a.w[i] = wpacked[i]; a.b[i] = bias[i]; }
  a.x = x; a.out = out; a.ldo = nout; a.ncols_out = nout; a.nrows = rows;
  {
    const int wf = wide_forced();
    // 64-row fp32 tiles under either policy: measured at batch 8 (profiles/point_chain_bench.py) 46.7 us on 128 CUs against
    // 49.0 us on 256 for the 32-row plane tiles; a launch too small to give 64 CUs a tile keeps the 32-row form under the
    // latency policy
    if (wf == 1 || (wf < 0 && !getenv("GN_POINT_WPC") && (point_tile_policy() == 1 || rows >= 64 * 64)))
      return approach_mlp_wide(a, as_stream(stream));
  }
  if (point_wpc(2) >= 2)
    return launch_point_split<PLOAD_PLAIN, 3, 256, 256, 256, 384, 320, 384, 32, false, 2>(a, as_stream(stream), "approach_mlp_split");
  return launch_point_split<PLOAD_PLAIN, 3, 256, 256, 256, 384, 320, 384, 64, false>(a, as_stream(stream), "approach_mlp_split");
}

extern "C" int gn_heads_mlp_split(const float *x, int rows, int cin, int nout, const void *const *wpacked,
                                  const float *const *bias, float *out, gn_stream_t stream) {
  GN_REQUIRE(rows >= 0, "heads_mlp_split: bad sizes");
  if (rows == 0) return GN_OK;
  GN_REQUIRE(x && wpacked && bias && out, "heads_mlp_split: null pointer");
  if (!(cin == 256 && nout == 48)) {
    set_error("heads_mlp_split: unsupported configuration cin=%d nout=%d (need 256, 48)", cin, nout);
    return GN_ERR_INVALID_ARG;
  }
  PointSplitArgs a{};
  for (int i = 0; i < 3; ++i) { a.w[i] = wpacked[i]; a.b[i] = bias[i]; }
  a.x = x; a.out = out; a.ldo = 48; a.ncols_out = 48; a.nrows = rows;
  // layer 2 is block-diagonal (two 128 -> 128 maps): each wave streams and multiplies its own half of K only.
  // GN_HEADS_DENSE=1 multiplies the zero blocks too (the round-1 form; A/B in tests/test_split_gpu.py)
  const char *de = getenv("GN_HEADS_DENSE");
  const bool dense = de && atoi(de) != 0;
  // enough rows to give (nearly) every CU a 128-row tile (or the throughput policy): the wide form streams a layer's weights
  // once per 128 rows instead of once per 64 (mlp_point_wide.hip; bit-identical outputs: tests/test_split_gpu.py)
  {
    const int wf = wide_forced();
    if (wf == 1 || (wf < 0 && !dense && !getenv("GN_POINT_WPC") && (rows >= 128 * 192 || point_tile_policy() == 1)))
      return heads_mlp_wide(a, as_stream(stream));
  }
  if (point_wpc(1) >= 2) {
    if (dense) return launch_point_split<PLOAD_PLAIN, 3, 256, 256, 256, 256, 256, 64, 32, false, 2>(a, as_stream(stream), "heads_mlp_split");
    return launch_point_split<PLOAD_PLAIN, 3, 256, 256, 256, 256, 256, 64, 32, false, 2, true>(a, as_stream(stream), "heads_mlp_split");
  }
  if (dense) return launch_point_split<PLOAD_PLAIN, 3, 256, 256, 256, 256, 256, 64, 64, false>(a, as_stream(stream), "heads_mlp_split");
  return launch_point_split<PLOAD_PLAIN, 3, 256, 256, 256, 256, 256, 64, 64, false, 1, true>(a, as_stream(stream), "heads_mlp_split");
}

extern "C" int gn_point_linear_split(const float *x, int rows, int cin, int cout, const void *wpacked, const float *bias,
                                     float *out, gn_stream_t stream) {
  GN_REQUIRE(rows >= 0, "point_linear_split: bad sizes");
  if (rows == 0) return GN_OK;
  GN_REQUIRE(x && wpacked && bias && out, "point_linear_split: null pointer");
  PointSplitArgs a{};
  a.w[0] = wpacked; a.b[0] = bias; a.x = x; a.out = out; a.ldo = cout; a.ncols_out = cout; a.nrows = rows;
  // one layer, ~3 us of MFMA + HBM work per launch: with ONE 64-row workgroup per CU (256 / 128 / 64 tiles at batch 8) a
  // CU sits through its tile's row fetch, split, weight stream and store one after the other; 32-row tiles, two
  // workgroups per CU (the default; GN_LINEAR_WPC=1: the 64-row form) let one tile's latencies lie under the other's work
  const char *le = getenv("GN_LINEAR_WPC");      // read per call: profiles/point_chain_bench.py A/Bs the shapes in one process
  const int lin_wpc = le ? atoi(le) : 2;          // measured at batch 8: 12.7 / 11.3 / 10.7 -> 10.2 / 10.8 / 9.8 us, outputs identical
  if (cin == 128 && cout == 128) {
    if (lin_wpc >= 2)
      return launch_point_split<PLOAD_PLAIN, 1, 128, 128, 0, 0, 0, 0, 32, false, 2>(a, as_stream(stream), "point_linear_split");
    return launch_point_split<PLOAD_PLAIN, 1, 128, 128, 0, 0, 0, 0, 64, false>(a, as_stream(stream), "point_linear_split");
  }
  if (cin == 256 && cout == 128) {
    if (lin_wpc >= 2)
      return launch_point_split<PLOAD_PLAIN, 1, 256, 128, 0, 0, 0, 0, 32, false, 2>(a, as_stream(stream), "point_linear_split");
    return launch_point_split<PLOAD_PLAIN, 1, 256, 128, 0, 0, 0, 0, 64, false>(a, as_stream(stream), "point_linear_split");
  }
  set_error("point_linear_split: unsupported configuration cin=%d cout=%d", cin, cout);
  return GN_ERR_INVALID_ARG;
}

GN_PROBE_READER(gn_phase_probe_read_point)
