// mlp_point.hip -- fused per-point MLP chains (no grouping): feature propagation, ApproachNet,
// OperationNet + ToleranceNet.
//
//   fp      : PointnetFPModule.forward (pointnet2/pointnet2_modules.py:394-416): inverse-distance
//             3-NN interpolation + skip concat + SharedMLP[512,256,256]
//   approach: ApproachNet.forward conv1/bn1/relu, conv2/bn2/relu, conv3 (models/modules.py:55-57)
//   heads   : OperationNet.forward + ToleranceNet.forward conv chains (models/modules.py:171-175,
//             216-220), evaluated as ONE 256->256->256->48 chain: conv1 of both nets stacked,
//             conv2/conv3 block-diagonal (built on the host, graspnet_amd/weights.py)
// Activations are position-major tiles in LDS, weights are streamed from L2 in K chunks as packed
// MFMA fragments (mlp_core.h); every intermediate stays on chip.
#include "mlp_core.h"

namespace gn {

struct PointMlpArgs {
  const float *w1, *b1, *w2, *b2, *w3, *b3;
  // plain loader
  const float *x;       // (rows, C0) channels-last
  // fp loader
  const float *known;   // (B, m, CK) channels-last features to interpolate
  const float *skip;    // (B, n, CS) channels-last skip features
  const float *dist2;   // (B, n, 3) squared distances from three_nn
  const int32_t *idx3;  // (B, n, 3)
  int n, m;
  float *out;           // (rows, ldo)
  int ldo, ncols_out;
  int nrows;            // total rows (B*n)
};

// rows of a plain channels-last tensor; rows beyond nrows are zero-filled
template <int C0, int TP>
__device__ __forceinline__ void load_plain_tile(float *__restrict__ xs, const float *__restrict__ x, int row0,
                                                int nrows, int tid) {
  constexpr int XS = lds_stride(C0), Q = C0 / 4;
  for (int e = tid; e < TP * Q; e += MLP_THREADS) {
    const int r = e / Q, c4 = e - r * Q;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + r < nrows) v = *reinterpret_cast<const f32x4 *>(x + (size_t)(row0 + r) * C0 + c4 * 4);
    *reinterpret_cast<f32x4 *>(xs + r * XS + c4 * 4) = v;
  }
}

// [ three_interpolate(known, idx, w) (CK) | skip (CS) ] with
// w = (1/(sqrt(d2)+1e-8)) / sum   (pointnet2_utils.py:138-140, pointnet2_modules.py:395-399)
template <int CK, int CS, int TP>
__device__ __forceinline__ void load_fp_tile(float *__restrict__ xs, const PointMlpArgs &a, int row0, int tid) {
  constexpr int XS = lds_stride(CK + CS), QK = CK / 4, QS = CS / 4;
  for (int e = tid; e < TP * QK; e += MLP_THREADS) {
    const int r = e / QK, c4 = e - r * QK;
    const int row = row0 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < a.nrows) {
      const int b = row / a.n;
      const float *d2 = a.dist2 + (size_t)row * 3;
      const int32_t *ix = a.idx3 + (size_t)row * 3;
      const float r0 = 1.0f / (sqrtf(d2[0]) + 1e-8f);
      const float r1 = 1.0f / (sqrtf(d2[1]) + 1e-8f);
      const float r2 = 1.0f / (sqrtf(d2[2]) + 1e-8f);
      const float norm = r0 + r1 + r2;
      const float w0 = r0 / norm, w1 = r1 / norm, w2 = r2 / norm;
      const float *kb = a.known + (size_t)b * a.m * CK + c4 * 4;
      const f32x4 f0 = *reinterpret_cast<const f32x4 *>(kb + (size_t)ix[0] * CK);
      const f32x4 f1 = *reinterpret_cast<const f32x4 *>(kb + (size_t)ix[1] * CK);
      const f32x4 f2 = *reinterpret_cast<const f32x4 *>(kb + (size_t)ix[2] * CK);
      v = f0 * w0 + f1 * w1 + f2 * w2;   // interpolate_gpu.cu:103-104 sum order
    }
    *reinterpret_cast<f32x4 *>(xs + r * XS + c4 * 4) = v;
  }
  for (int e = tid; e < TP * QS; e += MLP_THREADS) {
    const int r = e / QS, c4 = e - r * QS;
    const int row = row0 + r;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row < a.nrows) v = *reinterpret_cast<const f32x4 *>(a.skip + (size_t)row * CS + c4 * 4);
    *reinterpret_cast<f32x4 *>(xs + r * XS + CK + c4 * 4) = v;
  }
}

enum { LOAD_PLAIN = 0, LOAD_FP = 1 };

// C0 -> C1 (relu) -> C2 (relu iff 3 layers; else output, relu per RELU_OUT) [-> C3 output]
// CxP are padded sizes: C0P % 8 == 0; C1P, C2P, C3P multiples of 32 dividing over the waves;
// C1L / C2L = number of channels of layer 1 / 2 kept in LDS for the next layer (% 8 == 0).
template <int LOADER, int C0P, int C1P, int C1L, int C2P, int C2L, int C3P, int KC1, int KC2, int KC3, int TP,
          bool RELU_OUT>
__global__ __launch_bounds__(MLP_THREADS) void point_mlp_kernel(PointMlpArgs a, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int A_FLOATS = cmax(TP * lds_stride(C0P), C3P > 0 ? TP * lds_stride(C2L) : 0);
  float *bufA = smem;
  float *bufB = smem + A_FLOATS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int row0 = tile * TP;
    if constexpr (LOADER == LOAD_FP) load_fp_tile<C0P / 2, C0P / 2, TP>(bufA, a, row0, tid);
    else load_plain_tile<C0P, TP>(bufA, a.x, row0, a.nrows, tid);
    __syncthreads();
    StoreLds<true> e1{bufB, lds_stride(C1L), C1L, lane};
    run_layer_stream<TP, C0P, C1P, KC1>(a.w1, a.b1, bufA, wave, lane, e1);
    __syncthreads();
    const int nrows = min(TP, a.nrows - row0);
    if constexpr (C3P == 0) {
      StoreGlobal<RELU_OUT> e2{a.out + (size_t)row0 * a.ldo, a.ldo, a.ncols_out, nrows, lane};
      run_layer_stream<TP, C1L, C2P, KC2>(a.w2, a.b2, bufB, wave, lane, e2);
    } else {
      StoreLds<true> e2{bufA, lds_stride(C2L), C2L, lane};
      run_layer_stream<TP, C1L, C2P, KC2>(a.w2, a.b2, bufB, wave, lane, e2);
      __syncthreads();
      StoreGlobal<RELU_OUT> e3{a.out + (size_t)row0 * a.ldo, a.ldo, a.ncols_out, nrows, lane};
      run_layer_stream<TP, C2L, C3P, KC3>(a.w3, a.b3, bufA, wave, lane, e3);
    }
    __syncthreads();
  }
}

// one linear layer per point, no activation: out (rows, COUT) = x (rows, CIN) * W^T + b
template <int CIN, int COUT, int KC, int TP>
__global__ __launch_bounds__(MLP_THREADS) void point_linear_kernel(const float *__restrict__ x,
                                                                   const float *__restrict__ wp,
                                                                   const float *__restrict__ bp, int nrows,
                                                                   float *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row0 = blockIdx.x * TP;
  load_plain_tile<CIN, TP>(smem, x, row0, nrows, tid);
  __syncthreads();
  StoreGlobal<false> e{out + (size_t)row0 * COUT, COUT, COUT, min(TP, nrows - row0), lane};
  run_layer_stream<TP, CIN, COUT, KC>(wp, bp, smem, wave, lane, e);
}

template <int LOADER, int C0P, int C1P, int C1L, int C2P, int C2L, int C3P, int KC1, int KC2, int KC3, int TP,
          bool RELU_OUT>
static int launch_point_mlp(const PointMlpArgs &a, hipStream_t s, const char *name) {
  constexpr int A_FLOATS = cmax(TP * lds_stride(C0P), C3P > 0 ? TP * lds_stride(C2L) : 0);
  constexpr size_t lds = sizeof(float) * (A_FLOATS + TP * lds_stride(C1L));
  static_assert(lds <= 160 * 1024, "LDS budget");
  auto k = point_mlp_kernel<LOADER, C0P, C1P, C1L, C2P, C2L, C3P, KC1, KC2, KC3, TP, RELU_OUT>;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
      hipSuccess) {
    set_error("%s: cannot reserve %zu bytes of LDS", name, lds);
    return GN_ERR_LAUNCH;
  }
  const int ntiles = (a.nrows + TP - 1) / TP;
  hipLaunchKernelGGL(k, dim3(ntiles), dim3(MLP_THREADS), lds, s, a, ntiles);   // weights stream per tile anyway
  GN_CHECK_LAUNCH(name);
  return GN_OK;
}

}  // namespace gn

using namespace gn;

// PointnetFPModule tail: out (B,n,256) channels-last.
extern "C" int gn_fp_interp_mlp(const float *known_feat, const float *skip_feat, const float *dist2,
                                const int32_t *idx3, int b, int n, int m, int ck, int cs, const int *mlp,
                                const float *const *wpacked, const float *const *bias, float *out,
                                gn_stream_t stream) {
  GN_REQUIRE(b >= 0 && n > 0 && m > 0, "fp_interp_mlp: bad sizes");
  if (b == 0) return GN_OK;
  GN_REQUIRE(known_feat && skip_feat && dist2 && idx3 && mlp && wpacked && bias && out, "fp_interp_mlp: null pointer");
  if (!(ck == 256 && cs == 256 && mlp[0] == 256 && mlp[1] == 256)) {
    set_error("fp_interp_mlp: unsupported configuration ck=%d cs=%d mlp=[%d,%d]", ck, cs, mlp[0], mlp[1]);
    return GN_ERR_INVALID_ARG;
  }
  PointMlpArgs a{};
  a.w1 = wpacked[0]; a.b1 = bias[0]; a.w2 = wpacked[1]; a.b2 = bias[1];
  a.known = known_feat; a.skip = skip_feat; a.dist2 = dist2; a.idx3 = idx3; a.n = n; a.m = m;
  a.out = out; a.ldo = 256; a.ncols_out = 256; a.nrows = b * n;
  return launch_point_mlp<LOAD_FP, 512, 256, 256, 256, 256, 0, 64, 64, 0, 32, true>(a, as_stream(stream),
                                                                                       "fp_interp_mlp");
}

// ApproachNet conv chain: x (rows,256) channels-last -> out (rows,302) = [objectness(2) | view scores(300)]
extern "C" int gn_approach_mlp(const float *x, int rows, int cin, int nout, const float *const *wpacked,
                               const float *const *bias, float *out, gn_stream_t stream) {
  GN_REQUIRE(rows >= 0, "approach_mlp: bad sizes");
  if (rows == 0) return GN_OK;
  GN_REQUIRE(x && wpacked && bias && out, "approach_mlp: null pointer");
  if (!(cin == 256 && nout >= 3 && nout <= 320)) {
    set_error("approach_mlp: unsupported configuration cin=%d nout=%d (need 256 and 3 <= 2+num_view <= 320)", cin, nout);
    return GN_ERR_INVALID_ARG;
  }
  PointMlpArgs a{};
  a.w1 = wpacked[0]; a.b1 = bias[0]; a.w2 = wpacked[1]; a.b2 = bias[1]; a.w3 = wpacked[2]; a.b3 = bias[2];
  a.x = x; a.out = out; a.ldo = nout; a.ncols_out = nout; a.nrows = rows;
  return launch_point_mlp<LOAD_PLAIN, 256, 256, 256, 384, 320, 384, 64, 32, 40, 64, false>(a, as_stream(stream),
                                                                                             "approach_mlp");
}

// OperationNet + ToleranceNet: x (rows,256) -> out (rows,48) = [score(12) | angle cls(12) | width(12) | tolerance(12)]
extern "C" int gn_heads_mlp(const float *x, int rows, int cin, int nout, const float *const *wpacked,
                            const float *const *bias, float *out, gn_stream_t stream) {
  GN_REQUIRE(rows >= 0, "heads_mlp: bad sizes");
  if (rows == 0) return GN_OK;
  GN_REQUIRE(x && wpacked && bias && out, "heads_mlp: null pointer");
  if (!(cin == 256 && nout == 48)) {
    set_error("heads_mlp: unsupported configuration cin=%d nout=%d (need 256, 48)", cin, nout);
    return GN_ERR_INVALID_ARG;
  }
  PointMlpArgs a{};
  a.w1 = wpacked[0]; a.b1 = bias[0]; a.w2 = wpacked[1]; a.b2 = bias[1]; a.w3 = wpacked[2]; a.b3 = bias[2];
  a.x = x; a.out = out; a.ldo = 48; a.ncols_out = 48; a.nrows = rows;
  return launch_point_mlp<LOAD_PLAIN, 256, 256, 256, 256, 256, 64, 64, 64, 64, 64, false>(a, as_stream(stream),
                                                                                            "heads_mlp");
}

// Per-point linear map used to hoist layer 1 of a set-abstraction MLP: out (rows, cout) = x W^T + b.
extern "C" int gn_point_linear(const float *x, int rows, int cin, int cout, const float *wpacked, const float *bias,
                               float *out, gn_stream_t stream) {
  GN_REQUIRE(rows >= 0, "point_linear: bad sizes");
  if (rows == 0) return GN_OK;
  GN_REQUIRE(x && wpacked && bias && out, "point_linear: null pointer");
  constexpr int TP = 64;
  hipStream_t s = as_stream(stream);
  if (cin == 128 && cout == 128) {
    auto k = point_linear_kernel<128, 128, 64, TP>;
    constexpr size_t lds = sizeof(float) * TP * lds_stride(128);
    hipLaunchKernelGGL(k, dim3((rows + TP - 1) / TP), dim3(MLP_THREADS), lds, s, x, wpacked, bias, rows, out);
  } else if (cin == 256 && cout == 128) {
    auto k = point_linear_kernel<256, 128, 64, TP>;
    constexpr size_t lds = sizeof(float) * TP * lds_stride(256);
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) !=
        hipSuccess) {
      set_error("point_linear: cannot reserve LDS");
      return GN_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(k, dim3((rows + TP - 1) / TP), dim3(MLP_THREADS), lds, s, x, wpacked, bias, rows, out);
  } else {
    set_error("point_linear: unsupported configuration cin=%d cout=%d", cin, cout);
    return GN_ERR_INVALID_ARG;
  }
  GN_CHECK_LAUNCH("point_linear");
  return GN_OK;
}
