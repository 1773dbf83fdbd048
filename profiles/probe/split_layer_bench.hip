// Micro-benchmark of one split-bf16 layer (csrc/mlp_split.h) on one LDS tile: cycles per v_mfma_f32_32x32x16_bf16.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../include -I../../graspnet-baseline_amd/csrc split_layer_bench.hip ../../graspnet-baseline_amd/csrc/gn_common.cpp
#include <stdio.h>
#include <vector>
#include "mlp_split.h"
using namespace gn;

struct NoEpi {
  float sink;
  __device__ __forceinline__ void operator()(int, int, int, const f32x16 &acc) { sink += acc[0] + acc[15]; }
};

template <int MODE>
__global__ __launch_bounds__(256) void bench(const void *wp, float *out, unsigned long long *cyc, int reps, int mtc) {
  constexpr int TP = 96, CIN = 128, COUT = 256;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16 *x = reinterpret_cast<__bf16 *>(smem_raw);
  constexpr int PL = TP * sp_stride(CIN);
  float *gate = reinterpret_cast<float *>(x + 3 * PL);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * PL; i += 256) x[i] = (__bf16)((float)((i * 7) % 13) * 0.01f);
  for (int i = tid; i < TP * 4; i += 256) gate[i] = (i % 3) ? 0.f : -INFINITY;
  LayerWS<CIN, COUT> w;
#ifdef BENCH_PIN
  w.template load<true>(wp, wave, lane);
#else
  w.load(wp, wave, lane);
#endif
  __syncthreads();
  using LW = LayerWS<CIN, COUT>;
  float bias[LW::NTW] = {0.f, 0.f};
  PoolMasked<4, LW::NTW> pool{gate, lane, {}};
  pool.reset();
  NoEpi ne{0.f};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) run_layer_split_n<TP, CIN, COUT>(w, bias, x, PL, wave, lane, ne, mtc);
    if (MODE == 1) run_layer_split_n<TP, CIN, COUT>(w, bias, x, PL, wave, lane, pool, mtc);
    if constexpr (MODE == 2) run_layer_split_n_pool_pipe<TP, CIN, COUT, 4>(w, bias, x, PL, wave, lane, pool, mtc);
    if constexpr (MODE >= 3) run_layer_split_n_pool_pipe2<TP, CIN, COUT, 4, 4, MODE - 3>(w, bias, x, PL, wave, lane, pool, mtc);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = ne.sink;
  for (int d = 0; d < 4; ++d)
    for (int i = 0; i < LW::NTW; ++i) acc += pool.hold[d][i];
  out[blockIdx.x * 256 + tid] = acc;
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run(const char *name, const void *wp, float *out, unsigned long long *cyc, int blocks, int mtc) {
  const int reps = 200;
  const size_t lds = 2 * 3 * 96 * sp_stride(128) + 96 * 16;
  auto k = bench<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<<<blocks, 256, lds>>>(wp, out, cyc, 2, mtc);
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
  hipEventRecord(a);
  k<<<blocks, 256, lds>>>(wp, out, cyc, reps, mtc);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
  const double mfma = (double)mtc * 2 * 8 * 6 * reps;     // per wave = per SIMD
  printf("%-26s mtc=%d blocks=%d: %.3f ms  %.1f cycles per MFMA (ideal 32), wall %.2f ns\n", name, mtc, blocks, ms, avg / mfma, ms * 1e6 / mfma);
}


template <int MODE>
__global__ __launch_bounds__(256) void bench_t(const void *wp, float *out, unsigned long long *cyc, int reps, int mtc) {
  constexpr int TP = 96, CIN = 64, COUT = 128;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  __bf16 *x = reinterpret_cast<__bf16 *>(smem_raw);
  constexpr int PL = TP * sp_stride(CIN), PO = TP * sp_stride(COUT);
  __bf16 *y = x + 3 * PL;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 3 * PL; i += 256) x[i] = (__bf16)((float)((i * 7) % 13) * 0.01f);
  LayerWS<CIN, COUT> w;
#ifdef BENCH_PIN
  w.template load<true>(wp, wave, lane);
#else
  w.load(wp, wave, lane);
#endif
  __syncthreads();
  float bt[16];
  for (int r = 0; r < 16; ++r) bt[r] = 0.01f * r;
  StoreSplit e{y, sp_stride(COUT), PO, lane};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) run_layer_split_t<TP, CIN, COUT>(w, bt, x, PL, wave, lane, e, mtc);
    if constexpr (MODE == 1) run_layer_split_t_pipe<TP, CIN, COUT>(w, bt, x, PL, wave, lane, e, mtc);
    if constexpr (MODE >= 2) run_layer_split_t_pipe<TP, CIN, COUT, 4, MODE - 1>(w, bt, x, PL, wave, lane, e, mtc);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 256 + tid] = (float)y[tid];
  if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int MODE>
void run_t(const char *name, const void *wp, float *out, unsigned long long *cyc, int mtc) {
  const int reps = 200;
  const size_t lds = 2 * 3 * 96 * (sp_stride(64) + sp_stride(128));
  auto k = bench_t<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  k<<<1, 256, lds>>>(wp, out, cyc, 2, mtc);
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
  k<<<1, 256, lds>>>(wp, out, cyc, reps, mtc);
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  const double mfma = (double)mtc * 4 * 6 * reps;
  printf("%-26s mtc=%d: %.1f cycles per MFMA (ideal 32), %.0f cycles per M tile\n", name, mtc, (double)h[0] / mfma, (double)h[0] / (mtc * reps));
}

int main() {
  void *wp; float *out; unsigned long long *cyc;
  hipMalloc(&wp, 256 * 128 * 6); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&cyc, 4096 * 4 * 8);
  hipMemset(wp, 0, 256 * 128 * 6);
  for (int blocks : {256})
    for (int mtc : {3, 2}) {
      run<0>("split N, bare", wp, out, cyc, blocks, mtc);
      run<1>("split N + masked pool", wp, out, cyc, blocks, mtc);
      run<2>("split N + pool, pipelined", wp, out, cyc, blocks, mtc);
      run<3>("pipe2 (shared A frags)", wp, out, cyc, blocks, mtc);
      run<4>("pipe2, no gate reads", wp, out, cyc, blocks, mtc);
      run<5>("pipe2, gates, no pool math", wp, out, cyc, blocks, mtc);
      run<6>("pipe2, neither", wp, out, cyc, blocks, mtc);
    }
  for (int mtc : {3, 2}) {
    run_t<0>("split T 64->128 + store", wp, out, cyc, mtc); run_t<1>("split T, pipelined", wp, out, cyc, mtc);
    run_t<2>("T pipe, no LDS stores", wp, out, cyc, mtc); run_t<3>("T pipe, no split math", wp, out, cyc, mtc);
    run_t<4>("T pipe, neither", wp, out, cyc, mtc);
  }
  return 0;
}
