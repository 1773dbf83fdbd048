// Micro-benchmark of one resident-weight layer (csrc/mlp_core.h) on one LDS tile: cycles per 32x32x2 fp32 MFMA
// for the bare chain, with the plain epilogues and with the software-pipelined pooled layer.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../../include -I../../graspnet-baseline_amd/csrc mfma_layer_bench.hip ../../graspnet-baseline_amd/csrc/gn_common.cpp
#include <stdio.h>
#include <vector>
#include "mlp_core.h"
using namespace gn;

struct NoEpi {
  float sink;
  __device__ __forceinline__ void operator()(int, int, int, const f32x16 &acc) { sink += acc[0] + acc[15]; }
};

template <int MODE, int NWV, int CIN, int COUT>
__global__ __launch_bounds__(NWV * 64) void bench(const float *wp, const float *bp, float *out, unsigned long long *cyc,
                                                  int reps, int mtc) {
  constexpr int TP = 128;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *x = smem;                        // [TP][CIN+4]
  float *y = x + TP * lds_stride(CIN);    // [TP][COUT+4] (only for the store epilogue)
  float *gate = y + (MODE == 1 ? TP * lds_stride(COUT) : 64);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < TP * lds_stride(CIN); i += NWV * 64) x[i] = (float)((i * 7) % 13) * 0.01f;
  for (int i = tid; i < TP * 4; i += NWV * 64) gate[i] = (i % 3) ? 0.f : -INFINITY;
  LayerW<CIN, COUT, NWV> w;
  w.load(wp, bp, wave, lane);
  __syncthreads();
  using LW = LayerW<CIN, COUT, NWV>;
  PoolMasked<4, LW::NTW> pool{gate, lane, {}};
  pool.reset();
  NoEpi ne{0.f};
  StoreLds<true> se{y, lds_stride(COUT), COUT, lane};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
    if (MODE == 0) run_layer<TP, CIN, COUT, NoEpi, NWV>(w, x, wave, lane, ne, mtc);
    if (MODE == 1) run_layer<TP, CIN, COUT, StoreLds<true>, NWV>(w, x, wave, lane, se, mtc);
    if (MODE == 2) run_layer<TP, CIN, COUT, PoolMasked<4, LW::NTW>, NWV>(w, x, wave, lane, pool, mtc);
    if constexpr (MODE == 3) run_layer_pool_pipe<TP, CIN, COUT, 4, NWV>(w, x, wave, lane, pool, mtc);
    if (MODE == 1) __syncthreads();
    if (MODE == 5) { run_layer<TP, CIN, COUT, PoolMasked<4, LW::NTW>, NWV>(w, x, wave, lane, pool, mtc); __syncthreads(); }
    if (MODE == 6) { run_layer<TP, CIN, COUT, PoolMasked<4, LW::NTW>, NWV>(w, x, wave, lane, pool, mtc); __syncthreads(); if (wave >= 4) __builtin_amdgcn_s_sleep(8); }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float acc = ne.sink;
  for (int d = 0; d < 4; ++d)
    for (int i = 0; i < LW::NTW; ++i) acc += pool.hold[d][i];
  out[blockIdx.x * NWV * 64 + tid] = acc + y[tid];
  if (lane == 0) cyc[blockIdx.x * NWV + wave] = t1 - t0;
}

template <int MODE, int NWV, int CIN = 128, int COUT = 256>
void run(const char *name, const float *wp, const float *bp, float *out, unsigned long long *cyc, int blocks, int mtc) {
  const int reps = 200;
  const size_t lds = sizeof(float) * (128 * lds_stride(CIN) + (MODE == 1 ? 128 * lds_stride(COUT) : 64) + 128 * 4);
  auto k = bench<MODE, NWV, CIN, COUT>;
  hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  k<<<blocks, NWV * 64, lds>>>(wp, bp, out, cyc, 2, mtc);
  if (hipGetLastError() != hipSuccess) { printf("launch failed\n"); return; }
  hipEventRecord(a);
  k<<<blocks, NWV * 64, lds>>>(wp, bp, out, cyc, reps, mtc);
  hipEventRecord(b);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<unsigned long long> h(blocks * NWV);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += (double)v; avg /= h.size();
  // MFMAs issued per SIMD per rep: (NWV/4 waves) * tiles per wave * 64
  using LW = LayerW<CIN, COUT, NWV>;
  const double tiles_per_wave = (double)((mtc + LW::WM - 1) / LW::WM) * LW::NTW;   // busiest wave
  const double kmf = CIN / 2;
  const double mfma_per_simd = (NWV / 4) * tiles_per_wave * kmf * reps;
  printf("%-28s NWV=%d mtc=%d blocks=%d: %.3f ms  memtime ticks/rep %.0f  -> %.1f ticks per SIMD-MFMA, wall %.2f ns per SIMD-MFMA\n",
         name, NWV, mtc, blocks, ms, avg / reps, avg / mfma_per_simd, ms * 1e6 / mfma_per_simd);
}

int main() {
  float *wp, *bp, *out; unsigned long long *cyc;
  hipMalloc(&wp, 256 * 128 * 4); hipMalloc(&bp, 256 * 4); hipMalloc(&out, 4096 * 512 * 4); hipMalloc(&cyc, 4096 * 8 * 8);
  hipMemset(wp, 0, 256 * 128 * 4); hipMemset(bp, 0, 256 * 4);
  for (int blocks : {1}) {
    for (int mtc : {4, 3}) {
      run<0, 4>("bare chain", wp, bp, out, cyc, blocks, mtc);
      run<0, 4, 64, 128>("L2 bare chain", wp, bp, out, cyc, blocks, mtc);
      run<1, 4, 64, 128>("L2 store-to-LDS epilogue", wp, bp, out, cyc, blocks, mtc);
      run<1, 8, 64, 128>("L2 store-to-LDS epilogue", wp, bp, out, cyc, blocks, mtc);
      run<2, 4>("masked-pool epilogue", wp, bp, out, cyc, blocks, mtc);
      run<3, 4>("masked pool, pipelined", wp, bp, out, cyc, blocks, mtc);
      run<0, 8>("bare chain", wp, bp, out, cyc, blocks, mtc);
      run<2, 8>("masked-pool epilogue", wp, bp, out, cyc, blocks, mtc);
      run<5, 4>("masked pool + barrier/rep", wp, bp, out, cyc, blocks, mtc);
      run<5, 8>("masked pool + barrier/rep", wp, bp, out, cyc, blocks, mtc);
      run<6, 8>("same, waves 4-7 skewed", wp, bp, out, cyc, blocks, mtc);
    }
  }
  return 0;
}
