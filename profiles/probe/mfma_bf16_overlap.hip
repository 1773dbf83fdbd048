// v_mfma_f32_32x32x16_bf16 (XDL): cycles per MFMA of a dependent chain with NV independent VALU fillers per gap,
// one or two waves per SIMD.  Companion of mfma_valu_overlap.hip (the fp32 form, where nothing overlaps).
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int NV, int NT, int NCH = 1>
__global__ __launch_bounds__(NT) void k(float *out, unsigned long long *cyc, int reps) {
  f32x16 acc[4];
  for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)1.0f; }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
  const float one = 1.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m % NCH] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m % NCH], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[v % 8]) : "v"(one));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * NT + threadIdx.x] = acc[0][0] + acc[1][3] + acc[2][1] + acc[3][2] + s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int NV, int NT, int NCH = 1>
void run(float *out, unsigned long long *cyc) {
  const int reps = 2000;
  k<NV, NT, NCH><<<1, NT>>>(out, cyc, 10);
  hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  hipEventRecord(e0);
  k<NV, NT, NCH><<<1, NT>>>(out, cyc, reps);
  hipEventRecord(e1);
  k<NV, NT, NCH><<<256, NT>>>(out, cyc, reps);
  hipEventRecord(e2);
  hipDeviceSynchronize();
  float ms1, ms2; hipEventElapsedTime(&ms1, e0, e1); hipEventElapsedTime(&ms2, e1, e2);
  printf("   wall: 1 block %.3f ms, 256 blocks %.3f ms -> chip %.1f TFLOP/s | ", ms1, ms2, 256.0 * (NT / 64) * reps * 16.0 * 32768.0 / (ms2 * 1e-3) / 1e12);
  unsigned long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("bf16 32x32x16, waves/SIMD %d, chains %d, fillers/MFMA %2d: %.1f cycles per wave-MFMA\n", NT / 256, NCH, NV, (double)h[0] / (reps * 16.0));
}
int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1024);
  run<0, 256>(out, cyc); run<2, 256>(out, cyc); run<4, 256>(out, cyc); run<6, 256>(out, cyc); run<8, 256>(out, cyc); run<12, 256>(out, cyc);
  run<0, 512>(out, cyc); run<4, 512>(out, cyc); run<8, 512>(out, cyc); run<12, 512>(out, cyc);
  run<0, 256, 2>(out, cyc); run<0, 256, 4>(out, cyc); run<4, 256, 2>(out, cyc); run<0, 512, 2>(out, cyc); run<0, 1024, 1>(out, cyc);
  return 0;
}
