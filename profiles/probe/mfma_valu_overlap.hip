// How many independent VALU instructions hide in the shadow of one v_mfma_f32_32x32x2_f32 issued by the SAME wave?
// One wave per SIMD, a dependent MFMA chain (one accumulator) or two alternating chains, NV fillers after each MFMA.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int NV, int NCHAIN, int KIND, int NT = 256>
__global__ __launch_bounds__(NT) void k(float *out, unsigned long long *cyc, int reps) {
  f32x16 acc[2];
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  float a = threadIdx.x * 0.001f, b = 1.0f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      acc[m % NCHAIN] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m % NCHAIN], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[v % 8]) : "v"(b));
        if (KIND == 1) asm volatile("v_max_f32 %0, %0, %1" : "+v"(f[v % 8]) : "v"(b));
        if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&f[(v % 4) * 2]) : "v"(*(double *)&f[(v % 4) * 2]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * NT + threadIdx.x] = acc[0][0] + acc[1][3] + s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (NT / 64) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NV, int NCHAIN, int KIND, int NT = 256>
void run(float *out, unsigned long long *cyc) {
  const int reps = 2000;
  k<NV, NCHAIN, KIND, NT><<<1, NT>>>(out, cyc, 10);
  k<NV, NCHAIN, KIND, NT><<<1, NT>>>(out, cyc, reps);
  hipDeviceSynchronize();
  unsigned long long h[8];
  hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("waves/SIMD %d kind %d chains %d fillers/MFMA %2d: %.1f cycles per wave-MFMA (wave 0), %.1f (wave %d)\n", NT / 256, KIND, NCHAIN, NV,
         (double)h[0] / (reps * 16.0), (double)h[NT / 64 - 1] / (reps * 16.0), NT / 64 - 1);
}

int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 1024);
  run<0, 1, 0>(out, cyc); run<1, 1, 0>(out, cyc); run<2, 1, 0>(out, cyc); run<4, 1, 0>(out, cyc); run<6, 1, 0>(out, cyc);
  run<8, 1, 0>(out, cyc); run<10, 1, 0>(out, cyc); run<12, 1, 0>(out, cyc); run<16, 1, 0>(out, cyc);
  run<0, 2, 0>(out, cyc); run<2, 2, 0>(out, cyc); run<4, 2, 0>(out, cyc); run<8, 2, 0>(out, cyc); run<12, 2, 0>(out, cyc);
  run<4, 1, 1>(out, cyc); run<8, 1, 1>(out, cyc);
  run<2, 1, 2>(out, cyc); run<4, 1, 2>(out, cyc); run<8, 1, 2>(out, cyc);
  run<0, 1, 0, 512>(out, cyc); run<2, 1, 0, 512>(out, cyc); run<4, 1, 0, 512>(out, cyc); run<8, 1, 0, 512>(out, cyc); run<12, 1, 0, 512>(out, cyc); run<16, 1, 0, 512>(out, cyc);
  return 0;
}
