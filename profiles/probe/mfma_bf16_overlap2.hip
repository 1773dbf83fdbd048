// Which VALU fillers overlap a dependent v_mfma_f32_32x32x16_bf16 chain?  Explicit asm so register classes are known.
#include <hip/hip_runtime.h>
#include <stdio.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

// MODE 0: acc in VGPR, independent v_add fillers; 1: acc in AGPR, independent fillers; 2: acc AGPR, dependent chain
// add->max->max3; 3: acc AGPR, B operand in AGPR; 4: acc VGPR dependent chain; 5: acc AGPR + v_max3 only (independent)
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int reps) {
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  u32x4 a, b;
  for (int i = 0; i < 4; ++i) { a[i] = 0x3f803f80u; b[i] = 0x3f803f80u + threadIdx.x; }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = threadIdx.x + i;
  float one = 1.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll
    for (int m = 0; m < 16; ++m) {
      if (MODE == 0 || MODE == 4) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
      if (MODE == 1 || MODE == 2 || MODE == 5) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
      if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b));
      if (MODE == 0 || MODE == 1 || MODE == 3) {
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[v % 8]) : "v"(one));
      } else if (MODE == 5) {
#pragma unroll
        for (int v = 0; v < NV; ++v) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[v % 8]) : "v"(one), "v"(f[(v + 4) % 8]));
      } else {
#pragma unroll
        for (int v = 0; v < NV; v += 3)
          asm volatile("v_add_f32 %0, %1, %2\n\tv_add_f32 %3, %1, %4\n\tv_max_f32 %0, %0, %3" : "+v"(f[0]), "+v"(one) : "v"(f[2]), "v"(f[3]), "v"(f[4]));
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += f[i];
  out[threadIdx.x] = acc[0] + acc[3] + s;
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
}
template <int MODE, int NV>
void run(const char *what, float *out, unsigned long long *cyc) {
  const int reps = 2000;
  k<MODE, NV><<<1, 256>>>(out, cyc, 10);
  k<MODE, NV><<<1, 256>>>(out, cyc, reps);
  hipDeviceSynchronize();
  unsigned long long h[4];
  hipMemcpy(h, cyc, 32, hipMemcpyDeviceToHost);
  printf("%-44s fillers/MFMA %d: %.1f cycles per MFMA\n", what, NV, (double)h[0] / (reps * 16.0));
}
int main() {
  float *out; unsigned long long *cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 1024);
  run<0, 0>("acc VGPR", out, cyc); run<0, 3>("acc VGPR, independent v_add", out, cyc);
  run<1, 0>("acc AGPR", out, cyc); run<1, 3>("acc AGPR, independent v_add", out, cyc); run<1, 6>("acc AGPR, independent v_add", out, cyc);
  run<2, 3>("acc AGPR, dependent add,add,max", out, cyc); run<2, 6>("acc AGPR, dependent add,add,max x2", out, cyc);
  run<4, 3>("acc VGPR, dependent add,add,max", out, cyc);
  run<3, 3>("acc AGPR, B operand AGPR, independent v_add", out, cyc);
  run<5, 3>("acc AGPR, v_max3", out, cyc);
  return 0;
}
