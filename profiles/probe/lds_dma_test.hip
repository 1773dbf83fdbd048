// Where does global_load_lds_dwordx4 put each lane's 16 bytes?  (gfx950; csrc/mlp_point_wide.hip stages its input with it)
// build: hipcc --offload-arch=gfx950 -O3 profiles/probe/lds_dma_test.hip -o profiles/probe/lds_dma_test
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* __restrict__ g, float* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* lds = reinterpret_cast<float*>(smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 4 * 256; i += 256) lds[i] = -1.f;
  __syncthreads();
  // lane L fetches global piece (L ^ 5) of its wave's 1 KB block
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + wave * 256 + (lane ^ 5) * 4),
                                   (__attribute__((address_space(3))) void*)(lds + wave * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4 * 256; i += 256) out[i] = lds[i];
}
int main() {
  float h[1024], *g, *o, r[1024];
  for (int i = 0; i < 1024; ++i) h[i] = (float)i;
  hipMalloc(&g, sizeof(h)); hipMalloc(&o, sizeof(h));
  hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 4096, 0, g, o);
  hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int ok = 1;
  for (int w = 0; w < 4; ++w)
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) ok &= r[w * 256 + l * 4 + j] == (float)(w * 256 + (l ^ 5) * 4 + j);
  printf("lane-linear 16-byte slots behind the M0 base: %s\n", ok ? "YES" : "NO");
  if (!ok) for (int i = 0; i < 32; ++i) printf("%g ", r[i]);
  printf("\n");
  return !ok;
}
