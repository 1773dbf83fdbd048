#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(const float* in, float* o0, float* o1, float* om){
  float v = in[threadIdx.x];
  unsigned u = __builtin_bit_cast(unsigned, v);
  unsigned r[2] = {u, u};
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(r[0]), "+v"(r[1]));
  o0[threadIdx.x] = __builtin_bit_cast(float, r[0]);
  o1[threadIdx.x] = __builtin_bit_cast(float, r[1]);
  om[threadIdx.x] = fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}
int main(){
  float h[64], *d, *a, *b, *c; for(int i=0;i<64;i++) h[i]=i;
  hipMalloc(&d,256); hipMalloc(&a,256); hipMalloc(&b,256); hipMalloc(&c,256);
  hipMemcpy(d,h,256,hipMemcpyHostToDevice);
  k<<<1,64>>>(d,a,b,c);
  float x[64],y[64],z[64]; hipMemcpy(x,a,256,hipMemcpyDeviceToHost); hipMemcpy(y,b,256,hipMemcpyDeviceToHost); hipMemcpy(z,c,256,hipMemcpyDeviceToHost);
  for(int i=0;i<64;i+=8) printf("lane %2d: r0=%g r1=%g max=%g\n", i, x[i], y[i], z[i]);
  return 0;
}
