// checks the DPP compare-exchange steps of csrc/grid.hip against the ds_bpermute form, mask by mask, and sort64 / merge_low64.
// grid_cmpx_extract.inc = the helper block of grid.hip from `// One compare-exchange step` to `constexpr int GQ_WAVES`
// (cut out by a two-line python snippet, see profiles/README.md); build: hipcc -O3 --offload-arch=gfx950 dpp_cmpx_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
namespace gn {
#include "grid_cmpx_extract.inc"
}
__global__ void k(const int *in, int *out) {
  const int lane = threadIdx.x;
  int v = in[lane];
  int r = 0;
  const int masks[6] = {1, 2, 4, 8, 16, 32};
#pragma unroll
  for (int mi = 0; mi < 6; ++mi) {
    const int j = masks[mi];
    const int o = __shfl_xor(v, j);
    const bool km = (lane & j) == 0;
    const int want = km ? min(v, o) : max(v, o);
    const int got = gn::cmpx(v, j, km, lane);
    out[mi * 64 + lane] = (want == got) ? 0 : 1;
    r += want != got;
  }
  out[6 * 64 + lane] = gn::sort64(v, lane);
  const int a = gn::sort64(v, lane), b = gn::sort64(in[64 + lane], lane);
  out[7 * 64 + lane] = gn::merge_low64(a, b, lane);
}
int main() {
  int h[128], *d, *o, ho[8 * 64];
  srand(1);
  for (int i = 0; i < 128; ++i) h[i] = rand() % 100000;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
  hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
  for (int mi = 0; mi < 6; ++mi) { int bad = 0; for (int l = 0; l < 64; ++l) bad += ho[mi * 64 + l]; printf("mask %d: %d lanes wrong\n", 1 << mi, bad); }
  int s[64]; std::copy(h, h + 64, s); std::sort(s, s + 64);
  int bad = 0; for (int l = 0; l < 64; ++l) bad += s[l] != ho[6 * 64 + l];
  printf("sort64: %d wrong\n", bad);
  int m[128]; std::copy(h, h + 128, m); std::sort(m, m + 128);
  bad = 0; for (int l = 0; l < 64; ++l) bad += m[l] != ho[7 * 64 + l];
  printf("merge_low64: %d wrong\n", bad);
  return 0;
}
