// fps_util.h -- helpers shared by the furthest-point-sampling kernels (fps.hip, fps_cell.hip).
#pragma once
#include "gn_common.h"

namespace gn {

__device__ __forceinline__ int bitrev_bits(int v, int bits) {
  return bits ? (int)(__brev((unsigned)v) >> (32 - bits)) : 0;
}

// v_min_f32 without the canonicalising v_max pair hipcc wraps around fminf(): IEEE min, returns
// the non-NaN operand like CUDA's min(float,float) (sampling_gpu.cu:111).
__device__ __forceinline__ float vmin(float a, float b) {
  float r;
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float vmax(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// position in rank order -> original point index, or -1 for padding.
// qmagic = ceil(2^32 / Q): floor(p / Q) == umulhi(p, qmagic) for p < 2^15 <= 2^32 / Q (host checks).
__device__ __forceinline__ int pos_to_k(int p, int bs, int lg, int Q, unsigned qmagic, int n) {
  const int r = Q == 1 ? p : (int)__umulhi((unsigned)p, qmagic);
  const int q = p - r * Q;
  if (r >= bs) return -1;
  const int k = q * bs + bitrev_bits(r, lg);
  return k < n ? k : -1;
}

// pointnet2/_ext_src/include/cuda_utils.h:18-24 opt_n_threads(): the reference's block size, which fixes its tie order
static inline int fps_ref_block_size(int n) {
  int p = 0;
  while ((2 << p) <= n) ++p;  // floor(log2 n)
  int t = 1 << p;
  return t > 512 ? 512 : (t < 1 ? 1 : t);
}

// ---- cell kernel (fps_cell.hip): S x S regions of RS rows of 64 slots, state in an L2-resident workspace
constexpr int kFpsCellMaxN = 262144;
struct FpsCellCfg { int S, RS; };
// S*S*64*RS >= n; 16 x 16 regions of 1 / 2 / 4 rows up to 65 536 points, 32 x 32 beyond
static inline FpsCellCfg fps_cell_config(int n, int force_s = 0) {
  FpsCellCfg c;
  c.S = force_s ? force_s : (n <= 65536 ? 16 : 32);
  const int per = (n + c.S * c.S - 1) / (c.S * c.S);
  c.RS = per <= 64 ? 1 : (per <= 128 ? 2 : 4);
  return c;
}
static inline bool fps_cell_supports(int n, int force_s = 0) {
  if (n < 1 || n > kFpsCellMaxN) return false;
  const FpsCellCfg c = fps_cell_config(n, force_s);
  return (long)c.S * c.S * 64 * c.RS >= n;
}
static inline size_t fps_cell_workspace_bytes_one(int n, int force_s = 0) {
  const FpsCellCfg c = fps_cell_config(n, force_s);
  const size_t P = (size_t)c.S * c.S * 64 * c.RS;
  return P * 20 + (((size_t)n + 255) & ~(size_t)255);     // (x,y,z,temp) + key word per slot, slab id per point
}
namespace GN_FPNS {      // one copy per evaluation policy (gn_common.h)
int fps_cell_launch(const float *xyz, int b, int n, int m, int bs, int lg, int Q, unsigned qmagic, void *workspace,
                    int32_t *idx, float *new_xyz, int32_t *tie_out, int force_s, hipStream_t s);
}

}  // namespace gn
