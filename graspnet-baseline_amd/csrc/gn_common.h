// gn_common.h -- shared device/host helpers for libgraspnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "graspnet_hip.h"

// ---- evaluation policy of the index-deciding float expressions (round-4 VERDICT item 3) --------------------------------
// The reference's setup.py (pointnet2/setup.py:24-27, knn/setup.py) builds its kernels with plain `nvcc -O2`, i.e.
// --fmad=true: the distance expressions of sampling_gpu.cu:108-109, ball_query_gpu.cu:36-37, cylinder_query_gpu.cu:39-45,
// interpolate_gpu.cu:37 and knn.cu:89-90 are CONTRACTED into fused multiply-adds in the reference's own binary, in an order
// that cannot be observed here (no nvcc).  The files holding such expressions (fps.hip, fps_cell.hip, query.hip, grid.hip,
// knn.hip, group.hip's three_nn) are therefore compiled three times (csrc/Makefile, -DGN_FP=0/1/2), into the namespaces
// gn::fp0 / fp1 / fp2, and every C entry picks the copy gn_set_fp_policy() / the GN_FP environment variable names:
//   0 as_written    one rounding per operation, left to right, no contraction (this build's -ffp-contract=off) -- the default,
//                   what a CPU build of the same expressions computes (the reference's knn_cpu.cpp under gcc on x86-64);
//   1 contracted    a*a + b*b + c*c = fma(c, c, fma(a, a, b*b)): the order LLVM's DAG combiner (and with it NVVM / hipcc at
//                   their default -ffp-contract=fast) gives the expression tree (a*a + b*b) + c*c -- the left multiply of the
//                   inner add is fused first, then the right multiply of the outer one;
//   2 contracted_b  a*a + b*b + c*c = fma(c, c, fma(b, b, a*a)): the other order a compiler may pick (accumulate left to right).
// `s += t*t` (knn.cu:90) is fma(t, t, s) under 1 and 2.  Everything else (differences, radius*radius, compares) is unchanged.
#ifndef GN_FP
#define GN_FP 0
#endif
#if GN_FP == 0
#define GN_FPNS fp0
#elif GN_FP == 1
#define GN_FPNS fp1
#elif GN_FP == 2
#define GN_FPNS fp2
#else
#error "GN_FP must be 0, 1 or 2"
#endif
#define GN_CAT2_(a, b) a##b
#define GN_CAT_(a, b) GN_CAT2_(a, b)
// name of a policy-dependent C entry in this translation unit: the plain name in the policy-0 copy (which dispatches),
// name_fp1 / name_fp2 in the others
#if GN_FP == 0
#define GN_FP_ENTRY(name) name
#else
#define GN_FP_ENTRY(name) GN_CAT_(name, GN_CAT_(_, GN_FPNS))
#endif
// first statement of a policy-dependent C entry: in the policy-0 copy, hand the call to the copy the policy names
#if GN_FP == 0
#define GN_FP_DISPATCH(name, ...)                         \
  do {                                                    \
    const int p_ = gn::fp_policy();                       \
    if (p_ == 1) return name##_fp1(__VA_ARGS__);          \
    if (p_ == 2) return name##_fp2(__VA_ARGS__);          \
  } while (0)
#else
#define GN_FP_DISPATCH(name, ...) do { } while (0)
#endif

namespace gn {

int fp_policy();      // gn_common.cpp: gn_set_fp_policy(), else GN_FP from the environment (read once), else 0

namespace GN_FPNS {
// a*a + b*b + c*c
__host__ __device__ __forceinline__ float fp_sumsq3(float a, float b, float c) {
#if GN_FP == 0
  return a * a + b * b + c * c;
#elif GN_FP == 1
  return __builtin_fmaf(c, c, __builtin_fmaf(a, a, b * b));
#else
  return __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a));
#endif
}
// a*a + b*b
__host__ __device__ __forceinline__ float fp_sumsq2(float a, float b) {
#if GN_FP == 0
  return a * a + b * b;
#elif GN_FP == 1
  return __builtin_fmaf(a, a, b * b);
#else
  return __builtin_fmaf(b, b, a * a);
#endif
}
// a0*b0 + a1*b1 + a2*b2
__host__ __device__ __forceinline__ float fp_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
#if GN_FP == 0
  return a0 * b0 + a1 * b1 + a2 * b2;
#elif GN_FP == 1
  return __builtin_fmaf(a2, b2, __builtin_fmaf(a0, b0, a1 * b1));
#else
  return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, a0 * b0));
#endif
}
// s + t*t
__host__ __device__ __forceinline__ float fp_acc_sq(float s, float t) {
#if GN_FP == 0
  return s + t * t;
#else
  return __builtin_fmaf(t, t, s);
#endif
}
}  // namespace GN_FPNS

// ---- error plumbing (never exit(); cf. cuda_utils.h:35-44 in the reference)
void set_error(const char *fmt, ...);

#define GN_REQUIRE(cond, ...)                 \
  do {                                        \
    if (!(cond)) {                            \
      gn::set_error(__VA_ARGS__);             \
      return GN_ERR_INVALID_ARG;              \
    }                                         \
  } while (0)

#define GN_CHECK_LAUNCH(name)                                              \
  do {                                                                     \
    hipError_t e_ = hipGetLastError();                                     \
    if (e_ != hipSuccess) {                                                \
      gn::set_error("%s: kernel launch failed: %s", name, hipGetErrorString(e_)); \
      return GN_ERR_LAUNCH;                                                \
    }                                                                      \
  } while (0)

static inline hipStream_t as_stream(gn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// grad_det.hip: ascending order inside every destination's segment of an inverted index list
// skip (b, n) u8 or NULL: destinations flagged there keep their arbitrary (ticket) order
int seg_order(const int32_t *idx, int b, int n, int ne, const int32_t *off, int32_t *list, hipStream_t s, const uint8_t *skip = nullptr);

// ---- wave64 DPP reductions (row = 16 lanes).  dpp_ctrl encodings: quad_perm 0x00-0xFF,
// row_half_mirror 0x141, row_mirror 0x140.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}

// max over each 16-lane row with the DPP fused into v_max_f32 (one instruction per step; hipcc's
// fmaxf(v, dpp(v)) costs a v_mov_dpp plus two canonicalising v_max).  IEEE max: a NaN lane is
// ignored, matching fmaxf.  s_nop 1 covers the VALU-write -> DPP-read hazard inside the statement.
__device__ __forceinline__ float row16_max_fast(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}

// max over the whole wave in six DPP steps (the two row_bcast steps fold rows 0..2 into row 3: lane 63 ends up with
// the result), returned wave-uniform: one readlane instead of four readlanes and three scalar-operand max
__device__ __forceinline__ float wave_max_fast(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1"
      : "+v"(v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// max over lanes 0..7 only (three steps), valid in lanes 0..7
__device__ __forceinline__ float lanes8_max_fast(float v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}
__device__ __forceinline__ unsigned lanes8_max_u32_fast(unsigned v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}

// unsigned-integer flavour of row16_max_fast
__device__ __forceinline__ unsigned row16_max_u32_fast(unsigned v) {
  asm volatile(
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1\n\t"
      "v_max_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "s_nop 1"
      : "+v"(v));
  return v;
}

// max over each 16-lane row, result in every lane of the row
__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, dpp_f<0xB1>(v));   // quad_perm [1,0,3,2]
  v = fmaxf(v, dpp_f<0x4E>(v));   // quad_perm [2,3,0,1]
  v = fmaxf(v, dpp_f<0x141>(v));  // row_half_mirror
  v = fmaxf(v, dpp_f<0x140>(v));  // row_mirror
  return v;
}
__device__ __forceinline__ int row16_min_i(int v) {
  v = min(v, dpp_i<0xB1>(v));
  v = min(v, dpp_i<0x4E>(v));
  v = min(v, dpp_i<0x141>(v));
  v = min(v, dpp_i<0x140>(v));
  return v;
}
// max over the whole wave, wave-uniform result (scalar)
__device__ __forceinline__ float wave_max(float v) {
  v = row16_max(v);
  const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
  const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
  const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
  const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
  return fmaxf(fmaxf(a, b), fmaxf(c, d));
}
__device__ __forceinline__ int wave_min_i(int v) {
  v = row16_min_i(v);
  const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
  const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  return min(min(a, b), min(c, d));
}

__device__ __forceinline__ int lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

}  // namespace gn

// the policy copies of the policy-dependent C entries (defined by the -DGN_FP=1 / 2 objects; see GN_FP_ENTRY / GN_FP_DISPATCH)
#define GN_FP_DECLARE(name, ...) extern "C" int name##_fp1(__VA_ARGS__); extern "C" int name##_fp2(__VA_ARGS__);
GN_FP_DECLARE(gn_furthest_point_sampling, const float *, int, int, int, void *, int32_t *, float *, int32_t *, const int32_t *, gn_stream_t)
GN_FP_DECLARE(gn_ball_query, const float *, const float *, int, int, int, float, int, int32_t *, gn_stream_t)
GN_FP_DECLARE(gn_cylinder_query, const float *, const float *, const float *, int, int, int, float, float, float, int, int32_t *, gn_stream_t)
GN_FP_DECLARE(gn_cylinder_query_multi, const float *, const float *, const float *, int, int, int, float, float, const float *, int, int,
              int32_t *, gn_stream_t)
GN_FP_DECLARE(gn_cylinder_query_unique, const float *, const float *, const float *, int, int, int, float, float, const float *, int, int,
              int32_t *, int32_t *, gn_stream_t)
GN_FP_DECLARE(gn_ball_query_grid, const float *, const void *, int, int, int, float, int, int32_t *, gn_stream_t)
GN_FP_DECLARE(gn_cylinder_query_grid, const float *, const float *, const void *, int, int, int, float, float, float, int, int32_t *,
              gn_stream_t)
GN_FP_DECLARE(gn_cylinder_query_unique_grid, const float *, const float *, const void *, int, int, int, float, float, const float *, int,
              int, int32_t *, int32_t *, gn_stream_t)
GN_FP_DECLARE(gn_three_nn, const float *, const float *, int, int, int, float *, int32_t *, gn_stream_t)
GN_FP_DECLARE(gn_knn, const float *, const float *, int, int, int, int, int, int64_t *, float *, gn_stream_t)
