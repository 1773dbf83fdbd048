// gn_common.h -- shared device/host helpers for libgraspnet_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "graspnet_hip.h"

namespace gn {

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
int seg_order(const int32_t *idx, int b, int n, int ne, const int32_t *off, int32_t *list, hipStream_t s);

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
