// Shared device/host helpers for the PCAA HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/pcaa_hip.h"

typedef __bf16 bf16_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- errors
void pcaa_set_error(const char* fmt, ...);

#define PCAA_CHECK_ARG(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      pcaa_set_error(__VA_ARGS__);                \
      return PCAA_ERR_INVALID_ARG;                \
    }                                             \
  } while (0)

#define PCAA_RETURN_LAUNCH_STATUS(name)                                        \
  do {                                                                         \
    hipError_t e__ = hipGetLastError();                                        \
    if (e__ != hipSuccess) {                                                   \
      pcaa_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
      return PCAA_ERR_LAUNCH;                                                  \
    }                                                                          \
    return PCAA_OK;                                                            \
  } while (0)

static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- math
__device__ __forceinline__ float elu_f(float z) { return z > 0.f ? z : expm1f(z); }
// derivative of ELU evaluated from the pre-activation
__device__ __forceinline__ float elu_grad_from_pre(float z) { return z > 0.f ? 1.f : expf(z); }
// bf16-storage variants: v_exp_f32-based (2 instructions instead of ~20).  The result is rounded
// to 8 mantissa bits anyway; the fp32 parity path keeps expm1f / expf.  With libm exp the
// bf16 elementwise passes were VALU-bound, not HBM-bound.
template <typename T> __device__ __forceinline__ float elu_t(float z) {
  if constexpr (sizeof(T) == 2) return z > 0.f ? z : __expf(z) - 1.f;
  else return elu_f(z);
}
template <typename T> __device__ __forceinline__ float elu_grad_from_pre_t(float z) {
  if constexpr (sizeof(T) == 2) return z > 0.f ? 1.f : __expf(z);
  else return elu_grad_from_pre(z);
}
// derivative of ELU evaluated from the OUTPUT a = ELU(z):  z<=0 -> e^z = a + 1
__device__ __forceinline__ float elu_grad_from_out(float a) { return a > 0.f ? 1.f : a + 1.f; }

// ---------------------------------------------------------------- 4-wide typed access
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const bf16_t* p) {
  uint2 raw = *reinterpret_cast<const uint2*>(p);
  f32x4 r;
  r.x = __uint_as_float(raw.x << 16);
  r.y = __uint_as_float(raw.x & 0xffff0000u);
  r.z = __uint_as_float(raw.y << 16);
  r.w = __uint_as_float(raw.y & 0xffff0000u);
  return r;
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, f32x4 v) {
  bf16x4 b;
  b.x = (bf16_t)v.x; b.y = (bf16_t)v.y; b.z = (bf16_t)v.z; b.w = (bf16_t)v.w;
  *reinterpret_cast<bf16x4*>(p) = b;
}
// [hi | lo] fp16 image of fp32 values, the operand format of pcaa_gemm_split3: row r of an fp32 [rows, ch] tensor is
// stored as hi = fp16(s v) at img[r][c] and lo = fp16(s v - hi) at img[r][ch + c] (same bytes as the fp32 row);
// s is a power of two that puts the tensor's values into fp16's normal range (activations 1, weights 2^8,
// gradients 2^16): hi + lo then carries 22 mantissa bits, against 16 for a bf16 pair.
typedef _Float16 split_t;
typedef _Float16 split_x4 __attribute__((ext_vector_type(4)));
// Range guard of the images (round 4, advisor finding): fp16 holds |x| <= 65504, the fp32 tensors the images stand for
// have no such limit (ELU is unbounded above, a gradient spike can pass 2^-16 ... 1).  A value whose scaled magnitude
// leaves fp16's range is SATURATED (hi = +-65504, lo = 0: finite, so one outlier does not turn the whole product into
// inf - inf = NaN) and the caller's device flag is raised: pcaa_set_range_flag registers it, the producers' launchers
// hand it to their kernels, PCAATrainer.check() reads it and refuses the step's result (the exact-fp32 mode has no
// limit).  NaN inputs raise the flag too and stay NaN.
constexpr float SPLIT_MAX = 65504.f;
int* pcaa_range_flag_ptr();      // host: the registered device flag of the calling thread (may be null)
__device__ __forceinline__ float split_guard(float v, int* oflow) {
  if (!(fabsf(v) <= SPLIT_MAX)) {
    if (oflow) *oflow = 1;
    // a NaN stays NaN (fminf / fmaxf drop a NaN operand and would have turned it into -65504 -- a
    // diverged step must keep showing NaN losses, not finite ones); +-inf and finite outliers saturate
    if (v == v) v = fminf(fmaxf(v, -SPLIT_MAX), SPLIT_MAX);
  }
  return v;
}
__device__ __forceinline__ void store4_split(split_t* img, size_t row, unsigned ch, unsigned c, f32x4 v, float s,
                                             int* oflow) {
  v *= s;
  const float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
  if (!(m <= SPLIT_MAX) || v.x != v.x || v.y != v.y || v.z != v.z || v.w != v.w) {
    v.x = split_guard(v.x, oflow); v.y = split_guard(v.y, oflow);
    v.z = split_guard(v.z, oflow); v.w = split_guard(v.w, oflow);
    if (oflow) *oflow = 1;
  }
  split_x4 hi, lo;
  hi.x = (split_t)v.x; hi.y = (split_t)v.y; hi.z = (split_t)v.z; hi.w = (split_t)v.w;
  lo.x = (split_t)(v.x - (float)hi.x); lo.y = (split_t)(v.y - (float)hi.y);
  lo.z = (split_t)(v.z - (float)hi.z); lo.w = (split_t)(v.w - (float)hi.w);
  split_t* p = img + row * 2 * (size_t)ch + c;
  *reinterpret_cast<split_x4*>(p) = hi;
  *reinterpret_cast<split_x4*>(p + ch) = lo;
}
__device__ __forceinline__ float load1(const float* p) { return *p; }
__device__ __forceinline__ float load1(const bf16_t* p) { return (float)*p; }
__device__ __forceinline__ void store1(float* p, float v) { *p = v; }
__device__ __forceinline__ void store1(bf16_t* p, float v) { *p = (bf16_t)v; }

// One element of torch.optim.Adam (amsgrad off, no weight decay; step_size = lr / (1 - b1^t), inv_bc2_sqrt =
// 1 / sqrt(1 - b2^t)) with its rounding points FIXED (explicit fused multiply-adds, no contraction elsewhere):
// every kernel that applies the update -- the flat-buffer pass and the decoder's weight-gradient kernel --
// produces the same bits from the same operands.
__device__ __forceinline__ void adam_update(float& p, float& m, float& v, float g, float b1, float b2, float eps,
                                            float step_size, float inv_bc2_sqrt) {
#pragma clang fp contract(off)
  m = __builtin_fmaf(b1, m, (1.f - b1) * g);
  v = __builtin_fmaf(b2, v, ((1.f - b2) * g) * g);
  const float denom = __builtin_fmaf(sqrtf(v), inv_bc2_sqrt, eps);
  p = __builtin_fmaf(-step_size, m / denom, p);
}

// block-wide sum of one float (blockDim.x multiple of 64, <= 1024); result valid in thread 0
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
