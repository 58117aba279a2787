// Shared device/host helpers for libcsam_hip.so (gfx950 / CDNA4 only).
// Error convention: every entry returns 0 on success, <0 on error;
// csam_last_error() returns a thread-local message. No C++ exceptions cross the ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

#define CSAM_OK 0
#define CSAM_ERR_ARG (-1)
#define CSAM_ERR_HIP (-2)
#define CSAM_ERR_WORKSPACE (-3)

#ifdef __cplusplus
extern "C" {
#endif
void csam_set_error(const char* fmt, ...);
#ifdef __cplusplus
}
#endif

#define CSAM_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      csam_set_error(__VA_ARGS__);              \
      return CSAM_ERR_ARG;                      \
    }                                           \
  } while (0)

#define CSAM_LAUNCH_CHECK(name)                                              \
  do {                                                                       \
    hipError_t e__ = hipGetLastError();                                      \
    if (e__ != hipSuccess) {                                                 \
      csam_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return CSAM_ERR_HIP;                                                   \
    }                                                                        \
  } while (0)

static inline int csam_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// activation ids shared by the GEMM epilogues (include/csam.h: CSAM_ACT_*)
#define CSAM_ACT_NONE 0
#define CSAM_ACT_GELU 1
#define CSAM_ACT_RELU 2
// dtype ids (include/csam.h: CSAM_DT_*)
#define CSAM_DT_F16 0
#define CSAM_DT_F32 1

// erf via Abramowitz-Stegun 7.1.26 (|abs err| <= 1.5e-7): 1 rcp + 1 exp + 6 FMA instead of libm erff's
// ~50 instructions -- the exact-erf GELU (nn.GELU default) was the top VALU cost of the fused upscaler.
__device__ __forceinline__ float csam_erf_fast(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));   // v_rcp_f32 (1 ulp), not the 11-op IEEE divide
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float y = 1.0f - p * t * __expf(-ax * ax);
  return copysignf(y, x);
}
__device__ __forceinline__ float csam_gelu_erf(float x) {
  return 0.5f * x * (1.0f + csam_erf_fast(x * 0.70710678118654752440f));
}
__device__ __forceinline__ float csam_apply_act(float v, int act) {
  if (act == CSAM_ACT_GELU) return csam_gelu_erf(v);
  if (act == CSAM_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
__device__ __forceinline__ float csam_wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float csam_wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
