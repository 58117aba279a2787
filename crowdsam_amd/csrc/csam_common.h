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

// One-time per-DEVICE setup (hipFuncSetAttribute(MaxDynamicSharedMemorySize), CU count): a process may drive more than
// one GPU through the API (the bench is one process per GPU), so "once" is keyed by hipGetDevice(), not by process.
//   static csam_once_t once;  if (csam_first_call(once)) hipFuncSetAttribute(...);
//   const int n_cu = csam_cu_count();
#define CSAM_MAX_DEVICES 64
struct csam_once_t { bool done[CSAM_MAX_DEVICES]; };
static inline bool csam_first_call(csam_once_t& o) {
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= CSAM_MAX_DEVICES) return true;   // unknown: redo the (cheap) setup
  if (o.done[d]) return false;
  o.done[d] = true;
  return true;
}
static inline int csam_cu_count() {
  static int n_cu[CSAM_MAX_DEVICES];
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= CSAM_MAX_DEVICES) d = 0;
  if (n_cu[d] == 0) {
    hipDeviceProp_t prop;
    n_cu[d] = (hipGetDeviceProperties(&prop, d) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return n_cu[d];
}

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
// Packed-fp32 polynomial GELU for fp16-bound outputs (the fused upscaler, which PMC shows VALU-bound on GELU):
//   Phi(x) ~ 0.5 + xc * R(xc^2),  xc = clamp(x, +-4.4),  R = degree-8 minimax fit with Phi(4.4) = 1 exactly,
// |gelu error| <= 2.5e-5 for |x| < 3 and <= 4.5e-5 overall (tools/fit_gelu_poly.py) -- an order below the fp16
// rounding of the value it feeds -- in 13 instructions per PAIR (v_med3 x2, 11 v_pk_*_f32), no transcendental,
// against ~22 issue slots per element for the erf form above.
// (Round 5 tried the clamp on the OUTPUT instead -- the VOP3P `clamp` bit of the last packed FMA, which needs a one-instruction
// inline asm because hipcc does not fold it: -7 % VALU instructions in the upscaler, -0.7 % time, and the ISA lint
// (tools/lint_mfma_srcc.py) caught the asm overwriting an MFMA's SrcC quad two instructions behind it with no wait states:
// the hazard recogniser pads compiler-emitted VALU writes there, not asm ones.  Reverted; profiles/r05_upscale_fp16_gelu.txt.)
typedef float float2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float2_t csam_gelu_poly2(float2_t x) {
  const float c = 4.4f;
  const float2_t xc = {__builtin_amdgcn_fmed3f(x[0], -c, c), __builtin_amdgcn_fmed3f(x[1], -c, c)};
  const float2_t u = xc * xc;
  const float k[9] = {4.471991608e-11f, -4.528126140e-09f, 2.016253663e-07f, -5.250151905e-06f, 9.008348436e-05f,
                      -1.092016766e-03f, 9.773204936e-03f, -6.629599897e-02f, 3.988868129e-01f};
  float2_t r = {k[0], k[0]};
#pragma unroll
  for (int i = 1; i < 9; ++i) r = __builtin_elementwise_fma(r, u, (float2_t){k[i], k[i]});
  const float2_t ph = __builtin_elementwise_fma(xc, r, (float2_t){0.5f, 0.5f});
  return x * ph;
}

// N independent pairs at once, coefficient-major: consecutive instructions belong to different Horner chains.  A
// v_pk_fma_f32 whose input is the previous instruction's result costs an s_nop on gfx950 (the single-pair form above
// compiles to fma / s_nop / fma / s_nop ...: 9 wasted issue slots per pair); interleaved chains have none.
template <int N>
__device__ __forceinline__ void csam_gelu_poly2_n(float2_t (&x)[N]) {
  const float c = 4.4f;
  const float k[9] = {4.471991608e-11f, -4.528126140e-09f, 2.016253663e-07f, -5.250151905e-06f, 9.008348436e-05f,
                      -1.092016766e-03f, 9.773204936e-03f, -6.629599897e-02f, 3.988868129e-01f};
  float2_t xc[N], u[N], r[N];
#pragma unroll
  for (int p = 0; p < N; ++p) {
    xc[p] = float2_t{__builtin_amdgcn_fmed3f(x[p][0], -c, c), __builtin_amdgcn_fmed3f(x[p][1], -c, c)};
    u[p] = xc[p] * xc[p];
  }
#pragma unroll
  for (int p = 0; p < N; ++p) r[p] = __builtin_elementwise_fma((float2_t){k[0], k[0]}, u[p], (float2_t){k[1], k[1]});
#pragma unroll
  for (int i = 2; i < 9; ++i)
#pragma unroll
    for (int p = 0; p < N; ++p) r[p] = __builtin_elementwise_fma(r[p], u[p], (float2_t){k[i], k[i]});
#pragma unroll
  for (int p = 0; p < N; ++p) x[p] = x[p] * __builtin_elementwise_fma(xc[p], r[p], (float2_t){0.5f, 0.5f});
}

// exp2 for softmax arguments (<= 0, results in (0, 1]): the bare v_exp_f32.  exp2f() wraps it in a denormal-range
// rescue (compare + 2 selects + ldexp per call) that only matters for results below 2^-126, which a softmax
// weight may flush to zero.
__device__ __forceinline__ float csam_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

__device__ __forceinline__ float csam_apply_act(float v, int act) {
  if (act == CSAM_ACT_GELU) return csam_gelu_erf(v);
  if (act == CSAM_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
// Reductions with the partner 32 / 16 lanes away through v_permlane32_swap / v_permlane16_swap (gfx950) instead of __shfl_xor
// (= ds_bpermute_b32: an LDS round trip and an `s_waitcnt lgkmcnt(0)` that also drains every LDS read in flight).  Swapping a
// register with a copy of itself leaves (own, partner) in the even 16- / 32-lane groups and (partner, own) in the odd ones; +
// and max are commutative, so every lane gets the bits the shuffle form gave it.  The results are copied to scalars before the
// bit cast: __builtin_bit_cast of an ELEMENT of the builtin's vector result reads element 0 (clang; seen in the ISA).
// (The same idiom -- permlane32_swap(x, x), then op(result.x, result.y) -- is what the vendor's ck_tile fmha v3 pipeline uses on
// gfx950 for its row maximum and row sum: /opt/rocm/include/ck_tile/ops/fmha/pipeline/block_fmha_fwd_v3_pipeline.hpp.)
__device__ __forceinline__ void csam_swap32(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto s = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned x = s[0], y = s[1];
  a = __builtin_bit_cast(float, x);
  b = __builtin_bit_cast(float, y);
}
__device__ __forceinline__ void csam_swap16(float v, float& a, float& b) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const auto s = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned x = s[0], y = s[1];
  a = __builtin_bit_cast(float, x);
  b = __builtin_bit_cast(float, y);
}
#ifndef CSAM_SWAP_REDUCE      // developer A/B: -DCSAM_SWAP_REDUCE=0 = the __shfl_xor forms
#define CSAM_SWAP_REDUCE 1
#endif
#if CSAM_SWAP_REDUCE
__device__ __forceinline__ float csam_max_x32(float v) { float a, b; csam_swap32(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float csam_max_x16(float v) { float a, b; csam_swap16(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float csam_sum_x32(float v) { float a, b; csam_swap32(v, a, b); return a + b; }
__device__ __forceinline__ float csam_sum_x16(float v) { float a, b; csam_swap16(v, a, b); return a + b; }
#else
__device__ __forceinline__ float csam_max_x32(float v) { return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float csam_max_x16(float v) { return fmaxf(v, __shfl_xor(v, 16, 64)); }
__device__ __forceinline__ float csam_sum_x32(float v) { return v + __shfl_xor(v, 32, 64); }
__device__ __forceinline__ float csam_sum_x16(float v) { return v + __shfl_xor(v, 16, 64); }
#endif

__device__ __forceinline__ float csam_wave_sum(float v) {
  v = csam_sum_x32(v);
  v = csam_sum_x16(v);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float csam_wave_max(float v) {
  v = csam_max_x32(v);
  v = csam_max_x16(v);
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

