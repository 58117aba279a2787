// What does the VALU pipe of a gfx950 SIMD charge per instruction class?  (round 4, HISTORY.md §4.2e)
// Every wave runs ITER x 16 INDEPENDENT instructions of one class (inline asm, 16 separate destination registers, so neither a
// dependency nor the compiler is in the way) at 1 / 2 / 3 / 4 waves per SIMD; wave 0 of every workgroup brackets the loop
// with s_memtime.  Printed: shader cycles per instruction as ONE wave sees it, and per SIMD (= that / waves per SIMD) -- the
// second column is the pipe's price once it no longer falls with more waves.  The last rows time the upscaler's actual
// GELU (csam_gelu_poly2_n<4>, the source the kernel compiles) per PAIR of evaluations.
//   hipcc --offload-arch=gfx950 -O3 -I crowdsam_amd/csrc -o valu_rate tools/probe/valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "csam_common.h"

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 512;

#define REP16(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10) M(11) M(12) M(13) M(14) M(15)

// Round 6: the GELU off the VALU pipe -- Phi(x) from a piecewise-linear table in LDS, 1024 intervals of
// (value, slope) fp16 pairs on [-4.4, 4.4]: index + fraction = 1 FMA (clamp modifier) + 1 mul + 1 fract + 1 cvt + 1 shift, one
// ds_read_b32, 1 mixed-precision FMA + 1 mul.  OP 9: every lane its own argument (N(0,1)-like spread: the bank conflicts a
// real activation tile would see); OP 18: all lanes the same argument (LDS broadcast: the VALU side alone).
constexpr int TBL_N = 1024;
__device__ __forceinline__ float gelu_tbl(float x, const unsigned* tbl) {
  float t;
  asm("v_fma_f32 %0, %1, %2, 0.5 clamp" : "=v"(t) : "v"(x), "v"(1.0f / 8.8f));       // (x + 4.4) / 8.8 clamped to [0, 1]
  const float u = t * (float)TBL_N;
  const unsigned i = (unsigned)u;                                                       // v_cvt_u32_f32 (truncation = floor here)
  const float f = __builtin_amdgcn_fractf(u);
  const unsigned e = tbl[i];
  float ph;
  asm("v_fma_mix_f32 %0, %1, %2, %2 op_sel:[0,1,0] op_sel_hi:[0,1,1]" : "=v"(ph) : "v"(f), "v"(e));   // f * slope(hi) + value(lo)
  return x * ph;
}

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(unsigned long long* out, float seed) {
  float a[16];
  f2 p[16];
  __shared__ unsigned tbl[TBL_N + 1];
  if (OP == 9 || OP == 18) {
    for (int i = threadIdx.x; i <= TBL_N; i += 256) {
      const float x0 = -4.4f + 8.8f * i / TBL_N, x1 = x0 + 8.8f / TBL_N;
      const float v0 = 0.5f * (1.f + erff(x0 * 0.70710678f)), v1 = 0.5f * (1.f + erff(x1 * 0.70710678f));
      const _Float16 hv = (_Float16)v0, hs = (_Float16)(v1 - v0);
      tbl[i] = (unsigned)__builtin_bit_cast(unsigned short, hv) | ((unsigned)__builtin_bit_cast(unsigned short, hs) << 16);
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = seed + i + threadIdx.x;
    p[i] = f2{seed + i, seed - i};
    if (OP == 9) {                                  // per-lane arguments, spread like unit-variance activations
      const unsigned h = (threadIdx.x * 2654435761u + i * 40503u) >> 8;
      p[i] = f2{((h & 0xfff) / 4096.f - 0.5f) * 5.f, (((h >> 12) & 0xfff) / 4096.f - 0.5f) * 5.f};
    }
  }
  float k1 = 1.0001f + seed, k2 = 0.25f + seed;
  f2 q1 = {k1, k1}, q2 = {k2, k2};
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    if (OP == 0) {
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k1), "v"(k2));
      REP16(M)
#undef M
    } else if (OP == 1) {
#define M(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q1), "v"(q2));
      REP16(M)
#undef M
    } else if (OP == 2) {
#define M(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k1), "v"(k2));
      REP16(M)
#undef M
    } else if (OP == 3) {
#define M(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q1));
      REP16(M)
#undef M
    } else if (OP == 4) {
#define M(i) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(a[i]));
      REP16(M)
#undef M
    } else if (OP == 5) {
#define M(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
      REP16(M)
#undef M
    } else if (OP == 6) {
#define M(i) asm volatile("v_pk_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k1), "v"(k2));
      REP16(M)
#undef M
    } else if (OP == 10) {
#define M(i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a[i]) : "v"(k1));
      REP16(M)
#undef M
    } else if (OP == 11) {
#define M(i) asm volatile("v_pk_max_f16 %0, %0, %1" : "+v"(a[i]) : "v"(k1));
      REP16(M)
#undef M
    } else if (OP == 12) {
#define M(i) asm volatile("v_fma_f16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k1), "v"(k2));
      REP16(M)
#undef M
    } else if (OP == 13) {  // two products + accumulate per lane in one instruction (fp16 inputs, fp32 accumulator)
#define M(i) asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(a[i]) : "v"(k1), "v"(k2));
      REP16(M)
#undef M
    } else if (OP == 14) {  // does the transcendental unit run BESIDE the FMA pipe?  8 v_exp_f32 + 8 v_fma_f32, interleaved
#define M(i) if ((i) & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k1), "v"(k2));
      REP16(M)
#undef M
    } else if (OP == 15) {  // ... and beside the packed pipe: 8 v_exp_f32 + 8 v_pk_fma_f32
#define M(i) if ((i) & 1) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i])); else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(q1), "v"(q2));
      REP16(M)
#undef M
    } else if (OP == 16) {  // v_rcp_f32
#define M(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      REP16(M)
#undef M
    } else if (OP == 17) {  // fp16 -> fp32 mixed FMA (v_fma_mix_f32: fp16 or fp32 sources, fp32 result)
#define M(i) asm volatile("v_fma_mix_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(k1), "v"(k2));
      REP16(M)
#undef M
    } else if (OP == 7) {   // v_fma_f32 with the constants in SGPRs / literals as the compiler emits Horner steps
#define M(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(a[(i + 1) & 15]), "s"(k2));
      REP16(M)
#undef M
    } else if (OP == 9 || OP == 18) {   // LDS-table GELU, 16 pairs per iteration; the feedback keeps the arguments spread
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        const f2 z = {gelu_tbl(p[g][0], tbl), gelu_tbl(p[g][1], tbl)};
        p[g] = z * q2 + p[(g + 1) & 15] * q1;
      }
    } else if (OP == 8) {   // the upscaler's GELU, 4 pairs per call (13 instructions per pair), 4 calls per iteration
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f2 z[4] = {p[g * 4], p[g * 4 + 1], p[g * 4 + 2], p[g * 4 + 3]};
        csam_gelu_poly2_n<4>(z);
#pragma unroll
        for (int q = 0; q < 4; ++q) p[g * 4 + q] = z[q] + q1;
      }
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i][0] + p[i][1];
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = c1 - c0;
  if (s == -1.2345f) out[0] = 0;
}

template <int OP>
static void run(const char* name, double per_iter, const char* unit) {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  unsigned long long* d;
  (void)hipMalloc(&d, sizeof(unsigned long long) * n_cu * 4 * 4 * 4);
  printf("%-44s", name);
  for (int w = 1; w <= 4; ++w) {
    const int blocks = n_cu * w;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, 0.f);   // warm
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, 0.f);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double cyc = (double)h[h.size() / 2] / (ITER * per_iter);      // median wave
    printf("  w=%d: %6.2f /wave %6.2f /SIMD", w, cyc, cyc / w);
  }
  printf("   [cycles per %s]\n", unit);
  (void)hipFree(d);
}

int main() {
  printf("shader cycles (s_memtime) per instruction, 16 independent instructions per loop body, %d iterations; w = waves per SIMD\n", ITER);
  run<0>("v_fma_f32 (VGPR operands)", 16, "instruction");
  run<7>("v_fma_f32 (VGPR x VGPR + SGPR)", 16, "instruction");
  run<1>("v_pk_fma_f32", 16, "instruction");
  run<3>("v_pk_mul_f32", 16, "instruction");
  run<2>("v_med3_f32", 16, "instruction");
  run<4>("v_cvt_f16_f32", 16, "instruction");
  run<6>("v_pk_fma_f16", 16, "instruction");
  run<10>("v_pk_mul_f16", 16, "instruction");
  run<11>("v_pk_max_f16", 16, "instruction");
  run<12>("v_fma_f16", 16, "instruction");
  run<13>("v_dot2_f32_f16", 16, "instruction");
  run<17>("v_fma_mix_f32", 16, "instruction");
  run<5>("v_exp_f32", 16, "instruction");
  run<16>("v_rcp_f32", 16, "instruction");
  run<14>("8 v_exp_f32 + 8 v_fma_f32 interleaved", 16, "instruction");
  run<15>("8 v_exp_f32 + 8 v_pk_fma_f32 interleaved", 16, "instruction");
  run<8>("csam_gelu_poly2_n<4> (packed; + 1 pk_add)", 16, "PAIR of evaluations");
  run<9>("LDS-table GELU, per-lane arguments (+ 1 pk_fma + 1 pk_mul)", 16, "PAIR of evaluations");
  run<18>("LDS-table GELU, one argument for all lanes", 16, "PAIR of evaluations");
  return 0;
}
