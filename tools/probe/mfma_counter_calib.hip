// Calibration of SQ_VALU_MFMA_BUSY_CYCLES / SQ_INSTS_MFMA / SQ_BUSY_CYCLES on gfx950: kernels that
// issue a KNOWN number of back-to-back v_mfma_f32_16x16x32_f16 per wave, at 1, 2 and 4 waves per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O2 -o mfma_counter_calib mfma_counter_calib.hip
//   rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES --kernel-trace -d out -- ./mfma_counter_calib
// Each launch prints its wave count and MFMA count; dividing the counter by them gives the unit the counter charges.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int WPS>   // waves per SIMD
__global__ __launch_bounds__(256 * WPS) void mfma_only(float* out, int iters) {
  half8_t a = {1, 1, 1, 1, 1, 1, 1, 1}, b = {1, 0, 1, 0, 1, 0, 1, 0};
  floatx4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {            // 16 MFMAs per iteration, four independent accumulators
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
    }
  }
  if (c0[0] + c1[0] + c2[0] + c3[0] == -1.f) out[threadIdx.x] = c0[1];
}

// a VALU-only kernel of the same shape: what the MFMA counters read when no MFMA is issued
__global__ __launch_bounds__(256) void valu_only(float* out, int iters) {
  float x = threadIdx.x, y = 1.0001f;
  for (int i = 0; i < iters * 16; ++i) x = __builtin_fmaf(x, y, 0.5f);
  if (x == -1.f) out[threadIdx.x] = x;
}

template <int WPS>
static void run(float* d, int n_cu, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  mfma_only<WPS><<<n_cu, 256 * WPS>>>(d, 8);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  mfma_only<WPS><<<n_cu, 256 * WPS>>>(d, iters);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)n_cu * 4 * WPS, mfma = waves * iters * 16;
  printf("mfma_only<%d waves/SIMD>: grid %d x %d threads, %.0f waves, %.4g MFMA 16x16x32 f16 (%d per wave), %.3f ms -> %.1f TFLOP/s, "
         "%.2f ns per MFMA per SIMD\n", WPS, n_cu, 256 * WPS, waves, mfma, iters * 16, ms, mfma * 16384.0 / (ms * 1e-3) / 1e12,
         ms * 1e6 / (iters * 16.0 * WPS));
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int n_cu = p.multiProcessorCount;
  float* d;
  hipMalloc(&d, 4096);
  printf("device %s, %d CUs, clock %d kHz\n", p.name, n_cu, p.clockRate);
  const int iters = 20000;
  run<1>(d, n_cu, iters);
  run<2>(d, n_cu, iters);
  run<4>(d, n_cu, iters);
  valu_only<<<n_cu, 256>>>(d, iters);
  hipDeviceSynchronize();
  printf("valu_only: %d waves, 0 MFMA\n", n_cu * 4);
  return 0;
}
