// Shader clock under load: one wave reads s_memtime (core clock cycles) and s_memrealtime (constant 100 MHz) around a spin.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o libclock_probe.so clock_probe.hip   (loaded with ctypes by
//   tools/dev_gemm_ingraph.py and launched on a second stream beside the kernels being timed)
#include <hip/hip_runtime.h>
__global__ void clock_probe_kernel(unsigned long long* out, int spin) {
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  float x = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) x = __builtin_fmaf(x, 1.0001f, 0.5f);
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[0] = c1 - c0;
    out[1] = r1 - r0;
  }
  if (x == -1.f) out[2] = 1;
}
extern "C" int clock_probe(void* stream, void* out_u64x3, int spin) {
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (unsigned long long*)out_u64x3, spin);
  return (int)hipGetLastError();
}
