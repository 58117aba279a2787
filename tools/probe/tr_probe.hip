// developer probe: lane mapping of ds_read_b64_tr_b16 (gfx950).  LDS holds element index i at position i (fp16 can hold
// integers up to 2048 exactly); every lane reads at byte address lane*8 + OFF and the 4 halves it receives are dumped.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 h4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int mode) {
  __shared__ __fp16 sm[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = (__fp16)(float)(i & 2047);
  __syncthreads();
  const int l = threadIdx.x;
  int byte;
  if (mode == 0) byte = l * 8;                                   // contiguous 8-B pieces
  else byte = (l & 15) / 4 * 512 + ((l & 15) & 3) * 8 + (l >> 4) * 2048;   // rows 512 B apart: row = (l&15)/4 (+4*(l>>4)), 4-col chunk (l&3)
  h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)((__attribute__((address_space(3))) char*)sm + byte));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)v[j];
}
int main() {
  float* d; hipMalloc(&d, 64 * 4 * 4);
  float h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %5.0f %5.0f %5.0f %5.0f\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
