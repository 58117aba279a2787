// How many bytes per cycle does ONE CU move L2 -> LDS by LDS-DMA (global_load_lds_dwordx4), and L2 -> registers by
// global_load_dwordx4, with 4 / 8 / 16 waves per CU streaming a 2 MB (L2-resident) buffer?  (round 4, HISTORY.md 4.2f: the
// 64- / 96- / 128-row GEMM kernel moves its operands at ~20 B/clk/CU however its waves are organised.)
//   hipcc --offload-arch=gfx950 -O3 -o load_rate tools/probe/load_rate.hip && ./load_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int ITER = 256;

// MODE 0: LDS-DMA, 8 pieces (8 KB per wave) per iteration into a ring;  MODE 1: 8 dwordx4 loads per lane into registers
template <int MODE>
__global__ void rate_kernel(const char* __restrict__ src, long bytes, unsigned long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
  const long stride = (long)nw * 8 * 1024;                     // bytes the workgroup consumes per iteration
  long off = ((long)blockIdx.x * 977 * 1024) % bytes + (long)wave * 8 * 1024 + lane * 16;
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITER; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
        __builtin_amdgcn_global_load_lds((gptr_t)(src + (off + i * 1024) % bytes), (lptr_t)(smem + (wave * 8 + i) * 1024), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // the previous iteration's pieces have landed
    } else {
      floatx4 v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *(const floatx4*)(src + (off + i * 1024) % bytes);
#pragma unroll
      for (int i = 0; i < 8; ++i) acc += v[i];
    }
    off += stride;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) out[blockIdx.x * 16 + wave] = c1 - c0;
  if (acc[0] == 123.456f) sink[0] = acc[1] + smem[tid];
}

template <int MODE>
static void run(const char* name, const char* src, long bytes) {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int n_cu = prop.multiProcessorCount;
  unsigned long long* d;
  float* sink;
  (void)hipMalloc(&d, sizeof(unsigned long long) * n_cu * 16);
  (void)hipMalloc(&sink, 64);
  printf("%-34s", name);
  for (int nw : {4, 8, 16}) {
    const int smem = nw * 8 * 1024;
    (void)hipFuncSetAttribute((const void*)rate_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    for (int rep = 0; rep < 2; ++rep)
      hipLaunchKernelGGL(rate_kernel<MODE>, dim3(n_cu), dim3(nw * 64), smem, 0, src, bytes, d, sink);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(n_cu * 16);
    (void)hipMemcpy(h.data(), d, sizeof(unsigned long long) * n_cu * 16, hipMemcpyDeviceToHost);
    std::vector<unsigned long long> w;
    for (int b = 0; b < n_cu; ++b)
      for (int i = 0; i < nw; ++i) w.push_back(h[b * 16 + i]);
    std::sort(w.begin(), w.end());
    const double cyc = (double)w[w.size() / 2];
    printf("  %2d waves/CU: %6.1f B/clk/CU", nw, (double)nw * 8 * 1024 * ITER / cyc);
  }
  printf("\n");
  (void)hipFree(d);
  (void)hipFree(sink);
}

int main() {
  const long bytes = 2 << 20;
  char* src;
  (void)hipMalloc(&src, bytes + (64 << 10));
  (void)hipMemset(src, 1, bytes + (64 << 10));
  printf("one workgroup per CU on every CU, each wave 8 x 1 KB per iteration from a 2 MB buffer (L2 / MALL resident), %d iterations\n", ITER);
  run<0>("LDS-DMA (global_load_lds_dwordx4)", src, bytes);
  run<1>("global_load_dwordx4 -> registers", src, bytes);
  return 0;
}
