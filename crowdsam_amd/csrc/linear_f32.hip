// csam_linear_f32: out[M,N] = act(A[M,K] * W[N,K]^T + bias) (+ residual), everything fp32.
//
// For the decoder's small heads, whose outputs drive discrete decisions (argmax over the 4 fused
// scores, score/stability thresholds) and therefore stay in fp32 on the VALU:
//   mask_decoder.py:175-179 hyper-MLPs (256->256->256->32), :184 IoU head (256->256->256->4),
//   :192 point_classifier (256->256->n_class), :194-198 parallel_iou_head (512->256->256->1),
//   predictor.py:113-121 FG prior.  Arbitrary M, N, K and row strides (token gathers via lda).
// 32x64 output tile per 256-thread block, BK = 32 staged through LDS.
#include "csam_common.h"

namespace {

constexpr int LBM = 32, LBN = 64, LBK = 32;

__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ A, long lda,
                                                         const float* __restrict__ W, long ldw,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ R, long ldr,
                                                         float* __restrict__ C, long ldc, int M, int N,
                                                         int K, int act, long sA, long sW, long sB, long sC) {
  A += (long)blockIdx.z * sA;
  W += (long)blockIdx.z * sW;
  if (bias) bias += (long)blockIdx.z * sB;
  C += (long)blockIdx.z * sC;
  __shared__ float As[LBM][LBK + 1];
  __shared__ float Ws[LBN][LBK + 1];
  const int tid = threadIdx.x;
  const int tx = tid & 63, ty = tid >> 6;  // col, row-group (8 rows each)
  const int m0 = blockIdx.y * LBM, n0 = blockIdx.x * LBN;
  float acc[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) acc[r] = 0.f;
  for (int k0 = 0; k0 < K; k0 += LBK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int e = tid + i * 256;
      const int r = e >> 5, c = e & 31;
      const int m = m0 + r, k = k0 + c;
      As[r][c] = (m < M && k < K) ? A[(long)m * lda + k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int e = tid + i * 256;
      const int r = e >> 5, c = e & 31;
      const int n = n0 + r, k = k0 + c;
      Ws[r][c] = (n < N && k < K) ? W[(long)n * ldw + k] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < LBK; ++kk) {
      const float w = Ws[tx][kk];
#pragma unroll
      for (int r = 0; r < 8; ++r) acc[r] += As[ty * 8 + r][kk] * w;
    }
    __syncthreads();
  }
  const int n = n0 + tx;
  if (n >= N) return;
  const float b = bias ? bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int m = m0 + ty * 8 + r;
    if (m < M) {
      float v = csam_apply_act(acc[r] + b, act);
      if (R) v += R[(long)m * ldr + n];
      C[(long)m * ldc + n] = v;
    }
  }
}

// N <= 4 (IoU head 256 -> 4, parallel IoU head -> 1, one-class point classifier): a 64-column tile would leave 60+ lanes
// of every wave idle and spend 8 barrier pairs per 32 rows.  One wave per output row instead: the row's K values spread
// over the lanes (4 consecutive per lane and step), N running sums per lane, one wave reduction at the end.
template <int NMAX>
__global__ __launch_bounds__(256) void linear_f32_rowdot_kernel(const float* __restrict__ A, long lda,
                                                                const float* __restrict__ W, long ldw,
                                                                const float* __restrict__ bias,
                                                                const float* __restrict__ R, long ldr,
                                                                float* __restrict__ C, long ldc, int M, int N, int K,
                                                                int act) {
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (m >= M) return;
  float acc[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
  const float* a = A + (long)m * lda;
  for (int k = lane * 4; k < K; k += 256) {            // K % 4 == 0, rows 16-B aligned (checked by the launcher)
    const floatx4 av = *(const floatx4*)(a + k);
#pragma unroll
    for (int n = 0; n < NMAX; ++n) {
      if (n < N) {
        const floatx4 wv = *(const floatx4*)(W + (long)n * ldw + k);
        acc[n] = fmaf(av[0], wv[0], acc[n]);
        acc[n] = fmaf(av[1], wv[1], acc[n]);
        acc[n] = fmaf(av[2], wv[2], acc[n]);
        acc[n] = fmaf(av[3], wv[3], acc[n]);
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    if (n < N) {
      const float sum = csam_wave_sum(acc[n]);
      if (lane == 0) {
        float v = csam_apply_act(sum + (bias ? bias[n] : 0.f), act);
        if (R) v += R[(long)m * ldr + n];
        C[(long)m * ldc + n] = v;
      }
    }
  }
}

}  // namespace

extern "C" int csam_linear_f32(void* stream, const float* A, long lda, const float* W, long ldw,
                               const float* bias, const float* residual, long ldr, float* C, long ldc,
                               int M, int N, int K, int act) {
  CSAM_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0, "csam_linear_f32: bad args");
  if (N <= 4 && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && ((unsigned long)A & 15) == 0 && ((unsigned long)W & 15) == 0) {
    hipLaunchKernelGGL(linear_f32_rowdot_kernel<4>, dim3(csam_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, A, lda, W,
                       ldw, bias, residual, ldr, C, ldc, M, N, K, act);
    CSAM_LAUNCH_CHECK("csam_linear_f32");
    return CSAM_OK;
  }
  dim3 grid(csam_cdiv(N, LBN), csam_cdiv(M, LBM));
  hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, lda, W, ldw, bias, residual,
                     ldr, C, ldc, M, N, K, act, 0L, 0L, 0L, 0L);
  CSAM_LAUNCH_CHECK("csam_linear_f32");
  return CSAM_OK;
}

// `batch` independent problems (grid.z) with element strides; no residual (the 4 hyper-MLP output layers).
extern "C" int csam_linear_f32_batched(void* stream, const float* A, long lda, long strideA, const float* W, long ldw,
                                       long strideW, const float* bias, long strideBias, float* C, long ldc,
                                       long strideC, int M, int N, int K, int act, int batch) {
  CSAM_REQUIRE(A && W && C && M > 0 && N > 0 && K > 0 && batch > 0, "csam_linear_f32_batched: bad args");
  dim3 grid(csam_cdiv(N, LBN), csam_cdiv(M, LBM), batch);
  hipLaunchKernelGGL(linear_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, lda, W, ldw, bias,
                     (const float*)nullptr, 0L, C, ldc, M, N, K, act, strideA, strideW, strideBias, strideC);
  CSAM_LAUNCH_CHECK("csam_linear_f32_batched");
  return CSAM_OK;
}
