// Generic-head-dim attention support (head_dim != 64, e.g. ViT-H's 80): gather / softmax / scatter kernels around the
// batched MFMA GEMM, i.e. the reference's own materialised formulation of Attention.forward
// (segment_anything_cs/modeling/image_encoder.py:224-240 with window_partition :243-289 and
// add_decomposed_rel_pos :325-361) instead of the fused head_dim-64 kernels:
//   csam_head_gather    qkv [4096, 3 D] -> per (window, head) group: Qs = q*scale, K  ([G][Tp][128], zero-padded dims)
//                       and V^T ([G][128][Tp]); window mode folds window_partition in, pad tokens of edge windows take
//                       the qkv bias (they are real keys, SURVEY.md trap 4), slots >= T_valid are zero
//   (GEMM)              T = Qs . relcat^T  (rel-pos tables),  S = Qs . K^T
//   csam_softmax_relpos P = softmax_k(S + (Th[q,kh] + Tw[q,kw]) / scale) over the T_valid keys, fp16
//   (GEMM)              O = P . V
//   csam_head_scatter   O [G][Tp][128] -> out [4096, D] (window_unpartition, pad rows dropped)
// A completeness path: correct for any head_dim <= 128 (multiple of 8), not tuned.
#include "csam_common.h"

namespace {

constexpr int GEN_HP = 128;     // padded head dim

// token of slot i of window w (14x14 windows over the 70x70 padded grid), or -1 for a pad position
__device__ __forceinline__ int win_token(int w, int i) {
  const int y = (w / 5) * 14 + i / 14, x = (w % 5) * 14 + i % 14;
  return (y < 64 && x < 64) ? y * 64 + x : -1;
}

__global__ __launch_bounds__(256) void head_gather_kernel(const half_t* __restrict__ qkv, const float* __restrict__ bias,
                                                          half_t* __restrict__ Qs, half_t* __restrict__ Kp,
                                                          half_t* __restrict__ VpT, int D, int nH, int hd, int Tp,
                                                          int Tvalid, int window, float scale) {
  const int g = blockIdx.y, h = g % nH, w = g / nH;
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4);          // slot in the group
  const int c8 = (threadIdx.x & 15) * 8;                       // 8 consecutive dims
  if (i >= Tp) return;
  float q[8], k[8], v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) q[e] = k[e] = v[e] = 0.f;
  if (i < Tvalid && c8 < hd) {
    const int tok = window ? win_token(w, i) : i;
    const int col = h * hd + c8;
    if (tok >= 0) {
      const half_t* row = qkv + (long)tok * 3 * D;
      const half8_t a = *(const half8_t*)(row + col), b = *(const half8_t*)(row + D + col),
                    c = *(const half8_t*)(row + 2 * D + col);
#pragma unroll
      for (int e = 0; e < 8; ++e) { q[e] = (float)a[e]; k[e] = (float)b[e]; v[e] = (float)c[e]; }
    } else {   // zero-padded token after LayerNorm: q, k, v are the bias (in the precision of the qkv GEMM output)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        q[e] = (float)(half_t)bias[col + e];
        k[e] = (float)(half_t)bias[D + col + e];
        v[e] = (float)(half_t)bias[2 * D + col + e];
      }
    }
  }
  half8_t qs, kk;
#pragma unroll
  for (int e = 0; e < 8; ++e) { qs[e] = (half_t)(q[e] * scale); kk[e] = (half_t)k[e]; }
  const long base = ((long)g * Tp + i) * GEN_HP + c8;
  *(half8_t*)(Qs + base) = qs;
  *(half8_t*)(Kp + base) = kk;
#pragma unroll
  for (int e = 0; e < 8; ++e) VpT[((long)g * GEN_HP + c8 + e) * Tp + i] = (half_t)v[e];
}

// one wave per (group, query row)
__global__ __launch_bounds__(256) void softmax_relpos_kernel(const float* __restrict__ S, const float* __restrict__ traw,
                                                             half_t* __restrict__ P, int Tp, int Tvalid, int side,
                                                             float inv_scale) {
  const int g = blockIdx.y;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (q >= Tp) return;
  half_t* prow = P + ((long)g * Tp + q) * Tp;
  if (q >= Tvalid) {
    for (int k = lane; k < Tp; k += 64) prow[k] = (half_t)0.f;
    return;
  }
  const float* srow = S + ((long)g * Tp + q) * Tp;
  const float* th = traw + ((long)g * Tp + q) * 256 + (q / side) + side - 1;          // Th[kh] = th[-kh]
  const float* tw = traw + ((long)g * Tp + q) * 256 + 128 + (q % side) + side - 1;    // Tw[kw] = tw[-kw]
  float mx = -INFINITY;
  for (int k = lane; k < Tvalid; k += 64) {
    const float v = srow[k] + (th[-(k / side)] + tw[-(k % side)]) * inv_scale;
    mx = fmaxf(mx, v);
  }
  mx = csam_wave_max(mx);
  float sum = 0.f;
  for (int k = lane; k < Tvalid; k += 64) {
    const float v = srow[k] + (th[-(k / side)] + tw[-(k % side)]) * inv_scale;
    sum += __expf(v - mx);
  }
  sum = csam_wave_sum(sum);
  const float inv = 1.f / sum;
  for (int k = lane; k < Tp; k += 64) {
    float pv = 0.f;
    if (k < Tvalid) pv = __expf(srow[k] + (th[-(k / side)] + tw[-(k % side)]) * inv_scale - mx) * inv;
    prow[k] = (half_t)pv;
  }
}

// The same softmax with a row held in registers (round 3: the three-pass form above spent 6 of ViT-H's 21 ms encoder on
// integer divisions and re-reads).  NPL = Tp / 64 scores per lane; a key's (kh, kw) does not depend on the query, so the
// two table offsets of a lane's keys are worked out once per wave -- or not at all on the 64 x 64 global grid, where key
// k = lane + 64 j is simply (kh, kw) = (j, lane).  One wave walks ROWS query rows.  Same operations in the same order as
// the three-pass kernel (per-lane partial sums over ascending k, then the wave reduction): results are bit-identical.
template <int NPL, bool GRID64, int ROWS>
__global__ __launch_bounds__(256) void softmax_relpos_reg_kernel(const float* __restrict__ S, const float* __restrict__ traw,
                                                                 half_t* __restrict__ P, int Tp, int Tvalid, int side,
                                                                 float inv_scale) {
  const int g = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int q0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
  int kh[GRID64 ? 1 : NPL], kw[GRID64 ? 1 : NPL];
  if constexpr (!GRID64) {
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int k = lane + 64 * j;
      kh[j] = k / side;
      kw[j] = k - kh[j] * side;
    }
  }
  for (int r = 0; r < ROWS; ++r) {
    const int q = q0 + r;
    if (q >= Tp) return;
    half_t* prow = P + ((long)g * Tp + q) * Tp;
    if (q >= Tvalid) {
#pragma unroll
      for (int j = 0; j < NPL; ++j) prow[lane + 64 * j] = (half_t)0.f;
      continue;
    }
    const float* srow = S + ((long)g * Tp + q) * Tp;
    const float* th = traw + ((long)g * Tp + q) * 256 + (q / side) + side - 1;
    const float* tw = traw + ((long)g * Tp + q) * 256 + 128 + (q % side) + side - 1;
    float v[NPL];
    float mx = -INFINITY;
    const float twl = GRID64 ? tw[-lane] : 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int k = lane + 64 * j;
      if (k < Tvalid) {
        const float b = GRID64 ? (th[-j] + twl) : (th[-kh[j]] + tw[-kw[j]]);
        v[j] = srow[k] + b * inv_scale;
        mx = fmaxf(mx, v[j]);
      } else {
        v[j] = -INFINITY;
      }
    }
    mx = csam_wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < NPL; ++j)
      if (lane + 64 * j < Tvalid) sum += __expf(v[j] - mx);
    sum = csam_wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int j = 0; j < NPL; ++j) {
      const int k = lane + 64 * j;
      prow[k] = (half_t)(k < Tvalid ? __expf(v[j] - mx) * inv : 0.f);
    }
  }
}

__global__ __launch_bounds__(256) void head_scatter_kernel(const half_t* __restrict__ Op, half_t* __restrict__ out, int D,
                                                           int nH, int hd, int Tp, int Tvalid, int window) {
  const int g = blockIdx.y, h = g % nH, w = g / nH;
  const int i = blockIdx.x * 16 + (threadIdx.x >> 4);
  const int c8 = (threadIdx.x & 15) * 8;
  if (i >= Tvalid || c8 >= hd) return;
  const int tok = window ? win_token(w, i) : i;
  if (tok < 0) return;
  *(half8_t*)(out + (long)tok * D + h * hd + c8) = *(const half8_t*)(Op + ((long)g * Tp + i) * GEN_HP + c8);
}

}  // namespace

extern "C" int csam_head_gather(void* stream, const void* qkv_f16, const float* qkv_bias, void* Qs_f16, void* K_f16,
                                void* VT_f16, int D, int nH, int head_dim, int Tp, int T_valid, int window, float scale) {
  CSAM_REQUIRE(qkv_f16 && qkv_bias && Qs_f16 && K_f16 && VT_f16, "csam_head_gather: null pointer");
  CSAM_REQUIRE(head_dim > 0 && head_dim <= GEN_HP && head_dim % 8 == 0 && D == nH * head_dim && Tp % 128 == 0 &&
               T_valid <= Tp, "csam_head_gather: bad shape D=%d nH=%d hd=%d Tp=%d", D, nH, head_dim, Tp);
  const int G = window ? nH * 25 : nH;
  hipLaunchKernelGGL(head_gather_kernel, dim3(Tp / 16, G), dim3(256), 0, (hipStream_t)stream, (const half_t*)qkv_f16,
                     qkv_bias, (half_t*)Qs_f16, (half_t*)K_f16, (half_t*)VT_f16, D, nH, head_dim, Tp, T_valid, window, scale);
  CSAM_LAUNCH_CHECK("csam_head_gather");
  return CSAM_OK;
}

extern "C" int csam_softmax_relpos(void* stream, const float* S, const float* relpos_raw, void* P_f16, int G, int Tp,
                                   int T_valid, int side, float inv_scale) {
  CSAM_REQUIRE(S && relpos_raw && P_f16 && G > 0 && G <= 65535 && Tp > 0 && T_valid <= Tp && side > 0,
               "csam_softmax_relpos: bad args");
  const bool three_pass = false;                         // the row-in-registers kernels where they exist (round 3)
  if (!three_pass && Tp == 256)                          // a padded 14 x 14 window: 4 scores per lane, 4 rows per wave
    hipLaunchKernelGGL((softmax_relpos_reg_kernel<4, false, 4>), dim3(csam_cdiv(Tp, 16), G), dim3(256), 0, (hipStream_t)stream, S,
                       relpos_raw, (half_t*)P_f16, Tp, T_valid, side, inv_scale);
  else if (!three_pass && Tp == 4096 && side == 64)      // the global 64 x 64 grid: 64 scores per lane
    hipLaunchKernelGGL((softmax_relpos_reg_kernel<64, true, 1>), dim3(csam_cdiv(Tp, 4), G), dim3(256), 0, (hipStream_t)stream, S,
                       relpos_raw, (half_t*)P_f16, Tp, T_valid, side, inv_scale);
  else
    hipLaunchKernelGGL(softmax_relpos_kernel, dim3(csam_cdiv(Tp, 4), G), dim3(256), 0, (hipStream_t)stream, S, relpos_raw,
                       (half_t*)P_f16, Tp, T_valid, side, inv_scale);
  CSAM_LAUNCH_CHECK("csam_softmax_relpos");
  return CSAM_OK;
}

extern "C" int csam_head_scatter(void* stream, const void* O_f16, void* out_f16, int D, int nH, int head_dim, int Tp,
                                 int T_valid, int window) {
  CSAM_REQUIRE(O_f16 && out_f16 && head_dim > 0 && head_dim <= GEN_HP && D == nH * head_dim, "csam_head_scatter: bad args");
  const int G = window ? nH * 25 : nH;
  hipLaunchKernelGGL(head_scatter_kernel, dim3(csam_cdiv(T_valid, 16), G), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)O_f16, (half_t*)out_f16, D, nH, head_dim, Tp, T_valid, window);
  CSAM_LAUNCH_CHECK("csam_head_scatter");
  return CSAM_OK;
}
