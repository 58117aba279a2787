// csam_token_block_a / _b (round 4): the TOKEN side of a two-way decoder block for small prompt batches in two launches.
//
// Reference: segment_anything_cs/modeling/transformer.py:164-170 (token self-attention + norm1), :173-177 (the q projection and
// the out projection + norm2 around the token->image attention), :180-183 (MLP + norm3), :186-190 (the k / v projections of
// the image->token attention).
//
// The shipped EPS configuration decodes 32 prompts per batch, 16 batches one after the other.  At that size the token side
// -- 7 rows per prompt -- was 14 launches per layer (small GEMMs of 7-27 us on 2-8 workgroups, LayerNorms, the 7 x 7
// attention), every one a link of a serial chain, and beside the next frame's encoders every launch also queues behind the
// encoders' workgroups (profiles/r04_eps_batch_timeline.txt, r04_eps_pipeline_overlap.txt).  Here one 16-wave workgroup owns
// TWO prompts (14 token rows, one 16-row MFMA tile) and walks the whole token-side sequence with the activations in LDS:
//   A: [fp16(tokens)] -> qk / v projections -> 7 x 7 attention per head -> out projection (+ residual) -> norm1 -> q projection
//      of the token->image attention;
//   B: out projection of the token->image attention + residual -> norm2 -> MLP (256 -> 2048 ReLU -> 256) + residual -> norm3
//      -> k / v projections of the image->token attention [-> q projection of the NEXT token->image attention].
// Weights stream from L2 as MFMA A-fragments (16 output features x 32 inputs per instruction, 16-byte loads straight from the
// row-major [N][K] matrices, two fragment sets in flight per wave); the token rows are the B operand, read from LDS.
// Rounding points are those of the launch sequence it replaces: fp16 operands, fp32 accumulation in ascending K order, fp16
// outputs where the separate kernels wrote fp16, the LayerNorm expression of csam_layernorm_cast.
#include "csam_common.h"

namespace {

constexpr int TB_ROWS = 14;              // token rows of a workgroup: two prompts
constexpr int TB_WAVES = 16;
constexpr int TB_THREADS = TB_WAVES * 64;
constexpr int LD128 = 136, LD256 = 264, LD512 = 520, LD2048 = 2056;     // fp16 row pitches (+ 16 B: conflict-free fragment reads)
constexpr int LDY = 260;                 // fp32 row pitch

// out[token][n] = sum_k xs[token][k] * W[n][k] for the n-tiles nt = wave, wave + 16, ...; epi(nt, acc) receives
// acc[r] = out[token fr][nt * 16 + fg * 4 + r].  K in {128, 256, 2048}; LDW = row pitch of W.
template <int K, int LDX, int LDW = K, class Epi>
__device__ __forceinline__ void linear16(const half_t* __restrict__ W, int N, const half_t* xs, int wave, int lane, Epi epi) {
  constexpr int KS = K < 256 ? K / 32 : 8;            // k-steps per fragment set
  constexpr int NCH = K / (KS * 32);                  // fragment sets per n-tile
  const int fr = lane & 15, fg = lane >> 4;
  const int ntw = (N / 16 - wave + TB_WAVES - 1) / TB_WAVES;      // n-tiles of this wave
  const int total = ntw > 0 ? ntw * NCH : 0;
  if (total == 0) return;
  half8_t wa[KS], wb[KS];
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  auto issue = [&](half8_t (&w)[KS], int it) {
    const int nt = wave + (it / NCH) * TB_WAVES, c = it % NCH;
    const half_t* src = W + (long)(nt * 16 + fr) * LDW + c * (KS * 32) + fg * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) w[ks] = *(const half8_t*)(src + ks * 32);
  };
  auto consume = [&](half8_t (&w)[KS], int it) {
    const int nt = wave + (it / NCH) * TB_WAVES, c = it % NCH;
    if (c == 0) acc = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8_t xf = *(const half8_t*)(xs + fr * LDX + c * (KS * 32) + ks * 32 + fg * 8);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks], xf, acc, 0, 0, 0);
    }
    if (c == NCH - 1) epi(nt, acc);
  };
  issue(wa, 0);
  for (int it = 0; it < total; it += 2) {
    if (it + 1 < total) issue(wb, it + 1);
    consume(wa, it);
    if (it + 2 < total) issue(wa, it + 2);
    if (it + 1 < total) consume(wb, it + 1);
  }
}

// LayerNorm of LDS row `wave` (fp32 [256]) in csam_layernorm_cast's arithmetic: lane = 4 consecutive channels
__device__ __forceinline__ floatx4 ln_row(const float* y, const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float eps, int lane) {
  const floatx4 v = *(const floatx4*)(y + lane * 4);
  const float s = v[0] + v[1] + v[2] + v[3];
  const float mean = csam_wave_sum(s) / 256.f;
  float q = 0.f;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float d = v[e] - mean;
    q += d * d;
  }
  const float rstd = rsqrtf(csam_wave_sum(q) / 256.f + eps);
  const floatx4 g = *(const floatx4*)(gamma + lane * 4);
  const floatx4 b = *(const floatx4*)(beta + lane * 4);
  floatx4 o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * g[e] + b[e];
  return o;
}

__device__ __forceinline__ half4_t to_half4(floatx4 v) {
  return half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
}

struct TokA {
  const half_t* src_qk; const half_t* src_v; const float* tokens0; int from_tokens; const float* residual;
  const half_t* qk_w; const float* qk_b; const half_t* v_w; const float* v_b; const half_t* o_w; const float* o_b;
  const float* norm_g; const float* norm_b; float eps; const half_t* q_w; const float* q_b;
  float* queries; half_t* q16; half_t* qpe16; half_t* t2i_q; int M7;
};

constexpr int A_XP = 0;                                  // [16][LD256] fp16: qk operand, later fp16(queries + pe)
constexpr int A_XQ = A_XP + 16 * LD256 * 2;              // [16][LD256] fp16: v operand
constexpr int A_QK = A_XQ + 16 * LD256 * 2;              // [16][LD512] fp16: q | k
constexpr int A_VV = A_QK + 16 * LD512 * 2;              // [16][LD256] fp16: v, later the attention output
constexpr int A_AO = A_VV + 16 * LD256 * 2;              // [16][LD256] fp16
constexpr int A_Y = A_AO + 16 * LD256 * 2;               // [16][LDY] fp32
constexpr int A_SMEM = A_Y + 16 * LDY * 4;

__global__ __launch_bounds__(TB_THREADS) void token_block_a_kernel(TokA p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* xp = (half_t*)(smem + A_XP);
  half_t* xq = (half_t*)(smem + A_XQ);
  half_t* qk = (half_t*)(smem + A_QK);
  half_t* vv = (half_t*)(smem + A_VV);
  half_t* ao = (half_t*)(smem + A_AO);
  float* y = (float*)(smem + A_Y);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int row0 = blockIdx.x * TB_ROWS;
  const int nrow = min(TB_ROWS, p.M7 - row0);            // 14, or 7 in the last workgroup of an odd batch
  // ---- operands: rows >= nrow are zero
  for (int i = tid; i < 16 * 32; i += TB_THREADS) {       // 16 rows x 32 chunks of 8 channels
    const int r = i >> 5, c = (i & 31) * 8;
    half8_t a = {0, 0, 0, 0, 0, 0, 0, 0}, b = a;
    if (r < nrow) {
      const long off = (long)(row0 + r) * 256 + c;
      if (p.from_tokens) {                                 // layer 0: fp16(tokens) is both operands (transformer.py:164-166)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = (half_t)p.tokens0[off + e];
        b = a;
      } else {
        a = *(const half8_t*)(p.src_qk + off);
        b = *(const half8_t*)(p.src_v + off);
      }
    }
    *(half8_t*)(xp + r * LD256 + c) = a;
    *(half8_t*)(xq + r * LD256 + c) = b;
  }
  __syncthreads();
  // ---- q | k and v projections (fp16 outputs, as csam_gemm_f16 wrote them)
  linear16<256, LD256>(p.qk_w, 512, xp, wave, lane, [&](int nt, floatx4 acc) {
    const floatx4 b = *(const floatx4*)(p.qk_b + nt * 16 + fg * 4);
    *(half4_t*)(qk + fr * LD512 + nt * 16 + fg * 4) = to_half4(acc + b);
  });
  linear16<256, LD256>(p.v_w, 256, xq, wave, lane, [&](int nt, floatx4 acc) {
    const floatx4 b = *(const floatx4*)(p.v_b + nt * 16 + fg * 4);
    *(half4_t*)(vv + fr * LD256 + nt * 16 + fg * 4) = to_half4(acc + b);
  });
  __syncthreads();
  // ---- 7 x 7 attention per (prompt, head): thread = (prompt, head, query), csam_token_self_attn's arithmetic
  if (tid < 112) {
    const int pr = tid / 56, t = tid % 56, h = t / 7, qi = t % 7;
    const int r = pr * 7 + qi;
    if (r < nrow) {
      const half_t* qrow = qk + r * LD512 + h * 32;
      float q[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) q[c] = (float)qrow[c];
      float s[7], mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const half_t* krow = qk + (pr * 7 + j) * LD512 + 256 + h * 32;
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 32; ++c) a += q[c] * (float)krow[c];
        s[j] = a * 0.17677669529663687f;
        mx = fmaxf(mx, s[j]);
      }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        s[j] = __expf(s[j] - mx);
        sum += s[j];
      }
      const float inv = 1.f / sum;
      float o[32];
#pragma unroll
      for (int c = 0; c < 32; ++c) o[c] = 0.f;
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const half_t* vrow = vv + (pr * 7 + j) * LD256 + h * 32;
        const float pj = s[j] * inv;
#pragma unroll
        for (int c = 0; c < 32; ++c) o[c] += pj * (float)vrow[c];
      }
      half_t* orow = ao + r * LD256 + h * 32;
#pragma unroll
      for (int c = 0; c < 32; ++c) orow[c] = (half_t)o[c];
    }
  } else if (tid >= 128 && tid < 128 + 2 * 32 && nrow < 16) {
    // rows nrow .. 15 of the attention output are operands of the next product: zero them
    const int i = tid - 128;
    for (int r = nrow; r < 16; ++r) *(half8_t*)(ao + r * LD256 + (i & 31) * 8) = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
  }
  __syncthreads();
  // ---- out projection (+ residual) -> norm1
  linear16<256, LD256>(p.o_w, 256, ao, wave, lane, [&](int nt, floatx4 acc) {
    floatx4 v = acc + *(const floatx4*)(p.o_b + nt * 16 + fg * 4);
    if (p.residual && fr < nrow) v += *(const floatx4*)(p.residual + (long)(row0 + fr) * 256 + nt * 16 + fg * 4);
    *(floatx4*)(y + fr * LDY + nt * 16 + fg * 4) = v;
  });
  __syncthreads();
  {
    const int r = wave;                                    // 16 waves, 16 rows
    const floatx4 o = ln_row(y + r * LDY, p.norm_g, p.norm_b, p.eps, lane);
    half4_t hp = {0, 0, 0, 0};
    if (r < nrow) {
      const long off = (long)(row0 + r) * 256 + lane * 4;
      *(floatx4*)(p.queries + off) = o;
      *(half4_t*)(p.q16 + off) = to_half4(o);
      hp = to_half4(o + *(const floatx4*)(p.tokens0 + off));
      *(half4_t*)(p.qpe16 + off) = hp;
    }
    *(half4_t*)(xp + r * LD256 + lane * 4) = hp;
  }
  __syncthreads();
  // ---- q projection of the token->image attention
  linear16<256, LD256>(p.q_w, 128, xp, wave, lane, [&](int nt, floatx4 acc) {
    if (fr < nrow)
      *(half4_t*)(p.t2i_q + (long)(row0 + fr) * 128 + nt * 16 + fg * 4) = to_half4(acc + *(const floatx4*)(p.q_b + nt * 16 + fg * 4));
  });
}

struct TokB {
  const half_t* attn_o; float* queries; const float* tokens0;
  const half_t* o_w; const float* o_b; const float* n2_g; const float* n2_b;
  const half_t* m1_w; const float* m1_b; const half_t* m2_w; const float* m2_b; const float* n3_g; const float* n3_b;
  const half_t* k_w; const float* k_b; const half_t* v_w; const float* v_b; const half_t* q_w; const float* q_b; float eps;
  half_t* q16; half_t* qpe16; half_t* i2t_k; half_t* i2t_v; half_t* t2i_q; int M7;
};

constexpr int B_XO = 0;                                  // [16][LD128] fp16: attention output
constexpr int B_Y = B_XO + 16 * LD128 * 2;               // [16][LDY] fp32
constexpr int B_XQ = B_Y + 16 * LDY * 4;                 // [16][LD256] fp16(queries)
constexpr int B_XP = B_XQ + 16 * LD256 * 2;              // [16][LD256] fp16(queries + pe)
constexpr int B_H = B_XP + 16 * LD256 * 2;               // [16][LD2048] fp16: MLP hidden
constexpr int B_SMEM = B_H + 16 * LD2048 * 2;

__global__ __launch_bounds__(TB_THREADS) void token_block_b_kernel(TokB p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* xo = (half_t*)(smem + B_XO);
  float* y = (float*)(smem + B_Y);
  half_t* xq = (half_t*)(smem + B_XQ);
  half_t* xp = (half_t*)(smem + B_XP);
  half_t* hid = (half_t*)(smem + B_H);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int row0 = blockIdx.x * TB_ROWS;
  const int nrow = min(TB_ROWS, p.M7 - row0);
  if (tid < 16 * 16) {                                     // 16 rows x 16 chunks of 8 channels
    const int r = tid >> 4, c = (tid & 15) * 8;
    half8_t a = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < nrow) a = *(const half8_t*)(p.attn_o + (long)(row0 + r) * 128 + c);
    *(half8_t*)(xo + r * LD128 + c) = a;
  }
  __syncthreads();
  // ---- out projection of the token->image attention + residual -> norm2
  linear16<128, LD128>(p.o_w, 256, xo, wave, lane, [&](int nt, floatx4 acc) {
    floatx4 v = acc + *(const floatx4*)(p.o_b + nt * 16 + fg * 4);
    if (fr < nrow) v += *(const floatx4*)(p.queries + (long)(row0 + fr) * 256 + nt * 16 + fg * 4);
    *(floatx4*)(y + fr * LDY + nt * 16 + fg * 4) = v;
  });
  __syncthreads();
  {
    const int r = wave;
    const floatx4 o = ln_row(y + r * LDY, p.n2_g, p.n2_b, p.eps, lane);
    *(floatx4*)(y + r * LDY + lane * 4) = o;               // the residual of the MLP
    *(half4_t*)(xq + r * LD256 + lane * 4) = r < nrow ? to_half4(o) : half4_t{0, 0, 0, 0};
  }
  __syncthreads();
  // ---- MLP: 256 -> 2048 (ReLU, fp16 as csam_gemm_f16 wrote it) -> 256 + residual.  (Cutting the hidden layer into quarters
  // over four workgroups, the last arrival adding the partial products, was tried: 68 -> 53 us per launch, nothing end to end,
  // and the reordered sum moves fp16 roundings -- this form is bit-identical to the launch sequence it replaces.)
  linear16<256, LD256>(p.m1_w, 2048, xq, wave, lane, [&](int nt, floatx4 acc) {
    const floatx4 b = *(const floatx4*)(p.m1_b + nt * 16 + fg * 4);
    floatx4 v = acc + b;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    *(half4_t*)(hid + fr * LD2048 + nt * 16 + fg * 4) = to_half4(v);
  });
  __syncthreads();
  linear16<2048, LD2048>(p.m2_w, 256, hid, wave, lane, [&](int nt, floatx4 acc) {
    float* d = y + fr * LDY + nt * 16 + fg * 4;            // this lane is the only reader and writer of these four values
    *(floatx4*)d = acc + *(const floatx4*)(p.m2_b + nt * 16 + fg * 4) + *(const floatx4*)d;
  });
  __syncthreads();
  {
    const int r = wave;
    const floatx4 o = ln_row(y + r * LDY, p.n3_g, p.n3_b, p.eps, lane);
    half4_t hq = {0, 0, 0, 0}, hp = hq;
    if (r < nrow) {
      const long off = (long)(row0 + r) * 256 + lane * 4;
      *(floatx4*)(p.queries + off) = o;
      hq = to_half4(o);
      hp = to_half4(o + *(const floatx4*)(p.tokens0 + off));
      *(half4_t*)(p.q16 + off) = hq;
      *(half4_t*)(p.qpe16 + off) = hp;
    }
    *(half4_t*)(xq + r * LD256 + lane * 4) = hq;
    *(half4_t*)(xp + r * LD256 + lane * 4) = hp;
  }
  __syncthreads();
  // ---- k / v projections of the image->token attention [+ the q projection of the next token->image attention]
  linear16<256, LD256>(p.k_w, 128, xp, wave, lane, [&](int nt, floatx4 acc) {
    if (fr < nrow)
      *(half4_t*)(p.i2t_k + (long)(row0 + fr) * 128 + nt * 16 + fg * 4) = to_half4(acc + *(const floatx4*)(p.k_b + nt * 16 + fg * 4));
  });
  linear16<256, LD256>(p.v_w, 128, xq, wave, lane, [&](int nt, floatx4 acc) {
    if (fr < nrow)
      *(half4_t*)(p.i2t_v + (long)(row0 + fr) * 128 + nt * 16 + fg * 4) = to_half4(acc + *(const floatx4*)(p.v_b + nt * 16 + fg * 4));
  });
  if (p.q_w) {
    linear16<256, LD256>(p.q_w, 128, xp, wave, lane, [&](int nt, floatx4 acc) {
      if (fr < nrow)
        *(half4_t*)(p.t2i_q + (long)(row0 + fr) * 128 + nt * 16 + fg * 4) = to_half4(acc + *(const floatx4*)(p.q_b + nt * 16 + fg * 4));
    });
  }
}

}  // namespace

extern "C" int csam_token_block_a(void* stream, const void* src_qk_f16, const void* src_v_f16, const float* tokens0,
                                  int from_tokens, const float* residual_or_null, const void* qk_w_f16, const float* qk_b,
                                  const void* v_w_f16, const float* v_b, const void* o_w_f16, const float* o_b,
                                  const float* norm_g, const float* norm_b, float eps, const void* q_w_f16, const float* q_b,
                                  float* queries, void* q16, void* qpe16, void* t2i_q_f16, int B) {
  CSAM_REQUIRE(tokens0 && qk_w_f16 && qk_b && v_w_f16 && v_b && o_w_f16 && o_b && norm_g && norm_b && q_w_f16 && q_b &&
                   queries && q16 && qpe16 && t2i_q_f16 && B > 0 && (from_tokens || (src_qk_f16 && src_v_f16)),
               "csam_token_block_a: bad args");
  TokA a;
  a.src_qk = (const half_t*)src_qk_f16; a.src_v = (const half_t*)src_v_f16; a.tokens0 = tokens0; a.from_tokens = from_tokens;
  a.residual = residual_or_null; a.qk_w = (const half_t*)qk_w_f16; a.qk_b = qk_b; a.v_w = (const half_t*)v_w_f16; a.v_b = v_b;
  a.o_w = (const half_t*)o_w_f16; a.o_b = o_b; a.norm_g = norm_g; a.norm_b = norm_b; a.eps = eps;
  a.q_w = (const half_t*)q_w_f16; a.q_b = q_b; a.queries = queries; a.q16 = (half_t*)q16; a.qpe16 = (half_t*)qpe16;
  a.t2i_q = (half_t*)t2i_q_f16; a.M7 = B * 7;
  static csam_once_t once;
  if (csam_first_call(once))
    (void)hipFuncSetAttribute((const void*)token_block_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A_SMEM);
  hipLaunchKernelGGL(token_block_a_kernel, dim3(csam_cdiv(B, 2)), dim3(TB_THREADS), A_SMEM, (hipStream_t)stream, a);
  CSAM_LAUNCH_CHECK("csam_token_block_a");
  return CSAM_OK;
}

extern "C" int csam_token_block_b(void* stream, const void* attn_o_f16, float* queries, const float* tokens0,
                                  const void* o_w_f16, const float* o_b, const float* norm2_g, const float* norm2_b,
                                  const void* mlp1_w_f16, const float* mlp1_b, const void* mlp2_w_f16, const float* mlp2_b,
                                  const float* norm3_g, const float* norm3_b, const void* k_w_f16, const float* k_b,
                                  const void* v_w_f16, const float* v_b, const void* next_q_w_f16_or_null,
                                  const float* next_q_b_or_null, float eps, void* q16, void* qpe16, void* i2t_k_f16,
                                  void* i2t_v_f16, void* t2i_q_f16_or_null, int B) {
  CSAM_REQUIRE(attn_o_f16 && queries && tokens0 && o_w_f16 && o_b && norm2_g && norm2_b && mlp1_w_f16 && mlp1_b && mlp2_w_f16 &&
                   mlp2_b && norm3_g && norm3_b && k_w_f16 && k_b && v_w_f16 && v_b && q16 && qpe16 && i2t_k_f16 && i2t_v_f16 &&
                   B > 0 && (!next_q_w_f16_or_null || (next_q_b_or_null && t2i_q_f16_or_null)),
               "csam_token_block_b: bad args");
  TokB a;
  a.attn_o = (const half_t*)attn_o_f16; a.queries = queries; a.tokens0 = tokens0; a.o_w = (const half_t*)o_w_f16; a.o_b = o_b;
  a.n2_g = norm2_g; a.n2_b = norm2_b; a.m1_w = (const half_t*)mlp1_w_f16; a.m1_b = mlp1_b; a.m2_w = (const half_t*)mlp2_w_f16;
  a.m2_b = mlp2_b; a.n3_g = norm3_g; a.n3_b = norm3_b; a.k_w = (const half_t*)k_w_f16; a.k_b = k_b;
  a.v_w = (const half_t*)v_w_f16; a.v_b = v_b; a.q_w = (const half_t*)next_q_w_f16_or_null; a.q_b = next_q_b_or_null;
  a.eps = eps; a.q16 = (half_t*)q16; a.qpe16 = (half_t*)qpe16; a.i2t_k = (half_t*)i2t_k_f16; a.i2t_v = (half_t*)i2t_v_f16;
  a.t2i_q = (half_t*)t2i_q_f16_or_null; a.M7 = B * 7;
  static csam_once_t once;
  if (csam_first_call(once))
    (void)hipFuncSetAttribute((const void*)token_block_b_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, B_SMEM);
  hipLaunchKernelGGL(token_block_b_kernel, dim3(csam_cdiv(B, 2)), dim3(TB_THREADS), B_SMEM, (hipStream_t)stream, a);
  CSAM_LAUNCH_CHECK("csam_token_block_b");
  return CSAM_OK;
}
