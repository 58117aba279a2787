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
// Weights stream from L2 as MFMA A-fragments (16 output features x 32 inputs per instruction).  They are stored in FRAGMENT
// ORDER -- [N / 16][K / 32][lane = 16 (k % 32 / 8) + n % 16][8] instead of row-major [N][K], a plan-time permutation
// (hip.frag_order) -- so that a wave's load is one contiguous KB; two fragment sets are in flight per wave.  The token rows
// are the B operand, read from LDS.
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
    // fragment order: [n-tile][k-step][lane][8] -- a wave's load is 1 KB contiguous
    const half_t* src = W + (((long)nt * (LDW / 32) + c * KS) * 64 + lane) * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) w[ks] = *(const half8_t*)(src + ks * 512);
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

// the attention output of rows 0 .. nrow-1 into xo [16][LD128]: either the merged fp16 rows, or merged here from
// csam_t2i_fused's partial records (t2i_merge_kernel's arithmetic: thread = (prompt, head, query))
__device__ __forceinline__ void load_attn(half_t* xo, const half_t* attn_o, const float* part, int nparts, int row0, int nrow,
                                          int tid) {
  if (part) {
    if (tid < 112) {
      const int pr = tid / 56, t = tid % 56, h = t / 7, qi = t % 7, r = pr * 7 + qi;
      if (r < nrow) {
        const int b = row0 / 7 + pr;
        float m = -INFINITY, l = 0.f, acc[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) acc[d] = 0.f;
        for (int q = 0; q < nparts; ++q) {
          const float* src = part + ((((long)b * nparts + q) * 8 + h) * 7 + qi) * 18;
          const float mo = src[0], lo = src[1];
          const float mn = fmaxf(m, mo);
          const float a = csam_exp2(m - mn), bb = csam_exp2(mo - mn);
          l = l * a + lo * bb;
#pragma unroll
          for (int d = 0; d < 16; ++d) acc[d] = acc[d] * a + src[2 + d] * bb;
          m = mn;
        }
        const float inv = 1.f / l;
#pragma unroll
        for (int d = 0; d < 16; ++d) xo[r * LD128 + h * 16 + d] = (half_t)(acc[d] * inv);
      }
    } else if (tid >= 128 && tid < 128 + 16) {          // rows nrow .. 15: zero
      for (int r = nrow; r < 16; ++r) *(half8_t*)(xo + r * LD128 + (tid - 128) * 8) = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
    }
  } else if (tid < 16 * 16) {                            // 16 rows x 16 chunks of 8 channels
    const int r = tid >> 4, c = (tid & 15) * 8;
    half8_t a = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < nrow) a = *(const half8_t*)(attn_o + (long)(row0 + r) * 128 + c);
    *(half8_t*)(xo + r * LD128 + c) = a;
  }
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
  const half_t* attn_o; const float* part; int nparts; float* queries; const float* tokens0;
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
  load_attn(xo, p.attn_o, p.part, p.nparts, row0, nrow, tid);
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


// ---------------------------------------------------------------------------------------------------------------
// csam_token_heads: everything between the final token->image attention and the upscaler / PWD-Net branch, for small batches:
// out projection + residual + final LayerNorm (transformer.py:105-112), the four hyper-network MLPs (mask_decoder.py:175-179),
// the IoU head (:184) and Crowd-SAM's parallel residual IoU head (:194-198).  13 launches in one.  Per pair of prompts the first
// and second layers of the six MLPs are six 256-wide products each: every wave owns one 16-feature tile of each, and walks its
// 7 / 6 fragment sets back to back (two in flight).  The third layers stay fp32 on the VALU in the summation order of
// csam_linear_f32[_batched] (they feed discrete decisions): IoU outputs bit-identical to the launch sequence, hyper-network
// outputs to the last bit (that kernel's rows 6, 7 of 8 are scheduled without fma contraction).
// ---------------------------------------------------------------------------------------------------------------
struct TokH {
  const half_t* attn_o; const float* part; int nparts; const float* queries; const half_t* o_w; const float* o_b; const float* n_g; const float* n_b; float eps;
  const half_t* hw0; const float* hb0; const half_t* hw1; const float* hb1; const float* hw2; const float* hb2;
  const half_t* iw0; const float* ib0; const half_t* iw1; const float* ib1; const float* iw2; const float* ib2;
  const half_t* pw0; const float* pb0; const half_t* pw1; const float* pb1; const float* pw2; const float* pb2;
  float* hyper; float* iou0; float* res_iou; int M7; int B;
};

constexpr int H_XO = 0;                                  // [16][LD128] fp16
constexpr int H_Y = H_XO + 16 * LD128 * 2;               // [16][LDY] fp32
constexpr int H_HS = H_Y + 16 * LDY * 4;                 // [16][LD256] fp16: the final token state (rows 14, 15 zero)
constexpr int H_XF = H_HS + 16 * LD256 * 2;              // [8][LD512] fp16: [iou token | mask token l] of (prompt, l)
constexpr int H_ZERO = H_XF + 8 * LD512 * 2;             // [LD512] fp16 zeros: the operand row of unused tile rows
constexpr int H_L1 = H_ZERO + LD512 * 2;                 // first-layer outputs fp16: hyper [4][2] | iou [2] | par [8] rows of LD256
constexpr int H_L2 = H_L1 + 18 * LD256 * 2;              // second-layer outputs fp32: the same 18 rows of LDY
constexpr int H_I0 = H_L2 + 18 * LDY * 4;                // iou0 [2][4] fp32
constexpr int H_SMEM = H_I0 + 64;

struct HItem { const half_t* w; const half_t* x; int g; bool first, last; };

template <class Get, class Epi>
__device__ __forceinline__ void chunk_pass(int total, Get get, Epi epi) {
  half8_t wa[8], wb[8];
  floatx4 acc = {0.f, 0.f, 0.f, 0.f};
  auto issue = [&](half8_t (&w)[8], int it) {
    const HItem t = get(it);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w[ks] = *(const half8_t*)(t.w + ks * 512);
  };
  auto consume = [&](half8_t (&w)[8], int it) {
    const HItem t = get(it);
    if (t.first) acc = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const half8_t xf = *(const half8_t*)(t.x + ks * 32);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(w[ks], xf, acc, 0, 0, 0);
    }
    if (t.last) epi(t.g, acc);
  };
  issue(wa, 0);
  for (int it = 0; it < total; it += 2) {
    if (it + 1 < total) issue(wb, it + 1);
    consume(wa, it);
    if (it + 2 < total) issue(wa, it + 2);
    if (it + 1 < total) consume(wb, it + 1);
  }
}

__global__ __launch_bounds__(TB_THREADS) void token_heads_kernel(TokH p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* xo = (half_t*)(smem + H_XO);
  float* y = (float*)(smem + H_Y);
  half_t* hs = (half_t*)(smem + H_HS);
  half_t* xf = (half_t*)(smem + H_XF);
  half_t* zero = (half_t*)(smem + H_ZERO);
  half_t* l1 = (half_t*)(smem + H_L1);
  float* l2 = (float*)(smem + H_L2);
  float* i0 = (float*)(smem + H_I0);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int row0 = blockIdx.x * TB_ROWS;
  const int nrow = min(TB_ROWS, p.M7 - row0);
  const int npr = nrow / 7;                                // prompts of this workgroup: 2, or 1
  load_attn(xo, p.attn_o, p.part, p.nparts, row0, nrow, tid);
  if (tid >= 256 && tid < 256 + LD512 / 8) *(half8_t*)(zero + (tid - 256) * 8) = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();
  // ---- out projection of the final attention + residual -> final LayerNorm
  linear16<128, LD128>(p.o_w, 256, xo, wave, lane, [&](int nt, floatx4 acc) {
    floatx4 v = acc + *(const floatx4*)(p.o_b + nt * 16 + fg * 4);
    if (fr < nrow) v += *(const floatx4*)(p.queries + (long)(row0 + fr) * 256 + nt * 16 + fg * 4);
    *(floatx4*)(y + fr * LDY + nt * 16 + fg * 4) = v;
  });
  __syncthreads();
  {
    const int r = wave;
    const floatx4 o = ln_row(y + r * LDY, p.n_g, p.n_b, p.eps, lane);
    *(half4_t*)(hs + r * LD256 + lane * 4) = r < nrow ? to_half4(o) : half4_t{0, 0, 0, 0};
  }
  __syncthreads();
  // operand of the parallel head: row (prompt pr, mask l) = [token 0 | token 1 + l] of the prompt
  for (int i = tid; i < 8 * 64; i += TB_THREADS) {
    const int r = i >> 6, c = (i & 63) * 8, pr = r >> 2, l = r & 3;
    half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (pr < npr) v = *(const half8_t*)(hs + (pr * 7 + (c < 256 ? 0 : 1 + l)) * LD256 + (c & 255));
    *(half8_t*)(xf + r * LD512 + c) = v;
  }
  __syncthreads();
  // ---- first layers (fp16 out, ReLU): g = 0..3 hyper MLP l = g on token 1 + l, g = 4 IoU head on token 0, g = 5 parallel head
  // (K = 512: two fragment sets).  Tile row fr of product g is prompt fr (g < 5) / (prompt, mask) fr (g = 5); the rest reads zeros.
  {
    // weights in fragment order: tile `wave` of a [256][K] matrix starts at wave * (K / 32) * 512 halfs
    auto get = [&](int it) {
      HItem t;
      const int g = it < 6 ? it : 5, c = it == 6 ? 1 : 0;
      t.g = g; t.first = c == 0; t.last = g < 5 || c == 1;
      if (g < 4) {
        t.w = p.hw0 + (long)g * 256 * 256 + ((long)wave * 8 * 64 + lane) * 8;
        t.x = (fr < npr ? hs + (fr * 7 + 1 + g) * LD256 : zero) + fg * 8;
      } else if (g == 4) {
        t.w = p.iw0 + ((long)wave * 8 * 64 + lane) * 8;
        t.x = (fr < npr ? hs + (fr * 7) * LD256 : zero) + fg * 8;
      } else {
        t.w = p.pw0 + (((long)wave * 16 + c * 8) * 64 + lane) * 8;
        t.x = (fr < 8 ? xf + fr * LD512 + c * 256 : zero) + fg * 8;
      }
      return t;
    };
    chunk_pass(7, get, [&](int g, floatx4 acc) {
      const float* b = g < 4 ? p.hb0 + g * 256 : g == 4 ? p.ib0 : p.pb0;
      floatx4 v = acc + *(const floatx4*)(b + wave * 16 + fg * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      const int rows = g < 5 ? 2 : 8, base = g < 4 ? g * 2 : g == 4 ? 8 : 10;
      if (fr < rows) *(half4_t*)(l1 + (base + fr) * LD256 + wave * 16 + fg * 4) = to_half4(v);
    });
  }
  __syncthreads();
  // ---- second layers (fp32 out, ReLU)
  {
    auto get = [&](int it) {
      HItem t;
      t.g = it; t.first = t.last = true;
      const half_t* w = it < 4 ? p.hw1 + (long)it * 256 * 256 : it == 4 ? p.iw1 : p.pw1;
      t.w = w + ((long)wave * 8 * 64 + lane) * 8;
      const int rows = it < 5 ? 2 : 8, base = it < 4 ? it * 2 : it == 4 ? 8 : 10;
      t.x = (fr < rows ? l1 + (base + fr) * LD256 : zero) + fg * 8;
      return t;
    };
    chunk_pass(6, get, [&](int g, floatx4 acc) {
      const float* b = g < 4 ? p.hb1 + g * 256 : g == 4 ? p.ib1 : p.pb1;
      floatx4 v = acc + *(const floatx4*)(b + wave * 16 + fg * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      const int rows = g < 5 ? 2 : 8, base = g < 4 ? g * 2 : g == 4 ? 8 : 10;
      if (fr < rows) *(floatx4*)(l2 + (base + fr) * LDY + wave * 16 + fg * 4) = v;
    });
  }
  __syncthreads();
  // ---- third layers in fp32.  Hyper-network outputs: thread = (prompt, l, j), one accumulator, k ascending (csam_linear_f32_batched)
  if (tid < 256) {
    const int pr = tid >> 7, l = (tid >> 5) & 3, j = tid & 31;
    if (pr < npr) {
      const float* a = l2 + (l * 2 + pr) * LDY;
      const float* w = p.hw2 + ((long)l * 32 + j) * 256;
      float acc = 0.f;
      for (int k = 0; k < 256; k += 4) {
        const floatx4 wv = *(const floatx4*)(w + k);
        const floatx4 av = *(const floatx4*)(a + k);
        acc = __builtin_fmaf(av[0], wv[0], acc);
        acc = __builtin_fmaf(av[1], wv[1], acc);
        acc = __builtin_fmaf(av[2], wv[2], acc);
        acc = __builtin_fmaf(av[3], wv[3], acc);
      }
      p.hyper[((long)(blockIdx.x * 2 + pr)) * 128 + l * 32 + j] = acc + p.hb2[l * 32 + j];
    }
  } else if (wave >= 4 && wave < 6) {
    // IoU head output (4 per prompt): one wave per row, 4 consecutive k per lane, wave reduction (linear_f32_rowdot_kernel)
    const int pr = wave - 4;
    if (pr < npr) {
      const floatx4 av = *(const floatx4*)(l2 + (8 + pr) * LDY + lane * 4);
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const floatx4 wv = *(const floatx4*)(p.iw2 + n * 256 + lane * 4);
        float acc = 0.f;
        acc = fmaf(av[0], wv[0], acc);
        acc = fmaf(av[1], wv[1], acc);
        acc = fmaf(av[2], wv[2], acc);
        acc = fmaf(av[3], wv[3], acc);
        const float sum = csam_wave_sum(acc);
        if (lane == 0) {
          const float v = sum + p.ib2[n];
          i0[pr * 4 + n] = v;
          p.iou0[(long)(blockIdx.x * 2 + pr) * 4 + n] = v;
        }
      }
    }
  }
  __syncthreads();
  if (wave < 8) {                                          // parallel head: row (prompt, l), one output + the IoU head's as residual
    const int pr = wave >> 2, l = wave & 3;
    if (pr < npr) {
      const floatx4 av = *(const floatx4*)(l2 + (10 + wave) * LDY + lane * 4);
      const floatx4 wv = *(const floatx4*)(p.pw2 + lane * 4);
      float acc = 0.f;
      acc = fmaf(av[0], wv[0], acc);
      acc = fmaf(av[1], wv[1], acc);
      acc = fmaf(av[2], wv[2], acc);
      acc = fmaf(av[3], wv[3], acc);
      const float sum = csam_wave_sum(acc);
      if (lane == 0) p.res_iou[(long)(blockIdx.x * 2 + pr) * 4 + l] = (sum + p.pb2[0]) + i0[pr * 4 + l];
    }
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

extern "C" int csam_token_block_b(void* stream, const void* attn_o_f16, const float* t2i_partials_or_null, int nparts,
                                  float* queries, const float* tokens0,
                                  const void* o_w_f16, const float* o_b, const float* norm2_g, const float* norm2_b,
                                  const void* mlp1_w_f16, const float* mlp1_b, const void* mlp2_w_f16, const float* mlp2_b,
                                  const float* norm3_g, const float* norm3_b, const void* k_w_f16, const float* k_b,
                                  const void* v_w_f16, const float* v_b, const void* next_q_w_f16_or_null,
                                  const float* next_q_b_or_null, float eps, void* q16, void* qpe16, void* i2t_k_f16,
                                  void* i2t_v_f16, void* t2i_q_f16_or_null, int B) {
  CSAM_REQUIRE((attn_o_f16 || (t2i_partials_or_null && nparts > 0)) && queries && tokens0 && o_w_f16 && o_b && norm2_g && norm2_b && mlp1_w_f16 && mlp1_b && mlp2_w_f16 &&
                   mlp2_b && norm3_g && norm3_b && k_w_f16 && k_b && v_w_f16 && v_b && q16 && qpe16 && i2t_k_f16 && i2t_v_f16 &&
                   B > 0 && (!next_q_w_f16_or_null || (next_q_b_or_null && t2i_q_f16_or_null)),
               "csam_token_block_b: bad args");
  TokB a;
  a.attn_o = (const half_t*)attn_o_f16; a.part = t2i_partials_or_null; a.nparts = nparts; a.queries = queries; a.tokens0 = tokens0; a.o_w = (const half_t*)o_w_f16; a.o_b = o_b;
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

extern "C" int csam_token_heads(void* stream, const void* attn_o_f16, const float* t2i_partials_or_null, int nparts,
                                const float* queries, const void* o_w_f16, const float* o_b,
                                const float* norm_g, const float* norm_b, float eps, const void* hyper_w0_f16,
                                const float* hyper_b0, const void* hyper_w1_f16, const float* hyper_b1, const float* hyper_w2,
                                const float* hyper_b2, const void* iou_w0_f16, const float* iou_b0, const void* iou_w1_f16,
                                const float* iou_b1, const float* iou_w2, const float* iou_b2, const void* par_w0_f16,
                                const float* par_b0, const void* par_w1_f16, const float* par_b1, const float* par_w2,
                                const float* par_b2, float* hyper_out, float* iou0_out, float* res_iou_out, int B) {
  CSAM_REQUIRE((attn_o_f16 || (t2i_partials_or_null && nparts > 0)) && queries && o_w_f16 && o_b && norm_g && norm_b && hyper_w0_f16 && hyper_b0 && hyper_w1_f16 &&
                   hyper_b1 && hyper_w2 && hyper_b2 && iou_w0_f16 && iou_b0 && iou_w1_f16 && iou_b1 && iou_w2 && iou_b2 &&
                   par_w0_f16 && par_b0 && par_w1_f16 && par_b1 && par_w2 && par_b2 && hyper_out && iou0_out && res_iou_out &&
                   B > 0,
               "csam_token_heads: bad args");
  TokH a;
  a.attn_o = (const half_t*)attn_o_f16; a.part = t2i_partials_or_null; a.nparts = nparts; a.queries = queries;
  a.o_w = (const half_t*)o_w_f16; a.o_b = o_b; a.n_g = norm_g;
  a.n_b = norm_b; a.eps = eps; a.hw0 = (const half_t*)hyper_w0_f16; a.hb0 = hyper_b0; a.hw1 = (const half_t*)hyper_w1_f16;
  a.hb1 = hyper_b1; a.hw2 = hyper_w2; a.hb2 = hyper_b2; a.iw0 = (const half_t*)iou_w0_f16; a.ib0 = iou_b0;
  a.iw1 = (const half_t*)iou_w1_f16; a.ib1 = iou_b1; a.iw2 = iou_w2; a.ib2 = iou_b2; a.pw0 = (const half_t*)par_w0_f16;
  a.pb0 = par_b0; a.pw1 = (const half_t*)par_w1_f16; a.pb1 = par_b1; a.pw2 = par_w2; a.pb2 = par_b2; a.hyper = hyper_out;
  a.iou0 = iou0_out; a.res_iou = res_iou_out; a.M7 = B * 7; a.B = B;
  static csam_once_t once;
  if (csam_first_call(once))
    (void)hipFuncSetAttribute((const void*)token_heads_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, H_SMEM);
  hipLaunchKernelGGL(token_heads_kernel, dim3(csam_cdiv(B, 2)), dim3(TB_THREADS), H_SMEM, (hipStream_t)stream, a);
  CSAM_LAUNCH_CHECK("csam_token_heads");
  return CSAM_OK;
}
