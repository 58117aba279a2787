// Prompt-encoder + two-way-decoder kernels that are not plain GEMMs (all prompts of a batch at once).
//
// Reference: segment_anything_cs/modeling/prompt_encoder.py:75-93,189-218 (point PE),
// transformer.py:160-254 (self / token->image / image->token attention), mask_decoder.py:56-62,
// 172-198 (LayerNorm2d+GELU of the upscaler, hyper-network mask product, PWD-Net pooling).
//
// Layouts: tokens fp32 [B,7,256] (iou, mask0..3, point, not-a-point); per-prompt key state fp16
// [B,4096,256] token-major; attention operands fp16 with 8 heads x 16 dims packed as 128 columns.
// Round-1 structure: VALU kernels with packed-fp16 dot2 for QK; the GEMM-shaped parts of the
// decoder go through csam_gemm_f16.
#include "csam_common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;

// ---------------------------------------------------------------------------------------------
// Random-Fourier positional encoding (prompt_encoder.py:189-196) of points already in the
// 1024 frame: pe = [sin(2pi*c) | cos(2pi*c)], c = (2*(xy+0.5)/1024 - 1) @ G[2,128].
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float pe_value(float px, float py, const float* __restrict__ G, int ch) {
  const float cx = 2.f * ((px + 0.5f) / 1024.f) - 1.f;
  const float cy = 2.f * ((py + 0.5f) / 1024.f) - 1.f;
  const int j = ch & 127;
  float v = cx * G[j] + cy * G[128 + j];
  v = 6.283185307179586f * v;
  return ch < 128 ? sinf(v) : cosf(v);
}

// tokens[b] = [iou_token; mask_tokens(4); PE(point_b)+point_embed[1]; not_a_point]
// (mask_decoder.py:153-155, prompt_encoder.py:83-92 with label == 1 and the pad point label -1)
__global__ __launch_bounds__(256) void point_tokens_kernel(const float* __restrict__ coords,
                                                           const float* __restrict__ G,
                                                           const float* __restrict__ out_tokens5,
                                                           const float* __restrict__ point_embed1,
                                                           const float* __restrict__ not_a_point,
                                                           float* __restrict__ tokens) {
  const int b = blockIdx.x, ch = threadIdx.x;
  float* t = tokens + (long)b * 7 * 256;
#pragma unroll
  for (int r = 0; r < 5; ++r) t[r * 256 + ch] = out_tokens5[r * 256 + ch];
  t[5 * 256 + ch] = pe_value(coords[b * 2], coords[b * 2 + 1], G, ch) + point_embed1[ch];
  t[6 * 256 + ch] = not_a_point[ch];
}

// one point per prompt with an explicit label (prompt_encoder.py:88-92): 1 = foreground (point_embeddings[1]), 0 = background
// (point_embeddings[0]), -1 = "not a point" (the PE is zeroed and not_a_point_embed added); the padding token as above
__global__ __launch_bounds__(256) void point_tokens_labeled_kernel(const float* __restrict__ coords, const int* __restrict__ labels,
                                                                   const float* __restrict__ G,
                                                                   const float* __restrict__ out_tokens5,
                                                                   const float* __restrict__ point_embed0,
                                                                   const float* __restrict__ point_embed1,
                                                                   const float* __restrict__ not_a_point,
                                                                   float* __restrict__ tokens) {
  const int b = blockIdx.x, ch = threadIdx.x;
  float* t = tokens + (long)b * 7 * 256;
#pragma unroll
  for (int r = 0; r < 5; ++r) t[r * 256 + ch] = out_tokens5[r * 256 + ch];
  const int lb = labels[b];
  float v = lb == -1 ? 0.f : pe_value(coords[b * 2], coords[b * 2 + 1], G, ch);
  v += lb == -1 ? not_a_point[ch] : lb == 0 ? point_embed0[ch] : lb == 1 ? point_embed1[ch] : 0.f;
  t[5 * 256 + ch] = v;
  t[6 * 256 + ch] = not_a_point[ch];
}

// box prompts (prompt_encoder.py:95-102, :152-163 with points == None): tokens[b] = [iou_token; mask_tokens(4);
// PE(x0, y0) + point_embed[2]; PE(x1, y1) + point_embed[3]] -- two corner tokens and NO padding point, i.e. the same seven
// tokens per prompt the one-point form has, so every decoder kernel downstream serves box prompts unchanged.
__global__ __launch_bounds__(256) void box_tokens_kernel(const float* __restrict__ boxes, const float* __restrict__ G,
                                                         const float* __restrict__ out_tokens5,
                                                         const float* __restrict__ point_embed2,
                                                         const float* __restrict__ point_embed3, float* __restrict__ tokens) {
  const int b = blockIdx.x, ch = threadIdx.x;
  float* t = tokens + (long)b * 7 * 256;
#pragma unroll
  for (int r = 0; r < 5; ++r) t[r * 256 + ch] = out_tokens5[r * 256 + ch];
  t[5 * 256 + ch] = pe_value(boxes[b * 4], boxes[b * 4 + 1], G, ch) + point_embed2[ch];
  t[6 * 256 + ch] = pe_value(boxes[b * 4 + 2], boxes[b * 4 + 3], G, ch) + point_embed3[ch];
}

// pure PE rows for arbitrary points: used once per model for the dense 64x64 PE
// (prompt_encoder.py:64-73,198-209: grid point (i+0.5)/64 == pixel 16*i+7.5 in the 1024 frame)
__global__ __launch_bounds__(256) void pe_points_kernel(const float* __restrict__ coords,
                                                        const float* __restrict__ G, float* __restrict__ out) {
  const int p = blockIdx.x, ch = threadIdx.x;
  out[(long)p * 256 + ch] = pe_value(coords[p * 2], coords[p * 2 + 1], G, ch);
}

// ---------------------------------------------------------------------------------------------
// Token self-attention (transformer.py:164-169): 7 tokens, 8 heads x 32.  qk f16 [B*7, 512]
// (q | k projections), v f16 [B*7, 256] -> out f16 [B*7, 256].  One 64-thread block per prompt,
// thread = (head, query).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void token_self_attn_kernel(const half_t* __restrict__ qk,
                                                             const half_t* __restrict__ v,
                                                             half_t* __restrict__ out) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= 56) return;
  const int h = t / 7, qi = t % 7;
  const half_t* qrow = qk + ((long)b * 7 + qi) * 512 + h * 32;
  float q[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) q[c] = (float)qrow[c];
  float s[7], mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const half_t* krow = qk + ((long)b * 7 + j) * 512 + 256 + h * 32;
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) a += q[c] * (float)krow[c];
    s[j] = a * 0.17677669529663687f;  // 1/sqrt(32)
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
    const half_t* vrow = v + ((long)b * 7 + j) * 256 + h * 32;
    const float p = s[j] * inv;
#pragma unroll
    for (int c = 0; c < 32; ++c) o[c] += p * (float)vrow[c];
  }
  half_t* orow = out + ((long)b * 7 + qi) * 256 + h * 32;
#pragma unroll
  for (int c = 0; c < 32; ++c) orow[c] = (half_t)o[c];
}

// ---------------------------------------------------------------------------------------------
// token -> image attention (transformer.py:173-177, 105-112): 7 queries x T keys, 8 heads x 16.
// q f16 [B,7,128]; K,V f16 rows of 128 with row stride ldkv and per-prompt stride kv_bstride
// (0 when the layer-0 K/V of the shared image embedding are used).  Thread = (head, key slot),
// online softmax in chunks of 2 keys, (m,l,acc) merged across the wave by shuffles and across
// waves/splits through a partial buffer [B, nsplit*4, 8 heads, 7, 18] merged by t2i_merge_kernel.
// ---------------------------------------------------------------------------------------------
constexpr int T2I_REC = 18;  // m, l, acc[16]

__global__ __launch_bounds__(256) void attn_t2i_kernel(const half_t* __restrict__ q,
                                                       const half_t* __restrict__ K,
                                                       const half_t* __restrict__ V, long ldkv,
                                                       long kv_bstride, float* __restrict__ part, int T,
                                                       int nsplit) {
  const int b = blockIdx.x, split = blockIdx.y;
  const int tid = threadIdx.x, h = tid & 7, slot = tid >> 3;
  const int wave = tid >> 6;
  half2_t qh[7][8];
#pragma unroll
  for (int qi = 0; qi < 7; ++qi) {
    const half8_t a = *(const half8_t*)(q + ((long)b * 7 + qi) * 128 + h * 16);
    const half8_t c = *(const half8_t*)(q + ((long)b * 7 + qi) * 128 + h * 16 + 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qh[qi][e] = half2_t{a[2 * e], a[2 * e + 1]};
      qh[qi][4 + e] = half2_t{c[2 * e], c[2 * e + 1]};
    }
  }
  float m[7], l[7], acc[7][16];
#pragma unroll
  for (int qi = 0; qi < 7; ++qi) {
    m[qi] = -INFINITY;
    l[qi] = 0.f;
#pragma unroll
    for (int d = 0; d < 16; ++d) acc[qi][d] = 0.f;
  }
  const int per = T / nsplit;
  const half_t* Kb = K + (long)b * kv_bstride + h * 16;
  const half_t* Vb = V + (long)b * kv_bstride + h * 16;
  const float sc = 0.25f * LOG2E;  // 1/sqrt(16), base-2 softmax
  for (int key = split * per + slot; key < (split + 1) * per; key += 64) {
    // two keys per iteration: key and key+32 (per is a multiple of 64)
    half8_t k0a = *(const half8_t*)(Kb + (long)key * ldkv), k0b = *(const half8_t*)(Kb + (long)key * ldkv + 8);
    half8_t k1a = *(const half8_t*)(Kb + (long)(key + 32) * ldkv), k1b = *(const half8_t*)(Kb + (long)(key + 32) * ldkv + 8);
    half8_t v0a = *(const half8_t*)(Vb + (long)key * ldkv), v0b = *(const half8_t*)(Vb + (long)key * ldkv + 8);
    half8_t v1a = *(const half8_t*)(Vb + (long)(key + 32) * ldkv), v1b = *(const half8_t*)(Vb + (long)(key + 32) * ldkv + 8);
    float s0[7], s1[7];
#pragma unroll
    for (int qi = 0; qi < 7; ++qi) {
      float a0 = 0.f, a1 = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a0 = __builtin_amdgcn_fdot2(qh[qi][e], half2_t{k0a[2 * e], k0a[2 * e + 1]}, a0, false);
        a0 = __builtin_amdgcn_fdot2(qh[qi][4 + e], half2_t{k0b[2 * e], k0b[2 * e + 1]}, a0, false);
        a1 = __builtin_amdgcn_fdot2(qh[qi][e], half2_t{k1a[2 * e], k1a[2 * e + 1]}, a1, false);
        a1 = __builtin_amdgcn_fdot2(qh[qi][4 + e], half2_t{k1b[2 * e], k1b[2 * e + 1]}, a1, false);
      }
      s0[qi] = a0 * sc;
      s1[qi] = a1 * sc;
    }
    float vf0[16], vf1[16];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      vf0[d] = (float)v0a[d]; vf0[8 + d] = (float)v0b[d];
      vf1[d] = (float)v1a[d]; vf1[8 + d] = (float)v1b[d];
    }
#pragma unroll
    for (int qi = 0; qi < 7; ++qi) {
      const float mn = fmaxf(m[qi], fmaxf(s0[qi], s1[qi]));
      const float alpha = csam_exp2(m[qi] - mn);
      const float p0 = csam_exp2(s0[qi] - mn), p1 = csam_exp2(s1[qi] - mn);
      m[qi] = mn;
      l[qi] = l[qi] * alpha + p0 + p1;
#pragma unroll
      for (int d = 0; d < 16; ++d) acc[qi][d] = acc[qi][d] * alpha + p0 * vf0[d] + p1 * vf1[d];
    }
  }
  // ---- merge the 8 key-slot lanes of each head inside the wave (lane bits 3..5)
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
#pragma unroll
    for (int qi = 0; qi < 7; ++qi) {
      const float mo = __shfl_xor(m[qi], off, 64), lo = __shfl_xor(l[qi], off, 64);
      const float mn = fmaxf(m[qi], mo);
      const float a = csam_exp2(m[qi] - mn), bb = csam_exp2(mo - mn);
      l[qi] = l[qi] * a + lo * bb;
#pragma unroll
      for (int d = 0; d < 16; ++d) {
        const float ao = __shfl_xor(acc[qi][d], off, 64);
        acc[qi][d] = acc[qi][d] * a + ao * bb;
      }
      m[qi] = mn;
    }
  }
  if ((tid & 63) < 8) {
    float* dst = part + ((((long)b * nsplit + split) * 4 + wave) * 8 + h) * 7 * T2I_REC;
#pragma unroll
    for (int qi = 0; qi < 7; ++qi) {
      dst[qi * T2I_REC] = m[qi];
      dst[qi * T2I_REC + 1] = l[qi];
#pragma unroll
      for (int d = 0; d < 16; ++d) dst[qi * T2I_REC + 2 + d] = acc[qi][d];
    }
  }
}

// merge nparts partial records per (prompt, head, query) -> out f16 [B,7,128]
__global__ __launch_bounds__(64) void t2i_merge_kernel(const float* __restrict__ part, half_t* __restrict__ out,
                                                       int nparts) {
  const int b = blockIdx.x, t = threadIdx.x;
  if (t >= 56) return;
  const int h = t / 7, qi = t % 7;
  float m = -INFINITY, l = 0.f, acc[16];
#pragma unroll
  for (int d = 0; d < 16; ++d) acc[d] = 0.f;
  for (int p = 0; p < nparts; ++p) {
    const float* src = part + ((((long)b * nparts + p) * 8 + h) * 7 + qi) * T2I_REC;
    const float mo = src[0], lo = src[1];
    const float mn = fmaxf(m, mo);
    const float a = csam_exp2(m - mn), bb = csam_exp2(mo - mn);
    l = l * a + lo * bb;
#pragma unroll
    for (int d = 0; d < 16; ++d) acc[d] = acc[d] * a + src[2 + d] * bb;
    m = mn;
  }
  const float inv = 1.f / l;
  half_t* o = out + ((long)b * 7 + qi) * 128 + h * 16;
#pragma unroll
  for (int d = 0; d < 16; ++d) o[d] = (half_t)(acc[d] * inv);
}

// ---------------------------------------------------------------------------------------------
// image -> token attention (transformer.py:186-190): T image queries x 7 token keys, 8 x 16.
// Qi f16 rows of 128 (row stride ldq, per-prompt stride q_bstride; 0 for the shared layer-0
// projection); k, v f16 [B,7,128] -> out f16 [B*T, 128].  Thread = (head, token slot); the prompt's
// 7 keys/values of that head live in registers.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_i2t_kernel(const half_t* __restrict__ Qi, long ldq, long q_bstride,
                                                       const half_t* __restrict__ k,
                                                       const half_t* __restrict__ v,
                                                       half_t* __restrict__ out, int T, int nsplit) {
  const int b = blockIdx.x, split = blockIdx.y;
  const int tid = threadIdx.x, h = tid & 7, slot = tid >> 3;
  half2_t kh[7][8];
  float vf[7][16];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const half8_t a = *(const half8_t*)(k + ((long)b * 7 + j) * 128 + h * 16);
    const half8_t c = *(const half8_t*)(k + ((long)b * 7 + j) * 128 + h * 16 + 8);
    const half8_t va = *(const half8_t*)(v + ((long)b * 7 + j) * 128 + h * 16);
    const half8_t vc = *(const half8_t*)(v + ((long)b * 7 + j) * 128 + h * 16 + 8);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      kh[j][e] = half2_t{a[2 * e], a[2 * e + 1]};
      kh[j][4 + e] = half2_t{c[2 * e], c[2 * e + 1]};
    }
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      vf[j][d] = (float)va[d];
      vf[j][8 + d] = (float)vc[d];
    }
  }
  const int per = T / nsplit;
  const half_t* Qb = Qi + (long)b * q_bstride + h * 16;
  half_t* ob = out + (long)b * T * 128 + h * 16;
  const float sc = 0.25f * LOG2E;
  for (int t = split * per + slot; t < (split + 1) * per; t += 32) {
    const half8_t qa = *(const half8_t*)(Qb + (long)t * ldq), qb = *(const half8_t*)(Qb + (long)t * ldq + 8);
    half2_t qh[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      qh[e] = half2_t{qa[2 * e], qa[2 * e + 1]};
      qh[4 + e] = half2_t{qb[2 * e], qb[2 * e + 1]};
    }
    float s[7], mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      float a = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) a = __builtin_amdgcn_fdot2(qh[e], kh[j][e], a, false);
      s[j] = a * sc;
      mx = fmaxf(mx, s[j]);
    }
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      s[j] = csam_exp2(s[j] - mx);
      sum += s[j];
    }
    const float inv = 1.f / sum;
    float o[16];
#pragma unroll
    for (int d = 0; d < 16; ++d) o[d] = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const float p = s[j] * inv;
#pragma unroll
      for (int d = 0; d < 16; ++d) o[d] += p * vf[j][d];
    }
    half8_t oa, oc;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
      oa[d] = (half_t)o[d];
      oc[d] = (half_t)o[8 + d];
    }
    *(half8_t*)(ob + (long)t * 128) = oa;
    *(half8_t*)(ob + (long)t * 128 + 8) = oc;
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm2d(64) + GELU after the first ConvTranspose (mask_decoder.py:58-59), in place on the
// GEMM output viewed as rows of 64 channels (row = (prompt, pixel, sub-position)).  8 lanes/row.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ln64_gelu_kernel(half_t* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ bta, long rows, float eps) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long row = idx >> 3;
  const int part = (int)(idx & 7);
  if (row >= rows) return;
  half8_t v = *(half8_t*)(x + row * 64 + part * 8);
  float f[8], s = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    f[e] = (float)v[e];
    s += f[e];
  }
  s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
  const float mean = s * (1.f / 64.f);
  float qv = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float d = f[e] - mean;
    qv += d * d;
  }
  qv += __shfl_xor(qv, 1, 64); qv += __shfl_xor(qv, 2, 64); qv += __shfl_xor(qv, 4, 64);
  const float rstd = 1.0f / sqrtf(qv * (1.f / 64.f) + eps);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float y = (f[e] - mean) * rstd * g[part * 8 + e] + bta[part * 8 + e];
    v[e] = (half_t)csam_gelu_erf(y);
  }
  *(half8_t*)(x + row * 64 + part * 8) = v;
}

// ---------------------------------------------------------------------------------------------
// masks[b,l,Y,X] = sum_c hyper[b,l,c] * up[b,Y,X,c]  (mask_decoder.py:181) on the second
// ConvTranspose's GEMM output up2 f16 [B*4096*4, 128]: row = ((b*4096 + i*64+j)*4 + di*2+dj),
// col = (di2*2+dj2)*32 + c, output pixel Y = 4i+2di+di2, X = 4j+2dj+dj2.  One thread per pixel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void hyper_masks_kernel(const half_t* __restrict__ up2,
                                                          const float* __restrict__ hyper,
                                                          float* __restrict__ masks) {
  __shared__ float hs[128];
  const int b = blockIdx.y;
  if (threadIdx.x < 128) hs[threadIdx.x] = hyper[(long)b * 128 + threadIdx.x];
  __syncthreads();
  const int pix = blockIdx.x * 256 + threadIdx.x;  // Y*256 + X
  const int Y = pix >> 8, X = pix & 255;
  const int i = Y >> 2, di = (Y >> 1) & 1, di2 = Y & 1;
  const int j = X >> 2, dj = (X >> 1) & 1, dj2 = X & 1;
  const half_t* src = up2 + (((long)b * 4096 + i * 64 + j) * 4 + di * 2 + dj) * 128 + (di2 * 2 + dj2) * 32;
  float u[32];
#pragma unroll
  for (int c8 = 0; c8 < 4; ++c8) {
    const half8_t v = *(const half8_t*)(src + c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) u[c8 * 8 + e] = (float)v[e];
  }
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    float a = 0.f;
#pragma unroll
    for (int c = 0; c < 32; ++c) a += hs[l * 32 + c] * u[c];
    masks[(((long)b * 4 + l) << 16) + pix] = a;
  }
}

// ---------------------------------------------------------------------------------------------
// PWD-Net mask-weighted pooling (mask_decoder.py:186-190), re-associated:
//   pooled[b,l,:] = sum_hw softmax(masks[b,l])[hw] * bilinear_up(G)[:,hw]
//                 = ( sum_t (U^T e)[t] * G[t,:] ) / sum(e),   e = exp(masks - max)
// where U is the 73x73 -> 256x256 bilinear (align_corners=False) operator, so the 65536-long
// contraction collapses to 5329 after applying U^T (a gather with <= 8 taps per axis) to e.
// softmax_stats: per (b,l) max and sum(e).  pool_adjoint: w' = U^T e as f16 [B*4, ldw].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_stats_kernel(const float* __restrict__ masks,
                                                            float* __restrict__ stats) {
  __shared__ float red[4];
  const long r = blockIdx.x;
  const floatx4* src = (const floatx4*)(masks + (r << 16));
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < 16384; i += 256) {
    const floatx4 v = src[i];
    mx = fmaxf(mx, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
  }
  mx = csam_wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < 16384; i += 256) {
    const floatx4 v = src[i];
    s += __expf(v[0] - mx) + __expf(v[1] - mx) + __expf(v[2] - mx) + __expf(v[3] - mx);
  }
  s = csam_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    stats[r * 2] = mx;
    stats[r * 2 + 1] = red[0] + red[1] + red[2] + red[3];
  }
}

// taps: for each of the 73 coarse indices t, the fine indices X in [x0[t], x0[t]+n[t]) that touch
// it and their weights wt[t][k] (host-built from the align_corners=False formula).
struct AdjTaps {
  int x0[73];
  int n[73];
  float w[73][8];
};

__global__ __launch_bounds__(256) void pool_adjoint_kernel(const float* __restrict__ masks,
                                                           const float* __restrict__ stats,
                                                           const AdjTaps* __restrict__ taps,
                                                           half_t* __restrict__ wout, long ldw) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* tmp = (float*)smem;  // [256][73]  (Y, tj)
  __shared__ AdjTaps tp;
  const long r = blockIdx.x;
  for (int i = threadIdx.x; i < (int)(sizeof(AdjTaps) / 4); i += 256) ((int*)&tp)[i] = ((const int*)taps)[i];
  __syncthreads();
  const float mx = stats[r * 2];
  const float* src = masks + (r << 16);
  for (int it = threadIdx.x; it < 256 * 73; it += 256) {
    const int Y = it / 73, tj = it % 73;
    const int x0 = tp.x0[tj], n = tp.n[tj];
    float a = 0.f;
    for (int k = 0; k < n; ++k) a += tp.w[tj][k] * __expf(src[Y * 256 + x0 + k] - mx);
    tmp[it] = a;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < 73 * 73; it += 256) {
    const int ti = it / 73, tj = it % 73;
    const int y0 = tp.x0[ti], n = tp.n[ti];
    float a = 0.f;
    for (int k = 0; k < n; ++k) a += tp.w[ti][k] * tmp[(y0 + k) * 73 + tj];
    wout[r * ldw + it] = (half_t)a;
  }
}

// MFMA version of the same re-association: out = U^T E U with E = exp(x - max) in fp16 and U the dense form of
// the (banded) 256 -> 73 adjoint taps.  ONE WAVE PER PLANE, no LDS traffic for the data and no barriers:
//   T[Y,tj]   = sum_X E[Y,X] U[X,tj]    A = E rows straight from global (exp on the fly), B = packed U blocks
//   out[ti,tj] = sum_Y U[Y,ti] T[Y,tj]   B = the fp16 T accumulators of two Y tiles (register chaining, the
//                                        k-permutation is baked into the packed A blocks on the host)
// Only the (tile, k-step) blocks the band touches are stored and multiplied (13 + 13 of 40 + 40).  The kernel
// is then bound by the one fp32 read of the logits (1 MB / prompt) instead of LDS taps and 128 barriers.
// band of the 256 -> 73 geometry: tile j of 16 coarse indices touches the 32-wide fine k-steps ADJ_LO[j]..ADJ_HI[j];
// blocks are stored in (tile, k-step) order, so block id = ADJ_BASE[j] + kx - ADJ_LO[j]
__device__ constexpr int ADJ_LO[5] = {0, 1, 3, 5, 6}, ADJ_HI[5] = {1, 3, 5, 7, 7}, ADJ_BASE[5] = {0, 2, 5, 8, 11};

struct AdjMfma {
  int idx1[5][8];          // block of (tj tile, X k-step) in blk1, or -1
  int idx2[5][8];          // block of (ti tile, Y k-step) in blk2, or -1
  int pad[48];
  half8_t blk1[16][64];    // B fragments: lane (tj = 16j + l&15, g): U[32kx + 8g + e][tj]
  half8_t blk2[16][64];    // A fragments: lane (ti = 16i + l&15, g): U[(2s + (e>=4))*16 + 4g + (e&3)][ti]
};

#ifndef POOL_RING
#define POOL_RING 1
#endif
__global__ __launch_bounds__(256, 1) void pool_adjoint_mfma_kernel(const float* __restrict__ masks,
                                                                   float* __restrict__ stats,
                                                                   const AdjMfma* __restrict__ tab,
                                                                   half_t* __restrict__ wout, long ldw, int rows) {
  __shared__ AdjMfma T;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fg = lane >> 4;
  for (int i = tid; i < (int)(sizeof(AdjMfma) / 16); i += 256) ((floatx4*)&T)[i] = ((const floatx4*)tab)[i];
  __syncthreads();
  const long r = (long)blockIdx.x * 4 + wave;
  if (r >= rows) return;
  const float L2E = 1.4426950408889634f;
  const float nmx = -stats[r * 2] * L2E;
  const float* src = masks + (r << 16) + fr * 256 + fg * 8;
  floatx4 acc[5][5];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  float esum = 0.f;
  // POOL_RING (round 6, from the ISA): the kernel runs ONE wave per SIMD (376 registers) and the compiler kept two or three pairs
  // of 16-byte loads in flight per lane -- 16-24 KB per CU against the ~60 KB that 8 TB/s x ~2 us of loaded latency ask for,
  // which is the 0.58 of the HBM roof it measured at.  The logits now come through a ring of POOL_RING (a, b) pairs: pair q + 8
  // (the same k-step of the NEXT 16-row half block) is requested when pair q has been consumed, a scheduling barrier per
  // k-step keeps the requests where they are written: 16 loads = 64 KB per CU in flight.  Same loads, same arithmetic order.
  floatx4 ra[8], rb[8];
  auto ld_pair = [&](int q, floatx4& a, floatx4& b) {      // q = (2 s + h) * 8 + kx, clamped: the last half block re-requests itself
    const int qq = q < 128 ? q : q - 8;
    const float* rp = src + (qq >> 3) * 16 * 256 + (qq & 7) * 32;
    a = *(const floatx4*)rp;
    b = *(const floatx4*)(rp + 4);
  };
  if (POOL_RING) {
#pragma unroll
    for (int i = 0; i < 8; ++i) ld_pair(i, ra[i], rb[i]);
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll 1
  for (int s = 0; s < 8; ++s) {
    floatx4 t[2][5];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float* rowp = src + (2 * s + h) * 16 * 256;
#pragma unroll
      for (int j = 0; j < 5; ++j) t[h][j] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kx = 0; kx < 8; ++kx) {              // k-step outer: one E fragment live at a time
        floatx4 a, b;
        if (POOL_RING) {
          a = ra[kx];
          b = rb[kx];
          ld_pair((2 * s + h) * 8 + kx + 8, ra[kx], rb[kx]);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          a = *(const floatx4*)(rowp + kx * 32);
          b = *(const floatx4*)(rowp + kx * 32 + 4);
        }
        half8_t ef;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ea = csam_exp2(fmaf(a[e], L2E, nmx)), eb = csam_exp2(fmaf(b[e], L2E, nmx));
          esum += ea + eb;
          ef[e] = (half_t)ea;
          ef[4 + e] = (half_t)eb;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j)      // static band (checked against the host tables at launch): straight-line MFMAs
          if (kx >= ADJ_LO[j] && kx <= ADJ_HI[j])
            t[h][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ef, T.blk1[ADJ_BASE[j] + kx - ADJ_LO[j]][lane], t[h][j], 0, 0, 0);
      }
    }
    half8_t tb[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        tb[j][e] = (half_t)t[0][j][e];
        tb[j][4 + e] = (half_t)t[1][j][e];
      }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      if (s >= ADJ_LO[i] && s <= ADJ_HI[i]) {       // wave-uniform
        const half8_t af = T.blk2[ADJ_BASE[i] + s - ADJ_LO[i]][lane];
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, tb[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  esum = csam_wave_sum(esum);
  if (lane == 0) stats[r * 2 + 1] = esum;
  half_t* o = wout + r * ldw;
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const int tj = j * 16 + fr;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int ti = i * 16 + fg * 4 + e;
        if (ti < 73 && tj < 73) o[ti * 73 + tj] = (half_t)acc[i][j][e];
      }
    }
}

// pooled[r,c] = P[r,c] / sum[r] + bias[c]
__global__ __launch_bounds__(256) void rowscale_bias_kernel(const float* __restrict__ P,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int N) {
  const int r = blockIdx.x;
  const float inv = 1.f / stats[r * 2 + 1];
  for (int c = threadIdx.x; c < N; c += 256) out[(long)r * N + c] = P[(long)r * N + c] * inv + bias[c];
}

// split-K epilogue: out[r,c] = (sum_s P[s][r][c]) * (stats ? 1 / stats[r][1] : 1) + bias[c] + residual[r,c], the partial
// products summed in the order s = 0 .. S-1 (bit-repeatable).  4 columns per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ P, int S, long slab,
                                                            const float* __restrict__ stats,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ residual, long ldr,
                                                            float* __restrict__ out, long ldo, int rows, int N) {
  const int n4 = N >> 2;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)rows * n4) return;
  const int r = (int)(i / n4), c = (int)(i % n4) * 4;
  floatx4 acc = *(const floatx4*)(P + (long)r * N + c);
  for (int s = 1; s < S; ++s) acc += *(const floatx4*)(P + s * slab + (long)r * N + c);
  if (stats) {
    const float inv = 1.f / stats[r * 2 + 1];
    acc *= inv;
  }
  if (bias) acc += *(const floatx4*)(bias + c);
  if (residual) acc += *(const floatx4*)(residual + (long)r * ldr + c);
  *(floatx4*)(out + (long)r * ldo + c) = acc;
}

}  // namespace

extern "C" int csam_point_tokens(void* stream, const float* coords, const float* gauss, const float* out_tokens5,
                                 const float* point_embed1, const float* not_a_point, float* tokens, int B) {
  CSAM_REQUIRE(coords && gauss && out_tokens5 && point_embed1 && not_a_point && tokens && B > 0,
               "csam_point_tokens: bad args");
  hipLaunchKernelGGL(point_tokens_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, coords, gauss, out_tokens5,
                     point_embed1, not_a_point, tokens);
  CSAM_LAUNCH_CHECK("csam_point_tokens");
  return CSAM_OK;
}

extern "C" int csam_point_tokens_labeled(void* stream, const float* coords, const int* labels, const float* gauss,
                                         const float* out_tokens5, const float* point_embed0, const float* point_embed1,
                                         const float* not_a_point, float* tokens, int B) {
  CSAM_REQUIRE(coords && labels && gauss && out_tokens5 && point_embed0 && point_embed1 && not_a_point && tokens && B > 0,
               "csam_point_tokens_labeled: bad args");
  hipLaunchKernelGGL(point_tokens_labeled_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, coords, labels, gauss,
                     out_tokens5, point_embed0, point_embed1, not_a_point, tokens);
  CSAM_LAUNCH_CHECK("csam_point_tokens_labeled");
  return CSAM_OK;
}

extern "C" int csam_box_tokens(void* stream, const float* boxes_xyxy, const float* gauss, const float* out_tokens5,
                               const float* point_embed2, const float* point_embed3, float* tokens, int B) {
  CSAM_REQUIRE(boxes_xyxy && gauss && out_tokens5 && point_embed2 && point_embed3 && tokens && B > 0,
               "csam_box_tokens: bad args");
  hipLaunchKernelGGL(box_tokens_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, boxes_xyxy, gauss, out_tokens5,
                     point_embed2, point_embed3, tokens);
  CSAM_LAUNCH_CHECK("csam_box_tokens");
  return CSAM_OK;
}

extern "C" int csam_pe_points(void* stream, const float* coords, const float* gauss, float* out, int P) {
  CSAM_REQUIRE(coords && gauss && out && P > 0, "csam_pe_points: bad args");
  hipLaunchKernelGGL(pe_points_kernel, dim3(P), dim3(256), 0, (hipStream_t)stream, coords, gauss, out);
  CSAM_LAUNCH_CHECK("csam_pe_points");
  return CSAM_OK;
}

extern "C" int csam_token_self_attn(void* stream, const void* qk_f16, const void* v_f16, void* out_f16, int B) {
  CSAM_REQUIRE(qk_f16 && v_f16 && out_f16 && B > 0, "csam_token_self_attn: bad args");
  hipLaunchKernelGGL(token_self_attn_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const half_t*)qk_f16,
                     (const half_t*)v_f16, (half_t*)out_f16);
  CSAM_LAUNCH_CHECK("csam_token_self_attn");
  return CSAM_OK;
}

extern "C" long csam_attn_t2i_workspace_bytes(int B, int nsplit) {
  return (long)B * nsplit * 4 * 8 * 7 * T2I_REC * sizeof(float);
}

extern "C" int csam_attn_t2i(void* stream, const void* q_f16, const void* K_f16, const void* V_f16, long ldkv,
                             long kv_prompt_stride, void* out_f16, int B, int T, int nsplit, void* workspace,
                             long workspace_bytes) {
  CSAM_REQUIRE(q_f16 && K_f16 && V_f16 && out_f16 && workspace && B > 0, "csam_attn_t2i: bad args");
  CSAM_REQUIRE(nsplit >= 1 && T % (64 * nsplit) == 0, "csam_attn_t2i: T=%d must be a multiple of 64*nsplit", T);
  CSAM_REQUIRE(ldkv % 8 == 0 && kv_prompt_stride % 8 == 0, "csam_attn_t2i: alignment");
  if (workspace_bytes < csam_attn_t2i_workspace_bytes(B, nsplit)) {
    csam_set_error("csam_attn_t2i: workspace too small");
    return CSAM_ERR_WORKSPACE;
  }
  hipLaunchKernelGGL(attn_t2i_kernel, dim3(B, nsplit), dim3(256), 0, (hipStream_t)stream, (const half_t*)q_f16,
                     (const half_t*)K_f16, (const half_t*)V_f16, ldkv, kv_prompt_stride, (float*)workspace, T,
                     nsplit);
  hipLaunchKernelGGL(t2i_merge_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, (const float*)workspace,
                     (half_t*)out_f16, nsplit * 4);
  CSAM_LAUNCH_CHECK("csam_attn_t2i");
  return CSAM_OK;
}

// merge launcher shared with the fused kernel (decoder_fused.hip)
extern "C" int csam_t2i_merge_launch(void* stream, const float* part, void* out_f16, int B, int nparts) {
  hipLaunchKernelGGL(t2i_merge_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, part, (half_t*)out_f16, nparts);
  CSAM_LAUNCH_CHECK("csam_t2i_merge");
  return CSAM_OK;
}

extern "C" int csam_attn_i2t(void* stream, const void* Qi_f16, long ldq, long q_prompt_stride, const void* k_f16,
                             const void* v_f16, void* out_f16, int B, int T, int nsplit) {
  CSAM_REQUIRE(Qi_f16 && k_f16 && v_f16 && out_f16 && B > 0, "csam_attn_i2t: bad args");
  CSAM_REQUIRE(nsplit >= 1 && T % (32 * nsplit) == 0, "csam_attn_i2t: T must be a multiple of 32*nsplit");
  CSAM_REQUIRE(ldq % 8 == 0 && q_prompt_stride % 8 == 0, "csam_attn_i2t: alignment");
  hipLaunchKernelGGL(attn_i2t_kernel, dim3(B, nsplit), dim3(256), 0, (hipStream_t)stream, (const half_t*)Qi_f16,
                     ldq, q_prompt_stride, (const half_t*)k_f16, (const half_t*)v_f16, (half_t*)out_f16, T, nsplit);
  CSAM_LAUNCH_CHECK("csam_attn_i2t");
  return CSAM_OK;
}

extern "C" int csam_ln64_gelu(void* stream, void* x_f16, const float* gamma, const float* beta, long rows,
                              float eps) {
  CSAM_REQUIRE(x_f16 && gamma && beta && rows > 0, "csam_ln64_gelu: bad args");
  hipLaunchKernelGGL(ln64_gelu_kernel, dim3(csam_cdiv(rows * 8, 256)), dim3(256), 0, (hipStream_t)stream,
                     (half_t*)x_f16, gamma, beta, rows, eps);
  CSAM_LAUNCH_CHECK("csam_ln64_gelu");
  return CSAM_OK;
}

extern "C" int csam_hyper_masks(void* stream, const void* up2_f16, const float* hyper, float* masks, int B) {
  CSAM_REQUIRE(up2_f16 && hyper && masks && B > 0, "csam_hyper_masks: bad args");
  hipLaunchKernelGGL(hyper_masks_kernel, dim3(256, B), dim3(256), 0, (hipStream_t)stream, (const half_t*)up2_f16,
                     hyper, masks);
  CSAM_LAUNCH_CHECK("csam_hyper_masks");
  return CSAM_OK;
}

extern "C" int csam_softmax_stats(void* stream, const float* masks, float* stats, int rows) {
  CSAM_REQUIRE(masks && stats && rows > 0, "csam_softmax_stats: bad args");
  hipLaunchKernelGGL(softmax_stats_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, masks, stats);
  CSAM_LAUNCH_CHECK("csam_softmax_stats");
  return CSAM_OK;
}

extern "C" int csam_adj_taps_bytes(void) { return (int)sizeof(AdjTaps); }

extern "C" int csam_pool_adjoint(void* stream, const float* masks, const float* stats, const void* taps_dev,
                                 void* w_f16, long ldw, int rows) {
  CSAM_REQUIRE(masks && stats && taps_dev && w_f16 && rows > 0 && ldw >= 5329, "csam_pool_adjoint: bad args");
  const int smem = 256 * 73 * 4;
  static csam_once_t attr_set;
  if (csam_first_call(attr_set))
    hipFuncSetAttribute((const void*)pool_adjoint_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
  hipLaunchKernelGGL(pool_adjoint_kernel, dim3(rows), dim3(256), smem, (hipStream_t)stream, masks, stats,
                     (const AdjTaps*)taps_dev, (half_t*)w_f16, ldw);
  CSAM_LAUNCH_CHECK("csam_pool_adjoint");
  return CSAM_OK;
}

extern "C" int csam_adj_mfma_bytes(void) { return (int)sizeof(AdjMfma); }

// MFMA variant of csam_pool_adjoint_v2 (same contract); tables_dev: an AdjMfma built by the host from the taps
extern "C" int csam_pool_adjoint_mfma(void* stream, const float* masks, float* stats, const void* tables_dev,
                                      void* w_f16, long ldw, int rows) {
  CSAM_REQUIRE(masks && stats && tables_dev && w_f16 && rows > 0 && ldw >= 5329, "csam_pool_adjoint_mfma: bad args");
  hipLaunchKernelGGL(pool_adjoint_mfma_kernel, dim3(csam_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream, masks, stats,
                     (const AdjMfma*)tables_dev, (half_t*)w_f16, ldw, rows);
  CSAM_LAUNCH_CHECK("csam_pool_adjoint_mfma");
  return CSAM_OK;
}

extern "C" int csam_splitk_reduce(void* stream, const float* partials, int splits, long slab_stride,
                                  const float* stats_or_null, const float* bias_or_null, const float* residual_or_null,
                                  long ldr, float* out, long ldo, int rows, int N) {
  CSAM_REQUIRE(partials && out && splits > 0 && rows > 0 && N > 0 && N % 4 == 0 && ldo % 4 == 0 && slab_stride % 4 == 0 &&
                   (!residual_or_null || ldr % 4 == 0),
               "csam_splitk_reduce: bad args");
  const long n = (long)rows * (N / 4);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)csam_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, partials,
                     splits, slab_stride, stats_or_null, bias_or_null, residual_or_null, ldr, out, ldo, rows, N);
  CSAM_LAUNCH_CHECK("csam_splitk_reduce");
  return CSAM_OK;
}

extern "C" int csam_rowscale_bias(void* stream, const float* P, const float* stats, const float* bias, float* out,
                                  int rows, int N) {
  CSAM_REQUIRE(P && stats && bias && out && rows > 0 && N > 0, "csam_rowscale_bias: bad args");
  hipLaunchKernelGGL(rowscale_bias_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, P, stats, bias, out, N);
  CSAM_LAUNCH_CHECK("csam_rowscale_bias");
  return CSAM_OK;
}
