// csam_flash_attn: global multi-head attention, flash style (no [T,T] score tensor), head_dim 64.
//
// Serves (a) SAM's 4 global ViTDet blocks (image_encoder.py:224-240 with window_size == 0) including
// the decomposed relative-position bias (:325-361), and (b) every DINOv2 ViT-L/14 block (external
// dependency; plain softmax(q k^T / sqrt(d)) v over 1 + 73*73 tokens).
//
// Workgroup = 4 waves x 32 queries = 128 queries of one head; key tiles of 64 are staged into
// double-buffered LDS (K row-major, V transposed) through registers (issue-early / write-late),
// one barrier per tile.  Swapped MFMA orientation: keys on the accumulator rows, so each lane owns
// ONE query column -> online-softmax max/sum are in-lane + two shuffles, and the fp16 P registers
// feed P.V directly as the B operand.
//
// Rel-pos bias, SAM global blocks (64x64 grid, key tile t == key row kh = t, kw = key & 63):
//     S[q, (kh,kw)] = scale * (q.k + Th[q,kh]/scale + Tw[q,kw]/scale)
// The MFMA accumulator is INITIALISED with Tw[q, kw]/scale (16 registers per query tile, constant
// over all key tiles) + Th[q, t]/scale (one scalar per lane per tile): the bias costs one v_add per
// accumulator register and zero extra MFMA/LDS work.  Tables come from csam_relpos_tables.
#include "csam_common.h"

namespace {

constexpr int KT = 64;                 // keys per tile
constexpr int K_LD = 72, V_LD = 72;    // padded LDS row lengths (halfs)
constexpr int TILE_HALFS = KT * K_LD;  // 4608 halfs = 9216 B
constexpr int QPW = 32;                // queries per wave
constexpr int QPB = 128;               // queries per block

template <bool BIAS>
__global__ __launch_bounds__(256) void flash_attn_kernel(const half_t* __restrict__ qkv, long ld,
                                                         int q_off, int k_off, int v_off,
                                                         const float* __restrict__ th,
                                                         const float* __restrict__ tw,
                                                         half_t* __restrict__ out, long ldo, int T,
                                                         float scale) {
  __shared__ __attribute__((aligned(16))) half_t Ks[2][TILE_HALFS];
  __shared__ __attribute__((aligned(16))) half_t Vs[2][TILE_HALFS];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fg = lane >> 4;
  const int head = blockIdx.y;
  const int q0 = blockIdx.x * QPB + wave * QPW;
  const half_t* qp = qkv + q_off + head * 64;
  const half_t* kp = qkv + k_off + head * 64;
  const half_t* vp = qkv + v_off + head * 64;

  // ---- Q fragments (B operand: column = query fr), rows clamped for the ragged tail
  half8_t qf[2][2];
  int qrow[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    qrow[rt] = q0 + rt * 16 + fr;
    const int qc = qrow[rt] < T ? qrow[rt] : T - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[rt][ks] = *(const half8_t*)(qp + (long)qc * ld + (ks * 4 + fg) * 8);
  }
  // ---- rel-pos: Tw registers (constant over key tiles)
  floatx4 twr[2][4];
  const float* thp[2] = {nullptr, nullptr};
  if constexpr (BIAS) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int qc = qrow[rt] < T ? qrow[rt] : T - 1;
      const float* twq = tw + ((long)head * T + qc) * 64;
      thp[rt] = th + ((long)head * T + qc) * 64;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) twr[rt][kt] = *(const floatx4*)(twq + kt * 16 + fg * 4);
    }
  }

  floatx4 o[2][4];
  float m[2], l[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    m[rt] = -INFINITY;
    l[rt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[rt][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
  }

  // ---- staging: 512 16-B chunks per operand per tile, 2 per thread
  const int nt = (T + KT - 1) / KT;
  half8_t kreg[2], vreg[2];
  auto gload = [&](int t) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256;
      int key = t * KT + (c >> 3);
      key = key < T ? key : T - 1;
      kreg[i] = *(const half8_t*)(kp + (long)key * ld + (c & 7) * 8);
      vreg[i] = *(const half8_t*)(vp + (long)key * ld + (c & 7) * 8);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256;
      const int key = c >> 3, ch = c & 7;
      *(half8_t*)(&Ks[buf][key * K_LD + ch * 8]) = kreg[i];
#pragma unroll
      for (int e = 0; e < 8; ++e) Vs[buf][(ch * 8 + e) * V_LD + key] = vreg[i][e];
    }
  };
  gload(0);
  lstore(0);
  __syncthreads();

  const float sl2 = scale * 1.4426950408889634f;
  const float inv_scale_bias = 1.0f;  // tables already hold bias/scale

  for (int t = 0; t < nt; ++t) {
    const int cur = t & 1;
    if (t + 1 < nt) gload(t + 1);
    const half_t* Kc = Ks[cur];
    const half_t* Vc = Vs[cur];

    // ---- S^T = K Q^T (+ bias)
    floatx4 s[2][4];
    float thv[2] = {0.f, 0.f};
    if constexpr (BIAS) {
      thv[0] = thp[0][t] * inv_scale_bias;
      thv[1] = thp[1][t] * inv_scale_bias;
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        if constexpr (BIAS) {
          s[rt][kt] = twr[rt][kt] + thv[rt];
        } else {
          s[rt][kt] = floatx4{0.f, 0.f, 0.f, 0.f};
        }
      }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const half8_t kf = *(const half8_t*)(Kc + (kt * 16 + fr) * K_LD + (ks * 4 + fg) * 8);
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          s[rt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[rt][ks], s[rt][kt], 0, 0, 0);
      }
    }
    // ---- online softmax (base 2); lane holds keys kt*16 + fg*4 + j of query fr
    const int kbase = t * KT + fg * 4;
    const bool tail = (t + 1) * KT > T;
    half8_t pf[2][2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float v = s[rt][kt][j] * sl2;
          if (tail && kbase + kt * 16 + j >= T) v = -INFINITY;
          s[rt][kt][j] = v;
          mx = fmaxf(mx, v);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(m[rt], mx);
      const float alpha = exp2f(m[rt] - mnew);
      m[rt] = mnew;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float e = exp2f(s[rt][kt][j] - mnew);
          s[rt][kt][j] = e;
          ps += e;
        }
      l[rt] = l[rt] * alpha + ps;   // lane-partial sum (reduced over fg at the end)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[rt][dt] *= alpha;
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pf[rt][st][e] = (half_t)s[rt][2 * st][e];
          pf[rt][st][4 + e] = (half_t)s[rt][2 * st + 1][e];
        }
    }
    // ---- O^T += V^T P^T
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const half_t* vr = Vc + (dt * 16 + fr) * V_LD + fg * 4;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const half4_t v0 = *(const half4_t*)(vr + 32 * st);
        const half4_t v1 = *(const half4_t*)(vr + 32 * st + 16);
        const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          o[rt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[rt][st], o[rt][dt], 0, 0, 0);
      }
    }
    if (t + 1 < nt) lstore(cur ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    float lt = l[rt];
    lt += __shfl_xor(lt, 16, 64);
    lt += __shfl_xor(lt, 32, 64);
    const float inv = 1.0f / lt;
    if (qrow[rt] < T) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        half4_t h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (half_t)(o[rt][dt][j] * inv);
        *(half4_t*)(out + (long)qrow[rt] * ldo + head * 64 + dt * 16 + fg * 4) = h;
      }
    }
  }
}

// Th[h][q][kh] = (q . rel_pos_h[qh - kh + G-1]) / scale ; Tw likewise with (qw, kw).   G == 64.
// image_encoder.py:292-322 (get_rel_pos, table length == 2G-1 so no interpolation), :349-350.
__global__ __launch_bounds__(256) void relpos_tables_kernel(const half_t* __restrict__ qkv, long ld,
                                                            const float* __restrict__ rel_h,
                                                            const float* __restrict__ rel_w,
                                                            float* __restrict__ th, float* __restrict__ tw,
                                                            int nH, float inv_scale) {
  // block = (query q, head h) pair x 128 outputs (64 kh + 64 kw); 2 pairs per 256-thread block
  const int pair = blockIdx.x * 2 + (threadIdx.x >> 7);
  const int j = threadIdx.x & 127;
  const int q = pair / nH, h = pair % nH;
  if (q >= 4096) return;
  const int qh = q >> 6, qw = q & 63;
  const half_t* qv = qkv + (long)q * ld + h * 64;
  const float* R = (j < 64) ? rel_h + (qh - j + 63) * 64 : rel_w + (qw - (j - 64) + 63) * 64;
  float acc = 0.f;
#pragma unroll
  for (int c8 = 0; c8 < 8; ++c8) {
    const half8_t qq = *(const half8_t*)(qv + c8 * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc += (float)qq[e] * R[c8 * 8 + e];
  }
  acc *= inv_scale;
  float* dst = (j < 64) ? th + ((long)h * 4096 + q) * 64 + j : tw + ((long)h * 4096 + q) * 64 + (j - 64);
  *dst = acc;
}

}  // namespace

extern "C" int csam_relpos_tables(void* stream, const void* qkv_f16, long ld, const float* rel_pos_h,
                                  const float* rel_pos_w, float* th, float* tw, int nH, float scale) {
  CSAM_REQUIRE(qkv_f16 && rel_pos_h && rel_pos_w && th && tw && nH > 0, "csam_relpos_tables: bad args");
  hipLaunchKernelGGL(relpos_tables_kernel, dim3(4096 * nH / 2), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)qkv_f16, ld, rel_pos_h, rel_pos_w, th, tw, nH, 1.0f / scale);
  CSAM_LAUNCH_CHECK("csam_relpos_tables");
  return CSAM_OK;
}

extern "C" int csam_flash_attn(void* stream, const void* qkv_f16, long ld, int q_off, int k_off, int v_off,
                               const float* th, const float* tw, void* out_f16, long ldo, int T, int nH,
                               float scale) {
  CSAM_REQUIRE(qkv_f16 && out_f16 && T > 0 && nH > 0, "csam_flash_attn: bad args");
  CSAM_REQUIRE(ld % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && ldo % 4 == 0,
               "csam_flash_attn: alignment");
  CSAM_REQUIRE((th == nullptr) == (tw == nullptr), "csam_flash_attn: th/tw must come together");
  CSAM_REQUIRE(!th || T == 4096, "csam_flash_attn: rel-pos bias needs the 64x64 token grid");
  dim3 grid(csam_cdiv(T, QPB), nH), block(256);
  if (th)
    hipLaunchKernelGGL(flash_attn_kernel<true>, grid, block, 0, (hipStream_t)stream, (const half_t*)qkv_f16, ld,
                       q_off, k_off, v_off, th, tw, (half_t*)out_f16, ldo, T, scale);
  else
    hipLaunchKernelGGL(flash_attn_kernel<false>, grid, block, 0, (hipStream_t)stream, (const half_t*)qkv_f16, ld,
                       q_off, k_off, v_off, th, tw, (half_t*)out_f16, ldo, T, scale);
  CSAM_LAUNCH_CHECK("csam_flash_attn");
  return CSAM_OK;
}
