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

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

// V^T per head: vt[h][d][t] = qkv[t][v_off + h*64 + d], t < T (columns T..Tpad-1 stay zero).
// A 64x64 tile goes through LDS so that both the read (128 B per token) and the write (128 B per dim row)
// are coalesced.  ~20 MB of traffic per DINOv2 block: a few microseconds.
__global__ __launch_bounds__(256) void transpose_v_kernel(const half_t* __restrict__ qkv, long ld, int v_off,
                                                          half_t* __restrict__ vt, int T, int Tpad) {
  __shared__ half_t tile[64][66];
  const int t0 = blockIdx.x * 64, h = blockIdx.y;
  const int tid = threadIdx.x;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256;           // 512 16-B chunks: token c>>3, dims (c&7)*8..+7
    const int t = t0 + (c >> 3);
    half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t < T) v = *(const half8_t*)(qkv + (long)t * ld + v_off + h * 64 + (c & 7) * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[(c & 7) * 8 + e][c >> 3] = v[e];
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256;           // dim row c>>3, tokens (c&7)*8..+7
    const int d = c >> 3, tt = (c & 7) * 8;
    half8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[d][tt + e];
    *(half8_t*)(vt + ((long)h * 64 + d) * Tpad + t0 + tt) = v;
  }
}

// K-tile LDS swizzle: fragment reads touch key rows {8a + b (+4)}, a,b in 0..3 (see the key permutation
// below), so the XOR term must separate rows by bits 1 and 3..4 rather than by (row & 7).
__device__ __forceinline__ int kswz(int row) { return ((row >> 1) & 1) | (((row >> 3) & 3) << 1); }

#ifndef FLASH_NS
#define FLASH_NS 3
#endif
#ifndef FLASH_OCC
#define FLASH_OCC 1
#endif
template <bool BIAS>
__global__ __launch_bounds__(256, FLASH_OCC) void flash_attn_kernel(const half_t* __restrict__ qkv, long ld,
                                                         int q_off, int k_off,
                                                         const half_t* __restrict__ vt, int Tpad,
                                                         const float* __restrict__ traw,
                                                         half_t* __restrict__ out, long ldo, int T,
                                                         float scale) {
  // traw (BIAS): [nH][T][256] fp32 = q . [rel_pos_h (127 rows) | 0 | rel_pos_w (127 rows) | 0] from one
  // batched GEMM;  Th[q][kh] = traw[q][qh - kh + 63],  Tw[q][kw] = traw[q][128 + qw - kw + 63]
  // LDS: 3-deep ring x (K tile [64 keys][128 B] + V^T tile [64 dims][128 B]) = 48 KB, filled by global_load_lds;
  // tile t+2 is issued while tile t is consumed; every tile is retired (vmcnt(0)) one iteration before it is read
  __shared__ __attribute__((aligned(16))) char smem[FLASH_NS * 16384];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fg = lane >> 4;
  const int head = blockIdx.y;
  const int q0 = blockIdx.x * QPB + wave * QPW;
  const half_t* qp = qkv + q_off + head * 64;
  const half_t* kp = qkv + k_off + head * 64;
  const half_t* vtp = vt + (long)head * 64 * Tpad;

  half8_t qf[2][2];
  int qrow[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    qrow[rt] = q0 + rt * 16 + fr;
    const int qc = qrow[rt] < T ? qrow[rt] : T - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[rt][ks] = *(const half8_t*)(qp + (long)qc * ld + (ks * 4 + fg) * 8);
  }
  // Key permutation inside a 32-key step s: accumulator row (4g + r) of key tile kt holds key
  // 32s + 8g + r + 4*(kt&1), so that a lane's 8 P values of a step are 8 CONSECUTIVE keys and the V^T
  // fragment is one 16-B read.  Row this lane supplies as the A operand of S^T = K Q^T:
  const int krow_in_step = 8 * (fr >> 2) + (fr & 3);
  floatx4 twr[2][4];
  const float* thp[2] = {nullptr, nullptr};
  const float inv_scale = 1.0f / scale;
  if constexpr (BIAS) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int qc = qrow[rt] < T ? qrow[rt] : T - 1;
      const float* tq = traw + ((long)head * T + qc) * 256;
      thp[rt] = tq + (qc >> 6) + 63;                 // Th[q][t] = thp[-t]
      const float* twq = tq + 128 + (qc & 63) + 63;  // Tw[q][kw] = twq[-kw]
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int kw0 = (kt >> 1) * 32 + 8 * fg + 4 * (kt & 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) twr[rt][kt][j] = twq[-(kw0 + j)] * inv_scale;
      }
    }
  }
  floatx4 o[2][4];
  float m[2];
  floatx4 l[2];
  const half8_t ones8 = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    m[rt] = -INFINITY;
    l[rt] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[rt][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
  }

  const int nt = (T + KT - 1) / KT;
  auto stage = [&](int buf, int t) {
    // 512 16-B pieces per tile, 2 per thread: piece c -> row c>>3, LDS slot c&7
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256;
      const int row = c >> 3, sl = c & 7;
      int key = t * KT + row;
      key = key < T ? key : T - 1;
      glds16(kp + (long)key * ld + ((sl ^ kswz(row)) * 8), smem + buf * 16384 + (c & ~63) * 16);
      glds16(vtp + (long)row * Tpad + t * KT + ((sl ^ (row & 7)) * 8), smem + buf * 16384 + 8192 + (c & ~63) * 16);
    }
  };
  stage(0, 0);
  if (FLASH_NS == 3 && nt > 1) stage(1, 1);

  const float sl2 = scale * 1.4426950408889634f;
  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    // vmcnt(0): with the 3-slot ring this retires tile t+1 one iteration before it is read (a counted wait
    // followed by a same-phase read of other waves' LDS-DMA data is not safe, see decoder_fused.hip t2i)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // tile t landed for every wave; everybody is done reading tile t-1's buffer
    asm volatile("" ::: "memory");
    if (FLASH_NS == 3) {
      if (t + 2 < nt) stage(cur == 0 ? 2 : cur - 1, t + 2);
    } else {
      if (t + 1 < nt) stage(cur ^ 1, t + 1);
    }
    const char* Kc = smem + cur * 16384;
    const char* Vc = Kc + 8192;
    cur = cur + 1 == FLASH_NS ? 0 : cur + 1;

    floatx4 s[2][4];
    float thv[2] = {0.f, 0.f};
    if constexpr (BIAS) {
      thv[0] = thp[0][-t] * inv_scale;
      thv[1] = thp[1][-t] * inv_scale;
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
        const int krow = (kt >> 1) * 32 + krow_in_step + 4 * (kt & 1);
        const half8_t kf = *(const half8_t*)(Kc + krow * 128 + (((ks * 4 + fg) ^ kswz(krow)) << 4));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          s[rt][kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[rt][ks], s[rt][kt], 0, 0, 0);
      }
    }
    // ---- online softmax (base 2); lane holds keys t*64 + (kt>>1)*32 + 8 fg + 4 (kt&1) + j of query fr.
    // VALU-bound part of the kernel (PMC: VALU 72 % busy, MFMA 14 %), so it is kept to max3 / fma / exp / add per
    // score: the softmax scale lives in the exponent's fma (the running max is tracked on the raw scores, scale > 0),
    // out-of-range keys are masked on the last tile only, and O is rescaled only when some row's max moved.
    if ((t + 1) * KT > T) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (t * KT + (kt >> 1) * 32 + 8 * fg + 4 * (kt & 1) + j >= T) s[rt][kt][j] = -INFINITY;
    }
    half8_t pf[2][2];
    float alpha[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      float mx = fmaxf(fmaxf(s[rt][0][0], s[rt][0][1]), s[rt][0][2]);
      mx = fmaxf(fmaxf(mx, s[rt][0][3]), s[rt][1][0]);
      mx = fmaxf(fmaxf(mx, s[rt][1][1]), s[rt][1][2]);
      mx = fmaxf(fmaxf(mx, s[rt][1][3]), s[rt][2][0]);
      mx = fmaxf(fmaxf(mx, s[rt][2][1]), s[rt][2][2]);
      mx = fmaxf(fmaxf(mx, s[rt][2][3]), s[rt][3][0]);
      mx = fmaxf(fmaxf(mx, s[rt][3][1]), s[rt][3][2]);
      mx = fmaxf(mx, s[rt][3][3]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float mnew = fmaxf(m[rt], mx);          // raw-score units
      alpha[rt] = csam_exp2((m[rt] - mnew) * sl2);
      m[rt] = mnew;
      const float nm = -mnew * sl2;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int j = 0; j < 4; ++j) s[rt][kt][j] = csam_exp2(fmaf(s[rt][kt][j], sl2, nm));
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          pf[rt][st][e] = (half_t)s[rt][2 * st][e];
          pf[rt][st][4 + e] = (half_t)s[rt][2 * st + 1][e];
        }
    }
    if (__ballot(alpha[0] != 1.f || alpha[1] != 1.f) != 0ull) {     // exact: alpha == 1 when no max moved
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        l[rt] *= alpha[rt];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[rt][dt] *= alpha[rt];
      }
    }
    // row sums on the matrix pipe (the loop is VALU-bound): l^T += 1 . P^T, every lane of a query receives the sum of
    // the fp16 probabilities the PV product uses
#pragma unroll
    for (int st = 0; st < 2; ++st)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) l[rt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones8, pf[rt][st], l[rt], 0, 0, 0);
    // ---- O^T += V^T P^T: V^T fragment = dims row dt*16+fr, keys 32 st + 8 fg .. +7 (one 16-B read)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int vrow = dt * 16 + fr;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const half8_t vf = *(const half8_t*)(Vc + vrow * 128 + (((st * 4 + fg) ^ (vrow & 7)) << 4));
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
          o[rt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[rt][st], o[rt][dt], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const float inv = 1.0f / l[rt][0];
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

}  // namespace

extern "C" long csam_flash_attn_workspace_bytes(int T, int nH) {
  const long Tpad = (long)((T + 63) / 64) * 64;
  return (long)nH * 64 * Tpad * 2;
}

extern "C" int csam_flash_attn(void* stream, const void* qkv_f16, long ld, int q_off, int k_off, int v_off,
                               const float* relpos_raw, void* out_f16, long ldo, int T, int nH, float scale,
                               void* vt_workspace, long vt_workspace_bytes) {
  CSAM_REQUIRE(qkv_f16 && out_f16 && vt_workspace && T > 0 && nH > 0, "csam_flash_attn: bad args");
  if (vt_workspace_bytes < csam_flash_attn_workspace_bytes(T, nH)) {
    csam_set_error("csam_flash_attn: V^T workspace too small (must also be zero-initialised once)");
    return CSAM_ERR_WORKSPACE;
  }
  const int Tpad = ((T + 63) / 64) * 64;
  CSAM_REQUIRE(ld % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && ldo % 4 == 0,
               "csam_flash_attn: alignment");
  CSAM_REQUIRE(!relpos_raw || T == 4096, "csam_flash_attn: rel-pos bias needs the 64x64 token grid");
  hipLaunchKernelGGL(transpose_v_kernel, dim3(Tpad / 64, nH), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)qkv_f16, ld, v_off, (half_t*)vt_workspace, T, Tpad);
  dim3 grid(csam_cdiv(T, QPB), nH), block(256);
  if (relpos_raw)
    hipLaunchKernelGGL(flash_attn_kernel<true>, grid, block, 0, (hipStream_t)stream, (const half_t*)qkv_f16, ld,
                       q_off, k_off, (const half_t*)vt_workspace, Tpad, relpos_raw, (half_t*)out_f16, ldo, T, scale);
  else
    hipLaunchKernelGGL(flash_attn_kernel<false>, grid, block, 0, (hipStream_t)stream, (const half_t*)qkv_f16, ld,
                       q_off, k_off, (const half_t*)vt_workspace, Tpad, relpos_raw, (half_t*)out_f16, ldo, T, scale);
  CSAM_LAUNCH_CHECK("csam_flash_attn");
  return CSAM_OK;
}
