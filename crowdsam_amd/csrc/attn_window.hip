// csam_win_attn: ViTDet 14x14 windowed attention with decomposed relative position bias.
//
// Reference: segment_anything_cs/modeling/image_encoder.py:224-240 (Attention.forward),
// :243-289 (window_partition/unpartition), :325-361 (add_decomposed_rel_pos).
//
// One workgroup (4 waves) per (window, head).  The window gather/scatter is folded into the loads
// and stores: qkv is the token-major [4096, 3*D] GEMM output over the 64x64 grid, and the 6 padded
// rows/cols of the edge windows are synthesised in-kernel -- after LayerNorm the pad tokens are
// zero, so their q/k/v equal the qkv bias (SURVEY.md trap 4: pad tokens ARE real keys, not masked).
//
// Relative-position bias without a materialised [196,196] add: S = scale*(q.k) + Th[q,kh] + Tw[q,kw]
// is evaluated as ONE MFMA contraction over an extended K dimension
//     [ q (64) | Th/scale hi (14) | Tw/scale hi (14) | Th/scale lo (14) | Tw/scale lo (14) | 0 (8) ]
//   x [ k (64) | onehot(kh)       | onehot(kw)       | onehot(kh)       | onehot(kw)       | 0     ]
// with Th[q,kh] = q . rel_pos_h[qh-kh+13] (unscaled q, image_encoder.py:349-359) split into
// fp16 hi+lo parts so the bias keeps ~22 bits.  The one-hot key fragments are generated in
// registers.  Softmax runs in the "swapped" MFMA orientation (keys on the accumulator rows) so a
// lane owns 4 keys x 13 tiles of ONE query: row max/sum are in-lane plus two cross-lane shuffles.
// P is converted to fp16 in registers and fed straight back as the B operand of P.V with V held
// transposed in LDS.
#include "csam_common.h"

namespace {

constexpr int WS = 14, NTOK = 196, NPAD = 208;      // 13 tiles of 16
constexpr int QE_LD = 136;                          // halfs per Qe row: 128 + 8 pad (272 B)
constexpr int KE_LD = 72;                           // halfs per K row: 64 + 8 pad (144 B)
constexpr int VT_LD = 232;                          // halfs per Vt row: 224 + 8 pad (464 B)
constexpr int QE_BYTES = NPAD * QE_LD * 2;          // 56576
constexpr int KE_BYTES = NPAD * KE_LD * 2;          // 29952
constexpr int VT_BYTES = 64 * VT_LD * 2;            // 29696
constexpr int SMEM_BYTES = QE_BYTES + KE_BYTES + VT_BYTES;

__global__ __launch_bounds__(256) void win_attn_kernel(const half_t* __restrict__ qkv,
                                                       const float* __restrict__ qkv_bias,
                                                       const half_t* __restrict__ relcat,
                                                       half_t* __restrict__ out, int D, int nH,
                                                       float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Qe = (half_t*)smem;
  half_t* Ke = (half_t*)(smem + QE_BYTES);
  half_t* Vt = (half_t*)(smem + QE_BYTES + KE_BYTES);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int head = blockIdx.x % nH, win = blockIdx.x / nH;
  const int wy = win / 5, wx = win % 5;
  const long ld = 3L * D;

  // ---- stage q, k (row-major) and v (transposed); pad tokens take the bias values
  for (int it = tid; it < NPAD * 3 * 8; it += 256) {
    const int i = it / 24, r = it % 24, which = r >> 3, ch = r & 7;
    half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (i < NTOK) {
      const int y = wy * WS + i / WS, x = wx * WS + i % WS;
      const int col = which * D + head * 64 + ch * 8;
      if (y < 64 && x < 64) {
        v = *(const half8_t*)(qkv + (long)(y * 64 + x) * ld + col);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (half_t)qkv_bias[col + e];
      }
    }
    if (which == 0) {
      *(half8_t*)(Qe + i * QE_LD + ch * 8) = v;
    } else if (which == 1) {
      *(half8_t*)(Ke + i * KE_LD + ch * 8) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) Vt[(ch * 8 + e) * VT_LD + i] = v[e];
    }
  }
  // zero the key columns 208..231 of Vt (keys 196..207 were zero-filled above)
  for (int it = tid; it < 64 * 24; it += 256) Vt[(it / 24) * VT_LD + NPAD + it % 24] = (half_t)0.f;
  __syncthreads();

  // ---- rel-pos extension columns of Qe: [64..77] Th hi, [78..91] Tw hi, [92..105] Th lo,
  //      [106..119] Tw lo, [120..127] zero.  Values are T/scale so that S = scale * acc.
  // T[q][j] = q . relcat[j] (rows 0..26 rel_pos_h, 27..53 rel_pos_w, fp16 like every other weight) is one
  // small MFMA product per query tile: D[j][q] = relcat[j][:] . q[:], and each lane scatters its 16 values
  // to the (kh | kw) slot they belong to:  kh = qh + 13 - j,  kw = qw + 13 - (j - 27).
  const int fr = lane & 15, fg = lane >> 4;
  const float inv_scale = 1.0f / scale;
  for (int it = tid; it < NPAD * 8; it += 256) {           // zero all extension columns first
    *(half8_t*)(Qe + (it >> 3) * QE_LD + 64 + (it & 7) * 8) = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
  }
  __syncthreads();
  for (int rt = wave; rt < 13; rt += 4) {
    half8_t qf0[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf0[ks] = *(const half8_t*)(Qe + (rt * 16 + fr) * QE_LD + (ks * 4 + fg) * 8);
    const int qi = rt * 16 + fr;
    const int qh = qi / WS, qw = qi % WS;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      floatx4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const half8_t rf = *(const half8_t*)(relcat + (nt * 16 + fr) * 64 + (ks * 4 + fg) * 8);
        t = __builtin_amdgcn_mfma_f32_16x16x32_f16(rf, qf0[ks], t, 0, 0, 0);
      }
      if (qi < NTOK) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = nt * 16 + fg * 4 + r;
          int col = -1;
          if (j < 27) {
            const int kh = qh + 13 - j;
            if (kh >= 0 && kh < WS) col = kh;
          } else if (j < 54) {
            const int kw = qw + 13 - (j - 27);
            if (kw >= 0 && kw < WS) col = WS + kw;
          }
          if (col >= 0) {
            const float tv = t[r] * inv_scale;
            const half_t hi = (half_t)tv;
            Qe[qi * QE_LD + 64 + col] = hi;
            Qe[qi * QE_LD + 92 + col] = (half_t)(tv - (float)hi);
          }
        }
      }
    }
  }
  __syncthreads();

  const float sl2 = scale * 1.4426950408889634f;  // work in base 2

  for (int rt = wave; rt < 13; rt += 4) {
    half8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const half8_t*)(Qe + (rt * 16 + fr) * QE_LD + (ks * 4 + fg) * 8);

    floatx4 p[14];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 13; ++kt) {
      floatx4 acc = {0.f, 0.f, 0.f, 0.f};
      const int key = kt * 16 + fr;          // the key this lane supplies as an A-operand row
      const half_t* kr = Ke + key * KE_LD;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const half8_t kf = *(const half8_t*)(kr + (ks * 4 + fg) * 8);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], acc, 0, 0, 0);
      }
      const int kh = key / WS, kw = key % WS;
      const bool kvalid = key < NTOK;
#pragma unroll
      for (int ks = 2; ks < 4; ++ks) {
        half8_t kf;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int c = (ks - 2) * 32 + fg * 8 + e;
          const bool one = kvalid && (c == kh || c == 14 + kw || c == 28 + kh || c == 42 + kw);
          kf[e] = one ? (half_t)1.0f : (half_t)0.0f;
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], acc, 0, 0, 0);
      }
      // lane now holds S_raw[key = kt*16 + fg*4 + j][query = rt*16 + fr]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = kt * 16 + fg * 4 + j;
        const float s = (kk < NTOK) ? acc[j] * sl2 : -INFINITY;
        acc[j] = s;
        mx = fmaxf(mx, s);
      }
      p[kt] = acc;
    }
    p[13] = floatx4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 14; ++kt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e = csam_exp2(p[kt][j] - mx);
        p[kt][j] = e;
        sum += e;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    half8_t pf[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[s][e] = (half_t)p[2 * s][e];
        pf[s][4 + e] = (half_t)p[2 * s + 1][e];
      }
    }
    const int qi = rt * 16 + fr;
    const int y = wy * WS + qi / WS, x = wx * WS + qi % WS;
    const bool store = qi < NTOK && y < 64 && x < 64;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      floatx4 o = {0.f, 0.f, 0.f, 0.f};
      const half_t* vr = Vt + (dt * 16 + fr) * VT_LD + fg * 4;
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const half4_t v0 = *(const half4_t*)(vr + 32 * s);
        const half4_t v1 = *(const half4_t*)(vr + 32 * s + 16);
        const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[s], o, 0, 0, 0);
      }
      if (store) {
        half4_t h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (half_t)(o[j] * inv);
        *(half4_t*)(out + (long)(y * 64 + x) * D + head * 64 + dt * 16 + fg * 4) = h;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// v2 (round 2).  The round-1 kernel above spent most of its time OUTSIDE the matrix pipe: the one-hot key fragments
// that inject the rel-pos bias through the QK^T MFMA were rebuilt with ~1700 VALU instructions per 16-query tile
// (a wave64 VALU instruction occupies the SIMD for 4 cycles on gfx950), and its 116 KB of LDS (q + 64 extension
// columns per query) allowed one workgroup per CU, i.e. two rounds of the 25 x nH grid with four waves per CU.
// Here:
//   * q is never staged: a wave loads the fragments of its query tile straight from the qkv GEMM output (or the
//     bias for pad tokens); only K (row-major) and V^T live in LDS -> 67 KB, two workgroups per CU, the whole
//     grid (400 workgroups for ViT-L) is resident at once;
//   * the decomposed bias Th[q][kh] + Tw[q][kw] is computed once per query tile by the same small MFMA
//     (relcat . q), kept in fp32 in a 2 KB per-wave LDS table and ADDED to the scores after the QK^T product:
//     a lane's four keys are consecutive, so it needs Th for at most two key rows and four consecutive Tw
//     entries (the Tw row is stored with its first four entries repeated, so the run never wraps) -- two 8-byte
//     LDS reads, one select and two adds per score instead of two extra MFMAs per key tile;
//   * V^T is staged two keys at a time (4-byte transposed writes).
// Arithmetic: S = scale * (q.k) + Th + Tw with fp32 bias (round 1: hi+lo fp16 through the MFMA), base-2 softmax
// over the 196 keys INCLUDING the pad tokens of edge windows (SURVEY.md trap 4), P.V with P in fp16.
// ---------------------------------------------------------------------------------------------------------------
constexpr int T_LD = 30;                             // floats per query row of the bias table: Th[14] | Tw[14 + 2 spare]
constexpr int TW_OFF = 14;                           //   Tw at 14..27, its first 4 entries repeated at 28.. (see below)
constexpr int T2_LD = 34;                            // Th 0..13 | Tw 14..27 | Tw[0..3] again 28..31 | pad 2 -> 136 B rows
constexpr int T_BYTES = 4 * 16 * T2_LD * 4;          // per-wave tables
constexpr int SMEM2_BYTES = KE_BYTES + VT_BYTES + T_BYTES;

// The three MFMA groups of the kernel (bias table, scores, P.V) are inline-asm blocks with tied accumulators: built from
// the MFMA builtins hipcc selects the VGPR form here (launch bounds <= 256 registers) and recycled SrcC quads as
// ds_read destinations two instructions behind the MFMA that reads them -- the hazard analysed in attn_flash.hip
// (tools/lint_mfma_srcc.py flagged six such loads in this kernel).
#ifndef CSAM_WA_ABL
#define CSAM_WA_ABL 0      /* developer ablations: 1 = staging only, 2 = no staging */
#endif
#include "attn_window_asm.inc"

__global__ __launch_bounds__(256, 2) void win_attn2_kernel(const half_t* __restrict__ qkv,
                                                           const float* __restrict__ qkv_bias,
                                                           const half_t* __restrict__ relcat,
                                                           half_t* __restrict__ out, int D, int nH, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Ke = (half_t*)smem;
  half_t* Vt = (half_t*)(smem + KE_BYTES);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float* Tt = (float*)(smem + KE_BYTES + VT_BYTES) + wave * 16 * T2_LD;
  const int head = blockIdx.x % nH, win = blockIdx.x / nH;
  const int wy = win / 5, wx = win % 5;
  const long ld = 3L * D;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- stage K (row-major, 16-B writes) and V^T (two keys per 4-B write); pad tokens take the qkv bias, keys >= 196 zero.
  // A thread keeps ONE (k | v, 8-channel chunk) for all its token pairs (256 % 16 == 0), so the pad value is a per-thread
  // constant and the 14 loads of a thread are branch-free (clamped addresses, selected afterwards) and ALL IN FLIGHT AT ONCE:
  // round 4 found the compiler's branchy loop waiting for every load before issuing the next -- 13 serial round trips,
  // 9.5 us of the kernel's 29 with qkv in L2 and most of its 49 us behind the qkv GEMM, where the operands come from HBM
  const bool edge = (wy == 4) | (wx == 4);            // uniform: only these windows have pad tokens
#if CSAM_WA_ABL == 2
  if (qkv == nullptr)
#endif
  {
    const int r = tid & 15, which = 1 + (r >> 3), ch = r & 7;
    const int col = which * D + head * 64 + ch * 8;
    half8_t bv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (edge) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = (half_t)qkv_bias[col + e];
    }
    half8_t v[7][2];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int ip = (tid >> 4) + 16 * i;             // token pair (104 of them: the last round is half empty)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int tok = min(2 * ip + t, NTOK - 1);
        const int y = min(wy * WS + tok / WS, 63), x = min(wx * WS + tok % WS, 63);
        v[i][t] = *(const half8_t*)(qkv + (long)(y * 64 + x) * ld + col);
      }
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int ip = (tid >> 4) + 16 * i;
      if (ip < NPAD / 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int tok = 2 * ip + t;
          const bool pad = (wy * WS + tok / WS >= 64) | (wx * WS + tok % WS >= 64);
          if (tok >= NTOK) v[i][t] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
          else if (pad) v[i][t] = bv;
        }
        if (which == 1) {
          *(half8_t*)(Ke + (2 * ip) * KE_LD + ch * 8) = v[i][0];
          *(half8_t*)(Ke + (2 * ip + 1) * KE_LD + ch * 8) = v[i][1];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) *(half2_t*)(Vt + (ch * 8 + e) * VT_LD + 2 * ip) = half2_t{v[i][0][e], v[i][1][e]};
        }
      }
    }
  }
  for (int it = tid; it < 64 * 12; it += 256)       // keys 208..231 of V^T (read by the last, half-empty k-step)
    *(half2_t*)(Vt + (it / 12) * VT_LD + NPAD + 2 * (it % 12)) = half2_t{0, 0};

  // ---- per-lane constants: rel-pos fragments, and for every key tile where this lane's 4 keys sit in the window
  half8_t rf[4][2];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) rf[nt][ks] = *(const half8_t*)(relcat + (nt * 16 + fr) * 64 + (ks * 4 + fg) * 8);
  int koff[13];                                      // (kh0 << 8) | kw0 of key0 = 16 kt + 4 fg  (kw0 is even)
#pragma unroll
  for (int kt = 0; kt < 13; ++kt) {
    const int key0 = kt * 16 + fg * 4;
    koff[kt] = ((key0 / WS) << 8) | (key0 % WS);
  }
  // LDS byte addresses of this lane's fragments: K row fr of key tile 0, 16-B chunk fg (key tile: +16 rows, k-step: +64 B);
  // V^T row dt*16 + fr at keys 4 fg (k-step s2: +64 B, second half of the fragment: +32 B)
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem;
  const unsigned kaddr = lds0 + fr * (KE_LD * 2) + fg * 16;
  unsigned vaddr[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) vaddr[dt] = lds0 + KE_BYTES + (dt * 16 + fr) * (VT_LD * 2) + fg * 8;

  const float sl2 = scale * 1.4426950408889634f;     // scores in base-2 units
  const float l2e = 1.4426950408889634f;
#if CSAM_WA_ABL == 1
  if (scale != 123.f) { if (tid == 0) out[blockIdx.x] = Ke[win] + Vt[head]; return; }
#endif
  // q fragments: fetched a tile ahead, branch-free (clamped address, pad / out-of-range selected afterwards)
  half8_t qb[2] = {half8_t{0, 0, 0, 0, 0, 0, 0, 0}, half8_t{0, 0, 0, 0, 0, 0, 0, 0}};
  if (edge) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) qb[ks][e] = (half_t)qkv_bias[head * 64 + (ks * 4 + fg) * 8 + e];
  }
  auto load_q = [&](int rt, half8_t (&q)[2]) {
    const int qi = min(rt * 16 + fr, NTOK - 1);
    const int y = min(wy * WS + qi / WS, 63), x = min(wx * WS + qi % WS, 63);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) q[ks] = *(const half8_t*)(qkv + (long)(y * 64 + x) * ld + head * 64 + (ks * 4 + fg) * 8);
  };
  half8_t qn[2];
  load_q(wave, qn);
  __syncthreads();                                    // K / V^T staged (the first q fragments are already on their way)
  for (int rt = wave; rt < 13; rt += 4) {
    const int qi = rt * 16 + fr;
    const int qh = qi / WS, qw = qi % WS;
    const int y = wy * WS + qh, x = wx * WS + qw;
    const bool inside = qi < NTOK && y < 64 && x < 64;
    half8_t qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[ks] = inside ? qn[ks] : (qi < NTOK ? qb[ks] : half8_t{0, 0, 0, 0, 0, 0, 0, 0});
    load_q(min(rt + 4, 12), qn);                      // next tile (the last round re-reads tile 12: harmless)
    // bias table of this query tile: T[j][q] = relcat[j] . q, scattered to Th[kh = qh + 13 - j], Tw[kw = qw + 13 - (j - 27)]
    floatx4 t4[4];
    win_bias_mfma(t4, rf, qf);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const floatx4 t = t4[nt];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = nt * 16 + fg * 4 + r;
        int col = -1;
        if (j < 27) {
          const int kh = qh + 13 - j;
          if (kh >= 0 && kh < WS) col = kh;
        } else if (j < 54) {
          const int kw = qw + 13 - (j - 27);
          if (kw >= 0 && kw < WS) col = TW_OFF + kw;
        }
        if (col >= 0) {
          const float tv = t[r] * l2e;
          Tt[fr * T2_LD + col] = tv;
          if (col >= TW_OFF && col < TW_OFF + 4) Tt[fr * T2_LD + col + WS] = tv;     // Tw[0..3] repeated after Tw[13]
        }
      }
    }
    // (the table is wave-private: only this wave's own LDS writes must have landed before the reads below)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    floatx4 p[14];
    float mx = -INFINITY;
    const float* trow = Tt + fr * T2_LD;
    win_scores_mfma(p, qf, kaddr);
#pragma unroll
    for (int kt = 0; kt < 13; ++kt) {
      floatx4 acc = p[kt];
      // lane holds S_raw[key = kt*16 + fg*4 + j][query = rt*16 + fr]; its keys are (kh0, kw0 + j), wrapping to kh0 + 1
      const int kh0 = koff[kt] >> 8, kw0 = koff[kt] & 255;
      const float2_t th = *(const float2_t*)(trow + (kh0 & ~1));                  // Th[kh0 & ~1], Th[(kh0 & ~1) + 1]
      const float th0 = (kh0 & 1) ? th[1] : th[0];
      const float th1 = (kh0 & 1) ? trow[min(kh0 + 1, WS - 1)] : th[1];
      const float2_t twa = *(const float2_t*)(trow + TW_OFF + kw0);
      const float2_t twb = *(const float2_t*)(trow + TW_OFF + kw0 + 2);
      const float tw[4] = {twa[0], twa[1], twb[0], twb[1]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = kt * 16 + fg * 4 + j;
        const float bias = ((kw0 + j >= WS) ? th1 : th0) + tw[j];
        const float sv = (kk < NTOK) ? fmaf(acc[j], sl2, bias) : -INFINITY;
        acc[j] = sv;
        mx = fmaxf(mx, sv);
      }
      p[kt] = acc;
    }
    p[13] = floatx4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 14; ++kt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e = csam_exp2(p[kt][j] - mx);
        p[kt][j] = e;
        sum += e;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    half8_t pf[7];
#pragma unroll
    for (int s2 = 0; s2 < 7; ++s2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[s2][e] = (half_t)p[2 * s2][e];
        pf[s2][4 + e] = (half_t)p[2 * s2 + 1][e];
      }
    }
    floatx4 o4[4];
    win_pv_mfma(o4, pf, vaddr);
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const floatx4 o = o4[dt];
      if (inside) {
        half4_t h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (half_t)(o[j] * inv);
        *(half4_t*)(out + (long)(y * 64 + x) * D + head * 64 + dt * 16 + fg * 4) = h;
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// head_dim 80 (ViT-H, round 3): the same kernel with three k-steps -- 32 + 32 + 16, the last one a 16x16x16 MFMA on
// 8-byte fragments, so no dimension is padded -- and five 16-dim output tiles.  K rows are 176 B, V^T has 80 rows: 91 KB
// of LDS with per-wave bias tables, one workgroup per CU, so the workgroup has EIGHT waves (the 13 query tiles of a window
// in two rounds) and the 25 x nH grid runs in two rounds of 256.  relcat is [64, 80] here.
// ---------------------------------------------------------------------------------------------------------------
constexpr int HD80 = 80;
constexpr int KE80_LD = 88;                          // halfs per K row: 80 + 8 pad (176 B)
constexpr int KE80_BYTES = NPAD * KE80_LD * 2;       // 36608
constexpr int VT80_BYTES = HD80 * VT_LD * 2;         // 37120
constexpr int NW80 = 8;
constexpr int T80_BYTES = NW80 * 16 * T2_LD * 4;
constexpr int SMEM80_BYTES = KE80_BYTES + VT80_BYTES + T80_BYTES;

__global__ __launch_bounds__(64 * NW80, 1) void win_attn2_hd80_kernel(const half_t* __restrict__ qkv,
                                                                      const float* __restrict__ qkv_bias,
                                                                      const half_t* __restrict__ relcat,
                                                                      half_t* __restrict__ out, int D, int nH, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Ke = (half_t*)smem;
  half_t* Vt = (half_t*)(smem + KE80_BYTES);
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  float* Tt = (float*)(smem + KE80_BYTES + VT80_BYTES) + wave * 16 * T2_LD;
  const int head = blockIdx.x % nH, win = blockIdx.x / nH;
  const int wy = win / 5, wx = win % 5;
  const long ld = 3L * D;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- stage K (row-major, 16-B writes) and V^T (two keys per 4-B write); pad tokens take the qkv bias, keys >= 196 zero.
  // As in win_attn2_kernel: a thread keeps one (k | v, 8-channel chunk) -- 500 of the 512 threads, 25 token pairs per round,
  // five rounds -- so its ten loads are branch-free and in flight together
  const bool edge = (wy == 4) | (wx == 4);
  if (tid < 500) {
    const int r = tid % 20, which = 1 + r / 10, ch = r % 10;
    const int col = which * D + head * HD80 + ch * 8;
    half8_t bv = {0, 0, 0, 0, 0, 0, 0, 0};
    if (edge) {
#pragma unroll
      for (int e = 0; e < 8; ++e) bv[e] = (half_t)qkv_bias[col + e];
    }
    half8_t v[5][2];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int ip = tid / 20 + 25 * i;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int tok = min(2 * ip + t, NTOK - 1);
        const int y = min(wy * WS + tok / WS, 63), x = min(wx * WS + tok % WS, 63);
        v[i][t] = *(const half8_t*)(qkv + (long)(y * 64 + x) * ld + col);
      }
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int ip = tid / 20 + 25 * i;
      if (ip < NPAD / 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int tok = 2 * ip + t;
          const bool pad = (wy * WS + tok / WS >= 64) | (wx * WS + tok % WS >= 64);
          if (tok >= NTOK) v[i][t] = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
          else if (pad) v[i][t] = bv;
        }
        if (which == 1) {
          *(half8_t*)(Ke + (2 * ip) * KE80_LD + ch * 8) = v[i][0];
          *(half8_t*)(Ke + (2 * ip + 1) * KE80_LD + ch * 8) = v[i][1];
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) *(half2_t*)(Vt + (ch * 8 + e) * VT_LD + 2 * ip) = half2_t{v[i][0][e], v[i][1][e]};
        }
      }
    }
  }
  for (int it = tid; it < HD80 * 12; it += 64 * NW80)   // keys 208..231 of V^T (read by the last, half-empty k-step)
    *(half2_t*)(Vt + (it / 12) * VT_LD + NPAD + 2 * (it % 12)) = half2_t{0, 0};

  half8_t rf[4][2];
  half4_t rg[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) rf[nt][ks] = *(const half8_t*)(relcat + (nt * 16 + fr) * HD80 + (ks * 4 + fg) * 8);
    rg[nt] = *(const half4_t*)(relcat + (nt * 16 + fr) * HD80 + 64 + fg * 4);
  }
  int koff[13];
#pragma unroll
  for (int kt = 0; kt < 13; ++kt) {
    const int key0 = kt * 16 + fg * 4;
    koff[kt] = ((key0 / WS) << 8) | (key0 % WS);
  }
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)smem;
  const unsigned kaddr = lds0 + fr * (KE80_LD * 2) + fg * 16;
  const unsigned kaddr16 = lds0 + fr * (KE80_LD * 2) + 128 + fg * 8;
  unsigned vaddr[5];
#pragma unroll
  for (int dt = 0; dt < 5; ++dt) vaddr[dt] = lds0 + KE80_BYTES + (dt * 16 + fr) * (VT_LD * 2) + fg * 8;

  const float sl2 = scale * 1.4426950408889634f;
  const float l2e = 1.4426950408889634f;
  // q fragments a tile ahead, branch-free (see win_attn2_kernel)
  half8_t qb[2] = {half8_t{0, 0, 0, 0, 0, 0, 0, 0}, half8_t{0, 0, 0, 0, 0, 0, 0, 0}};
  half4_t qbg = {0, 0, 0, 0};
  if (edge) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) qb[ks][e] = (half_t)qkv_bias[head * HD80 + (ks * 4 + fg) * 8 + e];
#pragma unroll
    for (int e = 0; e < 4; ++e) qbg[e] = (half_t)qkv_bias[head * HD80 + 64 + fg * 4 + e];
  }
  auto load_q = [&](int rt, half8_t (&q)[2], half4_t& g) {
    const int qi = min(rt * 16 + fr, NTOK - 1);
    const int y = min(wy * WS + qi / WS, 63), x = min(wx * WS + qi % WS, 63);
    const half_t* src = qkv + (long)(y * 64 + x) * ld + head * HD80;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) q[ks] = *(const half8_t*)(src + (ks * 4 + fg) * 8);
    g = *(const half4_t*)(src + 64 + fg * 4);
  };
  half8_t qn[2];
  half4_t qng;
  load_q(min(wave, 12), qn, qng);
  __syncthreads();
  for (int rt = wave; rt < 13; rt += NW80) {
    const int qi = rt * 16 + fr;
    const int qh = qi / WS, qw = qi % WS;
    const int y = wy * WS + qh, x = wx * WS + qw;
    const bool inside = qi < NTOK && y < 64 && x < 64;
    half8_t qf[2];
    half4_t qg = inside ? qng : (qi < NTOK ? qbg : half4_t{0, 0, 0, 0});
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
      qf[ks] = inside ? qn[ks] : (qi < NTOK ? qb[ks] : half8_t{0, 0, 0, 0, 0, 0, 0, 0});
    load_q(min(rt + NW80, 12), qn, qng);
    floatx4 t4[4];
    win_bias_mfma80(t4, rf, rg, qf, qg);
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const floatx4 t = t4[nt];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = nt * 16 + fg * 4 + r;
        int col = -1;
        if (j < 27) {
          const int kh = qh + 13 - j;
          if (kh >= 0 && kh < WS) col = kh;
        } else if (j < 54) {
          const int kw = qw + 13 - (j - 27);
          if (kw >= 0 && kw < WS) col = TW_OFF + kw;
        }
        if (col >= 0) {
          const float tv = t[r] * l2e;
          Tt[fr * T2_LD + col] = tv;
          if (col >= TW_OFF && col < TW_OFF + 4) Tt[fr * T2_LD + col + WS] = tv;
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    floatx4 p[14];
    float mx = -INFINITY;
    const float* trow = Tt + fr * T2_LD;
    win_scores_mfma80(p, qf, qg, kaddr, kaddr16);
#pragma unroll
    for (int kt = 0; kt < 13; ++kt) {
      floatx4 acc = p[kt];
      const int kh0 = koff[kt] >> 8, kw0 = koff[kt] & 255;
      const float2_t th = *(const float2_t*)(trow + (kh0 & ~1));
      const float th0 = (kh0 & 1) ? th[1] : th[0];
      const float th1 = (kh0 & 1) ? trow[min(kh0 + 1, WS - 1)] : th[1];
      const float2_t twa = *(const float2_t*)(trow + TW_OFF + kw0);
      const float2_t twb = *(const float2_t*)(trow + TW_OFF + kw0 + 2);
      const float tw[4] = {twa[0], twa[1], twb[0], twb[1]};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kk = kt * 16 + fg * 4 + j;
        const float bias = ((kw0 + j >= WS) ? th1 : th0) + tw[j];
        const float sv = (kk < NTOK) ? fmaf(acc[j], sl2, bias) : -INFINITY;
        acc[j] = sv;
        mx = fmaxf(mx, sv);
      }
      p[kt] = acc;
    }
    p[13] = floatx4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 14; ++kt) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e = csam_exp2(p[kt][j] - mx);
        p[kt][j] = e;
        sum += e;
      }
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;

    half8_t pf[7];
#pragma unroll
    for (int s2 = 0; s2 < 7; ++s2) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        pf[s2][e] = (half_t)p[2 * s2][e];
        pf[s2][4 + e] = (half_t)p[2 * s2 + 1][e];
      }
    }
    floatx4 o5[5];
    win_pv_mfma80(o5, pf, vaddr);
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) {
      const floatx4 o = o5[dt];
      if (inside) {
        half4_t h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (half_t)(o[j] * inv);
        *(half4_t*)(out + (long)(y * 64 + x) * D + head * HD80 + dt * 16 + fg * 4) = h;
      }
    }
  }
}

}  // namespace

extern "C" int csam_win_attn(void* stream, const void* qkv_f16, const float* qkv_bias,
                             const void* relcat_f16, void* out_f16, int D, int nH, float scale) {
  CSAM_REQUIRE(qkv_f16 && qkv_bias && relcat_f16 && out_f16, "csam_win_attn: null pointer");
  CSAM_REQUIRE(nH > 0 && (D == nH * 64 || D == nH * 80), "csam_win_attn: head_dim must be 64 or 80 (D=%d nH=%d)", D, nH);
  if (D == nH * 80) {                                   // ViT-H: relcat is [64, 80]
    static csam_once_t set80;
    if (csam_first_call(set80))
      hipFuncSetAttribute((const void*)win_attn2_hd80_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM80_BYTES);
    hipLaunchKernelGGL(win_attn2_hd80_kernel, dim3(25 * nH), dim3(64 * NW80), SMEM80_BYTES, (hipStream_t)stream,
                       (const half_t*)qkv_f16, qkv_bias, (const half_t*)relcat_f16, (half_t*)out_f16, D, nH, scale);
    CSAM_LAUNCH_CHECK("csam_win_attn");
    return CSAM_OK;
  }
  static int version = -1;
  if (version < 0) {
    const char* e = getenv("CSAM_WIN_ATTN");          // 1 = the round-1 kernel (A/B and debugging)
    version = e ? atoi(e) : 2;
  }
  static csam_once_t set64;
  if (csam_first_call(set64)) {
    hipFuncSetAttribute((const void*)win_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM_BYTES);
    hipFuncSetAttribute((const void*)win_attn2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
  }
  if (version == 1)
    hipLaunchKernelGGL(win_attn_kernel, dim3(25 * nH), dim3(256), SMEM_BYTES, (hipStream_t)stream,
                       (const half_t*)qkv_f16, qkv_bias, (const half_t*)relcat_f16, (half_t*)out_f16, D, nH, scale);
  else
    hipLaunchKernelGGL(win_attn2_kernel, dim3(25 * nH), dim3(256), SMEM2_BYTES, (hipStream_t)stream,
                       (const half_t*)qkv_f16, qkv_bias, (const half_t*)relcat_f16, (half_t*)out_f16, D, nH, scale);
  CSAM_LAUNCH_CHECK("csam_win_attn");
  return CSAM_OK;
}
