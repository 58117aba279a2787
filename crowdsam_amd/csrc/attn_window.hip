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
// Image batch: the grid is 25 x nH x n_images workgroups; image b's tokens are rows b*4096 .. b*4096+4095 of qkv / out
// (the image-batched encoder pass of crowdsam_amd/encoder.py: one launch for the crops / look-ahead frames of a pass).
//
// (The round-1 kernel -- rel-pos bias injected through one-hot MFMA operands, q staged in LDS, 116 KB, one workgroup
// per CU -- measured 55.6 us against 28-34 us for the kernel below and is gone; profiles/r02_attention.txt.)
#include "csam_common.h"

namespace {

constexpr int WS = 14, NTOK = 196, NPAD = 208;      // 13 tiles of 16
constexpr int KE_LD = 72;                           // halfs per K row: 64 + 8 pad (144 B)
constexpr int VT_LD = 232;                          // halfs per Vt row: 224 + 8 pad (464 B)
constexpr int KE_BYTES = NPAD * KE_LD * 2;          // 29952
constexpr int VT_BYTES = 64 * VT_LD * 2;            // 29696

// ---------------------------------------------------------------------------------------------------------------
// The kernel (round 2).  The round-1 kernel spent most of its time OUTSIDE the matrix pipe: the one-hot key fragments
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
  const int head = blockIdx.x % nH, gwin = blockIdx.x / nH;
  const int win = gwin % 25;                           // window of image gwin / 25 (grid: 25 x nH x n_images)
  const int wy = win / 5, wx = win % 5;
  const long ld = 3L * D;
  qkv += (long)(gwin / 25) * 4096 * ld;
  out += (long)(gwin / 25) * 4096 * D;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- stage K (row-major, 16-B writes) and V^T (two keys per 4-B write); pad tokens take the qkv bias, keys >= 196 zero.
  // A thread keeps ONE (k | v, 8-channel chunk) for all its token pairs (256 % 16 == 0), so the pad value is a per-thread
  // constant and the 14 loads of a thread are branch-free (clamped addresses, selected afterwards) and ALL IN FLIGHT AT ONCE:
  // round 4 found the compiler's branchy loop waiting for every load before issuing the next -- 13 serial round trips,
  // 9.5 us of the kernel's 29 with qkv in L2 and most of its 49 us behind the qkv GEMM, where the operands come from HBM
  const bool edge = (wy == 4) | (wx == 4);            // uniform: only these windows have pad tokens
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
    mx = csam_max_x16(mx);
    mx = csam_max_x32(mx);
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
    sum = csam_sum_x16(sum);
    sum = csam_sum_x32(sum);
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
  const int head = blockIdx.x % nH, gwin = blockIdx.x / nH;
  const int win = gwin % 25;                           // window of image gwin / 25 (grid: 25 x nH x n_images)
  const int wy = win / 5, wx = win % 5;
  const long ld = 3L * D;
  qkv += (long)(gwin / 25) * 4096 * ld;
  out += (long)(gwin / 25) * 4096 * D;
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
    mx = csam_max_x16(mx);
    mx = csam_max_x32(mx);
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
    sum = csam_sum_x16(sum);
    sum = csam_sum_x32(sum);
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

extern "C" int csam_win_attn_batched(void* stream, const void* qkv_f16, const float* qkv_bias, const void* relcat_f16,
                                     void* out_f16, int D, int nH, float scale, int n_images) {
  CSAM_REQUIRE(qkv_f16 && qkv_bias && relcat_f16 && out_f16, "csam_win_attn: null pointer");
  CSAM_REQUIRE(nH > 0 && (D == nH * 64 || D == nH * 80), "csam_win_attn: head_dim must be 64 or 80 (D=%d nH=%d)", D, nH);
  CSAM_REQUIRE(n_images >= 1 && n_images <= 64, "csam_win_attn: n_images = %d", n_images);
  static csam_once_t set;
  if (csam_first_call(set)) {
    hipFuncSetAttribute((const void*)win_attn2_hd80_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM80_BYTES);
    hipFuncSetAttribute((const void*)win_attn2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM2_BYTES);
  }
  if (D == nH * 80)                                     // ViT-H: relcat is [64, 80]
    hipLaunchKernelGGL(win_attn2_hd80_kernel, dim3(25 * nH * n_images), dim3(64 * NW80), SMEM80_BYTES, (hipStream_t)stream,
                       (const half_t*)qkv_f16, qkv_bias, (const half_t*)relcat_f16, (half_t*)out_f16, D, nH, scale);
  else
    hipLaunchKernelGGL(win_attn2_kernel, dim3(25 * nH * n_images), dim3(256), SMEM2_BYTES, (hipStream_t)stream,
                       (const half_t*)qkv_f16, qkv_bias, (const half_t*)relcat_f16, (half_t*)out_f16, D, nH, scale);
  CSAM_LAUNCH_CHECK("csam_win_attn");
  return CSAM_OK;
}

extern "C" int csam_win_attn(void* stream, const void* qkv_f16, const float* qkv_bias,
                             const void* relcat_f16, void* out_f16, int D, int nH, float scale) {
  return csam_win_attn_batched(stream, qkv_f16, qkv_bias, relcat_f16, out_f16, D, nH, scale, 1);
}
