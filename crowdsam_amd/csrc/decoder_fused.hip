// Fused decoder kernels (prompt-batch fused, one pass over the per-prompt key state each).
//
// csam_i2t_fused: the whole image->token half of a TwoWayAttentionBlock (transformer.py:186-190)
//     keys_out = LayerNorm4( keys + out_proj( softmax( q(keys+pe) . k(tokens)^T / 4 ) v(tokens) ) )
// for every prompt of the batch in ONE kernel: per 128-token tile
//     [Q = X Wq^T (+ pe Wq^T + bq)]  MFMA, K = 256          (QMODE 1; QMODE 0 loads the hoisted layer-0 Q)
//     -> 7-key attention per (token, head) entirely in the accumulator registers
//     -> out-proj MFMA, K = 128, the fp16 O registers are fed straight back as the B operand
//     -> + bias + residual -> LayerNorm over 256 channels (in-lane + 2 shuffles) -> fp16 store.
// HBM traffic per prompt: read keys (2 MB) + write keys (2 MB) -- SURVEY.md section 8(d)'s R2+W2 --
// instead of the unfused Q-GEMM / attention / out-proj GEMM / LayerNorm chain (14 MB).
//
// Wave layout: 4 waves x 32 tokens, every wave owns ALL output channels of its tokens, so the
// per-head attention (head == one 16-wide N tile) and the 256-wide LayerNorm never cross waves.
// MFMA "swapped" orientation as in gemm_f16.hip: lane (fr = lane&15, fg = lane>>4) holds 4
// consecutive channels fg*4..+3 of N-tile ni for token fr of M-tile mi.  The O->out-proj register
// feed needs the out-proj weight columns permuted on the host:
//     k' = s*32 + g*8 + e   <->   k = (2s + (e>=4))*16 + g*4 + (e&3).
#include "csam_common.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

constexpr int I2T_TOK = 128;                 // tokens per workgroup
constexpr int I2T_STAGE = 64 * 1024;         // operand staging region (bytes)
constexpr int I2T_KV = 2 * 8 * 64 * 8;       // per-head MFMA A-fragment tables of the prompt's k and v^T
constexpr int I2T_PAR = I2T_STAGE + I2T_KV;    // bo | gamma | beta fp32 [3][256]
constexpr int I2T_SMEM = I2T_PAR + 3 * 256 * 4;

struct I2tArgs {
  const half_t* X; long x_bstride;           // keys in  [.,256] (per-prompt stride; 0 = shared src)
  const half_t* Q; long q_bstride;           // QMODE 0: hoisted Q [4096,128] (bias included)
  const half_t* Wq;                          // QMODE 1: [128,256]
  const float* qpe;                          // QMODE 1: pe Wq^T + bq  fp32 [4096,128]
  const half_t* kv_k; const half_t* kv_v;    // [B,7,128] token-side k / v projections
  const half_t* Wo;                          // [256,128], columns permuted (see header)
  const float* bo;                           // [256]
  const float* gamma; const float* beta; float eps;
  half_t* out;                               // keys out [B*4096, 256]
  int T;                                     // 4096
};

template <int QMODE>
__global__ __launch_bounds__(256, 2) void i2t_fused_kernel(I2tArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half4_t* kfr = (half4_t*)(smem + I2T_STAGE);   // [8 heads][64 lanes]: K_h   rows j,   k = dims fg*4..+3
  half4_t* vfr = kfr + 8 * 64;                   // [8 heads][64 lanes]: V_h^T rows d,   k = keys fg*4..+3
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * I2T_TOK;
  const half_t* Xb = p.X + (long)b * p.x_bstride + (long)t0 * 256;
  if (QMODE == 0) {
    for (int c = tid; c < 4096; c += 256) {          // see prefetch_wo below
      const int row = c >> 4, sl = c & 15;
      glds16(p.Wo + (long)row * 128 + ((sl ^ (row & 15)) * 8), smem + (c & ~63) * 16);
    }
  }

  // prompt's token-side k, v -> ready-made 16x16x16 MFMA A fragments (zero rows beyond the 7 keys)
  for (int i = tid; i < 8 * 64; i += 256) {
    const int h = i >> 6, l = i & 63, r16 = l & 15, g4 = (l >> 4) * 4;
    half4_t ka = {0, 0, 0, 0}, va = {0, 0, 0, 0};
    if (r16 < 7) ka = *(const half4_t*)(p.kv_k + ((long)b * 7 + r16) * 128 + h * 16 + g4);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (g4 + r < 7) va[r] = p.kv_v[((long)b * 7 + g4 + r) * 128 + h * 16 + r16];
    kfr[i] = ka;
    vfr[i] = va;
  }
  // Wo' (64 KB) -> staging region: 256 rows x 256 B = 4096 16-B chunks, slot = chunk ^ (row & 15).  QMODE 0 has
  // no phase 1 in that region, so its prefetch is the first thing the workgroup issues (the whole latency
  // hides under the table build and the Q loads); QMODE 1 issues it after phase 1.
  auto prefetch_wo = [&]() {
    for (int c = tid; c < 4096; c += 256) {
      const int row = c >> 4, sl = c & 15;
      glds16(p.Wo + (long)row * 128 + ((sl ^ (row & 15)) * 8), smem + (c & ~63) * 16);
    }
  };
  float* par = (float*)(smem + I2T_PAR);
  if (tid < 256) {
    par[tid] = p.bo[tid];
    par[256 + tid] = p.gamma[tid];
    par[512 + tid] = p.beta[tid];
  }

  floatx4 q[2][8];
  if (QMODE == 1) {
    // ---- phase 1: Q^T = Wq X^T  (tile 128 tokens x 128 n, K = 256), glds double-buffered
    const int srow = lane >> 3, sslot = lane & 7;
    const half_t* x_src[4];
    const half_t* w_src[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = (wave * 4 + i) * 8 + srow;
      const int chunk = sslot ^ (row & 7);
      x_src[i] = Xb + (long)row * 256 + chunk * 8;
      w_src[i] = p.Wq + (long)row * 256 + chunk * 8;
    }
    auto stage = [&](int buf, int k0) {
      char* xb = smem + buf * 32768;
      char* wb = xb + 16384;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int grp = (wave * 4 + i) * 1024;
        glds16(x_src[i] + k0, xb + grp);
        glds16(w_src[i] + k0, wb + grp);
      }
    };
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) q[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int sw = fr & 7;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int cur = kt & 1;
      if (kt + 1 < 4) stage(cur ^ 1, (kt + 1) * 64);
      const char* xb = smem + cur * 32768;
      const char* wb = xb + 16384;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int coff = ((kk * 4 + fg) ^ sw) << 4;
        half8_t xf[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) xf[mi] = *(const half8_t*)(xb + (wave * 32 + mi * 16 + fr) * 128 + coff);
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
          const half8_t wf = *(const half8_t*)(wb + (ni * 16 + fr) * 128 + coff);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            q[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf[mi], q[mi][ni], 0, 0, 0);
        }
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    // + (pe Wq^T + bq)[token]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const float* pq = p.qpe + (long)(t0 + wave * 32 + mi * 16 + fr) * 128 + fg * 4;
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) q[mi][ni] += *(const floatx4*)(pq + ni * 16);
    }
  } else {
    const half_t* Qb = p.Q + (long)b * p.q_bstride;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const half_t* qr = Qb + (long)(t0 + wave * 32 + mi * 16 + fr) * 128 + fg * 4;
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        const half4_t h = *(const half4_t*)(qr + ni * 16);
        q[mi][ni] = floatx4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      }
    }
    __syncthreads();   // fragment tables visible
  }

  // prefetch Wo' (64 KB) into the now-free staging region; it lands while phase 2 runs
  if (QMODE == 1) prefetch_wo();

  // ---- phase 2: softmax(q k^T / 4) v over the 7 token keys, per (token, head = ni), on the matrix
  // cores: S^T = K_h q^T and O^T = V_h^T P^T as 16x16x16 MFMAs whose B operands are the lane's own
  // accumulator registers (q -> fp16, P -> fp16); row max / sum need 2 shuffles each.
  half8_t of[2][4];
  const float sc = 0.25f * 1.4426950408889634f;
  // keys 4 fg + r >= 7 do not exist: a per-lane additive -inf folds the masking into the scale fma (exp2 of -inf is 0)
  floatx4 mb;
#pragma unroll
  for (int r = 0; r < 4; ++r) mb[r] = fg * 4 + r < 7 ? 0.f : -INFINITY;
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) {
    const half4_t ka = kfr[ni * 64 + lane];
    const half4_t va = vfr[ni * 64 + lane];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const floatx4 qq = q[mi][ni];
      const half4_t qb = {(half_t)qq[0], (half_t)qq[1], (half_t)qq[2], (half_t)qq[3]};
      floatx4 sacc = __builtin_amdgcn_mfma_f32_16x16x16f16(ka, qb, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const float s0 = fmaf(sacc[0], sc, mb[0]), s1 = fmaf(sacc[1], sc, mb[1]);
      const float s2 = fmaf(sacc[2], sc, mb[2]), s3 = fmaf(sacc[3], sc, mb[3]);
      float mx = fmaxf(fmaxf(s0, s1), fmaxf(s2, s3));
      mx = csam_max_x16(mx);
      mx = csam_max_x32(mx);
      const float p0 = csam_exp2(s0 - mx), p1 = csam_exp2(s1 - mx);
      const float p2 = csam_exp2(s2 - mx), p3 = csam_exp2(s3 - mx);
      float sum = (p0 + p1) + (p2 + p3);
      sum = csam_sum_x16(sum);
      sum = csam_sum_x32(sum);
      const half4_t pb = {(half_t)p0, (half_t)p1, (half_t)p2, (half_t)p3};
      const floatx4 o = __builtin_amdgcn_mfma_f32_16x16x16f16(va, pb, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int e = 0; e < 4; ++e) of[mi][ni >> 1][(ni & 1) * 4 + e] = (half_t)(o[e] * inv);
    }
  }

  // ---- phase 3: out-proj  D^T = Wo' O^T  (K = 128, N = 256) from the prefetched LDS image of Wo'
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  floatx4 acc[2][16];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 16; ++ni) acc[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
#pragma unroll
    for (int ni = 0; ni < 16; ++ni) {
      if ((ni & 3) == 0) asm volatile("" ::: "memory");   // bound the number of weight fragments in flight
      const int row = ni * 16 + fr;
      const half8_t wf = *(const half8_t*)(smem + row * 256 + (((s * 4 + fg) ^ (row & 15)) << 4));
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, of[mi][s], acc[mi][ni], 0, 0, 0);
    }
  }

  // ---- epilogue: + bias + residual, LayerNorm(256), fp16 store.
  // Residual rows come in and results go out through an LDS tile [128 tokens][512 B] whose 16-B slots are
  // XOR-swizzled with (row & 15): global traffic is whole coalesced 512-B rows (the accumulator layout
  // alone would give 8-byte pieces in 32-B segments -- store-issue bound), LDS accesses conflict-free.
  __syncthreads();                                  // all waves done reading Wo' from LDS
  {
    const half_t* xres = p.X + (long)b * p.x_bstride + (long)t0 * 256;
    for (int c = tid; c < 4096; c += 256) {         // 128 rows x 32 slots, lane-linear glds
      const int row = c >> 5, sl = c & 31;
      glds16(xres + (long)row * 256 + ((sl ^ (row & 15)) * 8), smem + (c & ~63) * 16);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int row = wave * 32 + mi * 16 + fr;
    char* lrow = smem + row * 512;
    float sum = 0.f;
#pragma unroll
    for (int ni = 0; ni < 16; ++ni) {
      if ((ni & 3) == 0) asm volatile("" ::: "memory");
      const floatx4 bb = *(const floatx4*)(par + ni * 16 + fg * 4);
      const int slot = (ni * 2 + (fg >> 1)) ^ (row & 15);
      const half4_t r = *(const half4_t*)(lrow + slot * 16 + (fg & 1) * 8);
      floatx4 v = acc[mi][ni] + bb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += (float)r[e];
        sum += v[e];
      }
      acc[mi][ni] = v;
    }
    sum = csam_sum_x16(sum);
    sum = csam_sum_x32(sum);
    const float mean = sum * (1.f / 256.f);
    float var = 0.f;
#pragma unroll
    for (int ni = 0; ni < 16; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = acc[mi][ni][e] - mean;
        var += d * d;
      }
    var = csam_sum_x16(var);
    var = csam_sum_x32(var);
    const float rstd = rsqrtf(var * (1.f / 256.f) + p.eps);
#pragma unroll
    for (int ni = 0; ni < 16; ++ni) {
      if ((ni & 3) == 0) asm volatile("" ::: "memory");
      const floatx4 g = *(const floatx4*)(par + 256 + ni * 16 + fg * 4);
      const floatx4 be = *(const floatx4*)(par + 512 + ni * 16 + fg * 4);
      half4_t h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (half_t)((acc[mi][ni][e] - mean) * rstd * g[e] + be[e]);
      const int slot = (ni * 2 + (fg >> 1)) ^ (row & 15);
      *(half4_t*)(lrow + slot * 16 + (fg & 1) * 8) = h;   // same positions this lane read: no hazard
    }
  }
  __syncthreads();
  {
    half_t* obase = p.out + ((long)b * p.T + t0) * 256;
    for (int c = tid; c < 4096; c += 256) {
      const int row = c >> 5, sl = c & 31;
      const half8_t v = *(const half8_t*)(smem + c * 16);
      *(half8_t*)(obase + (long)row * 256 + ((sl ^ (row & 15)) * 8)) = v;
    }
  }
}

}  // namespace

extern "C" int csam_i2t_fused(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16,
                              long q_prompt_stride, const void* Wq_f16, const float* qpe, const void* k_f16,
                              const void* v_f16, const void* Wo_perm_f16, const float* bo, const float* gamma,
                              const float* beta, float eps, void* out_f16, int B, int T) {
  CSAM_REQUIRE(X_f16 && k_f16 && v_f16 && Wo_perm_f16 && bo && gamma && beta && out_f16, "csam_i2t_fused: null pointer");
  CSAM_REQUIRE((Q_f16 != nullptr) != (Wq_f16 != nullptr), "csam_i2t_fused: give either Q (hoisted) or Wq");
  CSAM_REQUIRE(!Wq_f16 || qpe, "csam_i2t_fused: qpe required with Wq");
  CSAM_REQUIRE(B > 0 && T > 0 && T % I2T_TOK == 0, "csam_i2t_fused: T must be a multiple of %d", I2T_TOK);
  I2tArgs a;
  a.X = (const half_t*)X_f16; a.x_bstride = x_prompt_stride;
  a.Q = (const half_t*)Q_f16; a.q_bstride = q_prompt_stride;
  a.Wq = (const half_t*)Wq_f16; a.qpe = qpe;
  a.kv_k = (const half_t*)k_f16; a.kv_v = (const half_t*)v_f16;
  a.Wo = (const half_t*)Wo_perm_f16; a.bo = bo; a.gamma = gamma; a.beta = beta; a.eps = eps;
  a.out = (half_t*)out_f16; a.T = T;
  static csam_once_t attr_set;
  if (csam_first_call(attr_set)) {
    hipFuncSetAttribute((const void*)i2t_fused_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, I2T_SMEM);
    hipFuncSetAttribute((const void*)i2t_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, I2T_SMEM);
  }
  dim3 grid(T / I2T_TOK, B);
  if (Wq_f16)
    hipLaunchKernelGGL(i2t_fused_kernel<1>, grid, dim3(256), I2T_SMEM, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(i2t_fused_kernel<0>, grid, dim3(256), I2T_SMEM, (hipStream_t)stream, a);
  CSAM_LAUNCH_CHECK("csam_i2t_fused");
  return CSAM_OK;
}

// =====================================================================================================
// csam_i2t_stream: the same half-block as csam_i2t_fused, restructured as a PERSISTENT, weight-stationary stream.
// PMC + ablation of i2t_fused showed the per-workgroup fixed costs (64-128 KB of weights re-staged through LDS for
// every 64 KB of keys, two dependent memory round trips per tile, 2 workgroups of 4 waves per CU) set its floor, not
// HBM or the VALU.  Here ONE 8-wave workgroup per CU walks a contiguous range of 128-token tiles:
//   * weights live in REGISTERS for the whole launch: wave w owns head w of the Q projection (Wq rows 16w..+15,
//     32 VGPRs) and output channels 32w..+31 of the out projection (Wo rows, 32 VGPRs) + their bias / gamma / beta;
//   * the key tile (64 KB) is DMA'd straight into a double-buffered LDS image one tile ahead (global_load_lds,
//     retired by vmcnt(0) + barrier a whole tile later), every wave reads all of it as MFMA B fragments;
//   * heads split across waves, so the 7-key attention is wave-local (S^T = K_h q^T chained from the Q accumulators,
//     row sum by a ones-fragment MFMA); O is exchanged through 32 KB of the tile buffer for the out projection;
//   * residual = identity MFMA on the tile's own LDS fragments; LayerNorm statistics = in-lane partials reduced over
//     the 4 lane groups by an fp32 16x16x4 MFMA with a ones operand, then across waves through 8 KB of LDS;
//   * the normalised tile is written back over its own LDS image and leaves as whole 512-B rows.
// The token-side k must arrive pre-multiplied by 0.25*log2(e) (host folds it into the projection weights).
// Geometry, templated on the number of waves NW (8 or 4): a tile has 16*NW tokens, a wave owns 8/NW heads and
// 256/NW output channels.  NW = 8: one 512-thread workgroup per CU (128-token tiles, 139 KB LDS).  NW = 4: two
// independent 256-thread workgroups per CU (64-token tiles, 70 KB LDS each) whose phases drift apart, so one's LDS /
// barrier time overlaps the other's MFMA / VALU time.
template <int NW, int MI_>
struct I2S {
  static constexpr int TOK = 16 * MI_;
  static constexpr int NT = 64 * NW;                  // threads
  static constexpr int BUF = TOK * 512;               // one key tile, [TOK][256] fp16
  static constexpr int PART = 2 * BUF;                // float2 [NW waves][TOK tokens]
  static constexpr int PAR = PART + NW * TOK * 8;     // bo | gamma | beta fp32 [3][256]
  static constexpr int SMEM = PAR + 3 * 256 * 4;
  static constexpr int HPW = 8 / NW;                  // heads per wave
  static constexpr int NIO = 16 / NW;                 // 16-channel output tiles per wave
  static constexpr int MI = MI_;                      // 16-token tiles per workgroup tile
  static constexpr int NP = TOK * 32 / NT;            // 16-B pieces per thread and tile
};

// In-loop global memory traffic of the stream kernel goes through inline asm with scalar base + 32-bit lane offset:
// (1) no 64-bit address pairs to spill, (2) the compiler's waitcnt pass does not see these operations.  gfx9 counts
// loads and stores in ONE vmcnt and LLVM, seeing mixed types pending, answers every tracked load's first use with
// vmcnt(0) -- which would drain the next tile's LDS-DMA and the previous tile's stores in the middle of a tile.
// The kernel retires everything itself with the one s_waitcnt vmcnt(0) at barrier (d).
__device__ __forceinline__ void i2s_glds16(const char* sbase, unsigned voff, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ floatx4 i2s_load16(const char* sbase, unsigned voff) {
  floatx4 r;
  asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
__device__ __forceinline__ half4_t i2s_load8(const char* sbase, unsigned voff) {
  half4_t r;
  asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(r) : "v"(voff), "s"(sbase) : "memory");
  return r;
}
// gfx940+ data hazard: a VMEM store of more than 64 bits followed by a VALU write of its data VGPRs needs 2 wait states
// (LLVM's hazard recognizer pads compiler-emitted stores; it does not look inside inline asm).  Without the s_nop the
// next instruction -- e.g. the address computation of the next ds_read into the same quad -- can replace the first dword
// of the stored 16 bytes (seen in csam_i2t_t2i: LDS addresses in every fourth column of the key rows).
__device__ __forceinline__ void i2s_store16(char* sbase, unsigned voff, half8_t v) {
  asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(sbase) : "memory");
}

#define I2S_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")

template <int QMODE, int NW, int MI_>
__global__ __launch_bounds__(64 * NW, 8 / NW) void i2t_stream_kernel(I2tArgs p, int n_tiles, int tiles_per_wg) {
  typedef I2S<NW, MI_> G;
  constexpr int MI = G::MI, HPW = G::HPW, NIO = G::NIO, TOK = G::TOK;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // scalar: LDS-DMA bases and weight slices are per wave
  const int fr = lane & 15, fg = lane >> 4;
  // byte offset of this thread's 16-B piece inside a [TOK][256] fp16 tile (source-side swizzle of the lane-linear LDS
  // image).  The NT threads cover NT/32 rows per piece; with 8 rows per piece (NW = 4) bit 3 of the row alternates
  // between pieces, which flips bit 3 of the swizzled slot: xoff ^ ((i & 1) << 7).
  const unsigned xoff = ((tid >> 5) * 256 + (((tid & 31) ^ ((tid >> 5) & 15)) * 8)) * 2;
  constexpr int PIECE = G::NT * 16;                   // bytes one piece (one instruction of every thread) covers
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;    // LDS byte address of the dynamic segment
  const int tpp = p.T / TOK;                          // tiles per prompt
  const int first = blockIdx.x * tiles_per_wg;
  const int last = min(first + tiles_per_wg, n_tiles);
  if (first >= last) return;

  // ---- launch-resident operands
  half8_t wq[HPW][8];
  if (QMODE == 1) {
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks)
        wq[hh][ks] = *(const half8_t*)(p.Wq + (long)((wave * HPW + hh) * 16 + fr) * 256 + ks * 32 + fg * 8);
  }
  half8_t wo[NIO][4];
#pragma unroll
  for (int ni = 0; ni < NIO; ++ni)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      wo[ni][ks] = *(const half8_t*)(p.Wo + (long)((wave * NIO + ni) * 16 + fr) * 128 + ks * 32 + fg * 8);
  float* par = (float*)(smem + G::PAR);
  for (int i = tid; i < 256; i += G::NT) {
    par[i] = p.bo[i];
    par[256 + i] = p.gamma[i];
    par[512 + i] = p.beta[i];
  }
  half4_t eye, ones;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    eye[r] = (half_t)(fg * 4 + r == fr ? 1.f : 0.f);
    ones[r] = (half_t)(fg * 4 + r < 7 ? 1.f : 0.f);
  }
  const float mb3 = fg == 1 ? -INFINITY : 0.f;          // key 7 does not exist (lane group 1, r = 3)

  auto issue_x = [&](int t, int buf) {
    const int b = t / tpp, t0 = (t - b * tpp) * TOK;
    const char* src = (const char*)(p.X + (long)b * p.x_bstride + (long)t0 * 256);
    const unsigned dst = lds0 + buf * G::BUF + wave * 1024;
#pragma unroll
    for (int i = 0; i < G::NP; ++i)                   // TOK rows x 32 slots, lane-linear LDS-DMA, source-side swizzle
      i2s_glds16(src + i * PIECE, NW == 4 ? xoff ^ ((i & 1) << 7) : xoff, dst + i * PIECE);
  };
  // per-tile register operands, fetched one tile ahead: the Q accumulator seed (QMODE 1: pe Wq^T + bq) or the hoisted
  // Q itself (QMODE 0), and the prompt's k / v^T fragments of this wave's heads
  floatx4 qa[HPW][MI];
  half4_t qh[HPW][MI];
  half4_t ka[HPW], va[HPW];
  auto fetch_q = [&](int t) {
    const int b = t / tpp, t0 = (t - b * tpp) * TOK;
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
      const unsigned el = fr * 128 + (wave * HPW + hh) * 16 + fg * 4;      // element offset inside a 16-token slab
      if (QMODE == 1) {
        const char* base = (const char*)(p.qpe + (long)t0 * 128);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) qa[hh][mi] = i2s_load16(base + mi * 16 * 128 * 4, el * 4);
      } else {
        const char* base = (const char*)(p.Q + (long)b * p.q_bstride + (long)t0 * 128);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) qh[hh][mi] = i2s_load8(base + mi * 16 * 128 * 2, el * 2);
      }
    }
  };
  auto fetch_kv = [&](int b) {
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
      const int hc = (wave * HPW + hh) * 16;
      ka[hh] = half4_t{0, 0, 0, 0};
      if (fr < 7) ka[hh] = *(const half4_t*)(p.kv_k + ((long)b * 7 + fr) * 128 + hc + fg * 4);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = fg * 4 + r;
        va[hh][r] = j < 7 ? p.kv_v[((long)b * 7 + j) * 128 + hc + fr] : (half_t)0.f;
      }
    }
    // compiler-tracked loads: a use right here makes the compiler place its (full) wait in this once-per-prompt
    // path and not at the first use inside every tile
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) asm volatile("" : "+v"(ka[hh]), "+v"(va[hh]));
  };

  issue_x(first, 0);
  fetch_q(first);
  fetch_kv(first / tpp);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

  for (int t = first; t < last; ++t) {
    const int cur = (t - first) & 1;
    char* xb = smem + cur * G::BUF;
    const int b = t / tpp, t0 = (t - b * tpp) * TOK;
    // (a) tile t has landed -- every thread retired its own pieces at barrier (d) of the previous tile, BEFORE that
    // tile's stores were issued, so the stores drain under this tile instead of being waited for here (gfx9 counts
    // loads and stores in the same vmcnt) -- and every thread is done reading the other buffer (its store source)
    I2S_BARRIER();
    if (t > first && t0 == 0) fetch_kv(b);            // new prompt (once per T/TOK tiles; the compiler drains here)
    if (t + 1 < last) issue_x(t + 1, cur ^ 1);

    // ---- Q^T = Wq_h X^T (+ seed), K = 256: A = resident weight fragments, B = tile fragments from LDS
    // (fragments of step ks+1 are read while the MFMAs of step ks run; the fence keeps the compiler from hoisting
    // the whole tile into registers)
    if (QMODE == 1) {
      half8_t xf[2][MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[0][mi] = *(const half8_t*)(xb + (mi * 16 + fr) * 512 + ((fg ^ fr) << 4));
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            xf[(ks + 1) & 1][mi] = *(const half8_t*)(xb + (mi * 16 + fr) * 512 + ((((ks + 1) * 4 + fg) ^ fr) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int hh = 0; hh < HPW; ++hh)
            qa[hh][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wq[hh][ks], xf[ks & 1][mi], qa[hh][mi], 0, 0, 0);
        asm volatile("" ::: "memory");
      }
    }
    // ---- 7-key attention of this wave's heads for the tile's tokens
    half4_t oh[HPW][MI];
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        half4_t qb;
        if (QMODE == 1) {
          const floatx4 qq = qa[hh][mi];
          qb = half4_t{(half_t)qq[0], (half_t)qq[1], (half_t)qq[2], (half_t)qq[3]};
        } else {
          qb = qh[hh][mi];
        }
        const floatx4 sacc = __builtin_amdgcn_mfma_f32_16x16x16f16(ka[hh], qb, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const float s3 = sacc[3] + mb3;
        float mx = fmaxf(fmaxf(sacc[0], sacc[1]), fmaxf(sacc[2], s3));
        mx = csam_max_x16(mx);
        const half4_t pb = {(half_t)csam_exp2(sacc[0] - mx), (half_t)csam_exp2(sacc[1] - mx),
                            (half_t)csam_exp2(sacc[2] - mx), (half_t)csam_exp2(s3 - mx)};
        const floatx4 o = __builtin_amdgcn_mfma_f32_16x16x16f16(va[hh], pb, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const floatx4 sm = __builtin_amdgcn_mfma_f32_16x16x16f16(ones, pb, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const float inv = __builtin_amdgcn_rcpf(sm[0]);
        oh[hh][mi] = half4_t{(half_t)(o[0] * inv), (half_t)(o[1] * inv), (half_t)(o[2] * inv), (half_t)(o[3] * inv)};
      }
    asm volatile("" ::: "memory");
    // ---- out-proj accumulators start from bias + residual (identity MFMA on the tile's own fragments)
    floatx4 acc[NIO][MI];
#pragma unroll
    for (int ni = 0; ni < NIO; ++ni) {
      const floatx4 bo_r = *(const floatx4*)(par + (wave * NIO + ni) * 16 + fg * 4);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = mi * 16 + fr;
        const int chunk = (wave * NIO + ni) * 2 + (fg >> 1);
        const half4_t r = *(const half4_t*)(xb + row * 512 + ((chunk ^ fr) << 4) + (fg & 1) * 8);
        acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x16f16(eye, r, bo_r, 0, 0, 0);
      }
    }
    I2S_BARRIER();                                      // (e) every wave is done reading the key tile
    // ---- O exchange: [TOK][128 dims] fp16 over the first half of the tile buffer
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const int row = mi * 16 + fr;
        *(half4_t*)(xb + row * 256 + ((((wave * HPW + hh) * 2 + (fg >> 1)) ^ fr) << 4) + (fg & 1) * 8) = oh[hh][mi];
      }
    I2S_BARRIER();                                      // (b)
    // ---- out-proj, K = 128 (O fragments double-buffered like the key fragments above)
    {
      half8_t of[2][MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) of[0][mi] = *(const half8_t*)(xb + (mi * 16 + fr) * 256 + ((fg ^ fr) << 4));
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks + 1 < 4) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            of[(ks + 1) & 1][mi] = *(const half8_t*)(xb + (mi * 16 + fr) * 256 + ((((ks + 1) * 4 + fg) ^ fr) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NIO; ++ni)
            acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wo[ni][ks], of[ks & 1][mi], acc[ni][mi], 0, 0, 0);
        asm volatile("" ::: "memory");
      }
    }
    // next tile's register operands (their registers are dead now; the loads fly under the LayerNorm)
    if (t + 1 < last) fetch_q(t + 1);
    // ---- LayerNorm partials of this wave's channels: in-lane, then over the 4 lane groups by a ones MFMA (fp32)
    float2_t* part = (float2_t*)(smem + G::PART);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      float2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
      for (int ni = 0; ni < NIO; ++ni) {
        const float2_t a = {acc[ni][mi][0], acc[ni][mi][1]}, c2 = {acc[ni][mi][2], acc[ni][mi][3]};
        s2 += a;
        q2 = __builtin_elementwise_fma(a, a, q2);
        s2 += c2;
        q2 = __builtin_elementwise_fma(c2, c2, q2);
      }
      const floatx4 ssum = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, s2[0] + s2[1], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const floatx4 qsum = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, q2[0] + q2[1], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      if (fg == 0) part[wave * TOK + mi * 16 + fr] = float2_t{ssum[0], qsum[0]};
    }
    I2S_BARRIER();                                      // (c) partials visible; O no longer read
    // ---- combine over the waves, normalise, write the fp16 tile back over its LDS image
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      const int row = mi * 16 + fr;
      float2_t tot = part[row];
#pragma unroll
      for (int w = 1; w < NW; ++w) tot += part[w * TOK + row];
      const float mean = tot[0] * (1.f / 256.f);
      const float var = fmaxf(tot[1] * (1.f / 256.f) - mean * mean, 0.f);
      const float rstd = rsqrtf(var + p.eps);
      const float2_t rs2 = {rstd, rstd}, nm2 = {-mean * rstd, -mean * rstd};
#pragma unroll
      for (int ni = 0; ni < NIO; ++ni) {
        const floatx4 ga_r = *(const floatx4*)(par + 256 + (wave * NIO + ni) * 16 + fg * 4);
        const floatx4 be_r = *(const floatx4*)(par + 512 + (wave * NIO + ni) * 16 + fg * 4);
        const float2_t g0 = {ga_r[0], ga_r[1]}, g1 = {ga_r[2], ga_r[3]};
        const float2_t b0 = {be_r[0], be_r[1]}, b1 = {be_r[2], be_r[3]};
        const float2_t v0 = {acc[ni][mi][0], acc[ni][mi][1]}, v1 = {acc[ni][mi][2], acc[ni][mi][3]};
        const float2_t y0 = __builtin_elementwise_fma(__builtin_elementwise_fma(v0, rs2, nm2), g0, b0);
        const float2_t y1 = __builtin_elementwise_fma(__builtin_elementwise_fma(v1, rs2, nm2), g1, b1);
        const int chunk = (wave * NIO + ni) * 2 + (fg >> 1);
        *(half4_t*)(xb + row * 512 + ((chunk ^ fr) << 4) + (fg & 1) * 8) =
            half4_t{(half_t)y0[0], (half_t)y0[1], (half_t)y1[0], (half_t)y1[1]};
      }
    }
    // (d) tile complete; also retires the next tile's LDS-DMA (a whole tile old) and register operands
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
      char* obase = (char*)(p.out + ((long)b * p.T + t0) * 256);
#pragma unroll
      for (int i = 0; i < G::NP; ++i) {
        const half8_t v = *(const half8_t*)(xb + tid * 16 + i * PIECE);
        i2s_store16(obase + i * PIECE, NW == 4 ? xoff ^ ((i & 1) << 7) : xoff, v);
      }
    }
  }
}

template <int QMODE, int NW, int MI_>
static void i2t_stream_launch(const I2tArgs& a, int B, int n_cu, hipStream_t stream) {
  typedef I2S<NW, MI_> G;
  static csam_once_t set;
  if (csam_first_call(set))
    hipFuncSetAttribute((const void*)i2t_stream_kernel<QMODE, NW, MI_>, hipFuncAttributeMaxDynamicSharedMemorySize, G::SMEM);
  const int n_tiles = B * (a.T / G::TOK);
  const int slots = n_cu * (8 / NW);                  // resident workgroups
  const int per = csam_cdiv(n_tiles, slots);
  dim3 grid(csam_cdiv(n_tiles, per));
  hipLaunchKernelGGL((i2t_stream_kernel<QMODE, NW, MI_>), grid, dim3(G::NT), G::SMEM, stream, a, n_tiles, per);
}

extern "C" int csam_i2t_stream(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16,
                               long q_prompt_stride, const void* Wq_f16, const float* qpe, const void* k_scaled_f16,
                               const void* v_f16, const void* Wo_f16, const float* bo, const float* gamma,
                               const float* beta, float eps, void* out_f16, int B, int T) {
  CSAM_REQUIRE(X_f16 && k_scaled_f16 && v_f16 && Wo_f16 && bo && gamma && beta && out_f16, "csam_i2t_stream: null pointer");
  CSAM_REQUIRE((Q_f16 != nullptr) != (Wq_f16 != nullptr), "csam_i2t_stream: give either Q (hoisted) or Wq");
  CSAM_REQUIRE(!Wq_f16 || qpe, "csam_i2t_stream: qpe required with Wq");
  CSAM_REQUIRE(B > 0 && T > 0 && T % 128 == 0, "csam_i2t_stream: T must be a multiple of 128");
  I2tArgs a;
  a.X = (const half_t*)X_f16; a.x_bstride = x_prompt_stride;
  a.Q = (const half_t*)Q_f16; a.q_bstride = q_prompt_stride;
  a.Wq = (const half_t*)Wq_f16; a.qpe = qpe;
  a.kv_k = (const half_t*)k_scaled_f16; a.kv_v = (const half_t*)v_f16;
  a.Wo = (const half_t*)Wo_f16; a.bo = bo; a.gamma = gamma; a.beta = beta; a.eps = eps;
  a.out = (half_t*)out_f16; a.T = T;
  const int n_cu = csam_cu_count();
  // 4 waves: 32-token tiles in the projected form (Wq + Wo = 128 VGPRs), 64-token tiles with the hoisted Q (the 8-wave and
  // 32-token hoisted variants measured slower: profiles/r02_*; they remain template instances of the tests only)
  if (Wq_f16) i2t_stream_launch<1, 4, 2>(a, B, n_cu, (hipStream_t)stream);
  else i2t_stream_launch<0, 4, 4>(a, B, n_cu, (hipStream_t)stream);
  CSAM_LAUNCH_CHECK("csam_i2t_stream");
  return CSAM_OK;
}

// =====================================================================================================
// csam_i2t_rank: the hoisted-Q (layer 0) image->token half-block in its rank-56 form.  With q (and the residual source)
// shared by all prompts, a prompt enters only through its 7 token keys / values:
//     out_proj(softmax(q K_b^T) V_b) = P_b M_b,   M_b[(h, j), :] = Wo[:, h] v_b[j, h]     (56 x 256, built per prompt)
// so the PV product, the O exchange between waves and half of the out-proj MFMA work disappear, and -- the point --
// every WAVE can own 16 tokens with ALL 256 channels: attention, out-proj, residual, LayerNorm and the store staging
// are wave-local, there is NO barrier inside a prompt, and the waves of a workgroup run free (the ablation of the
// 5-barrier stream kernel showed its phases simply add up).  M_b^T (32 KB) sits in LDS per prompt; scores of two heads
// come from ONE 16x16x32 MFMA (A = [K_h0 | 0 ; 0 | K_h1], B = 32 q dims), two such results pack the 8 k-slots of a
// P.M operand: slot (ks, fg, e) <-> head 4 ks + 2 (e >> 2) + (fg >> 1), key 4 (fg & 1) + (e & 3).
// =====================================================================================================
namespace {

constexpr int IR_M_BYTES = 256 * 64 * 2;            // M_b^T [256 channels][64 k-slots] fp16
constexpr int IR_KP_BYTES = 64 * 256 * 2;           // Kp_b [64 (head, key) rows][256 channels] fp16 (projected form)
constexpr int IR_SLICE = 16 * 512;                  // per-wave output staging, 16 tokens x 256 ch fp16
template <bool PROJ, int NW>
struct IR {
  static constexpr int KP = IR_M_BYTES;                                  // Kp region (PROJ only)
  static constexpr int SLICES = IR_M_BYTES + (PROJ ? IR_KP_BYTES : 0);
  static constexpr int PAR = SLICES + NW * IR_SLICE;                     // bo | gamma | beta fp32
  static constexpr int SMEM = PAR + 3 * 256 * 4;
};

struct IrArgs {
  const half_t* X; long x_bstride; const half_t* Q; long q_bstride;
  const half_t* ks;          // [B,7,128] token-side k, pre-multiplied by 0.25 log2(e)
  const half_t* M;           // [B][256][64] from i2t_rank_prep_kernel
  const half_t* Kp;          // [B][64][256] from i2t_rank_kp_kernel (projected form) or null
  const float* bo; const float* gamma; const float* beta; float eps;
  half_t* out; int B; int T;
};

// M_b^T[c][slot] = sum_d Wo[c][head*16 + d] v[b][j][head*16 + d]  (j = 7: zero), slot order as in the header
// bo_fold (csam_i2t_t2i_fold bit 0): the out-projection bias rides in M_b -- each valid slot carries bo[c] / 8, and the softmax
// weights of a head's 7 keys sum to 1, so the 8 heads together add exactly bo[c] (up to the fp16 rounding of the M entries).
__global__ __launch_bounds__(256) void i2t_rank_prep_kernel(const half_t* __restrict__ v, const half_t* __restrict__ Wo,
                                                            half_t* __restrict__ M, const float* __restrict__ bo_fold = nullptr) {
  __shared__ float vs[7 * 128];
  const int b = blockIdx.x, c = threadIdx.x;
  for (int i = c; i < 7 * 128; i += 256) vs[i] = (float)v[(long)b * 7 * 128 + i];
  __syncthreads();
  float w[128];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const half8_t h = *(const half8_t*)(Wo + (long)c * 128 + i * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) w[i * 8 + e] = (float)h[e];
  }
  half_t* dst = M + ((long)b * 256 + c) * 64;
#pragma unroll
  for (int s8 = 0; s8 < 8; ++s8) {              // s8 = ks * 4 + fg
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ks = s8 >> 2, fg = s8 & 3;
      const int head = 4 * ks + 2 * (e >> 2) + (fg >> 1), j = 4 * (fg & 1) + (e & 3);
      float acc = 0.f;
      if (j < 7) {
#pragma unroll
        for (int d = 0; d < 16; ++d) acc = fmaf(w[head * 16 + d], vs[j * 128 + head * 16 + d], acc);
        if (bo_fold) acc += 0.125f * bo_fold[c];
      }
      o[e] = (half_t)acc;
    }
    *(half8_t*)(dst + s8 * 8) = o;
  }
}

// Projected form (layer 1: q = (X + pe) Wq^T + bq differs per prompt): the X-dependent part of the scores is
//     (X Wq_h^T) . k_b[j,h] = X . Kp_b[(h, j), :],    Kp_b[8 h + j][c] = sum_d ks[b][j][16 h + d] Wq[16 h + d][c]   (j = 7: zero)
// so the q projection of the 4096 image tokens is never formed: 56 back-projected token keys per prompt instead.
__global__ __launch_bounds__(256) void i2t_rank_kp_kernel(const half_t* __restrict__ ks, const half_t* __restrict__ Wq,
                                                          half_t* __restrict__ Kp, int B) {
  __shared__ float kk[7 * 128];
  const int c = threadIdx.x;
  float w[128];                                   // column c of Wq, kept over the workgroup's prompts
#pragma unroll
  for (int r = 0; r < 128; ++r) w[r] = (float)Wq[(long)r * 256 + c];
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int i = c; i < 7 * 128; i += 256) kk[i] = (float)ks[(long)b * 7 * 128 + i];
    __syncthreads();
    half_t* dst = Kp + (long)b * 64 * 256 + c;
#pragma unroll
    for (int head = 0; head < 8; ++head) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float acc = 0.f;
        if (j < 7) {
#pragma unroll
          for (int d = 0; d < 16; ++d) acc = fmaf(w[head * 16 + d], kk[j * 128 + head * 16 + d], acc);
        }
        dst[(long)(head * 8 + j) * 256] = (half_t)acc;
      }
    }
  }
}

template <bool PROJ, int NW>
__global__ __launch_bounds__(64 * NW, 8 / NW) void i2t_rank_kernel(IrArgs p, int prompts_per_wg) {
  typedef IR<PROJ, NW> G;
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;
  char* slice = smem + G::SLICES + wave * IR_SLICE;
  float* par = (float*)(smem + G::PAR);
  for (int i = tid; i < 256; i += NT) {
    par[i] = p.bo[i];
    par[256 + i] = p.gamma[i];
    par[512 + i] = p.beta[i];
  }
  const int tpp = p.T / 16;
  const int b_first = blockIdx.x * prompts_per_wg;
  const int b_last = min(b_first + prompts_per_wg, p.B);
  // residual: acc[2 nj + h] += E_h . X^T with E_0 = [I16 | 0], E_1 = [0 | I16] over a 32-channel (16 B per lane) fragment
  half8_t eye_lo, eye_hi;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    eye_lo[e] = (half_t)((fg < 2 && fg * 8 + e == fr) ? 1.f : 0.f);
    eye_hi[e] = (half_t)((fg >= 2 && (fg - 2) * 8 + e == fr) ? 1.f : 0.f);
  }
  const bool key7 = (fg & 1) == 1;                   // this lane's 4th key is the non-existent 8th of its head

  floatx4 qraw[4];                                   // 4 x half8: 32 q dims of head pair pr for token fr
  floatx4 xres[8];                                   // 8 x half8 residual fragments: channels nj*32 + fg*8 .. +7 of token fr
  for (int b = b_first; b < b_last; ++b) {
    __syncthreads();                                 // every wave is done with the previous prompt's M (and Kp)
    {
      const char* src = (const char*)(p.M + (long)b * 256 * 64);
#pragma unroll
      for (int i = 0; i < 2048 / NT; ++i) {          // 256 rows x 8 slots of 16 B, slot ^= row & 7
        const int c = tid + i * NT, row = c >> 3, sl = c & 7;
        i2s_glds16(src, (unsigned)(row * 128 + ((sl ^ (row & 7)) << 4)), lds0 + (unsigned)(wave * 1024 + i * NT * 16));
      }
      if constexpr (PROJ) {
        const char* ksrc = (const char*)(p.Kp + (long)b * 64 * 256);
#pragma unroll
        for (int i = 0; i < 2048 / NT; ++i) {        // 64 rows x 32 slots of 16 B, slot ^= row & 15
          const int c = tid + i * NT, row = c >> 5, sl = c & 31;
          i2s_glds16(ksrc, (unsigned)(row * 512 + ((sl ^ (row & 15)) << 4)),
                     lds0 + (unsigned)(G::KP + wave * 1024 + i * NT * 16));
        }
      }
    }
    // paired-head key fragments: rows 0-7 = K_h0 (dims in k 0..15), rows 8-15 = K_h1 (dims in k 16..31)
    half8_t kfr[4];
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
      half8_t kv = {0, 0, 0, 0, 0, 0, 0, 0};
      const int j = fr & 7, hsel = fr >> 3;          // row -> key j of head 2 pr + hsel
      if (j < 7 && (fg >> 1) == hsel)
        kv = *(const half8_t*)(p.ks + ((long)b * 7 + j) * 128 + (2 * pr + hsel) * 16 + (fg & 1) * 8);
      kfr[pr] = kv;
      asm volatile("" : "+v"(kfr[pr]));              // tracked loads: the compiler's wait lands here, once per prompt
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto prefetch = [&](int tile) {
      const int t0 = tile * 16;
      const char* qb = (const char*)(p.Q + (long)b * p.q_bstride + (long)t0 * 128);
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) qraw[pr] = i2s_load16(qb + pr * 64, (unsigned)(fr * 256 + fg * 16));
      const char* xb = (const char*)(p.X + (long)b * p.x_bstride + (long)t0 * 256);
#pragma unroll
      for (int nj = 0; nj < 8; ++nj) xres[nj] = i2s_load16(xb + nj * 64, (unsigned)(fr * 512 + fg * 16));
    };
    prefetch(wave);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    for (int tile = wave; tile < tpp; tile += NW) {
      const int t0 = tile * 16;
      // ---- scores of the 8 heads (two per MFMA) and their softmax over the 7 keys
      floatx4 sc[4];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        half8_t qf;
        __builtin_memcpy(&qf, &qraw[pr], 16);
        sc[pr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr[pr], qf, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      }
      if constexpr (PROJ) {
        // + X . Kp^T: A = Kp rows 16 pr + fr (head pair pr, same row order as kfr), channels nj*32 + fg*8 .. +7
#pragma unroll
        for (int nj = 0; nj < 8; ++nj) {
          half8_t xf;
          __builtin_memcpy(&xf, &xres[nj], 16);
#pragma unroll
          for (int pr = 0; pr < 4; ++pr) {
            const int row = pr * 16 + fr;
            const half8_t kp = *(const half8_t*)(smem + G::KP + row * 512 + (((nj * 4 + fg) ^ fr) << 4));
            sc[pr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kp, xf, sc[pr], 0, 0, 0);
          }
        }
      }
      half8_t pf[2];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const floatx4 s4 = sc[pr];
        const float s3 = key7 ? -INFINITY : s4[3];
        float mx = fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s3));
        mx = csam_max_x16(mx);
        const float p0 = csam_exp2(s4[0] - mx), p1 = csam_exp2(s4[1] - mx);
        const float p2 = csam_exp2(s4[2] - mx), p3 = csam_exp2(s3 - mx);
        float sum = (p0 + p1) + (p2 + p3);
        sum = csam_sum_x16(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        const int o = (pr & 1) * 4;
        pf[pr >> 1][o] = (half_t)(p0 * inv);
        pf[pr >> 1][o + 1] = (half_t)(p1 * inv);
        pf[pr >> 1][o + 2] = (half_t)(p2 * inv);
        pf[pr >> 1][o + 3] = (half_t)(p3 * inv);
      }
      // ---- accumulators = out-proj bias + residual (identity MFMA on the prefetched fragments)
      floatx4 acc[16];
#pragma unroll
      for (int nj = 0; nj < 8; ++nj) {
        half8_t xf;
        __builtin_memcpy(&xf, &xres[nj], 16);
        const floatx4 b0 = *(const floatx4*)(par + nj * 32 + fg * 4);
        const floatx4 b1 = *(const floatx4*)(par + nj * 32 + 16 + fg * 4);
        acc[2 * nj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eye_lo, xf, b0, 0, 0, 0);
        acc[2 * nj + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eye_hi, xf, b1, 0, 0, 0);
      }
      asm volatile("" ::: "memory");
      if (tile + NW < tpp) prefetch(tile + NW);        // the q / residual registers are free: next tile's operands
      // ---- P . M_b : [256 channels] x [16 tokens], K = 64 slots
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int ni = 0; ni < 16; ++ni) {
          if ((ni & 3) == 0) asm volatile("" ::: "memory");
          const int row = ni * 16 + fr;
          const half8_t mf = *(const half8_t*)(smem + row * 128 + (((ks * 4 + fg) ^ (row & 7)) << 4));
          acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(mf, pf[ks], acc[ni], 0, 0, 0);
        }
      }
      // ---- LayerNorm over the 256 channels of token fr: in-lane 64 values, lane groups by an fp32 ones-MFMA
      float2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
      for (int ni = 0; ni < 16; ++ni) {
        const float2_t a = {acc[ni][0], acc[ni][1]}, c2 = {acc[ni][2], acc[ni][3]};
        s2 += a;
        q2 = __builtin_elementwise_fma(a, a, q2);
        s2 += c2;
        q2 = __builtin_elementwise_fma(c2, c2, q2);
      }
      const floatx4 ssum = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, s2[0] + s2[1], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const floatx4 qsum = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, q2[0] + q2[1], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const float mean = ssum[0] * (1.f / 256.f);
      const float var = fmaxf(qsum[0] * (1.f / 256.f) - mean * mean, 0.f);
      const float rstd = rsqrtf(var + p.eps);
      const float2_t rs2 = {rstd, rstd}, nm2 = {-mean * rstd, -mean * rstd};
#pragma unroll
      for (int ni = 0; ni < 16; ++ni) {
        if ((ni & 3) == 0) asm volatile("" ::: "memory");
        const floatx4 g = *(const floatx4*)(par + 256 + ni * 16 + fg * 4);
        const floatx4 be = *(const floatx4*)(par + 512 + ni * 16 + fg * 4);
        const float2_t g0 = {g[0], g[1]}, g1 = {g[2], g[3]}, b0 = {be[0], be[1]}, b1 = {be[2], be[3]};
        const float2_t v0 = {acc[ni][0], acc[ni][1]}, v1 = {acc[ni][2], acc[ni][3]};
        const float2_t y0 = __builtin_elementwise_fma(__builtin_elementwise_fma(v0, rs2, nm2), g0, b0);
        const float2_t y1 = __builtin_elementwise_fma(__builtin_elementwise_fma(v1, rs2, nm2), g1, b1);
        const int chunk = ni * 2 + (fg >> 1);
        *(half4_t*)(slice + fr * 512 + ((chunk ^ fr) << 4) + (fg & 1) * 8) =
            half4_t{(half_t)y0[0], (half_t)y0[1], (half_t)y1[0], (half_t)y1[1]};
      }
      // next tile's operands have had the whole tile to land; retire them BEFORE this tile's stores are issued
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      char* obase = (char*)(p.out + ((long)b * p.T + t0) * 256);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int idx = lane + i * 64, row = idx >> 5, sl = idx & 31;
        const half8_t v = *(const half8_t*)(slice + idx * 16);
        i2s_store16(obase, (unsigned)(row * 512 + ((sl ^ (row & 15)) << 4)), v);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the slice is rewritten by the next tile
    }
  }
}

}  // namespace

extern "C" long csam_i2t_rank_workspace_bytes(int B) { return (long)B * 256 * 64 * 2; }
extern "C" long csam_i2t_rank_proj_workspace_bytes(int B) { return (long)B * (IR_M_BYTES + IR_KP_BYTES); }

static int ir_cus() {
  static csam_once_t once;
  const int n_cu = csam_cu_count();
  if (csam_first_call(once)) {
    constexpr int s0 = IR<false, 4>::SMEM, s1 = IR<true, 8>::SMEM, s2 = IR<false, 8>::SMEM;
    (void)hipFuncSetAttribute((const void*)i2t_rank_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, s0);
    (void)hipFuncSetAttribute((const void*)i2t_rank_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, s2);
    (void)hipFuncSetAttribute((const void*)i2t_rank_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, s1);
  }
  return n_cu;
}

extern "C" int csam_i2t_rank(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16, long q_prompt_stride,
                             const void* k_scaled_f16, const void* v_f16, const void* Wo_f16, const float* bo,
                             const float* gamma, const float* beta, float eps, void* out_f16, int B, int T,
                             void* workspace, long workspace_bytes) {
  CSAM_REQUIRE(X_f16 && Q_f16 && k_scaled_f16 && v_f16 && Wo_f16 && bo && gamma && beta && out_f16 && workspace,
               "csam_i2t_rank: null pointer");
  CSAM_REQUIRE(B > 0 && T > 0 && T % 64 == 0, "csam_i2t_rank: T must be a multiple of 64");
  if (workspace_bytes < csam_i2t_rank_workspace_bytes(B)) {
    csam_set_error("csam_i2t_rank: workspace too small");
    return CSAM_ERR_WORKSPACE;
  }
  const int n_cu = ir_cus();
  hipLaunchKernelGGL(i2t_rank_prep_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const half_t*)v_f16,
                     (const half_t*)Wo_f16, (half_t*)workspace);
  IrArgs a;
  a.X = (const half_t*)X_f16; a.x_bstride = x_prompt_stride; a.Q = (const half_t*)Q_f16; a.q_bstride = q_prompt_stride;
  a.ks = (const half_t*)k_scaled_f16; a.M = (const half_t*)workspace; a.Kp = nullptr; a.bo = bo; a.gamma = gamma;
  a.beta = beta; a.eps = eps; a.out = (half_t*)out_f16; a.B = B; a.T = T;
  {
    const int per = csam_cdiv(B, 2 * n_cu);
    constexpr int smem = IR<false, 4>::SMEM;
    hipLaunchKernelGGL((i2t_rank_kernel<false, 4>), dim3(csam_cdiv(B, per)), dim3(256), smem, (hipStream_t)stream, a, per);
  }
  CSAM_LAUNCH_CHECK("csam_i2t_rank");
  return CSAM_OK;
}

// Layer-1 form: the keys differ per prompt (x_prompt_stride = T * 256) and q = (X + pe) Wq^T + bq is never formed:
// qpe_f16 [T,128] = pe Wq^T + bq is the shared part (plays the hoisted Q of csam_i2t_rank), X . Kp_b^T the per-prompt part.
extern "C" int csam_i2t_rank_proj(void* stream, const void* X_f16, long x_prompt_stride, const void* qpe_f16,
                                  const void* Wq_f16, const void* k_scaled_f16, const void* v_f16, const void* Wo_f16,
                                  const float* bo, const float* gamma, const float* beta, float eps, void* out_f16, int B,
                                  int T, void* workspace, long workspace_bytes) {
  CSAM_REQUIRE(X_f16 && qpe_f16 && Wq_f16 && k_scaled_f16 && v_f16 && Wo_f16 && bo && gamma && beta && out_f16 &&
                   workspace,
               "csam_i2t_rank_proj: null pointer");
  CSAM_REQUIRE(B > 0 && T > 0 && T % 128 == 0, "csam_i2t_rank_proj: T must be a multiple of 128");
  if (workspace_bytes < csam_i2t_rank_proj_workspace_bytes(B)) {
    csam_set_error("csam_i2t_rank_proj: workspace too small");
    return CSAM_ERR_WORKSPACE;
  }
  const int n_cu = ir_cus();
  half_t* Mws = (half_t*)workspace;
  half_t* Kpws = Mws + (long)B * 256 * 64;
  const dim3 pgrid(B < 2 * n_cu ? B : 2 * n_cu);
  hipLaunchKernelGGL(i2t_rank_prep_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const half_t*)v_f16,
                     (const half_t*)Wo_f16, Mws);
  hipLaunchKernelGGL(i2t_rank_kp_kernel, pgrid, dim3(256), 0, (hipStream_t)stream, (const half_t*)k_scaled_f16,
                     (const half_t*)Wq_f16, Kpws, B);
  IrArgs a;
  a.X = (const half_t*)X_f16; a.x_bstride = x_prompt_stride; a.Q = (const half_t*)qpe_f16; a.q_bstride = 0;
  a.ks = (const half_t*)k_scaled_f16; a.M = Mws; a.Kp = Kpws; a.bo = bo; a.gamma = gamma; a.beta = beta; a.eps = eps;
  a.out = (half_t*)out_f16; a.B = B; a.T = T;
  const int per = csam_cdiv(B, n_cu);               // whole prompts per workgroup, one 8-wave workgroup per CU
  constexpr int smem = IR<true, 8>::SMEM;
  hipLaunchKernelGGL((i2t_rank_kernel<true, 8>), dim3(csam_cdiv(B, per)), dim3(512), smem, (hipStream_t)stream, a, per);
  CSAM_LAUNCH_CHECK("csam_i2t_rank_proj");
  return CSAM_OK;
}

// =====================================================================================================
// csam_upscale_fused: mask_decoder.py:172-181 in ONE kernel, one pass over the final key state:
//   up1 = ConvT(256->64,k2,s2)(keys)  ->  LayerNorm2d(64) -> GELU  ->  ConvT(64->32,k2,s2) -> GELU
//   masks[b,l,Y,X] = sum_c hyper[b,l,c] * up2[b,c,Y,X]
// Both transposed convolutions are per-token GEMMs (k=2,s=2: no overlap), so a token's 4x4 output
// pixels depend on that token only.  Workgroup = 64 consecutive tokens (one row of the 64x64 grid) of
// one prompt, 4 waves: wave = first-conv position (di,dj).
//   GEMM1 (K=256, 3-deep glds ring, swizzled)  -> the wave's 64 channels x 64 tokens in registers
//   -> LN over the 64 channels (in-lane + 2 shuffles) + GELU -> fp16 registers ARE GEMM2's B operand
//   GEMM2 (K=64, N=128 = 4 sub-positions x 32 ch, weights permuted on the host) -> GELU
//   -> hyper product as a 16x16x32 MFMA (hyper rows hi+lo fp16, rows 4..15 zero), again fed from
//      the accumulator registers -> 4 mask logits per (token, position, sub-position)
//   -> staged in LDS as full 256-pixel output rows -> coalesced fp32 stores.
// HBM per prompt: read 2 MB keys + write 1 MB logits (unfused chain: 19 MB).
// =====================================================================================================
namespace {

// LDS map (78.5 KB -> two workgroups per CU): operand ring 3 x (X 64x64 B + W1 256x64 B) | W2' | hyper frags | b2
constexpr int UP_NS = 3;                              // ring depth (K step = 32)
constexpr int UP_XB = 64 * 64, UP_WB = 256 * 64;      // bytes per stage
constexpr int UP_STAGE = UP_XB + UP_WB;               // 20 KB
constexpr int UP_W2S = UP_NS * UP_STAGE;              // 60 KB
constexpr int UP_HFR = UP_W2S + 16 * 1024;
constexpr int UP_B2 = UP_HFR + 2 * 64 * 16;           // b2 fp32 [128]
constexpr int UP_SMEM = UP_B2 + 128 * 4;

struct UpArgs {
  const half_t* X;          // keys [B*4096, 256]
  const half_t* W1;         // [256 = pos*64+co, 256 ci]
  const float* b1;          // [256]
  const float* ln_g; const float* ln_b; float eps;   // [64]
  const half_t* W2;         // [128 = pos2*32+co2, 64 ci2 (k-permuted)]
  const float* b2;          // [128]
  const float* hyper;       // [B,4,32]
  float* masks;             // [B,4,256,256]
  float* stats;             // optional [B*4][2]: running max of every mask plane (float atomic max), else NULL
};

// order-preserving float atomic max (sign-split integer trick); *addr must start at -inf
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
  if (v >= 0.f) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

// Workgroup = the 64 tokens of one row of the 64x64 grid, 4 waves: wave = first-conv position (di,dj), each
// wave owns 64 channels x 64 tokens.  Two workgroups are co-resident per CU (LDS 78.5 KB, 2 waves/SIMD), so
// one's GEMM1 load latency and store tail hide behind the other's LN/GELU/GEMM2 phases (the single 8-wave
// workgroup per CU this replaces idled there: measured 3.35 -> ms per 1024 prompts).
__global__ __launch_bounds__(256, 2) void upscale_fused_kernel(UpArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * 64;                  // first token of the grid row
  const int pos = wave;
  const half_t* Xb = p.X + ((long)b * 4096 + t0) * 256;

  // ---- one-time staging: W2' (16 KB, 128-B rows, swizzled) and the hyper A-fragments (hi/lo)
  {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int cc = tid + it * 256;                // 1024 16-B pieces
      const int row = cc >> 3, sl = cc & 7;
      glds16(p.W2 + (long)row * 64 + ((sl ^ (row & 7)) * 8), smem + UP_W2S + (cc & ~63) * 16);
    }
    if (tid < 64) {
      half8_t hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
      const int i = tid & 15, g = tid >> 4;
      if (i < 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int cch = ((e >= 4) ? 16 : 0) + g * 4 + (e & 3);
          const float h = p.hyper[((long)b * 4 + i) * 32 + cch];
          hi[e] = (half_t)h;
          lo[e] = (half_t)(h - (float)hi[e]);
        }
      }
      *(half8_t*)(smem + UP_HFR + tid * 16) = hi;
      *(half8_t*)(smem + UP_HFR + 1024 + tid * 16) = lo;
    }
    if (tid >= 64 && tid < 192) ((float*)(smem + UP_B2))[tid - 64] = p.b2[tid - 64];
  }

  // ---- phase 1: GEMM1, K = 256 in 8 steps of 32 through a 3-deep ring with counted vmcnt (64-B LDS rows:
  // slot ^= 3 * ((row >> 2) & 1) on the global source and on the fragment reads, conflict-free for ds_read_b128)
  const half_t* x_src;
  const half_t* w_src[4];
  {
    const int row = tid >> 2, sl = tid & 3;          // X: 64 rows x 4 slots = 256 pieces, 1 per thread
    x_src = Xb + (long)row * 256 + ((sl ^ (3 * ((row >> 2) & 1))) * 8);
#pragma unroll
    for (int it = 0; it < 4; ++it) {                 // W1: 256 rows x 4 slots = 1024 pieces, 4 per thread
      const int cc = tid + it * 256;
      const int wrow = cc >> 2, wsl = cc & 3;
      w_src[it] = p.W1 + (long)wrow * 256 + ((wsl ^ (3 * ((wrow >> 2) & 1))) * 8);
    }
  }
  auto stage = [&](int buf, int k0) {
    char* xb = smem + buf * UP_STAGE;
    glds16(x_src + k0, xb + (tid & ~63) * 16);
#pragma unroll
    for (int it = 0; it < 4; ++it) glds16(w_src[it] + k0, xb + UP_XB + ((tid + it * 256) & ~63) * 16);
  };
  floatx4 a1[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) a1[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int coff = (fg ^ (3 * ((fr >> 2) & 1))) << 4;
  stage(0, 0);
  stage(1, 32);
  {
    int cur = 0;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // retires stage kt+1 too; it is read one iteration later (see t2i)
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 2 < 8) stage(cur == 0 ? 2 : cur - 1, (kt + 2) * 32);
      const char* xb = smem + cur * UP_STAGE;
      const char* wb = xb + UP_XB;
      half8_t xf[4], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        xf[i] = *(const half8_t*)(xb + (i * 16 + fr) * 64 + coff);
        wf[i] = *(const half8_t*)(wb + (pos * 64 + i * 16 + fr) * 64 + coff);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          a1[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], xf[mi], a1[mi][ni], 0, 0, 0);
      cur = cur == 2 ? 0 : cur + 1;
    }
  }
  __syncthreads();                                   // ring is free (the output rows alias it)

  // ---- phases 2-4 per 16-token tile mi
  floatx4 b1v[4], gv[4], bv[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    b1v[ni] = *(const floatx4*)(p.b1 + pos * 64 + ni * 16 + fg * 4);
    gv[ni] = *(const floatx4*)(p.ln_g + ni * 16 + fg * 4);
    bv[ni] = *(const floatx4*)(p.ln_b + ni * 16 + fg * 4);
  }
  const half8_t hhi = *(const half8_t*)(smem + UP_HFR + lane * 16);
  const half8_t hlo = *(const half8_t*)(smem + UP_HFR + 1024 + lane * 16);
  float* outs = (float*)(smem);                     // [4 l][4 yy][256 X] fp32 = 16 KB (aliases the ring)
  // Per 16-token tile mi: LayerNorm2d(64) + GELU -> GEMM2 (MFMA) -> +bias, GELU -> hyper product (MFMA) -> LDS rows.
  // Software-pipelined by one tile: GEMM2 of tile mi+1 is issued before the GELU/hyper stage of tile mi, so the
  // matrix pipe works under the (dominant) VALU stream instead of in front of it.
  auto ln_gelu = [&](int mi, half8_t (&xf2)[2]) {
    float sum = 0.f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      a1[mi][ni] += b1v[ni];
      sum += (a1[mi][ni][0] + a1[mi][ni][1]) + (a1[mi][ni][2] + a1[mi][ni][3]);
    }
    sum = csam_sum_x16(sum);
    sum = csam_sum_x32(sum);
    const float mean = sum * (1.f / 64.f);
    float var = 0.f;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = a1[mi][ni][e] - mean;
        var += d * d;
      }
    var = csam_sum_x16(var);
    var = csam_sum_x32(var);
    const float rstd = 1.0f / sqrtf(var * (1.f / 64.f) + p.eps);
    const float nmr = -mean * rstd;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int e = 0; e < 4; e += 2) {
        // y = (a - mean) * rstd * g + b, two channels per packed op
        const float2_t av = {a1[mi][ni][e], a1[mi][ni][e + 1]};
        const float2_t g2 = {gv[ni][e], gv[ni][e + 1]}, b2 = {bv[ni][e], bv[ni][e + 1]};
        const float2_t nrm = __builtin_elementwise_fma(av, (float2_t){rstd, rstd}, (float2_t){nmr, nmr});
        const float2_t ge = csam_gelu_poly2(__builtin_elementwise_fma(nrm, g2, b2));
        xf2[ni >> 1][(ni & 1) * 4 + e] = (half_t)ge[0];
        xf2[ni >> 1][(ni & 1) * 4 + e + 1] = (half_t)ge[1];
      }
  };
  // GEMM2: [128 n2] x [16 tokens], K = 64 (weights k-permuted): 8 N tiles x 2 k-steps
  auto gemm2 = [&](const half8_t (&xf2)[2], floatx4 (&a2)[8]) {
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2) a2[n2] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < 2; ++s) {          // k-step outer: 8 independent accumulators back to back
#pragma unroll
      for (int n2 = 0; n2 < 8; ++n2) {
        const int row = n2 * 16 + fr;
        const half8_t wf = *(const half8_t*)(smem + UP_W2S + row * 128 + (((s * 4 + fg) ^ (row & 7)) << 4));
        a2[n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf2[s], a2[n2], 0, 0, 0);
      }
    }
  };
  // + bias, GELU, hyper product per sub-position pos2 (N tiles 2*pos2, 2*pos2+1)
  auto gelu_hyper = [&](int mi, const floatx4 (&a2)[8]) {
#pragma unroll
    for (int pos2 = 0; pos2 < 4; ++pos2) {
      half8_t ub;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const floatx4 bb = *(const floatx4*)(smem + UP_B2 + ((pos2 * 2 + h2) * 16 + fg * 4) * 4);
#pragma unroll
        for (int e = 0; e < 4; e += 2) {
          const float2_t z = (float2_t){a2[pos2 * 2 + h2][e], a2[pos2 * 2 + h2][e + 1]} + (float2_t){bb[e], bb[e + 1]};
          const float2_t ge = csam_gelu_poly2(z);
          ub[h2 * 4 + e] = (half_t)ge[0];
          ub[h2 * 4 + e + 1] = (half_t)ge[1];
        }
      }
      floatx4 m4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hhi, ub, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      m4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hlo, ub, m4, 0, 0, 0);
      if (fg == 0) {   // rows 0..3 of the product = the 4 mask logits of this pixel
        const int yy = (pos >> 1) * 2 + (pos2 >> 1);
        const int X = 4 * (mi * 16 + fr) + 2 * (pos & 1) + (pos2 & 1);
#pragma unroll
        for (int l = 0; l < 4; ++l) outs[(l * 4 + yy) * 256 + X] = m4[l];
      }
    }
  };
  {
    half8_t xfa[2], xfb[2];
    floatx4 a2a[8], a2b[8];
    ln_gelu(0, xfa);
    gemm2(xfa, a2a);
    ln_gelu(1, xfb);
    gemm2(xfb, a2b);
    gelu_hyper(0, a2a);
    ln_gelu(2, xfa);
    gemm2(xfa, a2a);
    gelu_hyper(1, a2b);
    ln_gelu(3, xfb);
    gemm2(xfb, a2b);
    gelu_hyper(2, a2a);
    gelu_hyper(3, a2b);
  }
  float* wmx = (float*)(smem + UP_W2S);   // [4 waves][4] (W2' is dead after the last GEMM2 of every wave: barrier below)
  __syncthreads();
  // ---- coalesced store of the 4 x 4 output rows (256 fp32 each); iteration `it` of a thread is plane l = it,
  // row yy = wave, so the per-plane running max for the PWD-Net softmax (saves a full pass over the logits
  // later) falls out of the same LDS reads
  {
    const int i0 = blockIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int x4 = tid & 63, rowid = it * 4 + wave;   // rowid = l*4 + yy
      const floatx4 v = *(const floatx4*)(outs + rowid * 256 + x4 * 4);
      float* dst = p.masks + ((((long)b * 4 + it) * 256) + (4 * i0 + wave)) * 256 + x4 * 4;
      *(floatx4*)dst = v;
      if (p.stats) {
        const float m = csam_wave_max(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
        if (lane == 0) wmx[wave * 4 + it] = m;
      }
    }
  }
  if (p.stats) {
    __syncthreads();
    if (tid < 4) {
      const float m = fmaxf(fmaxf(wmx[tid], wmx[4 + tid]), fmaxf(wmx[8 + tid], wmx[12 + tid]));
      atomic_max_float(p.stats + ((long)b * 4 + tid) * 2, m);   // one atomic per (workgroup, plane)
    }
  }
}

}  // namespace

// =====================================================================================================
// csam_upscale_stream: csam_upscale_fused as a persistent, weight-stationary stream.  PMC/ablation of the tile-per-
// workgroup kernel: GEMM1 alone 2.5 ms per 2048 prompts because every 64-token workgroup re-stages the 128 KB of W1
// from L2 through LDS (21 GB of L2->LDS traffic per launch, ~8.6 TB/s: L2-bound), phases 2-4 alone 3.3 ms (VALU).
// Here a 4-wave workgroup (two per CU) walks WHOLE prompts in 32-token tiles (half a row of the 64x64 grid):
//   * wave = first-conv position; its W1 slice (64 rows x 256) stays in 128 VGPRs for the launch;
//   * key tiles (16 KB) are LDS-DMA'd one tile ahead into a double buffer, every wave reads all of it as B fragments;
//   * LN2d + GELU -> GEMM2 (W2' resident in LDS) -> +bias, GELU -> hyper product exactly as in upscale_fused;
//   * the 16 output half-rows (4 masks x 4 rows x 128 px fp32) leave through LDS as whole 512-B segments, the stores
//     draining under the next tile; the per-plane max for the PWD-Net softmax is carried in registers across the
//     prompt and written once (no atomics, no init kernel dependence).
// =====================================================================================================
namespace {

// GELU of 4 channel pairs at once: four interleaved packed Horner chains (csam_common.h).  Scalar chains, wave priorities by
// phase and a rank-112 first conv at three waves per SIMD were measured in round 4 and are gone: profiles/r04_valu_rate.txt,
// r04_upscale_prio.txt, r04_upscale_rank_probe.txt
#define CSAM_UP_XDEPTH 3
#ifndef CSAM_UP_PIN
#define CSAM_UP_PIN 1
#endif
#ifndef CSAM_UP_ONE_STORE_BLOCK
#define CSAM_UP_ONE_STORE_BLOCK 1
#endif
constexpr int US_TOK = 32;
constexpr int US_BUF = US_TOK * 512;               // 16 KB key tile
constexpr int US_W2S = 2 * US_BUF;                 // W2' 16 KB
constexpr int US_OUT = US_W2S + 16 * 1024;         // [16 rows][128 px] fp32 = 8 KB
constexpr int US_HFR = US_OUT + 8 * 1024;          // hyper fragments hi | lo, 2 KB
constexpr int US_PAR = US_HFR + 2048;              // b2 [128] | b1 [256] | ln_g [64] | ln_b [64] fp32 = 2 KB
constexpr int US_WMX = US_PAR + 2048;              // [4 waves][2]
constexpr int US_SMEM = US_WMX + 64;

// developer probe (CSAM_DEFS_decoder_fused=-DCSAM_UP_RANKPROBE): the instruction mix and register footprint of a first conv
// with K = 128 (the rank-112 form of HISTORY.md §8(7): 64 registers of per-prompt operand instead of 128 of W1) plus the per-image
// table added to the accumulators, at THREE workgroups per CU.  Numerically meaningless (it multiplies half the key channels
// and adds the other half): it answers "does a third wave per SIMD pay" before the producer chain is rebuilt for it.
#define US_WG_PER_CU 2
#define US_KS 8
#define US_MG 2                               /* 16-token sub-tiles that share one pass over W2' */
// tiles_per_wg > 0 (round 4, batches that cannot give every workgroup whole prompts): the workgroup walks a RANGE of 32-token
// tiles that may start and end inside a prompt; the per-plane maxima then go to stats by atomic max (csam_upscale_fused's
// protocol: the launcher initialises stats first)
__global__ __launch_bounds__(256, US_WG_PER_CU) void upscale_stream_kernel(UpArgs p, int B, int prompts_per_wg, int tiles_per_wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const int pos = wave;
  const unsigned xoff = ((tid >> 5) * 256 + (((tid & 31) ^ ((tid >> 5) & 15)) * 8)) * 2;   // see i2t_stream_kernel
  constexpr int PIECE = 256 * 16;
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;
  constexpr int TPP = 4096 / US_TOK;                 // 128 tiles per prompt
  int first, last;
  if (tiles_per_wg > 0) {
    first = blockIdx.x * tiles_per_wg;
    last = min(first + tiles_per_wg, B * TPP);
  } else {
    first = blockIdx.x * prompts_per_wg * TPP;
    last = min(blockIdx.x * prompts_per_wg + prompts_per_wg, B) * TPP;
  }
  if (first >= last) return;

  // ---- launch-resident: W1 slice in registers, W2' and the small parameter vectors in LDS
  half8_t w1[4][US_KS];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int ks = 0; ks < US_KS; ++ks)
      w1[ni][ks] = *(const half8_t*)(p.W1 + (long)(pos * 64 + ni * 16 + fr) * 256 + ks * 32 + fg * 8);
#pragma unroll
  for (int it = 0; it < 4; ++it) {                   // W2': 128 rows x 8 slots, 128-B rows, slot ^= row & 7
    const int cc = tid + it * 256;
    const int row = cc >> 3, sl = cc & 7;
    glds16(p.W2 + (long)row * 64 + ((sl ^ (row & 7)) * 8), smem + US_W2S + (cc & ~63) * 16);
  }
  float* par = (float*)(smem + US_PAR);
  if (tid < 128) par[tid] = p.b2[tid];
  par[128 + tid] = p.b1[tid];
  if (tid < 64) {
    par[384 + tid] = p.ln_g[tid];
    par[448 + tid] = p.ln_b[tid];
  }
  float* outs = (float*)(smem + US_OUT);
  float* wmx = (float*)(smem + US_WMX);

  auto issue_x = [&](int t, int buf) {
    const char* src = (const char*)(p.X + (long)t * US_TOK * 256);
    const unsigned dst = lds0 + buf * US_BUF + wave * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) i2s_glds16(src + i * PIECE, xoff ^ ((i & 1) << 7), dst + i * PIECE);
  };

  half8_t hhi, hlo;
  float pmax0 = -INFINITY, pmax1 = -INFINITY;        // running max of the planes this thread stores (l0, l0 + 2)
  issue_x(first, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int t = first; t < last; ++t) {
    const int cur = (t - first) & 1;
    const char* xb = smem + cur * US_BUF;
    const int b = t / TPP, tp = t - b * TPP;
    const int i0 = tp >> 1, half = tp & 1;
    const bool newp = tp == 0 || t == first;         // first tile of a prompt for this workgroup
    if (newp && tid < 64) {                          // new prompt: hyper-network A fragments (hi/lo fp16 split)
      half8_t hi = {0, 0, 0, 0, 0, 0, 0, 0}, lo = {0, 0, 0, 0, 0, 0, 0, 0};
      const int i = tid & 15, g = tid >> 4;
      if (i < 4) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int cch = ((e >= 4) ? 16 : 0) + g * 4 + (e & 3);
          const float h = p.hyper[((long)b * 4 + i) * 32 + cch];
          hi[e] = (half_t)h;
          lo[e] = (half_t)(h - (float)hi[e]);
        }
      }
      *(half8_t*)(smem + US_HFR + tid * 16) = hi;
      *(half8_t*)(smem + US_HFR + 1024 + tid * 16) = lo;
    }
    // (a) tile t landed (retired at barrier (b) of the previous tile, before its stores); outs free again
    I2S_BARRIER();
    if (t + 1 < last) issue_x(t + 1, cur ^ 1);
    if (newp) {
      hhi = *(const half8_t*)(smem + US_HFR + lane * 16);
      hlo = *(const half8_t*)(smem + US_HFR + 1024 + lane * 16);
    }

    // ---- GEMM1 + LayerNorm2d + GELU per 16-token sub-tile: [64 co of this position] x [16 tokens], K = 256.
    // The ablation (r02) showed the GEMM1 phase at twice its MFMA time: with one fragment set in flight the
    // ~250-cycle LDS latency of a loaded CU is exposed on each of the 8 k-steps (4 MFMAs = 64 cycles of cover).  One
    // sub-tile at a time halves the live accumulators (16 instead of 32 registers) and pays for a 4-deep fragment
    // ring; sub-tile 1's MFMAs are independent of sub-tile 0's LayerNorm / GELU and may issue beneath them.
#pragma unroll
    for (int g = 0; g < 2 / US_MG; ++g) {
    half8_t xf2[US_MG][2];
#pragma unroll
    for (int m = 0; m < US_MG; ++m) {
      const int mi = g * US_MG + m;
      floatx4 a1[4];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) a1[ni] = *(const floatx4*)(par + 128 + pos * 64 + ni * 16 + fg * 4);
      {
        constexpr int XD = CSAM_UP_XDEPTH;              // fragment sets in flight
        half8_t xf[XD];
#pragma unroll
        for (int ks = 0; ks < XD - 1; ++ks) xf[ks] = *(const half8_t*)(xb + (mi * 16 + fr) * 512 + (((ks * 4 + fg) ^ fr) << 4));
        if (CSAM_UP_PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < US_KS; ++ks) {
          if (ks + XD - 1 < US_KS)
            xf[(ks + XD - 1) % XD] = *(const half8_t*)(xb + (mi * 16 + fr) * 512 + ((((ks + XD - 1) * 4 + fg) ^ fr) << 4));
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            a1[ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1[ni][ks], xf[ks % XD], a1[ni], 0, 0, 0);
          // CSAM_UP_PIN: a scheduling barrier per k-step.  With the memory fence alone the MFMAs float up to their fragment read:
          // `ds_read_b128; s_waitcnt lgkmcnt(0); 4 x v_mfma` on six of the eight k-steps (ISA of round 6), the ring collapsed to
          // one register quad
          if (CSAM_UP_PIN) __builtin_amdgcn_sched_barrier(0);
          else asm volatile("" ::: "memory");
        }
      }
      // LayerNorm2d statistics over the 64 channels of a pixel: in-lane partial sums (16 channels, packed), the
      // four lane groups reduced by an fp32 ones-MFMA (every lane of the token receives the totals)
      float2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) {
        const float2_t lo = {a1[ni][0], a1[ni][1]}, hi = {a1[ni][2], a1[ni][3]};
        s2 += lo;
        q2 = __builtin_elementwise_fma(lo, lo, q2);
        s2 += hi;
        q2 = __builtin_elementwise_fma(hi, hi, q2);
      }
      const floatx4 ssum = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, s2[0] + s2[1], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const floatx4 qsum = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, q2[0] + q2[1], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      const float mean = ssum[0] * (1.f / 64.f);
      const float var = fmaxf(qsum[0] * (1.f / 64.f) - mean * mean, 0.f);
      const float rstd = rsqrtf(var + p.eps);          // v_rsq_f32 (1 ulp), as every other LayerNorm here; the IEEE sqrt + divide was ~30 dependent VALU instructions per sub-tile
      const float nmr = -mean * rstd;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {                // two N tiles (4 channel pairs) at a time
        float2_t z[4];
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          const int ni = nh * 2 + n2;
          const floatx4 gv = *(const floatx4*)(par + 384 + ni * 16 + fg * 4);
          const floatx4 bv = *(const floatx4*)(par + 448 + ni * 16 + fg * 4);
#pragma unroll
          for (int e = 0; e < 4; e += 2) {
            const float2_t av = {a1[ni][e], a1[ni][e + 1]};
            const float2_t g2 = {gv[e], gv[e + 1]}, b2 = {bv[e], bv[e + 1]};
            const float2_t nrm = __builtin_elementwise_fma(av, (float2_t){rstd, rstd}, (float2_t){nmr, nmr});
            z[n2 * 2 + (e >> 1)] = __builtin_elementwise_fma(nrm, g2, b2);
          }
        }
        csam_gelu_poly2_n<4>(z);
#pragma unroll
        for (int q = 0; q < 4; ++q) {                 // ni = nh*2 + (q>>1), e = (q&1)*2 -> xf2[nh][(ni&1)*4 + e ..]
          xf2[m][nh][(q >> 1) * 4 + (q & 1) * 2] = (half_t)z[q][0];
          xf2[m][nh][(q >> 1) * 4 + (q & 1) * 2 + 1] = (half_t)z[q][1];
        }
      }
    }
    // second conv + GELU + hyper product in two halves of its 8 N tiles (= output sub-positions 0,1 then 2,3), so that
    // only 32 accumulator registers are live beside the 128 of the resident W1 slice
#if CSAM_UP_ONE_STORE_BLOCK
    floatx4 mres[2][2][US_MG];
#endif
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      floatx4 a2[US_MG][4];                             // seeded with the second conv's bias
#pragma unroll
      for (int n2 = 0; n2 < 4; ++n2) {
        const floatx4 b2v = *(const floatx4*)(par + (ph * 4 + n2) * 16 + fg * 4);
#pragma unroll
        for (int m = 0; m < US_MG; ++m) a2[m][n2] = b2v;
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
          const int row = (ph * 4 + n2) * 16 + fr;
          const half8_t wf = *(const half8_t*)(smem + US_W2S + row * 128 + (((s2 * 4 + fg) ^ (row & 7)) << 4));
#pragma unroll
          for (int m = 0; m < US_MG; ++m) a2[m][n2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf, xf2[m][s2], a2[m][n2], 0, 0, 0);
        }
      }
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {
        const int pos2 = ph * 2 + p2;
#pragma unroll
        for (int m = 0; m < US_MG; ++m) {
          const int mi = g * US_MG + m;
          float2_t z[4];
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
            for (int e = 0; e < 4; e += 2) z[h2 * 2 + (e >> 1)] = (float2_t){a2[m][p2 * 2 + h2][e], a2[m][p2 * 2 + h2][e + 1]};
          csam_gelu_poly2_n<4>(z);
          half8_t ub;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            ub[q * 2] = (half_t)z[q][0];
            ub[q * 2 + 1] = (half_t)z[q][1];
          }
          floatx4 m4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hhi, ub, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          m4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(hlo, ub, m4, 0, 0, 0);
#if CSAM_UP_ONE_STORE_BLOCK
          mres[ph][p2][m] = m4;
#else
          if (fg == 0) {   // rows 0..3 of the product = the 4 mask logits of this pixel
            const int yy = (pos >> 1) * 2 + (pos2 >> 1);
            const int X = 4 * (mi * 16 + fr) + 2 * (pos & 1) + (pos2 & 1);
#pragma unroll
            for (int l = 0; l < 4; ++l) outs[(l * 4 + yy) * 128 + X] = m4[l];
          }
#endif
        }
      }
    }
#if CSAM_UP_ONE_STORE_BLOCK
    if (fg == 0) {   // rows 0..3 of each product = the 4 mask logits of a pixel: ONE exec-masked block per tile instead of eight
#pragma unroll
      for (int ph = 0; ph < 2; ++ph)
#pragma unroll
        for (int p2 = 0; p2 < 2; ++p2)
#pragma unroll
          for (int m = 0; m < US_MG; ++m) {
            const int pos2 = ph * 2 + p2, mi = g * US_MG + m;
            const int yy = (pos >> 1) * 2 + (pos2 >> 1);
            const int X = 4 * (mi * 16 + fr) + 2 * (pos & 1) + (pos2 & 1);
#pragma unroll
            for (int l = 0; l < 4; ++l) outs[(l * 4 + yy) * 128 + X] = mres[ph][p2][m][l];
          }
    }
#endif
    }   // g
    // (b) output half-rows complete; also retires the next tile's LDS-DMA (a whole tile old) BEFORE this tile's stores
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int rowid = it * 8 + (tid >> 5), x4 = tid & 31;      // rowid = l*4 + yy
        const floatx4 v = *(const floatx4*)(outs + rowid * 128 + x4 * 4);
        const int l = rowid >> 2, yy = rowid & 3;
        char* dst = (char*)(p.masks + (((long)b * 4 * 256) + 4 * i0) * 256 + half * 128);     // uniform base
        const unsigned voff = (unsigned)(l * 65536 + yy * 256 + x4 * 4) * 4u;                 // plane, row, pixel
        asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(voff), "v"(v), "s"(dst) : "memory");   // see i2s_store16
        const float mxv = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (it == 0) pmax0 = fmaxf(pmax0, mxv);
        else pmax1 = fmaxf(pmax1, mxv);
      }
    }
    if ((tp == TPP - 1 || t == last - 1) && p.stats) {   // prompt (or this workgroup's part of it) complete: per-plane max (planes l0 = tid>>7 and l0 + 2)
      const float m0 = csam_wave_max(pmax0), m1 = csam_wave_max(pmax1);
      if (lane == 0) {
        wmx[wave * 2] = m0;
        wmx[wave * 2 + 1] = m1;
      }
      I2S_BARRIER();
      if (tid < 4) {   // plane tid: waves (tid & 1) * 2 and +1 hold it, slot tid >> 1
        const int w0 = (tid & 1) * 2, sl = tid >> 1;
        const float m = fmaxf(wmx[w0 * 2 + sl], wmx[(w0 + 1) * 2 + sl]);
        if (tiles_per_wg > 0) {
          atomic_max_float(p.stats + ((long)b * 4 + tid) * 2, m);
        } else {
          p.stats[((long)b * 4 + tid) * 2] = m;
          p.stats[((long)b * 4 + tid) * 2 + 1] = 0.f;
        }
      }
      pmax0 = -INFINITY;
      pmax1 = -INFINITY;
    }
  }
}


// -----------------------------------------------------------------------------------------------------
// (Round 4 built this kernel with SPECIALISED waves -- 4 matrix waves + 4 GELU waves per CU, lane-private LDS mailbox -- and
// measured it slower, 4.55 vs 3.87 ms per 2048 prompts: profiles/r04_upscale_wave_specialised.txt, HISTORY.md 4.2e.  The code is gone.)
}  // namespace

__global__ void stats_init_kernel(float* stats, int rows) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < rows) {
    stats[i * 2] = -INFINITY;
    stats[i * 2 + 1] = 0.f;
  }
}

extern "C" int csam_upscale_fused(void* stream, const void* keys_f16, const void* W1_f16, const float* b1,
                                  const float* ln_gamma, const float* ln_beta, float eps, const void* W2_perm_f16,
                                  const float* b2, const float* hyper, float* masks, float* stats_or_null, int B) {
  CSAM_REQUIRE(keys_f16 && W1_f16 && b1 && ln_gamma && ln_beta && W2_perm_f16 && b2 && hyper && masks && B > 0,
               "csam_upscale_fused: bad args");
  UpArgs a;
  a.X = (const half_t*)keys_f16; a.W1 = (const half_t*)W1_f16; a.b1 = b1; a.ln_g = ln_gamma; a.ln_b = ln_beta;
  a.eps = eps; a.W2 = (const half_t*)W2_perm_f16; a.b2 = b2; a.hyper = hyper; a.masks = masks;
  a.stats = stats_or_null;
  if (stats_or_null)
    hipLaunchKernelGGL(stats_init_kernel, dim3(csam_cdiv(B * 4, 256)), dim3(256), 0, (hipStream_t)stream, stats_or_null,
                       B * 4);
  static csam_once_t attr_set;
  if (csam_first_call(attr_set))
    hipFuncSetAttribute((const void*)upscale_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, UP_SMEM);
  hipLaunchKernelGGL(upscale_fused_kernel, dim3(64, B), dim3(256), UP_SMEM, (hipStream_t)stream, a);
  CSAM_LAUNCH_CHECK("csam_upscale_fused");
  return CSAM_OK;
}

extern "C" int csam_upscale_stream(void* stream, const void* keys_f16, const void* W1_f16, const float* b1,
                                   const float* ln_gamma, const float* ln_beta, float eps, const void* W2_perm_f16,
                                   const float* b2, const float* hyper, float* masks, float* stats_or_null, int B) {
  CSAM_REQUIRE(keys_f16 && W1_f16 && b1 && ln_gamma && ln_beta && W2_perm_f16 && b2 && hyper && masks && B > 0,
               "csam_upscale_stream: bad args");
  UpArgs a;
  a.X = (const half_t*)keys_f16; a.W1 = (const half_t*)W1_f16; a.b1 = b1; a.ln_g = ln_gamma; a.ln_b = ln_beta;
  a.eps = eps; a.W2 = (const half_t*)W2_perm_f16; a.b2 = b2; a.hyper = hyper; a.masks = masks;
  a.stats = stats_or_null;
  static csam_once_t once;
  const int n_cu = csam_cu_count();
  if (csam_first_call(once))
    (void)hipFuncSetAttribute((const void*)upscale_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, US_SMEM);
  if (B < US_WG_PER_CU * n_cu) {                       // fewer prompts than resident workgroups: ranges of tiles instead
    const int tiles = B * (4096 / US_TOK), tpw = csam_cdiv(tiles, US_WG_PER_CU * n_cu);
    if (stats_or_null)
      hipLaunchKernelGGL(stats_init_kernel, dim3(csam_cdiv(B * 4, 256)), dim3(256), 0, (hipStream_t)stream, stats_or_null, B * 4);
    hipLaunchKernelGGL(upscale_stream_kernel, dim3(csam_cdiv(tiles, tpw)), dim3(256), US_SMEM, (hipStream_t)stream, a, B, 0, tpw);
  } else {
    const int per = csam_cdiv(B, US_WG_PER_CU * n_cu);
    hipLaunchKernelGGL(upscale_stream_kernel, dim3(csam_cdiv(B, per)), dim3(256), US_SMEM, (hipStream_t)stream, a, B, per, 0);
  }
  CSAM_LAUNCH_CHECK("csam_upscale_stream");
  return CSAM_OK;
}

// =====================================================================================================
// csam_t2i_fused: token->image attention (transformer.py:173-177 and :105-112) with the K/V projections
// of the per-prompt key state fused in -- K and V never touch HBM.  Per workgroup: 128 tokens of one
// prompt, 4 waves x 32 tokens, two workgroups per CU.
//   K^T = Wk X^T            swapped MFMA orientation  -> lane (token fr, 4 consecutive dims)
//   V   = X Wv^T            NON-swapped orientation    -> lane (dim l&15, 4 consecutive tokens)
// (same LDS fragments, operands passed in the other order), which makes every later product chain
// through registers with no transpose:
//   S   = K q^T   16x16x16, A = the K registers, B = per-head q fragment -> lane (query j, 4 tokens)
//   P   = exp2(S - m_wg)  (m_wg: workgroup-wide max per (query, head) through LDS)
//   O  += P V     16x16x16, A = the P registers, B = the V registers    -> lane (dim d, queries 4g..)
// The workgroup writes ONE partial record (m, l, O[16]) per (head, query); csam's merge kernel combines
// the 32 partials of a prompt.  MODE 0 (layer 0): K / V^T of the shared image embedding are hoisted per
// image and simply loaded in those two layouts.
// HBM per prompt: read 2 MB keys (+ 65 KB partials) instead of KV GEMM write/read + attention reads (6 MB).
// =====================================================================================================
namespace {

constexpr int T2I_NW = 4;                       // waves per workgroup
constexpr int T2I_TOK = T2I_NW * 32;            // 128 tokens -> 32 partial records per prompt
constexpr int T2I_PARTS = 4096 / T2I_TOK;
constexpr int T2I_NS = 3;                       // operand ring depth (K step = 32)
constexpr int T2I_XB = T2I_TOK * 64, T2I_WB = 256 * 64;
constexpr int T2I_STG = T2I_XB + T2I_WB;        // 24 KB per stage
constexpr int T2I_QFR = T2I_NS * T2I_STG;       // [8 h][64 lanes] half4 = 4 KB behind the 72 KB ring
constexpr int T2I_SMEM = T2I_QFR + 4096;        // 76 KB: two workgroups per CU
constexpr int T2I_NREC = 18;

struct T2iArgs {
  const half_t* X;            // MODE 1: keys [B*4096, 256]
  const half_t* Wkv;          // MODE 1: [256,256]: rows 0..127 Wk, 128..255 Wv
  const float* kpe;           // MODE 1: pe Wk^T + bk, fp32 [4096,128]
  const float* bv;            // MODE 1: [128]
  const half_t* K0;           // MODE 0: hoisted K [4096,128] (bias + pe included)
  const half_t* V0T;          // MODE 0: hoisted V^T [128, 4096] (bias included)
  const half_t* q;            // [B,7,128] projected queries
  float* part;                // [B, T2I_PARTS, 8, 7, 18]
};

// 4 waves x 32 tokens per workgroup, 76 KB of LDS: two workgroups share a CU, so one's projection-load latency
// and record tail hide behind the other's MFMA / softmax work (the 8-wave, 132 KB, one-per-CU layout this replaces
// idled there).  Projection operands go through a 3-deep LDS ring of 32-wide K steps with counted vmcnt.
template <int MODE>
__global__ __launch_bounds__(256, 2) void t2i_fused_kernel(T2iArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fg = lane >> 4;
  const int b = blockIdx.y, tile = blockIdx.x;
  const int t0 = tile * T2I_TOK;
  half4_t* qfr = (half4_t*)(smem + T2I_QFR);

  // per-head q fragments (B operand of S = K q^T): lane (col j = l&15, dims 4g..4g+3); zero for j >= 7
#pragma unroll
  for (int i = tid; i < 512; i += 256) {
    const int h = i >> 6, l = i & 63, j = l & 15, g4 = (l >> 4) * 4;
    half4_t v = {0, 0, 0, 0};
    if (j < 7) v = *(const half4_t*)(p.q + ((long)b * 7 + j) * 128 + h * 16 + g4);
    qfr[i] = v;
  }

  half4_t kf[2][8], vf[2][8];   // fp16 K (A operand of S) and V (B operand of PV) per (mi, head)
  if (MODE == 1) {
    const half_t* Xb = p.X + ((long)b * 4096 + t0) * 256;
    // 64-B LDS rows: slot ^= 3 * ((row >> 2) & 1) on the global source and on the fragment reads
    const half_t* x_src[2];
    const half_t* w_src[4];
#pragma unroll
    for (int it = 0; it < 2; ++it) {              // X: 128 rows x 4 slots = 512 pieces
      const int cc = tid + it * 256, row = cc >> 2, sl = cc & 3;
      x_src[it] = Xb + (long)row * 256 + ((sl ^ (3 * ((row >> 2) & 1))) * 8);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {              // W: 256 rows x 4 slots = 1024 pieces
      const int cc = tid + it * 256, row = cc >> 2, sl = cc & 3;
      w_src[it] = p.Wkv + (long)row * 256 + ((sl ^ (3 * ((row >> 2) & 1))) * 8);
    }
    auto stage = [&](int buf, int k0) {
      char* xb = smem + buf * T2I_STG;
#pragma unroll
      for (int it = 0; it < 2; ++it) glds16(x_src[it] + k0, xb + ((tid + it * 256) & ~63) * 16);
#pragma unroll
      for (int it = 0; it < 4; ++it) glds16(w_src[it] + k0, xb + T2I_XB + ((tid + it * 256) & ~63) * 16);
    };
    floatx4 ak[2][8], av[2][8];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        ak[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
        av[mi][ni] = floatx4{0.f, 0.f, 0.f, 0.f};
      }
    const int coff = (fg ^ (3 * ((fr >> 2) & 1))) << 4;
    stage(0, 0);
    stage(1, 32);
    int cur = 0;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      // vmcnt(0), not a counted wait: stage kt+1 (issued one iteration ago) is retired here and read only in the
      // NEXT iteration.  A counted vmcnt(6) that retires stage kt and reads it right after the barrier was measured
      // racy (LDS-DMA data of other waves not yet visible: non-repeatable V rows); reads must trail the retiring
      // counted wait by a phase (cdna_hip_programming.md, 8-phase template rules).
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt + 2 < 8) stage(cur == 0 ? 2 : cur - 1, (kt + 2) * 32);
      const char* xb = smem + cur * T2I_STG;
      const char* wb = xb + T2I_XB;
      half8_t xf[2];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) xf[mi] = *(const half8_t*)(xb + (wave * 32 + mi * 16 + fr) * 64 + coff);
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        const half8_t wk = *(const half8_t*)(wb + (ni * 16 + fr) * 64 + coff);
        const half8_t wv = *(const half8_t*)(wb + (128 + ni * 16 + fr) * 64 + coff);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          ak[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk, xf[mi], ak[mi][ni], 0, 0, 0);   // [n][t]
          av[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[mi], wv, av[mi][ni], 0, 0, 0);   // [t][n]
        }
      }
      cur = cur == 2 ? 0 : cur + 1;
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const float* pk = p.kpe + (long)(t0 + wave * 32 + mi * 16 + fr) * 128 + fg * 4;
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        const floatx4 kb = *(const floatx4*)(pk + ni * 16);
        const float vb = p.bv[ni * 16 + fr];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kf[mi][ni][e] = (half_t)(ak[mi][ni][e] + kb[e]);
          vf[mi][ni][e] = (half_t)(av[mi][ni][e] + vb);
        }
      }
    }
  } else {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const int tb = t0 + wave * 32 + mi * 16;
#pragma unroll
      for (int ni = 0; ni < 8; ++ni) {
        kf[mi][ni] = *(const half4_t*)(p.K0 + (long)(tb + fr) * 128 + ni * 16 + fg * 4);
        vf[mi][ni] = *(const half4_t*)(p.V0T + (long)(ni * 16 + fr) * 4096 + tb + fg * 4);
      }
    }
  }
  __syncthreads();   // qfr visible; operand ring free (the reduction scratch below aliases it)

  // ---- S = K q^T per (mi, head): lane (query j = l&15, tokens 4g+r); workgroup max per (j, head)
  float* wmax = (float*)smem;                    // [NW waves][8 h][16 j]
  float* wred = (float*)(smem + 8192);           // [NW waves][8 h][7 j][17]: l, O[16]
  const float sc = 0.25f * 1.4426950408889634f;
  floatx4 s[2][8];
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) {
    const half4_t qb = qfr[ni * 64 + lane];
    float mx = -INFINITY;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      floatx4 a = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[mi][ni], qb, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      a *= sc;
      s[mi][ni] = a;
      mx = fmaxf(mx, fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3])));
    }
    mx = csam_max_x16(mx);
    mx = csam_max_x32(mx);
    if (fg == 0) wmax[(wave * 8 + ni) * 16 + fr] = mx;
  }
  __syncthreads();
  floatx4 o[8];
  float lsum[8], mwg[8];
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) {
    float m = wmax[ni * 16 + fr];
#pragma unroll
    for (int w = 1; w < T2I_NW; ++w) m = fmaxf(m, wmax[(w * 8 + ni) * 16 + fr]);
    mwg[ni] = m;
    o[ni] = floatx4{0.f, 0.f, 0.f, 0.f};
    float ls = 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      half4_t pb;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = csam_exp2(s[mi][ni][e] - m);
        ls += pe;
        pb[e] = (half_t)pe;
      }
      o[ni] = __builtin_amdgcn_mfma_f32_16x16x16f16(pb, vf[mi][ni], o[ni], 0, 0, 0);   // [j][d]
    }
    ls = csam_sum_x16(ls);
    ls = csam_sum_x32(ls);
    lsum[ni] = ls;
  }
  // per-wave partials -> LDS (disjoint from wmax): O lane = (d = l&15, queries j = 4g+r); l lane = (j = l&15)
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = fg * 4 + r;
      if (j < 7) wred[((wave * 8 + ni) * 7 + j) * 17 + 1 + fr] = o[ni][r];
    }
    if (fg == 0 && fr < 7) wred[((wave * 8 + ni) * 7 + fr) * 17] = lsum[ni];
  }
  // stash the workgroup max per (head, j) for the record writer
  float* mrec = (float*)(smem + 8192 + T2I_NW * 8 * 7 * 17 * 4);
  if (wave == 0 && fg == 0 && fr < 7) {
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) mrec[ni * 7 + fr] = mwg[ni];
  }
  __syncthreads();
  // reduce the waves (fixed order: deterministic) and write the record
  for (int i = tid; i < 8 * 7 * 17; i += 256) {
    const int e = i % 17, hj = i / 17;            // hj = h*7 + j
    float acc = 0.f;
#pragma unroll
    for (int w = 0; w < T2I_NW; ++w) acc += wred[(w * 56 + hj) * 17 + e];
    float* rec = p.part + (((long)b * T2I_PARTS + tile) * 56 + hj) * T2I_NREC;
    rec[1 + e] = acc;                              // e == 0 -> l, e >= 1 -> O[e-1]
    if (e == 0) rec[0] = mrec[hj];
  }
}

}  // namespace

// =====================================================================================================
// csam_t2i_shared: layer-0 token->image attention (transformer.py:173-177 with the shared image embedding).
// K and V of layer 0 are per-IMAGE constants (hoisted), so the whole prompt batch is one attention problem per
// head: (7 B) query rows x 4096 keys x 16 dims.  One wave = 16 query rows of one head, flash-style over all keys:
//   S^T = K_tile Q^T   16x16x16 (K = the head dim: one MFMA per 16 keys)   lane (query l&15, keys 4g..4g+3)
//   online softmax per 64 keys, scale in the exponent's fma
//   O^T += V^T_tile P^T  16x16x16, B operand = the P registers            lane (query, dims 4g..4g+3)
// K / V^T come as per-head tile-contiguous copies (512 B per 16 keys, built once per image), straight from L2 into
// the MFMA A operands: no LDS, no barriers, no partial records, no merge kernel.
// =====================================================================================================
namespace {

__global__ __launch_bounds__(256) void t2i_shared_kernel(const half_t* __restrict__ q, const half_t* __restrict__ Kh,
                                                         const half_t* __restrict__ Vh, half_t* __restrict__ out,
                                                         int R) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int fr = lane & 15, fg = lane >> 4;
  const int h = blockIdx.y;
  const int row0 = (blockIdx.x * 4 + wave) * 16;
  if (row0 >= R) return;
  const int row = min(row0 + fr, R - 1);
  const half4_t qb = *(const half4_t*)(q + (long)row * 128 + h * 16 + fg * 4);     // B operand: (query, dims 4g..)
  // per-head tiles: Kh [8][256 tiles][16 keys][16 d], Vh [8][256 tiles][16 d][16 keys]; lane reads its 8 bytes
  const half_t* kp = Kh + (long)h * 4096 * 16 + fr * 16 + fg * 4;
  const half_t* vp = Vh + (long)h * 4096 * 16 + fr * 16 + fg * 4;
  const float sl2 = 0.25f * 1.4426950408889634f;
  floatx4 o = {0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, l = 0.f;
  half4_t kf[4], vf[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    kf[i] = *(const half4_t*)(kp + i * 256);
    vf[i] = *(const half4_t*)(vp + i * 256);
  }
  for (int step = 0; step < 64; ++step) {
    half4_t kn[4], vn[4];
    const int nx = step + 1 < 64 ? step + 1 : step;          // prefetch the next 64 keys
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kn[i] = *(const half4_t*)(kp + (nx * 4 + i) * 256);
      vn[i] = *(const half4_t*)(vp + (nx * 4 + i) * 256);
    }
    floatx4 s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
      s[i] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[i], qb, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    float mx = fmaxf(fmaxf(s[0][0], s[0][1]), s[0][2]);
    mx = fmaxf(fmaxf(mx, s[0][3]), s[1][0]);
    mx = fmaxf(fmaxf(mx, s[1][1]), s[1][2]);
    mx = fmaxf(fmaxf(mx, s[1][3]), s[2][0]);
    mx = fmaxf(fmaxf(mx, s[2][1]), s[2][2]);
    mx = fmaxf(fmaxf(mx, s[2][3]), s[3][0]);
    mx = fmaxf(fmaxf(mx, s[3][1]), s[3][2]);
    mx = fmaxf(mx, s[3][3]);
    mx = csam_max_x16(mx);
    mx = csam_max_x32(mx);
    const float mnew = fmaxf(m, mx);
    const float alpha = csam_exp2((m - mnew) * sl2);
    m = mnew;
    const float nm = -mnew * sl2;
    float ps = 0.f;
    half4_t pb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = csam_exp2(fmaf(s[i][e], sl2, nm));
        ps += pe;
        pb[i][e] = (half_t)pe;
      }
    l = l * alpha + ps;
    o *= alpha;
#pragma unroll
    for (int i = 0; i < 4; ++i) o = __builtin_amdgcn_mfma_f32_16x16x16f16(vf[i], pb[i], o, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      kf[i] = kn[i];
      vf[i] = vn[i];
    }
  }
  l = csam_sum_x16(l);
  l = csam_sum_x32(l);
  if (row0 + fr < R) {
    const float inv = 1.f / l;
    half4_t r;
#pragma unroll
    for (int e = 0; e < 4; ++e) r[e] = (half_t)(o[e] * inv);
    *(half4_t*)(out + (long)(row0 + fr) * 128 + h * 16 + fg * 4) = r;
  }
}

}  // namespace

extern "C" int csam_t2i_shared(void* stream, const void* q_f16, const void* Kh_f16, const void* Vh_f16, void* out_f16,
                               int B) {
  CSAM_REQUIRE(q_f16 && Kh_f16 && Vh_f16 && out_f16 && B > 0, "csam_t2i_shared: bad args");
  const int R = B * 7;
  hipLaunchKernelGGL(t2i_shared_kernel, dim3(csam_cdiv(R, 64), 8), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)q_f16, (const half_t*)Kh_f16, (const half_t*)Vh_f16, (half_t*)out_f16, R);
  CSAM_LAUNCH_CHECK("csam_t2i_shared");
  return CSAM_OK;
}

// =====================================================================================================
// csam_t2i_stream: csam_t2i_fused<MODE 1> as a persistent, weight-stationary flash pass (the structure that took the
// image->token kernels from 5.7 to 3.9 ms per 2048 prompts).  A 4-wave workgroup (two per CU) walks WHOLE prompts:
//   * wave w owns heads 2w, 2w+1: its Wk and Wv row slices (4 x 16 rows x 256) stay in 128 VGPRs for the launch, so
//     the 128 KB of projection weights are no longer re-staged through LDS for every 128 tokens of keys;
//   * 32-key tiles (16 KB) are LDS-DMA'd one tile ahead into a double buffer; ONE barrier per tile;
//   * K^T = Wk X^T (+ pe Wk^T + bk as the accumulator seed, fetched a tile ahead) and V = X Wv^T (+ bv seed) chain
//     through registers into S = K q^T and O^T += V^T P^T with the query on the lane index in both, so the online
//     softmax (m, l, O rescale) is lane-local; the result leaves once per prompt -- no partial records, no merge.
// In-loop global traffic uses the inline-asm helpers above (no compiler-inserted vmcnt(0) drains).
// =====================================================================================================
namespace {

constexpr int T2S_MI = 2;                        // 16-key tiles per workgroup tile
constexpr int T2S_TOK = 16 * T2S_MI;             // 32 keys
constexpr int T2S_BUF = T2S_TOK * 512;           // 16 KB
constexpr int T2S_SMEM = 2 * T2S_BUF;
constexpr int T2S_NP = T2S_TOK * 32 / 256;       // 16-B pieces per thread and tile

struct T2sArgs {
  const half_t* X; const half_t* Wkv; const float* kpe; const float* bv; const half_t* q; half_t* out;
  int B; int T;
};

__global__ __launch_bounds__(256, 2) void t2i_stream_kernel(T2sArgs p, int prompts_per_wg) {
  constexpr int MI = T2S_MI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned xoff = ((tid >> 5) * 256 + (((tid & 31) ^ ((tid >> 5) & 15)) * 8)) * 2;   // see i2t_stream_kernel
  constexpr int PIECE = 256 * 16;
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;
  const int tpp = p.T / T2S_TOK;
  const int b_first = blockIdx.x * prompts_per_wg;
  const int b_last = min(b_first + prompts_per_wg, p.B);
  if (b_first >= b_last) return;
  const int first = b_first * tpp, last = b_last * tpp;

  // ---- launch-resident weight slices: rows of Wk (0..127) and Wv (128..255) of this wave's two heads
  half8_t wk[2][8], wv[2][8];
  float bv_r[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int row = (wave * 2 + hh) * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      wk[hh][ks] = *(const half8_t*)(p.Wkv + (long)row * 256 + ks * 32 + fg * 8);
      wv[hh][ks] = *(const half8_t*)(p.Wkv + (long)(128 + row) * 256 + ks * 32 + fg * 8);
    }
    bv_r[hh] = p.bv[row];
  }

  auto issue_x = [&](int t, int buf) {
    const char* src = (const char*)(p.X + (long)t * T2S_TOK * 256);      // prompts are contiguous: tile t of the batch
    const unsigned dst = lds0 + buf * T2S_BUF + wave * 1024;
#pragma unroll
    for (int i = 0; i < T2S_NP; ++i) i2s_glds16(src + i * PIECE, xoff ^ ((i & 1) << 7), dst + i * PIECE);
  };
  floatx4 ak[2][MI];                             // K^T accumulators, seeded with (pe Wk^T + bk)[key]
  auto fetch_seed = [&](int t) {
    const int t0 = (t % tpp) * T2S_TOK;
    const char* base = (const char*)(p.kpe + (long)t0 * 128);
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        ak[hh][mi] = i2s_load16(base + mi * 16 * 128 * 4, (fr * 128 + (wave * 2 + hh) * 16 + fg * 4) * 4);
  };
  half4_t qb[2];
  floatx4 o[2];
  float m[2], l[2];
  auto new_prompt = [&](int b) {
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      qb[hh] = half4_t{0, 0, 0, 0};
      if (fr < 7) qb[hh] = *(const half4_t*)(p.q + ((long)b * 7 + fr) * 128 + (wave * 2 + hh) * 16 + fg * 4);
      asm volatile("" : "+v"(qb[hh]));             // tracked load: the compiler's (full) wait lands here, once per prompt
      o[hh] = floatx4{0.f, 0.f, 0.f, 0.f};
      m[hh] = -INFINITY;
      l[hh] = 0.f;
    }
  };

  const float sc = 0.25f * 1.4426950408889634f;
  issue_x(first, 0);
  fetch_seed(first);
  for (int t = first; t < last; ++t) {
    const int cur = (t - first) & 1;
    const char* xb = smem + cur * T2S_BUF;
    const int tp = t % tpp;
    if (tp == 0) new_prompt(t / tpp);
    // tile t and its seeds have landed (issued a tile ago); everyone is done reading the other buffer
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 1 < last) issue_x(t + 1, cur ^ 1);

    // ---- projections, K = 256 channels in 8 steps; one key fragment feeds 4 MFMAs (K and V of both heads)
    floatx4 av[2][MI];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) av[hh][mi] = floatx4{bv_r[hh], bv_r[hh], bv_r[hh], bv_r[hh]};
    half8_t xf[2][MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) xf[0][mi] = *(const half8_t*)(xb + (mi * 16 + fr) * 512 + ((fg ^ fr) << 4));
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      if (ks + 1 < 8) {
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          xf[(ks + 1) & 1][mi] = *(const half8_t*)(xb + (mi * 16 + fr) * 512 + ((((ks + 1) * 4 + fg) ^ fr) << 4));
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          ak[hh][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wk[hh][ks], xf[ks & 1][mi], ak[hh][mi], 0, 0, 0);   // [d][key]
          av[hh][mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[ks & 1][mi], wv[hh][ks], av[hh][mi], 0, 0, 0);   // [key][d]
        }
      asm volatile("" ::: "memory");
    }
    half4_t kf[2][MI], vf[2][MI];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          kf[hh][mi][e] = (half_t)ak[hh][mi][e];
          vf[hh][mi][e] = (half_t)av[hh][mi][e];
        }
    asm volatile("" ::: "memory");
    if (t + 1 < last) fetch_seed(t + 1);           // the K accumulators are free: next tile's seeds fly under the softmax

    // ---- online softmax + PV per head; query on the lane index in S, P^T and O^T
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      floatx4 sa[MI];
      float mx = -INFINITY;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        sa[mi] = __builtin_amdgcn_mfma_f32_16x16x16f16(kf[hh][mi], qb[hh], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        mx = fmaxf(mx, fmaxf(fmaxf(sa[mi][0], sa[mi][1]), fmaxf(sa[mi][2], sa[mi][3])));
      }
      mx = csam_max_x16(mx);
      mx = csam_max_x32(mx);
      const float mnew = fmaxf(m[hh], mx);
      const float alpha = csam_exp2((m[hh] - mnew) * sc);
      m[hh] = mnew;
      const float nm = -mnew * sc;
      float ps = 0.f;
      half4_t pb[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = csam_exp2(fmaf(sa[mi][e], sc, nm));
          ps += pe;
          pb[mi][e] = (half_t)pe;
        }
      l[hh] = l[hh] * alpha + ps;
      o[hh] *= alpha;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) o[hh] = __builtin_amdgcn_mfma_f32_16x16x16f16(vf[hh][mi], pb[mi], o[hh], 0, 0, 0);
    }

    if (tp == tpp - 1) {                           // prompt complete: normalise and write [7][128] fp16
      const int b = t / tpp;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        float ls = l[hh];
        ls = csam_sum_x16(ls);
        ls = csam_sum_x32(ls);
        const float inv = 1.f / ls;
        if (fr < 7) {
          half4_t r;
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = (half_t)(o[hh][e] * inv);
          *(half4_t*)(p.out + ((long)b * 7 + fr) * 128 + (wave * 2 + hh) * 16 + fg * 4) = r;
        }
      }
    }
  }
}

}  // namespace

extern "C" int csam_t2i_stream(void* stream, const void* X_f16, const void* Wkv_f16, const float* kpe, const float* bv,
                               const void* q_f16, void* out_f16, int B, int T) {
  CSAM_REQUIRE(X_f16 && Wkv_f16 && kpe && bv && q_f16 && out_f16, "csam_t2i_stream: null pointer");
  CSAM_REQUIRE(B > 0 && T > 0 && T % T2S_TOK == 0, "csam_t2i_stream: T must be a multiple of %d", T2S_TOK);
  T2sArgs a;
  a.X = (const half_t*)X_f16; a.Wkv = (const half_t*)Wkv_f16; a.kpe = kpe; a.bv = bv;
  a.q = (const half_t*)q_f16; a.out = (half_t*)out_f16; a.B = B; a.T = T;
  const int n_cu = csam_cu_count();
  const int per = csam_cdiv(B, 2 * n_cu);           // whole prompts per workgroup, two workgroups per CU
  hipLaunchKernelGGL(t2i_stream_kernel, dim3(csam_cdiv(B, per)), dim3(256), T2S_SMEM, (hipStream_t)stream, a, per);
  CSAM_LAUNCH_CHECK("csam_t2i_stream");
  return CSAM_OK;
}

// =====================================================================================================
// csam_t2i_rank (round 2): the token->image attention of layers 1 / final in RANK-56 form.  csam_t2i_stream projects
// every key: K^T = Wk X^T and V = X Wv^T are 64 of its 72 MFMAs per 32-key tile, and that kernel is issue-bound (VALU +
// MFMA cycles add up on a SIMD, HISTORY.md section 4.1).  With only 7 queries per prompt the projections fold into the
// token side:
//     scores[(h,j), t] = q_hj . (Wk_h (x_t + pe_t) + bk_h) = (Wk_h^T q_hj) . x_t  +  q_hj . (Wk_h pe_t)  (+ const per row)
//     out[(h,j), :]    = sum_t p_t (Wv_h x_t + bv_h)       = Wv_h (sum_t p_t x_t) + bv_h
// i.e. S^T = X Qp^T (+ kpe qblk^T) with Qp = the 56 (padded 64) back-projected queries of the prompt (csam_t2i_rank_prep),
// and Y^T = X^T P^T, the probability-weighted sum of the RAW key rows; Wv and the out-projection are applied afterwards
// by one ordinary GEMM over K = 8 heads x 256 (weights folded per model on the host).  Per tile and wave: 18 + 16 MFMAs
// instead of 72, one softmax over 16 (head, query) rows on the lanes instead of two over 7-of-16 -- and no fp32->fp16
// casts of K / V.  The row constant q.bk does not move a softmax and is dropped.  X^T fragments come from the row-major
// LDS tile through ds_read_b64_tr_b16 (lane mapping: profiles/r02_ds_read_tr_probe.txt).
// Wave w owns heads 2w, 2w+1: softmax rows r = 8*hh + j (j = 7 padding).  Queries arrive pre-multiplied by
// 0.25 * log2(e), so the scores are already in base-2 units.
// =====================================================================================================
namespace {

struct T2rArgs {
  const half_t* X;        // [B, T, 256] key state
  const half_t* Qp;       // [B, 64, 256] back-projected queries, row = 16*wave + 8*hh + j  (csam_t2i_rank_prep)
  const half_t* qs;       // [B, 7, 128] scaled queries (for the key_pe term)
  const half_t* kpe;      // [T, 128] fp16  pe Wk^T
  half_t* Y;              // [B, 7, 8, 256] = softmax-weighted mean of the key rows per (query, head)
  int B; int T;
};

typedef __fp16 fp16x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ half4_t ds_tr_b64(const char* lds_ptr) {
  const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(lptr_t)lds_ptr);
  return half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
}

// Qp[b][16*w + 8*hh + j][c] = sum_d qs[b][j][(2w+hh)*16 + d] * Wk[(2w+hh)*16 + d][c]   (j = 7: zero row)
__global__ __launch_bounds__(256) void t2i_rank_prep_kernel(const half_t* __restrict__ qs, const half_t* __restrict__ Wk,
                                                            half_t* __restrict__ Qp, int B) {
  __shared__ float q[7 * 128];
  const int c = threadIdx.x;
  float w[128];                                   // column c of Wk, kept over the workgroup's prompts
#pragma unroll
  for (int r = 0; r < 128; ++r) w[r] = (float)Wk[(long)r * 256 + c];
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    for (int i = c; i < 7 * 128; i += 256) q[i] = (float)qs[(long)b * 7 * 128 + i];
    __syncthreads();
    half_t* dst = Qp + (long)b * 64 * 256 + c;
#pragma unroll
    for (int head = 0; head < 8; ++head) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float acc = 0.f;
        if (j < 7) {
#pragma unroll
          for (int d = 0; d < 16; ++d) acc = fmaf(w[head * 16 + d], q[j * 128 + head * 16 + d], acc);
        }
        dst[(long)(head * 8 + j) * 256] = (half_t)acc;        // head*8 + j == 16*(head>>1) + 8*(head&1) + j
      }
    }
  }
}

#define T2R_OCC 2
__global__ __launch_bounds__(256, T2R_OCC) void t2i_rank_kernel(T2rArgs p, int prompts_per_wg) {
  constexpr int MI = T2S_MI;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned xoff = ((tid >> 5) * 256 + (((tid & 31) ^ ((tid >> 5) & 15)) * 8)) * 2;   // see i2t_stream_kernel
  constexpr int PIECE = 256 * 16;
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;
  const int tpp = p.T / T2S_TOK;
  const int b_first = blockIdx.x * prompts_per_wg;
  const int b_last = min(b_first + prompts_per_wg, p.B);
  if (b_first >= b_last) return;
  const int first = b_first * tpp, last = b_last * tpp;

  auto issue_x = [&](int t, int buf) {
    const char* src = (const char*)(p.X + (long)t * T2S_TOK * 256);
    const unsigned dst = lds0 + buf * T2S_BUF + wave * 1024;
#pragma unroll
    for (int i = 0; i < T2S_NP; ++i) i2s_glds16(src + i * PIECE, xoff ^ ((i & 1) << 7), dst + i * PIECE);
  };
  // key_pe fragments of the NEXT tile (A operand of the pe term): keys mi*16 + fr, dims wave*32 + fg*8 .. +7
  floatx4 kpf[MI];
  auto fetch_kpe = [&](int t) {
    const int t0 = (t % tpp) * T2S_TOK;
    const char* base = (const char*)(p.kpe + (long)t0 * 128);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) kpf[mi] = i2s_load16(base + mi * 16 * 128 * 2, (fr * 128 + wave * 32 + fg * 8) * 2);
  };
  // transposed-read byte offsets inside a key tile: 16-lane group fg supplies the k-slots of keys {4fg..4fg+3} and
  // {16+4fg..16+4fg+3}; lane pl of the group points at row kb + (pl >> 2), dims (pl & 3)*4..+3 of a 16-dim block
  const int pl = lane & 15;
  unsigned troff[2];
#pragma unroll
  for (int kh = 0; kh < 2; ++kh) {
    const int row = kh * 16 + fg * 4 + (pl >> 2);
    troff[kh] = row * 512 + (pl & 1) * 8;                 // + ((chunk ^ (row & 15)) << 4) per 16-dim block, chunk = 2n + ((pl&3)>>1)
  }
  const int trow[2] = {(fg * 4 + (pl >> 2)) & 15, (16 + fg * 4 + (pl >> 2)) & 15};
  const int tsub = (pl & 3) >> 1;

  half8_t qp[8];              // B operand of S^T = X Qp^T: softmax row 16*wave + fr, dims ks*32 + fg*8 .. +7
  half8_t qblk;               // B operand of the pe term: dims of head 2*wave + (fg >> 1) for rows of that head, else 0
  floatx4 y[16];              // Y^T accumulators: dims n*16 + fg*4 + e, softmax row fr
  float m, l;
  auto new_prompt = [&](int b) {
    const half_t* src = p.Qp + ((long)b * 64 + wave * 16 + fr) * 256 + fg * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qp[ks] = *(const half8_t*)(src + ks * 32);
    qblk = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
    const int hh = fr >> 3, j = fr & 7;
    if (j < 7 && (fg >> 1) == hh)
      qblk = *(const half8_t*)(p.qs + ((long)b * 7 + j) * 128 + (wave * 2 + hh) * 16 + (fg & 1) * 8);
    asm volatile("" : "+v"(qblk), "+v"(qp[0]), "+v"(qp[7]));   // tracked loads: the compiler's wait lands here, once per prompt
#pragma unroll
    for (int n = 0; n < 16; ++n) y[n] = floatx4{0.f, 0.f, 0.f, 0.f};
    m = -INFINITY;
    l = 0.f;
  };

  issue_x(first, 0);
  fetch_kpe(first);
  for (int t = first; t < last; ++t) {
    const int cur = (t - first) & 1;
    const char* xb = smem + cur * T2S_BUF;
    const int tp = t % tpp;
    if (tp == 0) new_prompt(t / tpp);
    // tile t and its key_pe fragments have landed (issued a tile ago); everyone is done reading the other buffer
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (t + 1 < last) issue_x(t + 1, cur ^ 1);

    // ---- S^T[key][row] = sum_dim X[key][dim] Qp[row][dim]  +  kpe[key][.] . qs[row][.]
    floatx4 sa[MI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      half8_t kp8;
      __builtin_memcpy(&kp8, &kpf[mi], 16);
      sa[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kp8, qblk, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
    }
    {
      half8_t xf[2][MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) xf[0][mi] = *(const half8_t*)(xb + (mi * 16 + fr) * 512 + ((fg ^ fr) << 4));
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        if (ks + 1 < 8) {
#pragma unroll
          for (int mi = 0; mi < MI; ++mi)
            xf[(ks + 1) & 1][mi] = *(const half8_t*)(xb + (mi * 16 + fr) * 512 + ((((ks + 1) * 4 + fg) ^ fr) << 4));
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          sa[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[ks & 1][mi], qp[ks], sa[mi], 0, 0, 0);
        asm volatile("" ::: "memory");
      }
    }
    if (t + 1 < last) fetch_kpe(t + 1);            // next tile's key_pe fragments fly under the softmax

    // ---- online softmax over this lane's 8 keys of softmax row fr (base-2 scores); P^T as the B operand of Y^T += X^T P^T
    float mx = fmaxf(fmaxf(fmaxf(sa[0][0], sa[0][1]), fmaxf(sa[0][2], sa[0][3])),
                     fmaxf(fmaxf(sa[1][0], sa[1][1]), fmaxf(sa[1][2], sa[1][3])));
    mx = csam_max_x16(mx);
    mx = csam_max_x32(mx);
    const float mnew = fmaxf(m, mx);
    const float alpha = csam_exp2(m - mnew);
    m = mnew;
    float ps = 0.f;
    half8_t pb;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float pe = csam_exp2(sa[mi][e] - mnew);
        ps += pe;
        pb[mi * 4 + e] = (half_t)pe;               // k-slot 4*mi + e of lane group fg <-> key 16*mi + 4*fg + e
      }
    l = l * alpha + ps;
    if (__ballot(alpha != 1.f) != 0ull) {          // exact: alpha == 1 when no row max moved
#pragma unroll
      for (int n = 0; n < 16; ++n) y[n] *= alpha;
    }
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      // X^T fragment: dims n*16 + fr, k-slots = the same 8 keys, by two transposed 4x16 block reads
      const half4_t a0 = ds_tr_b64(xb + troff[0] + (((2 * n + tsub) ^ trow[0]) << 4));
      const half4_t a1 = ds_tr_b64(xb + troff[1] + (((2 * n + tsub) ^ trow[1]) << 4));
      const half8_t xt = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      y[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xt, pb, y[n], 0, 0, 0);
    }

    if (tp == tpp - 1) {                           // prompt complete: normalise and write Y[b][j][head][256] fp16
      const int b = t / tpp;
      float ls = l;
      ls = csam_sum_x16(ls);
      ls = csam_sum_x32(ls);
      const float inv = 1.f / ls;
      const int hh = fr >> 3, j = fr & 7;
      if (j < 7) {
        half_t* dst = p.Y + (((long)b * 7 + j) * 8 + wave * 2 + hh) * 256 + fg * 4;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
          half4_t r;
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = (half_t)(y[n][e] * inv);
          *(half4_t*)(dst + n * 16) = r;
        }
      }
    }
  }
}

}  // namespace

extern "C" int csam_t2i_rank(void* stream, const void* X_f16, const void* Wk_f16, const void* kpe_f16, const void* qs_f16,
                             void* Qp_workspace, long workspace_bytes, void* Y_f16, int B, int T) {
  CSAM_REQUIRE(X_f16 && Wk_f16 && kpe_f16 && qs_f16 && Qp_workspace && Y_f16, "csam_t2i_rank: null pointer");
  CSAM_REQUIRE(B > 0 && T > 0 && T % T2S_TOK == 0, "csam_t2i_rank: T must be a multiple of %d", T2S_TOK);
  if (workspace_bytes < (long)B * 64 * 256 * 2) {
    csam_set_error("csam_t2i_rank: workspace too small (%ld < %ld)", workspace_bytes, (long)B * 64 * 256 * 2);
    return CSAM_ERR_WORKSPACE;
  }
  const int n_cu = csam_cu_count();
  hipLaunchKernelGGL(t2i_rank_prep_kernel, dim3(B < 2 * n_cu ? B : 2 * n_cu), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)qs_f16, (const half_t*)Wk_f16, (half_t*)Qp_workspace, B);
  T2rArgs a;
  a.X = (const half_t*)X_f16; a.Qp = (const half_t*)Qp_workspace; a.qs = (const half_t*)qs_f16;
  a.kpe = (const half_t*)kpe_f16; a.Y = (half_t*)Y_f16; a.B = B; a.T = T;
  const int per = csam_cdiv(B, T2R_OCC * n_cu);     // whole prompts per workgroup, T2R_OCC workgroups per CU
  hipLaunchKernelGGL(t2i_rank_kernel, dim3(csam_cdiv(B, per)), dim3(256), T2S_SMEM, (hipStream_t)stream, a, per);
  CSAM_LAUNCH_CHECK("csam_t2i_rank");
  return CSAM_OK;
}

// =====================================================================================================
// csam_i2t_t2i (round 3): the image->token half-block of layer L AND the token->image attention of the NEXT block (layer
// L+1, or the final attention) in one pass over the key state.  csam_i2t_rank[_proj] writes the new keys (8.6 GB per 4096
// prompts) and csam_t2i_rank reads them straight back; the reader's queries depend on the token side only (the self-
// attention + norm1 of the next layer run BEFORE this kernel), so the read is folded into the write:
//   * one 8-wave workgroup per CU walks whole prompts; waves 0-3 are PRODUCERS (the csam_i2t_rank tile body: 16 tokens per
//     wave -> scores over 7 keys, P . M_b, residual, LayerNorm -> a swizzled [16][512 B] LDS slice -> coalesced stores),
//     waves 4-7 are READERS (the csam_t2i_rank tile body: wave w owns heads 2w, 2w+1 -> S^T = X Qp^T + kpe q^T, online
//     softmax, Y^T += X^T P^T through transposed LDS reads);
//   * the four producer slices of one step ARE a 64-key reader tile (same row pitch, same chunk ^ (row & 15) swizzle):
//     the readers consume step g - 1 from one half of a 2 x 32 KB buffer while the producers fill the other half with
//     step g; ONE workgroup barrier per 64 tokens;
//   * producer wave w and reader wave w share a SIMD: LayerNorm VALU of one runs under the MFMAs of the other
//     (tools/probe/valu_mfma_overlap.hip: a VALU wave and an MFMA wave on one SIMD overlap).
// The reader performs the arithmetic of csam_t2i_rank on bit-identical key rows in the same order, so Y is bit-identical
// to csam_i2t_rank[_proj] followed by csam_t2i_rank (tests/test_decoder_gpu.py).  The key_pe operand and Qp come from the
// same prep kernels.  HBM: the 2 (layer 0) / 2 (layer 1) MB per prompt the separate reader pulled back are gone.
// =====================================================================================================
namespace {

// The lane-group reductions of the two softmaxes (partners 16 and 32 lanes away) go through csam_max_x16 / csam_sum_x16 /
// csam_max_x32 (csam_common.h: v_permlane16_swap / v_permlane32_swap instead of __shfl_xor = ds_bpermute_b32 + `s_waitcnt
// lgkmcnt(0)`, eight per producer tile and four per reader step in the ISA of round 6; that wait also drained every fragment
// read in flight).  Bit-identical: max and + are commutative.
#define FUSE_RING 8       // M fragments in flight in the P . M phase
// FUSE_PM_PIPE (round 6, from the ISA): with the grouped form below (FUSE_RING loads, then FUSE_RING MFMAs, one fence per group)
// the machine scheduler sinks every fragment read to ONE MFMA before its use -- `ds_read_b128; s_waitcnt lgkmcnt(1); v_mfma` 32
// times per tile, two fragment registers alternating whatever FUSE_RING says (which is why rings of 4 / 8 / 16 measured equal in
// round 3) -- so each of the 32 MFMAs of the phase waits an LDS round trip less 16 cycles.  1 = the fenced pipeline, 0 = the
// grouped form.
#ifndef FUSE_PM_PIPE
#define FUSE_PM_PIPE 1
#endif
// FUSE_RD_PIPE: the same for the READER waves (score fragments pinned above their MFMAs; a ring of this many transposed-read pairs
// for the P^T . X product, its first pairs requested ahead of the softmax).  0 = the compiler's order.
#ifndef FUSE_RD_PIPE
#define FUSE_RD_PIPE 8
#endif
#define FUSE_PF_EARLY -1  // next tile's q / residual loads before P . M: 1 yes, 0 no, -1 = only in the projected form
// gamma / beta rows of the LayerNorm (layer 0: FOLD bit 1 clear) requested FUSE_GB_DEPTH channel blocks ahead of their use.  The
// compiler's own order requests block ni + 1 four packed FMAs (~30 cycles) before `s_waitcnt lgkmcnt(0)`: sixteen exposed LDS round
// trips per 16-token tile (ISA of round 6).  With a depth the first blocks are requested BEFORE the statistics (the M ring's 32
// registers are dead by then) and block ni + depth when block ni has been consumed; same loads, same arithmetic, same order of
// floating-point operations: bit-identical by construction.  0 = the compiler's order.
#ifndef FUSE_GB_DEPTH
#define FUSE_GB_DEPTH 4
#endif
template <bool PROJ>
struct IF {
  static constexpr int KP = IR_M_BYTES;
  static constexpr int TILES = IR_M_BYTES + (PROJ ? IR_KP_BYTES : 0);     // 2 x (4 producer slices = 64 keys x 512 B)
  static constexpr int STEP = 4 * IR_SLICE;                               // 32 KB
  static constexpr int PAR = TILES + 2 * STEP;
  static constexpr int SMEM = PAR + 3 * 256 * 4;
};

// FOLD (csam_i2t_t2i_fold): bit 0 = the out-projection bias is in M_b (no bias rows read from LDS: the residual's identity
// MFMAs start from zero); bit 1 = the LayerNorm's gamma / beta are folded into the CONSUMERS of these keys (upscaler first conv,
// final attention), the producer stores the plain normalised values (layer 1 only: its keys have no residual consumer).
template <bool PROJ, int FOLD>
__global__ __launch_bounds__(512, 1) void i2t_t2i_kernel(IrArgs p, T2rArgs r, int prompts_per_wg) {
  typedef IF<PROJ> G;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fg = lane >> 4;
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;
  float* par = (float*)(smem + G::PAR);
  for (int i = tid; i < 256; i += 512) {
    par[i] = p.bo[i];
    par[256 + i] = p.gamma[i];
    par[512 + i] = p.beta[i];
  }
  const int spp = p.T / 64;                          // steps (64 tokens) per prompt
  const int b_first = blockIdx.x * prompts_per_wg;
  const int b_last = min(b_first + prompts_per_wg, p.B);
  I2S_BARRIER();                                     // (0) par visible

  if (wave < 4) {
    // =============================== producers: csam_i2t_rank tile body, tile = 4 * step + wave ===============================
    const int ptid = tid;                            // 0 .. 255
    half8_t eye_lo, eye_hi;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      eye_lo[e] = (half_t)((fg < 2 && fg * 8 + e == fr) ? 1.f : 0.f);
      eye_hi[e] = (half_t)((fg >= 2 && (fg - 2) * 8 + e == fr) ? 1.f : 0.f);
    }
    const bool key7 = (fg & 1) == 1;
    floatx4 qraw[4];
    floatx4 xres[8];
    int g = 0;                                       // step counter of this workgroup (buffer parity)
    for (int b = b_first; b < b_last; ++b) {
      // every producer finished the previous prompt's last tile before the barrier that ended that step
      {
        const char* src = (const char*)(p.M + (long)b * 256 * 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int c = ptid + i * 256, row = c >> 3, sl = c & 7;
          i2s_glds16(src, (unsigned)(row * 128 + ((sl ^ (row & 7)) << 4)), lds0 + (unsigned)(wave * 1024 + i * 4096));
        }
        if constexpr (PROJ) {
          const char* ksrc = (const char*)(p.Kp + (long)b * 64 * 256);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int c = ptid + i * 256, row = c >> 5, sl = c & 31;
            i2s_glds16(ksrc, (unsigned)(row * 512 + ((sl ^ (row & 15)) << 4)),
                       lds0 + (unsigned)(G::KP + wave * 1024 + i * 4096));
          }
        }
      }
      half8_t kfr[4];
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        half8_t kv = {0, 0, 0, 0, 0, 0, 0, 0};
        const int j = fr & 7, hsel = fr >> 3;
        if (j < 7 && (fg >> 1) == hsel)
          kv = *(const half8_t*)(p.ks + ((long)b * 7 + j) * 128 + (2 * pr + hsel) * 16 + (fg & 1) * 8);
        kfr[pr] = kv;
        asm volatile("" : "+v"(kfr[pr]));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      I2S_BARRIER();                                 // (P) M_b / Kp_b of every producer wave have landed

      auto prefetch = [&](int tile) {
        const int t0 = tile * 16;
        const char* qb = (const char*)(p.Q + (long)b * p.q_bstride + (long)t0 * 128);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) qraw[pr] = i2s_load16(qb + pr * 64, (unsigned)(fr * 256 + fg * 16));
        const char* xb = (const char*)(p.X + (long)b * p.x_bstride + (long)t0 * 256);
#pragma unroll
        for (int nj = 0; nj < 8; ++nj) xres[nj] = i2s_load16(xb + nj * 64, (unsigned)(fr * 512 + fg * 16));
      };
      prefetch(wave);
      asm volatile("s_waitcnt vmcnt(0)"
                   : "+v"(qraw[0]), "+v"(qraw[1]), "+v"(qraw[2]), "+v"(qraw[3]), "+v"(xres[0]), "+v"(xres[1]), "+v"(xres[2]),
                     "+v"(xres[3]), "+v"(xres[4]), "+v"(xres[5]), "+v"(xres[6]), "+v"(xres[7])
                   :: "memory");

      for (int st = 0; st < spp; ++st, ++g) {
        const int tile = st * 4 + wave;
        const int t0 = tile * 16;
        char* slice = smem + G::TILES + (g & 1) * G::STEP + wave * IR_SLICE;
        floatx4 sc[4];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          half8_t qf;
          __builtin_memcpy(&qf, &qraw[pr], 16);
          sc[pr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr[pr], qf, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        // (see FUSE_PM_PIPE: product instantiations only)
        constexpr bool PM_PIPE = FUSE_PM_PIPE && ((!PROJ && FOLD == 1) || (PROJ && FOLD == 3));
        if constexpr (PROJ && PM_PIPE) {
          // Kp fragments j = nj * 4 + pr through a pinned ring, as P . M below (the compiler's order: one MFMA ahead)
          half8_t kr[FUSE_RING];
          auto kpread = [&](int j) {
            const int row = (j & 3) * 16 + fr;
            return *(const half8_t*)(smem + G::KP + row * 512 + ((((j >> 2) * 4 + fg) ^ fr) << 4));
          };
#pragma unroll
          for (int i = 0; i < FUSE_RING; ++i) kr[i] = kpread(i);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 32; ++j) {
            half8_t xf;
            __builtin_memcpy(&xf, &xres[j >> 2], 16);
            sc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kr[j % FUSE_RING], xf, sc[j & 3], 0, 0, 0);
            if (j + FUSE_RING < 32) kr[j % FUSE_RING] = kpread(j + FUSE_RING);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else if constexpr (PROJ) {
#pragma unroll
          for (int nj = 0; nj < 8; ++nj) {
            half8_t xf;
            __builtin_memcpy(&xf, &xres[nj], 16);
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {
              const int row = pr * 16 + fr;
              const half8_t kp = *(const half8_t*)(smem + G::KP + row * 512 + (((nj * 4 + fg) ^ fr) << 4));
              sc[pr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kp, xf, sc[pr], 0, 0, 0);
            }
          }
        }
        half8_t pf[2];
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {
          const floatx4 s4 = sc[pr];
          const float s3 = key7 ? -INFINITY : s4[3];
          float mx = fmaxf(fmaxf(s4[0], s4[1]), fmaxf(s4[2], s3));
          mx = csam_max_x16(mx);
          const float p0 = csam_exp2(s4[0] - mx), p1 = csam_exp2(s4[1] - mx);
          const float p2 = csam_exp2(s4[2] - mx), p3 = csam_exp2(s3 - mx);
          float sum = (p0 + p1) + (p2 + p3);
          sum = csam_sum_x16(sum);
          const float inv = __builtin_amdgcn_rcpf(sum);
          const int o = (pr & 1) * 4;
          pf[pr >> 1][o] = (half_t)(p0 * inv);
          pf[pr >> 1][o + 1] = (half_t)(p1 * inv);
          pf[pr >> 1][o + 2] = (half_t)(p2 * inv);
          pf[pr >> 1][o + 3] = (half_t)(p3 * inv);
        }
        floatx4 acc[16];
#pragma unroll
        for (int nj = 0; nj < 8; ++nj) {
          half8_t xf;
          __builtin_memcpy(&xf, &xres[nj], 16);
          floatx4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
          if constexpr (!(FOLD & 1)) {
            b0 = *(const floatx4*)(par + nj * 32 + fg * 4);
            b1 = *(const floatx4*)(par + nj * 32 + 16 + fg * 4);
          }
          acc[2 * nj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eye_lo, xf, b0, 0, 0, 0);
          acc[2 * nj + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(eye_hi, xf, b1, 0, 0, 0);
        }
        asm volatile("" ::: "memory");
        constexpr bool PF_EARLY = FUSE_PF_EARLY >= 0 ? FUSE_PF_EARLY != 0 : PROJ;
        if (PF_EARLY && st + 1 < spp) prefetch(tile + 4);
        // next tile's q / residual fragments: the hoisted-Q layer reads them from L2 (shared by all prompts) and issues them
        // AFTER P . M, which leaves their 48 registers to the M fragments in flight (3.92 vs 4.30 ms per 4096 prompts); the
        // projected layer reads per-prompt keys from HBM and needs the whole tile to hide them (4.63 vs 4.88 ms)
        // the product instantiations only (layer 0 <false, 1>, layer 1 <true, 3>): the comparator forms keep more rows in LDS
        // reads of their own and would spill
        if constexpr (PM_PIPE) {
          // P . M as an explicit software pipeline over k = ks * 16 + n: fragment k + FUSE_RING is requested right after MFMA k
          // and a scheduling barrier after every pair keeps both there (see FUSE_PM_PIPE above).  Per accumulator the two k-steps
          // still arrive in ascending order: bit-identical to the grouped form.
          half8_t mf[FUSE_RING];
          auto mread = [&](int k) {
            const int row = (k & 15) * 16 + fr;
            return *(const half8_t*)(smem + row * 128 + ((((k >> 4) * 4 + fg) ^ (row & 7)) << 4));
          };
#pragma unroll
          for (int i = 0; i < FUSE_RING; ++i) mf[i] = mread(i);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int k = 0; k < 32; ++k) {
            acc[k & 15] = __builtin_amdgcn_mfma_f32_16x16x32_f16(mf[k % FUSE_RING], pf[k >> 4], acc[k & 15], 0, 0, 0);
            if (k + FUSE_RING < 32) mf[k % FUSE_RING] = mread(k + FUSE_RING);
            __builtin_amdgcn_sched_barrier(0);                 // a memory fence is not enough: the MFMAs themselves float up to their reads
          }
        } else {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
          for (int n0 = 0; n0 < 16; n0 += FUSE_RING) {
            half8_t mf[FUSE_RING];
#pragma unroll
            for (int i = 0; i < FUSE_RING; ++i) {
              const int row = (n0 + i) * 16 + fr;
              mf[i] = *(const half8_t*)(smem + row * 128 + (((ks * 4 + fg) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FUSE_RING; ++i)
              acc[n0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(mf[i], pf[ks], acc[n0 + i], 0, 0, 0);
            asm volatile("" ::: "memory");
          }
        }
        }
        if (!PF_EARLY && st + 1 < spp) prefetch(tile + 4);   // lands under LayerNorm + stores
        constexpr int GBD = ((FOLD & 2) || PROJ) ? 0 : FUSE_GB_DEPTH;   // the projected form has no registers to spare (it would spill)
        floatx4 gmr[GBD > 0 ? GBD : 1], ber[GBD > 0 ? GBD : 1];
        if constexpr (GBD > 0) {
#pragma unroll
          for (int i = 0; i < GBD; ++i) {
            gmr[i] = *(const floatx4*)(par + 256 + i * 16 + fg * 4);
            ber[i] = *(const floatx4*)(par + 512 + i * 16 + fg * 4);
          }
          asm volatile("" ::: "memory");                      // the requests stay above the statistics
        }
        float2_t s2 = {0.f, 0.f}, q2 = {0.f, 0.f};
#pragma unroll
        for (int ni = 0; ni < 16; ++ni) {
          const float2_t a = {acc[ni][0], acc[ni][1]}, c2 = {acc[ni][2], acc[ni][3]};
          s2 += a;
          q2 = __builtin_elementwise_fma(a, a, q2);
          s2 += c2;
          q2 = __builtin_elementwise_fma(c2, c2, q2);
        }
        const floatx4 ssum = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, s2[0] + s2[1], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const floatx4 qsum = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, q2[0] + q2[1], floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        const float mean = ssum[0] * (1.f / 256.f);
        const float var = fmaxf(qsum[0] * (1.f / 256.f) - mean * mean, 0.f);
        const float rstd = rsqrtf(var + p.eps);
        const float2_t rs2 = {rstd, rstd}, nm2 = {-mean * rstd, -mean * rstd};
#pragma unroll
        for (int ni = 0; ni < 16; ++ni) {
          if (GBD > 0 || (ni & 3) == 0) asm volatile("" ::: "memory");
          const float2_t v0 = {acc[ni][0], acc[ni][1]}, v1 = {acc[ni][2], acc[ni][3]};
          float2_t y0 = __builtin_elementwise_fma(v0, rs2, nm2), y1 = __builtin_elementwise_fma(v1, rs2, nm2);
          if constexpr (!(FOLD & 2)) {
            floatx4 gm, be;
            if constexpr (GBD > 0) {
              gm = gmr[ni % (GBD > 0 ? GBD : 1)];
              be = ber[ni % (GBD > 0 ? GBD : 1)];
            } else {
              gm = *(const floatx4*)(par + 256 + ni * 16 + fg * 4);
              be = *(const floatx4*)(par + 512 + ni * 16 + fg * 4);
            }
            const float2_t g0 = {gm[0], gm[1]}, g1 = {gm[2], gm[3]}, b0 = {be[0], be[1]}, b1 = {be[2], be[3]};
            y0 = __builtin_elementwise_fma(y0, g0, b0);
            y1 = __builtin_elementwise_fma(y1, g1, b1);
          }
          const int chunk = ni * 2 + (fg >> 1);
          *(half4_t*)(slice + fr * 512 + ((chunk ^ fr) << 4) + (fg & 1) * 8) =
              half4_t{(half_t)y0[0], (half_t)y0[1], (half_t)y1[0], (half_t)y1[1]};
          if constexpr (GBD > 0) {
            if (ni + GBD < 16) {                               // block ni + depth into the registers block ni just left
              gmr[ni % (GBD > 0 ? GBD : 1)] = *(const floatx4*)(par + 256 + (ni + GBD) * 16 + fg * 4);
              ber[ni % (GBD > 0 ? GBD : 1)] = *(const floatx4*)(par + 512 + (ni + GBD) * 16 + fg * 4);
            }
          }
        }
        // next tile's operands land here; the asm RE-DEFINES them, so whatever copies the register allocator places on
        // the loop edge move landed data (an in-flight asm load result must never reach a compiler-made copy)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                     : "+v"(qraw[0]), "+v"(qraw[1]), "+v"(qraw[2]), "+v"(qraw[3]), "+v"(xres[0]), "+v"(xres[1]), "+v"(xres[2]),
                       "+v"(xres[3]), "+v"(xres[4]), "+v"(xres[5]), "+v"(xres[6]), "+v"(xres[7])
                     :: "memory");
        char* obase = (char*)(p.out + ((long)b * p.T + t0) * 256);
        {
          half8_t rb[8];                               // all eight read-backs in flight (the accumulators are dead), then the stores
#pragma unroll
          for (int i = 0; i < 8; ++i) rb[i] = *(const half8_t*)(slice + (lane + i * 64) * 16);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int idx = lane + i * 64, row = idx >> 5, sl = idx & 31;
            i2s_store16(obase, (unsigned)(row * 512 + ((sl ^ (row & 15)) << 4)), rb[i]);
          }
        }
        I2S_BARRIER();                               // (T) step g is in LDS; the readers are done with step g - 1
      }
    }
    I2S_BARRIER();                                   // (E) pairs with the readers' drain step
  } else {
    // =============================== readers: csam_t2i_rank tile body over the producers' slices ===============================
    constexpr int MI = T2S_MI;
    const int rw = wave - 4;                         // owns heads 2 rw, 2 rw + 1
    const int pl = lane & 15;
    unsigned troff[2];
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
      const int row = kh * 16 + fg * 4 + (pl >> 2);
      troff[kh] = row * 512 + (pl & 1) * 8;
    }
    const int trow[2] = {(fg * 4 + (pl >> 2)) & 15, (16 + fg * 4 + (pl >> 2)) & 15};
    const int tsub = (pl & 3) >> 1;
    floatx4 kpf[2][MI];                              // key_pe fragments of the NEXT step, both 32-key halves
    // the asm RE-DEFINES the fragments after the wait: copies the register allocator makes on loop edges move landed data
    auto land_kpe = [&]() {
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(kpf[0][0]), "+v"(kpf[0][1]), "+v"(kpf[1][0]), "+v"(kpf[1][1])::"memory");
    };
    auto fetch_kpe = [&](int st) {
      const char* base = (const char*)(r.kpe + (long)st * 64 * 128);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          kpf[h][mi] = i2s_load16(base + (h * 32 + mi * 16) * 128 * 2, (fr * 128 + rw * 32 + fg * 8) * 2);
    };
    half8_t qp[8];
    half8_t qblk;
    floatx4 y[16];
    float m, l;
    auto new_prompt = [&](int b) {
      const half_t* src = r.Qp + ((long)b * 64 + rw * 16 + fr) * 256 + fg * 8;
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) qp[ks] = *(const half8_t*)(src + ks * 32);
      qblk = half8_t{0, 0, 0, 0, 0, 0, 0, 0};
      const int hh = fr >> 3, j = fr & 7;
      if (j < 7 && (fg >> 1) == hh)
        qblk = *(const half8_t*)(r.qs + ((long)b * 7 + j) * 128 + (rw * 2 + hh) * 16 + (fg & 1) * 8);
      asm volatile("" : "+v"(qblk), "+v"(qp[0]), "+v"(qp[7]));
#pragma unroll
      for (int n = 0; n < 16; ++n) y[n] = floatx4{0.f, 0.f, 0.f, 0.f};
      m = -INFINITY;
      l = 0.f;
    };
    // step k of this workgroup: prompt b_first + k / spp, step k % spp.  `more` = another step follows: its key_pe
    // fragments are requested now (never request registers nobody waits for: the compiler would recycle them in flight)
    auto consume = [&](int k, bool more) {
      const int st = k % spp, b = b_first + k / spp;
      if (st == 0) new_prompt(b);
      floatx4 kcur[2][MI];                           // this step's key_pe fragments (landed: see land_kpe)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) kcur[h][mi] = kpf[h][mi];
      asm volatile("" ::: "memory");
      if (more) fetch_kpe(st + 1 < spp ? st + 1 : 0);   // the next prompt starts at step 0 again
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const char* xb = smem + G::TILES + (k & 1) * G::STEP + h * T2S_BUF;
        floatx4 sa[MI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          half8_t kp8;
          __builtin_memcpy(&kp8, &kcur[h][mi], 16);
          sa[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kp8, qblk, floatx4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        {
          // key fragments of the half-step j = ks * MI + mi.  The compiler's own order puts every read ONE MFMA before its use
          // (`ds_read_b128; s_waitcnt lgkmcnt(1); v_mfma`, ISA of round 6 -- whatever the source order says); FUSE_RD_PIPE pins a
          // ring of eight fragments (all sixteen at once spill the reader)
          if (FUSE_RD_PIPE) {
            constexpr int R1 = 8;
            half8_t xr[R1];
            auto kread = [&](int j) {
              return *(const half8_t*)(xb + ((j % MI) * 16 + fr) * 512 + ((((j / MI) * 4 + fg) ^ fr) << 4));
            };
#pragma unroll
            for (int j = 0; j < R1; ++j) xr[j] = kread(j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 8 * MI; ++j) {
              sa[j % MI] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xr[j % R1], qp[j / MI], sa[j % MI], 0, 0, 0);
              if (j + R1 < 8 * MI) xr[j % R1] = kread(j + R1);
              __builtin_amdgcn_sched_barrier(0);
            }
          } else {
            half8_t xf[8][MI];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
              for (int mi = 0; mi < MI; ++mi)
                xf[ks][mi] = *(const half8_t*)(xb + (mi * 16 + fr) * 512 + (((ks * 4 + fg) ^ fr) << 4));
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
              for (int mi = 0; mi < MI; ++mi)
                sa[mi] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf[ks][mi], qp[ks], sa[mi], 0, 0, 0);
          }
        }
        // transposed key fragments of the P^T . X product: the first FUSE_RD_PIPE pairs are requested here, ahead of the
        // softmax (they depend on the keys only), the rest as the ring drains; 0 = the compiler's order (one pair ahead)
        constexpr int RD = FUSE_RD_PIPE > 0 ? FUSE_RD_PIPE : 1;
        half4_t xa0[RD], xa1[RD];
        auto tr_pair = [&](int n, half4_t& a0, half4_t& a1) {
          a0 = ds_tr_b64(xb + troff[0] + (((2 * n + tsub) ^ trow[0]) << 4));
          a1 = ds_tr_b64(xb + troff[1] + (((2 * n + tsub) ^ trow[1]) << 4));
        };
        if (FUSE_RD_PIPE) {
#pragma unroll
          for (int i = 0; i < RD; ++i) tr_pair(i, xa0[i], xa1[i]);
          __builtin_amdgcn_sched_barrier(0);
        }
        float mx = fmaxf(fmaxf(fmaxf(sa[0][0], sa[0][1]), fmaxf(sa[0][2], sa[0][3])),
                         fmaxf(fmaxf(sa[1][0], sa[1][1]), fmaxf(sa[1][2], sa[1][3])));
        mx = csam_max_x16(mx);
        mx = csam_max_x32(mx);
        const float mnew = fmaxf(m, mx);
        const float alpha = csam_exp2(m - mnew);
        m = mnew;
        float ps = 0.f;
        half8_t pb;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pe = csam_exp2(sa[mi][e] - mnew);
            ps += pe;
            pb[mi * 4 + e] = (half_t)pe;
          }
        l = l * alpha + ps;
        if (__ballot(alpha != 1.f) != 0ull) {
#pragma unroll
          for (int n = 0; n < 16; ++n) y[n] *= alpha;
        }
        if (FUSE_RD_PIPE) {
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int n = 0; n < 16; ++n) {
            const half4_t a0 = xa0[n % RD], a1 = xa1[n % RD];
            const half8_t xt = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            y[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xt, pb, y[n], 0, 0, 0);
            if (n + RD < 16) tr_pair(n + RD, xa0[n % RD], xa1[n % RD]);
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int n = 0; n < 16; ++n) {
            half4_t a0, a1;
            tr_pair(n, a0, a1);
            const half8_t xt = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            y[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xt, pb, y[n], 0, 0, 0);
          }
        }
      }
      if (more) land_kpe();                          // requested a whole step ago: no exposed latency
      if (st == spp - 1) {
        float ls = l;
        ls = csam_sum_x16(ls);
        ls = csam_sum_x32(ls);
        const float inv = 1.f / ls;
        const int hh = fr >> 3, j = fr & 7;
        if (j < 7) {
          half_t* dst = r.Y + (((long)b * 7 + j) * 8 + rw * 2 + hh) * 256 + fg * 4;
#pragma unroll
          for (int n = 0; n < 16; ++n) {
            half4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (half_t)(y[n][e] * inv);
            *(half4_t*)(dst + n * 16) = o;
          }
        }
      }
    };
    fetch_kpe(0);
    land_kpe();
    int g = 0;
    for (int b = b_first; b < b_last; ++b) {
      I2S_BARRIER();                                 // (P)
      for (int st = 0; st < spp; ++st, ++g) {
        if (g > 0) consume(g - 1, true);
        I2S_BARRIER();                               // (T)
      }
    }
    if (g > 0) consume(g - 1, false);                // drain: the last step of the last prompt
    I2S_BARRIER();                                   // (E)
  }
}

}  // namespace

// image->token half-block + the next block's token->image attention.  Wq_f16 == null: the hoisted-Q layer-0 form
// (csam_i2t_rank's operands, Q_f16 = the shared image-side queries); else the projected form (csam_i2t_rank_proj's, Q_f16 =
// qpe16).  The reader operands are csam_t2i_rank's.  workspace: csam_i2t_t2i_workspace_bytes(B) = M_b | Kp_b | Qp_b.
extern "C" long csam_i2t_t2i_workspace_bytes(int B) { return (long)B * (IR_M_BYTES + IR_KP_BYTES + 64 * 256 * 2); }

static int i2t_t2i_launch(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16, long q_prompt_stride,
                          const void* Wq_f16, const void* k_scaled_f16, const void* v_f16, const void* Wo_f16,
                          const float* bo, const float* gamma, const float* beta, float eps, void* out_f16,
                          const void* t2i_Wk_f16, const void* t2i_kpe_f16, const void* t2i_qs_f16, void* Y_f16, int B, int T,
                          void* workspace, long workspace_bytes, int fold) {
  CSAM_REQUIRE(X_f16 && Q_f16 && k_scaled_f16 && v_f16 && Wo_f16 && bo && gamma && beta && out_f16 && t2i_Wk_f16 &&
                   t2i_kpe_f16 && t2i_qs_f16 && Y_f16 && workspace,
               "csam_i2t_t2i: null pointer");
  CSAM_REQUIRE(B > 0 && T > 0 && T % 64 == 0, "csam_i2t_t2i: T must be a multiple of 64");
  CSAM_REQUIRE(fold == 0 || fold == 1 || (fold == 3 && Wq_f16),
               "csam_i2t_t2i_fold: fold must be 0, 1 (bias in M_b) or 3 (+ gamma / beta folded downstream; projected form only)");
  if (workspace_bytes < csam_i2t_t2i_workspace_bytes(B)) {
    csam_set_error("csam_i2t_t2i: workspace too small");
    return CSAM_ERR_WORKSPACE;
  }
  static csam_once_t once;
  const int n_cu = csam_cu_count();
  if (csam_first_call(once)) {
    (void)hipFuncSetAttribute((const void*)i2t_t2i_kernel<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, IF<false>::SMEM);
    (void)hipFuncSetAttribute((const void*)i2t_t2i_kernel<true, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, IF<true>::SMEM);
    (void)hipFuncSetAttribute((const void*)i2t_t2i_kernel<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, IF<false>::SMEM);
    (void)hipFuncSetAttribute((const void*)i2t_t2i_kernel<true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, IF<true>::SMEM);
    (void)hipFuncSetAttribute((const void*)i2t_t2i_kernel<true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, IF<true>::SMEM);
  }
  hipStream_t s = (hipStream_t)stream;
  half_t* Mws = (half_t*)workspace;
  half_t* Kpws = Mws + (long)B * 256 * 64;
  half_t* Qpws = Kpws + (long)B * 64 * 256;
  const dim3 pgrid(B < 2 * n_cu ? B : 2 * n_cu);
  hipLaunchKernelGGL(i2t_rank_prep_kernel, dim3(B), dim3(256), 0, s, (const half_t*)v_f16, (const half_t*)Wo_f16, Mws,
                     (fold & 1) ? bo : (const float*)nullptr);
  if (Wq_f16)
    hipLaunchKernelGGL(i2t_rank_kp_kernel, pgrid, dim3(256), 0, s, (const half_t*)k_scaled_f16, (const half_t*)Wq_f16, Kpws, B);
  hipLaunchKernelGGL(t2i_rank_prep_kernel, pgrid, dim3(256), 0, s, (const half_t*)t2i_qs_f16, (const half_t*)t2i_Wk_f16, Qpws, B);
  IrArgs a;
  a.X = (const half_t*)X_f16; a.x_bstride = x_prompt_stride; a.Q = (const half_t*)Q_f16; a.q_bstride = q_prompt_stride;
  a.ks = (const half_t*)k_scaled_f16; a.M = Mws; a.Kp = Wq_f16 ? Kpws : nullptr; a.bo = bo; a.gamma = gamma; a.beta = beta;
  a.eps = eps; a.out = (half_t*)out_f16; a.B = B; a.T = T;
  T2rArgs t;
  t.X = nullptr; t.Qp = Qpws; t.qs = (const half_t*)t2i_qs_f16; t.kpe = (const half_t*)t2i_kpe_f16; t.Y = (half_t*)Y_f16;
  t.B = B; t.T = T;
  const int per = csam_cdiv(B, n_cu);               // whole prompts per workgroup, one 8-wave workgroup per CU
  const dim3 grid(csam_cdiv(B, per));
#define CSAM_I2T_T2I_GO(PROJ_, FOLD_) \
  hipLaunchKernelGGL((i2t_t2i_kernel<PROJ_, FOLD_>), grid, dim3(512), IF<PROJ_>::SMEM, s, a, t, per)
  if (Wq_f16) {
    if (fold == 3) CSAM_I2T_T2I_GO(true, 3);
    else if (fold == 1) CSAM_I2T_T2I_GO(true, 1);
    else CSAM_I2T_T2I_GO(true, 0);
  } else {
    if (fold == 1) CSAM_I2T_T2I_GO(false, 1);
    else CSAM_I2T_T2I_GO(false, 0);
  }
#undef CSAM_I2T_T2I_GO
  CSAM_LAUNCH_CHECK("csam_i2t_t2i");
  return CSAM_OK;
}

extern "C" int csam_i2t_t2i(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16, long q_prompt_stride,
                            const void* Wq_f16, const void* k_scaled_f16, const void* v_f16, const void* Wo_f16,
                            const float* bo, const float* gamma, const float* beta, float eps, void* out_f16,
                            const void* t2i_Wk_f16, const void* t2i_kpe_f16, const void* t2i_qs_f16, void* Y_f16, int B, int T,
                            void* workspace, long workspace_bytes) {
  return i2t_t2i_launch(stream, X_f16, x_prompt_stride, Q_f16, q_prompt_stride, Wq_f16, k_scaled_f16, v_f16, Wo_f16, bo, gamma,
                        beta, eps, out_f16, t2i_Wk_f16, t2i_kpe_f16, t2i_qs_f16, Y_f16, B, T, workspace, workspace_bytes, 0);
}

// csam_i2t_t2i with constants folded out of the producer's tile loop (include/csam.h).
extern "C" int csam_i2t_t2i_fold(void* stream, const void* X_f16, long x_prompt_stride, const void* Q_f16, long q_prompt_stride,
                                 const void* Wq_f16, const void* k_scaled_f16, const void* v_f16, const void* Wo_f16,
                                 const float* bo, const float* gamma, const float* beta, float eps, void* out_f16,
                                 const void* t2i_Wk_f16, const void* t2i_kpe_f16, const void* t2i_qs_f16, void* Y_f16, int B,
                                 int T, void* workspace, long workspace_bytes, int fold) {
  return i2t_t2i_launch(stream, X_f16, x_prompt_stride, Q_f16, q_prompt_stride, Wq_f16, k_scaled_f16, v_f16, Wo_f16, bo, gamma,
                        beta, eps, out_f16, t2i_Wk_f16, t2i_kpe_f16, t2i_qs_f16, Y_f16, B, T, workspace, workspace_bytes, fold);
}

extern "C" int csam_t2i_merge_launch(void* stream, const float* part, void* out_f16, int B, int nparts);

extern "C" long csam_t2i_fused_workspace_bytes(int B) { return (long)B * T2I_PARTS * 56 * T2I_NREC * sizeof(float); }

extern "C" int csam_t2i_fused(void* stream, const void* X_f16, const void* Wkv_f16, const float* kpe, const float* bv,
                              const void* K0_f16, const void* V0T_f16, const void* q_f16, void* out_f16, int B,
                              void* workspace, long workspace_bytes) {
  CSAM_REQUIRE(q_f16 && workspace && B > 0, "csam_t2i_fused: bad args");
  CSAM_REQUIRE((X_f16 != nullptr) != (K0_f16 != nullptr), "csam_t2i_fused: give either keys+weights or hoisted K0/V0T");
  CSAM_REQUIRE(!X_f16 || (Wkv_f16 && kpe && bv), "csam_t2i_fused: Wkv/kpe/bv required with keys");
  CSAM_REQUIRE(!K0_f16 || V0T_f16, "csam_t2i_fused: V0T required with K0");
  if (workspace_bytes < csam_t2i_fused_workspace_bytes(B)) {
    csam_set_error("csam_t2i_fused: workspace too small");
    return CSAM_ERR_WORKSPACE;
  }
  T2iArgs a;
  a.X = (const half_t*)X_f16; a.Wkv = (const half_t*)Wkv_f16; a.kpe = kpe; a.bv = bv;
  a.K0 = (const half_t*)K0_f16; a.V0T = (const half_t*)V0T_f16; a.q = (const half_t*)q_f16;
  a.part = (float*)workspace;
  static csam_once_t attr_set;
  if (csam_first_call(attr_set)) {
    hipFuncSetAttribute((const void*)t2i_fused_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, T2I_SMEM);
    hipFuncSetAttribute((const void*)t2i_fused_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, T2I_SMEM);
  }
  dim3 grid(T2I_PARTS, B);
  if (X_f16)
    hipLaunchKernelGGL(t2i_fused_kernel<1>, grid, dim3(T2I_NW * 64), T2I_SMEM, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(t2i_fused_kernel<0>, grid, dim3(T2I_NW * 64), T2I_SMEM, (hipStream_t)stream, a);
  CSAM_LAUNCH_CHECK("csam_t2i_fused");
  if (!out_f16) return CSAM_OK;       // the caller merges the csam_t2i_fused_parts() partial records itself (csam_token_block_b)
  return csam_t2i_merge_launch(stream, (const float*)workspace, out_f16, B, T2I_PARTS);
}

extern "C" int csam_t2i_fused_parts(void) { return T2I_PARTS; }
