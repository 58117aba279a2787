// csam_gemm_f16: C[M,N] = epilogue(A[M,K] * W[N,K]^T)  -- fp16 operands, fp32 MFMA accumulate.
//
// Replaces every nn.Linear / 1x1 conv / im2col'd conv on the hot path of the reference
// (segment_anything_cs/modeling/image_encoder.py:227,238 qkv/proj; common.py:25-26 MLP;
// image_encoder.py:88-104 neck; transformer.py:228-254 decoder projections;
// mask_decoder.py:56-62 ConvTranspose2d as GEMM).  W is the PyTorch Linear layout [N,K],
// so both operands are K-contiguous ("B^T input").
//
// CDNA4 design (cdna_hip_programming.md §5, step-3 structure):
//   * 128x128x64 tile, 256 threads = 4 waves in a 2x2 grid, each wave owns 64x64 = 4x4 MFMA
//     16x16x32 f16 tiles (16 floatx4 accumulators).
//   * operands staged HBM->LDS by global_load_lds (16 B/lane, no VGPR round trip), double
//     buffered, one barrier per K tile.
//   * LDS image is lane-linear (glds constraint), so the bank-conflict XOR swizzle
//     (chunk ^= row&7 within a 128-B row) is applied to the per-lane SOURCE address and again
//     on the ds_read_b128 fragment reads (rule 21: both sides or neither).
//   * MFMA is issued "swapped" (W rows as the A operand, activation rows as B) so that each
//     lane ends up with 4 consecutive N columns of one output row -> 8/16-byte stores.
//   * epilogue fused: +bias[n], GELU(erf)/ReLU, *colscale[n] (DINOv2 LayerScale),
//     +residual[m,n] (f16 or f32), cast to f16 or f32.
#include <type_traits>
#include "csam_common.h"

namespace {

#define GEMM_ST(ptr, val) (*(ptr) = (val))
constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand per stage

struct GemmArgs {
  const half_t* A; long lda;
  const half_t* W; long ldw;
  void* C; long ldc; int c_dt;
  const float* bias;
  const float* colscale;
  const void* R; long ldr; int r_dt;
  int res_mod;          // >0: residual row = m % res_mod (per-image constant broadcast over prompts)
  int xcd;              // XCD-aware tile order on/off
  int act;
  int M, N, K;
  long sA, sW, sC, sB;  // batch strides in elements (grid.z); sB: bias stride
  // LayerNorm folded into the GEMMs around it (csam_gemm_f16_ln; common.py:38-43 between image_encoder.py:166-182):
  //   producer (fp32 residual output): also writes C16 = fp16(C) and, per row and 128-column tile, (sum, sum of squares)
  //   consumer: C = act(rstd[m] * (A W'^T - mean[m] * colsum[n]) + bias[n]), mean / rstd from the producer's partials
  half_t* C16; long ldc16;
  float* st_out;
  const float* st_in; int st_np; float eps;
  const float* colsum;
};

// mean and 1/std of a row from its st_np (sum, sum of squares) partials, summed in tile order (deterministic)
__device__ __forceinline__ float2_t ln_row_stats(const float* st, int np, float eps) {
  float s = 0.f, q = 0.f;
  for (int i = 0; i < np; ++i) {
    s += st[2 * i];
    q += st[2 * i + 1];
  }
  const float inv = 1.0f / (float)(np * 128);
  const float mean = s * inv;
  const float var = fmaxf(q * inv - mean * mean, 0.f);
  return float2_t{mean, rsqrtf(var + eps)};
}

// The folded LayerNorm + bias of one accumulator value, with the two FMAs spelled out: every tile shape must round the same way
// (an image's features are bit-identical whichever kernel its batch size selects), so the contraction is not left to the compiler.
__device__ __forceinline__ float ln_fold1(float acc, float mean, float rstd, float cs, float b) {
  return __builtin_fmaf(rstd, __builtin_fmaf(-mean, cs, acc), b);
}

// (sum, sum of squares) of one 4-column piece of a LayerNorm-producer row, operation by operation (same reason as ln_fold1: the
// 128-column kernel and gemm4w must round alike); the 32 pieces of a 128-column tile are then summed as a butterfly over the
// piece index with strides 16, 8, 4, 2, 1.
__device__ __forceinline__ void ln_piece_stats(const floatx4 v, float& sm, float& sq) {
  sm = __fadd_rn(__fadd_rn(v[0], v[1]), __fadd_rn(v[2], v[3]));
  sq = __fadd_rn(__builtin_fmaf(v[0], v[0], __fmul_rn(v[1], v[1])), __builtin_fmaf(v[2], v[2], __fmul_rn(v[3], v[3])));
}

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
// Tile configuration: WM x WN waves, each owning MI x NI MFMA 16x16 tiles; the workgroup tile is
// (WM*MI*16) x 128 (WN*NI == 8).  NS = depth of the operand ring in LDS.  Stage kt+NS-1 is issued while stage
// kt is consumed.  The wait before each K step is vmcnt(0): a counted wait that retires a stage which is then read
// in the same phase is not safe for other waves' LDS-DMA data (measured in decoder_fused.hip), so with NS = 3 a stage
// is retired one iteration before its first read and the stage issued after the barrier is the one in flight.
// Shipped configurations:
//   <4,4,2,2,2>  128-row tile, 4 waves, 2 x 32 KB: two workgroups per CU
//   <2,4,2,2,3>   64-row tile, 4 waves, 3 x 24 KB: two workgroups per CU; for grids that would leave CUs idle
//   <3,4,2,2,2>   96-row tile, 4 waves, 2 x 28 KB: two workgroups per CU; fills one round of 512 where 128 rows do not
// (Round 4 measured producer waves, deeper rings, 192-row tiles, counted waits, interleaved LDS-DMA issue and non-temporal
// stores on these shapes: all within +-3 % or slower, HISTORY.md 4.2f / profiles/r04_gemm_*.txt; the code is gone.)
#ifndef G128_READS_FIRST
#define G128_READS_FIRST 1
#endif
#ifndef G128_FULL_PATH        // developer A/B: 0 = the guarded fp32 copy-out for every tile
#define G128_FULL_PATH 1
#endif
#ifndef G128_HOIST            // developer A/B: 0 = bias / column sums / column scales loaded per tile in the epilogue
#define G128_HOIST 1
#endif
template <int MI, int NI, int WM, int WN, int NS, int KB>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f16_kernel(GemmArgs p) {
  constexpr int NW = WM * WN, NT = NW * 64;
  constexpr int NL = NW;                         // waves that load
  constexpr int TBM = WM * MI * 16;              // rows of the workgroup tile
  static_assert(WN * NI * 16 == BN, "tile is 128 columns wide");
  static_assert(KB == 64 || KB == 32, "K step");
  constexpr int PITCH = KB * 2;                  // bytes per LDS row
  constexpr int RPI = 1024 / PITCH;              // rows covered by one wave-wide global_load_lds (8 / 16)
  constexpr int SPR = PITCH / 16;                // 16-B slots per row (8 / 4)
  constexpr int A_BYTES = TBM * PITCH;
  constexpr int STAGE = A_BYTES + BN * PITCH;    // [A: TBM rows][W: 128 rows]
  constexpr int LA = TBM / RPI / NL, LW = BN / RPI / NL;   // global_load_lds per lane per stage
  static_assert(LA >= 1 && LW >= 1 && LA * NL * RPI == TBM && LW * NL * RPI == BN, "staging split");
  constexpr int L = LA + LW;
  // bank-conflict swizzle of the 16-B slot index, applied to the global SOURCE chunk and to the fragment reads:
  // 128-B rows: slot ^= row & 7;  64-B rows: slot ^= 3 * ((row >> 2) & 1)  (both conflict-free for the 16-lane
  // groups ds_read_b128 is serviced in, MI355X_MICROARCH.md LDS table)
  auto swz = [](int row) { return KB == 64 ? (row & 7) : 3 * ((row >> 2) & 1); };
  extern __shared__ __attribute__((aligned(16))) char smem[];
  p.A += (long)blockIdx.z * p.sA;
  p.W += (long)blockIdx.z * p.sW;
  if (p.bias) p.bias += (long)blockIdx.z * p.sB;
  p.C = (p.c_dt == CSAM_DT_F32) ? (void*)((float*)p.C + (long)blockIdx.z * p.sC)
                                : (void*)((half_t*)p.C + (long)blockIdx.z * p.sC);
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int lw = wave;
  const int wm = wave / WN, wn = wave % WN;
  // XCD-aware tile order: workgroup b runs on XCD b % 8, so give every XCD a contiguous run of row-major
  // tiles (N fastest): an XCD's L2 then sees few A row-tiles and A crosses the fabric about once instead of
  // 8 times.  Bijective for any tile count.
  const int gx = p.N / BN;
  const int ntiles = gridDim.x;
  int t = blockIdx.x;
  if (p.xcd) {
    const int b = blockIdx.x, xcd = b & 7, q = ntiles >> 3, r = ntiles & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int bn0 = (t % gx) * BN;
  const int bm0 = (t / gx) * TBM;

  // ---- staging addresses: wave w, instr i covers tile rows (w*LX+i)*RPI .. +RPI-1
  const int srow = lane / SPR;           // row within the group
  const int sslot = lane % SPR;          // 16-B slot in the LDS row
  const half_t* a_src[LA];
  const half_t* w_src[LW];
#pragma unroll
  for (int i = 0; i < LW; ++i) {
    const int row = (lw * LW + i) * RPI + srow;
    const int chunk = sslot ^ swz(row);
    w_src[i] = p.W + (long)(bn0 + row) * p.ldw + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < LA; ++i) {
    const int row = (lw * LA + i) * RPI + srow;
    const int chunk = sslot ^ swz(row);
    int gm = bm0 + row;
    gm = gm < p.M ? gm : p.M - 1;        // clamp: rows past M are loaded but never stored
    a_src[i] = p.A + (long)gm * p.lda + chunk * 8;
  }
  auto stage = [&](int buf, int k0) {
    char* abase = smem + buf * STAGE;
    char* wbase = abase + A_BYTES;
#pragma unroll
    for (int i = 0; i < LW; ++i) glds16(w_src[i] + k0, wbase + (lw * LW + i) * 1024);
#pragma unroll
    for (int i = 0; i < LA; ++i) glds16(a_src[i] + k0, abase + (lw * LA + i) * 1024);
  };

  floatx4 acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment read offsets (bytes within a stage), swizzled
  const int fr = lane & 15, fg = lane >> 4;
  int a_off[MI], w_off[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) w_off[i] = A_BYTES + (wn * NI * 16 + i * 16 + fr) * PITCH;
#pragma unroll
  for (int i = 0; i < MI; ++i) a_off[i] = (wm * MI * 16 + i * 16 + fr) * PITCH;
  const int sw = swz(fr);  // tile-row offsets are multiples of 16, which the swizzle ignores

  const int nk = p.K / KB;
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nk) stage(s, s * KB);

  // the wave's NI x 4 bias / column-sum / column-scale values, requested here, under the whole main loop (as gemm4w does).  In the
  // epilogue each was a load behind a uniform branch with `s_waitcnt vmcnt(0)` right after it, once per (mi, ni) tile: up to
  // 3 MI NI serialised L2 round trips with nothing to cover them (ISA of round 6).  Same values: bit-identical.
  // (HOIST: the four-wave workgroups only -- two of them share a CU and have 256 registers per wave; the eight-wave 256-row form
  // must stay inside 128 to keep two workgroups per CU and loads these values in its epilogue as before)
  constexpr bool HOIST = G128_HOIST && NW <= 4;
  floatx4 bzv[NI], csv[NI], scv[NI];
  if constexpr (HOIST) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int n = bn0 + wn * NI * 16 + ni * 16 + (lane >> 4) * 4;
      bzv[ni] = p.bias ? *(const floatx4*)(p.bias + n) : floatx4{0.f, 0.f, 0.f, 0.f};
      csv[ni] = p.st_in ? *(const floatx4*)(p.colsum + n) : floatx4{0.f, 0.f, 0.f, 0.f};
      scv[ni] = p.colscale ? *(const floatx4*)(p.colscale + n) : floatx4{1.f, 1.f, 1.f, 1.f};
    }
  }

  int cur = 0;                                         // kt % NS
  for (int kt = 0; kt < nk; ++kt) {
    // my loads of stage kt have landed; the barrier then covers everybody's, and also says every wave is
    // done reading the buffer of stage kt-1, which the next prefetch overwrites
    wait_vmcnt<0>();     // NS >= 3: this also retires the stages issued ahead, one iteration before they are read
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (kt + NS - 1 < nk) stage(cur == 0 ? NS - 1 : cur - 1, (kt + NS - 1) * KB);
    const char* base = smem + cur * STAGE;
#if G128_READS_FIRST
    // every fragment of the K tile requested before its first MFMA, kept there by a scheduling barrier.  The compiler's own
    // order waits `lgkmcnt(0)` three times per 64-wide tile (k-step 0's reads, a late straggler, k-step 1's reads issued only
    // after k-step 0's MFMAs; ISA of round 6); this form exposes one LDS round trip per tile.  Per accumulator the k-steps still
    // arrive in ascending order: bit-identical.
    half8_t af[KB / 32][MI], wf[KB / 32][NI];
#pragma unroll
    for (int kk = 0; kk < KB / 32; ++kk) {
      const int coff = ((kk * 4 + fg) ^ sw) << 4;
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[kk][i] = *(const half8_t*)(base + w_off[i] + coff);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[kk][i] = *(const half8_t*)(base + a_off[i] + coff);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < KB / 32; ++kk)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[kk][ni], af[kk][mi], acc[mi][ni], 0, 0, 0);
#else
#pragma unroll
    for (int kk = 0; kk < KB / 32; ++kk) {
      const int coff = ((kk * 4 + fg) ^ sw) << 4;
      half8_t af[MI], wf[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) wf[i] = *(const half8_t*)(base + w_off[i] + coff);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const half8_t*)(base + a_off[i] + coff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
#endif
    cur = cur + 1 == NS ? 0 : cur + 1;
  }
  __syncthreads();                                     // operand ring is free: reuse it for the output tile
  // tiles taller than 128 rows go out in passes of 128 rows (PR): the staged fp32 slab stays at 64 KB, two workgroups per CU
  constexpr int NPASS = TBM > 128 ? TBM / 128 : 1, PR = TBM / NPASS;
  float* stab = (float*)(smem + PR * 512);             // [TBM][2] mean, rstd (LayerNorm-consumer launches only)
  if (p.st_in) {
    for (int r = tid; r < TBM; r += NT) {
      const int m = min(bm0 + r, p.M - 1);
      const float2_t ms = ln_row_stats(p.st_in + (long)m * p.st_np * 2, p.st_np, p.eps);
      stab[2 * r] = ms[0];
      stab[2 * r + 1] = ms[1];
    }
    __syncthreads();
  }

  // ---- epilogue.  Lane holds C[m = fr][n = fg*4 + j] of each 16x16 tile: stored straight from the
  // accumulators that is 8/16-byte pieces in 32/64-B segments (store-issue bound, measured).  Instead the
  // tile is staged in the (now free) operand LDS with an XOR-swizzled slot index and written / residual-added
  // as whole coalesced rows, 16 B per lane.
  const bool res_late = p.R && p.c_dt == CSAM_DT_F32 && p.r_dt == CSAM_DT_F32;   // residual added at copy-out
#pragma unroll
  for (int pass = 0; pass < NPASS; ++pass) {
  if (NPASS == 1 || (wm * MI * 16) / PR == pass) {
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int trow = wm * MI * 16 + mi * 16 + fr;    // row inside the workgroup tile
    const int row = trow - pass * PR;                // row inside this pass's slab
    const int m = bm0 + trow;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int col = wn * NI * 16 + ni * 16 + fg * 4;    // column inside the 128-col tile
      const int n = bn0 + col;
      floatx4 v = acc[mi][ni];
      floatx4 bb;
      if constexpr (HOIST) bb = bzv[ni];
      else bb = p.bias ? *(const floatx4*)(p.bias + n) : floatx4{0.f, 0.f, 0.f, 0.f};
      if (p.st_in) {                                      // folded LayerNorm: rstd * (acc - mean * colsum) + bias
        const float mean = stab[2 * trow], rstd = stab[2 * trow + 1];
        floatx4 cs;
        if constexpr (HOIST) cs = csv[ni];
        else cs = *(const floatx4*)(p.colsum + n);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ln_fold1(v[j], mean, rstd, cs[j], bb[j]);
      } else {
        v += bb;
      }
      if (p.act == CSAM_ACT_GELU && p.c_dt == CSAM_DT_F16) {
        // fp16-bound output: packed polynomial GELU (13 instructions per pair).  With the erf form the activation
        // was ~40 % of the fc1 GEMM's cycles at K = 1024 (64 GELUs per lane against 512 MFMA cycles per K step).
        const float2_t g0 = csam_gelu_poly2((float2_t){v[0], v[1]}), g1 = csam_gelu_poly2((float2_t){v[2], v[3]});
        v = floatx4{g0[0], g0[1], g1[0], g1[1]};
      } else if (p.act != CSAM_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = csam_apply_act(v[j], p.act);
      }
      if (p.colscale) {
        if constexpr (HOIST) v *= scv[ni];
        else v *= *(const floatx4*)(p.colscale + n);
      }
      if (p.R && !res_late && m < p.M) {
        const int mr = p.res_mod > 0 ? m % p.res_mod : m;
        if (p.r_dt == CSAM_DT_F32) {
          v += *(const floatx4*)((const float*)p.R + (long)mr * p.ldr + n);
        } else {
          const half4_t r = *(const half4_t*)((const half_t*)p.R + (long)mr * p.ldr + n);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] += (float)r[j];
        }
      }
      if (p.c_dt == CSAM_DT_F32) {                    // [TBM][512 B], 32 slots of 16 B, slot ^= row & 31
        const int slot = (col >> 2) ^ (row & 31);
        *(floatx4*)(smem + row * 512 + slot * 16) = v;
      } else {                                         // [TBM][256 B], 16 slots of 16 B, slot ^= row & 15
        half4_t h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (half_t)v[j];
        const int slot = (col >> 3) ^ (row & 15);
        *(half4_t*)(smem + row * 256 + slot * 16 + ((col >> 2) & 1) * 8) = h;
      }
    }
  }
  }
  __syncthreads();
  if (p.c_dt == CSAM_DT_F32) {
    // FULLC: the whole tile lies inside M (workgroup-uniform) -- the copy-out as ONE straight line.  Behind the per-row exec-mask
    // guard the compiler can neither hoist the residual loads of later pieces nor count the memory operations in flight, so every
    // piece was load -> `vmcnt(0)` -> add -> store: PR * 32 / NT dependent memory round trips per thread (ISA of round 6).
    auto copy_out_f32 = [&](auto FULLC) {
    constexpr bool FULL = decltype(FULLC)::value;
    constexpr int NIT = PR * 32 / NT;
    floatx4 rres[NIT];
    if (FULL && res_late) {                            // all residual pieces requested before the first is needed
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int c = tid + it * NT, row = c >> 5, sl = c & 31;
        const int m = bm0 + pass * PR + row;
        const int mr = p.res_mod > 0 ? m % p.res_mod : m;
        rres[it] = *(const floatx4*)((const float*)p.R + (long)mr * p.ldr + bn0 + ((sl ^ (row & 31)) << 2));
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = tid + it * NT;                     // PR*32 16-B pieces: row c>>5, LDS slot c&31
      const int row = c >> 5, sl = c & 31;
      const int m = bm0 + pass * PR + row;
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (FULL) {
        const int n = bn0 + ((sl ^ (row & 31)) << 2);
        v = *(const floatx4*)(smem + c * 16);
        if (res_late) v += rres[it];
        GEMM_ST((floatx4*)((float*)p.C + (long)m * p.ldc + n), v);
        if (p.C16) {
          half4_t h;
#pragma unroll
          for (int j = 0; j < 4; ++j) h[j] = (half_t)v[j];
          GEMM_ST((half4_t*)(p.C16 + (long)m * p.ldc16 + n), h);
        }
      } else if (m < p.M) {
        const int n = bn0 + ((sl ^ (row & 31)) << 2);
        v = *(const floatx4*)(smem + c * 16);
        if (res_late) {
          const int mr = p.res_mod > 0 ? m % p.res_mod : m;
          v += *(const floatx4*)((const float*)p.R + (long)mr * p.ldr + n);
        }
        GEMM_ST((floatx4*)((float*)p.C + (long)m * p.ldc + n), v);
        if (p.C16) {                                   // the next projection's fp16 operand (the LayerNorm is folded into it)
          half4_t h;
#pragma unroll
          for (int j = 0; j < 4; ++j) h[j] = (half_t)v[j];
          GEMM_ST((half4_t*)(p.C16 + (long)m * p.ldc16 + n), h);
        }
      }
      if (p.st_out) {                                  // 32 consecutive lanes hold one row's 128 columns
        float sm, sq;
        ln_piece_stats(v, sm, sq);
#pragma unroll
        for (int o = 16; o > 0; o >>= 1) {
          sm += __shfl_xor(sm, o, 64);
          sq += __shfl_xor(sq, o, 64);
        }
        if ((c & 31) == 0 && (FULL || m < p.M)) {
          float* d = p.st_out + ((long)m * (p.N / BN) + bn0 / BN) * 2;
          d[0] = sm;
          d[1] = sq;
        }
      }
    }
    };
    if (G128_FULL_PATH && NW <= 4 && bm0 + TBM <= p.M) copy_out_f32(std::true_type{});
    else copy_out_f32(std::false_type{});
  } else {
#pragma unroll
    for (int it = 0; it < PR * 16 / NT; ++it) {
      const int c = tid + it * NT;                     // PR*16 16-B pieces: row c>>4, LDS slot c&15
      const int row = c >> 4, sl = c & 15;
      const int m = bm0 + pass * PR + row;
      if (m < p.M) {
        const int n = bn0 + ((sl ^ (row & 15)) << 3);
        GEMM_ST((half8_t*)((half_t*)p.C + (long)m * p.ldc + n), *(const half8_t*)(smem + c * 16));
      }
    }
  }
  if (pass + 1 < NPASS) __syncthreads();               // the slab is free for the next pass
  }
}

// ---------------------------------------------------------------------------------------------------------
// 256 x 256 tile, 8 waves in two PING-PONG groups (cdna_hip_programming.md "256^2 8-phase" idea, own schedule):
// for the wide fp16-output projections (qkv N = 3072, fc1 N = 4096) whose grid is <= one workgroup per CU, where
// nothing else is resident to hide a workgroup's barrier / LDS-latency bubbles.  Group 1 (waves 4-7, rows
// 128..255) runs one barrier behind group 0, so while one group issues its ds_reads + global_load_lds the other
// owns the matrix pipe:
//     tick 4s   : G0 loads {A rows 0-63 of its half, B}  of stage s     | G1 MFMAs of its previous phase
//     tick 4s+1 : G0 16 MFMAs                                           | G1 loads ...
// K step 32 per stage (64-B LDS rows, swizzle slot ^= 3*((row>>2)&1)), 4-slot ring of 32 KB stages, stage s+2 is
// issued ahead of its use (never into a slot the lagging group may still be reading) and retired by ONE counted
// vmcnt per stage a full phase before anybody reads it.  Per wave 128 x 64 outputs = 32 accumulator tiles.
// ---------------------------------------------------------------------------------------------------------
constexpr int G2_STAGE = 2 * 256 * 64;                  // A 256 rows + W 256 rows, 64 B each
constexpr int G2_SMEM = 4 * G2_STAGE;                   // 128 KB ring; the 256x256 fp16 output tile reuses it
constexpr int G2_SMEM_ALL = G2_SMEM + 2048;             // + the [256][2] LayerNorm table of csam_gemm_f16_ln

__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wr = wave >> 2, wc = wave & 3;
  const int fr = lane & 15, fg = lane >> 4;
  const int gx = p.N / 256;
  int t = blockIdx.x;
  if (p.xcd) {
    const int ntiles = gridDim.x, b = blockIdx.x, xcd = b & 7, q = ntiles >> 3, r = ntiles & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int bn0 = (t % gx) * 256, bm0 = (t / gx) * 256;

  // staging: a stage half (A or W) is 256 rows x 4 slots = 1024 16-B pieces, 2 per thread
  const half_t* a_src[2];
  const half_t* w_src[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = (wave * 2 + i) * 64 + lane, row = c >> 2, sl = c & 3;
    const int chunk = sl ^ (3 * ((row >> 2) & 1));
    int gm = bm0 + row;
    gm = gm < p.M ? gm : p.M - 1;
    a_src[i] = p.A + (long)gm * p.lda + chunk * 8;
    w_src[i] = p.W + (long)(bn0 + row) * p.ldw + chunk * 8;
  }
  auto stage_a = [&](int slot, int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(a_src[i] + k0, smem + slot * G2_STAGE + (wave * 2 + i) * 1024);
  };
  auto stage_w = [&](int slot, int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) glds16(w_src[i] + k0, smem + slot * G2_STAGE + 16384 + (wave * 2 + i) * 1024);
  };

  floatx4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};
  // the wave's 4 x 4 bias values (columns wc*64 + j*16 + fg*4 ..): fetched now, under the whole main loop, instead of by
  // 32 loads at the head of the epilogue where nothing covers their latency
  floatx4 bz[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    bz[j] = p.bias ? *(const floatx4*)(p.bias + bn0 + (wave & 3) * 64 + j * 16 + (lane >> 4) * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
  // folded LayerNorm (consumer): the column sums of the wave's 16 columns and -- for thread r < 256 -- the (sum, sum of
  // squares) partials of tile row r are fetched here too; mean / rstd go through a 2 KB table behind the output tile
  floatx4 cz[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    cz[j] = p.st_in ? *(const floatx4*)(p.colsum + bn0 + (wave & 3) * 64 + j * 16 + (lane >> 4) * 4) : floatx4{0.f, 0.f, 0.f, 0.f};
  floatx4 pr[5];                                                  // st_np <= 10 partials of (sum, sum of squares), st_np even
#pragma unroll
  for (int i = 0; i < 5; ++i) pr[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  if (p.st_in && tid < 256) {
    const floatx4* s4 = (const floatx4*)(p.st_in + (long)min(bm0 + tid, p.M - 1) * p.st_np * 2);
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (2 * i < p.st_np) pr[i] = s4[i];
  }
  const int coff = (fg ^ (3 * ((fr >> 2) & 1))) << 4;
  const int a_base = (wr * 128 + fr) * 64 + coff;                 // + i * 1024 per 16-row tile
  const int w_base = 16384 + (wc * 64 + fr) * 64 + coff;          // + j * 1024

  const int nst = p.K / 32;
  // Prefetch: A(s+3) is issued in phase b of stage s, W(s+3) in phase a of stage s+1 -- two barriers after the last
  // read of the slot they overwrite (stage s-1, lagging group's phase b).  The one counted wait per stage sits in
  // phase a and retires stage s+1 while stage s+2 stays in flight.
  stage_a(0, 0); stage_w(0, 0);
  stage_a(1, 32); stage_w(1, 32);
  if (nst > 2) stage_a(2, 64);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // prologue: everything issued so far has landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  if (wr == 1) {                                                  // group 1 runs one barrier behind
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  int slot = 0;
  for (int s = 0; s < nst; ++s) {
    const char* base = smem + slot * G2_STAGE;
    half8_t af[4], wf[4];
    // ---- phase a: rows 0..63 of the wave's half x all its 64 columns
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *(const half8_t*)(base + w_base + j * 1024);
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *(const half8_t*)(base + a_base + i * 1024);
    if (s + 2 < nst) stage_w((slot + 2) & 3, (s + 2) * 32);
    // retire stage s+1 HERE, a full phase (two barriers) before either group reads it: a counted wait followed by
    // a read in the same phase is not safe for the other waves' LDS-DMA data.  A(s+2) and W(s+2) may stay in flight.
    if (s + 2 < nst) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- phase b: rows 64..127, same W fragments
#pragma unroll
    for (int i = 0; i < 4; ++i) af[i] = *(const half8_t*)(base + a_base + (4 + i) * 1024);
    if (s + 3 < nst) stage_a((slot + 3) & 3, (s + 3) * 32);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], af[i], acc[4 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    slot = (slot + 1) & 3;
  }
  if (wr == 0) {                                                  // rejoin
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  __syncthreads();
  float* stab = (float*)(smem + G2_SMEM);                         // [256][2] mean, rstd
  if (p.st_in) {
    if (tid < 256) {                                              // same summation order as ln_row_stats
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        sm += pr[i][0];
        sq += pr[i][1];
        sm += pr[i][2];
        sq += pr[i][3];
      }
      const float inv = 1.0f / (float)(p.st_np * 128);
      const float mean = sm * inv;
      stab[2 * tid] = mean;
      stab[2 * tid + 1] = rsqrtf(fmaxf(sq * inv - mean * mean, 0.f) + p.eps);
    }
    __syncthreads();
  }

  // ---- epilogue (fp16 out, bias + activation): tile staged in LDS [256 rows][512 B], 32 slots of 16 B, slot ^= row & 31
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = wr * 128 + i * 16 + fr;
    const float mean = p.st_in ? stab[2 * row] : 0.f, rstd = p.st_in ? stab[2 * row + 1] : 1.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = wc * 64 + j * 16 + fg * 4;
      floatx4 v = acc[i][j];
      if (p.st_in) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = ln_fold1(v[e], mean, rstd, cz[j][e], bz[j][e]);
      } else {
        v += bz[j];
      }
      if (p.act == CSAM_ACT_GELU) {
        const float2_t g0 = csam_gelu_poly2((float2_t){v[0], v[1]}), g1 = csam_gelu_poly2((float2_t){v[2], v[3]});
        v = floatx4{g0[0], g0[1], g1[0], g1[1]};
      } else if (p.act == CSAM_ACT_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
      }
      half4_t h;
#pragma unroll
      for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
      const int sl = (col >> 3) ^ (row & 31);
      *(half4_t*)(smem + row * 512 + sl * 16 + ((col >> 2) & 1) * 8) = h;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 16; ++it) {
    const int c = tid + it * 512;                    // 256 rows x 32 slots
    const int row = c >> 5, sl = c & 31;
    const int m = bm0 + row;
    if (m < p.M) {
      const int n = bn0 + ((sl ^ (row & 31)) << 3);
      GEMM_ST((half8_t*)((half_t*)p.C + (long)m * p.ldc + n), *(const half8_t*)(smem + c * 16));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile on FOUR waves, one per SIMD, 128 x 128 outputs per wave in a[0:255] -- the main loop is generated,
// hand-scheduled assembly (tools/gen/gen_gemm4w_asm.py -> gemm4w_asm.inc: one barrier per 64-wide K tile, LDS reads and LDS-DMA
// threaded one per MFMA gap).  16 ds_read_b128 per 64 MFMAs where the ping-pong kernel's 128 x 64 wave tile needs 12 per 32.
// LDS: two stages of [A: 256 rows x 128 B][W: 256 rows x 128 B], slot ^= row & 7; the fp16 output tile reuses them.
// Same operands, same ascending chain of 32-wide MFMA steps per output element as the other kernels: bit-identical results.
// ---------------------------------------------------------------------------------------------------------
#ifndef G4_GROUP_M
#define G4_GROUP_M 4
#endif
#ifndef G4_F32_MIN_TILES       // smallest 256 x 256 grid the fp32-epilogue form of gemm4w takes over from the 128-column kernel
#define G4_F32_MIN_TILES 192
#endif
constexpr int G4_STAGE = 65536;
constexpr int G4_SMEM = 2 * G4_STAGE + 2048;            // + the [256][2] LayerNorm table

#ifndef G4_ASM_INC          // developer A/B: tools/debug/gemm4w_variants.sh builds ablated / re-placed main loops
#define G4_ASM_INC "gemm4w_asm.inc"
#endif
#include G4_ASM_INC

// Epilogue of one PAIR of accumulator tiles (I, 2 JP) and (I, 2 JP + 1), straight from the registers: after the LayerNorm fold /
// bias / activation each lane holds 4 consecutive columns (8 B of fp16) of row fr in both tiles; v_permlane16_swap exchanges the
// odd 16-lane rows of the first tile's registers with the even rows of the second's, after which lanes fg = 0, 2 hold 8 consecutive
// columns (16 B) of tile 2 JP and lanes fg = 1, 3 of tile 2 JP + 1: ONE 16-byte store per lane and pair, 64-B segments per row,
// no LDS round trip and no barrier (cdna_hip_programming.md T21, the 16-lane form).
template <bool LN, int ACT, int I, int JP>
__device__ __forceinline__ void gemm4w_store_pair(half_t* crow, bool ok, const floatx4 (&bz)[8], const floatx4 (&cz)[8],
                                                  float mean, float rstd) {
  floatx4 v[2] = {gemm4w_acc<I * 8 + 2 * JP>(), gemm4w_acc<I * 8 + 2 * JP + 1>()};
  unsigned h[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (LN) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[t][e] = ln_fold1(v[t][e], mean, rstd, cz[2 * JP + t][e], bz[2 * JP + t][e]);
    } else {
      v[t] += bz[2 * JP + t];
    }
  }
  if (ACT == CSAM_ACT_GELU) {
    float2_t g[4] = {{v[0][0], v[0][1]}, {v[0][2], v[0][3]}, {v[1][0], v[1][1]}, {v[1][2], v[1][3]}};
    csam_gelu_poly2_n<4>(g);
    v[0] = floatx4{g[0][0], g[0][1], g[1][0], g[1][1]};
    v[1] = floatx4{g[2][0], g[2][1], g[3][0], g[3][1]};
  } else if (ACT == CSAM_ACT_RELU) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) v[t][e] = v[t][e] > 0.f ? v[t][e] : 0.f;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const half2_t hh = {(half_t)v[t][2 * q], (half_t)v[t][2 * q + 1]};
      h[t][q] = __builtin_bit_cast(unsigned, hh);
    }
  const auto s0 = __builtin_amdgcn_permlane16_swap(h[0][0], h[1][0], false, false);
  const auto s1 = __builtin_amdgcn_permlane16_swap(h[0][1], h[1][1], false, false);
  typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
  if (ok) *(uintx4*)(crow + JP * 32) = uintx4{s0[0], s1[0], s0[1], s1[1]};
}

template <bool LN, int ACT, int I>
__device__ __forceinline__ void gemm4w_store_row(half_t* crow, bool ok, const floatx4 (&bz)[8], const floatx4 (&cz)[8],
                                                 float mean, float rstd) {
  gemm4w_store_pair<LN, ACT, I, 0>(crow, ok, bz, cz, mean, rstd);
  gemm4w_store_pair<LN, ACT, I, 1>(crow, ok, bz, cz, mean, rstd);
  gemm4w_store_pair<LN, ACT, I, 2>(crow, ok, bz, cz, mean, rstd);
  gemm4w_store_pair<LN, ACT, I, 3>(crow, ok, bz, cz, mean, rstd);
}

// fp32-output epilogue of one 16-row tile row (the residual projections: image_encoder.py:238 proj, common.py:26 lin2):
// C = (acc + bias) [* colscale] [+ residual], its fp16 copy and the LayerNorm partials of the next block's folded norm
// (csam_gemm_f16_ln producer), all straight from the registers: per lane 8 x 16-byte residual loads (issued one tile row
// ahead), 8 x 16-byte fp32 stores, 4 x 16-byte fp16 stores (permlane16-widened), and the 128-column statistics as the same
// butterfly the 128-column kernel runs over lanes -- strides 16, 8, 4 are register pairs here, 2 and 1 are lane ^ 32, lane ^ 16.
// G4_RES_AHEAD (round 6, untimed): residual rows requested this many tile rows ahead of their use.  One row ahead (rounds 5-6 as
// measured) leaves 8 x 16 B per lane = 32 KB per CU in flight and a tile row's arithmetic (~0.4 us) as the only cover for an HBM
// access under a 168 MB burst: every one of the eight rows waits.  After the main loop the fragment registers v[128:255] are
// free, so three rows fit.  A lane reads exactly the elements it later overwrites, so in-place C == R stays fine at any depth.
#ifndef G4_RES_AHEAD
#define G4_RES_AHEAD 3
#endif
#ifndef G4_FULL_PATH          // developer A/B: 0 = every tile through the guarded epilogues (rounds 5-6 as measured)
#define G4_FULL_PATH 1
#endif
// FULL: every row of the workgroup's tile is inside M (all tiles of SAM's passes, all but the last tile row of DINOv2's).  The
// guarded form puts each store behind an exec-mask branch, and across those joins the compiler cannot count the memory operations
// in flight: it waits `vmcnt(0)` -- every residual row requested ahead AND every store issued so far -- at the first residual use
// of a tile row (ISA of round 6: two such waits per tile even with rows requested ahead; eight with one row ahead).  The
// straight-line form gets counted waits.
// HASR: p.R != null, as a template parameter for the same reason (no join behind a uniform branch either).
template <int I, bool FULL, bool HASR>
__device__ __forceinline__ void gemm4w_f32_row(const GemmArgs& p, int m, int n0, int fg, const floatx4 (&bz)[8], const floatx4 (&cz)[8],
                                               floatx4 (&rr)[G4_RES_AHEAD][8]) {
  constexpr int RA = G4_RES_AHEAD;
  const bool ok = FULL || m < p.M, okn = FULL || m + 16 * RA < p.M;
  floatx4 rn[8];
  floatx4 (&rc)[8] = rr[I % RA];
  if (I + RA < 8 && HASR) {                             // residual of tile row I + RA (in-place C == R is fine: other rows)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      rn[j] = okn ? *(const floatx4*)((const float*)p.R + (long)(m + 16 * RA) * p.ldr + n0 + j * 16) : floatx4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);                 // the requests stay ahead of this row's arithmetic and stores
  }
  floatx4 v[8] = {gemm4w_acc<I * 8 + 0>(), gemm4w_acc<I * 8 + 1>(), gemm4w_acc<I * 8 + 2>(), gemm4w_acc<I * 8 + 3>(),
                  gemm4w_acc<I * 8 + 4>(), gemm4w_acc<I * 8 + 5>(), gemm4w_acc<I * 8 + 6>(), gemm4w_acc<I * 8 + 7>()};
  float sm[8], sq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    v[j] += bz[j];
    if (p.colscale) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] = __fmul_rn(v[j][e], cz[j][e]);
    }
    if (HASR) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[j][e] = __fadd_rn(v[j][e], rc[j][e]);
    }
    if (ok) *(floatx4*)((float*)p.C + (long)m * p.ldc + n0 + j * 16) = v[j];
    ln_piece_stats(v[j], sm[j], sq[j]);
  }
  if (p.C16) {
    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
    half_t* crow = p.C16 + (long)m * p.ldc16 + (n0 - fg * 4) + (fg & 1) * 16 + (fg >> 1) * 8;
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
      unsigned h[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const half2_t hh = {(half_t)v[2 * jp + t][2 * q], (half_t)v[2 * jp + t][2 * q + 1]};
          h[t][q] = __builtin_bit_cast(unsigned, hh);
        }
      const auto s0 = __builtin_amdgcn_permlane16_swap(h[0][0], h[1][0], false, false);
      const auto s1 = __builtin_amdgcn_permlane16_swap(h[0][1], h[1][1], false, false);
      if (ok) *(uintx4*)(crow + jp * 32) = uintx4{s0[0], s1[0], s0[1], s1[1]};
    }
  }
  if (p.st_out) {                                      // piece index of (tile j, lane group fg) = 4 j + fg
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sm[j] = __fadd_rn(sm[j], sm[j + 4]);
      sq[j] = __fadd_rn(sq[j], sq[j + 4]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      sm[j] = __fadd_rn(sm[j], sm[j + 2]);
      sq[j] = __fadd_rn(sq[j], sq[j + 2]);
    }
    float a = __fadd_rn(sm[0], sm[1]), b = __fadd_rn(sq[0], sq[1]);
    a = __fadd_rn(a, __shfl_xor(a, 32, 64));
    b = __fadd_rn(b, __shfl_xor(b, 32, 64));
    a = __fadd_rn(a, __shfl_xor(a, 16, 64));
    b = __fadd_rn(b, __shfl_xor(b, 16, 64));
    if (fg == 0 && ok) *(float2_t*)(p.st_out + ((long)m * (p.N / BN) + (n0 >> 7)) * 2) = float2_t{a, b};
  }
  if (I + RA < 8 && HASR) {
#pragma unroll
    for (int j = 0; j < 8; ++j) rc[j] = rn[j];
  }
}

// the whole fp32 epilogue of a wave's 128 x 128 outputs as ONE straight line per (FULL, HASR)
template <bool FULL, bool HASR>
__device__ __forceinline__ void gemm4w_f32_epilogue(const GemmArgs& p, int m0, int n0, int fg, const floatx4 (&bz)[8],
                                                    const floatx4 (&cz)[8]) {
  floatx4 rr[G4_RES_AHEAD][8];
#pragma unroll
  for (int i = 0; i < G4_RES_AHEAD; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j)
      rr[i][j] = (HASR && (FULL || m0 + 16 * i < p.M)) ? *(const floatx4*)((const float*)p.R + (long)(m0 + 16 * i) * p.ldr + n0 + j * 16)
                                                        : floatx4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_sched_barrier(0);
  gemm4w_f32_row<0, FULL, HASR>(p, m0, n0, fg, bz, cz, rr);
  gemm4w_f32_row<1, FULL, HASR>(p, m0 + 16, n0, fg, bz, cz, rr);
  gemm4w_f32_row<2, FULL, HASR>(p, m0 + 32, n0, fg, bz, cz, rr);
  gemm4w_f32_row<3, FULL, HASR>(p, m0 + 48, n0, fg, bz, cz, rr);
  gemm4w_f32_row<4, FULL, HASR>(p, m0 + 64, n0, fg, bz, cz, rr);
  gemm4w_f32_row<5, FULL, HASR>(p, m0 + 80, n0, fg, bz, cz, rr);
  gemm4w_f32_row<6, FULL, HASR>(p, m0 + 96, n0, fg, bz, cz, rr);
  gemm4w_f32_row<7, FULL, HASR>(p, m0 + 112, n0, fg, bz, cz, rr);
}

template <bool LN, int ACT, bool F32>
__global__ __launch_bounds__(256) void gemm4w_kernel(GemmArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const int gx = p.N / 256;
  int t = blockIdx.x;
  if (p.xcd) {
    const int ntiles = gridDim.x, b = blockIdx.x, xcd = b & 7, q = ntiles >> 3, r = ntiles & 7;
    t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  // ... and inside an XCD's run, tile rows are taken G4_GROUP_M at a time, column-major inside the group: the 32 tiles an XCD has
  // in flight then cover about 4 x 8 tiles (12 operand panels per K step through its L2) instead of 2 x 16 (18)
  const int gy = (p.M + 255) / 256;
  const int tpg = G4_GROUP_M * gx, grp = t / tpg, row0 = grp * G4_GROUP_M, gsz = min(G4_GROUP_M, gy - row0);
  const int bn0 = ((t % tpg) / gsz) * 256, bm0 = (row0 + (t % tpg) % gsz) * 256;

  // LDS-DMA pieces: 1 KB = 8 rows x 128 B; wave w issues pieces w*8 .. w*8+7 of the activation tile and of the weight tile.
  // Lane l of a piece lands in LDS row l >> 3, slot l & 7, and fetches global chunk (l & 7) ^ (row & 7) (the bank swizzle).
  unsigned oa[8], ow[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int row = (wave * 8 + i) * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ (row & 7);
    const int gm = min(bm0 + row, p.M - 1) - bm0;      // rows past M are loaded (clamped) but never stored
    oa[i] = (unsigned)(gm * (int)p.lda * 2 + chunk * 16);
    ow[i] = (unsigned)(row * (int)p.ldw * 2 + chunk * 16);
  }
  const half_t* pa = p.A + (long)bm0 * p.lda;
  const half_t* pw = p.W + (long)bn0 * p.ldw;
  const unsigned lds0 = (unsigned)(size_t)smem;
  const unsigned ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * 8192);
  unsigned ra[2][2], rw[2][2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const unsigned sl = (unsigned)(((ks * 4 + fg) ^ (fr & 7)) << 4);
      ra[s][ks] = lds0 + s * G4_STAGE + (wm * 128 + fr) * 128 + sl;
      rw[s][ks] = lds0 + s * G4_STAGE + 32768 + (wn * 128 + fr) * 128 + sl;
    }
  // epilogue operands, fetched under the main loop: the lane's 8 x 4 bias values / column sums, tile row tid's LayerNorm partials
  floatx4 bz[8], cz[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int n = bn0 + wn * 128 + j * 16 + fg * 4;
    bz[j] = p.bias ? *(const floatx4*)(p.bias + n) : floatx4{0.f, 0.f, 0.f, 0.f};
    cz[j] = LN ? *(const floatx4*)(p.colsum + n) : (F32 && p.colscale) ? *(const floatx4*)(p.colscale + n) : floatx4{0.f, 0.f, 0.f, 0.f};
  }
  floatx4 pr[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) pr[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  if (LN) {
    const floatx4* s4 = (const floatx4*)(p.st_in + (long)min(bm0 + tid, p.M - 1) * p.st_np * 2);
#pragma unroll
    for (int i = 0; i < 5; ++i)
      if (2 * i < p.st_np) pr[i] = s4[i];
  }

  gemm4w_mainloop(pa, pw, ldsw, oa, ow, ra, rw, (p.K / 64 - 2) / 2);

#ifdef G4_SKIP_EPILOGUE                                 // developer timing ablation (tools/debug/gemm4w_variants.sh)
  if (p.M > 0) return;
#endif
  if (F32) {
    const int m0 = bm0 + wm * 128 + fr, n0 = bn0 + wn * 128 + fg * 4;
    const bool full = G4_FULL_PATH && bm0 + 256 <= p.M;   // workgroup-uniform, as is p.R
    if (full) {
      if (p.R) gemm4w_f32_epilogue<true, true>(p, m0, n0, fg, bz, cz);
      else gemm4w_f32_epilogue<true, false>(p, m0, n0, fg, bz, cz);
    } else {
      if (p.R) gemm4w_f32_epilogue<false, true>(p, m0, n0, fg, bz, cz);
      else gemm4w_f32_epilogue<false, false>(p, m0, n0, fg, bz, cz);
    }
    return;
  }
  float2_t ms[8];                                      // (mean, rstd) of the lane's row in each of its 8 row tiles
#pragma unroll
  for (int i = 0; i < 8; ++i) ms[i] = float2_t{0.f, 1.f};
  if (LN) {                                            // [256][2] table behind the operand stages; same summation order as ln_row_stats
    float* stab = (float*)(smem + 2 * G4_STAGE);
    float sm = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      sm += pr[i][0];
      sq += pr[i][1];
      sm += pr[i][2];
      sq += pr[i][3];
    }
    const float inv = 1.0f / (float)(p.st_np * 128);
    const float mean = sm * inv;
    stab[2 * tid] = mean;
    stab[2 * tid + 1] = rsqrtf(fmaxf(sq * inv - mean * mean, 0.f) + p.eps);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i) ms[i] = *(const float2_t*)(stab + 2 * (wm * 128 + i * 16 + fr));
  }
  // lane (fr, fg) stores 16 B of row fr of each row tile at column pair-base + (fg & 1) * 16 + (fg >> 1) * 8 (see gemm4w_store_pair)
  const int m0 = bm0 + wm * 128 + fr;
  half_t* c0 = (half_t*)p.C + (long)m0 * p.ldc + bn0 + wn * 128 + (fg & 1) * 16 + (fg >> 1) * 8;
  const long rs = 16 * p.ldc;
  if (G4_FULL_PATH && bm0 + 256 <= p.M) {      // whole tile inside M (workgroup-uniform): the 64 stores of a wave as one straight line, no exec-mask branches
    gemm4w_store_row<LN, ACT, 0>(c0, true, bz, cz, ms[0][0], ms[0][1]);
    gemm4w_store_row<LN, ACT, 1>(c0 + rs, true, bz, cz, ms[1][0], ms[1][1]);
    gemm4w_store_row<LN, ACT, 2>(c0 + 2 * rs, true, bz, cz, ms[2][0], ms[2][1]);
    gemm4w_store_row<LN, ACT, 3>(c0 + 3 * rs, true, bz, cz, ms[3][0], ms[3][1]);
    gemm4w_store_row<LN, ACT, 4>(c0 + 4 * rs, true, bz, cz, ms[4][0], ms[4][1]);
    gemm4w_store_row<LN, ACT, 5>(c0 + 5 * rs, true, bz, cz, ms[5][0], ms[5][1]);
    gemm4w_store_row<LN, ACT, 6>(c0 + 6 * rs, true, bz, cz, ms[6][0], ms[6][1]);
    gemm4w_store_row<LN, ACT, 7>(c0 + 7 * rs, true, bz, cz, ms[7][0], ms[7][1]);
    return;
  }
  gemm4w_store_row<LN, ACT, 0>(c0, m0 < p.M, bz, cz, ms[0][0], ms[0][1]);
  gemm4w_store_row<LN, ACT, 1>(c0 + rs, m0 + 16 < p.M, bz, cz, ms[1][0], ms[1][1]);
  gemm4w_store_row<LN, ACT, 2>(c0 + 2 * rs, m0 + 32 < p.M, bz, cz, ms[2][0], ms[2][1]);
  gemm4w_store_row<LN, ACT, 3>(c0 + 3 * rs, m0 + 48 < p.M, bz, cz, ms[3][0], ms[3][1]);
  gemm4w_store_row<LN, ACT, 4>(c0 + 4 * rs, m0 + 64 < p.M, bz, cz, ms[4][0], ms[4][1]);
  gemm4w_store_row<LN, ACT, 5>(c0 + 5 * rs, m0 + 80 < p.M, bz, cz, ms[5][0], ms[5][1]);
  gemm4w_store_row<LN, ACT, 6>(c0 + 6 * rs, m0 + 96 < p.M, bz, cz, ms[6][0], ms[6][1]);
  gemm4w_store_row<LN, ACT, 7>(c0 + 7 * rs, m0 + 112 < p.M, bz, cz, ms[7][0], ms[7][1]);
}

}  // namespace

static int gemm_launch(void* stream, const void* A, long lda, const void* W, long ldw, void* C, long ldc,
                       int c_dtype, const float* bias, const float* colscale, const void* residual,
                       long ldr, int r_dtype, int res_mod, int act, int M, int N, int K, int batch, long sA,
                       long sW, long sC, long sB = 0, void* C16 = nullptr, long ldc16 = 0, float* st_out = nullptr,
                       const float* st_in = nullptr, int st_np = 0, float eps = 0.f, const float* colsum = nullptr) {
  CSAM_REQUIRE(A && W && C, "csam_gemm_f16: null operand");
  CSAM_REQUIRE(M > 0 && N > 0 && K > 0, "csam_gemm_f16: bad shape M=%d N=%d K=%d", M, N, K);
  CSAM_REQUIRE(N % BN == 0, "csam_gemm_f16: N=%d must be a multiple of %d", N, BN);
  CSAM_REQUIRE(K % BK == 0, "csam_gemm_f16: K=%d must be a multiple of %d", K, BK);
  CSAM_REQUIRE(lda % 8 == 0 && ldw % 8 == 0 && ldc % (c_dtype == CSAM_DT_F16 ? 8 : 4) == 0, "csam_gemm_f16: ld alignment");
  CSAM_REQUIRE(!residual || ldr % 4 == 0, "csam_gemm_f16: ldr alignment");
  CSAM_REQUIRE(c_dtype == CSAM_DT_F16 || c_dtype == CSAM_DT_F32, "csam_gemm_f16: c_dtype");
  GemmArgs p;
  p.A = (const half_t*)A; p.lda = lda;
  p.W = (const half_t*)W; p.ldw = ldw;
  p.C = C; p.ldc = ldc; p.c_dt = c_dtype;
  p.bias = bias; p.colscale = colscale;
  p.R = residual; p.ldr = ldr; p.r_dt = r_dtype;
  p.res_mod = res_mod;
  p.act = act; p.M = M; p.N = N; p.K = K;
  p.sA = sA; p.sW = sW; p.sC = sC; p.sB = sB;
  p.C16 = (half_t*)C16; p.ldc16 = ldc16; p.st_out = st_out; p.st_in = st_in; p.st_np = st_np; p.eps = eps; p.colsum = colsum;
  CSAM_REQUIRE(!(C16 || st_out) || (c_dtype == CSAM_DT_F32 && batch == 1 && (!C16 || ldc16 % 4 == 0)),
               "csam_gemm_f16_ln: the fp16 copy / row statistics come with the fp32 output only");
  CSAM_REQUIRE(!st_in || (colsum && st_np > 0 && st_np * BN == K && batch == 1),
               "csam_gemm_f16_ln: row statistics must cover K = %d features in 128-column partials (got %d)", K, st_np);
  CSAM_REQUIRE(batch >= 1 && (batch == 1 || !residual), "csam_gemm_f16: batched call takes no residual");
  // Tile height of the 128-column kernel.  64 rows when 128-row tiles would leave CUs without a workgroup (measured: pays at
  // <= 256 tiles, loses at 336 because of the extra W re-reads and the uneven 2.6 workgroups/CU); 96 rows when 128-row tiles
  // would fill only part of ONE round of the 512 resident workgroups and 96-row tiles still fit that round (DINOv2 proj / fc2
  // at one image: M = 5330, N = 1024: 336 -> 448 workgroups); 128 rows otherwise (every image-batched shape)
  const long t128 = (long)(N / BN) * csam_cdiv(M, BM) * batch, t96 = (long)(N / BN) * csam_cdiv(M, 96) * batch;
  const bool small = t128 <= 256;
  const bool mid = !small && t128 < 512 && t96 <= 512;
  p.xcd = 1;
  // Wide fp16-output projections (qkv, fc1): the 256 x 256 ping-pong kernel, one workgroup per CU, when its grid is at most one
  // round of the chip or fills its last round to >= 85 % (image-batched encoders: M = B x 4096 / B x 5330 rows; one image of
  // DINOv2's fc1 -- 336 tiles, 1.3 rounds -- is cut into 4096 + 1234 rows by the plan instead)
  static csam_once_t set256;
  if (csam_first_call(set256))
    hipFuncSetAttribute((const void*)gemm256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, G2_SMEM_ALL);
  const long t256 = (long)(N / 256) * csam_cdiv(M, 256);
  const bool fills = t256 <= 256 || 100 * t256 >= 85 * 256 * csam_cdiv(t256, 256);
  // The residual projections (fp32 stream out, fp32 residual in, optionally the fp16 copy + LayerNorm partials): the four-wave kernel
  // with the fp32 epilogue when its grid is about one round of the chip or more
  if (batch == 1 && c_dtype == CSAM_DT_F32 && (!residual || r_dtype == CSAM_DT_F32) && res_mod == 0 && act == CSAM_ACT_NONE && !st_in &&
      N % 256 == 0 && K % 128 == 0 && K >= 256 && ldc % 4 == 0 && (!residual || ldr % 4 == 0) && t256 >= G4_F32_MIN_TILES) {
    static csam_once_t set4f;
    auto k4 = gemm4w_kernel<false, CSAM_ACT_NONE, true>;
    if (csam_first_call(set4f)) hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, G4_SMEM);
    p.xcd = 1;
    hipLaunchKernelGGL(k4, dim3((unsigned)t256), dim3(256), G4_SMEM, (hipStream_t)stream, p);
    CSAM_LAUNCH_CHECK("csam_gemm_f16");
    return CSAM_OK;
  }
  if (batch == 1 && c_dtype == CSAM_DT_F16 && !residual && !colscale && N % 256 == 0 && N >= 2048 && fills && K >= 64 &&
      (!st_in || (st_np % 2 == 0 && st_np <= 10))) {
    dim3 g256((unsigned)t256);
    if (K % 128 == 0 && K >= 256) {                    // the hand-scheduled four-wave kernel (two K tiles per loop trip)
#define CSAM_GEMM4W(LN_, ACT_)                                                                                   \
  {                                                                                                              \
    static csam_once_t set4w;                                                                                    \
    auto k4 = gemm4w_kernel<LN_, ACT_, false>;                                                                          \
    if (csam_first_call(set4w)) hipFuncSetAttribute((const void*)k4, hipFuncAttributeMaxDynamicSharedMemorySize, G4_SMEM); \
    hipLaunchKernelGGL(k4, g256, dim3(256), G4_SMEM, (hipStream_t)stream, p);                                    \
  }
      if (st_in) {
        if (act == CSAM_ACT_GELU) CSAM_GEMM4W(true, CSAM_ACT_GELU)
        else if (act == CSAM_ACT_RELU) CSAM_GEMM4W(true, CSAM_ACT_RELU)
        else CSAM_GEMM4W(true, CSAM_ACT_NONE)
      } else {
        if (act == CSAM_ACT_GELU) CSAM_GEMM4W(false, CSAM_ACT_GELU)
        else if (act == CSAM_ACT_RELU) CSAM_GEMM4W(false, CSAM_ACT_RELU)
        else CSAM_GEMM4W(false, CSAM_ACT_NONE)
      }
    } else {
      hipLaunchKernelGGL(gemm256_kernel, g256, dim3(512), G2_SMEM_ALL, (hipStream_t)stream, p);
    }
    CSAM_LAUNCH_CHECK("csam_gemm_f16");
    return CSAM_OK;
  }
  dim3 grid((N / BN) * csam_cdiv(M, mid ? 96 : small ? 64 : BM), 1, batch);
#define CSAM_GEMM_LAUNCH(MI_, NI_, WM_, WN_, NS_, KB_)                                                         \
  {                                                                                                            \
    constexpr int SM_RING = NS_ * (WM_ * MI_ * 16 + BN) * KB_ * 2;                                             \
    constexpr int SM_OUT = (WM_ * MI_ * 16 > 128 ? 128 : WM_ * MI_ * 16) * 512;                                \
    constexpr int SM = (SM_RING > SM_OUT + 2048 ? SM_RING : SM_OUT + 2048);  /* + [TBM][2] LayerNorm table */  \
    static csam_once_t set;                                                                                    \
    auto kern = gemm_f16_kernel<MI_, NI_, WM_, WN_, NS_, KB_>;                                                 \
    if (csam_first_call(set))                                                                                  \
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SM);                  \
    hipLaunchKernelGGL(kern, grid, dim3(WM_ * WN_ * 64), SM, (hipStream_t)stream, p);                          \
  }
  // 256-row tiles (8 waves, 3 x 24 KB ring, the fp32 slab staged in two passes of 128 rows: still two workgroups per CU) when they
  // fill whole rounds of the 512 resident workgroups: a quarter fewer operand bytes per flop and half the tile prologues.  SAM at
  // four images per pass (M = 16384, N = 1024: 512 tiles): proj 75.5 -> 68.1 us, fc2 171 -> 163; DINOv2's M = 21320 (672 tiles = 1.3
  // rounds) loses 8-17 % to the part-filled round and stays on 128 rows (profiles/r05_gemm_tall_tile.txt)
  const long t256r = (long)(N / BN) * csam_cdiv(M, 256);
  const bool tall = !small && !mid && batch == 1 && t256r >= 512 && 100 * t256r >= 90 * 512 * csam_cdiv(t256r, 512);
  if (tall) grid = dim3((N / BN) * csam_cdiv(M, 256), 1, batch);
  if (tall) CSAM_GEMM_LAUNCH(4, 4, 4, 2, 3, 32)
  else if (mid) CSAM_GEMM_LAUNCH(3, 4, 2, 2, 2, 64)
  else if (small) CSAM_GEMM_LAUNCH(2, 4, 2, 2, 3, 64)
  else CSAM_GEMM_LAUNCH(4, 4, 2, 2, 2, 64)
  CSAM_LAUNCH_CHECK("csam_gemm_f16");
  return CSAM_OK;
}

extern "C" int csam_gemm_f16(void* stream, const void* A, long lda, const void* W, long ldw,
                             void* C, long ldc, int c_dtype, const float* bias,
                             const float* colscale, const void* residual, long ldr, int r_dtype,
                             int act, int M, int N, int K) {
  return gemm_launch(stream, A, lda, W, ldw, C, ldc, c_dtype, bias, colscale, residual, ldr, r_dtype, 0, act,
                     M, N, K, 1, 0, 0, 0);
}

// LayerNorm folded into the two GEMMs around it (include/csam.h).
extern "C" int csam_gemm_f16_ln(void* stream, const void* A, long lda, const void* W, long ldw, void* C, long ldc, int c_dtype,
                                const float* bias, const float* colscale, const void* residual, long ldr, int r_dtype,
                                int act, int M, int N, int K, void* C16_out, long ldc16, float* rowstats_out,
                                const float* rowstats_in, int n_partials, float eps, const float* colsum) {
  return gemm_launch(stream, A, lda, W, ldw, C, ldc, c_dtype, bias, colscale, residual, ldr, r_dtype, 0, act, M, N, K, 1,
                     0, 0, 0, 0, C16_out, ldc16, rowstats_out, rowstats_in, n_partials, eps, colsum);
}

// residual row index taken modulo res_mod: adds a per-image [res_mod, N] constant to every prompt's
// [res_mod, N] slab of a prompt-stacked M = B*res_mod GEMM (hoisted key_pe projections).
extern "C" int csam_gemm_f16_resmod(void* stream, const void* A, long lda, const void* W, long ldw,
                                    void* C, long ldc, int c_dtype, const float* bias,
                                    const void* residual, long ldr, int r_dtype, int res_mod, int act,
                                    int M, int N, int K) {
  CSAM_REQUIRE(res_mod > 0, "csam_gemm_f16_resmod: res_mod must be > 0");
  return gemm_launch(stream, A, lda, W, ldw, C, ldc, c_dtype, bias, nullptr, residual, ldr, r_dtype, res_mod,
                     act, M, N, K, 1, 0, 0, 0);
}

// batch of independent GEMMs (grid.z) with element strides; no residual.
extern "C" int csam_gemm_f16_batched(void* stream, const void* A, long lda, long strideA, const void* W,
                                     long ldw, long strideW, void* C, long ldc, long strideC, int c_dtype,
                                     const float* bias, long strideBias, int act, int M, int N, int K,
                                     int batch) {
  return gemm_launch(stream, A, lda, W, ldw, C, ldc, c_dtype, bias, nullptr, nullptr, 0, 0, 0, act, M, N, K,
                     batch, strideA, strideW, strideC, strideBias);
}
