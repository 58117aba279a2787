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
#include "csam_common.h"

namespace {

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
  int act;
  int M, N, K;
  long sA, sW, sC, sB;  // batch strides in elements (grid.z); sB: bias stride
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

// MI = M tiles of 16 rows per wave: 4 -> 128-row workgroup tile, 2 -> 64-row tile (twice the workgroups:
// used when the 128-row grid would leave CUs idle, e.g. N = 1024 projections at M = 4096)
template <int MI>
__global__ __launch_bounds__(256) void gemm_f16_kernel(GemmArgs p) {
  constexpr int TBM = MI * 32;   // rows of the workgroup tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  p.A += (long)blockIdx.z * p.sA;
  p.W += (long)blockIdx.z * p.sW;
  if (p.bias) p.bias += (long)blockIdx.z * p.sB;
  p.C = (p.c_dt == CSAM_DT_F32) ? (void*)((float*)p.C + (long)blockIdx.z * p.sC)
                                : (void*)((half_t*)p.C + (long)blockIdx.z * p.sC);
  // layout: [stage][A|W][128 rows][128 B]
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int bn0 = blockIdx.x * BN;
  const int bm0 = blockIdx.y * TBM;

  // ---- staging addresses: wave w, instr i covers tile rows (w*4+i)*8 .. +7
  const int srow = lane >> 3;            // row within the 8-row group
  const int sslot = lane & 7;            // 16-B slot in the 128-B LDS row
  const half_t* a_src[MI];
  const half_t* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = (wave * 4 + i) * 8 + srow;
    const int chunk = sslot ^ (row & 7);
    w_src[i] = p.W + (long)(bn0 + row) * p.ldw + chunk * 8;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    const int row = (wave * MI + i) * 8 + srow;
    const int chunk = sslot ^ (row & 7);
    int gm = bm0 + row;
    gm = gm < p.M ? gm : p.M - 1;        // clamp: rows past M are loaded but never stored
    a_src[i] = p.A + (long)gm * p.lda + chunk * 8;
  }
  auto stage = [&](int buf, int k0) {
    char* abase = smem + buf * 2 * TILE_BYTES;
    char* wbase = abase + TILE_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) glds16(w_src[i] + k0, wbase + (wave * 4 + i) * 1024);
#pragma unroll
    for (int i = 0; i < MI; ++i) glds16(a_src[i] + k0, abase + (wave * MI + i) * 1024);
  };

  floatx4 acc[MI][4];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = floatx4{0.f, 0.f, 0.f, 0.f};

  // ---- fragment read offsets (bytes within a tile), swizzled
  const int fr = lane & 15, fg = lane >> 4;
  int a_off[MI], w_off[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) w_off[i] = (wn * 64 + i * 16 + fr) * 128;
#pragma unroll
  for (int i = 0; i < MI; ++i) a_off[i] = (wm * MI * 16 + i * 16 + fr) * 128;
  const int sw = fr & 7;  // row&7 (tile-row offsets are multiples of 16)

  const int nk = p.K / BK;
  stage(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) stage(cur ^ 1, (kt + 1) * BK);
    const char* abase = smem + cur * 2 * TILE_BYTES;
    const char* wbase = abase + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int coff = ((kk * 4 + fg) ^ sw) << 4;
      half8_t af[MI], wf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) wf[i] = *(const half8_t*)(wbase + w_off[i] + coff);
#pragma unroll
      for (int i = 0; i < MI; ++i) af[i] = *(const half8_t*)(abase + a_off[i] + coff);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ni], af[mi], acc[mi][ni], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue.  Lane holds C[m = fr][n = fg*4 + j] of each 16x16 tile: stored straight from the
  // accumulators that is 8/16-byte pieces in 32/64-B segments (store-issue bound, measured).  Instead the
  // tile is staged in the (now free) operand LDS with an XOR-swizzled slot index and written / residual-added
  // as whole coalesced rows, 16 B per lane.
  const bool res_late = p.R && p.c_dt == CSAM_DT_F32 && p.r_dt == CSAM_DT_F32;   // residual added at copy-out
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int row = wm * MI * 16 + mi * 16 + fr;     // row inside the workgroup tile
    const int m = bm0 + row;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int col = wn * 64 + ni * 16 + fg * 4;    // column inside the 128-col tile
      const int n = bn0 + col;
      floatx4 v = acc[mi][ni];
      if (p.bias) v += *(const floatx4*)(p.bias + n);
      if (p.act != CSAM_ACT_NONE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = csam_apply_act(v[j], p.act);
      }
      if (p.colscale) v *= *(const floatx4*)(p.colscale + n);
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
      if (p.c_dt == CSAM_DT_F32) {                    // [128][512 B], 32 slots of 16 B, slot ^= row & 31
        const int slot = (col >> 2) ^ (row & 31);
        *(floatx4*)(smem + row * 512 + slot * 16) = v;
      } else {                                         // [128][256 B], 16 slots of 16 B, slot ^= row & 15
        half4_t h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (half_t)v[j];
        const int slot = (col >> 3) ^ (row & 15);
        *(half4_t*)(smem + row * 256 + slot * 16 + ((col >> 2) & 1) * 8) = h;
      }
    }
  }
  __syncthreads();
  if (p.c_dt == CSAM_DT_F32) {
#pragma unroll
    for (int it = 0; it < 4 * MI; ++it) {
      const int c = tid + it * 256;                    // TBM*32 16-B pieces: row c>>5, LDS slot c&31
      const int row = c >> 5, sl = c & 31;
      const int m = bm0 + row;
      if (m < p.M) {
        const int n = bn0 + ((sl ^ (row & 31)) << 2);
        floatx4 v = *(const floatx4*)(smem + c * 16);
        if (res_late) {
          const int mr = p.res_mod > 0 ? m % p.res_mod : m;
          v += *(const floatx4*)((const float*)p.R + (long)mr * p.ldr + n);
        }
        *(floatx4*)((float*)p.C + (long)m * p.ldc + n) = v;
      }
    }
  } else {
#pragma unroll
    for (int it = 0; it < 2 * MI; ++it) {
      const int c = tid + it * 256;                    // TBM*16 16-B pieces: row c>>4, LDS slot c&15
      const int row = c >> 4, sl = c & 15;
      const int m = bm0 + row;
      if (m < p.M) {
        const int n = bn0 + ((sl ^ (row & 15)) << 3);
        *(half8_t*)((half_t*)p.C + (long)m * p.ldc + n) = *(const half8_t*)(smem + c * 16);
      }
    }
  }
}

}  // namespace

static int gemm_launch(void* stream, const void* A, long lda, const void* W, long ldw, void* C, long ldc,
                       int c_dtype, const float* bias, const float* colscale, const void* residual,
                       long ldr, int r_dtype, int res_mod, int act, int M, int N, int K, int batch, long sA,
                       long sW, long sC, long sB = 0) {
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
  CSAM_REQUIRE(batch >= 1 && (batch == 1 || !residual), "csam_gemm_f16: batched call takes no residual");
  // 64-row tiles when 128-row tiles would leave CUs without a workgroup (measured: pays at <= 256 tiles,
  // loses at 336 because of the extra W re-reads and the uneven 2.6 workgroups/CU)
  const bool small = (long)(N / BN) * csam_cdiv(M, BM) * batch <= 256;
  dim3 grid(N / BN, csam_cdiv(M, small ? 64 : BM), batch);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute((const void*)gemm_f16_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    hipFuncSetAttribute((const void*)gemm_f16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TILE_BYTES);
    attr_set = true;
  }
  if (small)
    hipLaunchKernelGGL(gemm_f16_kernel<2>, grid, dim3(256), 4 * TILE_BYTES, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(gemm_f16_kernel<4>, grid, dim3(256), 4 * TILE_BYTES, (hipStream_t)stream, p);
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
