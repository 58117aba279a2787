// csam_flash_attn: global multi-head attention, flash style (no [T,T] score tensor), head_dim 64.
//
// Serves (a) SAM's 4 global ViTDet blocks (image_encoder.py:224-240 with window_size == 0) including
// the decomposed relative-position bias (:325-361), and (b) every DINOv2 ViT-L/14 block (external
// dependency; plain softmax(q k^T / sqrt(d)) v over 1 + 73*73 tokens).
//
// Workgroup = 4 waves x 32 queries = 128 queries of one head; key tiles of 64 go into a 3-deep LDS ring
// (K row-major, V transposed) by global_load_lds, one barrier per tile.  Swapped MFMA orientation: keys on the
// accumulator rows, so each lane owns ONE query column -> softmax state is in-lane, and the fp16 P registers feed
// P.V directly as the B operand.
//
// Scores are in base-2 units from the start (q carries scale * log2(e), folded into the qkv projection by the
// plans) and relative to a per-query reference exponent that rides in the MFMA accumulator's initial value, so a
// score costs one v_exp_f32 and half a v_cvt_pk: no running maximum (see the loop comment).
//
// Rel-pos bias, SAM global blocks (64x64 grid, key tile t == key row kh = t, kw = key & 63):
//     S2[q, (kh,kw)] = q2.k + log2(e) * (Th[q,kh] + Tw[q,kw])
// The accumulator is INITIALISED with Tw (16 registers per query tile, constant over all key tiles) + Th[q, t] (one
// scalar per lane per tile) - reference: one v_add per accumulator register, zero extra MFMA/LDS work.
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
__global__ __launch_bounds__(256) void transpose_v_kernel(const half_t* qkv, long ld, int v_off, half_t* vt, int T, int Tpad) {
  __shared__ half_t tile[64][66];
  const int t0 = blockIdx.x * 64, h = blockIdx.y;
  const int tid = threadIdx.x;
  qkv += (long)blockIdx.z * T * ld;                        // image blockIdx.z of an image-batched pass
  vt += (long)blockIdx.z * gridDim.y * 64 * Tpad;
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

// MFMA groups as inline asm with TIED accumulators.  The softmax reads every score with the VALU, so the accumulators
// of S = K Q^T want to live in VGPRs (in AGPRs every score costs a v_accvgpr_read and every seed a v_accvgpr_write:
// +32 % on this loop).  Rounds 1-2 got that from hipcc's experimental -amdgpu-mfma-vgpr-form (and hipcc selects the
// VGPR form by itself once the launch bounds leave <= 256 registers); in that form the register allocator un-ties vDst
// from SrcC and re-uses the SrcC quad as the destination of the next ds_read_b128, `s_nop 2` behind the MFMA that reads
// it.  While hunting the intermittent wrong bias described at the Tw seeds below that pattern was the first suspect: a
// single-wave probe cannot make it fail (tools/probe/mfma_srcc_lds_war.hip, 0 wrong in 35 k trials) and the bias fault
// turned out to be elsewhere, so it is UNPROVEN as a hazard -- but it rests on the matrix pipe having taken SrcC for
// all 64 lanes before an LDS return can land, under any contention, and nothing documents that.  The MFMAs of this
// kernel therefore do not depend on it: every accumulator is tied ("+v": vDst == SrcC, nothing else may be aimed at
// the quad while the MFMA is pending), the fragments are fetched by the same blocks (ring of four quads, counted
// lgkmcnt waits -- LDS returns in order, so counting inside a block only ever over-waits for the compiler's own LDS
// operations), no load is left in flight when a block ends (the compiler may copy a block's outputs), and the wait
// states the hazard recogniser would insert around MFMAs it cannot see are written out: VALU write -> MFMA read 2,
// 8-pass XDL write -> VALU read 8 + 3 (LLVM GCNHazardRecognizer, gfx940 rows).  Same speed as the compiler-scheduled
// loop (173 vs 177 us at T = 5330); tools/lint_mfma_srcc.py reports the pattern per source file at build time.
// The K fragments of the tile are fetched by the same asm block (four ds_read_b128 in flight, counted lgkmcnt
// waits -- LDS returns in order, so waits counted inside the block only ever over-wait for the compiler's own LDS
// operations): fetched by C++ they ended up in ONE register quad, read / wait / two MFMAs eight times over.
// s: the 8 accumulator quads [rt][kt]; q: the four query fragments [rt][ks]; ka0 / ka1: LDS byte address of this lane's
// fragment in k-step 0 / 1 at key tile row block 0 (kt adds 512 B per odd kt and 4096 B per kt >= 2).
__device__ __forceinline__ void scores_mfma(floatx4 (&s)[2][4], const half8_t (&q)[2][2], unsigned ka0, unsigned ka1) {
  half8_t k0, k1, k2, k3;            // ring of four: a quad is re-filled for k-step 1 right behind the MFMAs that read it
  asm volatile(
      "ds_read_b128 %8, %16\n\t"
      "ds_read_b128 %9, %16 offset:512\n\t"
      "ds_read_b128 %10, %16 offset:4096\n\t"
      "ds_read_b128 %11, %16 offset:4608\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %8, %12, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %4, %8, %14, %4\n\t"
      "ds_read_b128 %8, %17\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %9, %12, %1\n\t"
      "v_mfma_f32_16x16x32_f16 %5, %9, %14, %5\n\t"
      "ds_read_b128 %9, %17 offset:512\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %10, %12, %2\n\t"
      "v_mfma_f32_16x16x32_f16 %6, %10, %14, %6\n\t"
      "ds_read_b128 %10, %17 offset:4096\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_mfma_f32_16x16x32_f16 %3, %11, %12, %3\n\t"
      "v_mfma_f32_16x16x32_f16 %7, %11, %14, %7\n\t"
      "ds_read_b128 %11, %17 offset:4608\n\t"
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %8, %13, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %4, %8, %15, %4\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %9, %13, %1\n\t"
      "v_mfma_f32_16x16x32_f16 %5, %9, %15, %5\n\t"
      "s_waitcnt lgkmcnt(1)\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %10, %13, %2\n\t"
      "v_mfma_f32_16x16x32_f16 %6, %10, %15, %6\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_mfma_f32_16x16x32_f16 %3, %11, %13, %3\n\t"
      "v_mfma_f32_16x16x32_f16 %7, %11, %15, %7\n\t"
      "s_nop 7\n\t"
      "s_nop 3"
      : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][2]), "+v"(s[0][3]), "+v"(s[1][0]), "+v"(s[1][1]), "+v"(s[1][2]),
        "+v"(s[1][3]), "=&v"(k0), "=&v"(k1), "=&v"(k2), "=&v"(k3)
      : "v"(q[0][0]), "v"(q[0][1]), "v"(q[1][0]), "v"(q[1][1]), "v"(ka0), "v"(ka1)
      : "memory");
}

// Row sums lt^T = 1 . P^T (tied accumulators, zeroed by the caller); the result is tested by the VALU right away,
// hence the 11 wait states.  (No loads are left in flight when an asm block ends: the compiler is free to copy a
// block's outputs, and a copy of a register whose load has not returned copies garbage.)
__device__ __forceinline__ void rowsum_mfma(floatx4 (&lt)[2], const half8_t (&p)[2][2], const half8_t& ones) {
  asm volatile(
      "s_nop 1\n\t"                                   // VALU (cvt / zero) write -> MFMA read
      "v_mfma_f32_16x16x32_f16 %0, %2, %3, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %2, %5, %1\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %2, %4, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %2, %6, %1\n\t"
      "s_nop 7\n\t"
      "s_nop 3"
      : "+v"(lt[0]), "+v"(lt[1])
      : "v"(ones), "v"(p[0][0]), "v"(p[0][1]), "v"(p[1][0]), "v"(p[1][1]));
}

// O^T += V^T P^T: 16 tied MFMAs over the 8 V^T fragments (dims row dt*16 + fr, keys 32 st + 8 fg .. +7), ring of four
// quads in the order (dt0,st0) (dt1,st0) (dt0,st1) (dt1,st1), dt 2 / 3 re-filled in place.
__device__ __forceinline__ void pv_mfma(floatx4 (&o)[2][4], const half8_t (&p)[2][2], unsigned va0, unsigned va1) {
  half8_t v0, v1, v2, v3;
  asm volatile(
      "ds_read_b128 %8, %16\n\t"                       // (dt 0, st 0)
      "ds_read_b128 %9, %16 offset:2048\n\t"           // (dt 1, st 0)
      "ds_read_b128 %10, %17\n\t"                      // (dt 0, st 1)
      "ds_read_b128 %11, %17 offset:2048\n\t"          // (dt 1, st 1)
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %8, %12, %0\n\t"    // o[0][0] += v(0,0) p[0][0]
      "v_mfma_f32_16x16x32_f16 %4, %8, %14, %4\n\t"    // o[1][0] += v(0,0) p[1][0]
      "ds_read_b128 %8, %16 offset:4096\n\t"           // (dt 2, st 0)
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %9, %12, %1\n\t"    // o[0][1] += v(1,0) p[0][0]
      "v_mfma_f32_16x16x32_f16 %5, %9, %14, %5\n\t"
      "ds_read_b128 %9, %16 offset:6144\n\t"           // (dt 3, st 0)
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %10, %13, %0\n\t"   // o[0][0] += v(0,1) p[0][1]
      "v_mfma_f32_16x16x32_f16 %4, %10, %15, %4\n\t"
      "ds_read_b128 %10, %17 offset:4096\n\t"          // (dt 2, st 1)
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %11, %13, %1\n\t"   // o[0][1] += v(1,1) p[0][1]
      "v_mfma_f32_16x16x32_f16 %5, %11, %15, %5\n\t"
      "ds_read_b128 %11, %17 offset:6144\n\t"          // (dt 3, st 1)
      "s_waitcnt lgkmcnt(3)\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %8, %12, %2\n\t"    // o[0][2] += v(2,0) p[0][0]
      "v_mfma_f32_16x16x32_f16 %6, %8, %14, %6\n\t"
      "s_waitcnt lgkmcnt(2)\n\t"
      "v_mfma_f32_16x16x32_f16 %3, %9, %12, %3\n\t"    // o[0][3] += v(3,0) p[0][0]
      "v_mfma_f32_16x16x32_f16 %7, %9, %14, %7\n\t"
      "s_waitcnt lgkmcnt(1)\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %10, %13, %2\n\t"   // o[0][2] += v(2,1) p[0][1]
      "v_mfma_f32_16x16x32_f16 %6, %10, %15, %6\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_mfma_f32_16x16x32_f16 %3, %11, %13, %3\n\t"   // o[0][3] += v(3,1) p[0][1]
      "v_mfma_f32_16x16x32_f16 %7, %11, %15, %7"
      : "+v"(o[0][0]), "+v"(o[0][1]), "+v"(o[0][2]), "+v"(o[0][3]), "+v"(o[1][0]), "+v"(o[1][1]), "+v"(o[1][2]),
        "+v"(o[1][3]), "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3)
      : "v"(p[0][0]), "v"(p[0][1]), "v"(p[1][0]), "v"(p[1][1]), "v"(va0), "v"(va1)
      : "memory");
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
__global__ __launch_bounds__(256, BIAS ? 2 : 3) void flash_attn_kernel(const half_t* __restrict__ qkv, long ld,
                                                         int q_off, int k_off,
                                                         const half_t* __restrict__ vt, int Tpad,
                                                         const float* __restrict__ traw,
                                                         half_t* __restrict__ out, long ldo, int T,
                                                         float qmul, float bmul, int xcd_heads, int nH) {
  // Scores live in BASE-2 units from the start: the caller's q already carries scale * log2(e) (folded into the qkv
  // projection, qmul == 1) or is multiplied by qmul here; the bias tables are multiplied by bmul.
  // traw (BIAS): [nH][T][256] fp32 = q . [rel_pos_h (127 rows) | 0 | rel_pos_w (127 rows) | 0] from one
  // batched GEMM;  Th[q][kh] = traw[q][qh - kh + 63],  Tw[q][kw] = traw[q][128 + qw - kw + 63]
  // LDS: 3-deep ring x (K tile [64 keys][128 B] + V^T tile [64 dims][128 B]) = 48 KB, filled by global_load_lds;
  // tile t+2 is issued while tile t is consumed; every tile is retired (vmcnt(0)) one iteration before it is read
  __shared__ __attribute__((aligned(16))) char smem[FLASH_NS * 16384];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fg = lane >> 4;
  // XCD-aware order (nH % 8 == 0): workgroup b runs on XCD b % 8, and all query blocks of a head share its K / V^T
  // (1.4 MB at T = 5330), so head h is served by XCD h % 8 only -- two heads' keys per 4 MB L2 instead of all sixteen
  int head, qblk, img;
  if (xcd_heads) {
    const int b = blockIdx.x, x = b & 7, i = b >> 3, nqb = gridDim.x / (xcd_heads * 8);
    head = x + 8 * (i / nqb);
    qblk = i % nqb;
    img = blockIdx.y;
  } else {
    head = blockIdx.y;
    qblk = blockIdx.x;
    img = blockIdx.z;
  }
  // image-batched pass: image `img` owns rows img*T .. img*T+T-1 of qkv / out, its own V^T block and bias tables
  qkv += (long)img * T * ld;
  out += (long)img * T * ldo;
  vt += (long)img * nH * 64 * Tpad;
  if constexpr (BIAS) traw += (long)img * nH * T * 256;
  const int q0 = qblk * QPB + wave * QPW;
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
    for (int ks = 0; ks < 2; ++ks) {
      qf[rt][ks] = *(const half8_t*)(qp + (long)qc * ld + (ks * 4 + fg) * 8);
      if (qmul != 1.f) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[rt][ks][e] = (half_t)((float)qf[rt][ks][e] * qmul);
      }
    }
  }
  // Key permutation inside a 32-key step s: accumulator row (4g + r) of key tile kt holds key
  // 32s + 8g + r + 4*(kt&1), so that a lane's 8 P values of a step are 8 CONSECUTIVE keys and the V^T
  // fragment is one 16-B read.  Row this lane supplies as the A operand of S^T = K Q^T:
  const int krow_in_step = 8 * (fr >> 2) + (fr & 3);
  // LDS byte address of this lane's K fragment at kt = 0: row krow_in_step, 16-B slot (ks*4 + fg) ^ kswz(row); kswz sees
  // only bits 1, 3, 4 of the row, which kt (+4, +32 rows) does not touch, so kt is an immediate offset
  const unsigned kaddr0 = (unsigned)(unsigned long)(lptr_t)smem + krow_in_step * 128 + ((fg ^ kswz(krow_in_step)) << 4);
  // V^T fragment: dims row dt*16 + fr (dt: +2048 B), 16-B slot (st*4 + fg) ^ (row & 7), at +8192 B in the tile slot
  const unsigned vaddr0 = (unsigned)(unsigned long)(lptr_t)smem + 8192 + fr * 128 + ((fg ^ (fr & 7)) << 4);
  const unsigned vaddr1 = (unsigned)(unsigned long)(lptr_t)smem + 8192 + fr * 128 + (((4 + fg) ^ (fr & 7)) << 4);
  const unsigned kaddr1 = (unsigned)(unsigned long)(lptr_t)smem + krow_in_step * 128 + (((4 + fg) ^ kswz(krow_in_step)) << 4);
  floatx4 twr[2][4];
  const float* thp[2] = {nullptr, nullptr};
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
        for (int j = 0; j < 4; ++j) twr[rt][kt][j] = twq[-(kw0 + j)] * bmul;
      }
    }
  }
  if constexpr (BIAS) {
    // The scaled Tw quads are FINISHED here, in the prologue.  Left to the scheduler, hipcc sinks the 16 v_pk_mul_f32
    // (raw table value x bmul) into the first key tile, right behind the first s_barrier and right in front of the
    // v_pk_add_f32 that build the accumulator seeds -- and on MI355X that sequence intermittently loses: in 1-8 % of the
    // launches (depending on what else sits between the barrier and the seeds) ONE seed register of ONE wave came out
    // as c0 + 0 in lanes 48..63, i.e. its v_pk_add_f32 saw the Tw operand before the v_pk_mul_f32 ~15 instructions
    // earlier had written that lane group (register dump of a failing launch: tests/dbg/flash_dump.py; key tile 0,
    // second query tile, lane group 3 only; never with a zero Tw table; the raw table loads were complete, the
    // accumulators tied, the scaled register itself correct a moment later).  Round 2 saw the same signature and
    // mis-filed it as a register-recycling artefact.  With the products pinned before the loop, and the LDS-DMA issue
    // between the barrier and the seeds, 0 of 1600 launches differ (tests/test_determinism_gpu.py keeps watching).
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      asm volatile("" : "+v"(twr[rt][0]), "+v"(twr[rt][1]), "+v"(twr[rt][2]), "+v"(twr[rt][3]));
  }
  floatx4 o[2][4];
  float mref[2];              // softmax reference exponent of the query (base-2 units): P = 2^(s - mref)
  float l[2];                 // softmax denominators (every lane of a query holds the same sum)
  const half8_t ones8 = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    mref[rt] = 0.f;
    l[rt] = 0.f;
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
  float thn[2] = {0.f, 0.f};          // Th[q][t] of the coming tile (raw table units)
  if constexpr (BIAS) {
    thn[0] = thp[0][0];
    thn[1] = thp[1][0];
  }
  stage(0, 0);
  if (FLASH_NS == 3 && nt > 1) stage(1, 1);

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
    const unsigned kbase = cur * 16384;
    cur = cur + 1 == FLASH_NS ? 0 : cur + 1;

    floatx4 s[2][4];
    float thv[2] = {0.f, 0.f};
    if constexpr (BIAS) {
      // fetched an iteration ahead, i.e. BEFORE this iteration's tile loads in program order: the counted wait the
      // compiler puts in front of the first use then leaves the just-issued tile loads in flight
      thv[0] = thn[0] * bmul;
      thv[1] = thn[1] * bmul;
      if (t + 1 < nt) {
        thn[0] = thp[0][-(t + 1)];
        thn[1] = thp[1][-(t + 1)];
      }
    }
    // S^T - mref = K Q^T with the accumulator INITIALISED to (bias - mref): the reference exponent costs no VALU op
    auto scores = [&]() {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const float c0 = thv[rt] - mref[rt];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          if constexpr (BIAS) {
            s[rt][kt] = twr[rt][kt] + c0;
          } else {
            s[rt][kt] = floatx4{c0, c0, c0, c0};
          }
        }
      }
      // tied-accumulator MFMAs + their K fragment reads (see scores_mfma): VALU-initialised quads in, VALU-readable out
      scores_mfma(s, qf, kaddr0 + kbase, kaddr1 + kbase);
      // lane holds keys t*64 + (kt>>1)*32 + 8 fg + 4 (kt&1) + j of query fr; out-of-range keys only on the last tile
      if ((t + 1) * KT > T) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (t * KT + (kt >> 1) * 32 + 8 * fg + 4 * (kt & 1) + j >= T) s[rt][kt][j] = -INFINITY;
      }
    };
    // P = 2^(s - mref) in fp16 (the B operand of both P.V and the row sums) and this tile's row sums on the matrix
    // pipe: lt^T = 1 . P^T, every lane of a query receives the sum of the fp16 probabilities the PV product uses
    half8_t pf[2][2];
    floatx4 lt[2];
    auto probs = [&]() {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int j = 0; j < 4; ++j) s[rt][kt][j] = csam_exp2(s[rt][kt][j]);
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pf[rt][st][e] = (half_t)s[rt][2 * st][e];
            pf[rt][st][4 + e] = (half_t)s[rt][2 * st + 1][e];
          }
      }
      lt[0] = floatx4{0.f, 0.f, 0.f, 0.f};
      lt[1] = floatx4{0.f, 0.f, 0.f, 0.f};
      rowsum_mfma(lt, pf, ones8);
    };
    // ---- softmax.  The loop is issue-bound (VALU and MFMA times of a SIMD add up), so the per-score work is ONE
    // v_exp_f32 and half a v_cvt_pk: no running maximum is tracked.  mref is set from the first tile's maximum and
    // stays until some probability of the wave leaves fp16's range (> 65504, i.e. a row maximum 16 octaves above its
    // reference), which shows as a non-finite row sum; only then (and on tile 0) the classic online-softmax update
    // runs: scores recomputed, row maximum, mref += delta, O and l rescaled by 2^-delta.  Exact up to rounding: the
    // softmax is shift-invariant, and a reference BELOW the true maximum only adds headroom at the small end.
    scores();
    bool renorm = t == 0;
    if (!renorm) {
      probs();
      renorm = __ballot(!(lt[0][0] < 1e30f) || !(lt[1][0] < 1e30f)) != 0ull;
      if (renorm) scores();                       // rare: s was consumed by the exponentials
    }
    if (renorm) {
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
        const float delta = t == 0 ? mx : fmaxf(mx, 0.f);      // s is relative to the old reference
        mref[rt] += delta;
        if (t > 0) {
          const float a = csam_exp2(-delta);
          l[rt] *= a;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) o[rt][dt] *= a;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[rt][kt] -= delta;
      }
      probs();
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) l[rt] += lt[rt][0];
    // ---- O^T += V^T P^T (tied-accumulator asm, see pv_mfma)
    pv_mfma(o, pf, vaddr0 + kbase, vaddr1 + kbase);
  }
  // the epilogue reads O with the VALU: XDL write -> VALU read wait states behind the last tile's MFMAs
  asm volatile("s_nop 7\n\ts_nop 3" : "+v"(o[0][0]), "+v"(o[0][1]), "+v"(o[0][2]), "+v"(o[0][3]), "+v"(o[1][0]),
               "+v"(o[1][1]), "+v"(o[1][2]), "+v"(o[1][3]));

#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const float inv = 1.0f / l[rt];
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


// =====================================================================================================================
// head_dim 80 (ViT-H global blocks, round 3).  Same kernel, three k-steps: dims 0..63 as above (the K tile keeps its 128-B
// swizzled rows), dims 64..79 in a 2 KB EXTENSION tile of 32-B rows read as 8-byte fragments into a 16x16x16 MFMA -- no
// padded dimension -- and five 16-dim output tiles (V^T tile of 80 rows).  Slot = K 8 KB | Kx 2 KB | V^T 10 KB; three slots
// = 60 KB, two workgroups per CU.  q is scaled in the kernel (qmul), the rel-pos tables by bmul.
// =====================================================================================================================
#include "attn_flash80_asm.inc"

constexpr int F80_KX = 8192, F80_VT = 10240, F80_SLOT = 20480;

// vt[h][d][t] = qkv[t][v_off + h*80 + d], d < 80 (columns T..Tpad-1 stay zero)
__global__ __launch_bounds__(256) void transpose_v80_kernel(const half_t* __restrict__ qkv, long ld, int v_off,
                                                            half_t* __restrict__ vt, int T, int Tpad) {
  __shared__ half_t tile[80][66];
  const int t0 = blockIdx.x * 64, h = blockIdx.y;
  const int tid = threadIdx.x;
  for (int c = tid; c < 640; c += 256) {             // token c / 10, dims (c % 10) * 8 .. +7
    const int t = t0 + c / 10, ch = c % 10;
    half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
    if (t < T) v = *(const half8_t*)(qkv + (long)t * ld + v_off + h * 80 + ch * 8);
#pragma unroll
    for (int e = 0; e < 8; ++e) tile[ch * 8 + e][c / 10] = v[e];
  }
  __syncthreads();
  for (int c = tid; c < 640; c += 256) {             // dim row c >> 3, tokens (c & 7) * 8 .. +7
    const int d = c >> 3, tt = (c & 7) * 8;
    half8_t v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = tile[d][tt + e];
    *(half8_t*)(vt + ((long)h * 80 + d) * Tpad + t0 + tt) = v;
  }
}

template <bool BIAS>
__global__ __launch_bounds__(256, 2) void flash_attn80_kernel(const half_t* __restrict__ qkv, long ld, int q_off, int k_off,
                                                              const half_t* __restrict__ vt, int Tpad,
                                                              const float* __restrict__ traw, half_t* __restrict__ out,
                                                              long ldo, int T, float qmul, float bmul, int xcd_heads) {
  __shared__ __attribute__((aligned(16))) char smem[3 * F80_SLOT];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int fr = lane & 15, fg = lane >> 4;
  int head, qblk;
  if (xcd_heads) {
    const int b = blockIdx.x, x = b & 7, i = b >> 3, nqb = gridDim.x / (xcd_heads * 8);
    head = x + 8 * (i / nqb);
    qblk = i % nqb;
  } else {
    head = blockIdx.y;
    qblk = blockIdx.x;
  }
  const int q0 = qblk * QPB + wave * QPW;
  const half_t* qp = qkv + q_off + head * 80;
  const half_t* kp = qkv + k_off + head * 80;
  const half_t* vtp = vt + (long)head * 80 * Tpad;

  half8_t qf[2][2];
  half4_t qg[2];
  int qrow[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    qrow[rt] = q0 + rt * 16 + fr;
    const int qc = qrow[rt] < T ? qrow[rt] : T - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      qf[rt][ks] = *(const half8_t*)(qp + (long)qc * ld + (ks * 4 + fg) * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[rt][ks][e] = (half_t)((float)qf[rt][ks][e] * qmul);
    }
    qg[rt] = *(const half4_t*)(qp + (long)qc * ld + 64 + fg * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) qg[rt][e] = (half_t)((float)qg[rt][e] * qmul);
  }
  const int krow_in_step = 8 * (fr >> 2) + (fr & 3);
  const unsigned lds0 = (unsigned)(unsigned long)(lptr_t)smem;
  const unsigned kaddr0 = lds0 + krow_in_step * 128 + ((fg ^ kswz(krow_in_step)) << 4);
  const unsigned kaddr1 = lds0 + krow_in_step * 128 + (((4 + fg) ^ kswz(krow_in_step)) << 4);
  const unsigned kxaddr = lds0 + F80_KX + krow_in_step * 32 + fg * 8;
  const unsigned vaddr0 = lds0 + F80_VT + fr * 128 + ((fg ^ (fr & 7)) << 4);
  const unsigned vaddr1 = lds0 + F80_VT + fr * 128 + (((4 + fg) ^ (fr & 7)) << 4);
  floatx4 twr[2][4];
  const float* thp[2] = {nullptr, nullptr};
  if constexpr (BIAS) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const int qc = qrow[rt] < T ? qrow[rt] : T - 1;
      const float* tq = traw + ((long)head * T + qc) * 256;
      thp[rt] = tq + (qc >> 6) + 63;
      const float* twq = tq + 128 + (qc & 63) + 63;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
        const int kw0 = (kt >> 1) * 32 + 8 * fg + 4 * (kt & 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) twr[rt][kt][j] = twq[-(kw0 + j)] * bmul;
      }
    }
    // finished and pinned in the prologue: see flash_attn_kernel
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
      asm volatile("" : "+v"(twr[rt][0]), "+v"(twr[rt][1]), "+v"(twr[rt][2]), "+v"(twr[rt][3]));
  }
  floatx4 o[2][5];
  float mref[2], l[2];
  const half8_t ones8 = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    mref[rt] = 0.f;
    l[rt] = 0.f;
#pragma unroll
    for (int dt = 0; dt < 5; ++dt) o[rt][dt] = floatx4{0.f, 0.f, 0.f, 0.f};
  }

  const int nt = (T + KT - 1) / KT;
  auto stage = [&](int buf, int t) {
    char* slot = smem + buf * F80_SLOT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {                    // K rows (dims 0..63) and V^T rows 0..63: 512 pieces each
      const int c = tid + i * 256;
      const int row = c >> 3, sl = c & 7;
      int key = t * KT + row;
      key = key < T ? key : T - 1;
      glds16(kp + (long)key * ld + ((sl ^ kswz(row)) * 8), slot + (c & ~63) * 16);
      glds16(vtp + (long)row * Tpad + t * KT + ((sl ^ (row & 7)) * 8), slot + F80_VT + (c & ~63) * 16);
    }
    if (tid < 128) {                                 // waves 0, 1: V^T rows 64..79 and the K extension (dims 64..79)
      const int c = tid;
      const int row = 64 + (c >> 3), sl = c & 7;
      glds16(vtp + (long)row * Tpad + t * KT + ((sl ^ (row & 7)) * 8), slot + F80_VT + 8192 + (c & ~63) * 16);
      int key = t * KT + (c >> 1);
      key = key < T ? key : T - 1;
      glds16(kp + (long)key * ld + 64 + (c & 1) * 8, slot + F80_KX + (c & ~63) * 16);
    }
  };
  float thn[2] = {0.f, 0.f};
  if constexpr (BIAS) {
    thn[0] = thp[0][0];
    thn[1] = thp[1][0];
  }
  stage(0, 0);
  if (nt > 1) stage(1, 1);

  int cur = 0;
  for (int t = 0; t < nt; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (t + 2 < nt) stage(cur == 0 ? 2 : cur - 1, t + 2);
    const unsigned kbase = cur * F80_SLOT;
    cur = cur + 1 == 3 ? 0 : cur + 1;

    floatx4 s[2][4];
    float thv[2] = {0.f, 0.f};
    if constexpr (BIAS) {
      thv[0] = thn[0] * bmul;
      thv[1] = thn[1] * bmul;
      if (t + 1 < nt) {
        thn[0] = thp[0][-(t + 1)];
        thn[1] = thp[1][-(t + 1)];
      }
    }
    auto scores = [&]() {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const float c0 = thv[rt] - mref[rt];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          if constexpr (BIAS) {
            s[rt][kt] = twr[rt][kt] + c0;
          } else {
            s[rt][kt] = floatx4{c0, c0, c0, c0};
          }
        }
      }
      asm volatile("s_nop 1" : "+v"(s[0][0]), "+v"(s[0][1]), "+v"(s[0][2]), "+v"(s[0][3]), "+v"(s[1][0]), "+v"(s[1][1]),
                   "+v"(s[1][2]), "+v"(s[1][3]));      // VALU-written seeds -> MFMA SrcC
      scores_mfma80(s, qf, qg, kaddr0 + kbase, kaddr1 + kbase, kxaddr + kbase);
      if ((t + 1) * KT > T) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              if (t * KT + (kt >> 1) * 32 + 8 * fg + 4 * (kt & 1) + j >= T) s[rt][kt][j] = -INFINITY;
      }
    };
    half8_t pf[2][2];
    floatx4 lt[2];
    auto probs = [&]() {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int j = 0; j < 4; ++j) s[rt][kt][j] = csam_exp2(s[rt][kt][j]);
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            pf[rt][st][e] = (half_t)s[rt][2 * st][e];
            pf[rt][st][4 + e] = (half_t)s[rt][2 * st + 1][e];
          }
      }
      lt[0] = floatx4{0.f, 0.f, 0.f, 0.f};
      lt[1] = floatx4{0.f, 0.f, 0.f, 0.f};
      rowsum_mfma(lt, pf, ones8);
    };
    scores();
    bool renorm = t == 0;
    if (!renorm) {
      probs();
      renorm = __ballot(!(lt[0][0] < 1e30f) || !(lt[1][0] < 1e30f)) != 0ull;
      if (renorm) scores();
    }
    if (renorm) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int j = 0; j < 4; ++j) mx = fmaxf(mx, s[rt][kt][j]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float delta = t == 0 ? mx : fmaxf(mx, 0.f);
        mref[rt] += delta;
        if (t > 0) {
          const float a = csam_exp2(-delta);
          l[rt] *= a;
#pragma unroll
          for (int dt = 0; dt < 5; ++dt) o[rt][dt] *= a;
        }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) s[rt][kt] -= delta;
      }
      probs();
    }
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) l[rt] += lt[rt][0];
    pv_mfma80(o, pf, vaddr0 + kbase, vaddr1 + kbase);
  }
  asm volatile("s_nop 7\n\ts_nop 3" : "+v"(o[0][0]), "+v"(o[0][1]), "+v"(o[0][2]), "+v"(o[0][3]), "+v"(o[0][4]), "+v"(o[1][0]),
               "+v"(o[1][1]), "+v"(o[1][2]), "+v"(o[1][3]), "+v"(o[1][4]));
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const float inv = 1.0f / l[rt];
    if (qrow[rt] < T) {
#pragma unroll
      for (int dt = 0; dt < 5; ++dt) {
        half4_t h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (half_t)(o[rt][dt][j] * inv);
        *(half4_t*)(out + (long)qrow[rt] * ldo + head * 80 + dt * 16 + fg * 4) = h;
      }
    }
  }
}

}  // namespace

extern "C" long csam_flash_attn_workspace_bytes(int T, int nH) {
  const long Tpad = (long)((T + 63) / 64) * 64;
  return (long)nH * 64 * Tpad * 2;
}

extern "C" int csam_flash_attn_batched(void* stream, const void* qkv_f16, long ld, int q_off, int k_off, int v_off,
                                       const float* relpos_raw, void* out_f16, long ldo, int T, int nH, float scale,
                                       void* vt_workspace, long vt_workspace_bytes, int q_prescaled, int n_images) {
  CSAM_REQUIRE(qkv_f16 && out_f16 && vt_workspace && T > 0 && nH > 0, "csam_flash_attn: bad args");
  CSAM_REQUIRE(n_images >= 1 && n_images <= 64, "csam_flash_attn: n_images = %d", n_images);
  if (vt_workspace_bytes < n_images * csam_flash_attn_workspace_bytes(T, nH)) {
    csam_set_error("csam_flash_attn: V^T workspace too small (must also be zero-initialised once)");
    return CSAM_ERR_WORKSPACE;
  }
  const int Tpad = ((T + 63) / 64) * 64;
  CSAM_REQUIRE(ld % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && ldo % 4 == 0,
               "csam_flash_attn: alignment");
  CSAM_REQUIRE(!relpos_raw || T == 4096, "csam_flash_attn: rel-pos bias needs the 64x64 token grid");
  hipLaunchKernelGGL(transpose_v_kernel, dim3(Tpad / 64, nH, n_images), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)qkv_f16, ld, v_off, (half_t*)vt_workspace, T, Tpad);
  // XCD-aware order (nH % 8 == 0): head h is served by XCD h % 8 only; the image index rides in grid.y then
  const int xcd_heads = (nH % 8 == 0) ? nH / 8 : 0;
  dim3 grid(csam_cdiv(T, QPB), nH, n_images), block(256);
  if (xcd_heads) grid = dim3(csam_cdiv(T, QPB) * nH, n_images);
  // q_prescaled: the caller folded scale * log2(e) into the q rows of the qkv projection (and relpos_raw was
  // computed from that q, so the bias tables carry the same factor and only 1/scale brings them to base-2 units)
  const float log2e = 1.4426950408889634f;
  const float qmul = q_prescaled ? 1.f : scale * log2e;
  const float bmul = q_prescaled ? 1.f / scale : log2e;
  if (relpos_raw)
    hipLaunchKernelGGL(flash_attn_kernel<true>, grid, block, 0, (hipStream_t)stream, (const half_t*)qkv_f16, ld,
                       q_off, k_off, (const half_t*)vt_workspace, Tpad, relpos_raw, (half_t*)out_f16, ldo, T, qmul, bmul,
                       xcd_heads, nH);
  else
    hipLaunchKernelGGL(flash_attn_kernel<false>, grid, block, 0, (hipStream_t)stream, (const half_t*)qkv_f16, ld,
                       q_off, k_off, (const half_t*)vt_workspace, Tpad, relpos_raw, (half_t*)out_f16, ldo, T, qmul, bmul,
                       xcd_heads, nH);
  CSAM_LAUNCH_CHECK("csam_flash_attn");
  return CSAM_OK;
}

extern "C" int csam_flash_attn(void* stream, const void* qkv_f16, long ld, int q_off, int k_off, int v_off,
                               const float* relpos_raw, void* out_f16, long ldo, int T, int nH, float scale,
                               void* vt_workspace, long vt_workspace_bytes, int q_prescaled) {
  return csam_flash_attn_batched(stream, qkv_f16, ld, q_off, k_off, v_off, relpos_raw, out_f16, ldo, T, nH, scale, vt_workspace,
                                 vt_workspace_bytes, q_prescaled, 1);
}

// head_dim 80 form (ViT-H): qkv heads are 80 wide, V^T workspace nH x 80 x Tpad fp16 (zero-initialised once); q is scaled
// inside the kernel.  relpos_raw as for csam_flash_attn (T == 4096 only).
extern "C" long csam_flash_attn80_workspace_bytes(int T, int nH) {
  const long Tpad = (long)((T + 63) / 64) * 64;
  return (long)nH * 80 * Tpad * 2;
}

extern "C" int csam_flash_attn80(void* stream, const void* qkv_f16, long ld, int q_off, int k_off, int v_off,
                                 const float* relpos_raw, void* out_f16, long ldo, int T, int nH, float scale,
                                 void* vt_workspace, long vt_workspace_bytes) {
  CSAM_REQUIRE(qkv_f16 && out_f16 && vt_workspace && T > 0 && nH > 0, "csam_flash_attn80: bad args");
  if (vt_workspace_bytes < csam_flash_attn80_workspace_bytes(T, nH)) {
    csam_set_error("csam_flash_attn80: V^T workspace too small (must also be zero-initialised once)");
    return CSAM_ERR_WORKSPACE;
  }
  const int Tpad = ((T + 63) / 64) * 64;
  CSAM_REQUIRE(ld % 8 == 0 && q_off % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && ldo % 4 == 0,
               "csam_flash_attn80: alignment");
  CSAM_REQUIRE(!relpos_raw || T == 4096, "csam_flash_attn80: rel-pos bias needs the 64x64 token grid");
  hipLaunchKernelGGL(transpose_v80_kernel, dim3(Tpad / 64, nH), dim3(256), 0, (hipStream_t)stream, (const half_t*)qkv_f16, ld,
                     v_off, (half_t*)vt_workspace, T, Tpad);
  dim3 grid(csam_cdiv(T, QPB), nH), block(256);
  const int xcd_heads = (nH % 8 == 0) ? nH / 8 : 0;
  if (xcd_heads) grid = dim3(csam_cdiv(T, QPB) * nH);
  const float log2e = 1.4426950408889634f;
  if (relpos_raw)
    hipLaunchKernelGGL(flash_attn80_kernel<true>, grid, block, 0, (hipStream_t)stream, (const half_t*)qkv_f16, ld, q_off, k_off,
                       (const half_t*)vt_workspace, Tpad, relpos_raw, (half_t*)out_f16, ldo, T, scale * log2e, log2e, xcd_heads);
  else
    hipLaunchKernelGGL(flash_attn80_kernel<false>, grid, block, 0, (hipStream_t)stream, (const half_t*)qkv_f16, ld, q_off, k_off,
                       (const half_t*)vt_workspace, Tpad, relpos_raw, (half_t*)out_f16, ldo, T, scale * log2e, log2e, xcd_heads);
  CSAM_LAUNCH_CHECK("csam_flash_attn80");
  return CSAM_OK;
}
