// PWD-Net selection + fused mask post-processing + EPS occupancy lookup.
//
// Reference: crowdsam/model.py:351-358 (score fusion, select_mask 'max_iou'), sam.py:153-161
// (postprocess_masks), crowdsam/model.py:371-389 (filters), amg.py:156-176 (stability score),
// amg.py:303-346 (batched_mask_to_box), crowdsam/model.py:238-246 (occupancy pruning).
//
// The reference up-samples ALL four candidate masks to (B,4,H,W) fp32 twice (50 MB/prompt of HBM
// traffic) and then discards three.  Here only the selected candidate's 256x256 logits are read
// (256 KB/prompt, L2-resident) and each output pixel is produced once: bilinear x4 -> thresholds
// -> u8 mask byte + wave-reduced stability counts and bbox extents (integer atomics, deterministic).
#include "csam_common.h"
#include <type_traits>

namespace {

// ---------------------------------------------------------------------------------------------
// score fusion + argmax (crowdsam/model.py:351,325,354): s_l = clamp(iou_l,0) * sigmoid(cls_l0);
// sel = first argmax_l s_l; category = first argmax_c cls[sel,c].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_kernel(const float* __restrict__ iou, const float* __restrict__ cls,
                                                     int C, int* __restrict__ sel, float* __restrict__ score,
                                                     int* __restrict__ category, float* __restrict__ fused, int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float best = -INFINITY;
  int bi = 0;
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    const float i = fmaxf(iou[b * 4 + l], 0.f);
    const float c = cls[((long)b * 4 + l) * C];
    const float s = i * (1.0f / (1.0f + expf(-c)));
    if (fused) fused[b * 4 + l] = s;
    if (s > best) {
      best = s;
      bi = l;
    }
  }
  sel[b] = bi;
  score[b] = best;
  int cat = 0;
  float cb = cls[((long)b * 4 + bi) * C];
  for (int c = 1; c < C; ++c) {
    const float v = cls[((long)b * 4 + bi) * C + c];
    if (v > cb) {
      cb = v;
      cat = c;
    }
  }
  category[b] = cat;
}

// ---------------------------------------------------------------------------------------------
// bilinear sample, torch upsample_bilinear2d(align_corners=False) semantics, fp32, no contraction
// ---------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ __forceinline__ void src_index(float scale, int dst, int in_size, int& i0, int& i1, float& lam) {
  float s = scale * ((float)dst + 0.5f) - 0.5f;
  s = s < 0.f ? 0.f : s;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  lam = s - (float)i0;
}

__device__ __forceinline__ float bilerp(const float* __restrict__ p, int sw, int y0, int y1, float ly, int x0,
                                        int x1, float lx) {
  const float w0x = 1.f - lx, w0y = 1.f - ly;
  const float top = w0x * p[y0 * sw + x0] + lx * p[y0 * sw + x1];
  const float bot = w0x * p[y1 * sw + x0] + lx * p[y1 * sw + x1];
  return w0y * top + ly * bot;
}

// Output geometry: (H, W).  Source: per-prompt plane [sh, sw] at src + b*src_bstride + sel*plane
// (sel == nullptr -> plane 0).  scale_y/x are the torch "in/out" ratios of THIS interpolate call:
// fast path (original_size == input_size): source = low-res logits, scale = 256/1024 (crop is a
// no-op on coordinates); general path stage 1 writes fp32 (mode 0), stage 2 thresholds (mode 1).
struct PostArgs {
  const float* src; long src_bstride; int plane; const int* sel;
  int sh, sw; float scale_y, scale_x;
  int H, W;
  float thr, off;
  float* out_f32;       // mode 0: [B,H,W] fp32 logits
  uint8_t* out_mask;    // mode 1: [B,H,W] u8
  int* inter; int* uni; int* box;   // mode 1: [B], [B], [B,4] = xmin,ymin,xmax,ymax (pre-initialised)
  const uint8_t* keep;  // optional [B]: prompts with keep[b]==0 are skipped entirely
  const int* slot;      // optional [B]: mask bytes of prompt b go to out_mask[slot[b]] (compacted store)
  int stats;            // mode 1: accumulate counts / bbox
  const float* score = nullptr;   // optional [B]: prompts with score[b] <= score_thr are skipped (they are dropped by
  float score_thr = 0.f;          // the predicted-IoU filter BEFORE stability is looked at, crowdsam/model.py:371-376)
};

constexpr int POST_ROWS = 16;   // output rows per row block
#ifndef CSAM_POST_RB
#define CSAM_POST_RB 4
#endif
constexpr int POST_RB = CSAM_POST_RB;   // row blocks per workgroup of the x4 statistics pass (the byte pass keeps 1:
                                        // it is store-bound and measured 10 % slower with fewer, longer workgroups)

template <int MODE>
__global__ __launch_bounds__(256) void mask_post_kernel(PostArgs a) {
  // grid: (row chunks of POST_ROWS, 1024-pixel column strips, B); 256 threads x 4 pixels cover one strip of a row, so
  // the output width is free (test.max_size > 1024: crowdsam/utils.py:141-156 makes it a knob)
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  if (a.keep && !a.keep[b]) return;
  if (a.score && !(a.score[b] > a.score_thr)) return;
  const int x4 = (blockIdx.y * 256 + tid) * 4;
  const float* p = a.src + (long)b * a.src_bstride + (a.sel ? (long)a.sel[b] * a.plane : 0);
  int cnt_i = 0, cnt_u = 0, xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;
  int x0[4], x1[4];
  float lx[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) src_index(a.scale_x, x4 + e, a.sw, x0[e], x1[e], lx[e]);
  const int yend = min(a.H, (int)(blockIdx.x + 1) * POST_ROWS);
  for (int y = blockIdx.x * POST_ROWS; y < yend; ++y) {
    if (x4 >= a.W) break;
    int y0, y1;
    float ly;
    src_index(a.scale_y, y, a.sh, y0, y1, ly);
    uint32_t packed = 0;
    float vv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int x = x4 + e;
      vv[e] = 0.f;
      if (x < a.W) {
        const float v = bilerp(p, a.sw, y0, y1, ly, x0[e], x1[e], lx[e]);
        vv[e] = v;
        if (MODE == 1) {
          cnt_i += v > (a.thr + a.off);
          cnt_u += v > (a.thr - a.off);
          if (v > a.thr) {
            packed |= 1u << (8 * e);
            xmin = min(xmin, x);
            xmax = max(xmax, x);
            ymin = min(ymin, y);
            ymax = max(ymax, y);
          }
        }
      }
    }
    const long o = ((long)((MODE == 1 && a.slot) ? a.slot[b] : b) * a.H + y) * a.W + x4;
    if (MODE == 0) {
      if (x4 + 3 < a.W && (o & 3) == 0) {
        *(floatx4*)(a.out_f32 + o) = floatx4{vv[0], vv[1], vv[2], vv[3]};
      } else {
        for (int e = 0; e < 4 && x4 + e < a.W; ++e) a.out_f32[o + e] = vv[e];
      }
    } else if (a.out_mask) {
      if (x4 + 3 < a.W && (o & 3) == 0) {
        *(uint32_t*)(a.out_mask + o) = packed;
      } else {
        for (int e = 0; e < 4 && x4 + e < a.W; ++e) a.out_mask[o + e] = (packed >> (8 * e)) & 1;
      }
    }
  }
  if (MODE == 1 && a.stats) {
    // block reduction -> ONE set of integer atomics per workgroup (order-independent, deterministic)
    __shared__ int red[4][6];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      cnt_i += __shfl_xor(cnt_i, o, 64);
      cnt_u += __shfl_xor(cnt_u, o, 64);
      xmin = min(xmin, __shfl_xor(xmin, o, 64));
      xmax = max(xmax, __shfl_xor(xmax, o, 64));
      ymin = min(ymin, __shfl_xor(ymin, o, 64));
      ymax = max(ymax, __shfl_xor(ymax, o, 64));
    }
    if ((tid & 63) == 0) {
      int* r = red[tid >> 6];
      r[0] = cnt_i; r[1] = cnt_u; r[2] = xmin; r[3] = ymin; r[4] = xmax; r[5] = ymax;
    }
    __syncthreads();
    if (tid == 0) {
      for (int w = 1; w < 4; ++w) {
        cnt_i += red[w][0]; cnt_u += red[w][1];
        xmin = min(xmin, red[w][2]); ymin = min(ymin, red[w][3]);
        xmax = max(xmax, red[w][4]); ymax = max(ymax, red[w][5]);
      }
      if (cnt_i) atomicAdd(a.inter + b, cnt_i);
      if (cnt_u) atomicAdd(a.uni + b, cnt_u);
      if (xmax >= 0) {
        atomicMin(a.box + b * 4 + 0, xmin);
        atomicMin(a.box + b * 4 + 1, ymin);
        atomicMax(a.box + b * 4 + 2, xmax);
        atomicMax(a.box + b * 4 + 3, ymax);
      }
    }
  }
}

// Fast path of csam_mask_post for the common geometry (original_size == input_size, source = the 256x256
// low-res logits, exact x4 up-sampling): the 16 output rows of a workgroup depend on 6 source rows and each
// thread's 4 output pixels on 3 source columns, so a thread holds its 18 source values in registers and the
// kernel is a pure streaming write of the mask bytes.  Arithmetic is IDENTICAL to the generic kernel
// (same src_index / bilerp expressions), so results are bit-identical to it.
template <int PASS>   // 0: statistics only (counts + bbox), 1: mask bytes of the KEPT prompts only
__global__ __launch_bounds__(256) void mask_post_x4_kernel(PostArgs a) {
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  if (a.keep && !a.keep[b]) return;
  if (a.score && !(a.score[b] > a.score_thr)) return;
  const int x4 = tid * 4;
  const float* p = a.src + (long)b * a.src_bstride + (long)a.sel[b] * a.plane;
  // A workgroup walks POST_RB consecutive 16-row blocks: with one block per workgroup (262 144 workgroups per 4096
  // prompts) the SIMDs were ~50 % busy -- a workgroup's life was mostly dispatch, first-load latency and the
  // reduction / atomics tail, and 83 VGPRs cap the residency at six of them per CU.
  int T_ci = 0, T_cu = 0, T_xmin = 1 << 30, T_xmax = -1, T_ymin = 1 << 30, T_ymax = -1;
  constexpr int RB = PASS == 0 ? POST_RB : 1;
  for (int ib = 0; ib < RB; ++ib) {
  const int i = blockIdx.x * RB + ib;            // 16 output rows: y = 16 i .. 16 i + 15
  if (i * 16 >= a.H) break;
  int cnt_i = 0, cnt_u = 0, xmin = 1 << 30, xmax = -1, ymin = 1 << 30, ymax = -1;
  // PASS 0 box: per-pixel work is one OR into a column flag and one into the row flag; the extents are taken from the
  // flags once per row (y) and once per thread (x) instead of four predicated min / max per pixel
  uint32_t colany[4] = {0u, 0u, 0u, 0u};
  // source columns t-1, t, t+1 and rows 4i-1 .. 4i+4 (clamped)
  const int c0 = max(tid - 1, 0), c1 = tid, c2 = min(tid + 1, 255);
  float v[6][3];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int r = min(max(4 * i - 1 + k, 0), 255);
    v[k][0] = p[r * 256 + c0];
    v[k][1] = p[r * 256 + c1];
    v[k][2] = p[r * 256 + c2];
  }
  // horizontal interpolation once per (source row, output column): hl[k][e] = w0x*v0 + lx*v1 -- exactly the
  // inner expression of bilerp(), so the vertical step below reproduces the generic kernel bit for bit
  float hl[6][4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    int x0, x1;
    float lx;
    src_index(0.25f, x4 + e, 256, x0, x1, lx);
    const float w0x = 1.f - lx;
    const int i0 = (x0 == c1) ? 1 : (x0 == c0 ? 0 : 2);
    const int i1 = (x1 == c1) ? 1 : (x1 == c0 ? 0 : 2);
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const float t0 = i0 == 1 ? v[k][1] : (i0 == 0 ? v[k][0] : v[k][2]);
      const float t1 = i1 == 1 ? v[k][1] : (i1 == 0 ? v[k][0] : v[k][2]);
      hl[k][e] = w0x * t0 + lx * t1;
    }
  }
  // The row loop exists twice: FULL (the whole 16 x 1024 tile lies inside the mask: no per-pixel bounds tests, which
  // hipcc otherwise turns into an exec-mask branch per pixel -- 24 VALU + 9 SALU instructions per pixel measured) and
  // the general edge version.
  const float hi_t = a.thr + a.off, lo_t = a.thr - a.off;
  auto rows = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int yy = 0; yy < 16; ++yy) {
      const int y = i * 16 + yy;
      if (!FULL && y >= a.H) break;
      // static row slots: y0 = 4i + o, o = floor(0.25*yy - 0.375) in {-1,0,1,2,3}; y1 is the next slot
      // (clamping only ever selects a row whose weight is exactly 0, or the same clamped row)
      const int k0 = (yy < 2) ? 0 : (yy < 6) ? 1 : (yy < 10) ? 2 : (yy < 14) ? 3 : 4;
      bool same = false;
      float ly;
      if (FULL) {      // interior tile: src = (y + 0.5) / 4 - 0.5 is never clamped, its fraction depends on yy only
        ly = (float)((yy + 2) & 3) * 0.25f + 0.125f;
      } else {
        int y0, y1;
        src_index(0.25f, y, 256, y0, y1, ly);
        same = (y1 == y0);
      }
      const float w0y = 1.f - ly;
      uint32_t packed = 0, rowany = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int x = x4 + e;
        if (FULL || x < a.W) {
          const float tt = hl[k0][e];
          const float bb = same ? hl[k0][e] : hl[k0 + 1][e];
          const float val = w0y * tt + ly * bb;
          if (PASS == 0) {
            cnt_i += val > hi_t;
            cnt_u += val > lo_t;
            colany[e] = val > a.thr ? 1u : colany[e];
            rowany = val > a.thr ? 1u : rowany;
          } else {
            packed |= (val > a.thr ? 1u : 0u) << (8 * e);
          }
        }
      }
      if (PASS == 0) {
        ymin = min(ymin, rowany ? y : (1 << 30));
        ymax = max(ymax, rowany ? y : -1);
      }
      if (PASS == 1) {
        const long o = ((long)(a.slot ? a.slot[b] : b) * a.H + y) * a.W + x4;
        if (FULL || (x4 + 3 < a.W && (o & 3) == 0)) {
          *(uint32_t*)(a.out_mask + o) = packed;
        } else {
          for (int e = 0; e < 4 && x4 + e < a.W; ++e) a.out_mask[o + e] = (packed >> (8 * e)) & 1;
        }
      }
    }
  };
  if (a.W == 1024 && a.H == 1024 && i > 0 && i < 63) {
    // Value-uniform strips.  Every output pixel of this thread is a convex combination (two nested fp32 lerps with
    // weights k/8) of its 18 source logits, so it lies in [min v, max v] up to 2 roundings (< 3e-7 relative): when a
    // whole wave's strip (16 rows x 256 px) clears both stability thresholds by 1e-6 relative -- the background and
    // the interior of a mask, i.e. almost all of the frame -- the per-pixel interpolation + compares are skipped and
    // the outcome they would have produced is written directly.  Decisions are identical to the evaluated path.
    float vmin = v[0][0], vmax = v[0][0];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        vmin = fminf(vmin, v[k][c]);
        vmax = fmaxf(vmax, v[k][c]);
      }
    const bool all_lo = vmax + fabsf(vmax) * 1e-6f < lo_t && a.off >= 0.f;
    const bool all_hi = vmin - fabsf(vmin) * 1e-6f > hi_t && a.off >= 0.f;
    if (__all(all_lo)) {
      if (PASS == 1) {
        const long o = ((long)(a.slot ? a.slot[b] : b) * 1024 + i * 16) * 1024 + x4;
#pragma unroll
        for (int yy = 0; yy < 16; ++yy) *(uint32_t*)(a.out_mask + o + yy * 1024) = 0u;
      }
    } else if (__all(all_hi)) {
      if (PASS == 0) {
        cnt_i = 64;
        cnt_u = 64;
        colany[0] = colany[1] = colany[2] = colany[3] = 1u;
        ymin = i * 16;
        ymax = i * 16 + 15;
      } else {
        const long o = ((long)(a.slot ? a.slot[b] : b) * 1024 + i * 16) * 1024 + x4;
#pragma unroll
        for (int yy = 0; yy < 16; ++yy) *(uint32_t*)(a.out_mask + o + yy * 1024) = 0x01010101u;
      }
    } else if (PASS == 0) {
      // Interior strip, statistics pass.  The VALU is the bound here (one wave64 VALU instruction = 4 cycles): the
      // vertical lerp runs on pixel PAIRS (v_pk_mul / v_pk_add; fp contraction is off, so the arithmetic is the
      // generic kernel's mul, mul, add bit for bit); the compares against the mask threshold and the upper stability
      // threshold write their 64-lane masks to SGPRs, where the count (s_bcnt1), the row flags and the column flags
      // (s_or) accumulate on the scalar unit -- 5.5 VALU + 4 SALU instructions per pixel instead of ~13 VALU.
      typedef float f2 __attribute__((ext_vector_type(2)));
      int sc_i = 0;                                 // wave-uniform counter
      unsigned long long colm[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
      for (int yy = 0; yy < 16; ++yy) {
        const int k0 = (yy < 2) ? 0 : (yy < 6) ? 1 : (yy < 10) ? 2 : (yy < 14) ? 3 : 4;
        const float ly = (float)((yy + 2) & 3) * 0.25f + 0.125f;
        const float w0y = 1.f - ly;
        unsigned long long rowm = 0ull;
#pragma unroll
        for (int e2 = 0; e2 < 2; ++e2) {
          const f2 tt = {hl[k0][2 * e2], hl[k0][2 * e2 + 1]}, bb = {hl[k0 + 1][2 * e2], hl[k0 + 1][2 * e2 + 1]};
          const f2 t1 = (f2){w0y, w0y} * tt;
          const f2 t2 = (f2){ly, ly} * bb;
          const f2 val = t1 + t2;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            // the scalar unit issues one instruction per cycle per CU, the same rate as the four SIMDs' VALUs together:
            // the work is split between them -- `inter` counts on the scalar side (compare -> mask -> s_bcnt1), `union`
            // counts on the vector side (compare -> add-with-carry), the mask bits as SGPR ORs
            sc_i += __popcll(__ballot(val[h] > hi_t));
            cnt_u += val[h] > lo_t;
            const unsigned long long m = __ballot(val[h] > a.thr);
            colm[2 * e2 + h] |= m;
            rowm |= m;
          }
        }
        if (rowm) {
          ymin = min(ymin, i * 16 + yy);
          ymax = max(ymax, i * 16 + yy);
        }
      }
      const int lane = tid & 63;
      cnt_i = lane == 0 ? sc_i : 0;                 // the wave total enters the block reduction once
#pragma unroll
      for (int e = 0; e < 4; ++e) colany[e] = (colm[e] >> lane) & 1ull ? 1u : 0u;
    } else {
      rows(std::true_type{});
    }
  } else if (x4 < a.W) {
    rows(std::false_type{});
  }
  if (PASS == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (colany[e]) {
        xmin = min(xmin, x4 + e);
        xmax = max(xmax, x4 + e);
      }
    T_ci += cnt_i;
    T_cu += cnt_u;
    T_xmin = min(T_xmin, xmin);
    T_xmax = max(T_xmax, xmax);
    T_ymin = min(T_ymin, ymin);
    T_ymax = max(T_ymax, ymax);
  }
  }   // row blocks
  if (PASS == 1) return;
  int cnt_i = T_ci, cnt_u = T_cu, xmin = T_xmin, xmax = T_xmax, ymin = T_ymin, ymax = T_ymax;
  __shared__ int red[4][6];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    cnt_i += __shfl_xor(cnt_i, o, 64);
    cnt_u += __shfl_xor(cnt_u, o, 64);
    xmin = min(xmin, __shfl_xor(xmin, o, 64));
    xmax = max(xmax, __shfl_xor(xmax, o, 64));
    ymin = min(ymin, __shfl_xor(ymin, o, 64));
    ymax = max(ymax, __shfl_xor(ymax, o, 64));
  }
  if ((tid & 63) == 0) {
    int* r = red[tid >> 6];
    r[0] = cnt_i; r[1] = cnt_u; r[2] = xmin; r[3] = ymin; r[4] = xmax; r[5] = ymax;
  }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) {
      cnt_i += red[w][0]; cnt_u += red[w][1];
      xmin = min(xmin, red[w][2]); ymin = min(ymin, red[w][3]);
      xmax = max(xmax, red[w][4]); ymax = max(ymax, red[w][5]);
    }
    if (cnt_i) atomicAdd(a.inter + b, cnt_i);
    if (cnt_u) atomicAdd(a.uni + b, cnt_u);
    if (xmax >= 0) {
      atomicMin(a.box + b * 4 + 0, xmin);
      atomicMin(a.box + b * 4 + 1, ymin);
      atomicMax(a.box + b * 4 + 2, xmax);
      atomicMax(a.box + b * 4 + 3, ymax);
    }
  }
}

__global__ __launch_bounds__(256) void post_init_kernel(int* inter, int* uni, int* box, int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  inter[b] = 0;
  uni[b] = 0;
  box[b * 4 + 0] = 1 << 30;
  box[b * 4 + 1] = 1 << 30;
  box[b * 4 + 2] = -1;
  box[b * 4 + 3] = -1;
}

// keep = (score > pred_iou_thresh) && (inter/union >= stability_thresh); empty mask -> box zeros
// (amg.py:341-343); occupancy flag = keep && score > filter_thresh (crowdsam/model.py:246).
__global__ __launch_bounds__(256) void post_finalize_kernel(const float* __restrict__ score, const int* __restrict__ inter,
                                                            const int* __restrict__ uni, int* __restrict__ box,
                                                            float pred_iou_thresh, float stab_thresh,
                                                            float filter_thresh, float* __restrict__ stability,
                                                            uint8_t* __restrict__ keep, uint8_t* __restrict__ occ,
                                                            int B) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float st = (float)inter[b] / (float)uni[b];
  stability[b] = st;
  bool k = true;
  if (pred_iou_thresh > 0.f) k = k && (score[b] > pred_iou_thresh);
  if (stab_thresh > 0.f) k = k && (st >= stab_thresh);
  if (box[b * 4 + 2] < box[b * 4 + 0] || box[b * 4 + 3] < box[b * 4 + 1]) {
    box[b * 4 + 0] = box[b * 4 + 1] = box[b * 4 + 2] = box[b * 4 + 3] = 0;
  }
  keep[b] = k;
  occ[b] = k && (score[b] > filter_thresh);
}

// Compacting variant: the same decisions, plus an in-kernel exclusive scan of the keep flags that assigns
// every surviving prompt a slot in the IMAGE-level result store (base = running device counter), scatters
// its small fields there and bumps the counter.  No host round trip, no gather copies afterwards: the second
// mask pass writes the bytes straight to store[slot].  One workgroup (B <= 4096).
// utils.is_box_near_crop_edge (crowdsam/utils.py:213-223): un-crop the box (box / downscale + crop offset, fp32 as torch
// computes it), drop it when a side lies within atol of the crop box but not of the image box
struct EdgeFilter {
  float crop[4], orig[4], downscale, atol;
  int enabled;
};

__global__ __launch_bounds__(1024) void post_finalize_compact_kernel(
    const float* __restrict__ score, const int* __restrict__ inter, const int* __restrict__ uni,
    const int* __restrict__ box, const int* __restrict__ category, const int* __restrict__ points,
    float pred_iou_thresh, float stab_thresh, float filter_thresh, uint8_t* __restrict__ keep,
    uint8_t* __restrict__ occ, int* __restrict__ slot, int* __restrict__ counter, float* __restrict__ o_score,
    float* __restrict__ o_stab, int* __restrict__ o_box, int* __restrict__ o_cat, int* __restrict__ o_pts, int B,
    int cap, EdgeFilter edge, const int* __restrict__ n_valid) {
  __shared__ int wsum[16];
  __shared__ int base_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base_s = *counter;
  const int nv = n_valid ? *n_valid : B;          // device-resident sampler: slots >= *n_valid hold no prompt
  __syncthreads();
  for (int b0 = 0; b0 < B; b0 += 1024) {
    const int b = b0 + tid;
    bool k = false;
    float st = 0.f;
    int bx[4] = {0, 0, 0, 0};
    if (b < B) {
      st = (float)inter[b] / (float)uni[b];
      k = b < nv;
      if (pred_iou_thresh > 0.f) k = k && (score[b] > pred_iou_thresh);
      if (stab_thresh > 0.f) k = k && (st >= stab_thresh);
#pragma unroll
      for (int e = 0; e < 4; ++e) bx[e] = box[b * 4 + e];
      if (bx[2] < bx[0] || bx[3] < bx[1]) bx[0] = bx[1] = bx[2] = bx[3] = 0;
      if (edge.enabled) {   // the reference applies it inside _process_batch, BEFORE the occupancy mask is built (:386-389)
        bool near = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // tensor / Python-scalar on the reference's device path is a multiply by the fp32 reciprocal (ATen's div-by-scalar)
          const float v = (float)bx[e] * (1.0f / edge.downscale) + edge.crop[e & 1];
          near = near || (fabsf(v - edge.crop[e]) <= edge.atol && !(fabsf(v - edge.orig[e]) <= edge.atol));
        }
        k = k && !near;
      }
      keep[b] = k;
      occ[b] = k && (score[b] > filter_thresh);
    }
    // exclusive scan of k over the 1024 threads
    const unsigned long long bal = __ballot(k);
    const int inwave = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wsum[wave] = __popcll(bal);
    __syncthreads();
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      if (w < wave) woff += wsum[w];
      total += wsum[w];
    }
    const int base = base_s;
    if (b < B) {
      int sl = -1;
      if (k) {
        sl = base + woff + inwave;
        if (sl < cap) {
          o_score[sl] = score[b];
          o_stab[sl] = st;
#pragma unroll
          for (int e = 0; e < 4; ++e) o_box[sl * 4 + e] = bx[e];
          o_cat[sl] = category[b];
          o_pts[sl * 2] = points[b * 2];
          o_pts[sl * 2 + 1] = points[b * 2 + 1];
        } else {
          sl = -1;   // store full: dropped (cap is sized for max_prompts, cannot happen in the driver)
          keep[b] = 0;
          occ[b] = 0;
        }
      }
      slot[b] = sl;
    }
    __syncthreads();
    if (tid == 0) base_s = min(base + total, cap);
    __syncthreads();
  }
  if (tid == 0) *counter = base_s;
}

// out[p] = OR_b occ[b] & mask[b, y_p, x_p]   (crowdsam/model.py:238,246)
__global__ __launch_bounds__(256) void occupancy_kernel(const int* __restrict__ pts, int P,
                                                        const uint8_t* __restrict__ masks,
                                                        const uint8_t* __restrict__ occ,
                                                        const int* __restrict__ slot, int B, int H, int W,
                                                        uint8_t* __restrict__ out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const int x = pts[p * 2], y = pts[p * 2 + 1];
  uint8_t r = 0;
  if (x >= 0 && x < W && y >= 0 && y < H) {
    for (int b = 0; b < B; ++b)
      if (occ[b]) r |= masks[((long)(slot ? slot[b] : b) * H + y) * W + x];
  }
  out[p] = r;
}

// ---- Efficient Prompt Sampler, device-resident (crowdsam/model.py:233-249).  The reference keeps the shuffled point list on
// the host, takes points[:batch_size] each round and drops the points that fall under the round's masks.  Here the list
// stays on the device with one alive flag per point: a round's batch = the first B points still alive, in list order (the
// same points, in the same order), and the pruning clears flags.  No host round trip per batch.
//
// eps_select: out_pts / out_coords [B] <- the first min(B, #alive) alive points (flags cleared: they are consumed), the rest
// of the B slots are filled with (0, 0) and reported through counts[0] = n_valid; counts[1] = points left alive afterwards.
// Coordinates go through ResizeLongestSide.apply_coords in float64 as the reference does on the host (transforms.py:33-41,
// trap 6): x * (new_w / old_w), y * (new_h / old_h), then the fp32 cast of torch.as_tensor(..., dtype=float).
__global__ __launch_bounds__(1024) void eps_select_kernel(const int* __restrict__ pts, uint8_t* __restrict__ alive, int P,
                                                          int B, double sx, double sy, int* __restrict__ out_pts,
                                                          float* __restrict__ out_coords, int* __restrict__ counts) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chunk = (P + 1023) / 1024;
  const int p0 = min(tid * chunk, P), p1 = min(p0 + chunk, P);
  int cnt = 0;
  for (int p = p0; p < p1; ++p) cnt += alive[p] != 0;
  int incl = cnt;                                  // inclusive scan inside the wave, then over the 16 wave totals
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int woff = 0, total = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) {
    if (w < wave) woff += wsum[w];
    total += wsum[w];
  }
  int rank = woff + incl - cnt;                    // alive points in front of this thread's chunk
  for (int p = p0; p < p1 && rank < B; ++p) {
    if (alive[p]) {
      const int x = pts[p * 2], y = pts[p * 2 + 1];
      out_pts[rank * 2] = x;
      out_pts[rank * 2 + 1] = y;
      out_coords[rank * 2] = (float)((double)x * sx);
      out_coords[rank * 2 + 1] = (float)((double)y * sy);
      alive[p] = 0;
      ++rank;
    }
  }
  const int nv = min(total, B);
  for (int i = nv + tid; i < B; i += 1024) {
    out_pts[i * 2] = out_pts[i * 2 + 1] = 0;
    out_coords[i * 2] = out_coords[i * 2 + 1] = 0.f;
  }
  if (tid == 0) {
    counts[0] = nv;
    counts[1] = total - nv;
  }
}

// alive[p] &= !(OR_b occ[b] & mask[b, y_p, x_p])   (crowdsam/model.py:238-239,246)
__global__ __launch_bounds__(256) void occupancy_prune_kernel(const int* __restrict__ pts, int P,
                                                              const uint8_t* __restrict__ masks,
                                                              const uint8_t* __restrict__ occ,
                                                              const int* __restrict__ slot, int B, int H, int W,
                                                              uint8_t* __restrict__ alive) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P || !alive[p]) return;
  const int x = pts[p * 2], y = pts[p * 2 + 1];
  if (x < 0 || x >= W || y < 0 || y >= H) return;
  uint8_t r = 0;
  for (int b = 0; b < B; ++b)
    if (occ[b]) r |= masks[((long)(slot ? slot[b] : b) * H + y) * W + x];
  if (r) alive[p] = 0;
}

}  // namespace

extern "C" int csam_eps_select(void* stream, const int* points_xy, void* alive_u8, int P, int B, double scale_x,
                               double scale_y, int* out_points_xy, float* out_coords, int* counts2) {
  CSAM_REQUIRE(points_xy && alive_u8 && out_points_xy && out_coords && counts2 && P > 0 && B > 0,
               "csam_eps_select: bad args");
  hipLaunchKernelGGL(eps_select_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, points_xy, (uint8_t*)alive_u8, P, B,
                     scale_x, scale_y, out_points_xy, out_coords, counts2);
  CSAM_LAUNCH_CHECK("csam_eps_select");
  return CSAM_OK;
}

extern "C" int csam_occupancy_prune(void* stream, const int* points_xy, int P, const void* masks_u8, const void* occ_u8,
                                    const int* slot_or_null, int B, int H, int W, void* alive_u8) {
  CSAM_REQUIRE(points_xy && masks_u8 && occ_u8 && alive_u8 && P > 0 && B > 0, "csam_occupancy_prune: bad args");
  hipLaunchKernelGGL(occupancy_prune_kernel, dim3(csam_cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, points_xy, P,
                     (const uint8_t*)masks_u8, (const uint8_t*)occ_u8, slot_or_null, B, H, W, (uint8_t*)alive_u8);
  CSAM_LAUNCH_CHECK("csam_occupancy_prune");
  return CSAM_OK;
}

extern "C" int csam_select_masks(void* stream, const float* iou, const float* cls, int n_class, int* sel,
                                 float* score, int* category, float* fused_or_null, int B) {
  CSAM_REQUIRE(iou && cls && sel && score && category && B > 0 && n_class > 0, "csam_select_masks: bad args");
  hipLaunchKernelGGL(select_kernel, dim3(csam_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, iou, cls, n_class,
                     sel, score, category, fused_or_null, B);
  CSAM_LAUNCH_CHECK("csam_select_masks");
  return CSAM_OK;
}

// Fused path of sam.py:153-161 for original_size == input_size (the normal case) and the general
// two-stage path otherwise (trap 9: 1023-sided frames).  tmp_f32 [B,in_h,in_w] is only needed when
// (out_h,out_w) != (in_h,in_w).  do_stats: counts + bbox (csam_mask_post); out_mask may be NULL (statistics
// only: the bytes of the prompts that survive the filters are produced later by csam_mask_write).
static int post_launch(hipStream_t s, const float* lowres, const int* sel, const uint8_t* keep, const int* slot,
                       int B, int in_h, int in_w, int out_h, int out_w, float thr, float off, void* out_mask_u8,
                       int* inter, int* uni, int* box, float* tmp_f32, int do_stats, const float* score = nullptr,
                       float score_thr = 0.f) {
  PostArgs a;
  a.score = score; a.score_thr = score_thr;
  a.thr = thr; a.off = off;
  a.inter = inter; a.uni = uni; a.box = box; a.keep = keep; a.slot = slot; a.stats = do_stats;
  if (in_h == out_h && in_w == out_w) {
    a.src = lowres; a.src_bstride = 4L * 65536; a.plane = 65536; a.sel = sel;
    a.sh = 256; a.sw = 256; a.scale_y = 256.0f / 1024.0f; a.scale_x = 256.0f / 1024.0f;
    a.H = out_h; a.W = out_w; a.out_f32 = nullptr; a.out_mask = (uint8_t*)out_mask_u8;
    dim3 grid0(csam_cdiv(csam_cdiv(out_h, POST_ROWS), POST_RB), 1, B), grid1(csam_cdiv(out_h, POST_ROWS), 1, B);
    if (do_stats) hipLaunchKernelGGL(mask_post_x4_kernel<0>, grid0, dim3(256), 0, s, a);
    if (out_mask_u8) hipLaunchKernelGGL(mask_post_x4_kernel<1>, grid1, dim3(256), 0, s, a);
  } else {
    if (!tmp_f32) {
      csam_set_error("csam_mask_post: tmp buffer required when original_size != input_size");
      return CSAM_ERR_ARG;
    }
    a.src = lowres; a.src_bstride = 4L * 65536; a.plane = 65536; a.sel = sel;
    a.sh = 256; a.sw = 256; a.scale_y = 0.25f; a.scale_x = 0.25f;
    a.H = in_h; a.W = in_w; a.out_f32 = tmp_f32; a.out_mask = nullptr;
    dim3 g0(csam_cdiv(in_h, POST_ROWS), 1, B);
    hipLaunchKernelGGL(mask_post_kernel<0>, g0, dim3(256), 0, s, a);
    a.src = tmp_f32; a.src_bstride = (long)in_h * in_w; a.plane = 0; a.sel = nullptr;
    a.sh = in_h; a.sw = in_w; a.scale_y = (float)in_h / (float)out_h; a.scale_x = (float)in_w / (float)out_w;
    a.H = out_h; a.W = out_w; a.out_f32 = nullptr; a.out_mask = (uint8_t*)out_mask_u8;
    dim3 g1(csam_cdiv(out_h, POST_ROWS), csam_cdiv(out_w, 1024), B);
    hipLaunchKernelGGL(mask_post_kernel<1>, g1, dim3(256), 0, s, a);
  }
  return CSAM_OK;
}

extern "C" int csam_mask_post(void* stream, const float* lowres, const int* sel, int B, int in_h, int in_w,
                              int out_h, int out_w, float thr, float off, void* out_mask_u8, int* inter, int* uni,
                              int* box, float* tmp_f32) {
  CSAM_REQUIRE(lowres && sel && inter && uni && box && B > 0, "csam_mask_post: bad args");
  CSAM_REQUIRE(in_h > 0 && in_w > 0 && in_h <= 1024 && in_w <= 1024 && out_h > 0 && out_w > 0,
               "csam_mask_post: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(post_init_kernel, dim3(csam_cdiv(B, 256)), dim3(256), 0, s, inter, uni, box, B);
  const int rc = post_launch(s, lowres, sel, nullptr, nullptr, B, in_h, in_w, out_h, out_w, thr, off, out_mask_u8,
                             inter, uni, box, tmp_f32, 1);
  if (rc) return rc;
  CSAM_LAUNCH_CHECK("csam_mask_post");
  return CSAM_OK;
}

// Statistics pass that skips the prompts the predicted-IoU filter drops anyway (score[b] <= score_thr, or
// score_thr <= 0: none): their inter / uni / box keep the initial values and post_finalize* rejects them on the score.
extern "C" int csam_mask_post_scored(void* stream, const float* lowres, const int* sel, const float* score,
                                     float score_thr, int B, int in_h, int in_w, int out_h, int out_w, float thr,
                                     float off, int* inter, int* uni, int* box, float* tmp_f32) {
  CSAM_REQUIRE(lowres && sel && score && inter && uni && box && B > 0, "csam_mask_post_scored: bad args");
  CSAM_REQUIRE(in_h > 0 && in_w > 0 && in_h <= 1024 && in_w <= 1024 && out_h > 0 && out_w > 0,
               "csam_mask_post_scored: bad sizes");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(post_init_kernel, dim3(csam_cdiv(B, 256)), dim3(256), 0, s, inter, uni, box, B);
  const int rc = post_launch(s, lowres, sel, nullptr, nullptr, B, in_h, in_w, out_h, out_w, thr, off, nullptr, inter, uni,
                             box, tmp_f32, 1, score_thr > 0.f ? score : nullptr, score_thr);
  if (rc) return rc;
  CSAM_LAUNCH_CHECK("csam_mask_post_scored");
  return CSAM_OK;
}

// Second pass of the two-pass mode: mask bytes (x > thr) of the prompts with keep[b] != 0 only
// (keep == NULL: all).  Rows of skipped prompts in out_mask are left untouched.
extern "C" int csam_mask_write(void* stream, const float* lowres, const int* sel, const void* keep_u8,
                               const int* slot_or_null, int B, int in_h, int in_w, int out_h, int out_w, float thr,
                               void* out_mask_u8, float* tmp_f32) {
  CSAM_REQUIRE(lowres && sel && out_mask_u8 && B > 0, "csam_mask_write: bad args");
  CSAM_REQUIRE(in_h > 0 && in_w > 0 && in_h <= 1024 && in_w <= 1024 && out_h > 0 && out_w > 0,
               "csam_mask_write: bad sizes");
  const int rc = post_launch((hipStream_t)stream, lowres, sel, (const uint8_t*)keep_u8, slot_or_null, B, in_h, in_w,
                             out_h, out_w, thr, 0.f, out_mask_u8, nullptr, nullptr, nullptr, tmp_f32, 0);
  if (rc) return rc;
  CSAM_LAUNCH_CHECK("csam_mask_write");
  return CSAM_OK;
}

// Generic bilinear resize (align_corners=False) of fp32 planes [n, sh, sw] -> [n, H, W]:
// predictor.py:116 (FG logits 73->256), crowdsam/model.py:202 (256 -> grid), predictor.py:104
// (1024 -> 1022 for third-party DINO models), and the API-compatible full postprocess_masks.
extern "C" int csam_bilinear_f32(void* stream, const float* src, int n, int sh, int sw, float* dst, int H, int W) {
  CSAM_REQUIRE(src && dst && n > 0 && sh > 0 && sw > 0 && H > 0 && W > 0, "csam_bilinear_f32: bad args");
  PostArgs a;
  a.src = src; a.src_bstride = (long)sh * sw; a.plane = 0; a.sel = nullptr;
  a.sh = sh; a.sw = sw; a.scale_y = (float)sh / (float)H; a.scale_x = (float)sw / (float)W;
  a.H = H; a.W = W; a.thr = 0.f; a.off = 0.f;
  a.out_f32 = dst; a.out_mask = nullptr; a.inter = nullptr; a.uni = nullptr; a.box = nullptr;
  a.keep = nullptr; a.slot = nullptr; a.stats = 0;
  dim3 grid(csam_cdiv(H, POST_ROWS), csam_cdiv(W, 1024), n);
  hipLaunchKernelGGL(mask_post_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, a);
  CSAM_LAUNCH_CHECK("csam_bilinear_f32");
  return CSAM_OK;
}

extern "C" int csam_post_finalize(void* stream, const float* score, const int* inter, const int* uni, int* box,
                                  float pred_iou_thresh, float stability_thresh, float filter_thresh,
                                  float* stability, void* keep_u8, void* occ_u8, int B) {
  CSAM_REQUIRE(score && inter && uni && box && stability && keep_u8 && occ_u8 && B > 0, "csam_post_finalize: bad args");
  hipLaunchKernelGGL(post_finalize_kernel, dim3(csam_cdiv(B, 256)), dim3(256), 0, (hipStream_t)stream, score, inter,
                     uni, box, pred_iou_thresh, stability_thresh, filter_thresh, stability, (uint8_t*)keep_u8,
                     (uint8_t*)occ_u8, B);
  CSAM_LAUNCH_CHECK("csam_post_finalize");
  return CSAM_OK;
}

// csam_post_finalize + in-kernel compaction into an image-level store (see the kernel comment).
extern "C" int csam_post_finalize_compact(void* stream, const float* score, const int* inter, const int* uni,
                                          const int* box, const int* category, const int* points_xy,
                                          float pred_iou_thresh, float stability_thresh, float filter_thresh,
                                          void* keep_u8, void* occ_u8, int* slot, int* counter, float* out_score,
                                          float* out_stability, int* out_box, int* out_category, int* out_points,
                                          int B, int capacity, const float* edge10_host,
                                          const int* n_valid_or_null) {
  CSAM_REQUIRE(score && inter && uni && box && category && points_xy && keep_u8 && occ_u8 && slot && counter &&
                   out_score && out_stability && out_box && out_category && out_points && B > 0 && capacity > 0,
               "csam_post_finalize_compact: bad args");
  EdgeFilter edge = {};
  if (edge10_host) {
    for (int e = 0; e < 4; ++e) edge.crop[e] = edge10_host[e], edge.orig[e] = edge10_host[4 + e];
    edge.downscale = edge10_host[8];
    edge.atol = edge10_host[9];
    edge.enabled = 1;
  }
  hipLaunchKernelGGL(post_finalize_compact_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, score, inter, uni, box,
                     category, points_xy, pred_iou_thresh, stability_thresh, filter_thresh, (uint8_t*)keep_u8,
                     (uint8_t*)occ_u8, slot, counter, out_score, out_stability, out_box, out_category, out_points, B,
                     capacity, edge, n_valid_or_null);
  CSAM_LAUNCH_CHECK("csam_post_finalize_compact");
  return CSAM_OK;
}

extern "C" int csam_occupancy_lookup(void* stream, const int* points_xy, int P, const void* masks_u8,
                                     const void* occ_u8, const int* slot_or_null, int B, int H, int W,
                                     void* out_u8) {
  CSAM_REQUIRE(points_xy && masks_u8 && occ_u8 && out_u8 && P > 0 && B > 0, "csam_occupancy_lookup: bad args");
  hipLaunchKernelGGL(occupancy_kernel, dim3(csam_cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, points_xy, P,
                     (const uint8_t*)masks_u8, (const uint8_t*)occ_u8, slot_or_null, B, H, W, (uint8_t*)out_u8);
  CSAM_LAUNCH_CHECK("csam_occupancy_lookup");
  return CSAM_OK;
}
