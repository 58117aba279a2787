// csam_box_nms: greedy box NMS with torchvision semantics, plus csam_rle_*: column-major RLE.
//
// Reference call sites: crowdsam/model.py:257-262, 429-434, 171-176 (torchvision.ops.batched_nms with
// all-zero category ids == plain nms): candidates in STABLE descending-score order; a box is
// suppressed when IoU > thr with an earlier kept box; IoU = inter / (area_i + area_j - inter) with
// area = (x2-x1)*(y2-y1), all fp32; kept indices are returned in descending-score order.
//
// Three kernels on one stream, no host sync:
//   1. bitonic sort of 64-bit keys (~orderable(score) << 32 | index) in LDS (N <= 16384),
//   2. upper-triangular IoU bitmask, one 64-bit word per (row, 64-column block) via per-lane loops,
//   3. a single-wave sequential scan: the "removed" bitmap lives in registers (one 64-bit word per
//      lane per 4096 boxes), rows are streamed through LDS 64 at a time.
#include "csam_common.h"

namespace {

constexpr int NMS_MAX = 16384;

__device__ __forceinline__ uint32_t orderable(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending order of floats
}

__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ scores, int N, int NP,
                                                        int* __restrict__ order) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* keys = (uint64_t*)smem;
  for (int i = threadIdx.x; i < NP; i += 1024) {
    uint64_t k = ~0ull;  // padding sorts last
    if (i < N) k = ((uint64_t)(~orderable(scores[i])) << 32) | (uint32_t)i;
    keys[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= NP; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < NP; i += 1024) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const uint64_t a = keys[i], b = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) {
            keys[i] = b;
            keys[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
  for (int i = threadIdx.x; i < N; i += 1024) order[i] = (int)(keys[i] & 0xffffffffu);
}

#pragma clang fp contract(off)
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const int* __restrict__ order,
                                                      int N, int nw, float thr, uint64_t* __restrict__ mask) {
  const int cb = blockIdx.x, rb = blockIdx.y;
  if (cb < rb) return;
  __shared__ float cbx[64][4];
  const int t = threadIdx.x;
  const int cj = cb * 64 + t;
  if (cj < N) {
    const float* bp = boxes + (long)order[cj] * 4;
    cbx[t][0] = bp[0]; cbx[t][1] = bp[1]; cbx[t][2] = bp[2]; cbx[t][3] = bp[3];
  }
  __syncthreads();
  const int i = rb * 64 + t;
  if (i >= N) return;
  const float* bi = boxes + (long)order[i] * 4;
  const float x1 = bi[0], y1 = bi[1], x2 = bi[2], y2 = bi[3];
  const float iarea = (x2 - x1) * (y2 - y1);
  uint64_t w = 0;
  const int jn = min(64, N - cb * 64);
  for (int jj = 0; jj < jn; ++jj) {
    const int j = cb * 64 + jj;
    if (j <= i) continue;
    const float xx1 = fmaxf(x1, cbx[jj][0]), yy1 = fmaxf(y1, cbx[jj][1]);
    const float xx2 = fminf(x2, cbx[jj][2]), yy2 = fminf(y2, cbx[jj][3]);
    const float ww = fmaxf(0.f, xx2 - xx1), hh = fmaxf(0.f, yy2 - yy1);
    const float inter = ww * hh;
    const float jarea = (cbx[jj][2] - cbx[jj][0]) * (cbx[jj][3] - cbx[jj][1]);
    const float ovr = inter / (iarea + jarea - inter);
    if (ovr > thr) w |= 1ull << jj;
  }
  mask[(long)i * nw + cb] = w;
}

__global__ __launch_bounds__(64) void nms_scan_kernel(const uint64_t* __restrict__ mask, const int* __restrict__ order,
                                                      int N, int nw, long* __restrict__ keep, int* __restrict__ count) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint64_t* rows = (uint64_t*)smem;   // [64][nw]
  const int lane = threadIdx.x;
  uint64_t removed[4] = {0, 0, 0, 0};  // word index w = lane + 64*s
  int nkeep = 0;
  const int ntile = (N + 63) / 64;
  for (int tile = 0; tile < ntile; ++tile) {
    const int r0 = tile * 64;
    const int nr = min(64, N - r0);
    uint64_t cur = __shfl(removed[tile >> 6], tile & 63, 64);
    // candidates of this tile already suppressed by earlier tiles can never be kept: their rows are not needed
    const uint64_t valid = nr == 64 ? ~0ull : ((1ull << nr) - 1ull);
    if ((cur & valid) == valid) continue;
    // stage the remaining rows r0..r0+nr-1, words >= tile only (lower words are never set)
    for (int r = 0; r < nr; ++r) {
      if ((cur >> r) & 1ull) continue;
      for (int w = lane; w < nw; w += 64) rows[r * nw + w] = (w >= tile) ? mask[(long)(r0 + r) * nw + w] : 0ull;
    }
    __syncthreads();
    for (int r = 0; r < nr; ++r) {
      if (!((cur >> r) & 1ull)) {
        if (lane == 0) keep[nkeep] = (long)order[r0 + r];
        ++nkeep;
        cur |= rows[r * nw + tile];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int w = lane + 64 * s;
          if (w < nw) removed[s] |= rows[r * nw + w];
        }
      }
    }
    __syncthreads();
  }
  if (lane == 0) *count = nkeep;
}

// ---------------------------------------------------------------------------------------------
// Mask "coverage" NMS (crowdsam/utils.py:422-467 mask_iou_nms + coverage): every mask is nearest-resampled
// to 150x150 (F.interpolate default mode: src = min(floorf(dst * (float)in / 150), in - 1)), then greedily in
// descending score a mask is dropped when max(inter/|A|, inter/|B|) > thr against a KEPT mask (fp32 division
// of the integer counts, NaN from an empty mask never suppresses -- torch.maximum / `>` semantics).
// Here: 150x150 bits packed into 352 64-bit words per mask (ballot), the N x N intersections are AND+popcount
// over LDS-staged column tiles, and the sequential part is the same single-wave bitmap scan box NMS uses.
// ---------------------------------------------------------------------------------------------
constexpr int MN_SIDE = 150, MN_BITS = MN_SIDE * MN_SIDE, MN_WORDS = (MN_BITS + 63) / 64;   // 352
constexpr int MN_CHUNK = 32;                                                                 // words per LDS stage

__global__ __launch_bounds__(256) void mask_pack_kernel(const uint8_t* __restrict__ masks, int H, int W,
                                                        uint64_t* __restrict__ packed, int* __restrict__ area) {
  const int n = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint8_t* m = masks + (long)n * H * W;
  const float sh = (float)H / (float)MN_SIDE, sw = (float)W / (float)MN_SIDE;
  int cnt = 0;
  for (int w = wave; w < MN_WORDS; w += 4) {
    const int idx = w * 64 + lane;
    bool bit = false;
    if (idx < MN_BITS) {
      const int y = idx / MN_SIDE, x = idx - y * MN_SIDE;
      const int sy = min((int)floorf((float)y * sh), H - 1), sx = min((int)floorf((float)x * sw), W - 1);
      bit = m[(long)sy * W + sx] != 0;
    }
    const unsigned long long word = __ballot(bit);
    if (lane == 0) packed[(long)n * MN_WORDS + w] = word;
    cnt += __popcll(word);
  }
  __shared__ int part[4];
  if (lane == 0) part[wave] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) area[n] = part[0] + part[1] + part[2] + part[3];
}

__global__ __launch_bounds__(64) void mask_cov_kernel(const uint64_t* __restrict__ packed, const int* __restrict__ area,
                                                      const int* __restrict__ order, int N, int nw, float thr,
                                                      uint64_t* __restrict__ mask) {
  const int cb = blockIdx.x, rb = blockIdx.y;
  if (cb < rb) return;
  __shared__ uint64_t cols[64][MN_CHUNK];
  const int t = threadIdx.x;
  const int i = rb * 64 + t;
  const uint64_t* myrow = packed + (long)order[min(i, N - 1)] * MN_WORDS;
  int acc[64];
#pragma unroll
  for (int jj = 0; jj < 64; ++jj) acc[jj] = 0;
  for (int c0 = 0; c0 < MN_WORDS; c0 += MN_CHUNK) {
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < MN_CHUNK; ++k) {                   // 64 masks x 32 words, 256 contiguous bytes per mask
      const int e = k * 64 + t, mj = e / MN_CHUNK, w = e % MN_CHUNK;
      const int j = cb * 64 + mj;
      cols[mj][w] = j < N ? packed[(long)order[j] * MN_WORDS + c0 + w] : 0ull;
    }
    uint64_t r[MN_CHUNK];
#pragma unroll
    for (int w = 0; w < MN_CHUNK; ++w) r[w] = myrow[c0 + w];
    __syncthreads();
#pragma unroll
    for (int jj = 0; jj < 64; ++jj) {
      int a = acc[jj];
#pragma unroll
      for (int w = 0; w < MN_CHUNK; ++w) a += __popcll(r[w] & cols[jj][w]);   // LDS broadcast reads
      acc[jj] = a;
    }
  }
  if (i >= N) return;
  const int ai = area[order[i]];
  uint64_t bits = 0;
#pragma unroll
  for (int jj = 0; jj < 64; ++jj) {
    const int j = cb * 64 + jj;
    if (j > i && j < N) {
      const int aj = area[order[j]];
      if (ai > 0 && aj > 0) {                              // 0/0 = NaN in the reference: never suppresses
        const float inter = (float)acc[jj];
        const float c = fmaxf(inter / (float)ai, inter / (float)aj);
        if (c > thr) bits |= 1ull << jj;
      }
    }
  }
  mask[(long)i * nw + cb] = bits;
}

// ---------------------------------------------------------------------------------------------
// Column-major RLE (amg.py:107-135 mask_to_rle_pytorch): positions i in [1, H*W) of the
// Fortran-order flattening (i = x*H + y) where the value changes.  Thread = column; pass 1 counts
// per column, pass 2 writes the sorted positions at the exclusive-scan offsets.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rle_count_kernel(const uint8_t* __restrict__ masks, const int* __restrict__ idx,
                                                        int H, int W, int* __restrict__ col_counts) {
  const int n = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const uint8_t* m = masks + (long)(idx ? idx[n] : n) * H * W;
  uint8_t prev = x > 0 ? m[(long)(H - 1) * W + x - 1] : m[x];   // x==0: no change at i==0
  int c = 0;
  for (int y = 0; y < H; ++y) {
    const uint8_t v = m[(long)y * W + x];
    c += (v != prev);
    prev = v;
  }
  col_counts[(long)n * W + x] = c;
}

// Four columns per thread (W % 4 == 0): one 4-byte load per row instead of four byte loads; the same counts / positions.
// ``boxes`` (optional, int32 [n][4] = x0, y0, x1, y1 with inclusive maxima: the masks' bounding boxes): every set pixel lies
// inside the box, so a column changes value only in rows y0 .. y1 + 1 -- and at row 0 against the end of the column to its
// left -- and only the columns x0 .. x1 + 1 change at all.  The scan then touches the box instead of the frame.
__device__ __forceinline__ void rle_rows(const int* __restrict__ boxes, int n, int x, int H, bool& active, int& ya, int& yb) {
  active = true;
  ya = 1;
  yb = H - 1;
  if (boxes) {
    const int bx0 = boxes[n * 4], by0 = boxes[n * 4 + 1], bx1 = boxes[n * 4 + 2], by1 = boxes[n * 4 + 3];
    active = x + 3 >= bx0 && x <= bx1 + 1;
    ya = max(by0, 1);
    yb = min(by1 + 1, H - 1);
  }
}

__global__ __launch_bounds__(256) void rle_count4_kernel(const uint8_t* __restrict__ masks, const int* __restrict__ idx,
                                                         int H, int W, int* __restrict__ col_counts,
                                                         const int* __restrict__ boxes) {
  const int n = blockIdx.y;
  const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x >= W) return;
  const uint8_t* m = masks + (long)(idx ? idx[n] : n) * H * W;
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  bool active;
  int ya, yb;
  rle_rows(boxes, n, x, H, active, ya, yb);
  if (active) {
    const uint32_t last = *(const uint32_t*)(m + (long)(H - 1) * W + x);      // last row of the four columns
    uint32_t prev = (last << 8) | (x > 0 ? m[(long)(H - 1) * W + x - 1] : (m[x] != 0));   // column c starts after column c-1 ends
    // bytes are 0 / 1: compare as packed bytes.  Row 0 against the end of the previous column, then rows ya .. yb against
    // the row above (rows 1 .. ya - 1 and yb + 1 .. H - 1 repeat the row above them: outside the box)
    {
      const uint32_t v = *(const uint32_t*)(m + x);
      const uint32_t d = v ^ prev;
      c0 += d & 1u; c1 += (d >> 8) & 1u; c2 += (d >> 16) & 1u; c3 += (d >> 24) & 1u;
    }
    prev = *(const uint32_t*)(m + (long)(ya - 1) * W + x);
#pragma unroll 8
    for (int y = ya; y <= yb; ++y) {
      const uint32_t v = *(const uint32_t*)(m + (long)y * W + x);
      const uint32_t d = v ^ prev;
      c0 += d & 1u; c1 += (d >> 8) & 1u; c2 += (d >> 16) & 1u; c3 += (d >> 24) & 1u;
      prev = v;
    }
  }
  int* cc = col_counts + (long)n * W + x;
  cc[0] = c0; cc[1] = c1; cc[2] = c2; cc[3] = c3;
}

__global__ __launch_bounds__(256) void rle_write4_kernel(const uint8_t* __restrict__ masks, const int* __restrict__ idx,
                                                         int H, int W, const int* __restrict__ col_offsets,
                                                         const long* __restrict__ mask_offsets, uint32_t* __restrict__ out,
                                                         const int* __restrict__ boxes) {
  const int n = blockIdx.y;
  const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x >= W) return;
  bool active;
  int ya, yb;
  rle_rows(boxes, n, x, H, active, ya, yb);
  if (!active) return;
  const uint8_t* m = masks + (long)(idx ? idx[n] : n) * H * W;
  const int* co = col_offsets + (long)n * W + x;
  uint32_t* ob = out + mask_offsets[n];
  uint32_t* o0 = ob + co[0]; uint32_t* o1 = ob + co[1]; uint32_t* o2 = ob + co[2]; uint32_t* o3 = ob + co[3];
  const uint32_t last = *(const uint32_t*)(m + (long)(H - 1) * W + x);
  uint32_t prev = (last << 8) | (x > 0 ? m[(long)(H - 1) * W + x - 1] : (m[x] != 0));
  auto step = [&](int y) {
    const uint32_t v = *(const uint32_t*)(m + (long)y * W + x);
    const uint32_t d = v ^ prev;
    if (d) {
      if (d & 1u) *o0++ = (uint32_t)(x * H + y);
      if (d & 0x100u) *o1++ = (uint32_t)((x + 1) * H + y);
      if (d & 0x10000u) *o2++ = (uint32_t)((x + 2) * H + y);
      if (d & 0x1000000u) *o3++ = (uint32_t)((x + 3) * H + y);
    }
    prev = v;
  };
  step(0);                                              // against the end of the previous column
  prev = *(const uint32_t*)(m + (long)(ya - 1) * W + x);
#pragma unroll 8
  for (int y = ya; y <= yb; ++y) step(y);
}

__global__ __launch_bounds__(1024) void rle_scan_kernel(int* __restrict__ col_counts, int W, int* __restrict__ totals) {
  // in-place exclusive scan over the W (<= 4096) columns of mask n
  __shared__ int buf[4096];
  const int n = blockIdx.x;
  int* c = col_counts + (long)n * W;
  for (int i = threadIdx.x; i < 4096; i += 1024) buf[i] = i < W ? c[i] : 0;
  __syncthreads();
  for (int off = 1; off < 4096; off <<= 1) {
    int v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int i = threadIdx.x + k * 1024;
      v[k] = i >= off ? buf[i - off] : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) buf[threadIdx.x + k * 1024] += v[k];
    __syncthreads();
  }
  for (int i = threadIdx.x; i < W; i += 1024) c[i] = i > 0 ? buf[i - 1] : 0;
  if (threadIdx.x == 0) totals[n] = buf[W - 1];
}

__global__ __launch_bounds__(256) void rle_write_kernel(const uint8_t* __restrict__ masks, const int* __restrict__ idx,
                                                        int H, int W, const int* __restrict__ col_offsets,
                                                        const long* __restrict__ mask_offsets,
                                                        uint32_t* __restrict__ out) {
  const int n = blockIdx.y;
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= W) return;
  const uint8_t* m = masks + (long)(idx ? idx[n] : n) * H * W;
  uint32_t* o = out + mask_offsets[n] + col_offsets[(long)n * W + x];
  uint8_t prev = x > 0 ? m[(long)(H - 1) * W + x - 1] : m[x];
  for (int y = 0; y < H; ++y) {
    const uint8_t v = m[(long)y * W + x];
    if (v != prev) *o++ = (uint32_t)(x * H + y);
    prev = v;
  }
}

}  // namespace

extern "C" long csam_box_nms_workspace_bytes(int N) {
  const long nw = (N + 63) / 64;
  return (long)N * sizeof(int) + (long)N * nw * sizeof(uint64_t) + 64;
}

extern "C" int csam_box_nms(void* stream, const float* boxes, const float* scores, int N, float thr, long* out_keep,
                            int* out_count, void* workspace, long workspace_bytes) {
  CSAM_REQUIRE(boxes && scores && out_keep && out_count && workspace, "csam_box_nms: null pointer");
  CSAM_REQUIRE(N > 0 && N <= NMS_MAX, "csam_box_nms: N=%d out of range (1..%d)", N, NMS_MAX);
  if (workspace_bytes < csam_box_nms_workspace_bytes(N)) {
    csam_set_error("csam_box_nms: workspace too small");
    return CSAM_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nw = (N + 63) / 64;
  int NP = 1;
  while (NP < N) NP <<= 1;
  uint64_t* mask = (uint64_t*)workspace;
  int* order = (int*)((char*)workspace + (long)N * nw * sizeof(uint64_t));
  static csam_once_t attr_set;
  if (csam_first_call(attr_set)) {
    hipFuncSetAttribute((const void*)nms_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NMS_MAX * 8);
    hipFuncSetAttribute((const void*)nms_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 256 * 8);
  }
  hipLaunchKernelGGL(nms_sort_kernel, dim3(1), dim3(1024), NP * 8, s, scores, N, NP, order);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nw, nw), dim3(64), 0, s, boxes, order, N, nw, thr, mask);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 64 * nw * 8, s, mask, order, N, nw, out_keep, out_count);
  CSAM_LAUNCH_CHECK("csam_box_nms");
  return CSAM_OK;
}

extern "C" long csam_mask_nms_workspace_bytes(int N) {
  const long nw = (N + 63) / 64;
  return (long)N * MN_WORDS * 8 + (long)N * nw * 8 + (long)N * 4 * 2 + 256;
}

extern "C" int csam_mask_nms(void* stream, const void* masks_u8, const float* scores, int N, int H, int W, float thr,
                             long* out_keep, int* out_count, void* workspace, long workspace_bytes) {
  CSAM_REQUIRE(masks_u8 && scores && out_keep && out_count && workspace, "csam_mask_nms: null pointer");
  CSAM_REQUIRE(N > 0 && N <= NMS_MAX && H > 0 && W > 0, "csam_mask_nms: bad shape N=%d H=%d W=%d", N, H, W);
  if (workspace_bytes < csam_mask_nms_workspace_bytes(N)) {
    csam_set_error("csam_mask_nms: workspace too small");
    return CSAM_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const int nw = (N + 63) / 64;
  int NP = 1;
  while (NP < N) NP <<= 1;
  char* w = (char*)workspace;
  uint64_t* packed = (uint64_t*)w;
  w += (long)N * MN_WORDS * 8;
  uint64_t* mask = (uint64_t*)w;
  w += (long)N * nw * 8;
  int* order = (int*)w;
  int* area = order + N;
  static csam_once_t attr_set;
  if (csam_first_call(attr_set)) {
    hipFuncSetAttribute((const void*)nms_sort_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, NMS_MAX * 8);
    hipFuncSetAttribute((const void*)nms_scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 256 * 8);
  }
  hipLaunchKernelGGL(mask_pack_kernel, dim3(N), dim3(256), 0, s, (const uint8_t*)masks_u8, H, W, packed, area);
  hipLaunchKernelGGL(nms_sort_kernel, dim3(1), dim3(1024), NP * 8, s, scores, N, NP, order);
  hipLaunchKernelGGL(mask_cov_kernel, dim3(nw, nw), dim3(64), 0, s, packed, area, order, N, nw, thr, mask);
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), 64 * nw * 8, s, mask, order, N, nw, out_keep, out_count);
  CSAM_LAUNCH_CHECK("csam_mask_nms");
  return CSAM_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// COCO compressed-RLE strings on the device (amg.py:294-300 coco_encode_rle -> pycocotools' rleFrPyObjects / rleToString):
// the change positions never visit the host as integer arrays -- a frame of noise-like masks has 1e7+ of them (100+ MB of
// uint32, then three int64 numpy passes and the C packer: the 20-40 ms "host stall" of single frames in rounds 3-4).
// Run lengths of mask i: differences of  E = [0, (0 if its first pixel is set), positions..., h*w];  count k is coded as
// x = cnt[k] - (k > 2 ? cnt[k-2] : 0) in 5-bit groups, low group first, bit 5 = "more", + 48 (rleToString).
// Three passes over the counts of ALL masks back to back (1024 counts per workgroup): characters per block, scan of the block
// sums, write.  Every thread recomputes its counts from four neighbouring positions (L2-resident: just written).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int coco_nchar(long x) {
  int n = 0;
  bool more = true;
  while (more) {
    const int c = (int)(x & 0x1f);
    x >>= 5;
    more = (c & 0x10) ? (x != -1) : (x != 0);
    ++n;
  }
  return n;
}

struct CocoSrc {
  const uint32_t* pos; const long* pos_off; const uint8_t* first; const long* cnt_off; int N; long hw;
};

// mask of global count index g (cnt_off[i] <= g < cnt_off[i + 1]); cnt_off is tiny and L1 / L2 resident
__device__ __forceinline__ int coco_find_mask(const long* __restrict__ cnt_off, int N, long g) {
  int lo = 0, hi = N;                  // invariant: cnt_off[lo] <= g < cnt_off[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cnt_off[mid] <= g) lo = mid; else hi = mid;
  }
  return lo;
}

// x of count k of mask i (k < number of counts of the mask)
__device__ __forceinline__ long coco_x(const CocoSrc& s, int i, long k) {
  const long p0 = s.pos_off[i], np = s.pos_off[i + 1] - p0;
  const long f = s.first[i] ? 1 : 0;
  auto E = [&](long j) -> long {       // j = 0 .. np + 1 + f
    if (j <= f) return 0;
    const long q = j - 1 - f;
    return q < np ? (long)s.pos[p0 + q] : s.hw;
  };
  const long c = E(k + 1) - E(k);
  return k > 2 ? c - (E(k - 1) - E(k - 2)) : c;
}

__global__ __launch_bounds__(1024) void coco_cnt_off_kernel(const long* __restrict__ pos_off, const uint8_t* __restrict__ first,
                                                            int N, long* __restrict__ cnt_off) {
  __shared__ long part[1024];
  __shared__ long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < N; base += 1024) {
    const int i = base + threadIdx.x;
    const long v = i < N ? pos_off[i + 1] - pos_off[i] + 1 + (first[i] ? 1 : 0) : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const long t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < N) cnt_off[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) cnt_off[N] = carry;
}

template <bool WRITE>
__global__ __launch_bounds__(256) void coco_pack_kernel(CocoSrc s, int* __restrict__ block_sums, const long* __restrict__ block_off,
                                                        char* __restrict__ out, long out_cap, long* __restrict__ str_off) {
  __shared__ int wsum[4];
  const long total = s.cnt_off[s.N];
  const long g0 = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  long x[4];
  int nc[4], mi[4];
  long kk[4];
  int mine = 0;
  int i = g0 < total ? coco_find_mask(s.cnt_off, s.N, g0) : 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const long g = g0 + e;
    nc[e] = 0; x[e] = 0; mi[e] = -1; kk[e] = 0;
    if (g < total) {
      while (g >= s.cnt_off[i + 1]) ++i;                // consecutive counts: at most a few steps (every mask has >= 1 count)
      const long k = g - s.cnt_off[i];
      x[e] = coco_x(s, i, k);
      nc[e] = coco_nchar(x[e]);
      mi[e] = i; kk[e] = k;
      mine += nc[e];
    }
  }
  // exclusive scan of `mine` over the 256 threads: wave shuffles + 4 wave sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int wbase = 0;
  for (int w = 0; w < wave; ++w) wbase += wsum[w];
  if (!WRITE) {
    if (threadIdx.x == 255) block_sums[blockIdx.x] = wbase + incl;
    return;
  }
  long p = block_off[blockIdx.x] + wbase + incl - mine;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (mi[e] < 0) continue;
    if (kk[e] == 0) str_off[mi[e]] = p;
    long v = x[e];
    for (int c = 0; c < nc[e]; ++c) {
      int ch = (int)(v & 0x1f);
      v >>= 5;
      if (c + 1 < nc[e]) ch |= 0x20;
      if (p < out_cap) out[p] = (char)(ch + 48);
      ++p;
    }
  }
}

__global__ __launch_bounds__(1024) void coco_block_scan_kernel(const int* __restrict__ block_sums, long nblocks,
                                                               long* __restrict__ block_off, long* __restrict__ total_out) {
  __shared__ long part[1024];
  __shared__ long carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (long base = 0; base < nblocks; base += 1024) {
    const long i = base + threadIdx.x;
    const long v = i < nblocks ? block_sums[i] : 0;
    part[threadIdx.x] = v;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
      const long t = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
      __syncthreads();
      part[threadIdx.x] += t;
      __syncthreads();
    }
    if (i < nblocks) block_off[i] = carry + part[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == 1023) carry += part[1023];
    __syncthreads();
  }
  if (threadIdx.x == 0) *total_out = carry;
}

extern "C" long csam_coco_rle_pack_workspace_bytes(int N, long max_counts) {
  const long nb = (max_counts + 1023) / 1024 + 1;
  return (long)(N + 1) * 8 + nb * 4 + nb * 8 + 64;
}

extern "C" int csam_coco_rle_pack(void* stream, const uint32_t* positions, const long* pos_offsets, const uint8_t* first_pixel,
                                  int N, long hw, long max_counts, void* workspace, long workspace_bytes, char* out_chars,
                                  long out_cap, long* str_offsets) {
  CSAM_REQUIRE(positions && pos_offsets && first_pixel && workspace && out_chars && str_offsets && N > 0 && hw > 0 &&
               max_counts > 0 && out_cap > 0, "csam_coco_rle_pack: bad args");
  if (workspace_bytes < csam_coco_rle_pack_workspace_bytes(N, max_counts)) {
    csam_set_error("csam_coco_rle_pack: workspace too small");
    return CSAM_ERR_WORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  const long nb = (max_counts + 1023) / 1024;
  long* cnt_off = (long*)workspace;
  long* block_off = cnt_off + (N + 1);
  int* block_sums = (int*)(block_off + nb + 1);
  CocoSrc src{positions, pos_offsets, first_pixel, cnt_off, N, hw};
  hipLaunchKernelGGL(coco_cnt_off_kernel, dim3(1), dim3(1024), 0, s, pos_offsets, first_pixel, N, cnt_off);
  hipLaunchKernelGGL(coco_pack_kernel<false>, dim3((unsigned)nb), dim3(256), 0, s, src, block_sums, (const long*)block_off,
                     out_chars, out_cap, str_offsets);
  hipLaunchKernelGGL(coco_block_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)block_sums, nb, block_off, str_offsets + N);
  hipLaunchKernelGGL(coco_pack_kernel<true>, dim3((unsigned)nb), dim3(256), 0, s, src, block_sums, (const long*)block_off,
                     out_chars, out_cap, str_offsets);
  CSAM_LAUNCH_CHECK("csam_coco_rle_pack");
  return CSAM_OK;
}

// masks_u8[idx[i]] for i < N when idx is given (store slots of the kept masks: no gather before the encoder), else masks_u8[i]
extern "C" int csam_rle_count_box(void* stream, const void* masks_u8, const int* idx_or_null, const int* boxes_or_null, int N,
                                  int H, int W, int* col_offsets, int* totals) {
  CSAM_REQUIRE(masks_u8 && col_offsets && totals && N > 0 && H > 0 && W > 0 && W <= 4096, "csam_rle_count: bad args");
  hipStream_t s = (hipStream_t)stream;
  if (W % 4 == 0 && ((long)H * W) % 4 == 0 && ((uintptr_t)masks_u8 & 3) == 0)
    hipLaunchKernelGGL(rle_count4_kernel, dim3(csam_cdiv(W, 1024), N), dim3(256), 0, s, (const uint8_t*)masks_u8, idx_or_null, H,
                       W, col_offsets, boxes_or_null);
  else
    hipLaunchKernelGGL(rle_count_kernel, dim3(csam_cdiv(W, 256), N), dim3(256), 0, s, (const uint8_t*)masks_u8, idx_or_null, H,
                       W, col_offsets);
  hipLaunchKernelGGL(rle_scan_kernel, dim3(N), dim3(1024), 0, s, col_offsets, W, totals);
  CSAM_LAUNCH_CHECK("csam_rle_count");
  return CSAM_OK;
}

extern "C" int csam_rle_write_box(void* stream, const void* masks_u8, const int* idx_or_null, const int* boxes_or_null, int N,
                                  int H, int W, const int* col_offsets, const long* mask_offsets, uint32_t* out_positions) {
  CSAM_REQUIRE(masks_u8 && col_offsets && mask_offsets && out_positions && N > 0, "csam_rle_write: bad args");
  if (W % 4 == 0 && ((long)H * W) % 4 == 0 && ((uintptr_t)masks_u8 & 3) == 0)
    hipLaunchKernelGGL(rle_write4_kernel, dim3(csam_cdiv(W, 1024), N), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)masks_u8, idx_or_null, H, W, col_offsets, mask_offsets, out_positions, boxes_or_null);
  else
    hipLaunchKernelGGL(rle_write_kernel, dim3(csam_cdiv(W, 256), N), dim3(256), 0, (hipStream_t)stream,
                       (const uint8_t*)masks_u8, idx_or_null, H, W, col_offsets, mask_offsets, out_positions);
  CSAM_LAUNCH_CHECK("csam_rle_write");
  return CSAM_OK;
}
