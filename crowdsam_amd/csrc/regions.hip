// csam_small_regions: hole filling + island removal on device (8-connected component labelling).
//
// Reference: segment_anything_cs/utils/amg.py:267-291 remove_small_regions (cv2.connectedComponentsWithStats,
// connectivity 8) as driven by crowdsam/model.py:394-443 postprocess_small_regions: per mask, first
// mode="holes" (components of the complement smaller than area_thresh are filled), then mode="islands"
// (foreground components smaller than area_thresh are dropped; when every component is small the largest
// one — first in raster order on ties — is kept).  `changed` is set when either pass modified the mask.
// The reference runs this on the host one mask at a time; here all masks of the frame go through five
// kernels per pass with no host round trip, and the post-edit XYXY boxes (amg.py:293-324 batched_mask_to_box)
// come out of the last kernel.
//
// Labelling is union-find over pixel indices (labels only ever decrease, roots are the raster-first pixel
// of a component, so "first label on ties" == smallest root), built in two levels (round 2):
//   tile     a 64 x 64 tile is labelled in LDS: a pixel's initial parent is the start of its horizontal run (ballot +
//            bit scan), north links come from the row bit masks with the usual decision tree (a pixel whose west
//            neighbour is set only needs NE when N is clear), LDS union-find, then every pixel is written out pointing at
//            its TILE root (global index);
//   border   the links that cross a tile edge, ONE union per contact run (the background of a frame would otherwise send
//            64 x 3 unions per edge at one root), on the global forest of tile roots;
//   count    per tile: pixels counted per tile root in LDS (wave-aggregated), then only the tile roots walk the global
//            forest, add their count to their global root's area and are re-pointed straight at it;
//   flags    per-root: any small / any big / arg-max area (64-bit atomicMax of area:~root);
//   apply    rewrites the mask (pixel -> tile root -> global root);  extent: box extents of the final masks.
// 3 % of the pixels touch global atomics; on the noise-like masks of the benchmark's crowded-tail leg the two passes over 355
// masks take 9.9 ms instead of 17.1 (CSAM_CC_TILE=0 runs the one-level form: init / merge / count kernels per pixel).
#include "csam_common.h"
#include <algorithm>

namespace {

struct RegionMeta {        // one per mask per pass
  int any_small;
  int any_big;
  unsigned long long best;   // (area << 32) | (0xffffffff - root)
};

__device__ __forceinline__ int uf_load(const int* L, int i) { return __atomic_load_n(L + i, __ATOMIC_RELAXED); }

__device__ __forceinline__ int uf_find(const int* L, int i) {
  int p;
  while ((p = uf_load(L, i)) != i) i = p;
  return i;
}

// find with path halving.  The shortcut store may race with other shortcuts / stale atomicMins, but any value it
// writes is an ancestor of i at some earlier time (same component, smaller index), which keeps the forest valid;
// root-time links (the only ones that define components) are never written by it because it never touches a root.
__device__ __forceinline__ int uf_find_halve(int* L, int i) {
  int p = uf_load(L, i);
  while (p != i) {
    const int g = uf_load(L, p);
    if (g == p) return p;
    __atomic_store_n(L + i, g, __ATOMIC_RELAXED);
    i = g;
    p = uf_load(L, i);
  }
  return i;
}

__device__ __forceinline__ void uf_union(int* L, int a, int b) {
  bool done;
  do {
    a = uf_find_halve(L, a);
    b = uf_find_halve(L, b);
    if (a < b) {
      const int old = atomicMin(L + b, a);
      done = (old == b);
      b = old;
    } else if (b < a) {
      const int old = atomicMin(L + a, b);
      done = (old == a);
      a = old;
    } else {
      done = true;
    }
  } while (!done);
}

// Row-segment coordinates shared by all kernels: wave -> (y, x0), lane -> x.
struct Seg {
  int y, x, lane;
  bool in;
};
__device__ __forceinline__ Seg seg_coords(int H, int W) {
  const int sw = (W + 63) >> 6;
  const int seg = blockIdx.x * 4 + (threadIdx.x >> 6);
  Seg s;
  s.lane = threadIdx.x & 63;
  s.y = seg / sw;
  s.x = (seg - s.y * sw) * 64 + s.lane;
  s.in = (s.y < H) && (s.x < W);
  return s;
}

template <int HOLES>
__device__ __forceinline__ bool work_at(const uint8_t* m, int W, int y, int x) {
  return HOLES ? (m[(long)y * W + x] == 0) : (m[(long)y * W + x] != 0);
}

template <int HOLES>
__global__ __launch_bounds__(256) void cc_init_kernel(const uint8_t* __restrict__ masks, int* __restrict__ L,
                                                      int* __restrict__ S, int H, int W) {
  const long base = (long)blockIdx.y * H * W;
  const Seg s = seg_coords(H, W);
  const bool wk = s.in && work_at<HOLES>(masks + base, W, s.y, s.x);
  const unsigned long long bits = __ballot(wk);
  if (!s.in) return;
  const int p = s.y * W + s.x;
  int lab = -1;
  if (wk) {
    const unsigned long long below = (s.lane == 0) ? 0ull : (~bits & ((1ull << s.lane) - 1ull));
    const int start = below ? (64 - __clzll(below)) : 0;
    lab = p - s.lane + start;
  }
  L[base + p] = lab;
  S[base + p] = 0;
}

template <int HOLES>
__global__ __launch_bounds__(256) void cc_merge_kernel(const uint8_t* __restrict__ masks, int* __restrict__ L,
                                                       int H, int W) {
  const long base = (long)blockIdx.y * H * W;
  const Seg s = seg_coords(H, W);
  if (!s.in) return;
  const uint8_t* m = masks + base;
  if (!work_at<HOLES>(m, W, s.y, s.x)) return;
  int* Lm = L + base;
  const int p = s.y * W + s.x;
  const bool w = s.x > 0 && work_at<HOLES>(m, W, s.y, s.x - 1);
  if (w && s.lane == 0) uf_union(Lm, p, p - 1);
  if (s.y == 0) return;
  const bool b = work_at<HOLES>(m, W, s.y - 1, s.x);
  const bool c = s.x + 1 < W && work_at<HOLES>(m, W, s.y - 1, s.x + 1);
  if (w) {
    if (c && !b) uf_union(Lm, p, p - W + 1);
    return;
  }
  if (b) {
    uf_union(Lm, p, p - W);
    return;
  }
  const bool a = s.x > 0 && work_at<HOLES>(m, W, s.y - 1, s.x - 1);
  if (a) uf_union(Lm, p, p - W - 1);
  if (c) uf_union(Lm, p, p - W + 1);
}

// ---- tile form of init + merge.  A 64 x 64 tile is labelled in LDS (run starts from ballots, north links from the row
// bit masks, union-find on a 16 KB LDS array), compressed, and written out as "global index of the tile-local root"; the
// links that cross a tile edge (west at column 0, north at row 0, the two diagonals at columns 0 / 63 and row 0) are made
// by cc_border_kernel on that forest with the global atomics of cc_merge_kernel.  Same components, same roots (the
// raster-first pixel: raster order inside a tile is global order), 3 % of the pixels touch global atomics instead of all.
__device__ __forceinline__ int lds_find(int* P, int i) {
  int p;
  while ((p = __atomic_load_n(P + i, __ATOMIC_RELAXED)) != i) i = p;
  return i;
}
// find with path halving (see uf_find_halve for why the racy shortcut stores are safe)
__device__ __forceinline__ int lds_find_halve(int* P, int i) {
  int p = __atomic_load_n(P + i, __ATOMIC_RELAXED);
  while (p != i) {
    const int g = __atomic_load_n(P + p, __ATOMIC_RELAXED);
    if (g == p) return p;
    __atomic_store_n(P + i, g, __ATOMIC_RELAXED);
    i = g;
    p = __atomic_load_n(P + i, __ATOMIC_RELAXED);
  }
  return i;
}
__device__ __forceinline__ void lds_union(int* P, int a, int b) {
  bool done;
  do {
    a = lds_find_halve(P, a);
    b = lds_find_halve(P, b);
    if (a < b) {
      const int old = atomicMin(P + b, a);
      done = (old == b);
      b = old;
    } else if (b < a) {
      const int old = atomicMin(P + a, b);
      done = (old == a);
      a = old;
    } else {
      done = true;
    }
  } while (!done);
}

template <int HOLES>
__global__ __launch_bounds__(256) void cc_tile_kernel(const uint8_t* __restrict__ masks, int* __restrict__ L,
                                                      int* __restrict__ S, int H, int W) {
  __shared__ unsigned long long rowbits[64];
  __shared__ int P[64 * 64];
  const long base = (long)blockIdx.y * H * W;
  const int sw = (W + 63) >> 6;
  const int ty = blockIdx.x / sw, tx = blockIdx.x - ty * sw;
  const int x0 = tx * 64, y0 = ty * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = x0 + lane;
  const uint8_t* m = masks + base;
  // runs inside the row segment: parent = start of the horizontal run (as cc_init_kernel)
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = wave * 16 + i, y = y0 + r;
    const bool wk = y < H && x < W && work_at<HOLES>(m, W, y, x);
    const unsigned long long bits = __ballot(wk);
    if (lane == 0) rowbits[r] = bits;
    int lab = -1;
    if (wk) {
      const unsigned long long below = (lane == 0) ? 0ull : (~bits & ((1ull << lane) - 1ull));
      const int start = below ? (64 - __clzll(below)) : 0;
      lab = r * 64 + start;
    }
    P[r * 64 + lane] = lab;
  }
  __syncthreads();
  // north links inside the tile, cc_merge_kernel's decision tree on the row bit masks
  for (int i = 0; i < 16; ++i) {
    const int r = wave * 16 + i;
    if (r == 0) continue;
    const unsigned long long cur = rowbits[r], up = rowbits[r - 1];
    if (!((cur >> lane) & 1ull)) continue;
    const bool w = lane > 0 && ((cur >> (lane - 1)) & 1ull);
    const bool b = (up >> lane) & 1ull;
    const bool c = lane < 63 && ((up >> (lane + 1)) & 1ull);
    const int p = r * 64 + lane;
    if (w) {
      if (c && !b) lds_union(P, p, p - 63);
      continue;
    }
    if (b) {
      lds_union(P, p, p - 64);
      continue;
    }
    const bool a = lane > 0 && ((up >> (lane - 1)) & 1ull);
    if (a) lds_union(P, p, p - 65);
    if (c) lds_union(P, p, p - 63);
  }
  __syncthreads();
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = wave * 16 + i, y = y0 + r;
    if (y >= H || x >= W) continue;
    int g = -1;
    if (P[r * 64 + lane] >= 0) {
      const int root = lds_find_halve(P, r * 64 + lane);
      g = (y0 + (root >> 6)) * W + x0 + (root & 63);
    }
    const long q = base + (long)y * W + x;
    L[q] = g;
    S[q] = 0;
  }
}

// Links across tile edges, ONE union per contact run: where a horizontal run of the row below a tile edge touches a run of
// the row above it over k consecutive columns, the k N-links (and the diagonals next to them) all join the same two tile
// components -- the background of a frame would otherwise send 64 x 3 unions per edge at one root -- so only the first
// column of a contact run links; the same along vertical edges (first row of a vertical contact run).  A diagonal links
// only where no straight contact next to it already does the job.
template <int HOLES>
__global__ __launch_bounds__(256) void cc_border_kernel(const uint8_t* __restrict__ masks, int* __restrict__ L, int H,
                                                        int W) {
  const long base = (long)blockIdx.y * H * W;
  const uint8_t* m = masks + base;
  int* Lm = L + base;
  auto at = [&](int y, int x) -> bool { return y >= 0 && x >= 0 && x < W && y < H && work_at<HOLES>(m, W, y, x); };
  // two waves per tile: the first walks the tile's first row (lane = column), the second its first / last column
  // (lane = row)
  const int sw = (W + 63) >> 6, tiles = ((H + 63) >> 6) * sw;
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (wv >= 2 * tiles) return;
  const int tile = wv < tiles ? wv : wv - tiles;
  const int ty = tile / sw, tx = tile - ty * sw;
  const int x0 = tx * 64, y0 = ty * 64;
  if (wv < tiles) {                                     // the row below a horizontal tile edge (or row 0)
    const int y = y0, x = x0 + lane;
    const bool in = x < W;
    const bool cur = in && at(y, x);
    const bool up = in && at(y - 1, x);
    const unsigned long long cb = __ballot(cur), ub = __ballot(up);
    if (!cur) return;
    const int p = y * W + x;
    if (lane == 0 && at(y, x - 1)) uf_union(Lm, p, p - 1);                   // W across the vertical edge
    if (y == 0) return;
    const unsigned long long cn = cb & ub;                                   // straight (N) contacts of this segment
    const bool n_here = (cn >> lane) & 1ull;
    const bool n_left = lane > 0 && ((cn >> (lane - 1)) & 1ull);
    const bool n_right = lane < 63 && ((cn >> (lane + 1)) & 1ull);
    if (n_here) {
      if (!n_left) uf_union(Lm, p, p - W);
      return;
    }
    const bool upl = lane > 0 ? ((ub >> (lane - 1)) & 1ull) : at(y - 1, x0 - 1);
    const bool upr = lane < 63 ? ((ub >> (lane + 1)) & 1ull) : at(y - 1, x0 + 64);
    if (upl && !n_left) uf_union(Lm, p, p - W - 1);
    if (upr && !n_right) uf_union(Lm, p, p - W + 1);
    return;
  }
  // other rows of the tile: only the columns next to a vertical tile edge
  const int y = y0 + lane;
  if (lane == 0 || y >= H) return;
  if (at(y, x0)) {                                      // first column: W / NW into the tile on the left
    const int p = y * W + x0;
    const bool n = at(y - 1, x0);                       // same tile: linked locally
    if (at(y, x0 - 1)) {
      if (!(n && at(y - 1, x0 - 1))) uf_union(Lm, p, p - 1);                // first row of a vertical W-contact run
    } else if (!n && at(y - 1, x0 - 1)) {
      uf_union(Lm, p, p - W - 1);                                           // NW only where neither W nor N carries it
    }
  }
  const int xr = x0 + 63;                               // last column: NE into the tile on the right
  if (xr + 1 < W && at(y, xr) && !at(y - 1, xr) && !at(y, xr + 1) && at(y - 1, xr + 1)) {
    const int p = y * W + xr;
    uf_union(Lm, p, p - W + 1);
  }
}

// Areas on the tile forest.  After cc_tile_kernel every pixel points at its TILE root, so a tile's pixels are counted per
// tile root in LDS (wave-aggregated), and only the tile roots walk the global forest: each adds its count to its global
// root's area and is re-pointed straight at it.  Pixels keep their tile root; cc_apply_kernel takes the second hop.
__global__ __launch_bounds__(256) void cc_count_tile_kernel(int* __restrict__ L, int* __restrict__ S, int H, int W) {
  __shared__ int cnt[64 * 64];
  const long base = (long)blockIdx.y * H * W;
  const int sw = (W + 63) >> 6;
  const int ty = blockIdx.x / sw, tx = blockIdx.x - ty * sw;
  const int x0 = tx * 64, y0 = ty * 64, origin = y0 * W + x0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 64 * 64; i += 256) cnt[i] = 0;
  __syncthreads();
  const float inv_w = 1.0f / (float)W;
  const int x = x0 + lane;
  for (int i = 0; i < 16; ++i) {
    const int y = y0 + wave * 16 + i;
    int r = -1;
    if (y < H && x < W) r = L[base + (long)y * W + x];
    unsigned long long active = __ballot(r >= 0);
    while (active) {
      const int leader = __ffsll((long long)active) - 1;
      const int rl = __shfl(r, leader);
      const unsigned long long same = __ballot(r == rl);
      if (lane == leader) {
        // cc_border_kernel's path halving may have re-pointed an edge pixel at a root in ANOTHER tile: those go to the
        // global forest directly (a few per tile edge)
        const int d = rl - origin;
        int row = -1, col = -1;
        if (d >= 0 && d < 64 * W) {                     // < 2^24: the float quotient is off by at most one
          row = (int)((float)d * inv_w);
          if (row * W > d) --row;
          if ((row + 1) * W <= d) ++row;
          col = d - row * W;
        }
        if (row >= 0 && row < 64 && col >= 0 && col < 64) {
          atomicAdd(&cnt[row * 64 + col], (int)__popcll(same));
        } else {
          atomicAdd(S + base + uf_find(L + base, rl), (int)__popcll(same));
        }
      }
      active &= ~same;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * 64; i += 256) {
    const int c = cnt[i];
    if (c == 0) continue;
    const int q = (y0 + (i >> 6)) * W + x0 + (i & 63);
    const int g = uf_find(L + base, q);
    if (g != q) L[base + q] = g;
    atomicAdd(S + base + g, c);
  }
}

// Path compression + per-root area.  A wave walks CC_ROWS vertically consecutive row segments and keeps a running
// (root, count) pair, so a solid region costs one atomic per 64 x CC_ROWS pixels instead of one per segment.
constexpr int CC_ROWS = 16;
__global__ __launch_bounds__(256) void cc_count_kernel(int* __restrict__ L, int* __restrict__ S, int H, int W) {
  const long base = (long)blockIdx.y * H * W;
  const int sw = (W + 63) >> 6;
  const int col = blockIdx.x * 4 + (threadIdx.x >> 6);     // (row band, x segment)
  const int lane = threadIdx.x & 63;
  const int band = col / sw;
  const int x = (col - band * sw) * 64 + lane;
  int cur = -1, cnt = 0;
  for (int i = 0; i < CC_ROWS; ++i) {
    const int y = band * CC_ROWS + i;
    int r = -1;
    if (y < H && x < W) {
      const int p = y * W + x;
      if (uf_load(L + base, p) >= 0) {
        r = uf_find(L + base, p);
        L[base + p] = r;
      }
    }
    unsigned long long active = __ballot(r >= 0);
    while (active) {
      const int leader = __ffsll((long long)active) - 1;
      const int rl = __shfl(r, leader);
      const unsigned long long same = __ballot(r == rl);
      const int k = (int)__popcll(same);
      if (rl == cur) {
        cnt += k;
      } else {
        if (cur >= 0 && lane == 0) atomicAdd(S + base + cur, cnt);
        cur = rl;
        cnt = k;
      }
      active &= ~same;
    }
  }
  if (cur >= 0 && lane == 0) atomicAdd(S + base + cur, cnt);
}

// Per-mask summary of the roots: any small / any big / arg-max area.  Same-address atomics from every wave that
// holds a root serialise in L2 (masks with ragged borders have 1e4+ roots), so this is a chunked reduction:
// grid (chunks, n), registers -> wave -> block, then at most three atomics per block.
__global__ __launch_bounds__(256) void cc_flags_kernel(const int* __restrict__ L, const int* __restrict__ S,
                                                       RegionMeta* __restrict__ meta, int H, int W, int thresh) {
  const long base = (long)blockIdx.y * H * W;
  const int sw = (W + 63) >> 6;
  const int segs = H * sw;
  const int per = (segs + gridDim.x - 1) / gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s0 = blockIdx.x * per, s1 = min(segs, s0 + per);
  unsigned long long key = 0ull;
  int flags = 0;                       // bit 0: small root seen, bit 1: big root seen
  for (int sg = s0 + wave; sg < s1; sg += 4) {
    const int y = sg / sw;
    const int x = (sg - y * sw) * 64 + lane;
    if (x >= W) continue;
    const int p = y * W + x;
    if (L[base + p] != p) continue;
    const int area = S[base + p];
    flags |= area < thresh ? 1 : 2;
    const unsigned long long k = ((unsigned long long)(unsigned)area << 32) | (0xffffffffu - (unsigned)p);
    key = k > key ? k : key;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_xor(key, off);
    key = o > key ? o : key;
    flags |= __shfl_xor(flags, off);
  }
  __shared__ unsigned long long rkey[4];
  __shared__ int rflags[4];
  if (lane == 0) { rkey[wave] = key; rflags[wave] = flags; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  for (int w = 1; w < 4; ++w) {
    key = rkey[w] > key ? rkey[w] : key;
    flags |= rflags[w];
  }
  RegionMeta* mm = meta + blockIdx.y;
  if (flags & 1) atomicOr(&mm->any_small, 1);
  if (flags & 2) atomicOr(&mm->any_big, 1);
  if (key) atomicMax(&mm->best, key);
}

// HOLES: out = mask | (complement component smaller than thresh).
// ISLANDS: out = mask & (component not small), or only the arg-max component when all are small.
template <int HOLES>
__global__ __launch_bounds__(256) void cc_apply_kernel(const uint8_t* in, uint8_t* out,
                                                       const int* __restrict__ L, const int* __restrict__ S,
                                                       const RegionMeta* __restrict__ meta, int H, int W,
                                                       int thresh) {
  const long base = (long)blockIdx.y * H * W;
  const Seg s = seg_coords(H, W);
  const RegionMeta mm = meta[blockIdx.y];
  bool v = false;
  if (s.in) {
    const int p = s.y * W + s.x;
    v = in[base + p] != 0;
    if (mm.any_small) {
      int r = L[base + p];
      // tile root -> global root: one more hop for almost every pixel (cc_count_tile_kernel points the counted tile roots
      // straight at their global root); a tile root nobody of its own tile points at any more keeps its chain
      if (r >= 0) r = uf_find(L + base, r);
      if (HOLES) {
        if (r >= 0 && S[base + r] < thresh) v = true;
      } else if (r >= 0) {
        if (mm.any_big) v = S[base + r] >= thresh;
        else v = (unsigned)r == 0xffffffffu - (unsigned)(mm.best & 0xffffffffull);
      }
    }
    out[base + p] = v ? 1 : 0;
  }
}

// Box extents of the final masks: grid (chunks, n); a block reduces its share of the row segments in
// registers and issues at most four atomics.
__global__ __launch_bounds__(256) void cc_extent_kernel(const uint8_t* __restrict__ masks, int* __restrict__ ext,
                                                        int H, int W) {
  const long base = (long)blockIdx.y * H * W;
  const int sw = (W + 63) >> 6;
  const int segs = H * sw;
  const int per = (segs + gridDim.x - 1) / gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s0 = blockIdx.x * per, s1 = min(segs, s0 + per);
  int x0 = W, y0 = H, x1 = -1, y1 = -1;
  for (int sg = s0 + wave; sg < s1; sg += 4) {
    const int y = sg / sw;
    const int x = (sg - y * sw) * 64 + lane;
    if (x < W && masks[base + (long)y * W + x]) {
      x0 = min(x0, x); x1 = max(x1, x);
      y0 = min(y0, y); y1 = max(y1, y);
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    x0 = min(x0, __shfl_xor(x0, off)); y0 = min(y0, __shfl_xor(y0, off));
    x1 = max(x1, __shfl_xor(x1, off)); y1 = max(y1, __shfl_xor(y1, off));
  }
  __shared__ int red[4][4];
  if (lane == 0) { red[wave][0] = x0; red[wave][1] = y0; red[wave][2] = x1; red[wave][3] = y1; }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; ++w) {
      x0 = min(x0, red[w][0]); y0 = min(y0, red[w][1]);
      x1 = max(x1, red[w][2]); y1 = max(y1, red[w][3]);
    }
    if (x1 >= 0) {
      int* e = ext + (long)blockIdx.y * 4;
      atomicMin(e + 0, x0); atomicMin(e + 1, y0);
      atomicMax(e + 2, x1); atomicMax(e + 3, y1);
    }
  }
}

__global__ void cc_prepare_kernel(RegionMeta* meta, int* ext, int n, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * n) {
    meta[i].any_small = 0;
    meta[i].any_big = 0;
    meta[i].best = 0ull;
  }
  if (i < n) {
    ext[i * 4 + 0] = W;
    ext[i * 4 + 1] = H;
    ext[i * 4 + 2] = -1;
    ext[i * 4 + 3] = -1;
  }
}

__global__ void cc_finish_kernel(const RegionMeta* meta, const int* ext, int n, int* changed, float* boxes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  changed[i] = (meta[i].any_small | meta[n + i].any_small) ? 1 : 0;
  const int* e = ext + i * 4;
  const bool empty = e[2] < e[0] || e[3] < e[1];
  for (int k = 0; k < 4; ++k) boxes[i * 4 + k] = empty ? 0.f : (float)e[k];
}

// fuse_simmap (crowdsam/model.py:273-286): mean over a mask's pixels of the prompt-prior map bilinearly resized
// (align_corners = False) from [fh,fw] to the frame.  The resized map is never materialised: each set pixel
// samples the small map (L1/L2 resident); partial sums in float64, one atomic pair per block.
__global__ __launch_bounds__(256) void mask_mean_bilinear_kernel(const uint8_t* __restrict__ masks,
                                                                 const float* __restrict__ sim, int fh, int fw, int lds,
                                                                 int H, int W, double* __restrict__ sum,
                                                                 int* __restrict__ cnt) {
  const long base = (long)blockIdx.y * H * W;
  const int sw = (W + 63) >> 6;
  const int segs = H * sw;
  const int per = (segs + gridDim.x - 1) / gridDim.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int s0 = blockIdx.x * per, s1 = min(segs, s0 + per);
  const float sy = (float)fh / (float)H, sx = (float)fw / (float)W;
  double acc = 0.0;
  int n = 0;
  for (int sg = s0 + wave; sg < s1; sg += 4) {
    const int y = sg / sw;
    const int x = (sg - y * sw) * 64 + lane;
    if (x < W && masks[base + (long)y * W + x]) {
      const float fy = fmaxf(sy * ((float)y + 0.5f) - 0.5f, 0.f), fx = fmaxf(sx * ((float)x + 0.5f) - 0.5f, 0.f);
      const int y0 = min((int)fy, fh - 1), x0 = min((int)fx, fw - 1);
      const int y1 = min(y0 + 1, fh - 1), x1 = min(x0 + 1, fw - 1);
      const float ly = fy - (float)y0, lx = fx - (float)x0;
      const float top = sim[y0 * lds + x0] * (1.f - lx) + sim[y0 * lds + x1] * lx;
      const float bot = sim[y1 * lds + x0] * (1.f - lx) + sim[y1 * lds + x1] * lx;
      acc += (double)(top * (1.f - ly) + bot * ly);
      ++n;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    acc += __shfl_xor(acc, off);
    n += __shfl_xor(n, off);
  }
  __shared__ double racc[4];
  __shared__ int rn[4];
  if (lane == 0) { racc[wave] = acc; rn[wave] = n; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tn = rn[0] + rn[1] + rn[2] + rn[3];
    if (tn) {
      atomicAdd(sum + blockIdx.y, (racc[0] + racc[1]) + (racc[2] + racc[3]));
      atomicAdd(cnt + blockIdx.y, tn);
    }
  }
}

inline long align256(long v) { return (v + 255) & ~255L; }

}  // namespace

extern "C" int csam_mask_mean_bilinear(void* stream, const uint8_t* masks, int n, int H, int W, const float* sim, int fh,
                                       int fw, int ld_sim, double* sum, int* count) {
  CSAM_REQUIRE(masks && sim && sum && count && n > 0 && n <= 65535 && H > 0 && W > 0 && fh > 0 && fw > 0 && ld_sim >= fw,
               "csam_mask_mean_bilinear: bad args");
  hipStream_t s = (hipStream_t)stream;
  hipMemsetAsync(sum, 0, (size_t)n * sizeof(double), s);
  hipMemsetAsync(count, 0, (size_t)n * sizeof(int), s);
  const int segs = H * ((W + 63) >> 6);
  const dim3 grid(std::min(std::min(std::max(1024 / n, 16), 256), std::max(1, segs / 8)), n);
  hipLaunchKernelGGL(mask_mean_bilinear_kernel, grid, dim3(256), 0, s, masks, sim, fh, fw, ld_sim, H, W, sum, count);
  CSAM_LAUNCH_CHECK("csam_mask_mean_bilinear");
  return CSAM_OK;
}

extern "C" long csam_small_regions_workspace_bytes(int n, int H, int W) {
  if (n <= 0 || H <= 0 || W <= 0) return 0;
  const long px = (long)n * H * W;
  return 2 * align256(px * 4) + align256((long)2 * n * sizeof(RegionMeta)) + align256((long)n * 16);
}

extern "C" int csam_small_regions(void* stream_, const uint8_t* masks, uint8_t* out, int* changed, float* boxes,
                                  int n, int H, int W, int min_area, void* ws, long ws_bytes) {
  CSAM_REQUIRE(masks && out && changed && boxes && ws, "csam_small_regions: null pointer");
  CSAM_REQUIRE(n > 0 && H > 0 && W > 0 && (long)H * W < (1L << 31), "csam_small_regions: bad shape");
  CSAM_REQUIRE(n <= 65535, "csam_small_regions: at most 65535 masks per call");
  CSAM_REQUIRE(ws_bytes >= csam_small_regions_workspace_bytes(n, H, W), "csam_small_regions: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  const long px = (long)n * H * W;
  char* w = (char*)ws;
  int* L = (int*)w;
  w += align256(px * 4);
  int* S = (int*)w;
  w += align256(px * 4);
  RegionMeta* meta = (RegionMeta*)w;
  w += align256((long)2 * n * sizeof(RegionMeta));
  int* ext = (int*)w;

  const int segs = H * ((W + 63) >> 6);
  const dim3 grid(csam_cdiv(segs, 4), n), block(256);
  // chunked reductions (flags, extents): enough blocks to fill the chip whatever n is
  const dim3 rgrid(std::min(std::min(std::max(1024 / n, 16), 256), std::max(1, segs / 8)), n);
  const dim3 cgrid(csam_cdiv(csam_cdiv(H, CC_ROWS) * ((W + 63) >> 6), 4), n);
  static int tiled = -1;
  if (tiled < 0) {
    const char* e = getenv("CSAM_CC_TILE");            // 0: per-pixel global union-find (init + merge kernels), for A/B
    tiled = e ? atoi(e) : 1;
  }
  const dim3 tgrid(csam_cdiv(H, 64) * ((W + 63) >> 6), n);
  const dim3 bgrid(csam_cdiv(2 * (int)tgrid.x, 4), n);
  hipLaunchKernelGGL(cc_prepare_kernel, dim3(csam_cdiv(2 * n, 256)), dim3(256), 0, stream, meta, ext, n, H, W);
  // pass 1: holes (components of the complement)
  if (tiled) {
    hipLaunchKernelGGL(cc_tile_kernel<1>, tgrid, block, 0, stream, masks, L, S, H, W);
    hipLaunchKernelGGL(cc_border_kernel<1>, bgrid, block, 0, stream, masks, L, H, W);
    if (tiled == 2) hipLaunchKernelGGL(cc_count_kernel, cgrid, block, 0, stream, L, S, H, W);
    else hipLaunchKernelGGL(cc_count_tile_kernel, tgrid, block, 0, stream, L, S, H, W);
  } else {
    hipLaunchKernelGGL(cc_init_kernel<1>, grid, block, 0, stream, masks, L, S, H, W);
    hipLaunchKernelGGL(cc_merge_kernel<1>, grid, block, 0, stream, masks, L, H, W);
    hipLaunchKernelGGL(cc_count_kernel, cgrid, block, 0, stream, L, S, H, W);
  }
  hipLaunchKernelGGL(cc_flags_kernel, rgrid, block, 0, stream, L, S, meta, H, W, min_area);
  hipLaunchKernelGGL(cc_apply_kernel<1>, grid, block, 0, stream, masks, out, L, S, meta, H, W, min_area);
  // pass 2: islands (components of the hole-filled mask)
  if (tiled) {
    hipLaunchKernelGGL(cc_tile_kernel<0>, tgrid, block, 0, stream, out, L, S, H, W);
    hipLaunchKernelGGL(cc_border_kernel<0>, bgrid, block, 0, stream, out, L, H, W);
    if (tiled == 2) hipLaunchKernelGGL(cc_count_kernel, cgrid, block, 0, stream, L, S, H, W);
    else hipLaunchKernelGGL(cc_count_tile_kernel, tgrid, block, 0, stream, L, S, H, W);
  } else {
    hipLaunchKernelGGL(cc_init_kernel<0>, grid, block, 0, stream, out, L, S, H, W);
    hipLaunchKernelGGL(cc_merge_kernel<0>, grid, block, 0, stream, out, L, H, W);
    hipLaunchKernelGGL(cc_count_kernel, cgrid, block, 0, stream, L, S, H, W);
  }
  hipLaunchKernelGGL(cc_flags_kernel, rgrid, block, 0, stream, L, S, meta + n, H, W, min_area);
  hipLaunchKernelGGL(cc_apply_kernel<0>, grid, block, 0, stream, out, out, L, S, meta + n, H, W, min_area);
  hipLaunchKernelGGL(cc_extent_kernel, rgrid, block, 0, stream, out, ext, H, W);
  hipLaunchKernelGGL(cc_finish_kernel, dim3(csam_cdiv(n, 256)), dim3(256), 0, stream, meta, ext, n, changed, boxes);
  CSAM_LAUNCH_CHECK("csam_small_regions");
  return CSAM_OK;
}
