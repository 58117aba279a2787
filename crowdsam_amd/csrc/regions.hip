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

// =====================================================================================================================
// Round 3: the same labelling WITHOUT per-pixel label arrays ("compact" form, csam_small_regions_idx).
// The two-level form above still keeps two int32 values per pixel (tile-root pointer L, area S): 8 B written and ~20 B
// read per mask byte and pass -- 355 one-megapixel masks move ~14 GB for a 0.36 GB input, and that traffic, not the
// labelling, is what its 10 ms are.  Only components that reach a tile edge can continue in another tile, so only the
// 252 RING pixels of a 64 x 64 tile need an identity outside the tile's LDS:
//   scan    label the tile in LDS (same run-start / north-link union-find), count every tile component, give each one
//           that owns ring pixels its smallest ring slot as representative; write per ring slot (256 per tile) its
//           parent slot RP, and at the representative the component's area inside the tile RA and its raster-first pixel
//           RF.  Components without ring pixels are complete: their small / big / arg-max flags go to the mask's meta now;
//   border  the cross-edge links of cc_border_kernel (one union per contact run) on the ring forest RP;
//   total   every representative adds its area to its global root (RT) and min-reduces the first pixel (RFT);
//   ringmeta  flags of the ring components (global roots) -> meta;
//   apply   no labelling: a pixel's tile root comes from a 2-byte-per-pixel array the scan left behind (R16), the tile's
//           root pixels resolve their component's area (own count, or RT of their representative's global root) into a
//           4 KB decision table, the tile's bytes are rewritten in place; the islands pass also reduces the box extents of
//           what it writes.
// Per pass: mask bytes read twice and written once, 2 B per pixel of roots written and read, 5 x 4 B per ring slot
// (1/16 of a pixel): ~8 B per pixel instead of ~28.
// Masks are addressed through an index list (store slots), so the NMS survivors are cleaned up where they lie.
// =====================================================================================================================
constexpr int C2_INF = 0x7fffffff;
constexpr unsigned short C2_TILE_EMPTY = 0xfffeu, C2_TILE_FULL = 0xfffdu;   // R16[tile][0] markers of trivial tiles
constexpr int C2_LDS = 64 * 8 + 4096 * 4 + 3 * 256 * 4; // row bit masks | P (parents, then rep : count per root) | ring records

__device__ __forceinline__ int c2_ring_slot(int r, int c) { return r == 0 ? c : r == 63 ? 64 + c : c == 0 ? 128 + r : 192 + r; }
// ring slot t -> tile-local (r, c); slots 128, 191, 192, 255 duplicate corner pixels owned by the row slots
__device__ __forceinline__ bool c2_slot_rc(int t, int& r, int& c) {
  if (t < 64) { r = 0; c = t; return true; }
  if (t < 128) { r = 63; c = t - 64; return true; }
  if (t < 192) { r = t - 128; c = 0; return r != 0 && r != 63; }
  r = t - 192; c = 63;
  return r != 0 && r != 63;
}
// start column of the horizontal run that contains `lane` in a row whose work bits are `bits` (lane is a work pixel)
__device__ __forceinline__ int c2_run_start(unsigned long long bits, int lane) {
  const unsigned long long below = (lane == 0) ? 0ull : (~bits & ((1ull << lane) - 1ull));
  return below ? (64 - __clzll(below)) : 0;
}

__device__ __forceinline__ void c2_meta_reduce(unsigned long long key, int flags, RegionMeta* mm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
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
  if (flags & 1) atomicOr(&mm->any_small, 1);
  if (flags & 2) atomicOr(&mm->any_big, 1);
  if (key) atomicMax(&mm->best, key);
}

// scan: labels one 64 x 64 tile in LDS and leaves, in global memory,
//   R16[tile][pixel]  0xffff: not a work pixel; < 0x1000: the pixel's tile root (tile-local index r*64 + c of the component's
//                     raster-first pixel); at the root pixel itself the component's record instead: 0x8000 | area inside the
//                     tile (complete component), or 0xC000 | representative ring slot
//   RP / RA / RF / RT / RFT[tile][ring slot]   the ring forest (see above)
// A pixel's first parent is the start of its horizontal run, so only run starts walk the forest; their roots stay in
// registers and reach the other pixels of the run by a wave shuffle (a row is one wave's iteration), which frees the
// parent array: the same 16 KB then hold the per-root (representative : area) words.  16.5 KB of LDS per workgroup keeps
// eight workgroups on a CU -- the union-find is a chain of dependent LDS atomics and lives on occupancy.
template <int HOLES>
__global__ __launch_bounds__(256, 8) void cc2_scan_kernel(const uint8_t* __restrict__ base, const int* __restrict__ idx,
                                                       unsigned short* __restrict__ R16,
                                                       int* __restrict__ RP, int* __restrict__ RA, int* __restrict__ RF,
                                                       int* __restrict__ RT, int* __restrict__ RFT,
                                                       RegionMeta* __restrict__ meta, int H, int W, int thresh, int no_trivial) {
  extern __shared__ __attribute__((aligned(16))) char c2smem[];
  unsigned long long* rowbits = (unsigned long long*)c2smem;
  int* P = (int*)(c2smem + 512);
  unsigned* RC = (unsigned*)P;                          // after the labelling: (representative ring slot | 0xffff) << 16 | area
  const int mi = blockIdx.y;
  const uint8_t* m = base + (long)(idx ? idx[mi] : mi) * H * W;
  const int sw = (W + 63) >> 6;
  const int ty = blockIdx.x / sw, tx = blockIdx.x - ty * sw;
  const int x0 = tx * 64, y0 = ty * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = x0 + lane;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int r = wave * 16 + i, y = y0 + r;
    const bool wk = y < H && x < W && work_at<HOLES>(m, W, y, x);
    const unsigned long long bits = __ballot(wk);
    if (lane == 0) rowbits[r] = bits;
    P[r * 64 + lane] = wk ? r * 64 + c2_run_start(bits, lane) : -1;
  }
  __syncthreads();
  {
    // Trivial tiles -- no work pixel, or (tile inside the frame) nothing but work pixels -- are most of a real mask's 256
    // tiles (a person covers a few percent of the frame: the islands pass sees empty tiles, the holes pass full ones).
    // They skip the union-find, the area table and the 8 KB of root records: R16[tile][0] carries a marker the apply
    // kernel checks first.  A full tile is ONE component rooted at its first pixel, represented by ring slot 0, with every
    // ring slot a member; an empty tile has no ring members at all.
    const unsigned long long rb = rowbits[lane];
    const bool all0 = __ballot(rb != 0ull) == 0ull;
    const bool all1 = __ballot(rb != ~0ull) == 0ull;   // 64 rows of 64 set bits: implies x0 + 63 < W and y0 + 63 < H
    if ((all0 || all1) && !no_trivial) {
      const int tile_id = mi * (int)gridDim.x + (int)blockIdx.x;
      const int cb = tile_id * 256, t = threadIdx.x;
      if (t == 0) R16[(long)tile_id * 4096] = all0 ? C2_TILE_EMPTY : C2_TILE_FULL;
      int rr, cc;
      const bool member = all1 && c2_slot_rc(t, rr, cc);
      RP[cb + t] = member ? cb : -1;
      RA[cb + t] = (all1 && t == 0) ? 4096 : 0;
      RF[cb + t] = (all1 && t == 0) ? y0 * W + x0 : C2_INF;
      RT[cb + t] = 0;
      RFT[cb + t] = C2_INF;
      return;                                           // no complete component here: nothing for the flags / best key
    }
  }
  for (int i = 0; i < 16; ++i) {                      // north links: cc_tile_kernel's decision tree
    // The tree has four union sites (NE behind a west neighbour | N | NW | NE); inlined one by one they are four divergent
    // loop nests of dependent LDS round trips per row, each entered by the whole wave when any lane needs it -- on noisy
    // masks nearly always.  A pixel makes at most two links, and the first one is exclusive across the branches: ONE
    // union site for it, a second (rare: NW and NE both set, N clear) for the other.  42 % of the kernel's time on noise.
    const int r = wave * 16 + i;
    if (r == 0) continue;
    const unsigned long long cur = rowbits[r], up = rowbits[r - 1];
    const bool wk = (cur >> lane) & 1ull;
    const bool w = lane > 0 && ((cur >> (lane - 1)) & 1ull);
    const bool bb = (up >> lane) & 1ull;
    const bool c = lane < 63 && ((up >> (lane + 1)) & 1ull);
    const bool a_ = lane > 0 && ((up >> (lane - 1)) & 1ull);
    const int p = r * 64 + lane;
    const bool has1 = wk && (w ? (c && !bb) : (bb || a_ || c));
    const int q1 = w ? p - 63 : bb ? p - 64 : a_ ? p - 65 : p - 63;
    if (has1) lds_union(P, p, q1);
    if (wk && !w && !bb && a_ && c) lds_union(P, p, p - 63);
  }
  __syncthreads();
  // run starts: the run's final root, two 16-bit values per register (0xffff: not a run start) -- the kernel must stay
  // under 64 registers to keep eight waves per SIMD
  unsigned sroot2[8];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const unsigned long long bits = rowbits[wave * 16 + i];
    const bool start = ((bits >> lane) & 1ull) && (lane == 0 || !((bits >> (lane - 1)) & 1ull));
    const unsigned v = start ? (unsigned)lds_find_halve(P, (wave * 16 + i) * 64 + lane) : 0xffffu;
    if (i & 1) sroot2[i >> 1] |= v << 16;
    else sroot2[i >> 1] = v;
  }
  auto sroot = [&](int i) -> int {
    const unsigned v = (sroot2[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
    return v == 0xffffu ? -1 : (int)v;
  };
  __syncthreads();                                      // every walk is done: the parent array becomes the RC table
  for (int i = threadIdx.x; i < 4096; i += 256) RC[i] = 0xffff0000u;
  __syncthreads();
  const int tile_id = mi * (int)gridDim.x + (int)blockIdx.x;
  unsigned short* r16 = R16 + (long)tile_id * 4096;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave * 16 + i;
    const unsigned long long bits = rowbits[r];
    const bool wk = (bits >> lane) & 1ull;
    const int sr = sroot(i);
    if (sr >= 0) {                                      // run length -> the root's area
      const unsigned long long rest = ~(bits >> lane);
      const int len = rest ? __ffsll((long long)rest) - 1 : 64 - lane;
      atomicAdd(&RC[sr], (unsigned)len);
    }
    const int from = __shfl(sr, wk ? c2_run_start(bits, lane) : lane);   // every pixel: root of its run
    if (!(wk && from == r * 64 + lane)) r16[r * 64 + lane] = wk ? (unsigned short)from : (unsigned short)0xffffu;
    if (wk && (r == 0 || r == 63 || lane == 0 || lane == 63)) {   // ring pixel: 16-bit min of the slot in the high half
      const unsigned slot = (unsigned)c2_ring_slot(r, lane);
      unsigned old = RC[from];
      while ((old >> 16) > slot) {
        const unsigned seen = atomicCAS(&RC[from], old, (slot << 16) | (old & 0xffffu));
        if (seen == old) break;
        old = seen;
      }
    }
  }
  __syncthreads();
  const int cb = tile_id * 256;
  int* ring = (int*)(c2smem + 512 + 4096 * 4);          // [3][256] staged ring records: RP | RA | RF
  ring[threadIdx.x] = -1;
  ring[256 + threadIdx.x] = 0;
  ring[512 + threadIdx.x] = C2_INF;
  __syncthreads();
  unsigned long long key = 0ull;
  int flags = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = wave * 16 + i, q = r * 64 + lane;
    const int sr = sroot(i);
    if (r == 0 || r == 63) {                            // ring rows: every work pixel of the row
      const unsigned long long bits = rowbits[r];
      const bool wk = (bits >> lane) & 1ull;
      const int from = __shfl(sr, wk ? c2_run_start(bits, lane) : lane);
      if (wk) {
        const int slot = c2_ring_slot(r, lane);
        const unsigned w = RC[from];
        ring[slot] = cb + (int)(w >> 16);
        if ((int)(w >> 16) == slot) {
          ring[256 + slot] = (int)(w & 0xffffu);
          ring[512 + slot] = (y0 + (from >> 6)) * W + x0 + (from & 63);
        }
      }
    } else {                                            // ring columns: lane 0 is its own run start, lane 63 asks its run start
      const unsigned long long bits = rowbits[r];
      const bool w63 = (bits >> 63) & 1ull;
      const int from63 = __shfl(sr, w63 ? c2_run_start(bits, 63) : 63);
      const bool mine = (lane == 0 && (bits & 1ull)) || (lane == 63 && w63);
      if (mine) {
        const int from = lane == 0 ? sr : from63;
        const int slot = c2_ring_slot(r, lane);
        const unsigned w = RC[from];
        ring[slot] = cb + (int)(w >> 16);
        if ((int)(w >> 16) == slot) {
          ring[256 + slot] = (int)(w & 0xffffu);
          ring[512 + slot] = (y0 + (from >> 6)) * W + x0 + (from & 63);
        }
      }
    }
    if (sr != q) continue;                              // tile roots: their R16 entry becomes the component's record
    const unsigned w = RC[q];
    if ((w >> 16) != 0xffffu) {
      r16[q] = (unsigned short)(0xC000u | (w >> 16));   // 0x8000: root record, 0x4000: ring component, low bits: ring slot
      continue;
    }
    const int area = (int)(w & 0xffffu);                // complete component (no ring pixel): area <= 4096, flags now
    r16[q] = (unsigned short)(0x8000u | (unsigned)area);
    flags |= area < thresh ? 1 : 2;
    const unsigned p = (unsigned)((y0 + r) * W + x0 + lane);
    const unsigned long long k = ((unsigned long long)(unsigned)area << 32) | (0xffffffffu - p);
    key = k > key ? k : key;
  }
  __syncthreads();
  RP[cb + threadIdx.x] = ring[threadIdx.x];
  RA[cb + threadIdx.x] = ring[256 + threadIdx.x];
  RF[cb + threadIdx.x] = ring[512 + threadIdx.x];
  RT[cb + threadIdx.x] = 0;
  RFT[cb + threadIdx.x] = C2_INF;
  c2_meta_reduce(key, flags, meta + mi);
}

// cc_border_kernel on the ring forest: the same contact-run logic, endpoints named by their compact ring slot
template <int HOLES>
__global__ __launch_bounds__(256) void cc2_border_kernel(const uint8_t* __restrict__ base, const int* __restrict__ idx,
                                                         int* __restrict__ RP, int H, int W) {
  const int mi = blockIdx.y;
  const uint8_t* m = base + (long)(idx ? idx[mi] : mi) * H * W;
  const int sw = (W + 63) >> 6, tiles = ((H + 63) >> 6) * sw;
  auto at = [&](int y, int x) -> bool { return y >= 0 && x >= 0 && x < W && y < H && work_at<HOLES>(m, W, y, x); };
  auto cs = [&](int y, int x) -> int { return (mi * tiles + (y >> 6) * sw + (x >> 6)) * 256 + c2_ring_slot(y & 63, x & 63); };
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
    if (lane == 0 && at(y, x - 1)) uf_union(RP, cs(y, x), cs(y, x - 1));
    if (y == 0) return;
    const unsigned long long cn = cb & ub;
    const bool n_here = (cn >> lane) & 1ull;
    const bool n_left = lane > 0 && ((cn >> (lane - 1)) & 1ull);
    const bool n_right = lane < 63 && ((cn >> (lane + 1)) & 1ull);
    if (n_here) {
      if (!n_left) uf_union(RP, cs(y, x), cs(y - 1, x));
      return;
    }
    const bool upl = lane > 0 ? ((ub >> (lane - 1)) & 1ull) : at(y - 1, x0 - 1);
    const bool upr = lane < 63 ? ((ub >> (lane + 1)) & 1ull) : at(y - 1, x0 + 64);
    if (upl && !n_left) uf_union(RP, cs(y, x), cs(y - 1, x - 1));
    if (upr && !n_right) uf_union(RP, cs(y, x), cs(y - 1, x + 1));
    return;
  }
  const int y = y0 + lane;
  if (lane == 0 || y >= H) return;
  if (at(y, x0)) {                                      // first column: W / NW into the tile on the left
    const bool n = at(y - 1, x0);
    if (at(y, x0 - 1)) {
      if (!(n && at(y - 1, x0 - 1))) uf_union(RP, cs(y, x0), cs(y, x0 - 1));
    } else if (!n && at(y - 1, x0 - 1)) {
      uf_union(RP, cs(y, x0), cs(y - 1, x0 - 1));
    }
  }
  const int xr = x0 + 63;                               // last column: NE into the tile on the right
  if (xr + 1 < W && at(y, xr) && !at(y - 1, xr) && !at(y, xr + 1) && at(y - 1, xr + 1))
    uf_union(RP, cs(y, xr), cs(y - 1, xr + 1));
}

__global__ __launch_bounds__(256) void cc2_total_kernel(int* __restrict__ RP, const int* __restrict__ RA,
                                                        const int* __restrict__ RF, int* __restrict__ RT,
                                                        int* __restrict__ RFT, long slots) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= slots) return;
  const int a = RA[i];
  if (a <= 0) return;
  const int root = uf_find(RP, (int)i);
  atomicAdd(RT + root, a);
  atomicMin(RFT + root, RF[i]);
}

__global__ __launch_bounds__(256) void cc2_ringmeta_kernel(const int* __restrict__ RP, const int* __restrict__ RT,
                                                           const int* __restrict__ RFT, RegionMeta* __restrict__ meta,
                                                           int per_mask, int thresh) {
  const int mi = blockIdx.y;
  const int per = (per_mask + gridDim.x - 1) / gridDim.x;
  const int s0 = blockIdx.x * per, s1 = min(per_mask, s0 + per);
  unsigned long long key = 0ull;
  int flags = 0;
  for (int s = s0 + threadIdx.x; s < s1; s += 256) {
    const int i = mi * per_mask + s;
    if (RP[i] != i) continue;
    const int area = RT[i];
    if (area <= 0) continue;
    flags |= area < thresh ? 1 : 2;
    const unsigned long long k = ((unsigned long long)(unsigned)area << 32) | (0xffffffffu - (unsigned)RFT[i]);
    key = k > key ? k : key;
  }
  c2_meta_reduce(key, flags, meta + mi);
}

// apply: no labelling -- a pixel's tile root comes from R16, the tile's root pixels work out their component's decision
// (area from its own R16 record, or from the ring forest's totals) into a 4 KB LDS table, every pixel looks its root's decision up.
template <int HOLES>
__global__ __launch_bounds__(256) void cc2_apply_kernel(const uint8_t* in_base, uint8_t* out_base, const int* __restrict__ idx,
                                                        const unsigned short* __restrict__ R16, const int* __restrict__ RP,
                                                        const int* __restrict__ RT, const int* __restrict__ RFT,
                                                        const RegionMeta* __restrict__ meta, int* __restrict__ ext, int H, int W,
                                                        int thresh) {
  __shared__ uint8_t dec[4096];
  const int mi = blockIdx.y;
  const long moff = (long)(idx ? idx[mi] : mi) * H * W;
  const uint8_t* m = in_base + moff;
  uint8_t* o = out_base + moff;
  const int sw = (W + 63) >> 6;
  const int ty = blockIdx.x / sw, tx = blockIdx.x - ty * sw;
  const int x0 = tx * 64, y0 = ty * 64;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = x0 + lane;
  const RegionMeta mm = meta[mi];
  int ex0 = W, ey0 = H, ex1 = -1, ey1 = -1;
  if (!mm.any_small) {                                 // nothing to edit in this pass: copy (or leave in place)
    for (int i = 0; i < 16; ++i) {
      const int y = y0 + wave * 16 + i;
      if (y >= H || x >= W) continue;
      const uint8_t v = m[(long)y * W + x] != 0;
      if (o != m) o[(long)y * W + x] = v;
      if (!HOLES && v) { ex0 = min(ex0, x); ex1 = max(ex1, x); ey0 = min(ey0, y); ey1 = max(ey1, y); }
    }
  } else {
    const int tile_id = mi * (int)gridDim.x + (int)blockIdx.x;
    const unsigned short* r16 = R16 + (long)tile_id * 4096;
    const int cb = tile_id * 256;
    const unsigned best_first = 0xffffffffu - (unsigned)(mm.best & 0xffffffffull);
    const unsigned short mark = r16[0];
    if (mark == C2_TILE_EMPTY || mark == C2_TILE_FULL) {
      // one decision for the whole tile; nothing is read back but the ring totals of slot 0 (full tiles)
      bool v;                                            // value of every output pixel of the tile
      if (mark == C2_TILE_EMPTY) {
        v = HOLES;                                       // holes pass: all foreground, stays; islands pass: all background
      } else {
        const int g = uf_find(RP, cb);
        const int area = RT[g];
        const unsigned first = (unsigned)RFT[g];
        v = HOLES ? (area < thresh) : (mm.any_big ? (area >= thresh) : (first == best_first));
      }
      const bool same = mark == C2_TILE_EMPTY ? true : (HOLES ? !v : v);   // output == input
      if (o != m || !same) {
        for (int i = 0; i < 16; ++i) {
          const int y = y0 + wave * 16 + i;
          if (y < H && x < W) o[(long)y * W + x] = v ? 1 : 0;
        }
      }
      if (!HOLES && v) {                                 // (a full tile lies inside the frame)
        ex0 = x; ex1 = x; ey0 = y0 + wave * 16; ey1 = y0 + wave * 16 + 15;
      }
    } else {
    unsigned short roots[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int q = (wave * 16 + i) * 64 + lane;
      const unsigned short rt = r16[q];
      roots[i] = rt;
      if (rt == 0xffffu || !(rt & 0x8000u)) continue;    // root pixels carry their component's record
      roots[i] = (unsigned short)q;
      int area = (int)(rt & 0x3fffu);
      unsigned first = (unsigned)((y0 + (q >> 6)) * W + x0 + (q & 63));
      if (rt & 0x4000u) {
        const int g = uf_find(RP, cb + (int)(rt & 0x3fffu));
        area = RT[g];
        first = (unsigned)RFT[g];
      }
      // 1 = the component's pixels are SET in the output
      dec[q] = HOLES ? (area < thresh) : (mm.any_big ? (area >= thresh) : (first == best_first));
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int y = y0 + wave * 16 + i;
      if (y >= H || x >= W) continue;
      const unsigned short rt = roots[i];
      bool v;
      if (HOLES) v = rt == 0xffffu || dec[rt] != 0;      // foreground stays; small background components are filled
      else v = rt != 0xffffu && dec[rt] != 0;
      o[(long)y * W + x] = v ? 1 : 0;
      if (!HOLES && v) { ex0 = min(ex0, x); ex1 = max(ex1, x); ey0 = min(ey0, y); ey1 = max(ey1, y); }
    }
    }
  }
  if (!HOLES) {                                        // box extents of the final mask: <= 4 atomics per tile
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      ex0 = min(ex0, __shfl_xor(ex0, off)); ey0 = min(ey0, __shfl_xor(ey0, off));
      ex1 = max(ex1, __shfl_xor(ex1, off)); ey1 = max(ey1, __shfl_xor(ey1, off));
    }
    __shared__ int red[4][4];
    if (lane == 0) { red[wave][0] = ex0; red[wave][1] = ey0; red[wave][2] = ex1; red[wave][3] = ey1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int w = 1; w < 4; ++w) {
        ex0 = min(ex0, red[w][0]); ey0 = min(ey0, red[w][1]);
        ex1 = max(ex1, red[w][2]); ey1 = max(ey1, red[w][3]);
      }
      if (ex1 >= 0) {
        int* e = ext + (long)mi * 4;
        atomicMin(e + 0, ex0); atomicMin(e + 1, ey0);
        atomicMax(e + 2, ex1); atomicMax(e + 3, ey1);
      }
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

extern "C" long csam_small_regions_idx_workspace_bytes(int n, int H, int W) {
  if (n <= 0 || H <= 0 || W <= 0) return 0;
  const long tiles = (long)n * ((H + 63) >> 6) * ((W + 63) >> 6);
  return 5 * align256(tiles * 256 * 4) + align256(tiles * 4096 * 2) + align256((long)2 * n * sizeof(RegionMeta)) +
         align256((long)n * 16);
}

// remove_small_regions (holes, then islands) on the masks masks_base[idx[i]] (idx == NULL: i), i < n, written to
// out_base[idx[i]] -- out_base may equal masks_base (in place).  Compact form: see the comment above cc2_scan_kernel.
extern "C" int csam_small_regions_idx(void* stream_, const uint8_t* masks_base, const int* idx, uint8_t* out_base,
                                      int* changed, float* boxes, int n, int H, int W, int min_area, void* ws,
                                      long ws_bytes) {
  CSAM_REQUIRE(masks_base && out_base && changed && boxes && ws, "csam_small_regions_idx: null pointer");
  CSAM_REQUIRE(n > 0 && H > 0 && W > 0 && (long)H * W < (1L << 31), "csam_small_regions_idx: bad shape");
  const int tiles = ((H + 63) >> 6) * ((W + 63) >> 6);
  const long slots = (long)n * tiles * 256;
  CSAM_REQUIRE(n <= 65535 && slots < (1L << 31), "csam_small_regions_idx: too many masks for one call");
  CSAM_REQUIRE(ws_bytes >= csam_small_regions_idx_workspace_bytes(n, H, W), "csam_small_regions_idx: workspace too small");
  hipStream_t stream = (hipStream_t)stream_;
  char* w = (char*)ws;
  int* RP = (int*)w; w += align256(slots * 4);
  int* RA = (int*)w; w += align256(slots * 4);
  int* RF = (int*)w; w += align256(slots * 4);
  int* RT = (int*)w; w += align256(slots * 4);
  int* RFT = (int*)w; w += align256(slots * 4);
  unsigned short* R16 = (unsigned short*)w; w += align256((long)n * tiles * 4096 * 2);
  RegionMeta* meta = (RegionMeta*)w; w += align256((long)2 * n * sizeof(RegionMeta));
  int* ext = (int*)w;
  static csam_once_t attr;
  if (csam_first_call(attr)) {
    (void)hipFuncSetAttribute((const void*)cc2_scan_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS);
    (void)hipFuncSetAttribute((const void*)cc2_scan_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, C2_LDS);
  }
  const dim3 block(256), tgrid(tiles, n), bgrid(csam_cdiv(2 * tiles, 4), n), sgrid((unsigned)csam_cdiv(slots, 256));
  const dim3 mgrid(std::min(16, std::max(1, tiles / 4)), n);
  hipLaunchKernelGGL(cc_prepare_kernel, dim3(csam_cdiv(2 * n, 256)), dim3(256), 0, stream, meta, ext, n, H, W);
  // pass 1: holes (components of the complement), masks -> out
  const int no_trivial = 0;                            // (1 = label trivial tiles like any other: the round-3 A/B)
  hipLaunchKernelGGL(cc2_scan_kernel<1>, tgrid, block, C2_LDS, stream, masks_base, idx, R16, RP, RA, RF, RT, RFT, meta, H, W,
                     min_area, no_trivial);
  hipLaunchKernelGGL(cc2_border_kernel<1>, bgrid, block, 0, stream, masks_base, idx, RP, H, W);
  hipLaunchKernelGGL(cc2_total_kernel, sgrid, block, 0, stream, RP, RA, RF, RT, RFT, slots);
  hipLaunchKernelGGL(cc2_ringmeta_kernel, mgrid, block, 0, stream, RP, RT, RFT, meta, tiles * 256, min_area);
  hipLaunchKernelGGL(cc2_apply_kernel<1>, tgrid, block, 0, stream, masks_base, out_base, idx, R16, RP, RT, RFT, meta, ext,
                     H, W, min_area);
  // pass 2: islands (components of the hole-filled mask), out -> out
  hipLaunchKernelGGL(cc2_scan_kernel<0>, tgrid, block, C2_LDS, stream, out_base, idx, R16, RP, RA, RF, RT, RFT, meta + n, H,
                     W, min_area, no_trivial);
  hipLaunchKernelGGL(cc2_border_kernel<0>, bgrid, block, 0, stream, out_base, idx, RP, H, W);
  hipLaunchKernelGGL(cc2_total_kernel, sgrid, block, 0, stream, RP, RA, RF, RT, RFT, slots);
  hipLaunchKernelGGL(cc2_ringmeta_kernel, mgrid, block, 0, stream, RP, RT, RFT, meta + n, tiles * 256, min_area);
  hipLaunchKernelGGL(cc2_apply_kernel<0>, tgrid, block, 0, stream, out_base, out_base, idx, R16, RP, RT, RFT, meta + n,
                     ext, H, W, min_area);
  hipLaunchKernelGGL(cc_finish_kernel, dim3(csam_cdiv(n, 256)), dim3(256), 0, stream, meta, ext, n, changed, boxes);
  CSAM_LAUNCH_CHECK("csam_small_regions_idx");
  return CSAM_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Window copies for the bounding-box-restricted clean-up (round 4).  remove_small_regions (amg.py:267-291) looks at the
// whole frame, but a mask's components all lie inside its bounding box: with the box padded by a ring of background the
// clean-up of the WINDOW is the clean-up of the frame (the ring is one background component, connected to everything
// outside, and large; islands never touch it).  csam_mask_window_copy gathers the windows of n masks into a dense
// [n, Hc, Wc] stack (zero outside the window) or scatters them back; the caller runs csam_small_regions_idx on the stack.
// win[i] = (x0, y0, w, h, ox, oy): mask i's window inside its H x W store slot and where it sits in the stack (ox + w <= Wc,
// oy + h <= Hc) -- a window cut by the right / bottom frame edge is placed flush with the stack's right / bottom edge, so that
// the zero padding never extends the frame beyond its true border.
// ---------------------------------------------------------------------------------------------------------------------
template <int TO_STORE>
__global__ __launch_bounds__(256) void mask_window_copy_kernel(uint8_t* __restrict__ store, const int* __restrict__ slots,
                                                               const int* __restrict__ win, const uint8_t* __restrict__ only,
                                                               uint8_t* __restrict__ crop, int H, int W, int Hc, int Wc) {
  const int n = blockIdx.z;
  if (TO_STORE && only && !only[n]) return;
  const int x0 = win[n * 6 + 0], y0 = win[n * 6 + 1], w = win[n * 6 + 2], h = win[n * 6 + 3], ox = win[n * 6 + 4],
            oy = win[n * 6 + 5];
  const int yc = blockIdx.y * 4 + (threadIdx.x >> 6);   // row of the stack
  if (yc >= Hc) return;
  const int y = yc - oy;                                // row of the window
  const bool yin = y >= 0 && y < h;
  const uint8_t* srow = store + ((long)(slots ? slots[n] : n) * H + (y0 + (yin ? y : 0))) * W + x0;
  uint8_t* crow = crop + ((long)n * Hc + yc) * Wc;
  for (int xc = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4; xc < Wc; xc += gridDim.x * 256) {
    if (TO_STORE) {
      if (yin) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int x = xc + e - ox;
          if (x >= 0 && x < w) const_cast<uint8_t*>(srow)[x] = crow[xc + e];
        }
      }
    } else {
      uint32_t v = 0;
      if (yin) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int x = xc + e - ox;
          if (x >= 0 && x < w) v |= (uint32_t)(srow[x] != 0) << (8 * e);
        }
      }
      *(uint32_t*)(crow + xc) = v;                      // Wc % 4 == 0
    }
  }
}

extern "C" int csam_mask_window_copy(void* stream, void* store_u8, const int* slots_or_null, const int* windows, const void* only_u8_or_null,
                                     void* crop_u8, int n, int H, int W, int Hc, int Wc, int to_store) {
  CSAM_REQUIRE(store_u8 && windows && crop_u8 && n > 0 && H > 0 && W > 0 && Hc > 0 && Wc > 0 && Wc % 4 == 0 && n <= 65535,
               "csam_mask_window_copy: bad args");
  const dim3 grid(csam_cdiv(Wc, 256), csam_cdiv(Hc, 4), n);
  if (to_store)
    hipLaunchKernelGGL(mask_window_copy_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (uint8_t*)store_u8, slots_or_null, windows,
                       (const uint8_t*)only_u8_or_null, (uint8_t*)crop_u8, H, W, Hc, Wc);
  else
    hipLaunchKernelGGL(mask_window_copy_kernel<0>, grid, dim3(256), 0, (hipStream_t)stream, (uint8_t*)store_u8, slots_or_null, windows,
                       (const uint8_t*)only_u8_or_null, (uint8_t*)crop_u8, H, W, Hc, Wc);
  CSAM_LAUNCH_CHECK("csam_mask_window_copy");
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
  const int tiled = 1;                                 // two-level labelling (tiles in LDS + cross-edge links)
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
