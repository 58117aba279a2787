// csam_caltech_match: the Caltech matching step of the CrowdHuman evaluator on device, float64, bit-exact.
//
// Reference: tools/crowdhuman_eval.py:113-143 Image.compare_caltech with :215-236 box_overlap_opr --
// per image, detections in descending score order are matched greedily to the positive ground-truth box
// of largest IoU (first index on ties, np.argmax) if that IoU exceeds thres; a matched GT column is zeroed
// for the detections that follow; an unmatched detection is dropped when an ignore region covers more
// than thres of ITS area (IoA), else it is a false positive.  `pos` = the detection overlaps some positive
// GT by more than thres before any zeroing (reference's 4th tuple field).
// The reference is a Python loop over a materialised [N,K] float64 matrix per image.  Here: one wave per
// image, lanes stride the GT boxes, the IoU row is recomputed per detection (never stored), the matched
// flags live in LDS, arg-max by wave reduction with the lowest-index tie rule.  All arithmetic is float64
// in numpy's evaluation order (no FMA contraction), so labels are identical to the reference's.
#include "csam_common.h"

namespace {

constexpr int EV_MAX_POS = 16384;   // matched flags in LDS (bytes)

#pragma clang fp contract(off)
__device__ __forceinline__ double ev_overlap(const double* d, const double* g, bool iou) {
  const double iw = fmin(d[2], g[2]) - fmax(d[0], g[0]);
  const double ih = fmin(d[3], g[3]) - fmax(d[1], g[1]);
  const double inter = fmax(0.0, iw) * fmax(0.0, ih);
  const double da = (d[2] - d[0]) * (d[3] - d[1]);
  if (iou) {
    const double ga = (g[2] - g[0]) * (g[3] - g[1]);
    return inter / (da + ga - inter + 1e-6);
  }
  return inter / (da + 1e-6);
}

__global__ __launch_bounds__(64) void caltech_match_kernel(const double* __restrict__ dt, const long* __restrict__ dt_off,
                                                           const double* __restrict__ gt, const long* __restrict__ gt_off,
                                                           const int* __restrict__ gt_npos, double thres,
                                                           signed char* __restrict__ label,
                                                           unsigned char* __restrict__ pos) {
  __shared__ unsigned char matched[EV_MAX_POS];
  const int img = blockIdx.x, lane = threadIdx.x;
  const long d0 = dt_off[img], d1 = dt_off[img + 1];
  const long g0 = gt_off[img], g1 = gt_off[img + 1];
  const int npos = gt_npos[img];
  const int nign = (int)(g1 - g0) - npos;
  if (g1 == g0) {                       // no ground truth at all: the reference drops every detection of the image
    for (long i = d0 + lane; i < d1; i += 64) { label[i] = -1; pos[i] = 0; }
    return;
  }
  for (int j = lane; j < npos; j += 64) matched[j] = 0;
  __syncthreads();
  const double* gp = gt + g0 * 5;
  const double* gi = gp + (long)npos * 5;
  for (long i = d0; i < d1; ++i) {
    const double* d = dt + i * 5;
    double best = -1.0;                 // IoU >= 0, so any column beats it; empty row keeps j = -1
    int bestj = 0x7fffffff;
    bool anypos = false;
    for (int j = lane; j < npos; j += 64) {
      const double v = ev_overlap(d, gp + (long)j * 5, true);
      anypos |= v > thres;
      const double vz = matched[j] ? 0.0 : v;
      if (vz > best) { best = vz; bestj = j; }          // strict: keeps the lowest j of this lane
    }
    bool anyign = false;
    for (int j = lane; j < nign; j += 64) anyign |= ev_overlap(d, gi + (long)j * 5, false) > thres;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_xor(best, off);
      const int oj = __shfl_xor(bestj, off);
      if (ob > best || (ob == best && oj < bestj)) { best = ob; bestj = oj; }
    }
    const bool ap = __ballot(anypos) != 0ull, ai = __ballot(anyign) != 0ull;
    const bool hit = npos > 0 && best > thres;
    if (lane == 0) {
      label[i] = hit ? 1 : (ai ? -1 : 0);
      pos[i] = ap ? 1 : 0;
      if (hit) matched[bestj] = 1;
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int csam_caltech_match(void* stream, const double* dt, const long* dt_off, const double* gt,
                                  const long* gt_off, const int* gt_npos, int n_img, int max_pos, double thres,
                                  signed char* label, unsigned char* pos) {
  CSAM_REQUIRE(dt_off && gt_off && gt_npos && label && pos && n_img > 0, "csam_caltech_match: bad args");
  CSAM_REQUIRE(max_pos <= EV_MAX_POS, "csam_caltech_match: at most %d positive boxes per image", EV_MAX_POS);
  hipLaunchKernelGGL(caltech_match_kernel, dim3(n_img), dim3(64), 0, (hipStream_t)stream, dt, dt_off, gt, gt_off,
                     gt_npos, thres, label, pos);
  CSAM_LAUNCH_CHECK("csam_caltech_match");
  return CSAM_OK;
}
