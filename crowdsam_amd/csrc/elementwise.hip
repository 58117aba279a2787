// HBM-bound helper kernels of the encoder path: LayerNorm, im2col (SAM 16x16 patches, DINOv2 14x14
// patches with the fused 1024->1022 bilinear resample, 3x3 neck conv), casts/adds.
// All are pure streaming kernels: 16-byte vector accesses, one wave per row for the reductions.
#include "csam_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm over the last dim (nn.LayerNorm, and LayerNorm2d on token-major NHWC data):
// reference image_encoder.py:168,180 (eps 1e-6), common.py:38-43, transformer.py norms (eps 1e-5).
// One wave per row, D % 4 == 0, D <= 1280.  Two-pass (mean, centred variance) in registers.
// ---------------------------------------------------------------------------------------------
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void layernorm_kernel(const TI* __restrict__ x, long ldx,
                                                        TO* __restrict__ y, long ldy,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, int M, int D,
                                                        float eps) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nchunk = D >> 2;
  constexpr int MAXC = 5;
  floatx4 v[MAXC];
  const TI* xr = x + (long)row * ldx;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int c = lane + 64 * j;
    if (c < nchunk) {
      if constexpr (sizeof(TI) == 4) {
        v[j] = *(const floatx4*)(xr + c * 4);
      } else {
        const half4_t h = *(const half4_t*)(xr + c * 4);
        v[j] = floatx4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
      }
      s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
  }
  const float mean = csam_wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int c = lane + 64 * j;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[j][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(csam_wave_sum(q) / (float)D + eps);
  TO* yr = y + (long)row * ldy;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int c = lane + 64 * j;
    if (c < nchunk) {
      const floatx4 g = *(const floatx4*)(gamma + c * 4);
      const floatx4 b = *(const floatx4*)(beta + c * 4);
      floatx4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * g[e] + b[e];
      if constexpr (sizeof(TO) == 4) {
        *(floatx4*)(yr + c * 4) = o;
      } else {
        half4_t h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)o[e];
        *(half4_t*)(yr + c * 4) = h;
      }
    }
  }
}

// LayerNorm of the decoder's token rows with the casts its consumers need folded in: y (fp32, the residual stream),
// y16 = fp16(y) (value / MLP operand) and ype16 = fp16(y + pe) (query / key operand: tokens + their positional
// embedding, transformer.py:164-190) -- one launch instead of LayerNorm + two add_cast.  fp32 rows, D <= 1280.
__global__ __launch_bounds__(256) void layernorm_cast_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int M, int D, float eps,
                                                             float* __restrict__ y, half_t* __restrict__ y16,
                                                             const float* __restrict__ pe, half_t* __restrict__ ype16) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nchunk = D >> 2;
  constexpr int MAXC = 5;
  floatx4 v[MAXC];
  const float* xr = x + (long)row * D;
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int c = lane + 64 * j;
    if (c < nchunk) {
      v[j] = *(const floatx4*)(xr + c * 4);
      s += v[j][0] + v[j][1] + v[j][2] + v[j][3];
    }
  }
  const float mean = csam_wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int c = lane + 64 * j;
    if (c < nchunk) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float d = v[j][e] - mean;
        q += d * d;
      }
    }
  }
  const float rstd = rsqrtf(csam_wave_sum(q) / (float)D + eps);
#pragma unroll
  for (int j = 0; j < MAXC; ++j) {
    const int c = lane + 64 * j;
    if (c < nchunk) {
      const floatx4 g = *(const floatx4*)(gamma + c * 4);
      const floatx4 b = *(const floatx4*)(beta + c * 4);
      floatx4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = (v[j][e] - mean) * rstd * g[e] + b[e];   // same expression as layernorm_kernel
      const long off = (long)row * D + c * 4;
      *(floatx4*)(y + off) = o;
      if (y16) {
        half4_t h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)o[e];
        *(half4_t*)(y16 + off) = h;
      }
      if (ype16) {
        const floatx4 pv = o + *(const floatx4*)(pe + off);                     // fp32 add, then the cast: add_cast's order
        half4_t h;
#pragma unroll
        for (int e = 0; e < 4; ++e) h[e] = (half_t)pv[e];
        *(half4_t*)(ype16 + off) = h;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// SAM patch-embed im2col with fused Sam.preprocess (sam.py:163-173): normalise with mean/std in
// fp32, zero-pad bottom/right to 1024^2, emit A[4096, 768] f16 with k = c*256 + dy*16 + dx so the
// conv weight [D,3,16,16] is used as the [D,768] GEMM operand unchanged (image_encoder.py:387-395).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sam_im2col_kernel(const float* __restrict__ img, int h, int w,
                                                         floatx4 mean, floatx4 stdv,
                                                         half_t* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;  // one thread = 8 consecutive dx
  if (idx >= 4096 * 96) return;
  const int t = idx / 96, kc = idx % 96;
  const int c = kc / 32, rem = kc % 32, dy = rem >> 1, dx0 = (rem & 1) * 8;
  const int py = t >> 6, px = t & 63;
  const int y = py * 16 + dy, x0 = px * 16 + dx0;
  half8_t o;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int x = x0 + e;
    float v = 0.f;
    if (y < h && x < w) v = (img[((long)c * h + y) * w + x] - mean[c]) / stdv[c];
    o[e] = (half_t)v;
  }
  *(half8_t*)(out + (long)t * 768 + kc * 8) = o;
}

// ---------------------------------------------------------------------------------------------
// DINOv2 input: bilinear (align_corners=False) resample of the normalised+padded 1024^2 tensor to
// 1022^2 (predictor.py:104) fused with the 14x14 patch im2col: A[5329, 640] f16, k = c*196+dy*14+dx,
// columns 588..639 zero (K padded to the GEMM's 64 multiple).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float norm_px(const float* img, int h, int w, int c, int y, int x,
                                         float mean, float stdv) {
  return (y < h && x < w) ? (img[((long)c * h + y) * w + x] - mean) / stdv : 0.f;
}

__global__ __launch_bounds__(256) void dino_im2col_kernel(const float* __restrict__ img, int h, int w, int S,
                                                          floatx4 mean, floatx4 stdv,
                                                          half_t* __restrict__ out) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 5329L * 640) return;
  const int t = (int)(idx / 640), k = (int)(idx % 640);
  float v = 0.f;
  if (k < 588) {
    const int c = k / 196, r = k % 196, dy = r / 14, dx = r % 14;
    const int Y = (t / 73) * 14 + dy, X = (t % 73) * 14 + dx;
    const float sc = (float)S / 1022.0f;   // frame size S (1024: padded SAM frame; 1022: already resized)
    float sy = sc * ((float)Y + 0.5f) - 0.5f;
    float sx = sc * ((float)X + 0.5f) - 0.5f;
    sy = sy < 0.f ? 0.f : sy;
    sx = sx < 0.f ? 0.f : sx;
    const int y0 = (int)sy, x0 = (int)sx;
    const int y1 = y0 + (y0 < S - 1 ? 1 : 0), x1 = x0 + (x0 < S - 1 ? 1 : 0);
    const float ly = sy - (float)y0, lx = sx - (float)x0;
    const float m = mean[c], s = stdv[c];
    const float v00 = norm_px(img, h, w, c, y0, x0, m, s), v01 = norm_px(img, h, w, c, y0, x1, m, s);
    const float v10 = norm_px(img, h, w, c, y1, x0, m, s), v11 = norm_px(img, h, w, c, y1, x1, m, s);
    v = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
  out[idx] = (half_t)v;
}

// ---------------------------------------------------------------------------------------------
// 3x3 / pad 1 im2col on token-major [64,64,C] f16 (neck conv, image_encoder.py:96-102):
// A[4096, 9*C], k = (ky*3+kx)*C + c.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col3x3_kernel(const half_t* __restrict__ in, half_t* __restrict__ out,
                                                        int C) {
  const int cpt = C >> 3;  // 16-B chunks per tap
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 4096L * 9 * cpt) return;
  const int t = (int)(idx / (9 * cpt));
  const int r = (int)(idx % (9 * cpt));
  const int tap = r / cpt, ch = r % cpt;
  const int y = (t >> 6) + tap / 3 - 1, x = (t & 63) + tap % 3 - 1;
  half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
  if (y >= 0 && y < 64 && x >= 0 && x < 64) v = *(const half8_t*)(in + ((long)(y * 64 + x)) * C + ch * 8);
  *(half8_t*)(out + (long)t * 9 * C + (long)tap * C + ch * 8) = v;
}

// y[M,N] (f16) = a[M,N] (f32|f16) + b[N or M,N] (f32), generic small helper
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       long b_row_stride, half_t* __restrict__ y16,
                                                       float* __restrict__ y32, long M, int N) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= M * (N >> 2)) return;
  const long m = idx / (N >> 2);
  const int n = (int)(idx % (N >> 2)) * 4;
  floatx4 v = *(const floatx4*)(a + m * N + n);
  if (b) v += *(const floatx4*)(b + m * b_row_stride + n);
  if (y32) *(floatx4*)(y32 + m * N + n) = v;
  if (y16) {
    half4_t h;
#pragma unroll
    for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
    *(half4_t*)(y16 + m * N + n) = h;
  }
}

// Sam.preprocess (sam.py:163-173) materialised: (x-mean)/std, zero pad to [3,1024,1024] fp32.
__global__ __launch_bounds__(256) void preprocess_pad_kernel(const float* __restrict__ img, int h, int w,
                                                             floatx4 mean, floatx4 stdv, float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= 3 * 1024 * 1024) return;
  const int c = idx >> 20, y = (idx >> 10) & 1023, x = idx & 1023;
  out[idx] = (y < h && x < w) ? (img[((long)c * h + y) * w + x] - mean[c]) / stdv[c] : 0.f;
}

// out[i] = max_c sigmoid(x[c, i])   (crowdsam/model.py:203)
__global__ __launch_bounds__(256) void sigmoid_max_kernel(const float* __restrict__ x, int C, int N,
                                                          float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  float m = -INFINITY;
  for (int c = 0; c < C; ++c) m = fmaxf(m, 1.0f / (1.0f + expf(-x[(long)c * N + i])));
  out[i] = m;
}

// cv2.resize(image, (w, h)) of the reference's frame resize (crowdsam/utils.py:149; INTER_LINEAR, uint8, 3 channels)
// on the already-uploaded frame: OpenCV's generic fixed-point path restated (11-bit coefficients built by the host
// exactly as cv::resize builds them; horizontal pass into int32, vertical pass
// (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2; the exact-2x decimation is INTER_AREA's (a+b+c+d+2)>>2).
// One thread per output pixel; writes the uint8 HWC frame and/or the fp32 CHW tensor the encoders read.
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw,
                                                               const int* __restrict__ xofs,
                                                               const short* __restrict__ xcoef,
                                                               const int* __restrict__ yofs,
                                                               const short* __restrict__ ycoef, int dh, int dw,
                                                               int area2x, uint8_t* __restrict__ dst_u8,
                                                               float* __restrict__ dst_chw) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= dh * dw) return;
  const int y = idx / dw, x = idx - y * dw;
  int v[3];
  if (area2x) {
    const uint8_t* p = src + ((long)(2 * y) * sw + 2 * x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) v[c] = (p[c] + p[3 + c] + p[(long)sw * 3 + c] + p[(long)sw * 3 + 3 + c] + 2) >> 2;
  } else {
    const int x0 = xofs[x], x1 = min(x0 + 1, sw - 1);
    const int a0 = xcoef[2 * x], a1 = xcoef[2 * x + 1];
    const int r0 = yofs[2 * y], r1 = yofs[2 * y + 1];
    const int b0 = ycoef[2 * y], b1 = ycoef[2 * y + 1];
    const uint8_t* p0 = src + (long)r0 * sw * 3;
    const uint8_t* p1 = src + (long)r1 * sw * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int s0 = p0[x0 * 3 + c] * a0 + p0[x1 * 3 + c] * a1;
      const int s1 = p1[x0 * 3 + c] * a0 + p1[x1 * 3 + c] * a1;
      v[c] = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int u = v[c] & 255;                      // uchar(...) cast of OpenCV (the value is already in 0..255)
    if (dst_u8) dst_u8[(long)idx * 3 + c] = (uint8_t)u;
    if (dst_chw) dst_chw[(long)c * dh * dw + idx] = (float)u;
  }
}

// uint8 HWC (3 channels) -> fp32 CHW: the layout change + cast of SamPredictor.set_image (predictor.py:52-56) in one pass
__global__ __launch_bounds__(256) void u8hwc_to_f32chw_kernel(const uint8_t* __restrict__ src, long n,
                                                              float* __restrict__ dst) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
#pragma unroll
  for (int c = 0; c < 3; ++c) dst[c * n + idx] = (float)src[idx * 3 + c];
}

// Pillow's Image.resize(size, BILINEAR) for uint8 images (ImagingResample, 8 bits per channel), one separable pass:
// out[o] = clip8((2^21 + sum_j in[xmin[o] + j] * k[o][j]) >> 22) with the 22-bit coefficient tables Pillow's
// precompute_coeffs / normalize_coeffs_8bpc build (crowdsam_amd/resize.py).  ResizeLongestSide.apply_image
// (segment_anything_cs/utils/transforms.py:26-31, via torchvision's resize(to_pil_image(.))) runs the horizontal pass,
// rounds to uint8, then the vertical pass -- two launches of this kernel.  `axis` 0: along x, 1: along y.
// Optionally also writes the fp32 CHW tensor the encoder kernels read (second pass).
constexpr int PIL_MAXTAPS = 8;
__global__ __launch_bounds__(256) void pil_resample_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw,
                                                              const int* __restrict__ xmin, const int* __restrict__ ntap,
                                                              const int* __restrict__ coef, int dh, int dw, int axis,
                                                              uint8_t* __restrict__ dst, float* __restrict__ dst_chw) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= dh * dw) return;
  const int y = idx / dw, x = idx - y * dw;
  const int o = axis == 0 ? x : y;
  const int x0 = xmin[o], n = ntap[o];
  int acc[3] = {1 << 21, 1 << 21, 1 << 21};
  for (int j = 0; j < n; ++j) {
    const int k = coef[o * PIL_MAXTAPS + j];
    const uint8_t* p = axis == 0 ? src + ((long)y * sw + x0 + j) * 3 : src + ((long)(x0 + j) * sw + x) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] += (int)p[c] * k;
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    int v = acc[c] >> 22;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    dst[(long)idx * 3 + c] = (uint8_t)v;
    if (dst_chw) dst_chw[(long)c * dh * dw + idx] = (float)v;
  }
}

}  // namespace

extern "C" int csam_preprocess_pad(void* stream, const float* img_chw, int h, int w, const float* mean3,
                                   const float* std3, float* out) {
  CSAM_REQUIRE(img_chw && out && h > 0 && w > 0 && h <= 1024 && w <= 1024, "csam_preprocess_pad: bad args");
  floatx4 m = {mean3[0], mean3[1], mean3[2], 0.f}, sd = {std3[0], std3[1], std3[2], 1.f};
  hipLaunchKernelGGL(preprocess_pad_kernel, dim3(3 * 1024 * 1024 / 256), dim3(256), 0, (hipStream_t)stream, img_chw,
                     h, w, m, sd, out);
  CSAM_LAUNCH_CHECK("csam_preprocess_pad");
  return CSAM_OK;
}

extern "C" int csam_resize_linear_u8(void* stream, const uint8_t* src_hwc, int sh, int sw, const int* xofs,
                                     const short* xcoef, const int* yofs, const short* ycoef, int dh, int dw,
                                     int area2x, uint8_t* dst_hwc, float* dst_chw_f32) {
  CSAM_REQUIRE(src_hwc && (dst_hwc || dst_chw_f32) && sh > 0 && sw > 0 && dh > 0 && dw > 0,
               "csam_resize_linear_u8: bad args");
  CSAM_REQUIRE(area2x ? (sh == 2 * dh && sw == 2 * dw) : (xofs && xcoef && yofs && ycoef),
               "csam_resize_linear_u8: tables missing / area2x on a non-2x shape");
  hipLaunchKernelGGL(resize_linear_u8_kernel, dim3(csam_cdiv((long)dh * dw, 256)), dim3(256), 0, (hipStream_t)stream,
                     src_hwc, sh, sw, xofs, xcoef, yofs, ycoef, dh, dw, area2x, dst_hwc, dst_chw_f32);
  CSAM_LAUNCH_CHECK("csam_resize_linear_u8");
  return CSAM_OK;
}

extern "C" int csam_pil_resample_u8(void* stream, const uint8_t* src_hwc, int sh, int sw, const int* xmin, const int* ntap,
                                    const int* coef, int dh, int dw, int axis, uint8_t* dst_hwc, float* dst_chw_f32) {
  CSAM_REQUIRE(src_hwc && xmin && ntap && coef && dst_hwc && sh > 0 && sw > 0 && dh > 0 && dw > 0 && (axis == 0 || axis == 1),
               "csam_pil_resample_u8: bad args");
  CSAM_REQUIRE(axis == 0 ? dh == sh : dw == sw, "csam_pil_resample_u8: one pass resizes one axis");
  hipLaunchKernelGGL(pil_resample_u8_kernel, dim3(csam_cdiv((long)dh * dw, 256)), dim3(256), 0, (hipStream_t)stream, src_hwc,
                     sh, sw, xmin, ntap, coef, dh, dw, axis, dst_hwc, dst_chw_f32);
  CSAM_LAUNCH_CHECK("csam_pil_resample_u8");
  return CSAM_OK;
}

extern "C" int csam_u8hwc_to_f32chw(void* stream, const uint8_t* src_hwc, int h, int w, float* dst_chw) {
  CSAM_REQUIRE(src_hwc && dst_chw && h > 0 && w > 0, "csam_u8hwc_to_f32chw: bad args");
  hipLaunchKernelGGL(u8hwc_to_f32chw_kernel, dim3(csam_cdiv((long)h * w, 256)), dim3(256), 0, (hipStream_t)stream,
                     src_hwc, (long)h * w, dst_chw);
  CSAM_LAUNCH_CHECK("csam_u8hwc_to_f32chw");
  return CSAM_OK;
}

extern "C" int csam_sigmoid_max(void* stream, const float* x, int C, int N, float* out) {
  CSAM_REQUIRE(x && out && C > 0 && N > 0, "csam_sigmoid_max: bad args");
  hipLaunchKernelGGL(sigmoid_max_kernel, dim3(csam_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, x, C, N, out);
  CSAM_LAUNCH_CHECK("csam_sigmoid_max");
  return CSAM_OK;
}

extern "C" int csam_layernorm(void* stream, const void* x, long ldx, int x_dtype, void* y, long ldy,
                              int y_dtype, const float* gamma, const float* beta, int M, int D,
                              float eps) {
  CSAM_REQUIRE(x && y && gamma && beta, "csam_layernorm: null pointer");
  CSAM_REQUIRE(D % 4 == 0 && D <= 1280 && D >= 4, "csam_layernorm: D=%d unsupported", D);
  CSAM_REQUIRE(ldx % 4 == 0 && ldy % 4 == 0, "csam_layernorm: ld alignment");
  dim3 grid(csam_cdiv(M, 4)), block(256);
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == CSAM_DT_F32 && y_dtype == CSAM_DT_F16)
    hipLaunchKernelGGL((layernorm_kernel<float, half_t>), grid, block, 0, s, (const float*)x, ldx, (half_t*)y, ldy, gamma, beta, M, D, eps);
  else if (x_dtype == CSAM_DT_F32 && y_dtype == CSAM_DT_F32)
    hipLaunchKernelGGL((layernorm_kernel<float, float>), grid, block, 0, s, (const float*)x, ldx, (float*)y, ldy, gamma, beta, M, D, eps);
  else if (x_dtype == CSAM_DT_F16 && y_dtype == CSAM_DT_F16)
    hipLaunchKernelGGL((layernorm_kernel<half_t, half_t>), grid, block, 0, s, (const half_t*)x, ldx, (half_t*)y, ldy, gamma, beta, M, D, eps);
  else if (x_dtype == CSAM_DT_F16 && y_dtype == CSAM_DT_F32)
    hipLaunchKernelGGL((layernorm_kernel<half_t, float>), grid, block, 0, s, (const half_t*)x, ldx, (float*)y, ldy, gamma, beta, M, D, eps);
  else {
    csam_set_error("csam_layernorm: bad dtypes");
    return CSAM_ERR_ARG;
  }
  CSAM_LAUNCH_CHECK("csam_layernorm");
  return CSAM_OK;
}

extern "C" int csam_layernorm_cast(void* stream, const float* x, const float* gamma, const float* beta, int M, int D,
                                   float eps, float* y_f32, void* y_f16_or_null, const float* pe_or_null,
                                   void* ype_f16_or_null) {
  CSAM_REQUIRE(x && gamma && beta && y_f32 && M > 0 && D % 4 == 0 && D <= 1280, "csam_layernorm_cast: bad args");
  CSAM_REQUIRE(!ype_f16_or_null || pe_or_null, "csam_layernorm_cast: ype needs pe");
  hipLaunchKernelGGL(layernorm_cast_kernel, dim3(csam_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, M, D,
                     eps, y_f32, (half_t*)y_f16_or_null, pe_or_null, (half_t*)ype_f16_or_null);
  CSAM_LAUNCH_CHECK("csam_layernorm_cast");
  return CSAM_OK;
}

extern "C" int csam_sam_im2col(void* stream, const float* img_chw, int h, int w, const float* mean3,
                               const float* std3, void* out_f16) {
  CSAM_REQUIRE(img_chw && out_f16 && h > 0 && w > 0 && h <= 1024 && w <= 1024, "csam_sam_im2col: bad args");
  floatx4 m = {mean3[0], mean3[1], mean3[2], 0.f}, sd = {std3[0], std3[1], std3[2], 1.f};
  hipLaunchKernelGGL(sam_im2col_kernel, dim3(csam_cdiv(4096 * 96, 256)), dim3(256), 0, (hipStream_t)stream,
                     img_chw, h, w, m, sd, (half_t*)out_f16);
  CSAM_LAUNCH_CHECK("csam_sam_im2col");
  return CSAM_OK;
}

extern "C" int csam_dino_im2col(void* stream, const float* img_chw, int h, int w, int frame, const float* mean3,
                                const float* std3, void* out_f16) {
  CSAM_REQUIRE(img_chw && out_f16 && h > 0 && w > 0 && h <= frame && w <= frame && (frame == 1024 || frame == 1022),
               "csam_dino_im2col: bad args");
  floatx4 m = {mean3[0], mean3[1], mean3[2], 0.f}, sd = {std3[0], std3[1], std3[2], 1.f};
  hipLaunchKernelGGL(dino_im2col_kernel, dim3(csam_cdiv(5329L * 640, 256)), dim3(256), 0, (hipStream_t)stream,
                     img_chw, h, w, frame, m, sd, (half_t*)out_f16);
  CSAM_LAUNCH_CHECK("csam_dino_im2col");
  return CSAM_OK;
}

extern "C" int csam_im2col3x3(void* stream, const void* in_f16, void* out_f16, int C) {
  CSAM_REQUIRE(in_f16 && out_f16 && C % 8 == 0, "csam_im2col3x3: bad args");
  hipLaunchKernelGGL(im2col3x3_kernel, dim3(csam_cdiv(4096L * 9 * (C / 8), 256)), dim3(256), 0,
                     (hipStream_t)stream, (const half_t*)in_f16, (half_t*)out_f16, C);
  CSAM_LAUNCH_CHECK("csam_im2col3x3");
  return CSAM_OK;
}

extern "C" int csam_add_cast(void* stream, const float* a, const float* b, long b_row_stride, void* y_f16,
                             float* y_f32, long M, int N) {
  CSAM_REQUIRE(a && (y_f16 || y_f32) && N % 4 == 0, "csam_add_cast: bad args");
  hipLaunchKernelGGL(add_cast_kernel, dim3(csam_cdiv(M * (N / 4), 256)), dim3(256), 0, (hipStream_t)stream,
                     a, b, b_row_stride, (half_t*)y_f16, y_f32, M, N);
  CSAM_LAUNCH_CHECK("csam_add_cast");
  return CSAM_OK;
}
