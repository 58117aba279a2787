// developer probe (gfx950): "v_mfma reads SrcC from VGPRs" -> "an LDS / global load RETURN overwrites those VGPRs".
// hipcc (ROCm 7.2) with -amdgpu-mfma-vgpr-form separates the two by s_nop 2 and then recycles the SrcC quad as a
// ds_read_b128 destination (csam_flash_attn main loop).  The VALU-overwrite form of this hazard is interlocked
// (profiles/r02_mfma_srcc_war_probe.txt); a load return is asynchronous.  With P MFMAs queued ahead of the reader the
// matrix pipe may take the SrcC operand late: does the returning load win the race?
// For P = 0..12 MFMAs ahead, N wait states, loader = ds_read_b128 | global_load_dwordx4: D must equal A.B + C_old.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define PRE4                                                     \
  "v_mfma_f32_16x16x32_f16 v[48:51], %6, %7, v[48:51]\n"         \
  "v_mfma_f32_16x16x32_f16 v[52:55], %6, %7, v[52:55]\n"         \
  "v_mfma_f32_16x16x32_f16 v[56:59], %6, %7, v[56:59]\n"         \
  "v_mfma_f32_16x16x32_f16 v[60:63], %6, %7, v[60:63]\n"
#define PRE1 "v_mfma_f32_16x16x32_f16 v[48:51], %6, %7, v[48:51]\n"
#define PRE2 "v_mfma_f32_16x16x32_f16 v[48:51], %6, %7, v[48:51]\n v_mfma_f32_16x16x32_f16 v[52:55], %6, %7, v[52:55]\n"

#define BODY(PRE, NOPS, LOAD)                                                                              \
  asm volatile(                                                                                            \
      "v_mov_b32 v40, %4\n v_mov_b32 v41, %4\n v_mov_b32 v42, %4\n v_mov_b32 v43, %4\n"                     \
      "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n"                        \
      "v_mov_b32 v52, 0\n v_mov_b32 v53, 0\n v_mov_b32 v54, 0\n v_mov_b32 v55, 0\n"                        \
      "v_mov_b32 v56, 0\n v_mov_b32 v57, 0\n v_mov_b32 v58, 0\n v_mov_b32 v59, 0\n"                        \
      "v_mov_b32 v60, 0\n v_mov_b32 v61, 0\n v_mov_b32 v62, 0\n v_mov_b32 v63, 0\n"                        \
      "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n" PRE                                                         \
      "v_mfma_f32_16x16x32_f16 v[44:47], %6, %7, v[40:43]\n" NOPS LOAD                                     \
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                                    \
      "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n"                     \
      "v_mov_b32 %0, v44\n v_mov_b32 %1, v45\n v_mov_b32 %2, v46\n v_mov_b32 %3, v47\n"                     \
      : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)                                                              \
      : "v"(cold), "v"(laddr), "v"(a), "v"(b), "v"(gaddr)                                                   \
      : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", \
        "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "memory")
#define LDS_LOAD "ds_read_b128 v[40:43], %5\n"
#define GLB_LOAD "global_load_dwordx4 v[40:43], %8, off\n"

template <int P, int N, int KIND>
__global__ void k(float* out, const float* gsrc) {
  __shared__ __attribute__((aligned(16))) float lds[256];
  const int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = 1000.f;
  __syncthreads();
  h8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.25f * ((l + e) % 7)); b[e] = (_Float16)(0.5f * ((l * 3 + e) % 5)); }
  const float cold = 100.f;
  const unsigned laddr = (unsigned)(unsigned long)(__attribute__((address_space(3))) void*)lds + l * 16;
  const float* gaddr = gsrc + l * 4;
  float d0, d1, d2, d3;
#define NOP_CASE(n, S) if constexpr (N == n) {                                                        \
    if constexpr (KIND == 0) {                                                                        \
      if constexpr (P == 0) BODY("", S, LDS_LOAD);                                                     \
      if constexpr (P == 1) BODY(PRE1, S, LDS_LOAD);                                                   \
      if constexpr (P == 2) BODY(PRE2, S, LDS_LOAD);                                                   \
      if constexpr (P == 4) BODY(PRE4, S, LDS_LOAD);                                                   \
      if constexpr (P == 6) BODY(PRE4 PRE2, S, LDS_LOAD);                                              \
      if constexpr (P == 8) BODY(PRE4 PRE4, S, LDS_LOAD);                                              \
      if constexpr (P == 12) BODY(PRE4 PRE4 PRE4, S, LDS_LOAD);                                        \
    } else {                                                                                          \
      if constexpr (P == 0) BODY("", S, GLB_LOAD);                                                     \
      if constexpr (P == 1) BODY(PRE1, S, GLB_LOAD);                                                   \
      if constexpr (P == 2) BODY(PRE2, S, GLB_LOAD);                                                   \
      if constexpr (P == 4) BODY(PRE4, S, GLB_LOAD);                                                   \
      if constexpr (P == 6) BODY(PRE4 PRE2, S, GLB_LOAD);                                              \
      if constexpr (P == 8) BODY(PRE4 PRE4, S, GLB_LOAD);                                              \
      if constexpr (P == 12) BODY(PRE4 PRE4 PRE4, S, GLB_LOAD);                                        \
    }                                                                                                 \
  }
  NOP_CASE(0, "")
  NOP_CASE(3, "s_nop 2\n")
  NOP_CASE(8, "s_nop 7\n")
  NOP_CASE(16, "s_nop 7\n s_nop 7\n")
  NOP_CASE(32, "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n")
  out[l * 4 + 0] = d0; out[l * 4 + 1] = d1; out[l * 4 + 2] = d2; out[l * 4 + 3] = d3;
}

static float ref[256];
static float *d, *g;
template <int P, int N, int KIND>
void run(int reps) {
  float h[256];
  int bad_launches = 0, bad_vals = 0;
  float ex = 0;
  for (int r = 0; r < reps; ++r) {
    hipLaunchKernelGGL((k<P, N, KIND>), dim3(1), dim3(64), 0, 0, d, g);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    if (P == 0 && N == 32 && KIND == 0 && r == 0) for (int i = 0; i < 256; ++i) ref[i] = h[i];
    int bad = 0;
    for (int i = 0; i < 256; ++i) if (h[i] != ref[i]) { ++bad; ex = h[i]; }
    bad_vals += bad;
    bad_launches += bad != 0;
  }
  printf("%s overwrite, %2d MFMAs queued ahead, %2d wait states: %3d of %d launches wrong (%d values%s", KIND ? "global_load" : "ds_read    ",
         P, N, bad_launches, reps, bad_vals, bad_vals ? ", e.g. got " : ")\n");
  if (bad_vals) printf("%.2f)\n", ex);
}
template <int P, int KIND>
void sweepN(int reps) { run<P, 0, KIND>(reps); run<P, 3, KIND>(reps); run<P, 8, KIND>(reps); run<P, 16, KIND>(reps); run<P, 32, KIND>(reps); }
int main() {
  hipMalloc(&d, 256 * 4);
  hipMalloc(&g, 256 * 4);
  float hv[256];
  for (int i = 0; i < 256; ++i) hv[i] = 1000.f;
  hipMemcpy(g, hv, sizeof(hv), hipMemcpyHostToDevice);
  run<0, 32, 0>(1);                       // reference: nothing queued, 32 wait states
  const int R = 50;
  sweepN<0, 0>(R); sweepN<1, 0>(R); sweepN<2, 0>(R); sweepN<4, 0>(R); sweepN<6, 0>(R); sweepN<8, 0>(R); sweepN<12, 0>(R);
  sweepN<0, 1>(R); sweepN<4, 1>(R); sweepN<8, 1>(R); sweepN<12, 1>(R);
  return 0;
}
