// developer probe (gfx950): how many wait states does "v_mfma_f32_16x16x32_f16 reads SrcC" -> "VALU overwrites that VGPR"
// need?  hipcc (ROCm 7.2 LLVM) separates the two by s_nop 2; csam_flash_attn's rel-pos variant produced wrong scores in
// lanes 48..63 exactly where the compiler had re-used an accumulator-init quad that way (HISTORY.md section 4.1).
// For N = 0..9 wait states and P = 0..3 independent MFMAs queued ahead: D must equal A.B + C_old in every lane.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define BODY(NOPS, PRE)                                                                                      \
  asm volatile(                                                                                              \
      "v_mov_b32 v40, %4\n v_mov_b32 v41, %4\n v_mov_b32 v42, %4\n v_mov_b32 v43, %4\n"                       \
      "v_mov_b32 v48, 0\n v_mov_b32 v49, 0\n v_mov_b32 v50, 0\n v_mov_b32 v51, 0\n"                          \
      "s_nop 7\n s_nop 7\n" PRE                                                                               \
      "v_mfma_f32_16x16x32_f16 v[44:47], %6, %7, v[40:43]\n" NOPS                                            \
      "v_add_f32 v40, %5, %5\n v_add_f32 v41, %5, %5\n v_add_f32 v42, %5, %5\n v_add_f32 v43, %5, %5\n"       \
      "s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7\n"                                                                \
      "v_mov_b32 %0, v44\n v_mov_b32 %1, v45\n v_mov_b32 %2, v46\n v_mov_b32 %3, v47\n"                       \
      : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)                                                                \
      : "v"(cold), "v"(cnew), "v"(a), "v"(b)                                                                  \
      : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51")
#define PRE1 "v_mfma_f32_16x16x32_f16 v[48:51], %6, %7, v[48:51]\n"

template <int N, int P>
__global__ void k(float* out) {
  const int l = threadIdx.x;
  h8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.25f * ((l + e) % 7)); b[e] = (_Float16)(0.5f * ((l * 3 + e) % 5)); }
  const float cold = 100.f, cnew = 1000.f;
  float d0, d1, d2, d3;
#define N_NOP(n) if constexpr (N == n)
#define RUN(NOPS)                                                 \
  {                                                               \
    if constexpr (P == 0) BODY(NOPS, "");                          \
    if constexpr (P == 1) BODY(NOPS, PRE1);                        \
    if constexpr (P == 2) BODY(NOPS, PRE1 PRE1);                   \
    if constexpr (P == 3) BODY(NOPS, PRE1 PRE1 PRE1);              \
  }
  N_NOP(0) RUN("")
  N_NOP(1) RUN("s_nop 0\n")
  N_NOP(2) RUN("s_nop 1\n")
  N_NOP(3) RUN("s_nop 2\n")
  N_NOP(4) RUN("s_nop 3\n")
  N_NOP(5) RUN("s_nop 4\n")
  N_NOP(6) RUN("s_nop 5\n")
  N_NOP(7) RUN("s_nop 6\n")
  N_NOP(8) RUN("s_nop 7\n")
  N_NOP(9) RUN("s_nop 7\n s_nop 0\n")
  N_NOP(10) RUN("s_nop 7\n s_nop 1\n")
  N_NOP(11) RUN("s_nop 7\n s_nop 3\n")
  out[l * 4 + 0] = d0; out[l * 4 + 1] = d1; out[l * 4 + 2] = d2; out[l * 4 + 3] = d3;
}

static float ref[256];
template <int N, int P>
void run(float* d) {
  float h[256];
  hipLaunchKernelGGL((k<N, P>), dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  if (N == 11 && P == 0) for (int i = 0; i < 256; ++i) ref[i] = h[i];
  int bad = 0, bad_lo = 64, bad_hi = -1;
  for (int i = 0; i < 256; ++i)
    if (h[i] != ref[i]) { ++bad; if (i / 4 < bad_lo) bad_lo = i / 4; if (i / 4 > bad_hi) bad_hi = i / 4; }
  printf("wait states %2d, %d MFMAs queued ahead: %3d of 256 accumulator values wrong", N, P, bad);
  if (bad) printf(" (lanes %d..%d, e.g. got %.2f want %.2f)", bad_lo, bad_hi, h[bad_lo * 4], ref[bad_lo * 4]);
  printf("\n");
}
template <int P>
void sweep(float* d) {
  run<0, P>(d); run<1, P>(d); run<2, P>(d); run<3, P>(d); run<4, P>(d); run<5, P>(d); run<6, P>(d); run<7, P>(d);
  run<8, P>(d); run<9, P>(d); run<10, P>(d);
}
int main() {
  float* d;
  hipMalloc(&d, 256 * 4);
  run<11, 0>(d);                       // reference: 12 wait states
  sweep<0>(d); sweep<1>(d); sweep<2>(d); sweep<3>(d);
  return 0;
}
