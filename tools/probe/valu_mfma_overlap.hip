// developer probe (gfx950): do VALU and MFMA issue cycles ADD or OVERLAP on one SIMD?  
//   * one wave: MFMA-only, VALU-only, MFMA + k independent VALU interleaved, MFMA block then VALU block;
//   * two waves on the SAME SIMD (waves w and w+4 of an 8-wave workgroup; HW_ID printed): MFMA-only beside VALU-only,
//     MFMA beside MFMA, VALU beside VALU, interleaved beside interleaved, block beside block (free-running);
//   * two waves on DIFFERENT SIMDs of one CU for comparison.
// Output: shader cycles (s_memtime) per 16-unit body iteration and per unit, per wave.
//   hipcc --offload-arch=gfx950 -O2 -o valu_mfma_overlap valu_mfma_overlap.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

#define M0 "v_mfma_f32_16x16x32_f16 v[0:3], v[20:23], v[24:27], v[0:3]\n"
#define M1 "v_mfma_f32_16x16x32_f16 v[4:7], v[20:23], v[24:27], v[4:7]\n"
#define M2 "v_mfma_f32_16x16x32_f16 v[8:11], v[20:23], v[24:27], v[8:11]\n"
#define M3 "v_mfma_f32_16x16x32_f16 v[12:15], v[20:23], v[24:27], v[12:15]\n"
#define F(r) "v_fma_f32 v" #r ", v" #r ", v28, v29\n"
#define P(r, r1) "v_pk_fma_f32 v[" #r ":" #r1 "], v[" #r ":" #r1 "], v[28:29], v[30:31]\n"
#define E(r) "v_exp_f32 v" #r ", v" #r "\n"
#define C(r, r1) "v_cvt_pk_f16_f32 v" #r ", v" #r ", v" #r1 "\n"

// 16 units per body; fillers rotate over v32..v55 (no filler depends on one issued < 12 instructions earlier)
#define U4(MM, A, B, Cc, D) MM##0 A MM##1 B MM##2 Cc MM##3 D
#define MM0 M0
#define MM1 M1
#define MM2 M2
#define MM3 M3
#define NN0 ""
#define NN1 ""
#define NN2 ""
#define NN3 ""

#define F3a F(32) F(33) F(34)
#define F3b F(35) F(36) F(37)
#define F3c F(38) F(39) F(40)
#define F3d F(41) F(42) F(43)
#define F2a F(32) F(33)
#define F2b F(34) F(35)
#define F2c F(36) F(37)
#define F2d F(38) F(39)
#define F4a F(32) F(33) F(34) F(35)
#define F4b F(36) F(37) F(38) F(39)
#define F4c F(40) F(41) F(42) F(43)
#define F4d F(44) F(45) F(46) F(47)
#define F1a F(32)
#define F1b F(33)
#define F1c F(34)
#define F1d F(35)
#define P3a P(32, 33) P(34, 35) P(36, 37)
#define P3b P(38, 39) P(40, 41) P(42, 43)
#define P3c P(44, 45) P(46, 47) P(48, 49)
#define P3d P(50, 51) P(52, 53) P(54, 55)
#define P2a P(32, 33) P(34, 35)
#define P2b P(36, 37) P(38, 39)
#define P2c P(40, 41) P(42, 43)
#define P2d P(44, 45) P(46, 47)
#define E3a E(32) E(33) E(34)
#define E3b E(35) E(36) E(37)
#define E3c E(38) E(39) E(40)
#define E3d E(41) E(42) E(43)
#define E2a E(32) E(33)
#define E2b E(34) E(35)
#define E2c E(36) E(37)
#define E2d E(38) E(39)

#define X4(Q) Q Q Q Q
#define BODY_M X4(U4(MM, "", "", "", ""))
#define BODY_F X4(U4(NN, F3a, F3b, F3c, F3d))
#define BODY_MF3 X4(U4(MM, F3a, F3b, F3c, F3d))
#define BODY_MF2 X4(U4(MM, F2a, F2b, F2c, F2d))
#define BODY_MF4 X4(U4(MM, F4a, F4b, F4c, F4d))
#define BODY_MF1 X4(U4(MM, F1a, F1b, F1c, F1d))
#define BODY_BLK BODY_M BODY_F
#define BODY_P X4(U4(NN, P3a, P3b, P3c, P3d))
#define BODY_MP3 X4(U4(MM, P3a, P3b, P3c, P3d))
#define BODY_MP2 X4(U4(MM, P2a, P2b, P2c, P2d))
#define BODY_E X4(U4(NN, E3a, E3b, E3c, E3d))
#define BODY_ME3 X4(U4(MM, E3a, E3b, E3c, E3d))
#define BODY_ME2 X4(U4(MM, E2a, E2b, E2c, E2d))
#define BODY_F2 X4(U4(NN, F2a, F2b, F2c, F2d))
#define BODY_BLK2 BODY_M BODY_M BODY_F BODY_F

enum { K_M, K_F, K_MF3, K_MF2, K_MF4, K_MF1, K_BLK, K_P, K_MP3, K_MP2, K_E, K_ME3, K_ME2, K_BLK2, K_IDLE, K_N };
static const char* NAMES[] = {"16 MFMA", "48 v_fma", "16x(MFMA+3 v_fma)", "16x(MFMA+2 v_fma)", "16x(MFMA+4 v_fma)",
                              "16x(MFMA+1 v_fma)", "16 MFMA ; 48 v_fma", "48 v_pk_fma", "16x(MFMA+3 v_pk_fma)",
                              "16x(MFMA+2 v_pk_fma)", "48 v_exp", "16x(MFMA+3 v_exp)", "16x(MFMA+2 v_exp)",
                              "32 MFMA ; 96 v_fma", "idle"};

#define RUN(BODY)                                                                                            \
  asm volatile(                                                                                              \
      "v_mov_b32 v20, %3\n v_mov_b32 v21, %3\n v_mov_b32 v22, %3\n v_mov_b32 v23, %3\n"                       \
      "v_mov_b32 v24, %3\n v_mov_b32 v25, %3\n v_mov_b32 v26, %3\n v_mov_b32 v27, %3\n"                       \
      "v_mov_b32 v28, %4\n v_mov_b32 v29, %5\n v_mov_b32 v30, %4\n v_mov_b32 v31, %5\n"                       \
      "s_mov_b32 s20, %6\n"                                                                                  \
      "s_memrealtime s[26:27]\n s_memtime s[22:23]\n s_waitcnt lgkmcnt(0)\n"                                 \
      "1:\n" BODY                                                                                            \
      "s_sub_u32 s20, s20, 1\n s_cmp_lg_u32 s20, 0\n s_cbranch_scc1 1b\n"                                    \
      "s_nop 7\n s_nop 7\n s_memtime s[24:25]\n s_memrealtime s[28:29]\n s_waitcnt lgkmcnt(0)\n"             \
      "s_sub_u32 s22, s24, s22\n s_subb_u32 s23, s25, s23\n s_sub_u32 s26, s28, s26\n"                       \
      "v_mov_b32 %0, s22\n v_add_f32 %1, v0, v32\n v_mov_b32 %2, s26\n"                                      \
      : "=v"(dt), "=v"(sink), "=v"(rt)                                                                       \
      : "v"(hz), "v"(c0), "v"(c1), "s"(iters)                                                                \
      : "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", \
        "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", \
        "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", \
        "v50", "v51", "v52", "v53", "v54", "v55", "s20", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29", "scc", \
        "memory")

__device__ __forceinline__ void run_mode(int mode, int iters, unsigned& dt, float& sink, unsigned& rt) {
  const unsigned hz = 0x00010001u;   // two tiny fp16 values: accumulators stay finite
  const float c0 = 0.999f, c1 = 1e-3f;
  dt = 0;
  rt = 0;
  sink = 0.f;
  switch (mode) {
    case K_M: RUN(BODY_M); break;
    case K_F: RUN(BODY_F); break;
    case K_MF3: RUN(BODY_MF3); break;
    case K_MF2: RUN(BODY_MF2); break;
    case K_MF4: RUN(BODY_MF4); break;
    case K_MF1: RUN(BODY_MF1); break;
    case K_BLK: RUN(BODY_BLK); break;
    case K_P: RUN(BODY_P); break;
    case K_MP3: RUN(BODY_MP3); break;
    case K_MP2: RUN(BODY_MP2); break;
    case K_E: RUN(BODY_E); break;
    case K_ME3: RUN(BODY_ME3); break;
    case K_ME2: RUN(BODY_ME2); break;
    case K_BLK2: RUN(BODY_BLK2); break;
    default: break;
  }
}

// wave w runs mode modes.m[w] (-1: leaves at once); 16-wave workgroup: waves w, w+4, w+8, w+12 share a SIMD
struct Modes { int m[16]; };
__global__ __launch_bounds__(1024) void k(Modes modes, int iters, unsigned* out, float* sinkp) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  __syncthreads();
  const int mode = modes.m[wave];
  if (mode < 0 || mode == K_IDLE) return;
  unsigned dt, rt;
  float sink;
  run_mode(mode, 64, dt, sink, rt);      // warm-up (instruction cache)
  run_mode(mode, iters, dt, sink, rt);
  if ((threadIdx.x & 63) == 0) {
    out[wave * 2] = dt;
    out[wave * 2 + 1] = hwid;
    out[32 + wave] = rt;                 // the same interval in s_memrealtime ticks (100 MHz)
  }
  if (sink == 12345.f) *sinkp = sink;
}

static unsigned* d_out;
static float* d_sink;

static void many(const int* waves, const int* mds, int n) {
  const int iters = 2000;
  unsigned h[64] = {0};
  Modes m;
  for (int i = 0; i < 16; ++i) m.m[i] = -1;
  for (int i = 0; i < n; ++i) m.m[waves[i]] = mds[i];
  hipMemset(d_out, 0, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, m, iters, d_out, d_sink);
  hipDeviceSynchronize();
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  printf(" ");
  for (int i = 0; i < n; ++i)
    printf(" w%d(S%u) %s: %.1f |", waves[i], (h[waves[i] * 2 + 1] >> 4) & 3, NAMES[mds[i]], (double)h[waves[i] * 2] / iters);
  printf("\n");
}

static void pair(int wa, int ma, int wb, int mb) {
  const int iters = 2000;
  unsigned h[64] = {0};
  Modes m;
  for (int i = 0; i < 16; ++i) m.m[i] = -1;
  m.m[wa] = ma;
  if (wb >= 0) m.m[wb] = mb;
  hipMemset(d_out, 0, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(1024), 0, 0, m, iters, d_out, d_sink);
  hipDeviceSynchronize();
  hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost);
  auto simd = [&](int w) { return (h[w * 2 + 1] >> 4) & 3; };
  printf("  wave %d (SIMD %u) %-22s: %8.1f cycles / body [%.1f ns]", wa, simd(wa), NAMES[ma], (double)h[wa * 2] / iters,
         10.0 * h[32 + wa] / iters);
  if (wb >= 0 && mb != K_IDLE)
    printf("   ||   wave %d (SIMD %u) %-22s: %8.1f cycles / body", wb, simd(wb), NAMES[mb], (double)h[wb * 2] / iters);
  printf("\n");
}

int main() {
  hipMalloc(&d_out, 256);
  hipMalloc(&d_sink, 4);
  printf("one wave alone (body = 16 units; 16 MFMA 16x16x32 f16 = 256 pipe cycles at 16 each)\n");
  for (int m = 0; m < K_IDLE; ++m) pair(0, m, -1, -1);
  printf("two waves on the SAME SIMD (waves 0 and 4)\n");
  pair(0, K_M, 4, K_F);
  pair(0, K_M, 4, K_M);
  pair(0, K_F, 4, K_F);
  pair(0, K_M, 4, K_P);
  pair(0, K_M, 4, K_E);
  pair(0, K_MF3, 4, K_MF3);
  pair(0, K_MF2, 4, K_MF2);
  pair(0, K_BLK, 4, K_BLK);
  pair(0, K_BLK2, 4, K_BLK2);
  pair(0, K_MP3, 4, K_MP3);
  pair(0, K_M, 4, K_IDLE);
  printf("n waves on the SAME SIMD (waves 0, 4, 8, 12 of a 16-wave workgroup); cycles per body per wave\n");
  {
    const int w4[4] = {0, 4, 8, 12};
    const int sets[][4] = {{K_F, K_F, K_F, K_F}, {K_P, K_P, K_P, K_P}, {K_E, K_E, K_E, K_E}, {K_M, K_F, K_F, K_F},
                           {K_M, K_P, K_P, K_P}, {K_M, K_M, K_F, K_F}, {K_BLK, K_BLK, K_BLK, K_BLK},
                           {K_BLK2, K_BLK2, K_BLK2, K_BLK2}, {K_MF2, K_MF2, K_MF2, K_MF2}, {K_M, K_M, K_M, K_M}};
    for (auto& st : sets) {
      many(w4, st, 3);
      many(w4, st, 4);
    }
  }
  printf("two waves on DIFFERENT SIMDs (waves 0 and 1)\n");
  pair(0, K_M, 1, K_F);
  pair(0, K_M, 1, K_M);
  pair(0, K_F, 1, K_F);
  pair(0, K_BLK, 1, K_BLK);
  return 0;
}
