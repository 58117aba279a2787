#!/usr/bin/env python
"""Generates crowdsam_amd/csrc/gemm4w_asm.inc: the hand-scheduled main loop of gemm4w_kernel (gemm_f16.hip) -- a 256 x 256 x 64
tile on FOUR waves, one per SIMD, each owning 128 x 128 outputs = 64 accumulator tiles in a[0:255] (the shape the vendor's
MT256x256x64 kernels use; image_encoder.py:227,238 / common.py:25-26 are the projections it serves).

One K tile (64 wide) per wave = 128 v_mfma_f32_16x16x32_f16 in two phases of 64 (k-step 0 / k-step 1), with everything else threaded
between them one instruction per MFMA gap:

    phase 0 of tile t : MFMAs on fragment set F0            | 16 ds_read_b128: set F1 of tile t (LDS stage t & 1)
                        s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier   -- tile t+1 has landed everywhere, stage t & 1 is read out
    phase 1 of tile t : MFMAs on fragment set F1            | 16 x (m0, global_load_lds 1 KB): tile t+2 into stage t & 1
                                                            | 16 ds_read_b128: set F0 of tile t+1 (stage (t+1) & 1)
                        s_waitcnt lgkmcnt(0)

so there is ONE barrier per 64-wide K tile, two LDS stages of 64 KB carry a prefetch distance of two tiles (the fragment sets in
registers are the third buffer), and an LDS-DMA piece is read no earlier than one barrier after the vmcnt that retired it.
Registers: W fragments of k-step ks in v[128 + 64 ks + 4 j], activation fragments in v[160 + 64 ks + 4 i] (i, j = 0..7),
acc[i][j] in a[(8 i + j) 4 ..]; the MFMA is issued "swapped" (W rows as SrcA) exactly as in gemm_f16_kernel, and the k order per
output element is the same ascending chain of 32-wide steps, so results are bit-identical to the other tile shapes.

    python tools/gen/gen_gemm4w_asm.py > crowdsam_amd/csrc/gemm4w_asm.inc
"""
import os
import sys

# developer ablations (timing only -- results are wrong): G4_ABLATE = comma list of noglds, noreads, nobarrier, flatglds
ABLATE = set(filter(None, os.environ.get("G4_ABLATE", "").split(",")))
MF = "v_mfma_f32_16x16x32_f16"
STAGE = 65536
W_OFF = 32768
# placement knobs (MFMA slot after which the r-th side instruction goes): "first:step" per stream, overridable for A/B builds with
# G4_PLACE="r0=0:2,g=0:3,r1=2:3,wait=counted"
PLACE = {"r0": (3, 3),      # phase 0: the 16 reads of fragment set F1
         "g": (1, 3),       # phase 1: m0 write at first + step * g, the LDS-DMA one slot later
         "r1": (3, 3),      # phase 1: the 16 reads of the next tile's set F0
         "wait": "phase"}   # "phase": lgkmcnt(0) at both phase ends; "counted": F0 reads stay in flight into phase 0 (counted waits)
#        "y": (slot,)       # optional: vmcnt wait + a second barrier at this phase-1 slot instead of vmcnt(0) at the phase-0 barrier
for kv in filter(None, os.environ.get("G4_PLACE", "").split(",")):
    k, v = kv.split("=")
    PLACE[k] = v if k == "wait" else tuple(int(x) for x in (v + ":1").split(":")[:2])
COUNTED = PLACE["wait"] == "counted"


def wreg(ks, j):
    b = 128 + 64 * ks + 4 * j
    return "v[%d:%d]" % (b, b + 3)


def areg(ks, i):
    b = 160 + 64 * ks + 4 * i
    return "v[%d:%d]" % (b, b + 3)


def acc(i, j):
    b = (8 * i + j) * 4
    return "a[%d:%d]" % (b, b + 3)


def mfma_list(ks):
    return ["%s %s, %s, %s, %s" % (MF, acc(i, j), wreg(ks, j), areg(ks, i), acc(i, j)) for i in range(8) for j in range(8)]


def frag_reads(ks, stage):
    """16 ds_read_b128 of fragment set ks from LDS stage `stage`, in the order the MFMA stream (i outer, j inner) first touches them:
    W0, A0, W1 .. W7, A1 .. A7.  Returns (register name, instruction)."""
    rw = lambda j: (("W", j), "ds_read_b128 %s, %%[rw%d%d] offset:%d" % (wreg(ks, j), stage, ks, j * 2048))
    ra = lambda i: (("A", i), "ds_read_b128 %s, %%[ra%d%d] offset:%d" % (areg(ks, i), stage, ks, i * 2048))
    return [rw(0), ra(0)] + [rw(j) for j in range(1, 8)] + [ra(i) for i in range(1, 8)]


BUF = "bufglds" in ABLATE       # LDS-DMA through buffer descriptors s[44:47] / s[48:51] + the K offset in s52


def glds_pairs(stage):
    """16 (m0 write, LDS-DMA) pairs: this wave's 8 activation pieces and 8 weight pieces of a stage."""
    out = []
    lda = "buffer_load_dwordx4 %%[oa%d], s[44:47], s52 offen lds" if BUF else "global_load_lds_dwordx4 %%[oa%d], s[40:41]"
    ldw = "buffer_load_dwordx4 %%[ow%d], s[48:51], s52 offen lds" if BUF else "global_load_lds_dwordx4 %%[ow%d], s[42:43]"
    for i in range(8):
        out.append(("s_add_u32 m0, %%[ldsw], %d" % (stage * STAGE + i * 1024), lda % i))
    for i in range(8):
        out.append(("s_add_u32 m0, %%[ldsw], %d" % (stage * STAGE + W_OFF + i * 1024), ldw % i))
    return out


ADVANCE = (["s_add_u32 s52, s52, 128"] if BUF else
           ["s_add_u32 s40, s40, 128", "s_addc_u32 s41, s41, 0", "s_add_u32 s42, s42, 128", "s_addc_u32 s43, s43, 0"])


def weave(mfmas, side):
    """side: dict slot -> list of instructions placed after MFMA `slot`."""
    out = []
    for k, m in enumerate(mfmas):
        out.append(m)
        out.extend(side.get(k, []))
    return out


def place(side, first_step, items):
    for r, ins in enumerate(items):
        side.setdefault(first_step[0] + first_step[1] * r, []).append(ins)


def phase0(stage, pending):
    """MFMAs on F0 with the F1 reads threaded in.  pending: F0 reads still in flight at entry (counted mode), oldest first, as register
    names; LDS operations return in order, so `s_waitcnt lgkmcnt(n)` before the first MFMA that touches a register, n = LDS operations
    issued after that register's read."""
    reads = [] if "noreads" in ABLATE else frag_reads(1, stage)
    side = {}
    place(side, PLACE["r0"], [ins for _, ins in reads])
    mf = mfma_list(0)
    if not pending:
        return weave(mf, side)
    out, queue, done = [], list(pending), set()     # queue: outstanding LDS ops in issue order (names; None for F1 reads)
    for k, m in enumerate(mf):
        i, j = divmod(k, 8)
        need = [len(queue) - 1 - queue.index(r) for r in (("W", j), ("A", i)) if r in queue]
        if need:
            c = min(min(need), 15)                      # the counter saturates at 15
            out.append("s_waitcnt lgkmcnt(%d)" % c)
            del queue[:len(queue) - c]
        out.append(m)
        for ins in side.get(k, []):
            out.append(ins)
            queue.append(None)
    return out


def tile(stage, with_glds, with_next):
    L = []
    if "noglds" in ABLATE:
        with_glds = False
    nxt = frag_reads(0, 1 - stage) if with_next and "noreads" not in ABLATE else []
    cur = frag_reads(0, stage) if "noreads" not in ABLATE else []
    # ---- phase 0
    L += phase0(stage, [name for name, _ in cur] if COUNTED else [])
    bar = [] if "nobarrier" in ABLATE else ["s_barrier"]
    LATE = "y" in PLACE                                  # second barrier: "the next tile has landed" moved into phase 1
    L += ["s_waitcnt lgkmcnt(0)" if LATE else "s_waitcnt vmcnt(0) lgkmcnt(0)"] + bar
    # ---- phase 1
    side = {}
    g_slots = []
    if with_glds:
        pairs = glds_pairs(stage)
        g_slots = [PLACE["g"][0] + 1 + PLACE["g"][1] * g for g in range(16)]
        assert g_slots[-1] + 4 <= 63, "LDS-DMA placement runs past the phase"
        place(side, PLACE["g"], [m0 for m0, _ in pairs])
        place(side, (PLACE["g"][0] + 1, PLACE["g"][1]), [ld for _, ld in pairs])
        place(side, (g_slots[-1] + 1, 1), ADVANCE)
    if LATE and with_next:
        ys = PLACE["y"][0]
        newer = sum(1 for q in g_slots if q < ys or (q == ys and False))   # pieces of tile t+2 issued before the wait
        # the wait + barrier go FIRST in their slot, ahead of any LDS-DMA placed in the same slot
        side.setdefault(ys, []).insert(0, "s_waitcnt vmcnt(%d)" % newer)
        for bi in bar:
            side[ys].insert(1, bi)
        assert PLACE["r1"][0] > ys, "the F0 reads must follow the second barrier"
    place(side, PLACE["r1"], [ins for _, ins in nxt])
    assert not nxt or PLACE["r1"][0] + PLACE["r1"][1] * 15 <= 63, "read placement runs past the phase"
    L += weave(mfma_list(1), side)
    if with_next and not COUNTED:
        L += ["s_waitcnt lgkmcnt(0)"]
    return L


def program():
    L = ["s_mov_b32 %[keep], m0", "s_mov_b64 s[40:41], %[pa]", "s_mov_b64 s[42:43], %[pw]"]
    if BUF:
        L += ["s_mov_b64 s[44:45], %[pa]", "s_and_b32 s45, s45, 0xffff", "s_mov_b32 s46, 0x80000000", "s_mov_b32 s47, 0x00020000",
              "s_mov_b64 s[48:49], %[pw]", "s_and_b32 s49, s49, 0xffff", "s_mov_b32 s50, 0x80000000", "s_mov_b32 s51, 0x00020000",
              "s_mov_b32 s52, 0", "s_nop 4"]
    # prologue: tiles 0 and 1 into stages 0 and 1
    for stage in range(2):
        for m0, ld in glds_pairs(stage):
            L += [m0, "s_nop 0", ld]
        L += ADVANCE
    for n in range(256):
        L.append("v_accvgpr_write_b32 a%d, 0" % n)
    L += ["s_waitcnt vmcnt(16)", "s_barrier"]
    L += [ins for _, ins in frag_reads(0, 0)]
    L += ([] if COUNTED else ["s_waitcnt lgkmcnt(0)"]) + ["s_cmp_eq_u32 %[niter], 0", "s_cbranch_scc1 L_g4_tail_%="]
    L += ["L_g4_loop_%=:"]
    L += tile(0, True, True)
    L += tile(1, True, True)
    L += ["s_sub_u32 %[niter], %[niter], 1", "s_cmp_lg_u32 %[niter], 0", "s_cbranch_scc1 L_g4_loop_%="]
    L += ["L_g4_tail_%=:"]
    L += tile(0, False, True)
    L += tile(1, False, False)
    L += ["s_nop 15", "s_mov_b32 m0, %[keep]"]
    return L


def emit():
    L = program()
    body = "\n".join('      "%s\\n\\t"' % l for l in L)
    ins = ['[pa] "s"(pa)', '[pw] "s"(pw)', '[ldsw] "s"(ldsw)']
    ins += ['[oa%d] "v"(oa[%d])' % (i, i) for i in range(8)]
    ins += ['[ow%d] "v"(ow[%d])' % (i, i) for i in range(8)]
    for s in range(2):
        for ks in range(2):
            ins.append('[ra%d%d] "v"(ra[%d][%d])' % (s, ks, s, ks))
            ins.append('[rw%d%d] "v"(rw[%d][%d])' % (s, ks, s, ks))
    clob = ['"memory"', '"scc"'] + ['"s%d"' % n for n in range(40, 53 if BUF else 44)] + ['"v%d"' % n for n in range(128, 256)] + ['"a%d"' % n for n in range(256)]

    def wrap(items, ind):
        out, cur = [], ind
        for it in items:
            if len(cur) + len(it) + 2 > 130:
                out.append(cur.rstrip())
                cur = ind
            cur += it + ", "
        out.append(cur.rstrip().rstrip(","))
        return "\n".join(out)

    print("// GENERATED by tools/gen/gen_gemm4w_asm.py -- do not edit.  %d instructions, %d MFMAs." % (len(L), sum(1 for l in L if l.startswith(MF))))
    print("// Main loop of gemm4w_kernel: K = 64 * (2 * niter + 2); accumulators are left in a[0:255] (acc[i][j] = a[(8 i + j) 4 ..]).")
    print("__device__ __forceinline__ void gemm4w_mainloop(const half_t* pa, const half_t* pw, unsigned ldsw, const unsigned (&oa)[8],")
    print("                                                const unsigned (&ow)[8], const unsigned (&ra)[2][2], const unsigned (&rw)[2][2],")
    print("                                                int niter) {")
    print("  unsigned keep;")
    print("  asm volatile(")
    print(body)
    print('      : [keep] "=&s"(keep), [niter] "+s"(niter)')
    print("      : " + wrap(ins, "        ").lstrip())
    print("      : " + wrap(clob, "        ").lstrip() + ");")
    print("}")
    print()
    print("// accumulator tile (i, j) of the wave, read back after the main loop (the MFMA -> accvgpr_read wait states are the s_nop 15 above)")
    print("template <int T>")
    print("__device__ __forceinline__ floatx4 gemm4w_acc() {")
    print("  float a, b, c, d;")
    print('  asm volatile("v_accvgpr_read_b32 %0, a[%c4]\\n\\tv_accvgpr_read_b32 %1, a[%c5]\\n\\tv_accvgpr_read_b32 %2, a[%c6]\\n\\tv_accvgpr_read_b32 %3, a[%c7]"')
    print('               : "=v"(a), "=v"(b), "=v"(c), "=v"(d)')
    print('               : "i"(4 * T), "i"(4 * T + 1), "i"(4 * T + 2), "i"(4 * T + 3));')
    print("  return floatx4{a, b, c, d};")
    print("}")


if __name__ == "__main__":
    emit()
