#!/usr/bin/env python
"""Generates crowdsam_amd/csrc/attn_window_asm.inc: the MFMA groups of win_attn2_kernel as inline-asm blocks with TIED
accumulators, their LDS fragment reads (ring of four quads, counted lgkmcnt waits) and the wait states the compiler's
hazard recogniser cannot insert around MFMAs it does not see (attn_flash.hip has the reasoning; the rule: an MFMA's SrcC
registers are written by nothing but that MFMA until it has completed).

Two sets: head_dim 64 (two 32-wide k-steps, four 16-dim output tiles, 144-B K rows) and head_dim 80 (ViT-H: 32 + 32 + 16 --
the last k-step is a 16x16x16 MFMA on 8-byte fragments, so nothing is padded --, five output tiles, 176-B K rows).

    python tools/gen/gen_attn_window_asm.py > crowdsam_amd/csrc/attn_window_asm.inc
"""
MF32 = "v_mfma_f32_16x16x32_f16"
MF16 = "v_mfma_f32_16x16x16_f16"


def block(lines):
    return "\n".join('      "%s\\n\\t"' % l for l in lines[:-1]) + '\n      "%s"' % lines[-1]


def ring_sequence(elems, load, mfma):
    """elems: list of fragment descriptors; load(e, ring_slot) / mfma(e, ring_slot) -> asm text.  Four loads in flight."""
    out = []
    n = len(elems)
    for i in range(min(4, n)):
        out.append(load(elems[i], i % 4))
    for i, e in enumerate(elems):
        out.append("s_waitcnt lgkmcnt(%d)" % min(3, n - 1 - i))
        out.append(mfma(e, i % 4))
        if i + 4 < n:
            out.append(load(elems[i + 4], i % 4))
    return out


def gen_bias():
    # operands: %0-%3 t[nt] (out); %4..%11 rf[nt][ks] = %(4 + 2*nt + ks); %12, %13 qf[ks]
    L = ["s_nop 1"]
    for ks in range(2):
        for nt in range(4):
            L.append("%s %%%d, %%%d, %%%d, %s" % (MF32, nt, 4 + 2 * nt + ks, 12 + ks, "0" if ks == 0 else "%%%d" % nt))
    L += ["s_nop 7", "s_nop 3"]
    return ("// T[j][q] = relcat[j] . q for the 64 relcat rows (4 row tiles), both k-steps\n"
            "__device__ __forceinline__ void win_bias_mfma(floatx4 (&t)[4], const half8_t (&rf)[4][2], const half8_t (&qf)[2]) {\n"
            "  asm volatile(\n" + block(L) + "\n"
            '      : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])\n'
            '      : "v"(rf[0][0]), "v"(rf[0][1]), "v"(rf[1][0]), "v"(rf[1][1]), "v"(rf[2][0]), "v"(rf[2][1]), "v"(rf[3][0]),\n'
            '        "v"(rf[3][1]), "v"(qf[0]), "v"(qf[1]));\n}\n')


def gen_scores():
    # operands: %0-%12 p[kt] (out); %13-%16 ring; %17, %18 qf[ks]; %19 kaddr
    groups = [(0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (10, 11, 12)]
    elems = [(kt, ks) for g in groups for ks in range(2) for kt in g]
    load = lambda e, r: "ds_read_b128 %%%d, %%19 offset:%d" % (13 + r, e[0] * 16 * 144 + e[1] * 64)
    mfma = lambda e, r: "%s %%%d, %%%d, %%%d, %s" % (MF32, e[0], 13 + r, 17 + e[1], "0" if e[1] == 0 else "%%%d" % e[0])
    L = ["s_nop 1"] + ring_sequence(elems, load, mfma) + ["s_nop 7", "s_nop 3"]
    outs = ", ".join('"=&v"(p[%d])' % i for i in range(13))
    return ("// S_raw[key tile kt][query] = K . q over both k-steps; K fragment of (kt, ks) at kaddr + kt*16*144 + ks*64 bytes\n"
            "__device__ __forceinline__ void win_scores_mfma(floatx4 (&p)[14], const half8_t (&qf)[2], unsigned kaddr) {\n"
            "  half8_t r0, r1, r2, r3;\n  asm volatile(\n" + block(L) + "\n"
            "      : " + outs + ',\n        "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)\n'
            '      : "v"(qf[0]), "v"(qf[1]), "v"(kaddr)\n      : "memory");\n}\n')


def gen_pv():
    # operands: %0-%3 o[dt] (out); %4-%7 ring; %8-%14 pf[s2]; %15-%18 vaddr[dt]
    elems = [(s2, dt) for s2 in range(7) for dt in range(4)]
    load = lambda e, r: "ds_read2_b64 %%%d, %%%d offset0:%d offset1:%d" % (4 + r, 15 + e[1], e[0] * 8, e[0] * 8 + 4)
    mfma = lambda e, r: "%s %%%d, %%%d, %%%d, %s" % (MF32, e[1], 4 + r, 8 + e[0], "0" if e[0] == 0 else "%%%d" % e[1])
    L = ["s_nop 1"] + ring_sequence(elems, load, mfma) + ["s_nop 7", "s_nop 3"]
    return ("// O^T[dt] = sum over the 7 k-steps of V^T(dt, s2) . P^T(s2); V^T fragment = two 8-byte halves 32 bytes apart\n"
            "__device__ __forceinline__ void win_pv_mfma(floatx4 (&o)[4], const half8_t (&pf)[7], const unsigned (&va)[4]) {\n"
            "  half8_t r0, r1, r2, r3;\n  asm volatile(\n" + block(L) + "\n"
            '      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)\n'
            '      : "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(pf[4]), "v"(pf[5]), "v"(pf[6]), "v"(va[0]), "v"(va[1]),\n'
            '        "v"(va[2]), "v"(va[3])\n      : "memory");\n}\n')


# ------------------------------------------------------------------------------------------------- head_dim 80
KP80 = 176        # bytes per K row: 80 halfs + 8 pad


def gen_bias80():
    # %0-%3 t[nt]; %4..%11 rf[nt][ks<2]; %12..%15 rg[nt] (half4, dims 64..79); %16, %17 qf[ks<2]; %18 qg (half4)
    L = ["s_nop 1"]
    for ks in range(3):
        for nt in range(4):
            if ks < 2:
                L.append("%s %%%d, %%%d, %%%d, %s" % (MF32, nt, 4 + 2 * nt + ks, 16 + ks, "0" if ks == 0 else "%%%d" % nt))
            else:
                L.append("%s %%%d, %%%d, %%18, %%%d" % (MF16, nt, 12 + nt, nt))
    L += ["s_nop 7", "s_nop 3"]
    return ("// head_dim 80: T[j][q] = relcat[j] . q over dims 0..31, 32..63 (16x16x32) and 64..79 (16x16x16)\n"
            "__device__ __forceinline__ void win_bias_mfma80(floatx4 (&t)[4], const half8_t (&rf)[4][2], const half4_t (&rg)[4],\n"
            "                                                const half8_t (&qf)[2], const half4_t& qg) {\n"
            "  asm volatile(\n" + block(L) + "\n"
            '      : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])\n'
            '      : "v"(rf[0][0]), "v"(rf[0][1]), "v"(rf[1][0]), "v"(rf[1][1]), "v"(rf[2][0]), "v"(rf[2][1]), "v"(rf[3][0]),\n'
            '        "v"(rf[3][1]), "v"(rg[0]), "v"(rg[1]), "v"(rg[2]), "v"(rg[3]), "v"(qf[0]), "v"(qf[1]), "v"(qg));\n}\n')


def gen_scores80():
    # %0-%12 p[kt]; %13-%16 ring (16 B); %17-%20 ring (8 B); %21, %22 qf[ks<2]; %23 qg; %24 kaddr (16-B chunks); %25 kaddr16
    groups = [(0, 1), (2, 3), (4, 5), (6, 7), (8, 9), (10, 11, 12)]
    elems = [(kt, ks) for g in groups for ks in range(3) for kt in g]

    def load(e, r):
        if e[1] < 2:
            return "ds_read_b128 %%%d, %%24 offset:%d" % (13 + r, e[0] * 16 * KP80 + e[1] * 64)
        return "ds_read_b64 %%%d, %%25 offset:%d" % (17 + r, e[0] * 16 * KP80)

    def mfma(e, r):
        if e[1] < 2:
            return "%s %%%d, %%%d, %%%d, %s" % (MF32, e[0], 13 + r, 21 + e[1], "0" if e[1] == 0 else "%%%d" % e[0])
        return "%s %%%d, %%%d, %%23, %%%d" % (MF16, e[0], 17 + r, e[0])

    L = ["s_nop 1"] + ring_sequence(elems, load, mfma) + ["s_nop 7", "s_nop 3"]
    outs = ", ".join('"=&v"(p[%d])' % i for i in range(13))
    return ("// head_dim 80: S_raw[key tile kt][query] = K . q; K fragment of (kt, ks < 2) at kaddr + kt*16*%d + ks*64 bytes,\n"
            "// of (kt, 2) -- dims 64 + 4 fg .. +3 -- at kaddr16 + kt*16*%d\n" % (KP80, KP80) +
            "__device__ __forceinline__ void win_scores_mfma80(floatx4 (&p)[14], const half8_t (&qf)[2], const half4_t& qg,\n"
            "                                                  unsigned kaddr, unsigned kaddr16) {\n"
            "  half8_t r0, r1, r2, r3;\n  half4_t h0, h1, h2, h3;\n  asm volatile(\n" + block(L) + "\n"
            "      : " + outs + ',\n        "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3)\n'
            '      : "v"(qf[0]), "v"(qf[1]), "v"(qg), "v"(kaddr), "v"(kaddr16)\n      : "memory");\n}\n')


def gen_pv80():
    # %0-%4 o[dt]; %5-%8 ring; %9-%15 pf[s2]; %16-%20 vaddr[dt]
    elems = [(s2, dt) for s2 in range(7) for dt in range(5)]
    load = lambda e, r: "ds_read2_b64 %%%d, %%%d offset0:%d offset1:%d" % (5 + r, 16 + e[1], e[0] * 8, e[0] * 8 + 4)
    mfma = lambda e, r: "%s %%%d, %%%d, %%%d, %s" % (MF32, e[1], 5 + r, 9 + e[0], "0" if e[0] == 0 else "%%%d" % e[1])
    L = ["s_nop 1"] + ring_sequence(elems, load, mfma) + ["s_nop 7", "s_nop 3"]
    return ("// head_dim 80: O^T[dt], dt < 5\n"
            "__device__ __forceinline__ void win_pv_mfma80(floatx4 (&o)[5], const half8_t (&pf)[7], const unsigned (&va)[5]) {\n"
            "  half8_t r0, r1, r2, r3;\n  asm volatile(\n" + block(L) + "\n"
            '      : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)\n'
            '      : "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]), "v"(pf[4]), "v"(pf[5]), "v"(pf[6]), "v"(va[0]), "v"(va[1]),\n'
            '        "v"(va[2]), "v"(va[3]), "v"(va[4])\n      : "memory");\n}\n')


if __name__ == "__main__":
    print("// GENERATED by tools/gen/gen_attn_window_asm.py -- do not edit.  Tied-accumulator MFMA groups of win_attn2_kernel.")
    print(gen_bias())
    print(gen_scores())
    print(gen_pv())
    print(gen_bias80())
    print(gen_scores80())
    print(gen_pv80())
