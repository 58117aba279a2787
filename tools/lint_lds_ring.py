#!/usr/bin/env python
"""Build-time report + gate: how far ahead of its MFMA is each LDS fragment read really issued?

What round 6 found in the ISA (no GPU needed to see it): for compiler-scheduled kernels LLVM's machine scheduler sinks every
`ds_read_b128` / `ds_read_b64_tr_b16` that feeds an MFMA to ONE MFMA before its use -- `ds_read; s_waitcnt lgkmcnt(0|1); v_mfma`
-- whatever ring depth the source spells and whatever `asm volatile("" ::: "memory")` fences stand between (the loads respect a
fence, the MFMAs float up to them).  csam_i2t_t2i's producers then wait an LDS round trip on each of the 32 P.M MFMAs of a tile
and its readers on each of their 64 per step; the upscaler's first conv on six of its eight k-steps.  `__builtin_amdgcn_
sched_barrier(0)` after every (MFMA, read) pair keeps the source order (decoder_fused.hip: FUSE_PM_PIPE, FUSE_RD_PIPE,
CSAM_UP_PIN).  This tool measures the result so that a compiler update (or an edit) that collapses a ring again is seen:

for every MFMA whose A or B operand was last written by an LDS read, DISTANCE = the number of MFMAs issued between that read
and the MFMA (same basic block or an earlier one of the same loop body; reads above a branch count from the branch).  Per kernel:
how many LDS-fed MFMAs, how many at distance <= 1 ("tight": less than ~32 cycles of cover for a >= 64-cycle LDS access).

    python tools/lint_lds_ring.py            # table for decoder_fused.hip's kernels; exit 1 when a gated kernel is over its bound
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# product kernels whose rings are pinned: (mangled-name fragment, max MFMAs at distance <= 1 per kernel body).  The bounds are
# what the pinned kernels show today plus slack for prologue / drain MFMAs; the collapsed forms had 150-400.
GATED = [
    ("14i2t_t2i_kernelILb0ELi1E", 40),      # layer 0: hoisted-Q producer + reader
    ("14i2t_t2i_kernelILb1ELi3E", 40),      # layer 1: projected producer, gamma / beta folded downstream + reader
    ("21upscale_stream_kernel", 24),        # first conv pinned; second conv / hyper product are interleaved with the GELU
]


def _regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.match(r"v(\d+)$", tok)
    if m:
        return int(m.group(1)), int(m.group(1))
    return None


def scan(path):
    """-> {kernel: (lds_fed_mfmas, tight, histogram {distance: count})}"""
    out = {}
    kernel = None
    last_read = {}          # vgpr -> MFMA counter at the time an LDS read wrote it
    n_mfma = 0
    for l in open(path):
        m = re.match(r"^(_Z\S+|\w+):\s", l)
        if m and not l.startswith(".L"):
            kernel = m.group(1)
            out[kernel] = [0, 0, {}]
            last_read, n_mfma = {}, 0
            continue
        if kernel is None:
            continue
        t = l.strip()
        if not t or t[0] in ";.":
            continue
        op = t.split()[0]
        if op.startswith(("ds_read", "ds_load")):
            m = re.match(r"\S+\s+(\S+?),", t)
            r = _regs(m.group(1)) if m else None
            if r:
                for v in range(r[0], r[1] + 1):
                    last_read[v] = n_mfma
            continue
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            m = re.search(r"v_s?mfma\S*\s+(\S+), (\S+), (\S+), (\S+?)(\s|$)", t)
            if m:
                dist = None
                for tok in (m.group(2), m.group(3)):
                    r = _regs(tok)
                    if r and r[0] in last_read:
                        d = n_mfma - last_read[r[0]]
                        dist = d if dist is None else min(dist, d)
                if dist is not None:
                    rec = out[kernel]
                    rec[0] += 1
                    rec[1] += dist <= 1
                    rec[2][dist] = rec[2].get(dist, 0) + 1
                # the destination is no longer LDS data
                r = _regs(m.group(1))
                if r:
                    for v in range(r[0], r[1] + 1):
                        last_read.pop(v, None)
            n_mfma += 1
            continue
        # any other VALU write to a register ends its "written by an LDS read" state
        if op.startswith("v_"):
            m = re.match(r"\S+\s+(\S+?),", t)
            r = _regs(m.group(1)) if m else None
            if r:
                for v in range(r[0], r[1] + 1):
                    last_read.pop(v, None)
    return {k: (v[0], v[1], v[2]) for k, v in out.items() if v[0]}


def compile_isa(src, out_dir, extra=()):
    from crowdsam_amd import build as b
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(out_dir, exist_ok=True)
    asm = os.path.join(out_dir, os.path.basename(src) + ("." + "_".join(x.strip("-").replace("=", "") for x in extra) if extra else "")
                       + ".s")
    cmd = [hipcc, "-x", "hip", f"--offload-arch={b.ARCH}", "-O3", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value",
           "-S", "--cuda-device-only"] + b.EXTRA_FLAGS.get(os.path.basename(src), []) + list(extra) + ["-o", asm, src]
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return asm


def main():
    from crowdsam_amd import build as b
    out_dir = os.environ.get("CSAM_LINT_DIR", "/tmp/csam_lint")
    src = os.path.join(b.CSRC, "decoder_fused.hip")
    extra = [a for a in sys.argv[1:] if a.startswith("-D")]
    res = scan(compile_isa(src, out_dir, extra))
    bad = 0
    print("%-64s %8s %8s   distance histogram (MFMAs between the LDS read and its MFMA)" % ("kernel", "LDS-fed", "tight"))
    for k in sorted(res):
        n, tight, hist = res[k]
        gate = next((g for g in GATED if g[0] in k), None)
        flag = ""
        if gate and not extra:
            flag = "  <= %d ok" % gate[1] if tight <= gate[1] else "  OVER %d" % gate[1]
            bad += tight > gate[1]
        h = " ".join("%d:%d" % (d, hist[d]) for d in sorted(hist)[:10])
        print("%-64s %8d %8d   %s%s" % (k.replace("_ZN12_GLOBAL__N_1", "")[:64], n, tight, h, flag))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
