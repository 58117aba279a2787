#!/usr/bin/env python
"""Build-time lint: an MFMA whose SrcC lives in VGPRs and is NOT its own vDst, followed closely by a load (LDS / global /
scratch return) or another non-MFMA write into those SrcC registers (a write-after-read on the accumulator input).

What is established (HISTORY.md section 4.2b): the hardware interlocks this case -- tools/probe/mfma_srcc_war.hip and
mfma_srcc_lds_war.hip found 0 wrong values with 0..32 wait states and 0..8 MFMAs queued ahead -- and for MFMAs the COMPILER
emits (intrinsics) LLVM's hazard recognizer additionally pads the overwrite (`s_nop 2` = 3 wait states in every site of this
tree).  The round-2 suspicion that such a site produced the intermittent flash-attention error was refuted in round 3 (the
cause was a VALU ordering slip around inline asm).  What is NOT covered by the compiler is an MFMA inside an inline-asm
block: nothing in the string is padded or tracked.  The rule therefore has two tiers, and no file is exempt:

  * MFMA inside `;;#ASMSTART ... ;;#ASMEND`: SrcC must be tied (vDst == SrcC) or in AGPRs, or nothing may overwrite it
    within WINDOW instructions (the attention kernels' generated blocks are tied by construction);
  * compiler-emitted MFMA: the overwrite must sit at least MIN_STATES wait states behind it -- i.e. the hazard recognizer's
    pad must be there.  A site closer than that means a compiler regression (or flags that disable the recognizer).

    python tools/lint_mfma_srcc.py            # compiles every csrc/*.hip to ISA with the build's flags and scans it
Exit status 1 on any violation; the per-file summary also counts the padded sites for the record."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WINDOW = 48     # instructions: ~ an LDS round trip under load
MIN_STATES = 3  # wait states LLVM puts between an MFMA's SrcC read and a load / VALU write of those VGPRs (`s_nop 2`)


def regs(tok):
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(3))
    m = re.match(r"([va])(\d+)$", tok)
    if m:
        return m.group(1), int(m.group(2)), int(m.group(2))
    return None


def scan(path):
    findings = []
    kernel = "?"
    lines = open(path).read().split("\n")
    in_asm = False
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+|\w+):\s", l)
        if m and not l.startswith(".L"):
            kernel = m.group(1)
        if ";;#ASMSTART" in l:
            in_asm = True
        elif ";;#ASMEND" in l:
            in_asm = False
        if "v_mfma" not in l and "v_smfmac" not in l:
            continue
        m = re.search(r"v_s?mfma\S*\s+(\S+), (\S+), (\S+), (\S+?)(\s|$)", l)
        if not m:
            continue
        d, a, b, c = m.group(1), m.group(2), m.group(3), m.group(4)
        rc = regs(c)
        if rc is None or rc[0] != "v" or c == d:
            continue
        states = 0
        for j in range(i + 1, min(i + 1 + WINDOW, len(lines))):
            t = lines[j].strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if t.startswith("s_endpgm") or t.startswith("s_branch") or t.startswith("s_cbranch"):
                break
            nop = re.match(r"s_nop\s+(\d+)", t)
            if nop:
                states += int(nop.group(1)) + 1
                continue
            mm = re.match(r"(\S+)\s+(\S+?),", t + ",")
            if not mm:
                states += 1
                continue
            op, dst = mm.group(1), mm.group(2)
            rd = regs(dst)
            if rd is None or rd[0] != "v" or rd[1] > rc[2] or rd[2] < rc[1]:
                states += 1
                continue
            kind = "load" if re.match(r"(ds_read|ds_load|global_load|buffer_load|flat_load|scratch_load)", op) else \
                   ("mfma" if "mfma" in op else "valu")
            # an MFMA taking the registers as its own accumulator chain is the normal dependent issue (0 states)
            bad = kind != "mfma" and (in_asm or states < MIN_STATES)
            findings.append((kernel, i + 1, l.strip(), j - i, t, kind, states, in_asm, bad))
            break
    return findings


def main():
    from crowdsam_amd import build as b
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out_dir = os.environ.get("CSAM_LINT_DIR", "/tmp/csam_lint")
    os.makedirs(out_dir, exist_ok=True)
    bad = 0
    for src in b.sources():
        if not src.endswith(".hip"):
            continue
        asm = os.path.join(out_dir, os.path.basename(src) + ".s")
        cmd = [hipcc, "-x", "hip", f"--offload-arch={b.ARCH}", "-O3", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value",
               "-S", "--cuda-device-only"] + b.EXTRA_FLAGS.get(os.path.basename(src), []) + ["-o", asm, src]
        subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        f = scan(asm)
        viol = [x for x in f if x[8]]
        padded = [x for x in f if x[5] != "mfma" and not x[8]]
        print("%-22s un-tied VGPR SrcC rewritten within %d instructions: %d by a load / %d by VALU behind the compiler's pad "
              "(min %s wait states), %d by the next MFMA of a chain; VIOLATIONS %d"
              % (os.path.basename(src), WINDOW, sum(x[5] == "load" for x in padded), sum(x[5] == "valu" for x in padded),
                 min([x[6] for x in padded], default="-"), sum(x[5] == "mfma" for x in f), len(viol)))
        for k, ln, ins, dist, t, kind, states, ia, _ in viol[:6]:
            print("    %s L%d%s: %s   ==> +%d instr / %d states  %s" % (k[:60], ln, " (inline asm)" if ia else "", ins, dist, states, t))
        bad += len(viol)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
