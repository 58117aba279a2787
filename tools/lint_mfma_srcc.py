#!/usr/bin/env python
"""Build-time lint for the gfx950 hazard found in round 3 (csrc/attn_flash.hip): an MFMA whose SrcC lives in VGPRs and
is NOT its own vDst, followed closely by a load (LDS / global / scratch return) or any other non-MFMA write into those
SrcC registers.  hipcc separates the two by `s_nop 2`; under matrix-pipe contention the load return can land before the
MFMA has taken SrcC for its last lane group.  AGPR accumulators (a[..]) and tied VGPR accumulators (vDst == SrcC) are
safe by construction.

    python tools/lint_mfma_srcc.py            # compiles every csrc/*.hip to ISA with the build's flags and scans it
Exit status 1 if a load overwrites an un-tied VGPR SrcC within WINDOW instructions."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
WINDOW = 48     # instructions: ~ an LDS round trip under load


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
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\S+|\w+):\s", l)
        if m and not l.startswith(".L"):
            kernel = m.group(1)
        if "v_mfma" not in l and "v_smfmac" not in l:
            continue
        m = re.search(r"v_s?mfma\S*\s+(\S+), (\S+), (\S+), (\S+?)(\s|$)", l)
        if not m:
            continue
        d, a, b, c = m.group(1), m.group(2), m.group(3), m.group(4)
        rc = regs(c)
        if rc is None or rc[0] != "v" or c == d:
            continue
        for j in range(i + 1, min(i + 1 + WINDOW, len(lines))):
            t = lines[j].strip()
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if t.startswith("s_endpgm") or t.startswith("s_branch") or t.startswith("s_cbranch"):
                break
            mm = re.match(r"(\S+)\s+(\S+?),", t + ",")
            if not mm:
                continue
            op, dst = mm.group(1), mm.group(2)
            rd = regs(dst)
            if rd is None or rd[0] != "v" or rd[1] > rc[2] or rd[2] < rc[1]:
                continue
            kind = "load" if re.match(r"(ds_read|ds_load|global_load|buffer_load|flat_load|scratch_load)", op) else \
                   ("mfma" if "mfma" in op else "valu")
            findings.append((kernel, i + 1, l.strip(), j - i, t, kind))
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
        loads = [x for x in f if x[5] == "load"]
        print("%-22s un-tied VGPR SrcC overwritten within %d instructions: %d by a load, %d by VALU, %d by another MFMA"
              % (os.path.basename(src), WINDOW, len(loads), sum(x[5] == "valu" for x in f), sum(x[5] == "mfma" for x in f)))
        for k, ln, ins, dist, t, kind in loads[:6]:
            print("    %s L%d: %s   ==> +%d  %s" % (k[:60], ln, ins, dist, t))
        bad += len(loads)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
