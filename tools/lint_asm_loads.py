#!/usr/bin/env python
"""Build-time lint for two inline-asm hazards found in round 3 (csrc/decoder_fused.hip, csam_i2t_t2i):

 1. IN-FLIGHT ASM LOAD READ.  `asm volatile("global_load_dwordx4 %0, ..." : "=v"(r))` returns immediately; the compiler
    believes r is defined and may COPY it (a v_mov on a loop edge, a live-range split) before the data has landed -- the
    copy carries stale registers.  Every instruction that reads a register written by an inline-asm global/buffer load
    must be preceded by an `s_waitcnt vmcnt(..)` issued after that load.  (Waits with N > 0 are counted as covering the
    load only if at most N vector-memory instructions were issued after it.)
 2. UNPADDED WIDE STORE.  gfx940+: a VMEM store of more than 64 bits followed by a VALU write of its data VGPRs needs 2
    wait states; LLVM's hazard recognizer pads its own stores but does not look inside inline asm.

    python tools/lint_asm_loads.py     # compiles every csrc/*.hip to ISA with the build's flags and scans it linearly
The scan is linear over the instruction stream of each kernel (fall-through order): loops are covered because the
back-edge copies sit at the bottom of the loop body, after the loads they would copy.  Exit status 1 on a finding."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VMEM = re.compile(r"^(global_|buffer_|flat_|scratch_)(load|store|atomic)")


def reg_range(tok):
    tok = tok.strip().rstrip(",")
    m = re.match(r"v\[(\d+):(\d+)\]$", tok)
    if m:
        return int(m.group(1)), int(m.group(2))
    m = re.match(r"v(\d+)$", tok)
    if m:
        return int(m.group(1)), int(m.group(1))
    return None


def operands(text):
    parts = text.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    return parts[0], [t.strip() for t in parts[1].split(",")]


def scan(path):
    findings = []
    kernel = "?"
    inflight = []          # [lo, hi, issue_index_of_vmem, line, text]
    vmem_issued = 0
    in_asm = False
    lines = open(path).read().split("\n")
    for i, raw in enumerate(lines):
        m = re.match(r"^(_Z\S+|\w+):\s", raw)
        if m and not raw.startswith(".L"):
            kernel, inflight, vmem_issued = m.group(1), [], 0
        t = raw.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        t = t.split(";")[0].strip()
        op, ops = operands(t)
        if op == "s_endpgm":
            inflight = []
            continue
        if op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", t)
            if m:
                n = int(m.group(1))
                inflight = [f for f in inflight if vmem_issued - f[2] <= n and n > 0]
            continue
        # reads of in-flight registers: every operand except the first (destination) -- and for stores / MFMA C all of them
        is_store = "store" in op
        srcs = ops if is_store else ops[1:]
        for s in srcs:
            r = reg_range(s)
            if r is None:
                continue
            for f in inflight:
                if r[0] <= f[1] and r[1] >= f[0]:
                    findings.append(("inflight", kernel, i + 1, t, f[3], f[4]))
        # a non-asm write into an in-flight register is the compiler recycling it: also a finding
        if ops and not is_store and not (in_asm and VMEM.match(op)):
            r = reg_range(ops[0])
            if r is not None:
                for f in inflight:
                    if r[0] <= f[1] and r[1] >= f[0]:
                        findings.append(("recycled", kernel, i + 1, t, f[3], f[4]))
        if VMEM.match(op):
            vmem_issued += 1
            if in_asm and "load" in op and "lds" not in op and ops:
                r = reg_range(ops[0])
                if r is not None:
                    inflight = [f for f in inflight if not (r[0] <= f[1] and r[1] >= f[0])]   # re-issued into the same quad
                    inflight.append([r[0], r[1], vmem_issued, i + 1, t])
            if in_asm and is_store and re.search(r"dwordx[34]", op):
                nxt = lines[i + 1].strip() if i + 1 < len(lines) else ""
                if not re.match(r"s_nop\s+[1-9]", nxt):
                    findings.append(("store", kernel, i + 1, t, i + 2, nxt))
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
        print("%-22s in-flight asm load read: %d, recycled: %d, unpadded wide asm store: %d"
              % (os.path.basename(src), sum(x[0] == "inflight" for x in f), sum(x[0] == "recycled" for x in f),
                 sum(x[0] == "store" for x in f)))
        for kind, k, ln, ins, l0, t0 in f[:8]:
            print("    [%s] %s L%d: %s   <== L%s %s" % (kind, k[:48], ln, ins, l0, t0))
        bad += len(f)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
