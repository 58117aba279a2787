"""Build libcsam_hip.so (gfx950) in-tree with hipcc.  No JIT cache: the .so travels with the repo."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# developer A/B builds: CSAM_BUILD_TAG=alt -> libcsam_hip_alt.so + build_alt/ (select it at run time with CSAM_LIB);
# CSAM_DEFS_<source stem> = extra hipcc flags for one source, e.g. CSAM_DEFS_decoder_fused="-DFOO -fno-slp-vectorize"
TAG = os.environ.get("CSAM_BUILD_TAG", "")
LIB = os.path.join(HERE, "libcsam_hip%s.so" % ("_" + TAG if TAG else ""))
ARCH = os.environ.get("CSAM_ARCH", "gfx950")     # developer: "gfx950:xnack+" for the sanitizer build (tools/asan_smoke.sh)
# CSAM_EXTRA_FLAGS: extra hipcc flags for EVERY source and the link (e.g. "-fsanitize=address -shared-libsan -g")
GLOBAL_FLAGS = os.environ.get("CSAM_EXTRA_FLAGS", "").split()
# ... except the sources named here (comma list of stems): the hand-scheduled inline-asm kernels do not survive instrumentation
# ("s" operands stop being provably uniform, 512-register kernels have no room for shadow checks)
GLOBAL_SKIP = set(filter(None, os.environ.get("CSAM_EXTRA_FLAGS_SKIP", "").split(",")))


# -amdgpu-mfma-vgpr-form (MFMA accumulators in the unified VGPR file instead of AGPRs: no v_accvgpr_read/write around a
# softmax) was used for the attention kernels in rounds 1-2 and is NOT any more: it lets the register allocator recycle an
# MFMA's SrcC quad as a ds_read destination `s_nop 2` later, and on gfx950 the LDS return can overtake the MFMA's late
# SrcC read under matrix-pipe contention (wrong scores in ~1 % of the flash-attention launches; csrc/attn_flash.hip has
# the analysis).  The score accumulators that the VALU reads are tied-operand inline-asm MFMAs in VGPRs instead.
# CSAM_VGPR_FORM=1 restores the old flags for A/B builds.
_VGPR_FORM = ["-mllvm", "-amdgpu-mfma-vgpr-form=1"] if os.environ.get("CSAM_VGPR_FORM") == "1" else []
EXTRA_FLAGS = {"attn_flash.hip": _VGPR_FORM + os.environ.get("CSAM_FLASH_DEFS", "").split(),
               "attn_window.hip": list(_VGPR_FORM)}


def sources():
    out = []
    for f in sorted(os.listdir(CSRC)):
        if f.endswith(".hip") or f.endswith(".cpp"):
            out.append(os.path.join(CSRC, f))
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build" + ("_" + TAG if TAG else ""))
    os.makedirs(objdir, exist_ok=True)
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(src)
                and all(os.path.getmtime(obj) > os.path.getmtime(os.path.join(CSRC, h))
                        for h in os.listdir(CSRC) if h.endswith((".h", ".inc")))):
            continue
        cmd = [hipcc, "-x", "hip", f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
               "-Wno-unused-result", "-Wno-unused-value"] + \
              ([] if os.path.basename(src).split(".")[0] in GLOBAL_SKIP else GLOBAL_FLAGS) + EXTRA_FLAGS.get(os.path.basename(src), []) + \
              os.environ.get("CSAM_DEFS_" + os.path.basename(src).split(".")[0], "").split() + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {src}")
        elif verbose and out.strip():
            print(out.decode())
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC"] + GLOBAL_FLAGS + ["-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
