"""The ISA lints of tools/ as a build gate (no GPU: hipcc cross-compiles to gfx950 assembly).

tools/lint_asm_loads.py: no instruction may read, and the compiler may not recycle, a register an inline-asm global load
is still filling (the bug class behind the intermittent Y differences of the first csam_i2t_t2i), and every inline-asm
store wider than 64 bits is padded for the gfx940+ store-data hazard."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_inflight_asm_load_is_read_or_recycled():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lint_asm_loads.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "decoder_fused.hip" in r.stdout


def test_lint_detects_the_hazards(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lint_asm_loads as L
    isa = tmp_path / "k.s"
    isa.write_text("""
_Z1kv:                                  ; @k
	;;#ASMSTART
	global_load_dwordx4 v[4:7], v1, s[2:3]
	;;#ASMEND
	v_mov_b64_e32 v[8:9], v[4:5]
	;;#ASMSTART
	s_waitcnt vmcnt(0)
	;;#ASMEND
	v_mov_b64_e32 v[10:11], v[6:7]
	;;#ASMSTART
	global_store_dwordx4 v1, v[8:11], s[2:3]
	;;#ASMEND
	v_add_u32_e32 v8, s1, v2
	s_endpgm
""")
    kinds = sorted(x[0] for x in L.scan(str(isa)))
    assert kinds == ["inflight", "store"], kinds


def test_mfma_srcc_lint_is_clean_on_every_source():
    """tools/lint_mfma_srcc.py over ALL of csrc/ (no file-level exemption): an un-tied VGPR SrcC may only
    be rewritten behind the compiler's pad (>= 3 wait states) for compiler-emitted MFMAs, and never near an inline-asm MFMA."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lint_mfma_srcc.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "decoder_fused.hip" in r.stdout and "VIOLATIONS 0" in r.stdout


def test_mfma_srcc_lint_detects_both_tiers(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lint_mfma_srcc as L
    isa = tmp_path / "k.s"
    isa.write_text("""
_Z1kv:                                  ; @k
	v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], v[4:7]
	s_nop 2
	ds_read_b128 v[4:7], v20
	v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], v[24:27]
	s_nop 0
	ds_read_b128 v[24:27], v20
	;;#ASMSTART
	v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], v[28:31]
	;;#ASMEND
	s_nop 7
	v_mov_b32_e32 v28, 0
	;;#ASMSTART
	v_mfma_f32_16x16x32_f16 v[32:35], v[8:11], v[12:15], v[32:35]
	;;#ASMEND
	v_mov_b32_e32 v32, 0
	s_endpgm
""")
    f = L.scan(str(isa))
    assert [(x[5], x[6], x[7], x[8]) for x in f] == [("load", 3, False, False), ("load", 1, False, True), ("valu", 8, True, True)], f


def test_generated_asm_blocks_are_up_to_date():
    """crowdsam_amd/csrc/*_asm.inc are generated (tools/gen/*.py); the committed files must be what the generators print."""
    for gen, inc in (("gen_attn_window_asm.py", "attn_window_asm.inc"), ("gen_attn_flash80_asm.py", "attn_flash80_asm.inc")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", gen)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
        have = open(os.path.join(ROOT, "crowdsam_amd", "csrc", inc)).read()
        assert r.stdout == have, "%s is stale: python tools/gen/%s > crowdsam_amd/csrc/%s" % (inc, gen, inc)


def _gemm4w_program():
    env = {k: v for k, v in os.environ.items() if not k.startswith("G4_")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen", "gen_gemm4w_asm.py")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_gemm4w_main_loop_is_up_to_date_and_ordered():
    """The hand-scheduled GEMM main loop (crowdsam_amd/csrc/gemm4w_asm.inc) is what tools/gen/gen_gemm4w_asm.py emits with default
    knobs, and its synchronisation reads as designed: per 64-wide K tile 128 MFMAs, 32 fragment reads, 16 LDS-DMA pieces each
    directly preceded by its m0 write; every fragment read of an LDS stage comes after a barrier that follows a `vmcnt(0)`
    issued after the LAST LDS-DMA into that stage (an LDS-DMA piece is ordered for another wave's ds_read only by the issuing
    wave's vmcnt + a barrier); an LDS-DMA into a stage comes after a barrier that follows the last read of that stage's previous
    tile (with its lgkmcnt(0) before the barrier)."""
    import re
    out = _gemm4w_program()
    assert out == open(os.path.join(ROOT, "crowdsam_amd", "csrc", "gemm4w_asm.inc")).read(), \
        "gemm4w_asm.inc is stale: python tools/gen/gen_gemm4w_asm.py > crowdsam_amd/csrc/gemm4w_asm.inc"
    ins = re.findall(r'^\s+"([^"\\]+)\\n\\t"', out, flags=re.M)
    assert sum(i.startswith("v_mfma") for i in ins) == 4 * 128        # two tiles in the loop body + two peeled tiles
    assert sum(i.startswith("ds_read_b128") for i in ins) == 16 + 3 * 32 + 16
    assert sum("global_load_lds" in i for i in ins) == 2 * 16 + 2 * 16  # prologue (tiles 0, 1) + the loop body's two tiles
    for k, i in enumerate(ins):
        if "global_load_lds" in i:
            prev = [j for j in ins[max(0, k - 3):k] if not j.startswith("v_mfma")]
            assert prev and prev[-1].startswith(("s_add_u32 m0", "s_nop")), (k, ins[k - 3:k + 1])

    def stage_of_read(i):       # operand names ra<stage><ks> / rw<stage><ks>
        return int(re.search(r"%\[r[aw](\d)\d\]", i).group(1))

    def stage_of_dma(k):        # the m0 write before it: offset // 65536
        j = k - 1
        while not ins[j].startswith("s_add_u32 m0"):
            j -= 1
        return int(ins[j].split(",")[-1]) // 65536

    # walk the straight-line stream twice around the loop body (prologue, body, body, tail): state per stage
    a, b = ins.index("L_g4_loop_%=:"), ins.index("L_g4_tail_%=:")
    stream = ins[:a] + [x for x in ins[a + 1:b] if "s_cbranch" not in x] * 2 + ins[b + 1:]
    dma_pending = {0: False, 1: False}      # LDS-DMA issued into the stage and not yet (vmcnt(0) -> barrier)-published
    waited = {0: True, 1: True}             # ... vmcnt(0) seen since the last DMA into the stage (publication needs a barrier next)
    reads_open = {0: False, 1: False}       # fragment reads of the stage issued and not yet (lgkmcnt(0) -> barrier)-retired
    lgkm_ok = True
    for k, i in enumerate(stream):
        if "global_load_lds" in i:
            # find its stage from the nearest preceding m0 write in the stream
            j = k - 1
            while not stream[j].startswith("s_add_u32 m0"):
                j -= 1
            s = int(stream[j].split(",")[-1]) // 65536
            assert not reads_open[s], "LDS-DMA into stage %d while its previous tile's reads are not retired (instruction %d)" % (s, k)
            dma_pending[s], waited[s] = True, False
        elif i.startswith("ds_read_b128"):
            s = stage_of_read(i)
            assert not dma_pending[s], "fragment read of stage %d before its LDS-DMA was published (instruction %d)" % (s, k)
            reads_open[s] = True
            lgkm_ok = False
        elif i.startswith("s_waitcnt"):
            if "vmcnt(0)" in i:
                waited = {0: True, 1: True}
            if "lgkmcnt(0)" in i:
                lgkm_ok = True
            m = re.search(r"vmcnt\((\d+)\)", i)
            if m and int(m.group(1)) == 16:          # prologue: tile 0's sixteen pieces have landed, tile 1's may be in flight
                waited[0] = True
        elif i == "s_barrier":
            for s in (0, 1):
                if dma_pending[s] and waited[s]:
                    dma_pending[s] = False
                if reads_open[s] and lgkm_ok:
                    reads_open[s] = False


def test_lds_fragment_rings_are_not_collapsed():
    """tools/lint_lds_ring.py: in the product kernels of the decoder sweep (csam_i2t_t2i both layers, csam_upscale_stream) the
    LDS reads that feed MFMAs are issued a ring ahead of them.  Round 6 found in the ISA that the machine scheduler had sunk every
    one of them to a single MFMA before its use (121 of 162 LDS-fed MFMAs of the layer-0 kernel waited an LDS round trip); the
    scheduling barriers of FUSE_PM_PIPE / FUSE_RD_PIPE / CSAM_UP_PIN keep the source order, and this gate notices when a compiler
    update or an edit collapses a ring again."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lint_lds_ring.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") >= 3 and "OVER" not in r.stdout, r.stdout


def test_lds_ring_lint_measures_the_distance(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import lint_lds_ring as L
    isa = tmp_path / "k.s"
    isa.write_text("""
_Z1kv:                                  ; @k
	ds_read_b128 v[4:7], v20
	ds_read_b128 v[8:11], v20 offset:16
	s_waitcnt lgkmcnt(1)
	v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[12:15], v[0:3]
	ds_read_b128 v[4:7], v20 offset:32
	s_waitcnt lgkmcnt(1)
	v_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], v[0:3]
	v_mfma_f32_16x16x32_f16 v[16:19], v[24:27], v[12:15], v[16:19]
	s_waitcnt lgkmcnt(0)
	v_mfma_f32_16x16x32_f16 v[0:3], v[4:7], v[12:15], v[0:3]
	s_endpgm
""")
    res = L.scan(str(isa))
    # three LDS-fed MFMAs at distances 0, 1 and 2 (the register-fed one in between does not count); two of them tight
    assert res == {"_Z1kv": (3, 2, {0: 1, 1: 1, 2: 1})}, res
