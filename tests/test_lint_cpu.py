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
