"""The sweep kernel's genotype prefetch parks asm-issued loads in AGPRs for a whole round; hipcc does not
count those loads.  tools/check_prefetch_regs.py proves on the generated code that nothing touches the
registers before the wait (make the invariant checkable)."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_prefetch_registers_are_untouched_until_the_wait():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "check_prefetch_regs.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert r.stdout.count("ok  ") == 42           # (AGPR tiles + staged LDS reads) x R in {1,2,4} x {no missing genotypes: long batches, short batches that cross stops,
                                                   #  long batches that cross stops; all markers (with / without the continuation); some markers (short and long batches)}


def _checker():
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_prefetch_regs", ROOT / "tools" / "check_prefetch_regs.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


STAGE_OK = """
	;;#ASMSTART
	ds_read_b128 v[10:13], v5
	ds_read_b128 v[14:17], v5 offset:16
	;;#ASMEND
	v_add_u32_e32 v1, v2, v3
	;;#ASMSTART
	ds_read_b128 v[20:23], v6
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt lgkmcnt(1)
	;;#ASMEND
	v_mov_b32_e32 v30, v10
	v_mov_b32_e32 v31, v17
	s_waitcnt lgkmcnt(0)
	v_mov_b32_e32 v32, v20
""".splitlines()


def test_stage_check_accepts_counted_waits_and_catches_early_uses():
    """Negative controls for the staged-LDS-read check: a use before the wait, a wait that leaves the
    register in flight, a scalar load in flight at a counted wait, a register pending at a block boundary."""
    chk = _checker()
    assert chk.check_stage("k", STAGE_OK) == []
    early = [ln.replace("v_add_u32_e32 v1, v2, v3", "v_add_u32_e32 v1, v2, v12") for ln in STAGE_OK]
    assert any("touches staged" in p for p in chk.check_stage("k", early))
    short = [ln.replace("lgkmcnt(1)", "lgkmcnt(2)") for ln in STAGE_OK]          # v[14:17] would still be in flight at its use
    assert any("touches staged" in p for p in chk.check_stage("k", short))
    smem = [ln.replace("v_add_u32_e32 v1, v2, v3", "s_load_dwordx2 s[0:1], s[4:5], 0x0") for ln in STAGE_OK]
    assert any("scalar load" in p for p in chk.check_stage("k", smem))
    branch = STAGE_OK[:8] + ["\ts_cbranch_execz .LBB0_1"] + STAGE_OK[8:]
    assert any("basic-block boundary" in p for p in chk.check_stage("k", branch))
