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
    assert r.stdout.count("ok  ") == 15           # R in {1,2,4} x {no, all (each with / without the continuation), some} markers with missing genotypes
