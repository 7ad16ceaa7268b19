"""The limits of the command-line host, checked without a GPU: bin/gmrm_hip refuses K > 8, G > 64 and N > 1 048 576 before
it touches a device (gmrm_amd/host/gmrm_main.cpp)."""
import subprocess
from pathlib import Path

import pytest

from tests import cases
from tests.test_gpu_cli import _write_inputs

ROOT = Path(__file__).resolve().parent.parent
BIN = ROOT / "bin" / "gmrm_hip"


@pytest.mark.parametrize("what", ["K", "G", "N"])
def test_cli_says_its_limits_out_loud(tmp_path, what):
    """options.cpp:222-286 accepts any number of groups and mixture components and phenotype.cpp:22 any N; this build
    supports K <= 8, G <= 64, N <= 1 048 576 and must say so in a FATAL line, not fail somewhere inside."""
    if not BIN.exists():
        subprocess.run(["make", "-s", "-C", str(ROOT / "gmrm_amd" / "host")], check=True)
    case = cases.Case("lim", 64, 8, 1, 4, 1, 0.0, 0, 3, 1, 2)
    inp = cases.make_inputs(case)
    phens = _write_inputs(tmp_path, case, inp)
    if what == "K":
        (tmp_path / "t.grm").write_text(" ".join(f"{v:.5f}" for v in [0.0] + [0.0001 * 2 ** i for i in range(8)]) + "\n")   # 9 components
        expect = "at most 8"
    elif what == "G":
        (tmp_path / "t.grm").write_text("0.00000 0.00010 0.00100 0.01000\n" * 65)
        (tmp_path / "t.gri").write_text("".join(f"{i} {i % 65}\n" for i in range(case.M)))
        expect = "at most 64"
    else:
        (tmp_path / "t.dim").write_text(f"{1048577} {case.M}\n")
        expect = "at most 1048576"
    cmd = [str(BIN), "--bed-file", str(tmp_path / "t.bed"), "--dim-file", str(tmp_path / "t.dim"),
           "--phen-files", ",".join(str(p) for p in phens), "--group-index-file", str(tmp_path / "t.gri"),
           "--group-mixture-file", str(tmp_path / "t.grm"), "--iterations", "1", "--out-dir", str(tmp_path / "out")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    fatal = [ln for ln in (r.stdout + r.stderr).splitlines() if ln.startswith("FATAL  :")]
    assert fatal and expect in fatal[-1], (r.stdout[-1500:], r.stderr[-500:])
