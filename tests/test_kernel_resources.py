"""The hand-written kernels' register / spill / scratch / code-size numbers against the committed
profiles/kernel_resources.json: a toolchain bump or an innocent edit that makes hipcc spill in the sweep
kernel costs tens of per cent of sweep time and shows in no numerics test.  Runs on the CPU: the numbers are read from
the gfx950 code objects inside the built libgmrm_hip.so (tools/kernel_resources.py)."""
import importlib.util
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _tool():
    spec = importlib.util.spec_from_file_location("kernel_resources", ROOT / "tools" / "kernel_resources.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_kernel_resources_match_the_committed_baseline():
    from gmrm_amd import build
    build.build()                                                   # (no-op when the library is current)
    kr = _tool()
    now = kr.read_library()
    base = json.loads((ROOT / "profiles" / "kernel_resources.json").read_text())
    problems = kr.compare(now, base)
    assert not problems, "\n".join(problems) + "\n(if intended: python tools/kernel_resources.py --write, and commit the json)"
    sweeps = [k for k in now if "k_sweep" in k]
    assert len(sweeps) >= 9 and all(now[k]["vgpr_count"] <= 512 for k in sweeps)
    # the kernels whose workgroups wait for each other must fit ONE workgroup per compute unit: dynamic LDS only
    assert all(now[k]["group_segment_fixed_size"] == 0 for k in sweeps)


def test_resource_guard_catches_regressions():
    """Negative controls: spills, scratch, SGPR spills +10 %, code +15 %, a vanished instantiation."""
    kr = _tool()
    k = "void gm::k_sweep<2, 0, false, true>(gm::SweepArgs)"
    base = {k: dict(vgpr_count=496, agpr_count=240, sgpr_count=106, sgpr_spill_count=300, vgpr_spill_count=0,
                    private_segment_fixed_size=0, group_segment_fixed_size=0, code_bytes=60000)}
    assert kr.compare(base, base) == []
    for field, val, word in (("vgpr_spill_count", 4, "VGPR spills"), ("private_segment_fixed_size", 64, "scratch"),
                             ("sgpr_spill_count", 400, "SGPR spills"), ("code_bytes", 70000, "code bytes")):
        worse = {k: dict(base[k], **{field: val})}
        assert any(word in p for p in kr.compare(worse, base)), field
    ok = {k: dict(base[k], sgpr_spill_count=320, code_bytes=65000)}     # inside the tolerances
    assert kr.compare(ok, base) == []
    assert any("not in the library" in p for p in kr.compare({}, base))
    assert any("new instantiation" in p for p in kr.compare(base, {}))
