"""The build gate of the cooperative kernel families (tools/scratch_gate.py, csrc/Makefile): a kernel that passes operands
between lanes in registers must not need scratch memory."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GATE = os.path.join(ROOT, "tools", "scratch_gate.py")

SAMPLE = """./pgps_rc2.hip.h:248:1: remark: Function Name: _ZN4pgps3rc211rc2_reduce1IdLi32ELb1EEEvNS_2wc6WcArgsIT_EE [-Rpass-analysis=kernel-resource-usage]
./pgps_rc2.hip.h:248:1: remark:     SGPRs: 102 [-Rpass-analysis=kernel-resource-usage]
./pgps_rc2.hip.h:248:1: remark:     VGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
./pgps_rc2.hip.h:248:1: remark:     AGPRs: 256 [-Rpass-analysis=kernel-resource-usage]
./pgps_rc2.hip.h:248:1: remark:     ScratchSize [bytes/lane]: %d [-Rpass-analysis=kernel-resource-usage]
./pgps_rc2.hip.h:248:1: remark:     Occupancy [waves/SIMD]: 1 [-Rpass-analysis=kernel-resource-usage]
./pgps_core.hip:10:1: remark: Function Name: _ZN4pgps7k_widenElPKfPd [-Rpass-analysis=kernel-resource-usage]
./pgps_core.hip:10:1: remark:     VGPRs: 8 [-Rpass-analysis=kernel-resource-usage]
./pgps_core.hip:10:1: remark:     AGPRs: 0 [-Rpass-analysis=kernel-resource-usage]
./pgps_core.hip:10:1: remark:     ScratchSize [bytes/lane]: 64 [-Rpass-analysis=kernel-resource-usage]
./pgps_core.hip:10:1: remark:     Occupancy [waves/SIMD]: 8 [-Rpass-analysis=kernel-resource-usage]
"""


def _run(paths):
    return subprocess.run([sys.executable, GATE] + paths, capture_output=True, text=True)


def test_gate_refuses_scratch_in_a_cooperative_kernel(tmp_path):
    bad, good = tmp_path / "bad.res", tmp_path / "good.res"
    bad.write_text(SAMPLE % 444)
    good.write_text(SAMPLE % 0)          # (scratch in a kernel outside pgps::rc / rc2 / qc is none of the gate's business)
    r = _run([str(bad)])
    assert r.returncode == 1 and "444 B/lane" in r.stderr and "rc2_reduce1" in r.stderr
    r = _run([str(good)])
    assert r.returncode == 0 and "0 with unexplained scratch" in r.stdout


def test_the_built_library_passed_the_gate():
    """build() has compiled the cooperative units with their resource remarks kept: every one of their kernels is at
    ScratchSize 0 (no allow-list entries are needed at the moment)."""
    res = sorted(glob.glob(os.path.join(ROOT, "parallel-gps_amd", "csrc", "build", "*.res")))
    if not res:
        import pytest
        pytest.skip("no build/*.res (the library was not built in this tree)")
    r = _run(res)
    assert r.returncode == 0, r.stderr
    assert "allow-listed" in r.stdout
