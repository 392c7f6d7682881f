"""The lean correctly-rounded sqrt / reciprocal sequences of path-tracing_amd/csrc/pt_fastfp.hpp, checked on the device
against the correctly rounded result for EVERY float of the range the kernel uses them for (about 10^9 values each).
The integrator's bit-exact parity with the oracle (IEEE sqrtf and division on the CPU) rests on this."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_fast_sqrt_and_reciprocal_are_correctly_rounded_everywhere(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "verify_fast_fp")
    b = subprocess.run([hipcc, "-O2", "-ffp-contract=off", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "path-tracing_amd", "csrc"),
                        os.path.join(ROOT, "tools", "fp", "verify_fast_fp.hip"), "-o", exe], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout, r.stderr[-1000:])
    assert "sqrt mismatches 0 " in r.stdout and "rcp mismatches 0 " in r.stdout and "checked 1015021568 floats" in r.stdout
