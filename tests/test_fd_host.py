"""Host build of the device field / curve code (za_amd/csrc/fd.h, ecd.h) with bound tracking:
compiles tests/cpp/test_fd.cpp with g++ -DZG_FD_DEBUG and runs it.  Every lazy-reduction constant
(sub<K>, pack) is asserted against a propagated upper bound, and results are compared with the
strict 8x32 implementation that the golden vectors pin."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lazy_field_and_curve_on_host(tmp_path):
    exe = str(tmp_path / "test_fd")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-DZG_FD_DEBUG", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "test_fd.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout
