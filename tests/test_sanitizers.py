"""`make -C oracle asan`: the host layer (level / list construction, MGBuild, cycle driver, BiCGStab, CLI) and the CPU
restatement under AddressSanitizer + UndefinedBehaviorSanitizer, CPU only (GPU ASan is not available on the pool)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_layer_is_clean_under_asan_and_ubsan():
    out = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "asan/ubsan: clean" in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
