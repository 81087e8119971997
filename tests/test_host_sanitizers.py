"""Host code of the product library and the kernels' lane code under AddressSanitizer + UBSan (CPU only):
tests/asan/host_asan.cpp includes the sources where they lie and drives them as the C ABI does."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_and_lane_code_are_clean_under_asan_ubsan(tmp_path):
    exe = tmp_path / "host_asan"
    src = os.path.join(ROOT, "tests", "asan", "host_asan.cpp")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-o", str(exe), src])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    out = subprocess.run([str(exe), str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.startswith("ok:")
