"""The checker under AddressSanitizer + UndefinedBehaviorSanitizer.  Every parity claim rests on
oracle/; this runs its known-answer and golden-vector tests against the sanitizer build (`make -C oracle asan`), in a
child process with the sanitizer runtimes preloaded (python itself is not instrumented).  `make -C oracle asan-test`
runs the WHOLE CPU suite the same way."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    rt = [subprocess.check_output(["g++", "-print-file-name=" + n], text=True).strip() for n in ("libasan.so", "libubsan.so")]
    assert all(os.path.isabs(p) and os.path.exists(p) for p in rt), rt
    env = dict(os.environ, LD_PRELOAD=" ".join(rt), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               VS_ORACLE_LIB=os.path.join(ROOT, "oracle", "libvs_oracle_asan.so"))
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "-m", "not gpu",
                          os.path.join(ROOT, "tests", "test_oracle_known_answers.py"), os.path.join(ROOT, "tests", "test_golden.py"),
                          os.path.join(ROOT, "tests", "test_phase_cpu.py")],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert "passed" in out.stdout and "AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
