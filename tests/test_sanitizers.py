"""CPU builds of the native code under AddressSanitizer + UndefinedBehaviorSanitizer.

`make -C oracle SAN=1` and the `PGPS_SAN=1` variants of the test harnesses build oracle/kalman_seq.c, oracle/kalman_par.c,
tests/cpu_math/emul.cpp (csrc/pgps_math.h + csrc/pgps_dual.h on the host) and csrc/pgps_seq_host.cpp with
-fsanitize=address,undefined; tests/san/driver.py runs them against the numpy oracle in a child process with libasan
preloaded.  CPU only: the GPU build is never sanitized (GPU ASan is not available on the pool)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_cpu_code_is_clean_under_asan_and_ubsan():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "SAN=1"], check=True)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], check=True, capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(asan) and os.path.exists(asan), "gcc's libasan.so not found"
    env = dict(os.environ, PGPS_SAN="1", LD_PRELOAD=asan, OMP_NUM_THREADS="3",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "san", "driver.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    report = p.stdout[-3000:] + "\n" + p.stderr[-6000:]
    assert p.returncode == 0, report
    assert "SAN OK" in p.stdout, report
    assert "AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr, report
