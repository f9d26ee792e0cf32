"""The C restatement of the IDT path (oracle/idt_oracle.c) under AddressSanitizer + UndefinedBehaviorSanitizer (SURVEY section 5:
the reference has no sanitizer story; here the CPU build of the checker gets one -- GPU ASan is not available).  The sanitized
library is built by `make -C oracle asan`, loaded into a fresh interpreter with libasan preloaded, and the golden-vector tests of
tests/test_oracle_idt_golden.py run through it: any out-of-bounds access, overflow or misaligned access aborts that process."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_idt_oracle_goldens_under_asan_ubsan():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lib = os.path.join(ROOT, "oracle", "_build", "liboracle_asan.so")
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("no libasan.so next to gcc")
    env = dict(os.environ, LD_PRELOAD=asan, CT_ORACLE_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle_idt_golden.py"), "-x", "-q", "-m", "not gpu",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
    # the sanitized library really was the one loaded
    probe = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from oracle import iterative as it; it._load() if hasattr(it, '_load') else None; "
                            "print(it._LIB_PATH)" % ROOT], capture_output=True, text=True, env=env, cwd=ROOT)
    assert probe.stdout.strip().endswith("liboracle_asan.so"), probe.stdout + probe.stderr
