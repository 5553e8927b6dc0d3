"""The host reorderers (gcn_amd/csrc/reorder.cpp) built with AddressSanitizer + UndefinedBehaviorSanitizer and run
on the CPU over graphs that stress their index arithmetic (GPU sanitizers are not available on the pool; the
device code is covered by the parity tests)."""
import os
import subprocess

from util import ROOT


def test_host_reorderers_are_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "sanitize_reorder")
    build = subprocess.run(["g++", "-std=c++20", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                            "-D_GLIBCXX_ASSERTIONS", os.path.join(ROOT, "tests", "sanitize_reorder.cpp"),
                            os.path.join(ROOT, "gcn_amd", "csrc", "reorder.cpp"), "-o", exe],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert run.stdout.count(" ok (") == 6
