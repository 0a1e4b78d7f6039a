"""the host-side scheduling code of the native library under AddressSanitizer + UBSan -- on the CPU build
only (GPU sanitizers are not available on the pool).  kiez_amd/csrc/kz_plan.h has no HIP dependency; tests/host/plan_sanitize.cpp
plans ~3000 random and all BASELINE shapes (narrow and wide workgroups) and checks coverage / layout invariants;
kiez_amd/csrc/kz_floor.h (the model behind the seeded candidate lists) is fitted to synthetic probes by tests/host/floor_sanitize.cpp."""
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("g++") is None, reason="no g++")
@pytest.mark.parametrize("name", ["plan_sanitize", "floor_sanitize"])
def test_host_code_is_clean_under_asan_and_ubsan(tmp_path, name):
    exe = tmp_path / name
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-Wall", "-Wextra",
                            "-Werror", str(ROOT / "tests" / "host" / (name + ".cpp")), "-o", str(exe)], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-3000:]
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600,
                         env={"ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"})
    assert run.returncode == 0, run.stdout[-3000:] + run.stderr[-3000:]
    assert "0 failures" in run.stdout, run.stdout[-2000:]
