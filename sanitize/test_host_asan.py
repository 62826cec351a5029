"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer over the HOST code of the library (SURVEY section 5, row "race detection /
sanitizers": run on the CPU build only -- the GPU pool refuses sanitizer runs, so this lives OUTSIDE tests/: run it with
`python -m pytest sanitize/ -q`).  `sanitize/build_host_asan.sh` compiles every source
host-only (`--cuda-host-only -fsanitize=address,undefined`, no device code objects) into libdrs_hip_asan.so in a temporary
directory outside the repository; the client
(sanitize/host_client.py) then walks the net tables, the buffer / variable layout, the filter-gradient cut and the argument validation of
every entry point that needs no device, in a child process with the sanitizer runtime preloaded."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_asan_and_ubsan(tmp_path):
    rt = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not rt:
        pytest.skip("no clang AddressSanitizer runtime in this image")
    out = str(tmp_path)              # outside the repository: the GPU pool refuses a tree that holds a sanitizer build
    subprocess.run(["bash", os.path.join(ROOT, "sanitize", "build_host_asan.sh"), out], check=True, capture_output=True)
    env = dict(os.environ, DRS_ASAN_LIB=os.path.join(out, "libdrs_hip_asan.so"), LD_PRELOAD=rt[-1], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "sanitize", "host_client.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "sanitized host run ok" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]
