"""CPU: include/drs.h is a C header and libdrs_hip.so a C library -- a C99 client (tests/c/abi_client.c) compiles against the
header with -Wall -Werror, links the library, and reads the Dilated8Pooling plan through the step-level entry points."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "dynamic-rs-segmentation_amd")


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no C compiler")
def test_c99_client_compiles_links_and_reads_the_plan(tmp_path):
    from drs_amd.nets import Plan
    exe = str(tmp_path / "abi_client")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_client.c"),
                    "-L", PKG, "-ldrs_hip", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib", "-o", exe], check=True)
    out = subprocess.run([exe, "dilated_grsl_rate8"], check=True, capture_output=True, text=True).stdout
    kv = dict(t.split("=") for t in out.split())
    p = Plan("dilated_grsl_rate8", 5, 6)
    assert int(kv["layers"]) == 8 and int(kv["params"]) == p.n_params == 2091590 and int(kv["decay"]) == p.n_decay and int(kv["bn"]) == p.n_bn
    assert (int(kv["x0_ld"]), int(kv["x0_halo"])) == p.buffers["x0"] and kv["conv8"] == "1"
    assert int(kv["unbound_step_rc"]) == 1 and abs(float(kv["lr"]) - 0.01) < 1e-9
    assert int(kv["bytes"]) > 1 << 30                      # ~15 GB of workspaces listed for (128, 85); nothing allocated
    assert subprocess.run([exe, "no_such_net"], capture_output=True, text=True).returncode == 2
