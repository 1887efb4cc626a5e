"""The C-ABI library loads on a CPU-only box and exports every symbol include/icsg3d.h declares
(no compute calls here: there is no GPU).  Also checks the ctypes table covers the header."""
import ctypes
import os
import re

import pytest

from icsg3d_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "icsg3d.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ics_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = header_functions()
    assert len(names) >= 35
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), "libicsg3d_hip.so does not export %s" % n
        assert n in _lib.SIGNATURES, "icsg3d_amd/_lib.py has no prototype for %s" % n
    assert sorted(_lib.SIGNATURES) == names, "binding declares symbols missing from the header"


def test_library_reports_version_and_errors_without_gpu():
    lib = _lib.load()
    assert b"gfx950" in lib.ics_version()
    # a failing call must return non-zero and leave a message; never crash, never fall back to CPU
    h = ctypes.c_void_p()
    rc = lib.ics_unet_create(None, ctypes.byref(h))
    assert rc != 0 and lib.ics_last_error()


def test_product_package_never_imports_the_oracle():
    """... nor do the entry scripts and the helpers under scripts/: the checkers that use the oracle live under tests/tools/.
    Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg touch oracle/."""
    trees = [os.path.join(ROOT, "icsg3d_amd"), os.path.join(ROOT, "scripts")]
    files = [os.path.join(ROOT, f) for f in ("train_unet.py", "train_vae.py", "generate.py")]
    for tree in trees:
        for dirpath, _, names in os.walk(tree):
            files += [os.path.join(dirpath, f) for f in names if f.endswith(".py")]
    assert len(files) > 30
    for f in files:
        text = open(f).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_host_code_is_clean_under_asan_ubsan():
    """`make asan`: the engine's HOST code (tensor tables, workspace / split-K / bucket planners, the whole C ABI)
    compiled host-only with AddressSanitizer + UBSan + LeakSanitizer against a malloc-backed stand-in for the HIP
    runtime (csrc/hoststub/), driven through every entry point.  Sanitizers run on the CPU build only."""
    import shutil
    import subprocess
    csrc = os.path.join(ROOT, "icsg3d_amd", "csrc")
    if shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no toolchain")
    p = subprocess.run(["make", "-C", csrc, "asan"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-3000:], p.stderr[-3000:])
    assert "no sanitizer report" in p.stdout
    assert "ERROR: AddressSanitizer" not in p.stderr and "runtime error" not in p.stderr
