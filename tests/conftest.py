"""Shared pytest setup: import paths, the ``gpu`` marker, and the GPU probe."""

import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT / "fast-forward-indexes_amd", ROOT / "oracle", ROOT / "tests", ROOT):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# if anything below libffx ends the process with abort(), the log should show the NATIVE call stack of the aborting thread, not
# only Python's (libffx installs the handler when it is loaded).  That -- with pytest.ini's Python-level capture, which lets the
# runtime's own last words through -- is how the rare abort of rounds 3 - 4 was traced to the HIP runtime's on-the-fly
# page-locking of heap memory (HISTORY.md section 10); libffx hands the runtime no ordinary memory since (csrc/host_xfer.h), and the handler stays armed
os.environ.setdefault("FFX_ABORT_BACKTRACE", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # if the runtime ever aborts the process, the log should say which test and which Python line the main thread was in
    import faulthandler

    if not faulthandler.is_enabled():
        faulthandler.enable(all_threads=True)
    # a fresh checkout has no built libraries (they are git-ignored): build them once, as
    # __graft_entry__.build() does, so the suite does not depend on a step having been run before it
    lib = ROOT / "fast-forward-indexes_amd" / "fast_forward_amd" / "libffx.so"
    if not lib.exists():
        import shutil
        import subprocess

        if shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists():
            subprocess.run(["make", "-C", str(ROOT / "fast-forward-indexes_amd" / "csrc")], check=False, capture_output=True)


def _have_gpu() -> bool:
    try:
        from fast_forward_amd import native

        return native.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def have_gpu():
    return _have_gpu()


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly rather than skip: nothing to do here.
    # Plain `pytest` (no -m) on the CPU container skips the gpu tests.
    if config.getoption("-m"):
        return
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU here; run with -m gpu on the GPU box")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _release_device_objects_between_tests(request):
    """Indexes, plans and communicators a GPU test did not close are released right after it (a collection pass), not
    whenever the interpreter gets to it in the middle of a later test: the device-side resources of ~200 tests (streams,
    events, page-locked blocks) then never pile up, and what a test sees does not depend on the garbage of its
    predecessors.  (Not what the rare SIGABRT of rounds 3 - 4 was about -- that was the runtime page-locking heap memory on the
    fly, HISTORY.md section 10 -- but a suite whose tests do not see each other's leftovers is worth having anyway.)"""
    yield
    if "gpu" in request.keywords and not os.environ.get("FFX_TEST_KEEP_GARBAGE"):  # (tools/abort_hunt.sh runs the suite without this)
        import gc

        gc.collect()
