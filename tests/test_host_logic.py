"""Host-side logic that needs no GPU: the C-ABI library loads and exports every symbol of
include/ffx.h, id tables, CSR grouping, argument checks that must fire before device work."""

import os
import re
from pathlib import Path

import numpy as np
import pytest

from fast_forward_amd import Ranking, native
from fast_forward_amd.index import InMemoryIndex, Mode
from fast_forward_amd.index.api import _grouped_by_query
from fast_forward_amd.index.ids import AVEP, FIRSTP, MAXP, PASSAGE, IdTable

ROOT = Path(__file__).resolve().parent.parent


def test_library_exports_every_declared_symbol():
    """Both headers against both lists of the binding: include/ffx.h (the stable contract INTEGRATION.md binds) and
    include/ffx_private.h (what the Python layer uses on top of it); nothing is declared twice, everything is exported."""
    found = {}
    for header, names in (("ffx.h", native.SYMBOLS), ("ffx_private.h", native.PRIVATE_SYMBOLS)):
        text = (ROOT / "include" / header).read_text()
        declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(ffx_\w+)\s*\(", text, flags=re.M))
        assert declared == set(names), (header, declared ^ set(names))
        found[header] = declared
    assert not (found["ffx.h"] & found["ffx_private.h"])
    assert len(found["ffx.h"]) <= 86  # the contract stays small: helpers go to the private header
    lib = native.lib()
    for name in found["ffx.h"] | found["ffx_private.h"]:
        assert hasattr(lib, name), name
    assert lib.ffx_abi_version() == 3
    # what the C examples use must come from the stable header alone
    for example in ("rerank.c", "rerank_sharded.c"):
        used = set(re.findall(r"\b(ffx_\w+)\s*\(", (ROOT / "tests" / "c_abi" / example).read_text()))
        assert not (used & found["ffx_private.h"]), (example, used & found["ffx_private.h"])


def test_no_cpu_fallback(have_gpu):
    if have_gpu:
        pytest.skip("a GPU is present")
    assert native.device_count() == 0
    with pytest.raises(native.FFXError) as e:
        native.DeviceIndex(8, 4)
    assert e.value.code == native.NO_DEVICE
    with pytest.raises(RuntimeError):
        native.require_device()
    idx = InMemoryIndex()
    with pytest.raises(native.FFXError):
        idx.add(np.zeros((2, 4), dtype=np.float32), doc_ids=["a", "b"])


def test_product_package_never_imports_the_oracle():
    pkg = ROOT / "fast-forward-indexes_amd"
    for f in pkg.rglob("*"):
        if f.suffix in (".py", ".hip", ".h"):
            text = f.read_text()
            assert "libffx_oracle" not in text, f  # never loads the oracle library
            assert not re.search(r"^\s*(import|from)\s+\S*oracle", text, flags=re.M), f  # never imports it
            assert not re.search(r"#include\s+\"[^\"]*oracle", text), f


def _table():
    t = IdTable()
    # rows:      0     1     2     3     4
    t.add(["d0", "d0", "d1", None, "d2"], ["p0", "p1", None, "p3", "p4"], 0)
    t.add(["d1", "d0"], [None, "p6"], 5)  # d1 and d0 re-opened: rows not contiguous
    return t


def test_id_table_csr_and_resolution():
    t = _table()
    assert t.doc_ids == {"d0", "d1", "d2"} and t.psg_ids == {"p0", "p1", "p3", "p4", "p6"}
    off, rows = t.doc_csr()
    assert off.tolist() == [0, 3, 5, 6] and rows.tolist() == [0, 1, 6, 2, 5, 4]
    assert t.psg_rows().tolist() == [0, 1, 3, 4, 6]
    assert t.resolve(np.array(["d2", "d0", "d2", "d1"]), MAXP).tolist() == [2, 0, 2, 1]
    assert t.resolve(["p6", "p0"], PASSAGE).tolist() == [4, 0]
    with pytest.raises(IndexError, match="ID dx not found in the index."):
        t.resolve(["d0", "dx", "dy"], FIRSTP)
    with pytest.raises(IndexError):
        t.resolve(["d0"], PASSAGE)  # a doc id is not a passage id
    docs, psgs = t.row_ids(7)
    assert docs == ["d0", "d0", "d1", None, "d2", "d1", "d0"]
    assert psgs == ["p0", "p1", None, "p3", "p4", None, "p6"]


def test_id_table_rows_for_follows_get_indices():
    t = _table()
    assert t.rows_for(["d0", "d1"], MAXP) == ([0, 1, 6, 2, 5], ["d0", "d0", "d0", "d1", "d1"])
    assert t.rows_for(["d0", "d1"], AVEP)[0] == [0, 1, 6, 2, 5]
    assert t.rows_for(["d1", "d0"], FIRSTP) == ([2, 0], ["d1", "d0"])
    assert t.rows_for(["p3"], PASSAGE) == ([3], ["p3"])
    with pytest.raises(IndexError):
        t.rows_for(["p3"], MAXP)


def test_duplicate_passage_id_leaves_table_untouched():
    t = _table()
    v = t.version
    with pytest.raises(RuntimeError, match="Passage ID p1 already exists."):
        t.add(["d9", "d9"], ["p9", "p1"], 7)
    with pytest.raises(RuntimeError):
        t.add([None, None], ["p8", "p8"], 7)
    assert t.version == v and "d9" not in t.doc_ids and "p9" not in t.psg_ids


def test_grouping_pairs_by_query():
    perm, off = _grouped_by_query(np.array([0, 0, 1, 1, 1, 3]), 4)
    assert perm is None and off.tolist() == [0, 2, 5, 5, 6]
    perm, off = _grouped_by_query(np.array([2, 0, 2, 1, 0]), 3)
    assert off.tolist() == [0, 2, 3, 5] and perm.tolist() == [1, 4, 3, 0, 2]


def test_argument_checks_fire_before_device_work():
    idx = InMemoryIndex()
    no_queries = Ranking.from_run({"q1": {"d0": 1.0}})
    with pytest.raises(ValueError, match="no queries"):
        idx(no_queries)
    with_q = Ranking.from_run({"q1": {"d0": 1.0}}, queries={"q1": "x"})
    with pytest.raises(ValueError, match="Early stopping"):
        idx(with_q, early_stopping=10, early_stopping_alpha=None, early_stopping_depths=(10,))
    with pytest.raises(ValueError, match="Early stopping"):
        idx(with_q, early_stopping=10, early_stopping_alpha=0.5)
    with pytest.raises(RuntimeError, match="query encoder"):
        idx.encode_queries(["x"])
    with pytest.raises(ValueError):
        idx.add(np.zeros((2, 3)), doc_ids=None, psg_ids=None)
    with pytest.raises(ValueError):
        idx.add(np.zeros((3, 3)), doc_ids=["a"], psg_ids=None)
    with pytest.raises(ValueError):
        idx.add(np.zeros((2, 3)), doc_ids=[None, "a"], psg_ids=[None, "p"])
    assert len(idx) == 0 and idx.dim is None and idx.mode == Mode.MAXP
    idx.mode = Mode.FIRSTP
    assert idx.mode == Mode.FIRSTP


def test_header_is_plain_c(tmp_path):
    """include/ffx.h must be consumable from C (the drop-in boundary is a C ABI, not C++)."""
    import subprocess

    src = tmp_path / "use_ffx.c"
    src.write_text('#include "ffx.h"\nint main(void) { ffx_stats s; (void)s; return ffx_abi_version() == FFX_ABI_VERSION ? 0 : 1; }\n')
    lib_dir = ROOT / "fast-forward-indexes_amd" / "fast_forward_amd"
    exe = tmp_path / "use_ffx"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", f"-I{ROOT / 'include'}", str(src), "-o", str(exe),
           f"-L{lib_dir}", "-l:libffx.so", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert subprocess.run([str(exe)]).returncode == 0


def test_loading_the_library_changes_no_setting_of_the_runtime():
    """Rounds 3 - 4 set GPU_PINNED_MIN_XFER_SIZE from a library constructor and from the binding.  A library must not
    reconfigure the runtime of its host process: loading libffx -- through the binding or through dlopen alone -- leaves the
    variable as the process had it (unset stays unset).  Checked in child processes."""
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    lib = root / "fast-forward-indexes_amd" / "fast_forward_amd" / "libffx.so"
    via_binding = ("import os, sys; sys.path.insert(0, %r); from fast_forward_amd import native; native.lib(); "
                   "print(os.environ.get('GPU_PINNED_MIN_XFER_SIZE'))" % str(root / "fast-forward-indexes_amd"))
    via_dlopen = ("import ctypes; ctypes.CDLL(%r); libc = ctypes.CDLL('libc.so.6'); libc.getenv.restype = ctypes.c_char_p; "
                  "print((libc.getenv(b'GPU_PINNED_MIN_XFER_SIZE') or b'None').decode())" % str(lib))
    base = {k: v for k, v in os.environ.items() if k != "GPU_PINNED_MIN_XFER_SIZE"}
    for code in (via_binding, via_dlopen):
        for extra, want in (({}, "None"), ({"GPU_PINNED_MIN_XFER_SIZE": "7"}, "7")):
            out = subprocess.run([sys.executable, "-c", code], env=dict(base, **extra), capture_output=True, text=True, timeout=120)
            assert out.returncode == 0, out.stderr[-2000:]
            assert out.stdout.strip().splitlines()[-1] == want, (code[:40], extra, out.stdout)


def test_the_default_library_ships_no_tuning_hooks():
    """`make` builds what the dispatcher can reach; the tuning hooks of FFX_DEBUG_VARIANT (ffx_experiment) and the kernels only
    they select (gather=N: strided, persistent, two-wave forms) are compiled with `make EXPERIMENTS=1` into a library of their
    own.  The default library says so, and the sources keep every such hook behind ffx_experiment."""
    from fast_forward_amd import native

    if not os.environ.get("FFX_LIBRARY"):
        assert native.has_experiments() is False
    csrc = ROOT / "fast-forward-indexes_amd" / "csrc"
    text = (csrc / "ffx.hip").read_text() + (csrc / "comm.hip").read_text()
    for hook in ("gather", "anyd_u", "generic_staged", "generic_plain", "sort_block_from", "score_plans", "score_no_ramp", "merge_on_results"):
        assert f'ffx_debug_variant("{hook}"' not in text and f'ffx_experiment("{hook}"' in text, hook
    header = (csrc / "gather_dot.hip.h").read_text()
    assert header.index("#ifdef FFX_EXPERIMENTS") < header.index("gather_dot_exact_lds_persistent_kernel(")


def test_caller_memory_reaches_the_runtime_only_through_host_xfer():
    """What replaces the setting: every copy between host memory and the device in ffx.hip / comm.hip goes through
    ffx_h2d / ffx_d2h (csrc/host_xfer.h: ordinary memory above 1 MiB is staged through page-locked blocks of the library), or
    names a page-locked block of the library itself.  A raw hipMemcpy of either direction anywhere else fails this test --
    the lines below are the known page-locked ones, by what they copy."""
    import re

    csrc = ROOT / "fast-forward-indexes_amd" / "csrc"
    page_locked = ("(dev, host, bytes,", "(host + kSmall,", "(h_small,", "(h_local_off,", "(h_got,", "(8 bytes:", ", idx->stage[s],", ", idx->pull_list_host,",
                   ", s_rows,", "(s_out,", "(b_ff,", "(b_in,", "(b_or,", "(pl->out_stage[lane]", "hipMemcpyDeviceToHost, os)")
    raw = re.compile(r"hipMemcpy(Async)?\(.*hipMemcpy(HostToDevice|DeviceToHost)")
    loose = []
    for name in ("ffx.hip", "comm.hip", "host_io.cpp", "run_file.cpp"):
        for n, line in enumerate((csrc / name).read_text().splitlines(), 1):
            if raw.search(line) and not line.lstrip().startswith("//") and not any(word in line for word in page_locked):
                loose.append(f"{name}:{n}: {line.strip()}")
    assert not loose, "\n".join(loose)
    text = (csrc / "ffx.hip").read_text()
    assert "setenv(" not in text and "setenv(" not in (csrc / "comm.hip").read_text()


def test_helpers_around_page_locked_memory_degrade_quietly_without_a_device():
    """Without a device (this container) nothing can be page-locked: a large ranking's arrays stay ordinary memory
    (``page_locked_copy`` hands the array itself back), and the helper thread of a first call (``reserve_pooled``) runs what
    it is given first, drops whatever that raises -- the caller's own thread does the work again and reports -- and adds no
    block to the pool."""
    from fast_forward_amd import native

    if native.device_count() > 0:
        pytest.skip("a device is present: covered by the GPU tests")
    big = np.arange(native.INPUT_BLOCKS_FROM // 4 + 5, dtype=np.int32)
    assert native.page_locked_copy(big) is big
    assert native.page_locked_or_empty(big.shape[0], np.float32).base is None
    two_d = np.zeros((4, 4), dtype=np.float32)
    assert native.page_locked_copy(two_d) is two_d
    ran = []

    def first():
        ran.append(1)
        raise RuntimeError("the helper thread must not take the process down")

    idle = native._POOL_IDLE[0]
    t = native.reserve_pooled([(1 << 20, np.float32)], first=first)
    t.join(30)
    assert not t.is_alive() and ran == [1] and native._POOL_IDLE[0] == idle
    assert native.drain_result_pool() == 0
