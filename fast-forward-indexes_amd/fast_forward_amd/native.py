"""ctypes binding of ``libffx.so`` (C ABI in ``include/ffx.h``).

There is no CPU fallback anywhere in this package: if the shared library is missing, or no
HIP device is visible when device work is requested, the error is raised to the caller.
"""

from __future__ import annotations

import ctypes as C
import weakref
from pathlib import Path

import numpy as np

import os as _os  # noqa: E402

# FFX_LIBRARY: another build of the same sources (tools/kernel_ab.py & co. name libffx_experiments.so, `make EXPERIMENTS=1`,
# which carries the tuning hooks and the kernels only they reach; the default library has neither)
LIB_PATH = Path(_os.environ["FFX_LIBRARY"]).resolve() if _os.environ.get("FFX_LIBRARY") else Path(__file__).resolve().parent / "libffx.so"

OK, BAD_ARG, OOM, HIP_ERROR, RCCL_ERROR, NOT_FOUND, NO_DEVICE = 0, -1, -2, -3, -4, -5, -6
FLAG_EXACT_SUM, FLAG_SORT, FLAG_SORT_FF, FLAG_ES_BOUND_F32, FLAG_KEEP_DENSE = 1, 2, 4, 8, 16
TOKEN_INT, TOKEN_NUMBER, TOKEN_NA, TOKEN_TEXT = 1, 2, 4, 8
RUN_SHORT_LINE, RUN_LONG_LINE, RUN_QUOTES = 1, 2, 4
BUF_FF, BUF_INTERP, BUF_ORDER, BUF_QUERIES, BUF_PAIR_IDS, BUF_SPARSE = 0, 1, 2, 3, 4, 5

# every symbol include/ffx.h declares: the stable contract (tests check the library exports all of them) ...
SYMBOLS = (
    "ffx_abi_version", "ffx_last_error", "ffx_device_count", "ffx_device_info", "ffx_index_create", "ffx_index_create_typed", "ffx_index_dtype",
    "ffx_index_set_codebook", "ffx_index_query_dim", "ffx_index_destroy", "ffx_index_dim", "ffx_index_capacity", "ffx_index_device_ptr", "ffx_index_reseat",
    "ffx_index_upload", "ffx_index_upload_rows", "ffx_index_copy_rows", "ffx_index_gather", "ffx_index_set_doc_rows", "ffx_index_set_psg_rows",
    "ffx_plan_create", "ffx_plan_destroy", "ffx_plan_set_queries", "ffx_plan_set_queries_device", "ffx_plan_set_pairs", "ffx_plan_run",
    "ffx_plan_set_pairs_lazy", "ffx_plan_early_stopping", "ffx_plan_fetch_scored", "ffx_plan_sync", "ffx_plan_fetch", "ffx_plan_stats",
    "ffx_plan_device_ptr", "ffx_plan_stream", "ffx_plan_set_stream", "ffx_comm_unique_id", "ffx_comm_init", "ffx_comm_destroy", "ffx_comm_rank",
    "ffx_comm_stream", "ffx_comm_barrier", "ffx_comm_allreduce_f64", "ffx_comm_allgather", "ffx_comm_exchange", "ffx_comm_exchange_host",
    "ffx_query_partition", "ffx_route", "ffx_sharded_create", "ffx_sharded_destroy", "ffx_sharded_set_queries", "ffx_sharded_set_pairs",
    "ffx_sharded_set_pairs_held", "ffx_sharded_stage_owned", "ffx_sharded_set_code_table", "ffx_sharded_stage_queries", "ffx_sharded_run", "ffx_sharded_fetch_async",
    "ffx_sharded_score", "ffx_sharded_sync", "ffx_sharded_fetch", "ffx_sharded_owned", "ffx_sharded_stats", "ffx_sharded_device_ptr",
    "ffx_aggregate", "ffx_ids_create", "ffx_ids_destroy", "ffx_ids_count", "ffx_ids_intern", "ffx_ids_lookup", "ffx_ids_get", "ffx_run_open",
    "ffx_run_close", "ffx_run_counts", "ffx_run_query_id", "ffx_run_tag", "ffx_run_export", "ffx_run_write", "ffx_score", "ffx_score_ranked", "ffx_index_kept", "ffx_index_drop_kept", "ffx_rank_kept",
    "ffx_host_alloc", "ffx_host_free", "ffx_host_pin", "ffx_host_unpin",
)
# ... and include/ffx_private.h: what this Python layer uses on top of it, no stability promise
PRIVATE_SYMBOLS = (
    "ffx_index_upload_rows_async", "ffx_index_pull_rows_async", "ffx_index_upload_wait", "ffx_index_upload_after", "ffx_index_fill_synthetic",
    "ffx_index_prepare_score", "ffx_host_fill_runs", "ffx_host_xfer_stats", "ffx_host_xfer_release", "ffx_build_has_experiments", "ffx_index_probe_placement",
    "ffx_plan_copy_pairs", "ffx_plan_set_pairs_device", "ffx_plan_recent_gather_ms", "ffx_merge_create", "ffx_merge_destroy", "ffx_merge_set",
    "ffx_merge_set_device", "ffx_merge_run", "ffx_merge_sync", "ffx_merge_fetch", "ffx_merge_device_ptr", "ffx_comm_rehearsal_inbox",
    "ffx_ids_lookup_packed", "ffx_ids_intern_packed", "ffx_ids_blob", "ffx_strings_code", "ffx_strings_code_offsets", "ffx_strings_code_offsets32", "ffx_strings_code_fixed", "ffx_host_repeated_in_groups", "ffx_buffer_free",
    "ffx_host_owner_crc32", "ffx_host_code_u64", "ffx_run_flags", "ffx_run_ids_blob", "ffx_run_names", "ffx_host_stage_report", "ffx_run_save", "ffx_run_save_coded", "ffx_host_order",
    "ffx_host_order_blocks", "ffx_host_rank_blocks", "ffx_host_take32", "ffx_host_interpolate", "ffx_host_unique_rows", "ffx_host_rows_unique", "ffx_host_segment_offsets",
    "ffx_host_segment_rows_unique", "ffx_host_outer_join",
)
COMM_RCCL, COMM_HOST, COMM_REHEARSAL, COMM_ID_BYTES = 0, 1, 2, 128
OP_SUM, OP_MAX, OP_MIN = 0, 1, 2


class Stats(C.Structure):
    _fields_ = [
        ("n_queries", C.c_int64), ("n_pairs", C.c_int64), ("n_rows", C.c_int64),
        ("algorithmic_bytes", C.c_double),
        ("ms_expand", C.c_float), ("ms_gather_dot", C.c_float), ("ms_reduce", C.c_float),
        ("ms_sort", C.c_float), ("ms_total", C.c_float),
        ("kernel_variant", C.c_int32), ("reserved", C.c_int32),
    ]

    def as_dict(self) -> dict:
        return {name: getattr(self, name) for name, _ in self._fields_ if name != "reserved"}


class ReseatReport(C.Structure):
    """``ffx_reseat_report`` (include/ffx.h)."""
    _fields_ = [("candidates", C.c_int32), ("chosen", C.c_int32), ("probe_ms", C.c_float * 8), ("bytes", C.c_int64), ("settle_s", C.c_float),
                ("probe_ms_settled", C.c_float)]


class ShardStats(C.Structure):
    _fields_ = [
        ("local", Stats), ("own_queries", C.c_int64), ("own_pairs", C.c_int64), ("send_bytes", C.c_int64), ("recv_bytes", C.c_int64),
        ("rank", C.c_int32), ("world", C.c_int32), ("transport", C.c_int32), ("reserved", C.c_int32),
        ("ms_exchange", C.c_float), ("ms_merge", C.c_float),
        ("stage_h2d_bytes", C.c_int64), ("stage_route_send_bytes", C.c_int64), ("stage_route_recv_bytes", C.c_int64),
        ("ms_stage_host", C.c_float), ("ms_stage_device", C.c_float),
    ]

    def as_dict(self) -> dict:
        d = {name: getattr(self, name) for name, _ in self._fields_ if name not in ("reserved", "local")}
        d.update(self.local.as_dict())
        return d


class FFXError(RuntimeError):
    """A libffx call failed (status code + the library's message)."""

    def __init__(self, code: int, message: str):
        super().__init__(f"libffx error {code}: {message}")
        self.code = code


class NotFound(FFXError):
    """An ordinal named no row (the reference's IndexError, index/util.py:38-39)."""

    def __init__(self, code: int, message: str, pair: int):
        super().__init__(code, message)
        self.pair = pair


_lib = None
TORCH_LOADED_FIRST = False


def lib() -> C.CDLL:
    """Load libffx.so (once).  Raises if it has not been built: there is nothing to fall back to.

    One process can hold only one HIP/HSA runtime.  PyTorch-ROCm wheels bundle their own
    ``libamdhip64.so.7`` (same soname as /opt/rocm's), so whichever of torch and libffx is loaded
    first decides which runtime both use.  Nothing in this package imports torch (the multi-GPU
    path has its own communicator, csrc/comm.hip); a host program that does (a transformer query
    encoder) may load either first -- ``TORCH_LOADED_FIRST`` records which it was.  Loading the
    library changes no setting of the HIP runtime: transfers from or into ordinary memory are
    staged by the library itself (csrc/host_xfer.h)."""
    global _lib, TORCH_LOADED_FIRST
    if _lib is not None:
        return _lib
    import sys

    TORCH_LOADED_FIRST = "torch" in sys.modules
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C fast-forward-indexes_amd/csrc`).  This package has no CPU fallback."
        )
    import os

    # read by the HSA runtime when it starts (the first HIP call): this host's driver supports dmabuf IPC only,
    # and RCCL's intra-node transport needs IPC (harmless for a single-GPU process)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    L = C.CDLL(str(LIB_PATH))
    vp, i32, i64, f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float
    P = C.POINTER
    sig = {
        "ffx_abi_version": ([], C.c_int),
        "ffx_last_error": ([], C.c_char_p),
        "ffx_device_count": ([P(C.c_int)], C.c_int),
        "ffx_device_info": ([C.c_int, C.c_char_p, C.c_size_t, P(i64), P(i64), P(i32)], C.c_int),
        "ffx_index_create": ([C.c_int, i64, i32, P(vp)], C.c_int),
        "ffx_index_create_typed": ([C.c_int, i64, i32, i32, P(vp)], C.c_int),
        "ffx_index_dtype": ([vp, P(i32)], C.c_int),
        "ffx_index_set_codebook": ([vp, vp, i32, i32, i32], C.c_int),
        "ffx_index_query_dim": ([vp, P(i32)], C.c_int),
        "ffx_index_destroy": ([vp], C.c_int),
        "ffx_index_dim": ([vp, P(i32)], C.c_int),
        "ffx_index_capacity": ([vp, P(i64)], C.c_int),
        "ffx_index_device_ptr": ([vp, P(vp)], C.c_int),
        "ffx_index_reseat": ([vp, C.c_int32, P(ReseatReport)], C.c_int),
        "ffx_index_upload": ([vp, vp, i64, i64], C.c_int),
        "ffx_index_upload_rows": ([vp, vp, i64, i64, vp, i64, i64, i32], C.c_int),
        "ffx_index_upload_rows_async": ([vp, vp, i64, i64, vp, i64, i64, i32], C.c_int),
        "ffx_index_pull_rows_async": ([vp, vp, i64, i64, vp, i64, i64], C.c_int),
        "ffx_index_prepare_score": ([vp, i64, i64, C.c_int32, C.c_int32, C.c_int32], C.c_int),
        "ffx_host_fill_runs": ([vp, i64, C.c_int32, vp], C.c_int),
        "ffx_index_upload_after": ([vp, vp], C.c_int),
        "ffx_index_upload_wait": ([vp], C.c_int),
        "ffx_index_copy_rows": ([vp, vp, i64], C.c_int),
        "ffx_index_gather": ([vp, vp, i64, vp], C.c_int),
        "ffx_index_fill_synthetic": ([vp, C.c_uint64, f32, i64, i64, i64], C.c_int),
        "ffx_index_set_doc_rows": ([vp, vp, i64, vp, i64], C.c_int),
        "ffx_index_set_psg_rows": ([vp, vp, i64], C.c_int),
        "ffx_plan_create": ([vp, P(vp)], C.c_int),
        "ffx_plan_destroy": ([vp], C.c_int),
        "ffx_plan_set_queries": ([vp, vp, i64], C.c_int),
        "ffx_plan_set_queries_device": ([vp, vp, i64], C.c_int),
        "ffx_plan_copy_pairs": ([vp, vp], C.c_int),
        "ffx_plan_set_pairs_device": ([vp, vp, vp, i64, vp, C.c_int, P(i64)], C.c_int),
        "ffx_plan_set_pairs": ([vp, vp, i64, vp, vp, C.c_int, P(i64)], C.c_int),
        "ffx_plan_run": ([vp, C.c_double, C.c_uint32], C.c_int),
        "ffx_plan_set_pairs_lazy": ([vp, vp, i64, vp, vp, C.c_int], C.c_int),
        "ffx_plan_early_stopping": ([vp, i32, C.c_double, vp, i32, C.c_uint32, P(i64), P(i64), P(i32)], C.c_int),
        "ffx_plan_fetch_scored": ([vp, i64, vp, vp, P(i64)], C.c_int),
        "ffx_plan_recent_gather_ms": ([vp, i32, vp], C.c_int),
        "ffx_plan_sync": ([vp], C.c_int),
        "ffx_plan_fetch": ([vp, vp, vp, vp], C.c_int),
        "ffx_plan_stats": ([vp, P(Stats)], C.c_int),
        "ffx_plan_device_ptr": ([vp, C.c_int, P(vp), P(i64)], C.c_int),
        "ffx_plan_stream": ([vp, P(vp)], C.c_int),
        "ffx_plan_set_stream": ([vp, vp], C.c_int),
        "ffx_merge_create": ([C.c_int, P(vp)], C.c_int),
        "ffx_merge_destroy": ([vp], C.c_int),
        "ffx_merge_set": ([vp, vp, i64, vp, vp], C.c_int),
        "ffx_merge_set_device": ([vp, vp, i64, vp, vp], C.c_int),
        "ffx_merge_run": ([vp, vp, C.c_double, C.c_uint32, vp], C.c_int),
        "ffx_merge_sync": ([vp], C.c_int),
        "ffx_merge_fetch": ([vp, vp, vp, vp], C.c_int),
        "ffx_merge_device_ptr": ([vp, C.c_int, P(vp), P(i64)], C.c_int),
        "ffx_aggregate": ([C.c_int, vp, vp, i64, C.c_int, vp, C.c_double, vp, i64, C.c_uint32, vp, vp, vp], C.c_int),
        "ffx_ids_create": ([P(vp)], C.c_int),
        "ffx_ids_destroy": ([vp], C.c_int),
        "ffx_ids_count": ([vp, P(i64)], C.c_int),
        "ffx_ids_intern": ([vp, vp, i32, i64, vp], C.c_int),
        "ffx_ids_lookup": ([vp, vp, i32, i64, vp, P(i64)], C.c_int),
        "ffx_ids_lookup_packed": ([vp, C.c_char_p, i64, C.c_char, i64, vp, P(i64)], C.c_int),
        "ffx_ids_intern_packed": ([vp, C.c_char_p, i64, C.c_char, i64, vp], C.c_int),
        "ffx_ids_blob": ([vp, C.c_char, vp, i64, P(i64)], C.c_int),
        "ffx_strings_code": ([C.c_char_p, i64, C.c_char, i64, i32, vp, P(vp), P(i64), P(i64)], C.c_int),
        "ffx_strings_code_offsets": ([vp, vp, i64, C.c_char, i32, vp, P(vp), P(i64), P(i64)], C.c_int),
        "ffx_strings_code_offsets32": ([vp, vp, i64, C.c_char, i32, vp, P(vp), P(i64), P(i64)], C.c_int),
        "ffx_strings_code_fixed": ([vp, i32, i64, C.c_char, i32, vp, P(vp), P(i64), P(i64)], C.c_int),
        "ffx_host_repeated_in_groups": ([vp, vp, i64, i64, i32, P(i64)], C.c_int),
        "ffx_buffer_free": ([vp], C.c_int),
        "ffx_host_code_u64": ([vp, i64, i32, vp, vp, P(i64)], C.c_int),
        "ffx_host_owner_crc32": ([vp, i32, vp, i32, i64, i32, i32, vp], C.c_int),
        "ffx_ids_get": ([vp, i32, P(vp), P(i32)], C.c_int),
        "ffx_run_open": ([C.c_char_p, vp, P(vp), P(i64)], C.c_int),
        "ffx_run_close": ([vp], C.c_int),
        "ffx_run_counts": ([vp, P(i64), P(i64)], C.c_int),
        "ffx_run_query_id": ([vp, i64, P(vp), P(i32)], C.c_int),
        "ffx_run_tag": ([vp, P(vp), P(i32)], C.c_int),
        "ffx_run_export": ([vp, i64, vp, vp, vp, P(i64)], C.c_int),
        "ffx_run_write": ([vp, C.c_char_p, vp, vp, i64, C.c_char_p], C.c_int),
        "ffx_run_flags": ([vp, P(i32), P(i32), P(i32)], C.c_int),
        "ffx_run_ids_blob": ([vp, vp, i64, P(i64)], C.c_int),
        "ffx_run_names": ([vp, P(vp), P(i64), P(i64)], C.c_int),
        "ffx_host_stage_report": ([vp, i64, P(i64)], C.c_int),
        "ffx_host_xfer_stats": ([P(C.c_uint64), P(C.c_uint64)], C.c_int),
        "ffx_build_has_experiments": ([], C.c_int),
        "ffx_index_probe_placement": ([vp, P(C.c_float)], C.c_int),
        "ffx_host_xfer_release": ([], C.c_int),
        "ffx_run_save": ([C.c_char_p, C.c_char_p, i64, i64, vp, C.c_char_p, i64, i64, vp, C.c_char_p], C.c_int),
        "ffx_run_save_coded": ([C.c_char_p, C.c_char_p, i64, i64, vp, C.c_char_p, i64, i64, vp, i64, vp, C.c_char_p], C.c_int),
        "ffx_score": ([vp, vp, i64, vp, vp, vp, C.c_int, C.c_double, C.c_uint32, vp, vp, vp, P(i64), P(Stats)], C.c_int),
        "ffx_score_ranked": ([vp, vp, i64, vp, vp, vp, vp, C.c_int, C.c_double, C.c_uint32, vp, vp, vp, P(i64), P(Stats)], C.c_int),
        "ffx_index_kept": ([vp, P(C.c_uint64), P(i64), P(i64)], C.c_int),
        "ffx_index_drop_kept": ([vp], C.c_int),
        "ffx_rank_kept": ([vp, C.c_uint64, i64, vp, vp, C.c_double, C.c_uint32, vp, vp, vp, P(Stats)], C.c_int),
        "ffx_host_alloc": ([C.c_size_t, P(vp)], C.c_int),
        "ffx_host_free": ([vp], C.c_int),
        "ffx_host_pin": ([vp, C.c_size_t], C.c_int),
        "ffx_host_unpin": ([vp], C.c_int),
        "ffx_host_order": ([vp, vp, i64, i32, vp], C.c_int),
        "ffx_host_order_blocks": ([vp, vp, vp, i64, i32, vp], C.c_int),
        "ffx_host_rank_blocks": ([vp, vp, vp, vp, i64, i32, vp, vp, vp], C.c_int),
        "ffx_host_take32": ([vp, i64, vp, i64, i32, vp], C.c_int),
        "ffx_host_interpolate": ([vp, vp, vp, i64, C.c_float, C.c_float, i32, vp, P(i32)], C.c_int),
        "ffx_host_outer_join": ([vp, vp, vp, vp, i64, i32, vp, vp, vp, vp], C.c_int),
        "ffx_host_unique_rows": ([vp, i64, i64, i32, vp, P(i64), vp], C.c_int),
        "ffx_host_rows_unique": ([vp, i64, vp, i64, vp, i64, i64, i32, vp, P(i64), vp], C.c_int),
        "ffx_host_segment_offsets": ([vp, i64, vp, i64, i32, vp], C.c_int),
        "ffx_host_segment_rows_unique": ([vp, i64, vp, i64, vp, i64, vp, i64, i32, vp, P(i64), vp], C.c_int),
        "ffx_comm_unique_id": ([C.c_int, vp], C.c_int),
        "ffx_comm_init": ([C.c_int, C.c_int, C.c_int, C.c_int, vp, P(vp)], C.c_int),
        "ffx_comm_destroy": ([vp], C.c_int),
        "ffx_comm_rank": ([vp, P(C.c_int), P(C.c_int)], C.c_int),
        "ffx_comm_stream": ([vp, P(vp)], C.c_int),
        "ffx_comm_barrier": ([vp], C.c_int),
        "ffx_comm_allreduce_f64": ([vp, vp, i32, C.c_int], C.c_int),
        "ffx_comm_allgather": ([vp, vp, i64, vp], C.c_int),
        "ffx_comm_exchange": ([vp, vp, vp, vp, vp, vp], C.c_int),
        "ffx_comm_exchange_host": ([vp, vp, vp, vp, vp], C.c_int),
        "ffx_comm_rehearsal_inbox": ([vp, vp, i64], C.c_int),
        "ffx_query_partition": ([i64, C.c_int, vp], C.c_int),
        "ffx_route": ([vp, i64, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, P(i64)], C.c_int),
        "ffx_sharded_create": ([vp, vp, vp, P(vp)], C.c_int),
        "ffx_sharded_destroy": ([vp], C.c_int),
        "ffx_sharded_set_queries": ([vp, vp, i64], C.c_int),
        "ffx_sharded_set_pairs": ([vp, vp, i64, vp, vp, C.c_int, P(i64)], C.c_int),
        "ffx_sharded_set_pairs_held": ([vp, vp, i64, vp, vp, vp, C.c_int, P(i64)], C.c_int),
        "ffx_sharded_run": ([vp, C.c_double, C.c_uint32], C.c_int),
        "ffx_sharded_sync": ([vp], C.c_int),
        "ffx_sharded_fetch": ([vp, vp, vp, vp], C.c_int),
        "ffx_sharded_owned": ([vp, P(i64), P(i64), P(i64), P(i64)], C.c_int),
        "ffx_sharded_stats": ([vp, P(ShardStats)], C.c_int),
        "ffx_sharded_device_ptr": ([vp, C.c_int, P(vp), P(i64)], C.c_int),
        "ffx_sharded_stage_owned": ([vp, vp, vp, vp, vp, vp, vp, C.c_int, P(i64)], C.c_int),
        "ffx_sharded_stage_queries": ([vp, vp, i64], C.c_int),
        "ffx_sharded_set_code_table": ([vp, vp, i64], C.c_int),
        "ffx_sharded_fetch_async": ([vp, vp, vp, vp], C.c_int),
        "ffx_sharded_score": ([vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_double, C.c_uint32, i32, vp, vp, vp, P(i64)], C.c_int),
    }
    for name, (argtypes, restype) in sig.items():
        fn = getattr(L, name)
        fn.argtypes = argtypes
        fn.restype = restype
    if L.ffx_abi_version() != 3:
        raise ImportError(f"libffx ABI {L.ffx_abi_version()} does not match this binding (3)")
    _lib = L
    return L


def stage_report() -> dict:
    """Where the time of this thread's last ``ffx_run_open`` / ``ffx_run_write`` / ``ffx_run_save*`` call went: {stage: seconds}."""
    need = C.c_int64(0)
    _check(lib().ffx_host_stage_report(None, 0, C.byref(need)))
    buf = C.create_string_buffer(max(1, need.value))
    _check(lib().ffx_host_stage_report(buf, need.value, C.byref(need)))
    return {k: float(v) for k, v in (item.split("=") for item in buf.value.decode().split(";") if item)}


def has_experiments() -> bool:
    """True for the experiments build (``make EXPERIMENTS=1``): the tuning hooks of FFX_DEBUG_VARIANT are compiled in."""
    return bool(lib().ffx_build_has_experiments())


def xfer_stats() -> dict:
    """Transfers between ordinary host memory and the device that libffx staged through its own page-locked blocks since the
    process began (csrc/host_xfer.h): {"transfers": n, "bytes": n}."""
    n, b = C.c_uint64(0), C.c_uint64(0)
    _check(lib().ffx_host_xfer_stats(C.byref(n), C.byref(b)))
    return {"transfers": int(n.value), "bytes": int(b.value)}


def xfer_release() -> None:
    """Release the library's page-locked staging blocks (they come back on the next staged transfer)."""
    _check(lib().ffx_host_xfer_release())


def usable_cores() -> int:
    """Cores this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container on a
    256-core host is often allowed 16: more busy threads than that get the whole group throttled)."""
    import os

    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), None))):
        try:
            quota, period = parse(open(path).read())
            if period is None:
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
            break
        except (OSError, ValueError):
            continue
    return max(1, n)


def _close_quietly(self) -> None:
    """``__del__`` of the handle classes: at interpreter shutdown the module's globals may already be
    gone (the process is ending and the driver frees everything anyway), which is not worth a
    traceback on stderr."""
    try:
        self.close()
    except Exception:
        pass


def _check(rc: int, bad_pair: int | None = None) -> None:
    if rc == OK:
        return
    msg = (lib().ffx_last_error() or b"").decode("utf-8", "replace")
    if rc == NOT_FOUND:
        raise NotFound(rc, msg, -1 if bad_pair is None else bad_pair)
    if rc == OOM:
        raise MemoryError(f"libffx: {msg}")
    raise FFXError(rc, msg)


def _ptr(a: np.ndarray | None):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


def device_count() -> int:
    n = C.c_int(0)
    _check(lib().ffx_device_count(C.byref(n)))
    return n.value


def device_info(device: int = 0) -> dict:
    name = C.create_string_buffer(256)
    total, free, cus = C.c_int64(0), C.c_int64(0), C.c_int32(0)
    _check(lib().ffx_device_info(device, name, 256, C.byref(total), C.byref(free), C.byref(cus)))
    # some boxes report no marketing name, only "(gfx950:...)"
    return {"name": name.value.decode().strip(), "total_bytes": total.value, "free_bytes": free.value, "compute_units": cus.value}


def require_device() -> None:
    if device_count() < 1:
        raise RuntimeError("no HIP device is visible; fast_forward_amd has no CPU fallback")


DTYPE_F32, DTYPE_F16, DTYPE_PQ8 = 0, 1, 2
_DTYPE_CODE = {np.dtype(np.float32): DTYPE_F32, np.dtype(np.float16): DTYPE_F16, np.dtype(np.uint8): DTYPE_PQ8}


class DeviceIndex:
    """Owns one ``ffx_index``: float32 / float16 rows, or uint8 product-quantizer codes (then
    ``dim`` is the code length M and ``set_codebook`` must follow), resident in HBM."""

    def __init__(self, capacity_rows: int, dim: int, device: int = 0, dtype=np.float32):
        self.dtype = np.dtype(dtype)
        if self.dtype not in _DTYPE_CODE:
            raise ValueError(f"unsupported row dtype {self.dtype}: float32, float16 or uint8 (PQ codes)")
        h = C.c_void_p()
        _check(lib().ffx_index_create_typed(device, capacity_rows, dim, _DTYPE_CODE[self.dtype], C.byref(h)))
        self._h = h
        self.capacity, self.dim, self.device = int(capacity_rows), int(dim), int(device)
        self.query_dim = 0 if self.dtype == np.uint8 else int(dim)
        self.codebook_shape: tuple[int, int, int] | None = None
        self._dependents = weakref.WeakSet()  # plans / sharded scorers built on this index: closed before it

    def set_codebook(self, codewords: np.ndarray) -> None:
        """Codewords ``[M, Ks, Ds]`` float32 of the product quantizer whose codes the rows hold."""
        codewords = _arr(codewords, np.float32)
        m, ks, ds = codewords.shape
        _check(lib().ffx_index_set_codebook(self._h, _ptr(codewords), m, ks, ds))
        self.query_dim = m * ds
        self.codebook_shape = (m, ks, ds)

    def close(self) -> None:
        if getattr(self, "_h", None):
            for dep in list(getattr(self, "_dependents", ())):  # their device state refers to this index
                dep.close()
            lib().ffx_index_destroy(self._h)
            self._h = None

    __del__ = _close_quietly

    def upload(self, rows: np.ndarray, row_offset: int) -> None:
        rows = _arr(rows, self.dtype)
        assert rows.ndim == 2 and rows.shape[1] == self.dim
        _check(lib().ffx_index_upload(self._h, _ptr(rows), row_offset, rows.shape[0]))

    def upload_wait(self) -> None:
        """Wait for an ``upload_rows(..., wait=False)`` to land."""
        _check(lib().ffx_index_upload_wait(self._h))

    def upload_rows(self, chunks, rows, dst_row: int = 0, threads: int = 0, wait: bool = True) -> None:
        """Stored rows ``rows`` (numbers over the concatenation of the equally sized 2-D arrays
        ``chunks``, e.g. the memory maps of an on-disk index) -> device rows from ``dst_row`` on,
        gathered by native threads through page-locked staging (``ffx_index_upload_rows``).  ``wait=False``:
        return when the last copies are queued (``upload_wait`` before the rows are used)."""
        chunk_rows = int(chunks[0].shape[0]) if len(chunks) else 1
        for c in chunks:
            if c.dtype != self.dtype or c.ndim != 2 or c.shape != (chunk_rows, self.dim) or not c.flags["C_CONTIGUOUS"]:
                raise ValueError("chunks must be contiguous (chunk_rows, dim) arrays of the index dtype")
        ptrs = (C.c_void_p * max(1, len(chunks)))(*[c.ctypes.data for c in chunks])
        rows = _arr(rows, np.int64)
        fn = lib().ffx_index_upload_rows if wait else lib().ffx_index_upload_rows_async
        _check(fn(self._h, ptrs, len(chunks), chunk_rows, _ptr(rows), rows.shape[0], int(dst_row), int(threads)))

    def upload_after(self, before: "DeviceIndex") -> None:
        """Asynchronous uploads of this index queued from now on run after those ``before`` has queued so far."""
        _check(lib().ffx_index_upload_after(self._h, before._h))

    def pull_rows(self, chunks, rows, dst_row: int = 0) -> None:
        """The rows of :meth:`upload_rows` pulled by the device itself out of PAGE-LOCKED chunk maps
        (:func:`pin_maps`): ``ffx_index_pull_rows_async``; ``upload_wait`` before the rows are used."""
        chunk_rows = int(chunks[0].shape[0]) if len(chunks) else 1
        key = tuple(c.ctypes.data for c in chunks)
        cached = getattr(self, "_pull_ptrs", None)
        if cached is None or cached[0] != key:
            for c in chunks:
                if c.dtype != self.dtype or c.ndim != 2 or c.shape != (chunk_rows, self.dim) or not c.flags["C_CONTIGUOUS"]:
                    raise ValueError("chunks must be contiguous (chunk_rows, dim) arrays of the index dtype")
            cached = self._pull_ptrs = (key, (C.c_void_p * max(1, len(chunks)))(*key))
        rows = _arr(rows, np.int64)
        _check(lib().ffx_index_pull_rows_async(self._h, cached[1], len(chunks), chunk_rows, _ptr(rows), rows.shape[0], int(dst_row)))

    def device_ptr(self) -> int:
        """Device address of row 0 (rows contiguous, in the index dtype)."""
        p = C.c_void_p()
        _check(lib().ffx_index_device_ptr(self._h, C.byref(p)))
        return p.value or 0

    def reseat(self, tries: int = -1) -> dict:
        """``ffx_index_reseat``: the rows move into the fastest of up to ``tries`` further allocations (a random gather over an
        allocation runs 3 - 9 % faster or slower with the physical memory it was handed).  ``tries=-1``: as many as half of the
        free memory holds, at most 7.  Transient memory: ``tries`` x the index.  Returns what was measured."""
        rep = ReseatReport()
        _check(lib().ffx_index_reseat(self._h, int(tries), C.byref(rep)))
        return {"candidates": int(rep.candidates), "chosen": int(rep.chosen), "probe_ms": [float(rep.probe_ms[i]) for i in range(rep.candidates)],
                "bytes": int(rep.bytes), "settle_s": float(rep.settle_s), "probe_ms_settled": float(rep.probe_ms_settled)}

    def probe_placement(self) -> float:
        """Milliseconds of ``ffx_index_reseat``'s probe over the block this index has (a tool's view of what reseat compares)."""
        ms = C.c_float()
        _check(lib().ffx_index_probe_placement(self._h, C.byref(ms)))
        return float(ms.value)

    def copy_rows_from(self, src: "DeviceIndex", n_rows: int) -> None:
        _check(lib().ffx_index_copy_rows(self._h, src._h, n_rows))

    def gather(self, rows) -> np.ndarray:
        rows = _arr(rows, np.int32)
        out = np.empty((rows.shape[0], self.dim), dtype=self.dtype)
        _check(lib().ffx_index_gather(self._h, _ptr(rows), rows.shape[0], _ptr(out)))
        return out

    def fill_synthetic(self, seed: int, scale: float, row_offset: int, n_rows: int, first_global_row: int | None = None) -> None:
        g = row_offset if first_global_row is None else first_global_row
        _check(lib().ffx_index_fill_synthetic(self._h, seed, scale, row_offset, n_rows, g))

    def set_doc_rows(self, doc_off, doc_rows) -> None:
        doc_off, doc_rows = _arr(doc_off, np.int64), _arr(doc_rows, np.int32)
        _check(lib().ffx_index_set_doc_rows(self._h, _ptr(doc_off), doc_off.shape[0] - 1, _ptr(doc_rows), doc_rows.shape[0]))

    def prepare_score(self, n_queries: int, n_pairs: int, has_sparse: bool, ranked: bool, page_locked_buffers: bool) -> None:
        """``ffx_index_prepare_score``: the first ``score`` / ``score_ranked`` call's allocations, made now."""
        _check(lib().ffx_index_prepare_score(self._h, int(n_queries), int(n_pairs), int(bool(has_sparse)), int(bool(ranked)), int(bool(page_locked_buffers))))

    def set_psg_rows(self, psg_row) -> None:
        psg_row = _arr(psg_row, np.int32)
        _check(lib().ffx_index_set_psg_rows(self._h, _ptr(psg_row), psg_row.shape[0]))


class _DevArray:
    """Zero-copy view of a plan buffer for consumers that speak ``__cuda_array_interface__``
    (``torch.as_tensor(view, device="cuda")``)."""

    def __init__(self, ptr: int, n: int, typestr: str, owner):
        self._owner = owner
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (ptr, False), "version": 2}


class Plan:
    """Owns one ``ffx_plan``: device workspace and launch sequence for a batch of queries."""

    def __init__(self, index: DeviceIndex):
        h = C.c_void_p()
        _check(lib().ffx_plan_create(index._h, C.byref(h)))
        self._h = h
        self._index = index  # keep the index alive as long as the plan
        index._dependents.add(self)
        self.n_pairs = 0
        self.has_sparse = False

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ffx_plan_destroy(self._h)
            self._h = None

    __del__ = _close_quietly

    def set_queries(self, queries: np.ndarray) -> None:
        queries = _arr(queries, np.float32)
        assert queries.ndim == 2 and queries.shape[1] == self._index.query_dim, "query dimensionality does not match the index"
        _check(lib().ffx_plan_set_queries(self._h, _ptr(queries), queries.shape[0]))

    def set_queries_device(self, dev_ptr: int, n_queries: int) -> None:
        """Query vectors that already sit in device memory of this GPU (fp32, ``[n_queries, query_dim]``)."""
        _check(lib().ffx_plan_set_queries_device(self._h, C.c_void_p(dev_ptr), int(n_queries)))

    def set_pairs(self, pair_off, pair_ids, sparse, mode: int) -> None:
        pair_off, pair_ids = _arr(pair_off, np.int64), _arr(pair_ids, np.int32)
        sp = None if sparse is None else _arr(sparse, np.float32)
        assert pair_ids.shape[0] == int(pair_off[-1])
        bad = C.c_int64(-1)
        rc = lib().ffx_plan_set_pairs(self._h, _ptr(pair_off), pair_off.shape[0] - 1, _ptr(pair_ids), _ptr(sp), int(mode), C.byref(bad))
        _check(rc, bad.value)
        self.n_pairs = int(pair_off[-1])
        self.has_sparse = sp is not None

    def run(self, alpha: float = 0.0, exact_sum: bool = True, sort: bool = False, sort_ff: bool = False) -> None:
        flags = (FLAG_EXACT_SUM if exact_sum else 0) | (FLAG_SORT if sort else 0) | (FLAG_SORT_FF if sort_ff else 0)
        _check(lib().ffx_plan_run(self._h, float(alpha), flags))

    def set_pairs_lazy(self, pair_off, pair_ids, sparse, mode: int) -> None:
        """Candidate lists for :meth:`early_stopping` (rank order inside a query); ordinals are not
        validated here: -1 marks an id the host could not resolve."""
        pair_off, pair_ids, sp = _arr(pair_off, np.int64), _arr(pair_ids, np.int32), _arr(sparse, np.float32)
        assert pair_ids.shape[0] == sp.shape[0] == int(pair_off[-1])
        _check(lib().ffx_plan_set_pairs_lazy(self._h, _ptr(pair_off), pair_off.shape[0] - 1, _ptr(pair_ids), _ptr(sp), int(mode)))
        self.n_pairs, self.has_sparse = int(pair_off[-1]), True

    def early_stopping(self, cutoff: int, alpha: float, depths, exact_sum: bool = True, bound_f32: bool = False) -> tuple[int, int]:
        """``ffx_plan_early_stopping``: returns (scored pairs, depth intervals run); afterwards
        ``fetch(ff=True, interp=True)`` holds NaN where a candidate was not scored.  ``NotFound``
        (with ``.pair``) when an unresolved ordinal is reached.  ``bound_f32``: the criterion's bound in float32
        (NumPy >= 2 scalar promotion) instead of float64 (NumPy < 2, the reference's pinned range)."""
        d = _arr(sorted(int(x) for x in depths), np.int32)
        bad, scored, intervals = C.c_int64(-1), C.c_int64(0), C.c_int32(0)
        flags = (FLAG_EXACT_SUM if exact_sum else 0) | (FLAG_ES_BOUND_F32 if bound_f32 else 0)
        rc = lib().ffx_plan_early_stopping(self._h, int(cutoff), float(alpha), _ptr(d), d.shape[0], flags,
                                           C.byref(bad), C.byref(scored), C.byref(intervals))
        _check(rc, bad.value)
        return scored.value, intervals.value

    def fetch_scored(self, n: int) -> tuple[np.ndarray, np.ndarray]:
        """After :meth:`early_stopping`: (positions in the candidate list, dense scores) of the ``n`` candidates that
        were scored (``ffx_plan_fetch_scored``), positions ascending."""
        pos, ff = np.empty(int(n), dtype=np.int32), np.empty(int(n), dtype=np.float32)
        got = C.c_int64(0)
        _check(lib().ffx_plan_fetch_scored(self._h, int(n), _ptr(pos), _ptr(ff), C.byref(got)))
        if got.value != n:
            raise FFXError(BAD_ARG, f"{got.value} candidates were scored, {n} expected")
        return pos, ff

    def recent_gather_ms(self, n: int) -> np.ndarray:
        """gather_dot's duration of the last ``n`` runs (HIP events), oldest first; the runs may have been queued back to back."""
        out = np.empty(int(n), dtype=np.float32)
        _check(lib().ffx_plan_recent_gather_ms(self._h, int(n), _ptr(out)))
        return out

    def sync(self) -> None:
        _check(lib().ffx_plan_sync(self._h))

    def fetch(self, ff: bool = True, interp: bool = False, order: bool = False, out=None):
        """Results as host arrays.  ``out``: optional ``(ff, interp, order)`` arrays (float32, float32, int32, each
        holding at least ``n_pairs`` elements; e.g. page-locked ones from :func:`pinned_empty`) to write into
        instead of fresh ones -- a fresh array costs a page fault per 4 KB when it is first written."""
        n = self.n_pairs

        def buf(want: bool, given, dtype):
            if not want:
                return None
            if given is None:
                return np.empty(n, dtype=dtype)
            if given.dtype != dtype or given.size < n or not given.flags["C_CONTIGUOUS"]:
                raise ValueError("out arrays must be contiguous, of the result dtype and hold every pair")
            return given[:n]

        o = out if out is not None else (None, None, None)
        a, b, c = buf(ff, o[0], np.float32), buf(interp, o[1], np.float32), buf(order, o[2], np.int32)
        _check(lib().ffx_plan_fetch(self._h, _ptr(a), _ptr(b), _ptr(c)))
        return a, b, c

    def stats(self) -> dict:
        s = Stats()
        _check(lib().ffx_plan_stats(self._h, C.byref(s)))
        return s.as_dict()

    def device_array(self, which: int) -> _DevArray:
        ptr, n = C.c_void_p(), C.c_int64(0)
        _check(lib().ffx_plan_device_ptr(self._h, which, C.byref(ptr), C.byref(n)))
        typestr = "<i4" if which in (BUF_ORDER, BUF_PAIR_IDS) else "<f4"
        return _DevArray(ptr.value or 0, n.value, typestr, self)

    def stream(self) -> int:
        s = C.c_void_p()
        _check(lib().ffx_plan_stream(self._h, C.byref(s)))
        return s.value or 0

    def set_stream(self, stream: int | None) -> None:
        """Issue the plan's work on a caller-owned hipStream_t (``None``: the plan's own)."""
        _check(lib().ffx_plan_set_stream(self._h, C.c_void_p(stream or 0)))


class Merge:
    """Owns one ``ffx_merge``: owner-side scatter + interpolation + ordering of scores that
    arrived from the ranks of a row-sharded index."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        _check(lib().ffx_merge_create(device, C.byref(h)))
        self._h = h
        self.n_pairs = 0
        self.has_sparse = False

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ffx_merge_destroy(self._h)
            self._h = None

    __del__ = _close_quietly

    def set(self, pair_off, recv_slot, sparse=None) -> None:
        pair_off, recv_slot = _arr(pair_off, np.int64), _arr(recv_slot, np.int32)
        sp = None if sparse is None else _arr(sparse, np.float32)
        _check(lib().ffx_merge_set(self._h, _ptr(pair_off), pair_off.shape[0] - 1, _ptr(recv_slot), _ptr(sp)))
        self.n_pairs = int(pair_off[-1])
        self.has_sparse = sp is not None

    def run(self, recv_dev_ptr: int, alpha: float, sort: bool = False, stream: int | None = None) -> None:
        flags = FLAG_SORT if sort else 0
        _check(lib().ffx_merge_run(self._h, C.c_void_p(recv_dev_ptr), float(alpha), flags, C.c_void_p(stream or 0)))

    def sync(self) -> None:
        _check(lib().ffx_merge_sync(self._h))

    def fetch(self, ff: bool = True, interp: bool = False, order: bool = False):
        n = self.n_pairs
        a = np.empty(n, dtype=np.float32) if ff else None
        b = np.empty(n, dtype=np.float32) if interp else None
        c = np.empty(n, dtype=np.int32) if order else None
        _check(lib().ffx_merge_fetch(self._h, _ptr(a), _ptr(b), _ptr(c)))
        return a, b, c

    def device_array(self, which: int) -> _DevArray:
        ptr, n = C.c_void_p(), C.c_int64(0)
        _check(lib().ffx_merge_device_ptr(self._h, which, C.byref(ptr), C.byref(n)))
        return _DevArray(ptr.value or 0, n.value, "<i4" if which == BUF_ORDER else "<f4", self)


def _free_pinned(self) -> None:
    addr = C.addressof(self)
    if addr and _lib is not None:
        _lib.ffx_host_free(C.c_void_p(addr))


def pinned_empty(shape, dtype) -> np.ndarray:
    """``np.empty`` in page-locked host memory (``ffx_host_alloc``): copies to and from the device
    run at link rate and overlap the kernels.  The memory is released with its last view."""
    dtype = np.dtype(dtype)
    shape = (int(shape),) if np.isscalar(shape) else tuple(int(x) for x in shape)
    n = int(np.prod(shape, dtype=np.int64)) if shape else 1
    nbytes = max(1, n * dtype.itemsize)
    p = C.c_void_p()
    _check(lib().ffx_host_alloc(nbytes, C.byref(p)))
    # every view keeps this ctypes object alive through the buffer protocol; it frees on collection
    owner = type("PinnedBytes", (C.c_char * nbytes,), {"__del__": _free_pinned}).from_address(p.value)
    return np.frombuffer(owner, dtype=dtype, count=n).reshape(shape)


# ---- a pool of page-locked result blocks --------------------------------------------------------------------
# The arrays a large result is handed out in.  A fresh 140 MB NumPy array costs a page fault per 4 KB when it is first
# written (~10 ms) and an munmap when it dies (~5 ms), and the device cannot copy into it at link rate; a page-locked block
# costs ~40 ms to make.  So result blocks are page-locked ONCE and go back to this pool when the last view of them dies: the
# next call's results land straight in them (no gather, no copy, no fault).  Bounded: at most FFX_RESULT_POOL_BYTES (default
# 4 GiB) of idle blocks are kept; beyond that a dying block is released.
_POOL: dict = {}
_POOL_IDLE = [0]
# the bookkeeping is touched by the caller's thread, by reserve_pooled's helper thread and by finalisers (which may run inside
# any of them, hence re-entrant)
import threading as _threading  # noqa: E402

_POOL_LOCK = _threading.RLock()


def _pool_limit() -> int:
    import os

    try:
        return int(os.environ.get("FFX_RESULT_POOL_BYTES", str(4 << 30)))
    except ValueError:
        return 4 << 30


def _pool_class(nbytes: int) -> int:
    """Block size a request is served from: the next of 1, 1.25, 1.5, 1.75 x 2^k (at most 25 % over)."""
    nbytes = max(nbytes, 1 << 16)
    k = nbytes.bit_length() - 1
    for q in (4, 5, 6, 7, 8):
        if (q << k) >> 2 >= nbytes:
            return (q << k) >> 2
    return 2 << k


def _return_pooled(self) -> None:
    addr = C.addressof(self)
    if not addr or _lib is None:
        return
    cap = C.sizeof(self)
    try:
        with _POOL_LOCK:
            if _POOL_IDLE[0] + cap <= _pool_limit():
                _POOL.setdefault(cap, []).append(addr)
                _POOL_IDLE[0] += cap
                return
    except Exception:  # interpreter shutdown: the module's globals may be gone
        pass
    _lib.ffx_host_free(C.c_void_p(addr))


def pooled_empty(n: int, dtype) -> np.ndarray:
    """``np.empty(n, dtype)`` in a page-locked block of the result pool (see above): released back to the pool, not to
    the system, when its last view dies."""
    dtype = np.dtype(dtype)
    n = int(n)
    cap = _pool_class(max(1, n * dtype.itemsize))
    addr = None
    with _POOL_LOCK:
        free = _POOL.get(cap)
        if free:
            addr = free.pop()
            _POOL_IDLE[0] -= cap
    if addr is None:
        p = C.c_void_p()
        _check(lib().ffx_host_alloc(cap, C.byref(p)))
        addr = p.value
    owner = type("PooledBytes", (C.c_char * cap,), {"__del__": _return_pooled}).from_address(addr)
    return np.frombuffer(owner, dtype=dtype, count=n)


_RESERVING: list = []  # helper threads of reserve_pooled that may still be page-locking


def reserve_pooled(requests, first=None) -> "object":
    """Make sure the pool can serve ``requests`` (``[(n, dtype), ...]``) without page-locking anything, on a helper thread:
    page-locking 2 x 140 MB takes ~0.1 s of driver time that a first call can spend looking its ids up instead.  Blocks that
    exist already are left alone.  ``first``: something else for that thread to do before (an index brings its device-side
    state up to date: ``InMemoryIndex._warm_up``); whatever it raises is dropped -- the caller's own thread does the same
    work again where it needs it, and reports.  Returns the thread (``join`` it before the blocks are needed).

    Tried and dropped (tools/cold_api_probe.py): not waiting for the page-locking -- results in ordinary arrays when the
    blocks are not ready, the thread carrying on beside the device call.  A call whose results land in fresh ordinary
    memory takes 25 - 43 ms instead of 19 by itself, but 100 ms with the page-locking thread beside it (the driver serialises
    the two): worse than waiting the ~60 ms a new pair of blocks costs, once per number of results alive at a time."""
    import threading

    lib()
    need: dict = {}
    for n, dtype in requests:
        cap = _pool_class(max(1, int(n) * np.dtype(dtype).itemsize))
        need[cap] = need.get(cap, 0) + 1
    with _POOL_LOCK:
        todo = [cap for cap, k in need.items() for _ in range(max(0, k - len(_POOL.get(cap, ()))))]

    def work():
        try:
            if first is not None:
                first()
        except Exception:  # noqa: BLE001
            pass
        for cap in todo:
            p = C.c_void_p()
            if _lib.ffx_host_alloc(cap, C.byref(p)) == OK:
                with _POOL_LOCK:
                    _POOL.setdefault(cap, []).append(p.value)
                    _POOL_IDLE[0] += cap

    t = threading.Thread(target=work, daemon=True)
    _RESERVING[:] = [x for x in _RESERVING if x.is_alive()] + [t]
    t.start()
    return t


def drain_result_pool(wait: bool = True) -> int:
    """Release every idle block of the result pool; returns the bytes freed.  ``wait=False`` (an index being closed, possibly
    from a finaliser): the helper threads of ``reserve_pooled`` are not joined -- a block one of them adds afterwards stays
    for the next call."""
    if _lib is None:
        return 0
    if wait:
        for t in list(_RESERVING):  # nobody adds a block behind this call's back
            t.join()
        _RESERVING.clear()
    freed = 0
    with _POOL_LOCK:
        taken = [(cap, blocks[:]) for cap, blocks in _POOL.items()]
        for blocks in _POOL.values():
            del blocks[:]
        _POOL_IDLE[0] = 0
    for cap, blocks in taken:
        for addr in blocks:
            _lib.ffx_host_free(C.c_void_p(addr))
            freed += cap
    return freed


class pinned:
    """Context manager: page-lock existing arrays for the duration (``ffx_host_pin``)."""

    min_bytes = 1 << 20  # smaller arrays gain nothing and may share pages with one another

    def __init__(self, *arrays):
        self._arrays = [a for a in arrays if a is not None and a.nbytes >= self.min_bytes]
        self._done = []

    def __enter__(self):
        for a in self._arrays:
            if not a.flags["C_CONTIGUOUS"]:
                raise ValueError("only contiguous arrays can be page-locked")
            _check(lib().ffx_host_pin(_ptr(a), a.nbytes))
            self._done.append(a)
        return self

    def __exit__(self, *exc):
        for a in reversed(self._done):
            lib().ffx_host_unpin(_ptr(a))
        self._done = []
        return False


class PinnedMaps:
    """Arrays page-locked for as long as this object lives (``ffx_host_pin`` on the pages each one spans).  Memory maps
    of a file qualify: the file's pages then stay in host memory and the device can read them
    (:meth:`DeviceIndex.pull_rows`)."""

    def __init__(self, arrays):
        self._done: list[tuple[int, int]] = []
        page = 4096
        spans = []
        for a in arrays:
            if not a.flags["C_CONTIGUOUS"]:
                raise ValueError("only contiguous arrays can be page-locked")
            if a.nbytes:
                spans.append((a.ctypes.data // page * page, -(-(a.ctypes.data + a.nbytes) // page) * page))
        merged: list[list[int]] = []  # views of one map may share pages: every page is locked once
        for lo, hi in sorted(spans):
            if merged and lo <= merged[-1][1]:
                merged[-1][1] = max(merged[-1][1], hi)
            else:
                merged.append([lo, hi])
        try:
            for lo, hi in merged:
                _check(lib().ffx_host_pin(C.c_void_p(lo), hi - lo))
                self._done.append((lo, hi - lo))
        except BaseException:
            self.close()
            raise

    def close(self) -> None:
        for lo, _ in reversed(getattr(self, "_done", [])):
            lib().ffx_host_unpin(C.c_void_p(lo))
        self._done = []

    __del__ = _close_quietly


def score(index: DeviceIndex, queries, pair_off, pair_ids, mode: int, sparse=None, alpha: float = 0.0,
          exact_sum: bool = True, sort: bool = False, out=None, want_ff: bool = True):
    """One-shot ``ffx_score``: host arrays in, host arrays out.  Returns (ff, interp, order, stats).

    ``out``: optional ``(ff, interp, order)`` arrays to write into (e.g. from :func:`pinned_empty`).
    ``want_ff=False``: the dense scores stay on the device (``ff`` is None)."""
    queries = _arr(queries, np.float32)
    pair_off, pair_ids = _arr(pair_off, np.int64), _arr(pair_ids, np.int32)
    sp = None if sparse is None else _arr(sparse, np.float32)
    n = int(pair_off[-1])
    o_ff, o_interp, o_order = out if out is not None else (None, None, None)

    def _out(a, dtype):
        if a is None:
            return np.empty(n, dtype=dtype)
        if a.dtype != dtype or a.size < n or not a.flags["C_CONTIGUOUS"]:
            raise ValueError("out arrays must be contiguous, of the result dtype and hold every pair")
        return a[:n]

    ff = _out(o_ff, np.float32) if want_ff else None
    interp = None if sp is None else _out(o_interp, np.float32)
    order = _out(o_order, np.int32) if sort else None
    bad, st = C.c_int64(-1), Stats()
    flags = (FLAG_EXACT_SUM if exact_sum else 0) | (FLAG_SORT if sort else 0)
    rc = lib().ffx_score(index._h, _ptr(queries), queries.shape[0], _ptr(pair_off), _ptr(pair_ids), _ptr(sp), int(mode),
                         float(alpha), flags, _ptr(ff), _ptr(interp), _ptr(order), C.byref(bad), C.byref(st))
    _check(rc, bad.value)
    return ff, interp, order, st.as_dict()


def score_ranked(index: DeviceIndex, queries, pair_off, pair_ids, mode: int, sparse=None, alpha: float = 0.0, carry=None,
                 exact_sum: bool = True, want_order: bool = False, sort_ff: bool = False, keep: bool = False, pool: bool = True):
    """``ffx_score_ranked``: the run in FINAL order -- (scores, carried words | None, order | None, stats), the scores
    being the interpolated ones when ``sparse`` is given, else the dense ones; ``carry``: one int32 per candidate that is
    permuted along (a ranking's id codes).  Large results land in blocks of the page-locked result pool.  ``keep``: the
    dense scores and the carried words stay in HBM with the index (``FFX_FLAG_KEEP_DENSE``; ``kept_token`` / ``rank_kept``)."""
    queries = _arr(queries, np.float32)
    pair_off, pair_ids = _arr(pair_off, np.int64), _arr(pair_ids, np.int32)
    sp = None if sparse is None else _arr(sparse, np.float32)
    ca = None if carry is None else _arr(carry, np.int32)
    n = int(pair_off[-1])
    make = pooled_empty if (pool and n >= (1 << 18)) else (lambda m, dt: np.empty(m, dtype=dt))
    scores = make(n, np.float32)
    out_carry = None if ca is None else make(n, np.int32)
    order = make(n, np.int32) if want_order else None
    bad, st = C.c_int64(-1), Stats()
    flags = (FLAG_EXACT_SUM if exact_sum else 0) | FLAG_SORT | (FLAG_SORT_FF if sort_ff else 0) | (FLAG_KEEP_DENSE if keep else 0)
    rc = lib().ffx_score_ranked(index._h, _ptr(queries), queries.shape[0], _ptr(pair_off), _ptr(pair_ids), _ptr(sp), _ptr(ca), int(mode), float(alpha), flags,
                                _ptr(scores), _ptr(out_carry), _ptr(order), C.byref(bad), C.byref(st))
    _check(rc, bad.value)
    return scores, out_carry, order, st.as_dict()


def kept_token(index: DeviceIndex) -> int:
    """``ffx_index_kept``: the token of the dense scores the index keeps (0: none)."""
    tok = C.c_uint64(0)
    _check(lib().ffx_index_kept(index._h, C.byref(tok), None, None))
    return int(tok.value)


def drop_kept(index: DeviceIndex) -> None:
    """``ffx_index_drop_kept``: release the dense scores the index keeps."""
    _check(lib().ffx_index_drop_kept(index._h))


def rank_kept(index: DeviceIndex, token: int, pair_off, sparse, alpha: float, want_carry: bool = True, want_order: bool = False):
    """``ffx_rank_kept``: (scores, carried words | None, order | None, stats) as ``score_ranked`` would return them for the
    candidate lists whose dense scores the index keeps under ``token`` -- interpolation and ordering only."""
    pair_off = _arr(pair_off, np.int64)
    sp = None if sparse is None else _arr(sparse, np.float32)
    n = int(pair_off[-1])
    make = pooled_empty if n >= (1 << 18) else (lambda m, dt: np.empty(m, dtype=dt))
    scores = make(n, np.float32)
    out_carry = make(n, np.int32) if want_carry else None
    order = make(n, np.int32) if want_order else None
    st = Stats()
    _check(lib().ffx_rank_kept(index._h, int(token), pair_off.shape[0] - 1, _ptr(pair_off), _ptr(sp), float(alpha), FLAG_SORT, _ptr(scores), _ptr(out_carry),
                               _ptr(order), C.byref(st)))
    return scores, out_carry, order, st.as_dict()


def host_order(scores, pair_off, threads: int = 0, dst_off=None) -> np.ndarray:
    """``ffx_host_order``: per-query order (scores descending, exact ties by position) computed by host
    threads; positions are indices into ``scores``.  ``dst_off``: where every query's block starts in the output
    (``ffx_host_order_blocks``: the queries laid out in another sequence in the same pass)."""
    scores, pair_off = _arr(scores, np.float32), _arr(pair_off, np.int64)
    out = np.empty(scores.shape[0], dtype=np.int32)
    if dst_off is None:
        _check(lib().ffx_host_order(_ptr(scores), _ptr(pair_off), pair_off.shape[0] - 1, int(threads), _ptr(out)))
    else:
        dst = _arr(dst_off, np.int64)
        assert dst.shape[0] >= pair_off.shape[0] - 1
        _check(lib().ffx_host_order_blocks(_ptr(scores), _ptr(pair_off), _ptr(dst), pair_off.shape[0] - 1, int(threads), _ptr(out)))
    return out


def rank_blocks(scores, carry, pair_off, dst_off=None, threads: int = 0, want_order: bool = False):
    """``ffx_host_rank_blocks``: every query ordered (scores descending, exact ties by position) and its scores and a
    second int32 column (id codes) written in that order at the query's block (``dst_off``; None: where it stands):
    ``(scores, carry[, order])`` of the whole run in final layout, one threaded pass."""
    scores, carry, pair_off = _arr(scores, np.float32), _arr(carry, np.int32), _arr(pair_off, np.int64)
    assert scores.shape[0] == carry.shape[0]
    dst = None if dst_off is None else _arr(dst_off, np.int64)
    assert dst is None or dst.shape[0] >= pair_off.shape[0] - 1
    out_s, out_c = page_locked_or_empty(scores.shape[0], np.float32), page_locked_or_empty(scores.shape[0], np.int32)
    order = np.empty(scores.shape[0], dtype=np.int32) if want_order else None
    _check(lib().ffx_host_rank_blocks(_ptr(scores), _ptr(carry), _ptr(pair_off), _ptr(dst), pair_off.shape[0] - 1, int(threads), _ptr(order),
                                      _ptr(out_s), _ptr(out_c)))
    return (out_s, out_c, order) if want_order else (out_s, out_c)


_HAVE_DEVICE: list = [None]
INPUT_BLOCKS_FROM = 32 << 20  # bytes from which a ranking's own arrays are kept in page-locked blocks (page_locked_copy)


def page_locked_or_empty(n: int, dtype) -> np.ndarray:
    """``np.empty(n, dtype)``, in a page-locked block of the result pool when the array is large and a device is there.
    For the arrays a large ranking hands to the device on EVERY call (its first-stage scores, its id codes, the candidates'
    ordinals): out of page-locked memory they go up by DMA at link rate while the caller carries on; out of ordinary memory
    the runtime stages them through its own blocks on the calling thread (30 GB/s, and libffx keeps the runtime from
    page-locking ordinary memory itself: csrc/ffx.hip).  ~35 ms per 140 MB to make, once; back to the pool with the array."""
    dtype = np.dtype(dtype)
    if int(n) * dtype.itemsize >= INPUT_BLOCKS_FROM:
        try:
            if _HAVE_DEVICE[0] is None:
                _HAVE_DEVICE[0] = device_count() > 0
            if _HAVE_DEVICE[0]:
                return pooled_empty(n, dtype)
        except Exception:  # noqa: BLE001  (no library, no device, no page-locked memory to be had: ordinary memory does)
            _HAVE_DEVICE[0] = False
    return np.empty(int(n), dtype=dtype)


def page_locked_copy(a: np.ndarray) -> np.ndarray:
    """``a`` itself, or -- a large contiguous 1-d array of 4-byte elements with a device present -- its copy in a page-locked
    block (``page_locked_or_empty``)."""
    if a.ndim != 1 or a.dtype.itemsize != 4 or a.nbytes < INPUT_BLOCKS_FROM or not a.flags["C_CONTIGUOUS"]:
        return a
    out = page_locked_or_empty(a.shape[0], a.dtype)
    if out.base is None:  # (an ordinary array came back: nothing gained by copying)
        return a
    np.copyto(out, a)
    return out


def run_numbers(off: np.ndarray, threads: int = 0) -> np.ndarray:
    """``np.repeat(np.arange(len(off) - 1), np.diff(off))`` as int64 (``ffx_host_fill_runs``: host threads share the page
    faults of the fresh array); small inputs take NumPy's route."""
    off = _arr(off, np.int64)
    n_runs = off.shape[0] - 1
    if n_runs <= 0:
        return np.zeros(0, dtype=np.int64)
    if int(off[-1]) < (1 << 20):
        return np.repeat(np.arange(n_runs, dtype=np.int64), np.diff(off))
    out = np.empty(int(off[-1]), dtype=np.int64)
    _check(lib().ffx_host_fill_runs(_ptr(off), n_runs, int(threads), _ptr(out)))
    return out


def take32(src: np.ndarray, index: np.ndarray, threads: int = 0, page_locked: bool = False) -> np.ndarray:
    """``src[index]`` for a contiguous 1-d array of 4-byte elements and an int32 index, by host threads
    (``ffx_host_take32``); small or otherwise shaped inputs take NumPy's route.  ``page_locked``: a large result lands in a
    page-locked block (``page_locked_or_empty``)."""
    if (src.ndim != 1 or src.dtype.itemsize != 4 or not src.flags["C_CONTIGUOUS"] or index.dtype != np.int32 or index.ndim != 1
            or not index.flags["C_CONTIGUOUS"] or index.shape[0] < (1 << 18)):
        return src[index]
    out = page_locked_or_empty(index.shape[0], src.dtype) if page_locked else np.empty(index.shape[0], dtype=src.dtype)
    _check(lib().ffx_host_take32(_ptr(src), src.shape[0], _ptr(index), index.shape[0], int(threads), _ptr(out)))
    return out


def host_interpolate(sparse: np.ndarray, dense_of: np.ndarray, source_row: np.ndarray, alpha: float, threads: int = 0):
    """``ffx_host_interpolate``: (scores in the sparse ranking's row order, whether any is NaN) for
    ``alpha * sparse + (1 - alpha) * dense`` where row i of the dense ranking is row ``source_row[i]`` of the
    sparse one.  float32 arithmetic with float32-rounded weights, as NumPy does for a Python-float weight."""
    sparse, dense_of, source_row = _arr(sparse, np.float32), _arr(dense_of, np.float32), _arr(source_row, np.int32)
    if not (sparse.shape == dense_of.shape == source_row.shape):
        raise ValueError("the three arrays must have one entry per row")
    out = np.empty(sparse.shape[0], dtype=np.float32)
    nan = C.c_int32(0)
    _check(lib().ffx_host_interpolate(_ptr(sparse), _ptr(dense_of), _ptr(source_row), sparse.shape[0], float(np.float32(alpha)),
                                      float(np.float32(1 - alpha)), int(threads), _ptr(out), C.byref(nan)))
    return out, bool(nan.value)


def code_strings(blob: bytes, n: int, threads: int = 0) -> tuple[np.ndarray, bytes, int]:
    """``ffx_strings_code``: (int32 code of each of the ``n`` strings of the ``'\\n'``-joined ``blob``, the distinct
    strings in code order joined by ``'\\n'``, how many there are)."""
    codes = np.empty(int(n), dtype=np.int32)
    names, n_bytes, n_names = C.c_void_p(), C.c_int64(0), C.c_int64(0)
    _check(lib().ffx_strings_code(blob, len(blob), b"\n", int(n), int(threads), _ptr(codes), C.byref(names), C.byref(n_bytes), C.byref(n_names)))
    try:
        joined = C.string_at(names.value, n_bytes.value) if names.value and n_bytes.value > 0 else b""
    finally:
        if names.value:
            lib().ffx_buffer_free(names)
    return codes, joined, int(n_names.value)


def _coded_call(fn, n: int, *args, threads: int = 0):
    """One of the ``ffx_strings_code*`` calls: (int32 codes, the distinct strings as :class:`Names`)."""
    from fast_forward_amd.names import Names

    codes = np.empty(int(n), dtype=np.int32)
    names, n_bytes, n_names = C.c_void_p(), C.c_int64(0), C.c_int64(0)
    _check(fn(*args, int(n), b"\n", int(threads), _ptr(codes), C.byref(names), C.byref(n_bytes), C.byref(n_names)))
    try:
        joined = C.string_at(names.value, n_bytes.value) if names.value and n_bytes.value > 0 else b""
    finally:
        if names.value:
            lib().ffx_buffer_free(names)
    return codes, Names(joined, int(n_names.value))


def code_arrow_strings(arr, threads: int = 0):
    """(int32 code of every element, the distinct strings as :class:`Names`) of a pyarrow ``string`` / ``large_string`` array
    (or chunked array), numbered straight from its offsets and data buffers -- no Python object per row, no copy of the
    text.  None when that does not apply (another type, nulls, a string holding a line break)."""
    import pyarrow as pa

    if isinstance(arr, pa.ChunkedArray):
        arr = arr.chunk(0) if arr.num_chunks == 1 else arr.combine_chunks()
        if isinstance(arr, pa.ChunkedArray):  # (no chunk at all)
            arr = pa.array([], type=arr.type)
    large = pa.types.is_large_string(arr.type)
    if not (large or pa.types.is_string(arr.type)) or arr.null_count:
        return None
    n = len(arr)
    if n == 0:
        from fast_forward_amd.names import Names

        return np.zeros(0, dtype=np.int32), Names(b"", 0)
    bufs = arr.buffers()  # [validity, offsets, data]
    width = 8 if large else 4
    offsets = C.c_void_p(bufs[1].address + arr.offset * width)
    data = C.c_void_p(bufs[2].address if bufs[2] is not None else 0)
    fn = lib().ffx_strings_code_offsets if large else lib().ffx_strings_code_offsets32
    codes, names = _coded_call(fn, n, data, offsets, threads=threads)
    if names.blob.count(b"\n") != max(0, names.n - 1):
        return None
    return codes, names


def code_fixed_strings(a: np.ndarray, threads: int = 0):
    """The same for a NumPy array of fixed-width bytes (``S<w>``); ``U<w>`` arrays are converted when they are ASCII."""
    if a.dtype.kind == "U":
        try:
            a = a.astype("S")
        except UnicodeEncodeError:
            return None
    if a.dtype.kind != "S" or a.ndim != 1:
        return None
    a = np.ascontiguousarray(a)
    if a.shape[0] == 0 or a.dtype.itemsize == 0:
        from fast_forward_amd.names import Names

        return np.zeros(a.shape[0], dtype=np.int32), Names(b"", 1 if a.shape[0] else 0)
    codes, names = _coded_call(lib().ffx_strings_code_fixed, a.shape[0], _ptr(a), int(a.dtype.itemsize), threads=threads)
    if names.blob.count(b"\n") != max(0, names.n - 1):
        return None
    return codes, names


def repeated_in_groups(codes: np.ndarray, group_off: np.ndarray, n_codes: int, threads: int = 0) -> int:
    """``ffx_host_repeated_in_groups``: the first group of rows holding one code twice, -1 if none."""
    codes, group_off = _arr(codes, np.int32), _arr(group_off, np.int64)
    group = C.c_int64(-1)
    _check(lib().ffx_host_repeated_in_groups(_ptr(codes), _ptr(group_off), group_off.shape[0] - 1, int(n_codes), int(threads), C.byref(group)))
    return int(group.value)


def owner_of_rows(doc_ids: "np.ndarray | None", psg_ids: "np.ndarray | None", world: int, threads: int = 0) -> np.ndarray:
    """``ffx_host_owner_crc32``: the rank holding every row, from fixed-width byte arrays (numpy ``S``) of document and
    passage ids (either may be None); equals ``zlib.crc32(id.encode()) % world`` on the row's document id, or on its
    passage id when it has no document id; 255 where a row has neither."""
    cols = []
    for a in (doc_ids, psg_ids):
        if a is None:
            cols.append((None, 0))
        else:
            a = np.ascontiguousarray(a)
            if a.dtype.kind != "S" or a.ndim != 1:
                raise ValueError("ids must be a 1-d array of fixed-width bytes")
            cols.append((a, a.dtype.itemsize))
    n = next(a.shape[0] for a, _ in cols if a is not None)
    out = np.empty(n, dtype=np.uint8)
    _check(lib().ffx_host_owner_crc32(_ptr(cols[0][0]), cols[0][1], _ptr(cols[1][0]), cols[1][1], n, int(world), int(threads), _ptr(out)))
    return out


def code_u64(address: int, n: int, threads: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """``ffx_host_code_u64`` on the ``n`` 8-byte values at ``address``: (int32 number of every value, a row holding
    each number)."""
    codes, first = np.empty(int(n), dtype=np.int32), np.empty(int(n), dtype=np.int64)
    n_distinct = C.c_int64(0)
    _check(lib().ffx_host_code_u64(C.c_void_p(int(address)), int(n), int(threads), _ptr(codes), _ptr(first), C.byref(n_distinct)))
    return codes, first[:n_distinct.value].copy()


def code_object_column(col: np.ndarray, threads: int = 0, names_as_objects: bool = True) -> "tuple[np.ndarray, np.ndarray | None, tuple[bytes, int] | None] | None":
    """(int32 code of every element, the distinct strings by code, those strings once more as one ``'\n'``-joined
    UTF-8 buffer with their number -- or None if one of them holds a newline) of an object array of Python
    strings; None when some element is not a ``str``.  The array holds POINTERS: they are numbered first
    (rows that share an object share a code at once, no string is looked at), then only the distinct objects are
    read as strings and numbered by value (``ffx_strings_code``) -- a run of 34.9 M rows names 8.7 M documents
    and 7 k queries, and pandas' own hash of 34.9 M Python strings is what the constructor used to wait for.
    ``names_as_objects=False``: where the joined buffer can stand for the name table, the object array of the distinct strings
    is not made (None comes back in its place): scattering 8.7 M string objects into it is a quarter of a second nobody
    needs while the names are only ever looked up, written or compared as bytes."""
    if col.dtype != object or col.ndim != 1 or not col.flags["C_CONTIGUOUS"] or col.itemsize != 8:
        return None
    n = col.shape[0]
    if n == 0:
        return np.zeros(0, dtype=np.int32), np.empty(0, dtype=object), None
    by_object, first = code_u64(col.ctypes.data, n, threads)
    if n >= (1 << 16) and first.shape[0] > n // 2:
        # (nearly) every cell its own object -- a frame converted from fixed-width arrays, a JSON dump: the pointers share
        # nothing worth a second level, the strings of all rows are numbered directly, in row order (objects made one
        # after the other lie one after the other in memory: a sequential read)
        got = _code_distinct_objects(col, threads)
        if got is None:
            return None
        codes, blob = got
        n_names = int(codes.max()) + 1
        packed = (blob, n_names) if (blob.count(b"\n") == n_names - 1 and b"\0" not in blob) else None
        if packed is not None and not names_as_objects:
            return codes, None, packed
        names = np.empty(n_names, dtype=object)
        names[codes] = col
        return codes, names, packed
    distinct = col[first]
    if distinct.shape[0] < 4096:  # few objects: joined in Python, numbered by the same coder (one numbering rule for any size)
        items = distinct.tolist()
        if any(type(s) is not str or "\n" in s or "\0" in s for s in items):
            if any(type(s) is not str for s in items):
                return None
            table = {s: j for j, s in enumerate(sorted(set(items)))}  # (a newline in an id: no joined buffer; numbered in sorted order)
            of_object = np.fromiter((table[s] for s in items), dtype=np.int32, count=len(items))
            names = np.empty(len(table), dtype=object)
            names[:] = list(table)
            return take32(of_object, by_object), names, None
        of_object, blob, n_names = code_strings("\n".join(items).encode("utf-8"), len(items), threads)
        names = np.empty(n_names, dtype=object)
        names[of_object] = distinct
    else:
        got = _code_distinct_objects(distinct, threads)
        if got is None:
            return None
        of_object, blob = got
        n_names = int(of_object.max()) + 1
        if not names_as_objects and blob.count(b"\n") == n_names - 1 and b"\0" not in blob:
            return take32(of_object, by_object), None, (blob, n_names)
        names = np.empty(n_names, dtype=object)
        names[of_object] = distinct  # the strings themselves: any object of a value stands for it
    packed = (blob, len(names)) if (blob.count(b"\n") == len(names) - 1 and b"\0" not in blob) else None
    return take32(of_object, by_object), (None if (packed is not None and not names_as_objects) else names), packed


def _code_distinct_objects(distinct: np.ndarray, threads: int = 0) -> "tuple[np.ndarray, bytes] | None":
    """(int32 number of every string of an object array -- equal strings, equal numbers; dense --, the distinct strings in
    number order joined by ``'\n'``), None if some element is not a ``str``.  With pyarrow at hand the strings are laid back to back by its converter (one C++ pass, no joined
    Python string, no separator to check for) and numbered from there (``ffx_strings_code_offsets``); otherwise
    through one ``'\n'``-joined buffer (``ffx_strings_code``)."""
    n = distinct.shape[0]
    try:
        import pyarrow as pa
    except ImportError:
        pa = None
    if pa is not None:
        try:
            arr = pa.array(distinct, type=pa.large_string() if n > (1 << 26) else pa.string(), from_pandas=False)
        except (pa.ArrowInvalid, pa.ArrowTypeError, TypeError, ValueError, UnicodeError):
            return None
        if arr.null_count:
            return None
        bufs = arr.buffers()  # [validity, offsets, data]
        off_t = np.int64 if pa.types.is_large_string(arr.type) else np.int32
        offsets = np.frombuffer(bufs[1], dtype=off_t, count=n + 1 + arr.offset)[arr.offset:].astype(np.int64)
        data_address = bufs[2].address if bufs[2] is not None else 0
        codes = np.empty(n, dtype=np.int32)
        names, n_bytes, n_names = C.c_void_p(), C.c_int64(0), C.c_int64(0)
        _check(lib().ffx_strings_code_offsets(C.c_void_p(data_address), _ptr(offsets), n, b"\n", int(threads), _ptr(codes), C.byref(names),
                                              C.byref(n_bytes), C.byref(n_names)))
        try:
            joined = C.string_at(names.value, n_bytes.value) if names.value and n_bytes.value > 0 else b""
        finally:
            if names.value:
                lib().ffx_buffer_free(names)
        return codes, joined
    packed = _packed(distinct)
    if packed is None:
        return None
    codes, joined, _ = code_strings(*packed, threads)
    return codes, joined


def unique_rows(rows: np.ndarray, n_index_rows: int, threads: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """``numpy.unique(rows, return_inverse=True)`` for row numbers below ``n_index_rows`` (``ffx_host_unique_rows``:
    bitmap + ranks by host threads, no sort); the inverse is int32."""
    rows = _arr(rows, np.int64)
    n = rows.shape[0]
    uniq = np.empty(max(1, min(n, int(n_index_rows))), dtype=np.int64)
    inverse = np.empty(n, dtype=np.int32)
    count = C.c_int64(0)
    _check(lib().ffx_host_unique_rows(_ptr(rows), n, int(n_index_rows), int(threads), _ptr(uniq), C.byref(count), _ptr(inverse)))
    return uniq[:count.value], inverse


def rows_unique(ids, table1: np.ndarray, table2: np.ndarray | None, n_index_rows: int, threads: int = 0) -> tuple[np.ndarray, np.ndarray]:
    """``unique_rows(table2[table1[ids]])`` (``table1[ids]`` without ``table2``) in one native pass
    (``ffx_host_rows_unique``): (distinct rows ascending, int32 slot of every id's row among them)."""
    ids = _arr(ids, np.int32)
    assert table1.dtype == np.int32 and table1.flags["C_CONTIGUOUS"] and (table2 is None or (table2.dtype == np.int32 and table2.flags["C_CONTIGUOUS"]))
    n = ids.shape[0]
    uniq = np.empty(max(1, min(n, int(n_index_rows))), dtype=np.int64)
    inverse = np.empty(n, dtype=np.int32)
    count = C.c_int64(0)
    _check(lib().ffx_host_rows_unique(_ptr(ids), n, _ptr(table1), table1.shape[0], _ptr(table2), 0 if table2 is None else table2.shape[0],
                                      int(n_index_rows), int(threads), _ptr(uniq), C.byref(count), _ptr(inverse)))
    return uniq[:count.value], inverse


def segment_rows_unique(ids, entry_off: np.ndarray, entry_row: np.ndarray, n_index_rows: int, threads: int = 0):
    """Candidates that contribute all rows of their document (``entry_off[id] .. entry_off[id + 1]`` index ``entry_row``):
    (segment offsets [n + 1], distinct rows ascending, int32 slot of every gathered row among them) in two native passes
    (``ffx_host_segment_offsets``, ``ffx_host_segment_rows_unique``)."""
    ids = _arr(ids, np.int32)
    assert entry_off.dtype == np.int32 and entry_row.dtype == np.int32 and entry_off.flags["C_CONTIGUOUS"] and entry_row.flags["C_CONTIGUOUS"]
    n = ids.shape[0]
    seg_off = np.empty(n + 1, dtype=np.int64)
    _check(lib().ffx_host_segment_offsets(_ptr(ids), n, _ptr(entry_off), entry_off.shape[0] - 1, int(threads), _ptr(seg_off)))
    total = int(seg_off[-1])
    uniq = np.empty(max(1, min(total, int(n_index_rows))), dtype=np.int64)
    inverse = np.empty(total, dtype=np.int32)
    count = C.c_int64(0)
    _check(lib().ffx_host_segment_rows_unique(_ptr(ids), n, _ptr(entry_off), entry_off.shape[0] - 1, _ptr(entry_row), entry_row.shape[0], _ptr(seg_off),
                                              int(n_index_rows), int(threads), _ptr(uniq), C.byref(count), _ptr(inverse)))
    return seg_off, uniq[:count.value], inverse


def outer_join(off_a, code_a, off_b, code_b, threads: int = 0):
    """``ffx_host_outer_join``: (u_off, u_code, u_a, u_b) -- per query the union of two id-code lists in ascending
    code, with the row of A / of B that holds each pair (-1 where a side lacks it)."""
    off_a, off_b = _arr(off_a, np.int64), _arr(off_b, np.int64)
    code_a, code_b = _arr(code_a, np.int32), _arr(code_b, np.int32)
    n_q = off_a.shape[0] - 1
    assert off_b.shape[0] == n_q + 1
    room = int(off_a[-1] + off_b[-1])
    u_off = np.empty(n_q + 1, dtype=np.int64)
    u_code, u_a, u_b = (np.empty(max(1, room), dtype=np.int32) for _ in range(3))
    _check(lib().ffx_host_outer_join(_ptr(off_a), _ptr(code_a), _ptr(off_b), _ptr(code_b), n_q, int(threads), _ptr(u_off), _ptr(u_code), _ptr(u_a), _ptr(u_b)))
    n = int(u_off[-1])
    return u_off, u_code[:n], u_a[:n], u_b[:n]


def _as_fixed_bytes(strings) -> np.ndarray:
    """Strings -> contiguous NUL-padded byte fields (numpy 'S<w>'), the ABI's string-array form."""
    a = np.asarray(strings)
    if a.dtype.kind == "S":
        return np.ascontiguousarray(a)
    if a.size == 0:
        return np.empty(0, dtype="S1")
    try:  # ASCII ids (the usual case): one vectorised conversion
        return np.ascontiguousarray(a.astype("S"))
    except (UnicodeEncodeError, ValueError, TypeError):
        return np.array([s.encode("utf-8") if isinstance(s, str) else bytes(s) for s in a.tolist()], dtype="S")


def _packed(strings) -> "tuple[bytes, int] | None":
    """Many Python strings as ONE '\n'-joined UTF-8 buffer (a fraction of the cost of a fixed-width copy
    of each), or None when that form does not apply: few strings, not all ``str``, or a string that
    holds the separator or a NUL byte."""
    if isinstance(strings, np.ndarray):
        if strings.dtype != object or strings.shape[0] < 4096:
            return None
        items = strings.tolist()
    elif isinstance(strings, (list, tuple)) and len(strings) >= 4096:
        items = strings
    else:
        return None
    try:
        blob = "\n".join(items).encode("utf-8")
    except TypeError:  # not all str
        return None
    n = len(items)
    if blob.count(b"\n") != n - 1 or b"\0" in blob:
        return None
    return blob, n


class IdStrings:
    """Owns one ``ffx_ids``: an insertion-ordered string -> ordinal table on the native side."""

    def __init__(self):
        h = C.c_void_p()
        _check(lib().ffx_ids_create(C.byref(h)))
        self._h = h

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ffx_ids_destroy(self._h)
            self._h = None

    __del__ = _close_quietly

    def __len__(self) -> int:
        n = C.c_int64(0)
        _check(lib().ffx_ids_count(self._h, C.byref(n)))
        return n.value

    def intern(self, strings) -> np.ndarray:
        packed = _packed(strings)
        if packed is not None:
            blob, n = packed
            out = np.empty(n, dtype=np.int32)
            _check(lib().ffx_ids_intern_packed(self._h, blob, len(blob), b"\n", n, _ptr(out)))
            return out
        b = _as_fixed_bytes(strings)
        out = np.empty(b.shape[0], dtype=np.int32)
        if b.shape[0]:
            _check(lib().ffx_ids_intern(self._h, _ptr(b), b.dtype.itemsize, b.shape[0], _ptr(out)))
        return out

    def intern_blob(self, blob: bytes, n: int) -> np.ndarray:
        """``intern`` for ``n`` strings that already are one ``'\\n'``-joined UTF-8 buffer (no Python string each)."""
        out = np.empty(int(n), dtype=np.int32)
        if n:
            _check(lib().ffx_ids_intern_packed(self._h, blob, len(blob), b"\n", int(n), _ptr(out)))
        return out

    def blob(self) -> bytes:
        """All strings in ordinal order, joined by ``'\\n'``."""
        need = C.c_int64(0)
        _check(lib().ffx_ids_blob(self._h, b"\n", None, 0, C.byref(need)))
        if len(self) == 0:
            return b""
        buf = C.create_string_buffer(max(1, need.value))
        _check(lib().ffx_ids_blob(self._h, b"\n", buf, need.value, C.byref(need)))
        return buf.raw[:need.value]

    def lookup(self, strings, missing_ok: bool = False, packed: "tuple[bytes, int] | None" = None) -> np.ndarray:
        """Ordinals of ``strings``; ``IndexError`` naming the first unknown one, or -1 in its place
        with ``missing_ok``.  ``packed``: the same strings as one ``'\n'``-joined buffer and their number, if the
        caller already holds them that way (the name table of a coded ranking)."""
        if packed is None or packed[1] != len(strings):
            packed = _packed(strings)
        if packed is not None:
            blob, n = packed
            out = np.empty(n, dtype=np.int32)
            miss = C.c_int64(-1)
            rc = lib().ffx_ids_lookup_packed(self._h, blob, len(blob), b"\n", n, _ptr(out), C.byref(miss))
            if rc == NOT_FOUND:
                if missing_ok:
                    return out
                raise IndexError(f"ID {strings[miss.value]} not found in the index.")
            _check(rc)
            return out
        b = _as_fixed_bytes(strings)
        out = np.empty(b.shape[0], dtype=np.int32)
        if b.shape[0] == 0:
            return out
        miss = C.c_int64(-1)
        rc = lib().ffx_ids_lookup(self._h, _ptr(b), b.dtype.itemsize, b.shape[0], _ptr(out), C.byref(miss))
        if rc == NOT_FOUND:
            if missing_ok:
                return out
            raise IndexError(f"ID {b[miss.value].decode('utf-8', 'replace')} not found in the index.")
        _check(rc)
        return out

    def get(self, ordinal: int) -> str:
        p, n = C.c_void_p(), C.c_int32(0)
        _check(lib().ffx_ids_get(self._h, int(ordinal), C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value).decode("utf-8")

    def strings(self) -> list[str]:
        """Every string of the table, in ordinal order."""
        need = C.c_int64(0)
        _check(lib().ffx_ids_blob(self._h, b"\0", None, 0, C.byref(need)))
        if len(self) == 0:
            return []
        buf = C.create_string_buffer(max(1, need.value))
        _check(lib().ffx_ids_blob(self._h, b"\0", buf, need.value, C.byref(need)))
        return buf.raw[:need.value].decode("utf-8").split("\0")


def _joined(strings) -> bytes | None:
    """'\n'-joined UTF-8 of plain strings, or None when a CSV writer would have to quote one of them
    (separator, quote or line-break characters inside, or an empty string)."""
    items = strings.tolist() if isinstance(strings, np.ndarray) else list(strings)
    try:
        blob = "\n".join(items).encode("utf-8")
    except TypeError:
        return None
    n = len(items)
    if n == 0:
        return b""
    if blob.count(b"\n") != n - 1 or any(c in blob for c in (b"\t", b'"', b"\r", b"\0")):
        return None
    if blob.startswith(b"\n") or blob.endswith(b"\n") or b"\n\n" in blob:
        return None
    return blob


def save_run(path, q_names, pair_off, ids, scores, tag: str) -> bool:
    """``ffx_run_save``: write a run grouped by query; False when the strings need a real CSV writer."""
    q_blob, id_blob = _joined(q_names), _joined(ids)
    tag_b = str(tag).encode("utf-8")
    if q_blob is None or id_blob is None or any(c in tag_b for c in (b"\t", b'"', b"\r", b"\n", b"\0")) or not tag_b:
        return False
    pair_off, scores = _arr(pair_off, np.int64), _arr(scores, np.float32)
    _check(lib().ffx_run_save(str(path).encode(), q_blob, len(q_blob), len(pair_off) - 1, _ptr(pair_off), id_blob, len(id_blob),
                              scores.shape[0], _ptr(scores), tag_b))
    return True


def save_run_coded(path, q_names, pair_off, id_codes, id_names, scores, tag: str) -> bool:
    """``ffx_run_save_coded``: the same for rows that hold id codes into ``id_names`` (an array / list of strings, or
    the ``'\\n'``-joined bytes of them with their number: ``(blob, n)``)."""
    q_blob = _joined(q_names)
    if isinstance(id_names, tuple):
        name_blob, n_names = id_names
    else:
        name_blob, n_names = _joined(id_names), len(id_names)
    tag_b = str(tag).encode("utf-8")
    if q_blob is None or name_blob is None or any(c in tag_b for c in (b"\t", b'"', b"\r", b"\n", b"\0")) or not tag_b:
        return False
    pair_off, scores, id_codes = _arr(pair_off, np.int64), _arr(scores, np.float32), _arr(id_codes, np.int32)
    _check(lib().ffx_run_save_coded(str(path).encode(), q_blob, len(q_blob), len(pair_off) - 1, _ptr(pair_off), name_blob, len(name_blob),
                                    int(n_names), _ptr(id_codes), scores.shape[0], _ptr(scores), tag_b))
    return True


class RunFile:
    """Owns one ``ffx_run``: a TREC run parsed natively into the kernels' CSR (and written back)."""

    def __init__(self, path, ids: "IdStrings | None"):
        """``ids`` None: nothing to resolve the candidate ids against; they stay strings
        (:meth:`id_strings`) and :meth:`flags` tells how pandas would have typed the columns."""
        h, bad = C.c_void_p(), C.c_int64(-1)
        rc = lib().ffx_run_open(str(path).encode(), None if ids is None else ids._h, C.byref(h), C.byref(bad))
        if rc == NOT_FOUND:
            msg = (lib().ffx_last_error() or b"").decode("utf-8", "replace")
            raise IndexError(msg)
        if rc == BAD_ARG:
            raise ValueError((lib().ffx_last_error() or b"").decode("utf-8", "replace"))
        _check(rc)
        self._h, self._ids = h, ids
        nq, npairs = C.c_int64(0), C.c_int64(0)
        _check(lib().ffx_run_counts(h, C.byref(nq), C.byref(npairs)))
        self.n_queries, self.n_pairs = nq.value, npairs.value

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ffx_run_close(self._h)
            self._h = None

    __del__ = _close_quietly

    def query_ids(self) -> list[str]:
        out = []
        p, n = C.c_void_p(), C.c_int32(0)
        for q in range(self.n_queries):
            _check(lib().ffx_run_query_id(self._h, q, C.byref(p), C.byref(n)))
            out.append(C.string_at(p.value, n.value).decode("utf-8"))
        return out

    @property
    def tag(self) -> str:
        p, n = C.c_void_p(), C.c_int32(0)
        _check(lib().ffx_run_tag(self._h, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value).decode("utf-8") if n.value else ""

    def flags(self) -> tuple[int, int, int]:
        """(q_id column, id column, file shape): FFX_TOKEN_* / FFX_RUN_* bits of a run opened without ids."""
        a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        _check(lib().ffx_run_flags(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def ids_blob(self) -> bytes:
        """Candidate ids of a run opened without an id table, grouped by query, file order inside, joined by '\\n'."""
        need = C.c_int64(0)
        _check(lib().ffx_run_ids_blob(self._h, None, 0, C.byref(need)))
        if self.n_pairs == 0:
            return b""
        buf = C.create_string_buffer(max(1, need.value))
        _check(lib().ffx_run_ids_blob(self._h, buf, need.value, C.byref(need)))
        return buf.raw[:need.value]

    def id_strings(self) -> list[str]:
        """... as Python strings."""
        return self.ids_blob().decode("utf-8").split("\n") if self.n_pairs else []

    def names(self) -> tuple[bytes, int]:
        """A run opened without an id table numbers its candidate ids while it is parsed: (the DISTINCT ids in code order
        joined by ``'\\n'``, how many); :meth:`export` hands out every row's code."""
        p, n_bytes, n_names = C.c_void_p(), C.c_int64(0), C.c_int64(0)
        _check(lib().ffx_run_names(self._h, C.byref(p), C.byref(n_bytes), C.byref(n_names)))
        return (C.string_at(p.value, n_bytes.value) if p.value and n_bytes.value > 0 else b""), int(n_names.value)

    def scores(self) -> tuple[np.ndarray, np.ndarray]:
        """(pair_off, first-stage scores) without touching ordinals (any run)."""
        pair_off = np.empty(self.n_queries + 1, dtype=np.int64)
        sparse = np.empty(self.n_pairs, dtype=np.float32)
        n = C.c_int64(0)
        _check(lib().ffx_run_export(self._h, 0, _ptr(pair_off), None, _ptr(sparse), C.byref(n)))
        return pair_off, sparse

    def export(self, cutoff: int = 0):
        """(pair_off, pair_ids, sparse); ``cutoff`` > 0 keeps each query's best ``cutoff`` first."""
        pair_off = np.empty(self.n_queries + 1, dtype=np.int64)
        n = C.c_int64(0)
        if cutoff > 0:  # trim inside the run first, then size the outputs
            _check(lib().ffx_run_export(self._h, cutoff, _ptr(pair_off), None, None, C.byref(n)))
            self.n_pairs = n.value
        pair_ids = np.empty(self.n_pairs, dtype=np.int32)
        sparse = np.empty(self.n_pairs, dtype=np.float32)
        _check(lib().ffx_run_export(self._h, 0, _ptr(pair_off), _ptr(pair_ids), _ptr(sparse), C.byref(n)))
        return pair_off, pair_ids, sparse

    def write(self, path, scores: np.ndarray, order: np.ndarray, keep: int = 0, tag: str | None = None) -> None:
        scores, order = _arr(scores, np.float32), _arr(order, np.int32)
        assert scores.shape[0] == self.n_pairs == order.shape[0]
        from pathlib import Path

        Path(path).parent.mkdir(parents=True, exist_ok=True)
        _check(lib().ffx_run_write(self._h, str(path).encode(), _ptr(scores), _ptr(order), keep,
                                   None if tag is None else tag.encode()))


def aggregate(row_scores, seg_off, mode: int, sparse=None, alpha: float = 0.0, pair_off=None, sort: bool = False, device: int = 0):
    """``ffx_aggregate``: segment max/first/mean of host row scores, interpolation and per-query
    order on the device; ``seg_off`` None = one row per candidate.  Returns (dense, interpolated|None, order|None)."""
    row_scores = _arr(row_scores, np.float32)
    seg_off = None if seg_off is None else _arr(seg_off, np.int64)
    n = row_scores.shape[0] if seg_off is None else seg_off.shape[0] - 1
    sp = None if sparse is None else _arr(sparse, np.float32)
    po = None if pair_off is None else _arr(pair_off, np.int64)
    ff = np.empty(n, dtype=np.float32)
    interp = None if sp is None else np.empty(n, dtype=np.float32)
    order = np.empty(n, dtype=np.int32) if sort else None
    _check(lib().ffx_aggregate(device, _ptr(row_scores), _ptr(seg_off), n, int(mode), _ptr(sp), float(alpha), _ptr(po),
                               0 if po is None else po.shape[0] - 1, FLAG_SORT if sort else 0, _ptr(ff), _ptr(interp), _ptr(order)))
    return ff, interp, order


# ---- multi-GPU: communicator, routing, one rank of a row-sharded index ------------------------------------
class Comm:
    """Owns one ``ffx_comm``.  ``Comm.unique_id`` on one rank, the bytes handed to the others by any
    means, then ``Comm(transport, device, rank, world, id)`` on every rank (collective)."""

    @staticmethod
    def unique_id(transport: int) -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        _check(lib().ffx_comm_unique_id(int(transport), buf))
        return buf.raw

    def __init__(self, transport: int, device: int, rank: int, world: int, uid: bytes):
        if len(uid) != COMM_ID_BYTES:
            raise ValueError(f"a communicator id has {COMM_ID_BYTES} bytes")
        h = C.c_void_p()
        _check(lib().ffx_comm_init(int(transport), int(device), int(rank), int(world), uid, C.byref(h)))
        self._h = h
        self.transport, self.device, self.rank, self.world = int(transport), int(device), int(rank), int(world)
        self._dependents = weakref.WeakSet()  # sharded scorers using this communicator's stream: closed before it

    def close(self) -> None:
        if getattr(self, "_h", None):
            for dep in list(getattr(self, "_dependents", ())):
                dep.close()
            lib().ffx_comm_destroy(self._h)
            self._h = None

    __del__ = _close_quietly

    def barrier(self) -> None:
        _check(lib().ffx_comm_barrier(self._h))

    def allreduce(self, values, op: int = OP_SUM) -> np.ndarray:
        a = np.array(values, dtype=np.float64, ndmin=1)
        _check(lib().ffx_comm_allreduce_f64(self._h, _ptr(a), a.shape[0], int(op)))
        return a

    def allgather_bytes(self, mine: bytes) -> list[bytes]:
        """Same-length byte strings of every rank, in rank order (control plane)."""
        n = len(mine)
        out = C.create_string_buffer(max(1, n * self.world))
        _check(lib().ffx_comm_allgather(self._h, mine, n, out))
        return [out.raw[r * n:(r + 1) * n] for r in range(self.world)]

    def allgather_array(self, a: np.ndarray) -> np.ndarray:
        """``(world, *a.shape)``: every rank's array of the same shape and dtype."""
        a = np.ascontiguousarray(a)
        out = np.empty((self.world,) + a.shape, dtype=a.dtype)
        _check(lib().ffx_comm_allgather(self._h, _ptr(a), a.nbytes, _ptr(out)))
        return out

    def rehearsal_inbox(self, dev_ptr: "int | None", n_words: int = 0) -> None:
        """Rehearsal transport only: the device words the peers WOULD send on the routing lane from now on."""
        _check(lib().ffx_comm_rehearsal_inbox(self._h, C.c_void_p(dev_ptr) if dev_ptr else None, int(n_words)))

    def exchange_host(self, send: np.ndarray, send_counts, recv_counts) -> np.ndarray:
        """All-to-all-v of float32 HOST arrays (host transport)."""
        send = _arr(send, np.float32)
        sc, rc = _arr(send_counts, np.int64), _arr(recv_counts, np.int64)
        assert sc.shape[0] == rc.shape[0] == self.world and int(sc.sum()) == send.shape[0]
        recv = np.empty(int(rc.sum()), dtype=np.float32)
        _check(lib().ffx_comm_exchange_host(self._h, _ptr(send), _ptr(sc), _ptr(recv), _ptr(rc)))
        return recv


def query_partition(n_queries: int, world: int) -> np.ndarray:
    b = np.empty(world + 1, dtype=np.int64)
    _check(lib().ffx_query_partition(int(n_queries), int(world), _ptr(b)))
    return b


def route(pair_off, pair_ids, rank: int, world: int, doc_bounds=None, holder=None) -> dict:
    """``ffx_route`` (host only): what rank ``rank`` scores and what it finalises."""
    pair_off = _arr(pair_off, np.int64)
    pair_ids = _arr(pair_ids, np.int32)
    n_q = pair_off.shape[0] - 1
    db = None if doc_bounds is None else _arr(doc_bounds, np.int64)
    ho = None if holder is None else _arr(holder, np.int32)
    local_off = np.empty(n_q + 1, dtype=np.int64)
    send, recv, own = np.empty(world, dtype=np.int64), np.empty(world, dtype=np.int64), np.empty(4, dtype=np.int64)
    bad = C.c_int64(-1)

    def call(local_ids, slot):
        rc = lib().ffx_route(_ptr(pair_off), n_q, _ptr(pair_ids), _ptr(ho), _ptr(db), int(rank), int(world), _ptr(local_off), _ptr(local_ids),
                             _ptr(send), _ptr(recv), _ptr(slot), _ptr(own), C.byref(bad))
        if rc == NOT_FOUND:
            raise IndexError(f"pair {bad.value} names document ordinal {int(pair_ids[bad.value])}, outside the sharded index")
        _check(rc)

    call(None, None)  # sizes
    local_ids = np.empty(int(local_off[-1]), dtype=np.int32)
    slot = np.empty(int(own[3]), dtype=np.int32)
    call(local_ids, slot)
    return {"local_pair_off": local_off, "local_pair_ids": local_ids, "send_counts": send, "recv_counts": recv, "recv_slot": slot,
            "q_lo": int(own[0]), "q_hi": int(own[1]), "own_first_pair": int(own[2]), "own_pairs": int(own[3])}


class Sharded:
    """Owns one ``ffx_sharded``: this rank's share of a row-sharded index (local plans + exchange +
    owner-side merge)."""

    def __init__(self, shard: DeviceIndex, comm: Comm, doc_bounds=None):
        db = None if doc_bounds is None else _arr(doc_bounds, np.int64)
        if db is not None and db.shape[0] != comm.world + 1:
            raise ValueError("doc_bounds must have world + 1 entries")
        h = C.c_void_p()
        _check(lib().ffx_sharded_create(shard._h, comm._h, _ptr(db), C.byref(h)))
        self._h, self._shard, self._comm = h, shard, comm
        shard._dependents.add(self)
        comm._dependents.add(self)
        self.has_sparse = False
        self.q_lo = self.q_hi = self.own_first_pair = self.own_pairs = 0

    def close(self) -> None:
        if getattr(self, "_h", None):
            lib().ffx_sharded_destroy(self._h)
            self._h = None

    __del__ = _close_quietly

    def set_queries(self, queries: np.ndarray) -> None:
        queries = _arr(queries, np.float32)
        assert queries.ndim == 2 and queries.shape[1] == self._shard.query_dim, "query dimensionality does not match the index"
        _check(lib().ffx_sharded_set_queries(self._h, _ptr(queries), queries.shape[0]))

    def _owned(self) -> None:
        a, b, c, d = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        _check(lib().ffx_sharded_owned(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        self.q_lo, self.q_hi, self.own_first_pair, self.own_pairs = a.value, b.value, c.value, d.value

    def set_pairs(self, pair_off, pair_ids, sparse, mode: int, holder=None) -> None:
        """CSR of ALL queries.  ``pair_ids``: global document ordinals (shards = the ``doc_bounds``
        ranges), or with ``holder`` the holder-local ordinal of every pair."""
        pair_off, pair_ids = _arr(pair_off, np.int64), _arr(pair_ids, np.int32)
        sp = None if sparse is None else _arr(sparse, np.float32)
        assert pair_ids.shape[0] == int(pair_off[-1])
        bad = C.c_int64(-1)
        if holder is None:
            rc = lib().ffx_sharded_set_pairs(self._h, _ptr(pair_off), pair_off.shape[0] - 1, _ptr(pair_ids), _ptr(sp), int(mode), C.byref(bad))
        else:
            ho = _arr(holder, np.int32)
            rc = lib().ffx_sharded_set_pairs_held(self._h, _ptr(pair_off), pair_off.shape[0] - 1, _ptr(ho), _ptr(pair_ids), _ptr(sp), int(mode), C.byref(bad))
        _check(rc, bad.value)
        self.has_sparse = sp is not None
        self._owned()

    def _owned_args(self, q_bounds, own_queries, own_pair_off, own_pair_ids, own_sparse, own_holder):
        qb = _arr(q_bounds, np.int64)
        if qb.shape[0] != self._comm.world + 1:
            raise ValueError("q_bounds must have world + 1 entries")
        n_own = int(qb[self._comm.rank + 1] - qb[self._comm.rank])
        off, ids = _arr(own_pair_off, np.int64), _arr(own_pair_ids, np.int32)
        if off.shape[0] != n_own + 1 or ids.shape[0] != int(off[-1]):
            raise ValueError("own_pair_off / own_pair_ids do not describe this rank's block of the batch")
        qv = None
        if own_queries is not None:
            qv = _arr(own_queries, np.float32)
            if qv.shape != (n_own, self._shard.query_dim):
                raise ValueError("own_queries must hold one vector per owned query")
        sp = None if own_sparse is None else _arr(own_sparse, np.float32)
        ho = None if own_holder is None else _arr(own_holder, np.int32)
        if (sp is not None and sp.shape[0] != ids.shape[0]) or (ho is not None and ho.shape[0] != ids.shape[0]):
            raise ValueError("own_sparse / own_holder must have one entry per owned pair")
        return qb, n_own, qv, off, ids, sp, ho

    def stage_owned(self, q_bounds, own_queries, own_pair_off, own_pair_ids, own_sparse, mode: int, own_holder=None) -> None:
        """``ffx_sharded_stage_owned``: a FRESH batch of which this rank hands over only the queries it owns
        (``q_bounds[rank] .. q_bounds[rank + 1]``): their vectors, candidate lists (offsets from 0) and first-stage
        scores.  Collective; may be called while the previous batch runs.  ``run`` then runs the batch staged last."""
        qb, n_own, qv, off, ids, sp, ho = self._owned_args(q_bounds, own_queries, own_pair_off, own_pair_ids, own_sparse, own_holder)
        bad = C.c_int64(-1)
        rc = lib().ffx_sharded_stage_owned(self._h, _ptr(qb), _ptr(qv), _ptr(off), _ptr(ids), _ptr(ho), _ptr(sp), int(mode), C.byref(bad))
        _check(rc, bad.value)
        self.has_sparse = sp is not None
        self.q_lo, self.q_hi, self.own_first_pair, self.own_pairs = int(qb[self._comm.rank]), int(qb[self._comm.rank + 1]), 0, int(off[-1])

    def set_code_table(self, ordinal_of_code: "np.ndarray | None") -> None:
        """``ffx_sharded_set_code_table``: with explicit holders, candidates travel as codes and the holder translates them
        (``ordinal_of_code[c]``: this rank's ordinal of code c, -1 where it does not hold it); None: ordinals again."""
        t = None if ordinal_of_code is None else _arr(ordinal_of_code, np.int32)
        _check(lib().ffx_sharded_set_code_table(self._h, _ptr(t), 0 if t is None else t.shape[0]))

    def stage_queries(self, all_queries: np.ndarray) -> None:
        """``ffx_sharded_stage_queries``: ALL query vectors of the batch about to be staged, replicated on every rank (then
        ``stage_owned(..., own_queries=None, ...)``).  The array must stay alive until that call returns."""
        q = _arr(all_queries, np.float32)
        assert q.ndim == 2 and q.shape[1] == self._shard.query_dim, "query dimensionality does not match the index"
        self._staged_queries = q
        _check(lib().ffx_sharded_stage_queries(self._h, _ptr(q), q.shape[0]))

    def fetch_async(self, out, ff: bool = True, interp: bool = False, order: bool = False) -> None:
        """``ffx_sharded_fetch_async``: the batch merged last (after ``run``: the one BEFORE the batch it ran) copied into
        ``out = (ff, interp, order)`` arrays (page-locked ones for a truly asynchronous copy); complete after ``sync``."""
        a, b, c = (out[0] if ff else None), (out[1] if interp else None), (out[2] if order else None)
        _check(lib().ffx_sharded_fetch_async(self._h, _ptr(a), _ptr(b), _ptr(c)))

    def score_owned(self, q_bounds, own_queries, own_pair_off, own_pair_ids, own_sparse, mode: int, alpha: float = 0.0, exact_sum: bool = True,
                    sort: bool = False, n_slices: int = 0, own_holder=None, out=None):
        """``ffx_sharded_score``: this rank's block of a fresh batch from host arrays to host arrays, cut into ``n_slices``
        slices (0: by size; the same on every rank) that are routed, scored, exchanged, merged and fetched back to back.
        Returns (dense, interpolated|None, order|None) of the owned queries."""
        qb, n_own, qv, off, ids, sp, ho = self._owned_args(q_bounds, own_queries, own_pair_off, own_pair_ids, own_sparse, own_holder)
        n = ids.shape[0]
        o = out if out is not None else (None, None, None)
        ff = o[0][:n] if o[0] is not None else np.empty(n, dtype=np.float32)
        interp = None if sp is None else (o[1][:n] if o[1] is not None else np.empty(n, dtype=np.float32))
        order = None if not sort else (o[2][:n] if o[2] is not None else np.empty(n, dtype=np.int32))
        flags = (FLAG_EXACT_SUM if exact_sum else 0) | (FLAG_SORT if sort else 0)
        bad = C.c_int64(-1)
        rc = lib().ffx_sharded_score(self._h, _ptr(qb), _ptr(qv), _ptr(off), _ptr(ids), _ptr(ho), _ptr(sp), int(mode), float(alpha), flags, int(n_slices),
                                     _ptr(ff), _ptr(interp), _ptr(order), C.byref(bad))
        _check(rc, bad.value)
        self.has_sparse = sp is not None
        self.q_lo, self.q_hi, self.own_first_pair, self.own_pairs = int(qb[self._comm.rank]), int(qb[self._comm.rank + 1]), 0, n
        return ff, interp, order

    def run(self, alpha: float = 0.0, exact_sum: bool = True, sort: bool = False, sort_ff: bool = False) -> None:
        flags = (FLAG_EXACT_SUM if exact_sum else 0) | (FLAG_SORT if sort else 0) | (FLAG_SORT_FF if sort_ff else 0)
        _check(lib().ffx_sharded_run(self._h, float(alpha), flags))

    def sync(self) -> None:
        _check(lib().ffx_sharded_sync(self._h))

    def fetch(self, ff: bool = True, interp: bool = False, order: bool = False, out=None):
        """Results of the owned queries.  ``out``: optional ``(ff, interp, order)`` arrays to write into (float32,
        float32, int32, each holding at least the owned pairs; e.g. page-locked ones) instead of fresh ones."""
        n = self.own_pairs

        def buf(want: bool, given, dtype):
            if not want:
                return None
            if given is None:
                return np.empty(n, dtype=dtype)
            if given.dtype != dtype or given.size < n or not given.flags["C_CONTIGUOUS"]:
                raise ValueError("out arrays must be contiguous, of the result dtype and hold every owned pair")
            return given[:n]

        o = out if out is not None else (None, None, None)
        a, b, c = buf(ff, o[0], np.float32), buf(interp, o[1], np.float32), buf(order, o[2], np.int32)
        _check(lib().ffx_sharded_fetch(self._h, _ptr(a), _ptr(b), _ptr(c)))
        return a, b, c

    def stats(self) -> dict:
        s = ShardStats()
        _check(lib().ffx_sharded_stats(self._h, C.byref(s)))
        return s.as_dict()
