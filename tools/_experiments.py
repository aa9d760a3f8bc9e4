"""For the tools that flip tuning hooks of FFX_DEBUG_VARIANT (ffx_experiment, csrc/ffx_internal.h): those hooks, and the kernels
only they reach, are compiled into libffx_experiments.so (``make -C fast-forward-indexes_amd/csrc EXPERIMENTS=1``), not into the
default library.  ``use()`` -- before fast_forward_amd is imported -- points the binding at that build."""
import os
import sys
from pathlib import Path

LIB = Path(__file__).resolve().parent.parent / "fast-forward-indexes_amd" / "fast_forward_amd" / "libffx_experiments.so"


def use() -> None:
    if os.environ.get("FFX_LIBRARY"):
        return
    if not LIB.exists():
        sys.exit(f"{LIB} is missing: build it with `make -C fast-forward-indexes_amd/csrc EXPERIMENTS=1` (the default library has no tuning hooks)")
    os.environ["FFX_LIBRARY"] = str(LIB)
