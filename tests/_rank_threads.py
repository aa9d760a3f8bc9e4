"""Several ranks of one job in ONE process, a thread each (FFX_RANKS_HERE="a,b,..." beside WORLD_SIZE): how eight ranks share a GPU
on a box that admits six processes to it.  Every rank has its own communicator handle, indexes and scorer; the calls that wait
for other ranks are C calls, which release the interpreter.  Used by the sharded workers' ``__main__``."""

import os
import sys
import threading
import traceback


def run(main) -> None:
    """``main()`` as the one rank of this process (RANK / WORLD_SIZE from the environment), or ``main(rank, world)`` on a thread per
    rank named in FFX_RANKS_HERE; exits non-zero if any rank failed."""
    here = os.environ.get("FFX_RANKS_HERE")
    if not here:
        main()
        return
    world = int(os.environ["WORLD_SIZE"])
    failed = []

    def one_rank(r):
        try:
            main(r, world)
        except BaseException:  # noqa: BLE001 (SystemExit too: a rank that gives up must fail the process)
            failed.append(r)
            traceback.print_exc()

    threads = [threading.Thread(target=one_rank, args=(int(r),)) for r in here.split(",")]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    sys.exit(1 if failed else 0)
