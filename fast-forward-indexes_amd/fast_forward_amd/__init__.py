"""fast_forward_amd: the Fast-Forward re-ranking path of mrjleo/fast-forward-indexes, rebuilt for
AMD MI355X (gfx950).  Same surface as the reference for this path --

    from fast_forward_amd import Ranking
    from fast_forward_amd.index import InMemoryIndex, Mode
    from fast_forward_amd.encoder import LambdaEncoder

-- with the index resident in HBM and scoring done by hand-written HIP kernels behind the C ABI
of ``libffx.so`` (``include/ffx.h``).  There is no CPU fallback.
"""

from fast_forward_amd import encoder, index, util  # noqa: F401
from fast_forward_amd.ranking import Ranking

__all__ = ["Ranking", "encoder", "index", "util"]
__version__ = "0.1.0"
