"""PyTerrier transformers over a Fast-Forward index (reference ``src/fast_forward/util/pyterrier.py:15-83``).

``FFScore`` computes dense scores for every (qid, docno) of a PyTerrier result frame and moves
the previous scores to ``score_0``; ``FFInterpolate`` combines them as
``alpha * score_0 + (1 - alpha) * score``.  ``python-terrier`` is an optional dependency: when it
is importable the classes derive from ``pt.Transformer`` and use ``pt.model.add_ranks``; without
it they are plain callables with the same ``transform`` (ranks added here), which is also how the
tests exercise them.
"""

from __future__ import annotations

import pandas as pd

from fast_forward_amd.ranking import Ranking

try:  # optional
    import pyterrier as pt

    _Base = pt.Transformer
except Exception:  # pragma: no cover - python-terrier is not installed in this image
    pt = None
    _Base = object


def _add_ranks(df: pd.DataFrame) -> pd.DataFrame:
    """PyTerrier's convention: rank 0 is the best document of a query."""
    if pt is not None:
        return pt.model.add_ranks(df, single_query=False)
    out = df.copy()
    out["rank"] = out.groupby("qid", sort=False)["score"].rank(ascending=False, method="first").astype(int) - 1
    return out.sort_values(["qid", "rank"]).reset_index(drop=True)


class FFScore(_Base):
    """Scores all query-document pairs of a PyTerrier frame with a Fast-Forward index."""

    def __init__(self, index) -> None:
        self._index = index
        if pt is not None:
            super().__init__()

    def transform(self, inp: pd.DataFrame) -> pd.DataFrame:
        ranking = Ranking(inp.rename(columns={"qid": "q_id", "docno": "id"}), copy=False, is_sorted=True)
        dense = self._index(ranking)._df.rename(columns={"q_id": "qid", "id": "docno"})
        merged = dense[["qid", "docno", "score", "query"]].merge(inp[["qid", "docno", "score"]], on=["qid", "docno"],
                                                                 suffixes=(None, "_0"))
        return _add_ranks(merged)

    __call__ = transform

    def __repr__(self) -> str:
        return f"{self.__class__.__name__}({id(self._index)}, {id(self._index._query_encoder)})"


class FFInterpolate(_Base):
    """``alpha * score_0 + (1 - alpha) * score`` on the output of ``FFScore``."""

    def __init__(self, alpha: float) -> None:
        self.alpha = alpha  # this exact attribute name is what pyterrier.GridScan tunes
        if pt is not None:
            super().__init__()

    def transform(self, inp: pd.DataFrame) -> pd.DataFrame:
        out = inp[["qid", "docno", "query"]].copy()
        out["score"] = self.alpha * inp["score_0"] + (1 - self.alpha) * inp["score"]
        return _add_ranks(out)

    __call__ = transform
