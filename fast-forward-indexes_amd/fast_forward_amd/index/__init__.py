"""Indexes: ``Index`` (abstract), ``Mode``, ``InMemoryIndex`` (HBM-resident), ``OnDiskIndex``
(file-backed, streamed through HBM while scoring)."""

from fast_forward_amd.index.api import Index, Mode
from fast_forward_amd.index.hbm import InMemoryIndex
from fast_forward_amd.index.ondisk import OnDiskIndex

__all__ = ["Index", "Mode", "InMemoryIndex", "OnDiskIndex"]
