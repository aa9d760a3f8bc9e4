"""Indexes: ``Index`` (abstract), ``Mode``, ``InMemoryIndex`` (HBM-resident).

``OnDiskIndex`` of the reference (HDF5 streaming) is not part of this
build yet."""

from fast_forward_amd.index.api import Index, Mode
from fast_forward_amd.index.hbm import InMemoryIndex

__all__ = ["Index", "Mode", "InMemoryIndex"]
