"""Process-wide GPU context for the API layer (one process per GPU: the device is LOCAL_RANK)."""
from __future__ import annotations

import os

from . import _native

_ctx: _native.Context | None = None


def context() -> _native.Context:
    """The lazily created context of this process.  Raises if the library or the GPU is missing — no CPU path."""
    global _ctx
    if _ctx is None:
        _ctx = _native.Context(int(os.environ.get("DOTRING_DEVICE", os.environ.get("LOCAL_RANK", "0"))))
    return _ctx


def set_context(ctx: _native.Context | None) -> None:
    global _ctx
    _ctx = ctx
