"""deepvideocodec_amd: the DMC inter-frame path on AMD Instinct MI355X (gfx950).

`deepvideocodec_amd.models.DMC` mirrors the reference's `models.DMC`; all per-frame
tensor work runs in the hand-written HIP kernels of `lib/libdvc_hip.so`
(declared in include/dvc_hip.h), reached through `deepvideocodec_amd.ops`.
"""
from . import _lib  # noqa: F401

__all__ = ["models", "ops"]
