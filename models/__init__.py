"""`from models import DMC`, the import the reference's scripts use (dmc/test.py:24, dmc/train.py:55): with this
repository's root on sys.path the same statement binds the MI355X implementation."""
from deepvideocodec_amd.models import DMC  # noqa: F401

__all__ = ["DMC"]
