from .video_model import DMC  # noqa: F401
