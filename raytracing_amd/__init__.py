"""raytracing_amd -- MI355X-native path-tracing core behind the reference's
scene-upload / dispatch / accumulate surface."""
from . import wire, scenes  # noqa: F401

__all__ = ["wire", "scenes"]
