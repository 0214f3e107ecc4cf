"""Counterpart of the reference's `stereo_vision` Python package (reference: stereo_vision/__init__.py:1)."""
from .sv import *  # noqa: F401,F403
