"""`python -m <package>.stereo_vision` (reference: stereo_vision/__main__.py:1-4)."""
from .sv import main

main()
