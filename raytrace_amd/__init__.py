"""raytrace_amd — MI355X (gfx950) render path for alexandervandenbroucke/raytrace.

Only the hot path lives here: the per-pixel loop `parallelTrace` (reference src/Main.hs:623-637)
as hand-written HIP kernels behind a C ABI (include/raytrace_gpu.h, raytrace_amd/csrc/), plus a
host-side mirror of the reference's scene constructors (raytrace_amd/api.py).
"""
from .api import *  # noqa: F401,F403
from . import api, scenes  # noqa: F401
