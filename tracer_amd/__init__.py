"""tracer_amd -- MI355X (gfx950) implementation of nopjia/tracer's per-bounce path-tracing loop.

The product is the C-ABI shared library `libtracer_amd.so` (include/tracer_abi.h, sources in
tracer_amd/csrc/). This package is only the thin ctypes binding the tests, bench.py and
__graft_entry__.py drive it through. There is NO CPU fallback: importing `tracer_amd.api`
without the built library raises.
"""
from .api import (Tracer, Scene, camera_default, write_ppm, write_pfm, Camera, Counters, TracerError, lib_path, load_library,  # noqa: F401
                  RNG_REFERENCE, RNG_DECORRELATED, PIPE_WAVEFRONT, PIPE_COMPACTED, PIPE_FUSED, PIPE_QUEUE, PIPE_AUTO)
