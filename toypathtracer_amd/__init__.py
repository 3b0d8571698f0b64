"""toypathtracer_amd -- MI355X (gfx950 / HIP) implementation of ToyPathTracer's Trace/HitWorld/Scatter
hot path behind the reference's Test API (Cpp/Source/Test.h:10-17).

The product is the C-ABI shared library ``lib/libtoypathtracer_hip.so`` (sources in ``csrc/``,
interface in ``include/tpt_hip.h``).  This package is only the thin Python host side: a ctypes
mirror of the reference API (``api``), row-stripe sharding + RCCL gather for one-process-per-GPU
rendering (``sharding``) and scene helpers (``scenes``).  There is no CPU rendering path: without
the HIP library and a GPU, ``InitializeTest`` raises.
"""
from .api import (  # noqa: F401
    TptError, kFlagAnimate, kFlagProgressive, SEED_ROW_SERIAL, SEED_PER_PIXEL, FOLD_RECURSIVE, FOLD_FORWARD,
    InitializeTest, ShutdownTest, UpdateTest, DrawTest, GetObjectCount, GetSceneDesc, library_path, load_library,
)
