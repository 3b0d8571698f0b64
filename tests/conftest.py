import os
import sys

import pytest

# One hardware queue per in-flight trace kernel: the HIP runtime reads this when it STARTS, and a test that touches torch.cuda
# before the library is loaded would start it with the default of 4 (the library then measures 3 usable queues and runs a 2-deep
# pipeline: round 5, a launcher test that asked torch for the device count first turned five later tests red).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The checker's own redundancy (tests/oracle_lib.py renders every frame twice on big hosts): how often it was used and how often
    the two runs disagreed -- printed at the END of the run so that it shows in the tail a driver keeps."""
    if "oracle_lib" in sys.modules:
        o = sys.modules["oracle_lib"].Oracle
        terminalreporter.write_line("oracle self-check: redundant renders %s, disagreements between the checker's own runs: %d" % ("on" if o.redundant else "off", o.disagreements))


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle.get()


@pytest.fixture(scope="session")
def ref():
    from oracle_lib import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libtpt_ref.so not built (needs /root/reference)")
    return Ref.get()


@pytest.fixture(scope="session")
def emu():
    """tests/lane_emu.cpp: the per-lane device logic compiled for the host (test harness only)."""
    import ctypes as C
    import subprocess
    out_dir = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "liblane_emu.so")
    src = os.path.join(ROOT, "tests", "lane_emu.cpp")
    inc = os.path.join(ROOT, "toypathtracer_amd", "csrc")
    deps = [src] + [os.path.join(inc, f) for f in ("tpt_math.h", "tpt_trace.h", "tpt_scene.h")]
    if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wno-unknown-pragmas",
                               "-I", inc, src, "-o", so])
    lib = C.CDLL(so)
    lib.emu_render.restype = C.c_int64
    lib.emu_render.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p] + [C.c_int] * 6 + [C.c_uint] + [C.c_int] * 3 + [C.c_void_p]
    lib.emu_default_scene.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.emu_default_camera.argtypes = [C.c_void_p, C.c_int, C.c_int]
    for f in (lib.emu_sinf, lib.emu_cosf, lib.emu_pow5f):
        f.restype = C.c_float
        f.argtypes = [C.c_float]
    return lib


@pytest.fixture(scope="session")
def tpt():
    """The product, initialised on the GPU (gpu tests only). Fails loudly if the HIP library is missing."""
    from toypathtracer_amd import api
    api.InitializeTest()
    yield api
    api.ShutdownTest()


def _reset_knobs(tpt):
    tpt.set_scene(None)
    tpt.set_camera(None)
    tpt.set_samples_per_pixel(4)
    tpt.set_config(True, 0.9, False)
    tpt.set_seed_mode(tpt.SEED_PER_PIXEL)
    tpt.set_fold_mode(tpt.FOLD_RECURSIVE)
    tpt.set_kernel_variant(0, 3, -1)
    tpt.set_row_shard(0, 1, 0)
    tpt.set_frame_overlap(16)
    tpt.set_host_buffer_mode(False)
    tpt.set_host_lookahead(2)
    tpt.set_stream_batching(os.environ.get("TPT_FORCE_STREAM_BATCH", "1") == "1")  # the library's default; TPT_FORCE_STREAM_BATCH=0 runs the suite without it
    return tpt


@pytest.fixture()
def tpt_defaults(tpt):
    """Reset every run-time knob to its default before a test."""
    return _reset_knobs(tpt)


@pytest.fixture(scope="session")
def _hooks_session(tpt):
    yield tpt
    tpt.shutdown_hooks()


@pytest.fixture()
def tpt_hooks(_hooks_session):
    """The HOOKS build of the library (libtoypathtracer_hip_hooks.so: the product's sources + include/tpt_test_hooks.h) for the
    unit tests of the math layer / HitSpheres / the matrix filter.  Inside the test every call of the api module goes to that
    build (its own context); the product library stays loaded and initialised beside it."""
    tpt = _hooks_session
    with tpt.using_hooks():
        _reset_knobs(tpt)
        yield tpt
