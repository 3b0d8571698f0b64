"""The HOST side of the library on the CPU: csrc/tpt_host*.cpp -- unmodified -- compiled against tests/hostemu (a stand-in for the HIP
runtime calls it makes: streams as queues of closures, events, "device" memory with poison-on-free; the kernels' launch functions
restated on the lane headers) and driven through the ctypes mirror by tests/hostemu_driver.py, whose scenarios are the GPU suite's at CPU
sizes: streaming and synchronous callers, look-ahead, DrawTest on host pointers in both seed modes, batches, stream batching, animated
scenes, resizes, the sharded loopback exchange, scene / camera changes in a stream.  Every image and ray count is held against the oracle.

Each scenario list runs under several schedules of the emulated device: eager (everything executes when it is enqueued), lazy (only
what a wait needs, at the latest legal moment -- a buffer freed, re-armed or overwritten while queued work still uses it turns into wrong
pixels or a hard stop), and seeded random interleavings.  Two runs use a "device" of two CUs so that the tail helpers (csrc/tpt_device.h)
are actually launched; the emulated launch checks the state of the counter block it meets.

TEST INFRASTRUCTURE: nothing here is reachable from the product (toypathtracer_amd/ has no CPU path and never loads these libraries)."""
import os
import subprocess
import sys

import pytest

from oracle_lib import ROOT

HERE = os.path.dirname(os.path.abspath(__file__))
BUILD = os.path.join(HERE, "_build")
HOST_UNITS = [os.path.join(ROOT, "toypathtracer_amd", "csrc", u + ".cpp") for u in ("tpt_host", "tpt_host_pipeline", "tpt_host_draw", "tpt_host_shard", "tpt_host_hooks")]
EMU_UNITS = [os.path.join(HERE, "hostemu", "hostemu_kernels.cpp"), os.path.join(HERE, "hostemu", "hip_shim.cpp")]
SOURCES = HOST_UNITS + EMU_UNITS
PIPELINE = HOST_UNITS[1]  # (where the mutation below is made)


def build(name, extra, host_source=None):
    out = os.path.join(BUILD, name)
    os.makedirs(BUILD, exist_ok=True)
    deps = SOURCES + [os.path.join(ROOT, "toypathtracer_amd", "csrc", h) for h in os.listdir(os.path.join(ROOT, "toypathtracer_amd", "csrc")) if h.endswith(".h")]
    deps += [os.path.join(HERE, "hostemu", "hip", "hip_runtime.h"), os.path.join(HERE, "hostemu", "rccl", "rccl.h")]
    if os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wno-unknown-pragmas",
           "-include", "hip/hip_runtime.h", "-I", os.path.join(HERE, "hostemu"), "-I", os.path.join(ROOT, "toypathtracer_amd", "csrc")]
    subprocess.check_call(cmd + extra + ([u if u != PIPELINE else host_source for u in SOURCES] if host_source else SOURCES) + ["-o", out, "-ldl", "-lpthread"])
    return out


@pytest.fixture(scope="module")
def runs():
    """all schedules at once, one process each (a run is ~30 s of oracle and emulation)"""
    plain = build("libtpt_hostemu.so", [])
    helpers = plain  # (the tail helpers are part of the library since round 5: HOSTEMU_CUS=2 makes its grids small enough to be helped)
    jobs = {}
    # A mutant of the host code for the harness's own sanity: the stream wait that keeps a colour slot's next trace launch behind
    # the slot's previous blend is taken out (the host's pacing loop, which normally hides such a slip, is off in that run).
    src = open(PIPELINE).read()
    wait = "        HIPCHK(hipStreamWaitEvent(ts, g.evResolve[slot], 0)); // colour buffer free again"
    assert src.count(wait) == 1, "the mutation site moved: update tests/test_host_logic.py"
    mutant_src = os.path.join(BUILD, "tpt_host_pipeline_mutant.cpp")
    os.makedirs(BUILD, exist_ok=True)
    text = src.replace(wait, "        // MUTANT (tests/test_host_logic.py): no wait for the slot's previous blend").replace('#include "tpt_context.h"', '#include "%s/toypathtracer_amd/csrc/tpt_context.h"' % ROOT)
    if not os.path.exists(mutant_src) or open(mutant_src).read() != text:
        open(mutant_src, "w").write(text)
    mutant = build("libtpt_hostemu_mutant.so", [], host_source=mutant_src)
    for key, lib, policy, cus, extra_env, args in [
            ("eager", plain, "eager", None, {}, []), ("lazy", plain, "lazy", None, {}, []), ("random", plain, "random:1", None, {}, []),
            ("lazy, no host pacing", plain, "lazy", None, {"TPT_HOST_PACE": "0"}, []),
            ("helpers lazy", helpers, "lazy", "2", {}, []), ("helpers random", helpers, "random:2", "2", {}, []),
            ("fuzz", plain, "random:7", None, {"TPT_HOST_PACE": "0"}, ["fuzz:7:30"]), ("fuzz helpers", helpers, "lazy", "2", {}, ["fuzz:5:30"]),
            ("mutant", mutant, "lazy", None, {"TPT_HOST_PACE": "0"}, ["streaming 44"])]:
        env = dict(os.environ, TPT_LIB=lib, HOSTEMU_POLICY=policy, **extra_env)
        env.pop("TPT_LIB_DIR", None)
        if cus:
            env["HOSTEMU_CUS"] = cus
        jobs[key] = subprocess.Popen([sys.executable, os.path.join(HERE, "hostemu_driver.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    out = {}
    for key, p in jobs.items():
        text = p.communicate(timeout=900)[0].decode()
        out[key] = (p.returncode, text)
    return out


@pytest.mark.parametrize("schedule", ["eager", "lazy", "random", "lazy, no host pacing"])
def test_host_logic_against_the_oracle(runs, schedule):
    rc, text = runs[schedule]
    lines = [ln for ln in text.splitlines() if ln.startswith(("OK", "FAIL"))]
    assert rc == 0 and len(lines) >= 24 and all(ln.startswith("OK") for ln in lines), text[-3000:]
    assert "synchronous device caller  " in text  # (look-ahead hits reported)


@pytest.mark.parametrize("schedule", ["helpers lazy", "helpers random"])
def test_tail_helpers_against_the_oracle(runs, schedule):
    rc, text = runs[schedule]
    lines = [ln for ln in text.splitlines() if ln.startswith(("OK", "FAIL"))]
    assert rc == 0 and len(lines) >= 24 and all(ln.startswith("OK") for ln in lines), text[-3000:]
    import re
    m = re.search(r"helper grids: (\d+) found their launch closed, (\d+) the pool dry, (\d+) took chunks", text)
    assert m and int(m.group(1)) + int(m.group(3)) > 0, "no helper grid was launched: the scenario no longer exercises the tail helpers"


@pytest.mark.parametrize("walk", ["fuzz", "fuzz helpers"])
def test_random_walk_over_the_api(runs, walk):
    """30 episodes of random frame shapes, sample counts, seed modes, kernel variants, pipeline depths, look-ahead, stream batching and
    calling patterns (the sharded loopback and scene changes among them) without re-initialising in between (tests/hostemu_driver.py: fuzz);
    105 more seeds were run by hand when this was written."""
    rc, text = runs[walk]
    assert rc == 0 and "OK   fuzz:" in text, text[-3000:]


def test_the_harness_catches_a_missing_stream_wait(runs):
    """the mutant above renders wrong pixels under the lazy schedule as soon as colour slots are reused"""
    rc, text = runs["mutant"]
    assert rc != 0 and "FAIL streaming 44 frames" in text and "words differ" in text, text[-2000:]


@pytest.mark.parametrize("n", [2, 3, 8])
def test_the_librarys_own_multi_gpu_path_with_n_processes(n, tmp_path):
    """tptCommInit / tptDrawSharded / tptDrawShardedBatch / tptShardedFinish with MORE THAN ONE rank: one process per rank on the
    emulation build, tests/hostemu/fake_rccl.cpp answering the library's dlopen of librccl.so.1 (ranks meet in a directory, ncclGather
    = rccl.h's semantics as a unit of work on the emulated stream).  72 x 52 pixels in stripes of 8 rows: 6.5 stripes, dealt unevenly,
    the last one partial, and with 8 ranks one rank owns nothing.  Rank 0 holds the assembled image and the sum of the ranks' ray
    counters -- which ride in the gathered tiles -- against the oracle's 1-GPU render, byte for byte; every rank runs under a different
    random schedule.  (What stays untested without a multi-GPU box is RCCL and xGMI themselves, not this code.)"""
    lib = build("libtpt_hostemu.so", [])
    fake_dir = os.path.join(BUILD, "fakerccl")
    os.makedirs(fake_dir, exist_ok=True)
    fake = os.path.join(fake_dir, "librccl.so.1")
    src = os.path.join(HERE, "hostemu", "fake_rccl.cpp")
    if not os.path.exists(fake) or os.path.getmtime(fake) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-include", "hip/hip_runtime.h", "-I", os.path.join(HERE, "hostemu"), src, "-o", fake, "-ldl"])
    procs = []
    for r in range(n):
        env = dict(os.environ, TPT_LIB=lib, FAKE_RCCL_DIR=str(tmp_path), HOSTEMU_POLICY="random:%d" % (r + 1) if r else "lazy",
                   LD_LIBRARY_PATH=fake_dir + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
        env.pop("TPT_LIB_DIR", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "hostemu_rank.py"), str(r), str(n), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    for r, (p, text) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and ("OK rank %d of %d" % (r, n)) in text, "rank %d:\n%s" % (r, text[-2000:])
    assert "equal the oracle's" in outs[0]


_SCENE_INFO = r"""
import json, sys
sys.path.insert(0, sys.argv[1])
from toypathtracer_amd import api
from toypathtracer_amd.scenes import stress_scene
api.InitializeTest()
out = {"default": api.scene_info()}
s, m = stress_scene(512, 8)
api.set_scene(s, m)
out["stress512"] = api.scene_info()
api.set_kernel_variant(3, 3, -1)
out["stress512_valu"] = api.scene_info()
api.set_kernel_variant(4, 3, -1)
out["stress512_matrix"] = api.scene_info()
api.set_kernel_variant(0, 3, -1)
out["stress512_again"] = api.scene_info()
api.ShutdownTest()
print(json.dumps(out))
"""


@pytest.mark.parametrize("queues", [None, "22", "32"])
def test_groups_bounds_are_off_the_matrix_cores_unless_the_host_asks(queues, tmp_path):
    """in a time-sliced process a wave that has run the matrix-core filter of the groups' bounds now and then gets wrong
    data from a member gather behind it (rounds 5-6).  Grouped scenes therefore take the two-level packed VALU filter by default --
    whatever GPU_MAX_HW_QUEUES says: nothing the library computes depends on the environment any more -- and the matrix-core bounds only
    after tptSetKernelVariant(4, ..); tptGetSceneInfo tells."""
    import json
    lib = build("libtpt_hostemu.so", [])
    script = tmp_path / "scene_info.py"
    script.write_text(_SCENE_INFO)
    env = dict(os.environ, TPT_LIB=lib, HOSTEMU_POLICY="eager")
    env.pop("TPT_LIB_DIR", None)
    env.pop("GPU_MAX_HW_QUEUES", None)
    if queues:
        env["GPU_MAX_HW_QUEUES"] = queues
    out = json.loads(subprocess.check_output([sys.executable, str(script), ROOT], env=env, timeout=300).decode().strip().splitlines()[-1])
    assert out["default"] == dict(spheres=46, groups=0, bounds_on_matrix_cores=False)
    assert out["stress512"]["spheres"] == 512 and out["stress512"]["groups"] > 0
    assert out["stress512"]["bounds_on_matrix_cores"] is False
    assert out["stress512_valu"]["groups"] == out["stress512"]["groups"] and out["stress512_valu"]["bounds_on_matrix_cores"] is False
    assert out["stress512_matrix"]["groups"] == out["stress512"]["groups"] and out["stress512_matrix"]["bounds_on_matrix_cores"] is True
    assert out["stress512_again"] == out["stress512"]


def test_a_grouped_scene_renders_the_same_in_a_process_that_exported_32_queues():
    """the scene / camera change scenario (a 300-sphere scene enters a stream) in a process that exported 32 hardware queues: nothing
    depends on that variable any more -- the scene set is staged without the groups' matrix table (the default), every frame equals the oracle"""
    lib = build("libtpt_hostemu.so", [])
    env = dict(os.environ, TPT_LIB=lib, HOSTEMU_POLICY="lazy", GPU_MAX_HW_QUEUES="32")
    env.pop("TPT_LIB_DIR", None)
    p = subprocess.run([sys.executable, os.path.join(HERE, "hostemu_driver.py"), "custom scene and camera"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert p.returncode == 0, p.stdout.decode()[-3000:]


def test_the_emulation_is_test_infrastructure_only():
    """nothing under toypathtracer_amd/, include/, examples/ or bench.py names the emulation"""
    for base, _, files in os.walk(ROOT):
        if any(part in base for part in (os.sep + "tests", os.sep + ".git", os.sep + "profiles", os.sep + "gpurun_out", os.sep + "tools")):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".sh")):
                assert "hostemu" not in open(os.path.join(base, f), errors="ignore").read(), os.path.join(base, f)
