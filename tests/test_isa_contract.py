"""What DESIGN.md says about the gfx950 code of the path-queue kernels, checked on the code object INSIDE the shipped library
(no GPU needed: the fat binary is unbundled and disassembled with the ROCm LLVM tools).  Each assertion is a property that cost
a profiling session to find when it was lost:

  * the LDS rings / pair lists / stack level 0 are DS operations -- a volatile access through a generic pointer compiles to a
    FLAT load, which reaches LDS through the vector-memory path (rounds 2-3: flat_load_ushort in the pop and push spins);
  * phase 1 of HitSpheres runs on the matrix cores (v_mfma_f32_32x32x16_f16), 8 of them for the <= 64-sphere table;
  * 120 VGPRs for the kernels that stage the scene in LDS: four waves per SIMD, i.e. two 8-wave workgroups per CU, and room left for
    the resolve kernel's waves; 128 for the grouped-scene instantiations (round 5: 32 -> 22 spilled registers, HBM traffic 11.3 x -> 5.9 x);
  * register spills stay where they were measured in round 5 (headline: 2 VGPRs, 12 B of scratch, 33 SGPRs spilled to lanes; grouped: 22 VGPRs, 92 B);
  * the hot path holds no IEEE division expansion beyond the cold fallbacks of the short forms (tpt_math.h).
"""
import os
import re
import shutil
import subprocess

import pytest

from oracle_lib import ROOT

LLVM = "/opt/rocm/lib/llvm/bin"
BUNDLER = os.path.join(LLVM, "clang-offload-bundler")
OBJDUMP = os.path.join(LLVM, "llvm-objdump")
READELF = os.path.join(LLVM, "llvm-readelf")
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"

pytestmark = pytest.mark.skipif(not all(os.path.exists(p) for p in (BUNDLER, OBJDUMP, READELF)) or shutil.which("objcopy") is None,
                                reason="ROCm LLVM tools not installed")

QUEUE = "_ZN3tpt19tptTraceQueueKernelILb%dELb%dEEEvNS_10KernelArgsE"  # <LDS_SCENE, BATCH>


@pytest.fixture(scope="module")
def code_object(tmp_path_factory):
    from toypathtracer_amd import api
    d = tmp_path_factory.mktemp("isa")
    fat, co = str(d / "fat.bin"), str(d / "kernels.co")
    subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", api.library_path(), fat])
    targets = subprocess.check_output([BUNDLER, "--list", "--type=o", "--input=" + fat]).decode().split()
    assert [t for t in targets if t.startswith("hipv4-amdgcn")] == [TARGET], "the library carries gfx950 code only: %r" % targets
    subprocess.check_call([BUNDLER, "--unbundle", "--type=o", "--targets=" + TARGET, "--input=" + fat, "--output=" + co])
    dis = subprocess.check_output([OBJDUMP, "-d", co]).decode()
    notes = subprocess.check_output([READELF, "--notes", co]).decode()
    bodies = {}
    for m in re.finditer(r"^[0-9a-f]+ <(\w+)>:\n(.*?)(?=^[0-9a-f]+ <\w+>:|\Z)", dis, flags=re.S | re.M):
        # one instruction per line: "\t<mnemonic> operands  // address: encoding"
        bodies[m.group(1)] = [ln.split("//")[0].split() for ln in m.group(2).splitlines() if ln.startswith("\t")]
    meta = {}
    for blk in re.split(r"\n\s+- (?=\.agpr_count)", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        meta[name] = {k: int(v) for k, v in re.findall(r"\.(\w+):\s+(\d+)\s*$", blk, flags=re.M)}
    return bodies, meta


def count(body, pattern):
    rx = re.compile(pattern)
    return sum(1 for ins in body if ins and rx.match(ins[0]))


def test_every_kernel_of_the_library_is_there(code_object):
    bodies, meta = code_object
    names = set(meta)
    for lds in (0, 1):
        for batch in (0, 1):
            assert QUEUE % (lds, batch) in names
    assert sum(1 for n in names if "tptTraceKernel" in n) == 8      # the lane-refill fallback: HS x PERSIST x LDS
    for n in ("tptResolveKernel", "tptResolveMirrorKernel", "tptResolveBatchKernel", "tptAssembleKernel", "tptDisplayKernel", "tptChunkOrderKernel"):
        assert any(n in k for k in names), n
    assert not any("Test" in k for k in names), "unit-test kernels belong to the hooks build only"
    assert set(bodies) >= names


def test_queue_kernels_keep_lds_traffic_on_ds_instructions(code_object):
    bodies, _ = code_object
    for lds in (0, 1):
        for batch in (0, 1):
            body = bodies[QUEUE % (lds, batch)]
            assert count(body, r"flat_") == 0, "a FLAT instruction in %s: an LDS pointer lost its address space" % (QUEUE % (lds, batch))
            assert count(body, r"ds_(read|load)") >= 30 and count(body, r"ds_(write|store)") >= 15
            assert count(body, r"buffer_(load|store|atomic)") == 0


def test_phase_one_runs_on_the_matrix_cores(code_object):
    bodies, _ = code_object
    for batch in (0, 1):
        # <= 64 spheres: two sphere tiles x two k steps x two ray tiles
        assert count(bodies[QUEUE % (1, batch)], r"v_mfma_f32_32x32x16_f16") == 8
        # grouped scenes: NO matrix-core instruction in the product's instantiation since round 6 (a wave that has executed the
        # matrix-core filter of the groups' bounds is not safe in a time-sliced process; the hooks build carries that path)
        assert count(bodies[QUEUE % (0, batch)], r"v_mfma") == 0
        assert count(bodies[QUEUE % (0, batch)], r"v_permlane32_swap") == 0
    for name, body in bodies.items():
        assert count(body, r"v_mfma") == count(body, r"v_mfma_f32_32x32x16_f16"), name


def test_register_budget_of_the_queue_kernels(code_object):
    bodies, meta = code_object
    for lds in (0, 1):
        for batch in (0, 1):
            m = meta[QUEUE % (lds, batch)]
            assert m["vgpr_count"] <= 128, "more than 128 VGPRs: three waves per SIMD, one workgroup per CU"
            assert m["agpr_count"] == 0
            assert m["max_flat_workgroup_size"] == 512 and m["wavefront_size"] == 64
    head = meta[QUEUE % (1, 0)]
    assert head["vgpr_count"] <= 120, head  # (the resolve kernel's waves start beside a machine full of these: DESIGN 3.4)
    assert head["vgpr_spill_count"] <= 2 and head["private_segment_fixed_size"] <= 12, head
    assert head["sgpr_spill_count"] <= 36, head  # round 5: 63 -> 33 (scalars made where they are used: uniformHere)
    assert meta[QUEUE % (1, 1)]["vgpr_count"] <= 120 and meta[QUEUE % (1, 1)]["vgpr_spill_count"] <= 2
    # the grouped-scene kernels hold the dealing state on top: 128 registers since round 5; since round 6 (no matrix-core
    # path in them) nothing is spilled inside the traversal -- what is left are two or three binary64 constants of pow5 / sin-cos that LLVM
    # hoists to the kernel's entry and parks in scratch (4-6 registers: a store each at the entry, a load each in the class code)
    for batch in (0, 1):
        m = meta[QUEUE % (0, batch)]
        assert m["vgpr_spill_count"] <= 6 and m["private_segment_fixed_size"] <= 28, m
        body = bodies[QUEUE % (0, batch)]
        assert count(body, r"scratch_") <= 6, count(body, r"scratch_")
        assert count(body, r"scratch_store") == count(body, r"scratch_load") <= 3  # (dwordx2 each: one per constant)


def test_big_spheres_of_a_grouped_scene_come_through_scalar_loads(code_object):
    """hitSpheresGroupedDeal reads {centre, r^2} and the index of every big sphere (ground, lights) wave-uniformly: as s_load_dwordx4 through
    the constant address space.  As vector loads with a wait behind each they were five serial L2 round trips per call (10.7 % of the wave
    time at configs[4], profiles/r06/r06_run22.log / r06_run24.log)."""
    bodies, _ = code_object
    for batch in (0, 1):
        assert count(bodies[QUEUE % (0, batch)], r"s_load_dwordx4") >= 6, count(bodies[QUEUE % (0, batch)], r"s_load_dwordx4")


def test_divisions_of_the_hot_path_are_the_short_forms(code_object):
    bodies, _ = code_object
    for lds in (0, 1):
        body = bodies[QUEUE % (lds, 0)]
        # what is left are the cold fallbacks: tdivSafeNum / tdivByPi / trsqrt2 out of their proven ranges, the dielectric's
        # and the camera's few general quotients (v_div_fixup closes one IEEE expansion each)
        assert count(body, r"v_div_fixup_f32") <= 10, count(body, r"v_div_fixup_f32")
        assert count(body, r"v_\w+_f64") <= 61  # sin / cos / pow5 in binary64 (glibc's algorithms), nothing else
