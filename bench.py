#!/usr/bin/env python
"""bench.py -- Mray/s of the Trace/HitWorld/Scatter hot path on MI355X (BASELINE.json's metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c5]     (N > 1: launches its N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W                                    (the same ranks under a launcher)
    multi-GPU configs: --gpus 8 --workload c3 (BASELINE configs[3]), --gpus 8 --workload c5 (configs[4])

A step = UpdateTest + DrawTest of ONE frame of the workload (default: BASELINE.json configs[1],
1280x720, 4 spp, built-in 46-sphere scene, progressive accumulation on), with the accumulation
buffer resident in HBM when the timed region starts.  Mray/s = rays (HitWorld calls: camera + bounce +
shadow, exactly the reference's counter, Test.cpp:122,199) of the K timed frames, summed over all
ranks, divided by the max-over-ranks wall time between two barrier+synchronize brackets.
For N > 1 the frame's rows are dealt out in 8-row stripes round-robin over the ranks and every step
includes its exchange -- ONE RCCL gather to rank 0 of each rank's blended tile plus a row carrying its
64-bit ray counter, software-pipelined against the next frames.  That exchange is the product's own:
the C ABI a Test.h host uses (tptCommGetUniqueId / tptCommInit / tptDrawSharded / tptShardedFinish,
include/tpt_hip.h section 3; torch.distributed only carries the 128-byte id and the timing reductions)
-- "exchange": "cabi" in the JSON line.  Total work is fixed as N grows -> "scaling": "strong".

One JSON line on rank 0, with
  roofline     : the trace kernel against the HBM roofline the north_star names (algorithmic bytes =
                 W*H*16 B written per frame) -- plus the FP32 VALU fraction, which is what actually binds;
                 durations from HIP events on the kernel's stream over the timed region;
  cpu_baseline : the pristine reference (oracle/_ref, SIMD path, all host cores via enkiTS) timed on
                 the same workload for a bounded ~10 s sample, rank 0 at N = 1 only;
  image_fnv / parity_checked : FNV-1a-32 of the final float image of THIS run, and whether it (and the
                 run's ray total) equalled the oracle's render of the same frames (untimed checker leg,
                 like cpu_baseline; default scene, runs of up to 64 frames).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# one hardware queue per in-flight trace kernel; read by the HIP runtime when it initialises (before torch does that)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (width, height, spp, scene, label)
    "c2": (1280, 720, 4, "default", "configs[1]: default 46-sphere scene, 1280x720, 4 spp/frame, progressive"),
    "c3": (3840, 2160, 16, "default", "configs[2]: default 46-sphere scene, 3840x2160, 16 spp/frame, progressive"),
    "c5": (1920, 1080, 8, "stress", "configs[4]: stress scene 4096 spheres, 1920x1080, 8 spp/frame, progressive"),
    "c1": (640, 360, 1, "default", "configs[0]: default scene, 640x360, 1 spp"),
}
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PEAK_FP32_TFLOPS = 157.3       # MI355X_MICROARCH.md: peak FP32 vector (counts FMA as 2)
PEAK_VALU_GINSTR = 256 * 4 * 2.4 / 2  # G wave64 VALU instructions per second: each SIMD issues one over 2 cycles (MI355X_MICROARCH.md)
FLOP_PER_SPHERE_TEST = 17      # per (ray, sphere) test
FLAG_PROGRESSIVE = 2
FLAG_ANIMATE = 1
PORT_TAKEN_RC = 98             # exit code of a rank whose rendezvous port was already in use (EADDRINUSE): self_launch retries with another port
# set by launchers that start one process per rank themselves (Open MPI, MPICH / PMI, Slurm): bench.py must not spawn N more ranks under them
FOREIGN_LAUNCHER_VARS = ("OMPI_COMM_WORLD_SIZE", "PMI_SIZE", "PMIX_RANK", "SLURM_NTASKS", "SLURM_PROCID")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def time_reference_build(variant, width, height, spp, budget_s):
    """One build of the pristine reference (oracle/_ref), all host threads via its own enkiTS scheduler."""
    from oracle_lib import Ref
    ref = Ref.get(variant)
    ref.set_spp(spp)
    bb = np.zeros((height, width, 4), np.float32)
    for f in range(2):  # 2 warm-up frames discarded
        ref.update(0.0, f, width, height, FLAG_PROGRESSIVE)
        ref.draw(0.0, f, width, height, bb, FLAG_PROGRESSIVE)
    rays, frames, t0 = 0, 0, time.perf_counter()
    while True:
        ref.update(0.0, frames + 2, width, height, FLAG_PROGRESSIVE)
        rays += ref.draw(0.0, frames + 2, width, height, bb, FLAG_PROGRESSIVE)
        frames += 1
        if (time.perf_counter() - t0 > budget_s and frames >= 10) or frames >= 200:
            break
    dt = time.perf_counter() - t0
    return rays / dt / 1e6, frames, dt


def cpu_baseline(width, height, spp, budget_s=5.0):
    """Times the checker/reference on the host cores (reported baseline, never the product path): the three builds
    BASELINE.md section 3 / SURVEY 8(d) ask for, CPU model and thread count stated."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, Ref
    cores = os.cpu_count() or 1
    model = cpu_model()
    builds = {"simd": "SIMD path as in the repo, -O2 -msse4.1 -ffp-contract=off (image bit-identical to the scalar path)",
              "fast": "SIMD path as shipped, -O3 -ffast-math -mavx2 -mfma (different bits)",
              "scalar": "scalar path (-D__EMSCRIPTEN__, Config.h:9-13), -O2 -ffp-contract=off: the parity target"}
    if Ref.available("simd"):
        res = {}
        for v in ("simd", "fast", "scalar"):
            if Ref.available(v):
                mr, frames, dt = time_reference_build(v, width, height, spp, budget_s)
                res[v] = dict(value=mr, frames=frames, seconds=dt, build=builds[v])
        return dict(value=res["simd"]["value"], unit="Mray/s", cores=cores, cpu_model=model, kind="reference",
                    sample="%d frames of %dx%dx%dspp after 2 warm-up frames, pristine reference (oracle/_ref/libtpt_ref.so: %s) "
                           "+ enkiTS on %d hardware threads, %.1f s" % (res["simd"]["frames"], width, height, spp, builds["simd"], cores,
                                                                        res["simd"]["seconds"]),
                    builds={k: dict(value=v["value"], frames=v["frames"], build=v["build"]) for k, v in res.items()})
    o = Oracle.get()
    s, m = o.default_scene()
    cam = o.default_camera(width, height)
    bb = np.zeros((height, width, 4), np.float32)
    rays, frames, t0 = 0, 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        r, _ = o.render(s, m, cam, width, height, spp, frames, backbuffer=bb)
        rays += r
        frames += 1
    dt = time.perf_counter() - t0
    return dict(value=rays / dt / 1e6, unit="Mray/s", cores=cores, cpu_model=model, kind="port",
                sample="%d frames of %dx%dx%dspp, oracle/tpt_oracle.c (OpenMP over rows), %.1f s" % (frames, width, height, spp, dt))


def image_parity(image, rays_total, width, height, spp, frames, max_frames=64, max_samples=1.6e8):
    """Checker leg (untimed, never the product path): the final image of the run that was just timed -- every frame from
    frame 0 on a zeroed tile -- against the oracle's PER_PIXEL render of the same frames, byte for byte, ray totals equal.
    -> dict for the JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import SEED_PER_PIXEL, Oracle, fnv1a
    img = np.ascontiguousarray(image.detach().cpu().numpy() if hasattr(image, "detach") else image, np.float32)
    out = {"image_fnv": "%08x" % fnv1a(img), "parity_checked": False}
    if frames > max_frames or float(frames) * width * height * spp > max_samples:
        out["parity_note"] = ("run of %d frames = %.3g camera samples: the oracle leg is bounded to %d frames and %.3g samples "
                              "(--parity-frames / --parity-samples raise it)" % (frames, float(frames) * width * height * spp, max_frames, max_samples))
        return out
    t0 = time.perf_counter()
    ro, bo = Oracle.get().render_frames(width, height, spp, frames, seed_mode=SEED_PER_PIXEL)
    out.update(parity_checked=True, parity_ok=bool(img.tobytes() == bo.tobytes() and int(rays_total) == int(ro)),
               oracle_fnv="%08x" % fnv1a(bo), oracle_rays=int(ro), run_rays=int(rays_total), parity_frames=frames,
               parity_seconds=time.perf_counter() - t0,
               parity_note="final image + ray total of this run (frames 0..%d, zeroed tile) vs oracle/tpt_oracle.c, PER_PIXEL seeds, bytes equal" % (frames - 1))
    return out


def drain_lookahead(api):
    """Frames the library traced ahead of a synchronous caller are dropped and the GPU is idle: a secondary leg neither inherits
    speculative work from the one before it nor leaves its own unfinished behind its end time."""
    api.set_host_lookahead(2)  # (the default; setting it drops what was traced ahead)
    api.synchronize()


def drawtest_host_path(api, width, height, frames=24):
    """The reference's own contract: synchronous DrawTest on a HOST backbuffer (upload + trace + blend + download)."""
    bb = np.zeros((height, width, 4), np.float32)
    for f in range(4):
        api.UpdateTest(0.0, f, width, height, FLAG_PROGRESSIVE)
        api.DrawTest(0.0, f, width, height, bb, FLAG_PROGRESSIVE)
    drain_lookahead(api)  # the timed frames start with nothing traced ahead (the first one pays a full trace) ...
    rays, t0 = 0, time.perf_counter()
    for f in range(4, 4 + frames):
        api.UpdateTest(0.0, f, width, height, FLAG_PROGRESSIVE)
        rays += api.DrawTest(0.0, f, width, height, bb, FLAG_PROGRESSIVE)
    drain_lookahead(api)  # ... and the frames still being traced ahead of the last call are waited for inside the timed region
    dt = time.perf_counter() - t0
    return dt / frames * 1e3, rays / dt / 1e6


def batched_rate(api, torch, width, height, per_launch=8, launches=25):
    """tptDrawDeviceBatch: `per_launch` frames of the static scene per launch (same bits as one launch per frame)."""
    tile = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")  # (default-stream fill: the library's own stream is ordered behind it)
    api.UpdateTest(0.0, 0, width, height, FLAG_PROGRESSIVE)
    f = 0
    for _ in range(16):
        api.draw_device_batch(0.0, f, per_launch, width, height, tile.data_ptr(), FLAG_PROGRESSIVE)
        f += per_launch
    r0 = api.ray_counter_read()  # synchronises
    t0 = time.perf_counter()
    for _ in range(launches):
        api.draw_device_batch(0.0, f, per_launch, width, height, tile.data_ptr(), FLAG_PROGRESSIVE)
        f += per_launch
    rays = api.ray_counter_read() - r0
    dt = time.perf_counter() - t0
    return dt / (launches * per_launch) * 1e3, rays / dt / 1e6


def sync_caller_rate(api, torch, width, height, frames=60):
    """tptDrawDevice on a device tile with a synchronise after every frame: the reference's synchronous DrawTest contract
    without the PCIe copies (the library traces the next frames ahead of such a caller)."""
    tile = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")  # (default-stream fill: the library's own stream is ordered behind it)
    for f in range(8):
        api.UpdateTest(0.0, f, width, height, FLAG_PROGRESSIVE)
        api.draw_device(0.0, f, width, height, tile.data_ptr(), FLAG_PROGRESSIVE)
        api.synchronize()
    drain_lookahead(api)
    r0 = api.ray_counter_read()
    t0 = time.perf_counter()
    for f in range(8, 8 + frames):
        api.UpdateTest(0.0, f, width, height, FLAG_PROGRESSIVE)
        api.draw_device(0.0, f, width, height, tile.data_ptr(), FLAG_PROGRESSIVE)
        api.synchronize()
    drain_lookahead(api)
    dt = time.perf_counter() - t0
    rays = api.ray_counter_read() - r0
    return dt / frames * 1e3, rays / dt / 1e6


def row_serial_rate(api, width, height, frames=96):
    """ROW_SERIAL seeds through the reference's own contract: synchronous DrawTest(host buffer), frame by frame, the reference's
    exact image (one RNG stream per row; the library traces the next 32 frames of a static scene ahead as one launch)."""
    api.set_seed_mode(0)
    bb = np.zeros((height, width, 4), np.float32)
    for f in range(3):  # untimed: the library launches its 32-frame batches for a caller that has shown three consecutive frames
        api.UpdateTest(0.0, f, width, height, FLAG_PROGRESSIVE)
        api.DrawTest(0.0, f, width, height, bb, FLAG_PROGRESSIVE)
    drain_lookahead(api)
    rays, t0 = 0, time.perf_counter()
    for f in range(3, 3 + frames):
        api.UpdateTest(0.0, f, width, height, FLAG_PROGRESSIVE)
        rays += api.DrawTest(0.0, f, width, height, bb, FLAG_PROGRESSIVE)
    drain_lookahead(api)
    dt = time.perf_counter() - t0
    api.set_seed_mode(1)
    return dt / frames * 1e3, rays / dt / 1e6


def row_serial_batched_rate(api, torch, width, height, per_launch=32, launches=8):
    """ROW_SERIAL seeds through tptDrawDeviceBatch: per_launch frames x rows lanes per launch (rows AND frames are independent
    RNG streams in the reference, Test.cpp:280) -- the reference's exact image, bit for bit, at GPU speed."""
    api.set_seed_mode(0)
    tile = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")  # (default-stream fill: the library's own stream is ordered behind it)
    api.UpdateTest(0.0, 0, width, height, FLAG_PROGRESSIVE)
    api.draw_device_batch(0.0, 0, per_launch, width, height, tile.data_ptr(), FLAG_PROGRESSIVE)
    r0 = api.ray_counter_read()  # synchronises
    t0 = time.perf_counter()
    f = per_launch
    for _ in range(launches):
        api.draw_device_batch(0.0, f, per_launch, width, height, tile.data_ptr(), FLAG_PROGRESSIVE)
        f += per_launch
    rays = api.ray_counter_read() - r0
    dt = time.perf_counter() - t0
    api.set_seed_mode(1)
    return dt / (launches * per_launch) * 1e3, rays / dt / 1e6


def golden_case(section, **key):
    """A committed golden vector of tests/golden/goldens.json (per_pixel_cases: made by the reference-compiled per-pixel build,
    oracle/build_ref.sh PERPIXEL=1; oracle_cases: made by oracle/tpt_oracle.c for the scene the reference does not have)."""
    path = os.path.join(ROOT, "tests", "golden", "goldens.json")
    if not os.path.exists(path):
        return None
    for c in json.load(open(path)).get(section, []):
        if all(c.get(k) == v for k, v in key.items()):
            return c
    return None


def static_counters(workload, grid_blocks):
    """HBM bytes and executed VALU wave-instructions per trace launch for this workload at this launch geometry, from
    profiles/pmc_traffic.json (rocprofv3 --pmc passes serialise kernels: never taken inside a timed region)."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(tpath):
        return None
    return json.load(open(tpath)).get("by_workload_and_grid", {}).get(workload, {}).get(str(grid_blocks))


def secondary_leg(api, torch, name, untimed, steps, overlap):
    """A short fenced leg of another BASELINE.json config behind the headline run: frames 0 .. untimed + steps - 1 of workload `name`
    blended into one zeroed device tile, the last `steps` of them timed (barrier-less at N = 1: synchronise on both sides, HIP
    events around the region and around every trace launch), the final image hash and the ray total compared with a COMMITTED
    golden vector -- reference-compiled for the default scene, the oracle's for the 4096-sphere scene (whose frames 0-2 are
    checked one per tile before the timed frames).  -> dict for the JSON line's "secondary" object."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import fnv1a
    width, height, spp, scene, label = WORKLOADS[name]
    api.set_scene(None)
    api.set_camera(None)
    n_spheres = 46
    if scene == "stress":
        from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene
        s, m = stress_scene(4096, 64)
        api.set_scene(s, m)
        api.set_camera(**STRESS_CAMERA)
        n_spheres = 4096
    api.set_samples_per_pixel(spp)
    api.set_frame_overlap(overlap)
    out = {"workload": label, "steps": steps, "untimed_frames": untimed, "frame_overlap": overlap}
    pre = []
    if scene == "stress":  # parity first: frames 0, 1, 2, each into its own zeroed tile, in flight together -- the oracle's hashes
        tiles = [torch.zeros((height, width, 4), dtype=torch.float32, device="cuda") for _ in range(3)]
        r0 = api.ray_counter_read()
        for f in range(3):
            api.UpdateTest(0.0, f, width, height, FLAG_PROGRESSIVE)
            api.draw_device(0.0, f, width, height, tiles[f].data_ptr(), FLAG_PROGRESSIVE)
        rays3 = api.ray_counter_read() - r0
        want = [golden_case("oracle_cases", name="c5", frame=f) for f in range(3)]
        got = ["%08x" % fnv1a(t.cpu().numpy()) for t in tiles]
        pre = got
        if all(want):
            out.update(parity_checked=True, parity_ok=bool(got == [c["fnv"] for c in want] and rays3 == sum(c["rays"] for c in want)),
                       parity_source="tests/golden/goldens.json oracle_cases (oracle/tpt_oracle.c, brute force over the 4096 spheres; the reference has no such scene): "
                                     "frames 0-2, one zeroed tile each, three in flight", parity_image_fnv=got, parity_rays=int(rays3))
        else:
            out.update(parity_checked=False, parity_note="no committed hash for this workload")
        del tiles
    tile = torch.zeros((height, width, 4), dtype=torch.float32, device="cuda")
    ra = api.ray_counter_read()
    for f in range(untimed):
        api.UpdateTest(0.0, f, width, height, FLAG_PROGRESSIVE)
        api.draw_device(0.0, f, width, height, tile.data_ptr(), FLAG_PROGRESSIVE)
    api.synchronize()
    torch.cuda.synchronize()
    r0 = api.ray_counter_read()
    api.kernel_timing_begin(steps)
    api.timer_begin()
    t0 = time.perf_counter()
    for f in range(untimed, untimed + steps):
        api.UpdateTest(0.0, f, width, height, FLAG_PROGRESSIVE)
        api.draw_device(0.0, f, width, height, tile.data_ptr(), FLAG_PROGRESSIVE)
    pipeline_ms = api.timer_end()
    api.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    launch_ms_sum, launches = api.kernel_timing_end()
    rays = api.ray_counter_read() - r0
    rays_all = api.ray_counter_read() - ra
    info = api.launch_info()
    sinfo = api.scene_info()
    k_ms = launch_ms_sum / max(launches, 1)
    fpl = steps / max(launches, 1)            # frames per launch (the library batches small frames of a stream)
    p_ms = pipeline_ms / steps
    img = np.ascontiguousarray(tile.cpu().numpy(), np.float32)
    out.update(value=rays / dt / 1e6, unit="Mray/s", ms_per_step=dt / steps * 1e3, pipeline_ms_per_step=p_ms, trace_launch_ms_avg=k_ms,
               frames_per_launch=fpl, rays_per_step=rays / steps, image_fnv="%08x" % fnv1a(img), grid_blocks=info["grid_blocks"],
               blocks_per_cu=info["blocks_per_cu"], groups=sinfo["groups"], bounds_on_matrix_cores=sinfo["bounds_on_matrix_cores"])
    if scene == "default":
        c = golden_case("per_pixel_cases", width=width, height=height, spp=spp, frames=untimed + steps, flags=FLAG_PROGRESSIVE)
        if c:
            out.update(parity_checked=True, parity_ok=bool(out["image_fnv"] == c["fnv"] and int(rays_all) == c["rays"]), golden_fnv=c["fnv"], golden_rays=c["rays"], run_rays=int(rays_all),
                       parity_source="tests/golden/goldens.json per_pixel_cases: the reference's scalar CPU path compiled from /root/reference with its own GPU seed "
                                     "formula injected (oracle/build_ref.sh PERPIXEL=1), frames 0-%d on a zeroed tile" % (untimed + steps - 1))
        else:
            out.update(parity_checked=False, parity_note="no committed hash for %d frames of this workload" % (untimed + steps))
    # rooflines, per launch and for the chip (see the headline's objects): HBM write (north_star), then what binds
    px = width * height * fpl
    hbm = px * 16 / (p_ms * fpl * 1e-3) / 1e9
    ent = static_counters(name, info["grid_blocks"]) if fpl == 1 else None
    out["roofline"] = {"bound": "hbm", "achieved": hbm, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": hbm / PEAK_HBM_GBS,
                       "traffic": ent["bytes_per_launch"] if ent else None, "traffic_static": True, "traffic_source": ent["source"] if ent else None,
                       "launch_ms_avg": k_ms, "launches": launches, "kernel": "tptTraceQueueKernel"}
    if sinfo["groups"] > 0:
        vi = ent.get("valu_insts_per_launch") if ent else None
        g = vi / (p_ms * fpl * 1e-3) / 1e9 if vi else None
        out["roofline_valu"] = {"bound": "valu_issue", "achieved": g, "peak": PEAK_VALU_GINSTR, "unit": "G wave-instr/s", "frac": g / PEAK_VALU_GINSTR if g else None,
                                "insts_per_launch": vi, "static": True, "source": ent.get("valu_source") if ent else None,
                                "brute_force_equivalent_tflops": rays / steps * FLOP_PER_SPHERE_TEST * n_spheres / (p_ms * 1e-3) / 1e12}
    else:
        tf = rays / steps * FLOP_PER_SPHERE_TEST * n_spheres / (p_ms * 1e-3) / 1e12
        out["roofline_valu"] = {"bound": "valu_fp32", "achieved": tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP32_TFLOPS}
        if ent and ent.get("valu_insts_per_launch"):
            g = ent["valu_insts_per_launch"] / (p_ms * fpl * 1e-3) / 1e9
            out["valu_issue"] = {"insts_per_launch": ent["valu_insts_per_launch"], "achieved": g, "peak": PEAK_VALU_GINSTR, "unit": "G wave-instr/s", "frac": g / PEAK_VALU_GINSTR, "static": True}
    del tile
    return out



def self_launch(n):
    """`python bench.py --gpus N` without a launcher: this command once per GPU of this node, with the environment
    torch.distributed.run would give each rank (RANK / LOCAL_RANK / WORLD_SIZE, rendezvous on 127.0.0.1 at a free port).
    Rank 0 prints the JSON line.  A rank that fails takes the others down; -> the first non-zero exit code."""
    import socket
    import subprocess
    import time as _time
    rc = 0
    for attempt in range(3):  # the rendezvous port is found by binding and closing a socket: another process may take it in between
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), OMP_NUM_THREADS=os.environ.get("OMP_NUM_THREADS", "1"))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes needs it on this driver
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
        rc, failed_at = 0, None
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc, failed_at = code, _time.monotonic()
            # the others would wait for the failed rank in a collective for ever -- but give them the time to say what THEY found first
            # (a rank still importing torch on a busy host has not printed its own "needs a GPU" yet)
            if failed_at is not None and _time.monotonic() - failed_at > 60.0:
                for q in procs:
                    q.terminate()
                failed_at = float("inf")
            _time.sleep(0.05)
        if rc != PORT_TAKEN_RC:
            return rc
        print("bench.py: rendezvous port %d was taken by another process, trying another one (%d / 3)" % (port, attempt + 1), file=sys.stderr, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--stripe-rows", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-pointer DrawTest and ROW_SERIAL legs after the timed region")
    ap.add_argument("--extras", default="host,sync,batched,row_serial", help="which secondary legs run after the timed region (comma list of host, sync, batched, row_serial)")
    ap.add_argument("--secondary", default="auto",
                    help="short fenced legs of the other configs behind the headline run, each with its own roofline objects and a parity flag "
                         "against a committed golden hash (comma list of c2_steady, c3, c5; auto = all three when the headline is the default "
                         "1-GPU configs[1] run, none otherwise; none = skip)")
    ap.add_argument("--hit-spheres", type=int, default=0, help="0 two-phase: matrix-core filter for <= 64 spheres, grouped traversal for >= 256 (default); 1 simple loop; 2 two-phase brute force; 3 as 0 with the flat packed VALU filter everywhere (no matrix cores, no second level over the groups); 4 as 0 with a grouped scene's bounds on the matrix cores (opt-in: not in a time-sliced process, DESIGN.md 2.2)")
    ap.add_argument("--persistent", type=int, default=3, choices=[0, 1, 3], help="3 path queues (default) 1 persistent waves with lane refill (the fallback kernel) 0 one thread per pixel (the lane-refill kernel with re-filling off: the north_star's shape, for A/B runs)")
    ap.add_argument("--fold", type=int, default=0, help="0 recursive (reference order, default) 1 forward")
    ap.add_argument("--lds-scene", type=int, default=-1)
    ap.add_argument("--batch", type=int, default=1,
                    help="frames per launch (tptDrawDeviceBatch; sharded: also per exchange).  1 = one launch and one exchange per frame, the "
                         "reference's own contract and the default; prime / warmup / steps must be multiples of it")
    ap.add_argument("--animate", action="store_true",
                    help="kFlagAnimate: spheres 1 and 8 move every frame (time = frame/60 s), the scene is re-uploaded per frame")
    ap.add_argument("--overlap", type=int, default=0,
                    help="trace kernels of up to this many consecutive frames may be in flight (0 = auto: 16, or 8 when the frame is "
                         "sharded over more than 2 ranks -- with small tiles the per-packet latency of many active queues costs more "
                         "than the extra overlap buys)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "none", "cabi"],
                    help="how a sharded frame is exchanged: cabi = the library's own RCCL gather (tptCommInit / tptDrawSharded / "
                         "tptShardedFinish, what a C++ Test.h host uses; default for N > 1, and at N = 1 a one-rank communicator so that the "
                         "gather path really runs), none = no exchange (default for N = 1: plain tptDrawDevice on a device tile)")
    ap.add_argument("--parity-frames", type=int, default=64, help="check the final image against the oracle when the run has at most this many frames (0 = never)")
    ap.add_argument("--parity-samples", type=float, default=1.6e8,
                    help="... and at most this many camera samples in total (frames x width x height x spp): bounds the oracle leg to ~10 s "
                         "of the host's cores -- 41 frames of configs[1], ONE frame of configs[2] (--workload c3 --prime 0 --warmup 0 --steps 1)")
    ap.add_argument("--prime", type=int, default=-1,
                    help="untimed frames rendered BEFORE the warm-up so that the frame pipeline (buffers of all slots, the library's "
                         "estimate of how deep this caller pipelines) is in its steady state when warm-up and timing start; "
                         "-1 = as many as frames may be in flight; reported as config.untimed_priming_frames")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        foreign = [v for v in FOREIGN_LAUNCHER_VARS if v in os.environ]
        if foreign:
            sys.exit("bench.py: started under another launcher (%s is set) without RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment: "
                     "export those per rank (as torch.distributed.run does) instead of letting every rank spawn %d more" % (foreign[0], args.gpus))
        # typed the way the driver types its 1-GPU run (`python3 bench.py --gpus N ...`): launch the ranks ourselves
        sys.exit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world  # (the launcher's world size is what runs)
    who = "bench.py rank %d of %d" % (rank, world)
    if not torch.cuda.is_available():
        sys.exit("%s: needs a GPU (the product has no CPU path)" % who)
    if torch.cuda.device_count() < world:
        sys.exit("%s: %d GPUs needed, %d visible" % (who, world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    os.environ.setdefault("TPT_DEVICE", str(local_rank))
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        except Exception as e:  # noqa: BLE001 -- rank 0 could not bind the rendezvous port: tell self_launch to pick another one
            if "EADDRINUSE" in str(e) or "address already in use" in str(e).lower():
                print("%s: %s" % (who, e), file=sys.stderr)
                sys.exit(PORT_TAKEN_RC)
            raise

    from toypathtracer_amd import api

    width, height, spp, scene, label = WORKLOADS[args.workload]
    api.InitializeTest()
    api.set_samples_per_pixel(spp)
    api.set_fold_mode(args.fold)
    api.set_kernel_variant(args.hit_spheres, args.persistent, args.lds_scene)
    if args.overlap <= 0:
        args.overlap = 16 if world <= 2 else 8
    api.set_frame_overlap(args.overlap)
    n_spheres = 46
    if scene == "stress":
        from toypathtracer_amd.scenes import STRESS_CAMERA, stress_scene
        s, m = stress_scene(4096, 64)
        api.set_scene(s, m)
        api.set_camera(**STRESS_CAMERA)
        n_spheres = 4096
    exchange = args.exchange
    if exchange == "auto":
        exchange = "cabi" if world > 1 else "none"
    if exchange == "none" and world > 1:
        sys.exit("--exchange none needs --gpus 1")
    flags = FLAG_PROGRESSIVE | (FLAG_ANIMATE if args.animate else 0)
    rccl_ranks = 0
    image_on_root = None
    if exchange == "cabi":
        # the product's own multi-GPU path: RCCL inside the library, nothing of torch.distributed in the data path
        uid = [api.comm_get_unique_id() if rank == 0 else None]
        if dist is not None:
            dist.broadcast_object_list(uid, src=0, device=device)
        api.comm_init(uid[0], world, rank, args.stripe_rows)
        rccl_ranks, rccl_rank, _lb = api.comm_info()  # what RCCL itself says (ncclCommCount / ncclCommUserRank), not the arguments above
        assert (rccl_ranks, rccl_rank) == (world, rank), ("communicator of %d ranks, this is rank %d; expected %d / %d" % (rccl_ranks, rccl_rank, world, rank))
        image_on_root = torch.zeros((height, width, 4), dtype=torch.float32, device=device)  # (filled on torch's default stream: the library's ordered streams wait for it)
        img_ptr = image_on_root.data_ptr()

        def step(frame):
            t = frame / 60.0 if args.animate else 0.0
            api.UpdateTest(t, frame, width, height, flags)
            if args.batch > 1:
                api.draw_sharded_batch(t, frame, args.batch, width, height, img_ptr, flags)
            else:
                api.draw_sharded(t, frame, width, height, img_ptr, flags)

        def rays_so_far():  # rank 0: sum over the ranks as of the last gathered frame; others: their own (both exact after a drain)
            return api.sharded_finish()

        def fence():
            api.sharded_finish()
            api.synchronize()
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
    else:
        # one GPU, no exchange: the product's plain device path -- tptDrawDevice on a tile that stays resident in HBM, the library's own
        # (blocking) stream and ray counter
        api.set_row_shard(0, 1, 0)
        tile = torch.zeros((height, width, 4), dtype=torch.float32, device=device)  # (filled on torch's default stream: the library's stream is ordered behind it)
        tile_ptr = tile.data_ptr()

        def step(frame):
            t = frame / 60.0 if args.animate else 0.0
            api.UpdateTest(t, frame, width, height, flags)
            if args.batch > 1:
                api.draw_device_batch(t, frame, args.batch, width, height, tile_ptr, flags)
            else:
                api.draw_device(t, frame, width, height, tile_ptr, flags)

        def rays_so_far():
            return api.ray_counter_read()

        def fence():
            api.synchronize()
            torch.cuda.synchronize()

    if args.prime < 0:
        args.prime = args.overlap
    B = args.batch
    if B > 1:
        if args.animate:
            sys.exit("--batch needs a static scene")
        args.prime, args.warmup = -(-args.prime // B) * B, -(-args.warmup // B) * B
        if args.steps % B:
            sys.exit("--steps must be a multiple of --batch")
    for f in range(0, args.prime, B):  # untimed, not part of --warmup either (stated in the JSON line)
        step(f)
    fence()
    for f in range(args.prime, args.prime + args.warmup, B):
        step(f)
    fence()
    rays0 = rays_so_far()
    api.kernel_timing_begin(args.steps // B)  # a HIP event pair around every trace launch, on the stream it is launched on
    api.timer_begin()                     # + one pair around the whole timed region on the context's stream
    t0 = time.perf_counter()
    for f in range(args.prime + args.warmup, args.prime + args.warmup + args.steps, B):
        step(f)
    pipeline_ms = api.timer_end()         # records + synchronises the end event on the render stream
    fence()
    dt = time.perf_counter() - t0
    launch_ms_sum, launches = api.kernel_timing_end()
    kernel_ms = launch_ms_sum / max(launches, 1) * args.steps  # = steps x average duration of one trace launch (a launch of --batch frames counts once per frame: trace_launch_ms_avg is per LAUNCH)
    frames_per_launch = args.steps / max(launches, 1)          # measured: --batch, or the library's stream batching of small frames (tptSetStreamBatching)
    rays_end = rays_so_far()
    rays_local = rays_end - rays0
    if exchange == "cabi":
        image = image_on_root
        rays_all_frames = rays_end  # rank 0: every rank's rays since tptInitialize
        if rank != 0:
            rays_local = 0  # rank 0's difference already is the sum over the ranks (the counters ride in the gathered tiles)
    else:
        image = tile
        rays_all_frames = rays_end

    stats = torch.tensor([dt, float(rays_local), kernel_ms, pipeline_ms], dtype=torch.float64, device=device)
    if dist is not None:
        tmax = stats.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = stats.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, rays_total, kernel_ms, pipeline_ms = float(tmax[0]), float(tsum[1]), float(tmax[2]), float(tmax[3])
        if rays_all_frames is None:
            tot = torch.tensor([float(rays_end)], dtype=torch.float64, device=device)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            rays_all_frames = int(tot[0])
    else:
        rays_total = float(rays_local)
        rays_all_frames = rays_end

    if rank == 0:
        info = api.launch_info()
        k_ms = kernel_ms / args.steps                      # average duration of one trace launch (this rank's rows),
        #                                                    what a rocprofv3 kernel trace reports per dispatch; with
        #                                                    frame overlap two launches share the GPU, so k_ms ~ 2 x the
        #                                                    pipeline time per frame (p_ms)
        p_ms = pipeline_ms / args.steps
        pl_ms = p_ms * frames_per_launch                   # pipeline time per LAUNCH (= per frame unless several frames share a launch)
        px = width * height / world * frames_per_launch     # pixels one launch of one rank covers (all frames of a batched launch)
        rays_per_launch = rays_total / args.steps / world * frames_per_launch
        hbm_write_gbs = px * 16 / (k_ms * 1e-3) / 1e9      # 16 B written per pixel
        valu_tflops = rays_per_launch * FLOP_PER_SPHERE_TEST * n_spheres / (k_ms * 1e-3) / 1e12
        # HBM traffic per trace launch: counter passes serialise kernels and cannot run inside the timed region, so the
        # figure is a STATIC one from profiles/pmc_traffic.json -- and only the entry measured for this workload at this
        # very launch geometry (workgroups per launch); no entry -> null, never a neighbour's number.
        traffic, traffic_src = None, "profiles/pmc_traffic.json has no entry for (%s, %d workgroups per launch)" % (args.workload, info["grid_blocks"])
        valu_insts, valu_src = None, traffic_src
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and args.persistent == 3 and frames_per_launch == 1 and world == 1:
            ent = json.load(open(tpath)).get("by_workload_and_grid", {}).get(args.workload, {}).get(str(info["grid_blocks"]))
            if ent:
                traffic, traffic_src = ent["bytes_per_launch"], ent["source"]
                if ent.get("valu_insts_per_launch") and args.hit_spheres == 0:
                    valu_insts, valu_src = ent["valu_insts_per_launch"], ent.get("valu_source", ent["source"])
        # EXECUTED vector work (SQ_INSTS_VALU wave-instructions per launch, a static figure like the traffic) over the pipeline time
        # per frame, against one wave64 VALU instruction per SIMD every 2 cycles (MI355X_MICROARCH.md): what the VALU pipes really do
        grouped = api.scene_info()["groups"] > 0  # (what the launch really does: a scene the grouping refuses is traversed flat and keeps the flop roofline)
        issue = None
        if valu_insts:
            g_instr_s = valu_insts / (pl_ms * 1e-3) / 1e9
            issue = {"insts_per_launch": valu_insts, "achieved": g_instr_s, "peak": PEAK_VALU_GINSTR, "unit": "G wave-instr/s", "frac": g_instr_s / PEAK_VALU_GINSTR,
                     "static": True, "source": valu_src,
                     "note": "SQ_INSTS_VALU per trace launch (rocprofv3 --pmc, kernels serialised) over the pipeline time per launch; peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles"}
        out = {
            "metric": "Mray/s", "value": rays_total / dt / 1e6, "unit": "Mray/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "exchange": exchange, "rccl_ranks": rccl_ranks if exchange == "cabi" else 0,
            "config": {"workload": label, "width": width, "height": height, "spp": spp, "spheres": n_spheres,
                       "seed_mode": "per_pixel", "fold": "forward" if args.fold else "recursive",
                       "hit_spheres": ["two_phase" + ("+groups+two_level_valu_bounds" if n_spheres >= 256 else "+matrix_core_filter" if n_spheres <= 64 else ""), "simple", "two_phase_brute_force", "two_phase_flat_valu_filter",
                                       "two_phase+groups+matrix_core_bounds"][args.hit_spheres], "kernel": {0: "thread_per_pixel", 1: "persistent_waves", 3: "path_queues"}[args.persistent], "frame_overlap": args.overlap, "frames_per_launch": frames_per_launch,
                       "untimed_priming_frames": args.prime,
                       "flags": "progressive|animate" if args.animate else "progressive",
                       "sharding": "none" if world == 1 else "row stripes of %d, round-robin over %d ranks, pipelined gather to rank 0" % (args.stripe_rows, world),
                       "device": api.device_name(), "grid_blocks": info["grid_blocks"], "blocks_per_cu": info["blocks_per_cu"],
                       "lds_bytes_per_block": info["lds_bytes"],
                       # grouped scenes: whether the groups' bounds are filtered on the matrix cores (opt-in, --hit-spheres 4: DESIGN.md 2.2;
                       # the default is the two-level packed VALU filter) -- and the queues this process asked for
                       "groups": api.scene_info()["groups"], "bounds_on_matrix_cores": api.scene_info()["bounds_on_matrix_cores"],
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES")},
            "rays_per_step": rays_total / args.steps,
            "trace_launch_ms_avg": k_ms,
            "pipeline_ms_per_step": p_ms,
            "pipeline_Mray_s": rays_per_launch / frames_per_launch * world / (p_ms * 1e-3) / 1e6,
            # roofline.frac is the chip-level figure: the frame's algorithmic bytes over the time a frame occupies the
            # pipeline (ms_per_step measured by HIP events on the render stream) -- up to `frame_overlap` launches share
            # the GPU, so bytes / one launch's own duration (frac_per_launch) understates the chip by that factor
            "roofline": {"bound": "hbm", "achieved": hbm_write_gbs * k_ms / pl_ms, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": hbm_write_gbs * k_ms / pl_ms / PEAK_HBM_GBS, "traffic": traffic,
                         "traffic_static": True, "traffic_source": traffic_src,
                         "traffic_note": ("NOT measured in this run: bytes per trace launch on the L2's fabric side (WRITE_SIZE + 2 x FETCH_SIZE, "
                                          "separate rocprofv3 --pmc passes, tools/traffic.sh) taken for this workload at THIS launch geometry "
                                          "(config.grid_blocks workgroups per launch); null when no such measurement is on file"),
                         "achieved_per_launch": hbm_write_gbs, "frac_per_launch": hbm_write_gbs / PEAK_HBM_GBS,
                         "achieved_read_plus_write": 2 * hbm_write_gbs * k_ms / pl_ms, "launch_ms_avg": k_ms, "launches": launches,
                         "note": "north_star's HBM-write roofline (W*H*16 B per frame).  The kernel is FP32-VALU bound (arithmetic "
                                 "intensity ~440 flop/B against a machine balance of ~20): see roofline_valu, the binding one",
                         "kernel": {0: "tptTraceKernel", 1: "tptTraceKernel", 3: "tptTraceQueueKernel"}[args.persistent]},
            "roofline_valu": ({"bound": "valu_fp32", "achieved": valu_tflops * k_ms / pl_ms, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                               "frac": valu_tflops * k_ms / pl_ms / PEAK_FP32_TFLOPS,
                               "achieved_per_launch": valu_tflops, "frac_per_launch": valu_tflops / PEAK_FP32_TFLOPS,
                               "note": "algorithmic flops = rays x 17 flop x spheres (SURVEY 8d: the reference's own per-test count) over the "
                                       "pipeline time per frame; peak counts FMA as 2 flop.  The exact arithmetic (phase 2 of HitSpheres, Scatter) "
                                       "may not contract to FMA (parity); phase 1 is a conservative filter"} if not grouped else
                              # a scene traversed through sphere groups never executes most of the 17 x N flops: a fraction of the FP32 peak
                              # would exceed 1 and say nothing.  What the kernel EXECUTES is reported instead (null without a measurement).
                              {"bound": "valu_issue", "achieved": issue["achieved"] if issue else None, "peak": PEAK_VALU_GINSTR, "unit": "G wave-instr/s",
                               "frac": issue["frac"] if issue else None,
                               "brute_force_equivalent_tflops": valu_tflops * k_ms / pl_ms,
                               "note": "grouped traversal (>= 256 spheres): executed VALU wave-instructions per second against the issue rate of the "
                                       "SIMDs; the brute-force-equivalent flop rate (rays x 17 x spheres, tests that are never executed) is kept "
                                       "beside it for comparison with the flat kernels and is not a roofline fraction"}),
        }
        if issue:
            out["valu_issue"] = issue
        total_frames = args.prime + args.warmup + args.steps
        if scene == "default" and not args.animate and args.parity_frames > 0:
            out.update(image_parity(image, rays_all_frames, width, height, spp, total_frames, args.parity_frames, args.parity_samples))
        else:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            from oracle_lib import fnv1a
            out.update(image_fnv="%08x" % fnv1a(np.ascontiguousarray(image.detach().cpu().numpy(), np.float32)), parity_checked=False,
                       parity_note="oracle leg runs for the static default scene only (this workload's parity: tests/test_gpu_parity.py); image_fnv lets two runs of the same frames be compared")
        if scene == "default" and not args.animate and world == 1:
            # ... and against the REFERENCE ITSELF where a golden vector for exactly these frames is committed (41 = the driver's
            # command, 236 = the default command): made by the reference's scalar CPU path compiled with its own GPU seed formula
            c = golden_case("per_pixel_cases", width=width, height=height, spp=spp, frames=total_frames, flags=FLAG_PROGRESSIVE)
            if c:
                out["reference_golden"] = {"fnv": c["fnv"], "rays": c["rays"], "ok": bool(out["image_fnv"] == c["fnv"] and int(rays_all_frames) == c["rays"]),
                                           "source": "tests/golden/goldens.json per_pixel_cases (oracle/_ref/libtpt_ref_perpixel.so: Test.cpp + Maths.cpp compiled from "
                                                     "/root/reference, ComputeShader.hlsl:380's seed injected at Test.cpp:281), frames 0-%d" % (total_frames - 1)}
        extras = set() if args.no_extras else set(x for x in args.extras.split(",") if x)
        if world == 1 and exchange != "cabi" and extras:
            # the same workload through the reference's own contract (host backbuffer, synchronous) and in its own seed mode
            api.set_ray_counter(None)
            api.set_stream(None)
            api.set_tile_mirror(None)
            api.set_row_shard(0, 1, 0)
            if "host" in extras:
                ms, mr = drawtest_host_path(api, width, height)
                out["drawtest_host_ms"], out["drawtest_host_Mray_s"] = ms, mr
                out["drawtest_host_note"] = ("synchronous DrawTest(host float* backbuffer) per frame, the reference's own calling contract: backbuffer "
                                             "upload + blend + download over PCIe in every call (default host-buffer mode), the next two frames "
                                             "traced ahead of the caller (tptSetHostLookahead); never the headline value")
            if "sync" in extras and not args.animate and width * height <= 1280 * 720:
                ms, mr = sync_caller_rate(api, torch, width, height)
                out["sync_device_caller_ms"], out["sync_device_caller_Mray_s"] = ms, mr
                out["sync_device_caller_note"] = ("tptDrawDevice + a synchronise after EVERY frame (device tile, no PCIe): the next two frames are "
                                                  "traced ahead of such a caller; never the headline value")
            if "batched" in extras and args.persistent == 3 and args.hit_spheres != 1 and not args.animate and width * height <= 1280 * 720:
                for k in (4, 8):
                    ms, mr = batched_rate(api, torch, width, height, per_launch=k, launches=max(4, 200 // k))
                    out["batched_%d_ms_per_frame" % k], out["batched_%d_Mray_s" % k] = ms, mr
                out["batched_note"] = ("tptDrawDeviceBatch: k frames of the static scene per launch, blended in order by one more -- the same bits as "
                                       "k tptDrawDevice calls; a secondary figure, the headline value is one launch per frame")
            if "row_serial" in extras and scene == "default" and width * height <= 1280 * 720:
                ms, mr = row_serial_rate(api, width, height)
                out["row_serial_ms"], out["row_serial_Mray_s"] = ms, mr
                out["row_serial_note"] = ("seed mode 0: the reference's per-row RNG streams (bit-identical CPU image) through synchronous DrawTest(host buffer), frame by "
                                          "frame; the library traces the next 32 frames of the static scene ahead as one launch (rows x frames lanes)")
                ms, mr = row_serial_batched_rate(api, torch, width, height)
                out["row_serial_batched_32_ms_per_frame"], out["row_serial_batched_32_Mray_s"] = ms, mr
                out["row_serial_batched_note"] = ("seed mode 0 through tptDrawDeviceBatch: 32 frames x rows lanes per launch on a device tile -- the reference's exact "
                                                  "image (golden hashes 609aacda / 46afd557 reproduced by tests/test_gpu_parity.py) at GPU speed")
        sec = args.secondary
        if sec == "auto":
            sec = "c2_steady,c3,c5" if (world == 1 and exchange == "none" and args.workload == "c2" and args.persistent == 3 and args.hit_spheres == 0 and args.fold == 0
                                        and args.batch == 1 and not args.animate and not args.no_extras) else "none"
        if sec != "none" and world == 1 and exchange == "none":
            # the other configs in front of whoever runs this command: C2 in steady state (200 timed frames), C3 (3840x2160x16spp) and
            # C5 (4096 spheres), each fenced, each with its own rooflines and a parity flag against a committed golden vector
            api.set_ray_counter(None)
            api.set_stream(None)
            api.set_tile_mirror(None)
            api.set_row_shard(0, 1, 0)
            api.set_seed_mode(1)
            out["secondary"] = {}
            plan = {"c2_steady": ("c2", 36, 200, 16), "c3": ("c3", 4, 9, 16), "c5": ("c5", 16, 16, 16)}
            for leg in [x for x in sec.split(",") if x]:
                wl, untimed, steps, ov = plan[leg]
                drain_lookahead(api)
                out["secondary"][leg] = secondary_leg(api, torch, wl, untimed, steps, ov)
            api.set_scene(None)
            api.set_camera(None)
            api.set_samples_per_pixel(spp)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(width, height, spp)
        print(json.dumps(out), flush=True)

    if exchange == "cabi":
        api.comm_destroy()
    api.set_tile_mirror(None)
    api.set_ray_counter(None)
    api.set_stream(None)
    api.ShutdownTest()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
