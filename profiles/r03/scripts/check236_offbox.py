import sys, time
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from oracle_lib import SEED_PER_PIXEL, Oracle, fnv1a
t0 = time.time()
ro, bo = Oracle.get().render_frames(1280, 720, 4, 236, seed_mode=SEED_PER_PIXEL)
print("oracle 236 frames: fnv %08x rays %d (%.0f s)" % (fnv1a(bo), ro, time.time() - t0))
