"""Exhaustive interleaving check of the hand-shake between a path-queue launch and the helper grids of the tail-helper experiment
(csrc/tpt_device.h; tpt_kernels.hip: the helper's prologue, the launch's last wave).  A MODEL of the protocol, not
the kernel: every shared access below is one returning atomic at the device's coherence point, in program order per actor (the kernel
gets that order from feeding each atomic's result into the next), and the explorer runs every interleaving of those steps.

    counter block of a frame slot:  W next chunk | B helper workgroups registered | C serial of the last launch that closed here
    helper(gen):      B += 1;  c = C;  closed if c >= gen -> B -= 1, leave.   else: claim chunks (k = W++ while k < N), finish them, B -= 1
    launch(gen):      its own workgroups claim chunks the same way; when the pool is dry and they are done, the last wave:
                      C = gen;  wait until B == 0;  W = 0 (re-arm for the next launch on this slot);  kernel ends

Two launches use the same block one after the other (gen 1, then gen 2 -- the second starts when the first has ended, stream order);
a helper grid waits (on the stream it is put on) for the event its launch's stream carries right before the launch -- so a helper of
launch 2 cannot start before launch 1 has ended, but it may start before launch 2 itself does; apart from that helpers of either
serial start at ANY time, including long after their launch has ended (a stale helper).  (Without that event the model fails: a
helper of launch 2 takes chunks of launch 1's pool -- the first version of this file found it.)  Checked in every
interleaving: each chunk of each launch is processed exactly once and for the right launch; no chunk is claimed from a pool that has
been re-armed for the next launch by a helper of the previous one; a launch ends only when all its chunks are finished."""
import sys

N = 2  # chunks per launch


def explore(helper_gens):
    """DFS over all interleavings.  State: shared (W, B, C), per-actor program counters and locals, bookkeeping of who processed what."""
    sys.setrecursionlimit(10000)
    # actors: 'L1', 'L2' (one worker each + the last-wave epilogue), helpers h0..; each actor is a generator-like state machine
    seen = set()
    violations = []
    ended_states = [0]

    def step(state):
        key = state
        if key in seen:
            return
        seen.add(key)
        W, B, C, l1, l2, hs, done1, done2, ended1, ended2 = state
        # done1 / done2: tuple per chunk: 0 = untouched, 1 = claimed (in progress), 2 = finished
        progressed = False

        def launch_step(gen, pc, done, ended, other_ended_ok):
            """one atomic step of launch `gen`; returns list of (W, B, C, pc, done, ended)"""
            out = []
            if ended:
                return out
            if gen == 2 and not other_ended_ok:
                return out  # stream order: launch 2 starts when launch 1 has ended
            kind = pc[0]
            if kind == "claim":            # k = W++
                k = W
                if k < N:
                    d = list(done); assert d[k] == 0, "chunk claimed twice"
                    d[k] = 1
                    out.append((W + 1, B, C, ("work", k), tuple(d), ended))
                else:
                    out.append((W + 1, B, C, ("close",), done, ended))
            elif kind == "work":           # finish the chunk (its stores), then claim again
                d = list(done); d[pc[1]] = 2
                out.append((W, B, C, ("claim",), tuple(d), ended))
            elif kind == "close":          # last wave: C = gen
                out.append((W, B, gen, ("wait",), done, ended))
            elif kind == "wait":           # read B
                if B == 0:
                    out.append((W, B, C, ("rearm",), done, ended))
                else:
                    out.append(None)       # spin: no state change (other actors must move)
            elif kind == "rearm":          # W = 0, kernel ends
                if not all(x == 2 for x in done):
                    violations.append(("launch %d ended with unfinished chunks" % gen, state))
                out.append((0, B, C, ("end",), done, True))
            return out

        # launch 1
        for nxt in launch_step(1, l1, done1, ended1, True):
            if nxt is None:
                continue
            progressed = True
            step((nxt[0], nxt[1], nxt[2], nxt[3], l2, hs, nxt[4], done2, nxt[5], ended2))
        # launch 2
        for nxt in launch_step(2, l2, done2, ended2, ended1):
            if nxt is None:
                continue
            progressed = True
            step((nxt[0], nxt[1], nxt[2], l1, nxt[3], hs, done1, nxt[4], ended1, nxt[5]))
        # helpers
        for i, (gen, pc) in enumerate(hs):
            kind = pc[0]
            nW, nB, nC, npc, nd1, nd2 = W, B, C, None, done1, done2
            if kind == "start" and gen == 2 and not ended1:
                continue                   # the helper grid waits for what its launch waits for (evPre): the slot's previous launch has ended
            if kind == "start":            # B += 1
                nB, npc = B + 1, ("look",)
            elif kind == "look":           # c = C
                npc = ("leave",) if C >= gen else ("claim",)
            elif kind == "claim":          # k = W++
                k = W
                nW = W + 1
                if k < N:
                    # which pool is this?  the block serves launch 1 until it has ended, then launch 2
                    serving = 1 if not ended1 else 2
                    if serving != gen:
                        violations.append(("helper of launch %d claimed chunk %d of launch %d" % (gen, k, serving), state))
                    d = list(done1 if serving == 1 else done2)
                    if d[k] != 0:
                        violations.append(("chunk %d of launch %d claimed twice" % (k, serving), state))
                    d[k] = 1
                    if serving == 1:
                        nd1 = tuple(d)
                    else:
                        nd2 = tuple(d)
                    npc = ("work", k, serving)
                else:
                    npc = ("leave",)
            elif kind == "work":
                d = list(done1 if pc[2] == 1 else done2); d[pc[1]] = 2
                if pc[2] == 1:
                    nd1 = tuple(d)
                else:
                    nd2 = tuple(d)
                npc = ("claim",)
            elif kind == "leave":          # B -= 1
                nB, npc = B - 1, ("gone",)
            else:
                continue
            progressed = True
            nhs = hs[:i] + ((gen, npc),) + hs[i + 1:]
            step((nW, nB, nC, l1, l2, nhs, nd1, nd2, ended1, ended2))
        if not progressed:
            if not (ended1 and ended2 and all(pc[0] == "gone" for _, pc in hs)):
                violations.append(("deadlock", state))
            ended_states[0] += 1

    hs0 = tuple((g, ("start",)) for g in helper_gens)
    step((0, 0, 0, ("claim",), ("claim",), hs0, (0,) * N, (0,) * N, False, False))
    return len(seen), ended_states[0], violations


def test_every_interleaving_of_one_launch_pair_and_its_helpers():
    for helpers in [(1,), (2,), (1, 1), (1, 2), (2, 2), (1, 1, 2)]:
        states, ends, violations = explore(helpers)
        assert ends > 0 and states > 50
        assert not violations, (helpers, violations[:3])


def test_the_model_finds_the_bug_when_the_order_is_wrong():
    """Sanity of the checker itself: a helper that looks BEFORE it registers (the wrong order) is caught."""
    global N
    import types
    src = open(__file__).read()
    broken = src.replace('            if kind == "start":            # B += 1\n                nB, npc = B + 1, ("look",)',
                         '            if kind == "start":            # (wrong order) c = C first\n                npc = ("leave0",) if C >= gen else ("reg",)')
    broken = broken.replace('elif kind == "look":           # c = C\n                npc = ("leave",) if C >= gen else ("claim",)',
                            'elif kind == "reg":\n                nB, npc = B + 1, ("claim",)\n            elif kind == "leave0":\n                npc = ("gone",)')
    assert broken != src
    mod = types.ModuleType("broken_model")
    exec(compile(broken, "broken_model", "exec"), mod.__dict__)
    _, _, violations = mod.explore((1,))
    assert violations, "the wrong order must produce a violation"
