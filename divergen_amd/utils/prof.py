"""Host side of libdgx's launch accounting (csrc/prof.hip): per-family HIP-event time and algorithmic FLOP / bytes of the heavy
entry points, used by bench.py for its `roofline` objects.  Launches replayed from hipGraphs carry no events; their work is
known from the capture (utils/graphs.py adds a segment's captured tally once per replay)."""
import ctypes

from .. import _lib as L

FAMILIES = {"gemm_nt": 0, "wgrad": 1, "attn_fwd": 2, "attn_bwd": 3}
# work replayed from graphs since the last enable(): family -> [launches, flops, bytes]
REPLAYED = {k: [0, 0.0, 0.0] for k in FAMILIES}
ON = False


def enable(on=True):
    """Switch the accounting on or off; either way every tally (library and replay side) restarts from zero."""
    global ON
    L.check(L.lib().dgx_prof_enable(int(on)), "dgx_prof_enable")
    for v in REPLAYED.values():
        v[0], v[1], v[2] = 0, 0.0, 0.0
    ON = bool(on)


def pause(paused=True):
    """Stop / resume the accounting without resetting it (bench.py samples every n-th step)."""
    global ON
    L.check(L.lib().dgx_prof_pause(int(paused)), "dgx_prof_pause")
    ON = not paused


def captured_snapshot():
    """(launches, flops, bytes) recorded into graphs so far, per family -- differenced around a capture by graphs.py."""
    out = {}
    for name, fid in FAMILIES.items():
        st = L.ProfStats()
        L.check(L.lib().dgx_prof_read(fid, ctypes.byref(st)), "dgx_prof_read")
        out[name] = (st.captured_launches, st.captured_flops, st.captured_bytes)
    return out


def add_replay(delta):
    for name, (n, f, b) in delta.items():
        r = REPLAYED[name]
        r[0] += n
        r[1] += f
        r[2] += b


def read():
    """family -> dict(ms, launches, flops, bytes: event-timed eager launches; graph_launches, graph_flops, graph_bytes:
    launches replayed from hipGraphs over the same period).  Waits for the recorded events."""
    out = {}
    for name, fid in FAMILIES.items():
        st = L.ProfStats()
        L.check(L.lib().dgx_prof_read(fid, ctypes.byref(st)), "dgx_prof_read")
        r = REPLAYED[name]
        out[name] = {"ms": st.ms, "launches": st.launches, "flops": st.flops, "bytes": st.bytes,
                     "graph_launches": r[0], "graph_flops": r[1], "graph_bytes": r[2]}
    return out
