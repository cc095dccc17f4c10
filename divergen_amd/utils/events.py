"""EventStorage / writers.  Surface of D2/utils/events.py:50-485 used by the training loop."""
import json
import logging
import os
import time
from collections import defaultdict
from contextlib import contextmanager

_CURRENT_STORAGE_STACK = []


def get_event_storage():
    assert len(_CURRENT_STORAGE_STACK), "get_event_storage() has to be called inside a 'with EventStorage(...)' context!"
    return _CURRENT_STORAGE_STACK[-1]


def has_event_storage():
    return len(_CURRENT_STORAGE_STACK) > 0


class DeferredScalar:
    """A statistic whose arithmetic waits for a reader: `fn` over the host values of the raw device counters in `tensors`,
    evaluated in float().  The training step then carries no launches for numbers only a writer looks at (every 20 iterations)."""
    __slots__ = ("_fn", "_t")

    def __init__(self, fn, *tensors):
        self._fn, self._t = fn, tuple(t.detach() for t in tensors)

    def __float__(self):
        return float(self._fn(*[t.tolist() for t in self._t]))


class EventStorage:
    def __init__(self, start_iter=0):
        self._history = defaultdict(list)
        self._latest = {}
        self._iter = start_iter
        self._prefix = ""

    def put_scalar(self, name, value, smoothing_hint=True):
        name = self._prefix + name
        # device tensors are kept as-is and converted when a writer reads them: no host sync per iter
        value = value if isinstance(value, DeferredScalar) else value.detach() if hasattr(value, "detach") else float(value)
        self._history[name].append((value, self._iter))
        self._latest[name] = (value, self._iter)

    def put_scalars(self, *, smoothing_hint=True, **kwargs):
        for k, v in kwargs.items():
            self.put_scalar(k, v, smoothing_hint)

    def history(self, name):
        return self._history[name]

    def latest(self):
        return self._latest

    def latest_with_smoothing_hint(self, window_size=20):
        out = {}
        for k, (v, it) in self._latest.items():
            vals = sorted(float(x[0]) for x in self._history[k][-window_size:])
            out[k] = (vals[len(vals) // 2] if len(vals) % 2 else 0.5 * (vals[len(vals) // 2 - 1] + vals[len(vals) // 2]), it)
        return out

    def step(self):
        self._iter += 1

    @property
    def iter(self):
        return self._iter

    @iter.setter
    def iter(self, v):
        self._iter = int(v)

    @contextmanager
    def name_scope(self, name):
        old = self._prefix
        self._prefix = name.rstrip("/") + "/"
        yield
        self._prefix = old

    def __enter__(self):
        _CURRENT_STORAGE_STACK.append(self)
        return self

    def __exit__(self, *a):
        assert _CURRENT_STORAGE_STACK[-1] is self
        _CURRENT_STORAGE_STACK.pop()


class JSONWriter:
    """One JSON object per write into OUTPUT_DIR/metrics.json (events.py:181-272)."""

    def __init__(self, json_file, window_size=20):
        self._fh = open(json_file, "a")
        self._window = window_size
        self._last = -1

    def write(self):
        st = get_event_storage()
        per_iter = defaultdict(dict)
        for k, (v, it) in st.latest_with_smoothing_hint(self._window).items():
            if it <= self._last:
                continue
            per_iter[it][k] = v
        if per_iter:
            self._last = max(per_iter)
        for it, d in sorted(per_iter.items()):
            d["iteration"] = it
            self._fh.write(json.dumps(d, sort_keys=True) + "\n")
        self._fh.flush()

    def close(self):
        self._fh.close()


class CommonMetricPrinter:
    """iteration, losses, time, data_time, lr, max_mem to the logger (events.py:274-380)."""

    def __init__(self, max_iter=None, window_size=20):
        self.logger = logging.getLogger("divergen_amd")
        self._max_iter = max_iter
        self._window = window_size

    def write(self):
        import torch
        st = get_event_storage()
        it = st.iter
        lat = st.latest_with_smoothing_hint(self._window)
        losses = "  ".join("%s: %.4g" % (k, v[0]) for k, v in lat.items() if "loss" in k)
        t = lat.get("time", (None,))[0]
        dt = lat.get("data_time", (None,))[0]
        lr = lat.get("lr", (None,))[0]
        mem = torch.cuda.max_memory_allocated() / 1024.0 / 1024.0 if torch.cuda.is_available() else None
        self.logger.info(" iter: %d  %s  %s%slr: %s  %s" % (
            it, losses, "time: %.4f  " % t if t is not None else "", "data_time: %.4f  " % dt if dt is not None else "",
            "%.5g" % lr if lr is not None else "N/A", "max_mem: %.0fM" % mem if mem is not None else ""))

    def close(self):
        pass
