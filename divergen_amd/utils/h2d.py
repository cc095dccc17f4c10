"""Small host->device uploads that do not stall the stream.

`torch.tensor(list, device="cuda")` copies from pageable memory, which blocks the host until everything queued on
the stream has run -- a full device synchronisation for a dozen integers (per-image offsets, paste descriptors).
Here the values are written into a slot of a ring of PINNED buffers and copied with non_blocking=True; a slot is
reused only after the event recorded behind its previous copy has completed."""
import numpy as np
import torch

_SLOTS, _CAP = 64, 2048
_ring = {}


def upload_i32(values, device):
    """values: sequence / ndarray of integers (<= 2048 of them) -> int32 tensor on `device`, asynchronously."""
    arr = np.asarray(values, dtype=np.int32).reshape(-1)
    n = arr.size
    device = torch.device(device)
    if device.type != "cuda" or n > _CAP:
        return torch.from_numpy(arr.copy()).to(device)
    key = device.index if device.index is not None else torch.cuda.current_device()
    st = _ring.get(key)
    if st is None:
        st = _ring[key] = {"bufs": [torch.empty(_CAP, dtype=torch.int32).pin_memory() for _ in range(_SLOTS)],
                           "ev": [None] * _SLOTS, "i": 0}
    i = st["i"]
    st["i"] = (i + 1) % _SLOTS
    if st["ev"][i] is not None:
        st["ev"][i].synchronize()          # almost always long done
    buf = st["bufs"][i]
    buf[:n].numpy()[:] = arr
    out = torch.empty(n, dtype=torch.int32, device=device)
    out.copy_(buf[:n], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st["ev"][i] = ev
    return out
