"""Checkpoint / resume.  Surface of D2/checkpoint/detection_checkpoint.py:15-120 + fvcore Checkpointer
as DG/train_net.py:139-152,304 uses it: files model_XXXXXXX.pth / model_final.pth / last_checkpoint,
contents {model, optimizer, scheduler, model_ema, iteration}; rank 0 writes; .pkl weights in D2 format
(dict with 'model' of numpy arrays) load with exact-then-longest-suffix key matching (the name-matching
heuristic of D2/checkpoint/c2_model_loading.py:209-335, reduced to what Swin/R50 pkls need)."""
import logging
import os
import pickle

import numpy as np
import torch

from ..utils import comm


def _match_keys(model_keys, ckpt_keys):
    """Map checkpoint keys to model keys (D2/checkpoint/c2_model_loading.py:align_and_update_state_dicts): every MODEL key
    takes the checkpoint key that is its longest '.'-aligned suffix (e.g. 'backbone.bottom_up.layers.0...' <- 'layers.0...'),
    independent of the order of either list; an exact match is the longest possible suffix.  A checkpoint key claimed by
    several model keys is ambiguous and raises, as the reference does."""
    cset = set(ckpt_keys)
    mapping = {}
    for k in model_keys:
        best = None
        if k in cset:
            best = k
        else:
            parts = k.split(".")
            for i in range(1, len(parts)):               # longest suffix first
                suf = ".".join(parts[i:])
                if suf in cset:
                    best = suf
                    break
        if best is None:
            continue
        if best in mapping:
            raise ValueError("Cannot match one checkpoint key to multiple keys in the model: %s <- %s, %s" % (best, mapping[best], k))
        mapping[best] = k
    return mapping


class DetectionCheckpointer:
    def __init__(self, model, save_dir="", *, save_to_disk=None, **checkpointables):
        self.model, self.save_dir = model, save_dir
        self.checkpointables = dict(checkpointables)
        self.save_to_disk = comm.is_main_process() if save_to_disk is None else save_to_disk
        self.logger = logging.getLogger("divergen_amd")

    # ---- save
    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return
        data = {"model": self.model.state_dict()}
        for k, obj in self.checkpointables.items():
            data[k] = obj.state_dict()
        data.update(kwargs)
        os.makedirs(self.save_dir, exist_ok=True)
        basename = "{}.pth".format(name)
        path = os.path.join(self.save_dir, basename)
        torch.save(data, path)
        with open(os.path.join(self.save_dir, "last_checkpoint"), "w") as f:
            f.write(basename)
        self.logger.info("Saved checkpoint to %s", path)

    # ---- load
    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, "last_checkpoint"))

    def get_checkpoint_file(self):
        with open(os.path.join(self.save_dir, "last_checkpoint")) as f:
            return os.path.join(self.save_dir, f.read().strip())

    def resume_or_load(self, path, *, resume=True, model_key="model"):
        """model_key="model_ema": the weights evaluated by `--eval-only` when SOLVER.MODEL_EMA > 0 (DG/train_net.py:340-350
        copies ckpt['model_ema'] over ckpt['model'] through a temporary file before loading)."""
        if resume and self.has_checkpoint():
            return self.load(self.get_checkpoint_file(), model_key=model_key)
        return self.load(path, checkpointables=[], model_key=model_key)

    def _load_file(self, filename):
        if filename.endswith(".pkl"):
            with open(filename, "rb") as f:
                data = pickle.load(f, encoding="latin1")
            if "model" in data and "__author__" in data:
                return data
            if "blobs" in data:
                data = data["blobs"]
            data = {k: v for k, v in data.items() if not k.endswith("_momentum")}
            return {"model": data, "__author__": "Caffe2", "matching_heuristics": True}
        loaded = torch.load(filename, map_location="cpu", weights_only=False)
        if "model" not in loaded:
            loaded = {"model": loaded}
        return loaded

    def load(self, path, checkpointables=None, model_key="model"):
        if not path:
            self.logger.info("No checkpoint found. Initializing model from scratch")
            return {}
        if not os.path.isfile(path):
            raise FileNotFoundError("Checkpoint {} not found!".format(path))
        ckpt = self._load_file(path)
        if model_key != "model":
            if model_key not in ckpt:
                raise KeyError("checkpoint %s has no '%s' entry (keys: %s)" % (path, model_key, sorted(ckpt)))
            ckpt["model"] = ckpt[model_key]
        sd = ckpt.pop("model")
        sd = {k[7:] if k.startswith("module.") else k: (torch.from_numpy(v) if isinstance(v, np.ndarray) else v)
              for k, v in sd.items()}
        model_sd = self.model.state_dict()
        mapping = _match_keys(list(model_sd.keys()), list(sd.keys()))
        new_sd, skipped = {}, []
        for ck, mk in mapping.items():
            if tuple(model_sd[mk].shape) == tuple(sd[ck].shape):
                new_sd[mk] = sd[ck]
            else:
                skipped.append((ck, tuple(sd[ck].shape), tuple(model_sd[mk].shape)))
        incompatible = self.model.load_state_dict(new_sd, strict=False)
        for ck, a, b in skipped:
            self.logger.warning("Skip loading parameter '%s': checkpoint shape %s vs model shape %s", ck, a, b)
        if incompatible.missing_keys:
            self.logger.info("Keys not found in the checkpoint: %d", len(incompatible.missing_keys))
        for k in (self.checkpointables if checkpointables is None else checkpointables):
            if k in ckpt:
                self.checkpointables[k].load_state_dict(ckpt.pop(k))
        return ckpt


class PeriodicCheckpointer:
    def __init__(self, checkpointer, period, max_iter=None):
        self.checkpointer, self.period, self.max_iter = checkpointer, int(period), max_iter

    def step(self, iteration, **kwargs):
        iteration = int(iteration)
        extra = {"iteration": iteration}
        extra.update(kwargs)
        if (iteration + 1) % self.period == 0:
            self.checkpointer.save("model_{:07d}".format(iteration), **extra)
        if self.max_iter is not None and iteration >= self.max_iter - 1:
            self.checkpointer.save("model_final", **extra)
