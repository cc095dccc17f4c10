"""Batched-input contract types: Boxes, Instances, ImageList, BitMasks.
Surface of D2/structures/{boxes.py:130-357, instances.py, image_list.py:59-110, masks.py:88-245}."""
import itertools
from typing import Any, Dict, List, Tuple

import torch
import torch.nn.functional as F


class Boxes:
    """(N,4) fp32 XYXY boxes."""

    def __init__(self, tensor):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor, dtype=torch.float32)
        tensor = tensor.to(torch.float32)
        if tensor.numel() == 0:
            tensor = tensor.reshape((-1, 4))
        assert tensor.dim() == 2 and tensor.size(-1) == 4, tensor.size()
        self.tensor = tensor

    def clone(self):
        return Boxes(self.tensor.clone())

    def to(self, device):
        return Boxes(self.tensor.to(device=device, non_blocking=torch.device(device).type == "cuda"))      # host -> device only

    def pin_memory(self):
        return Boxes(self.tensor.pin_memory())

    def area(self):
        b = self.tensor
        return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

    def clip(self, box_size):
        h, w = box_size
        x1 = self.tensor[:, 0].clamp(min=0, max=w)
        y1 = self.tensor[:, 1].clamp(min=0, max=h)
        x2 = self.tensor[:, 2].clamp(min=0, max=w)
        y2 = self.tensor[:, 3].clamp(min=0, max=h)
        self.tensor = torch.stack((x1, y1, x2, y2), dim=-1)

    def nonempty(self, threshold=0.0):
        b = self.tensor
        return ((b[:, 2] - b[:, 0]) > threshold) & ((b[:, 3] - b[:, 1]) > threshold)

    def __getitem__(self, item):
        if isinstance(item, int):
            return Boxes(self.tensor[item].view(1, -1))
        b = self.tensor[item]
        assert b.dim() == 2, "Indexing on Boxes with {} failed to return a matrix!".format(item)
        return Boxes(b)

    def __len__(self):
        return self.tensor.shape[0]

    def __repr__(self):
        return "Boxes(" + str(self.tensor) + ")"

    def get_centers(self):
        return (self.tensor[:, :2] + self.tensor[:, 2:]) / 2

    @property
    def device(self):
        return self.tensor.device

    @classmethod
    def cat(cls, boxes_list):
        if len(boxes_list) == 0:
            return cls(torch.empty(0))
        return cls(torch.cat([b.tensor for b in boxes_list], dim=0))


def pairwise_iou(boxes1: Boxes, boxes2: Boxes):
    """boxes.py:310-357 (torch ops; the training path uses layers.iou_match instead)."""
    b1, b2 = boxes1.tensor, boxes2.tensor
    wh = (torch.min(b1[:, None, 2:], b2[:, 2:]) - torch.max(b1[:, None, :2], b2[:, :2])).clamp_(min=0)
    inter = wh.prod(dim=2)
    return torch.where(inter > 0, inter / (boxes1.area()[:, None] + boxes2.area() - inter),
                       torch.zeros(1, dtype=inter.dtype, device=inter.device))


class BitMasks:
    """(N,H,W) bool masks (D2/structures/masks.py:87-255).

    Indexing with an index / boolean tensor is LAZY: the result shares the mask storage and carries the row
    indices.  The RoI heads index the ground-truth masks of an image with its 512 sampled proposals (and again with
    the foreground subset); materialising that is a 0.5 GB gather per image at 1024^2 that the reference pays
    (Instances.__getitem__), while the only consumer -- crop_and_resize -- reads a few rows through the index."""

    def __init__(self, tensor, index=None):
        if not isinstance(tensor, torch.Tensor):
            tensor = torch.as_tensor(tensor)
        tensor = tensor.to(torch.bool)
        assert tensor.dim() == 3, tensor.size()
        self.image_size = tensor.shape[1:]
        self._base = tensor
        self._index = index          # None = identity

    @property
    def tensor(self):
        if self._index is not None:      # materialise on demand
            self._base, self._index = self._base[self._index], None
        return self._base

    def to(self, *args, **kwargs):
        base = self._base.to(*args, **kwargs)
        return BitMasks(base, None if self._index is None else self._index.to(base.device))

    def pin_memory(self):
        return BitMasks(self._base.pin_memory(), None if self._index is None else self._index.pin_memory())

    @property
    def device(self):
        return self._base.device

    def __getitem__(self, item):
        if isinstance(item, int):
            return BitMasks(self.tensor[item].unsqueeze(0))
        if isinstance(item, torch.Tensor) and item.dim() == 1 and item.dtype in (torch.int64, torch.int32, torch.bool):
            cur = self._index if self._index is not None else torch.arange(self._base.shape[0], device=self._base.device)
            return BitMasks(self._base, cur[item.to(cur.device)])
        if isinstance(item, slice):          # stays lazy: slicing a sampled proposal list must not gather full-resolution masks
            if self._index is None:
                return BitMasks(self._base[item])
            return BitMasks(self._base, self._index[item])
        m = self.tensor[item]
        assert m.dim() == 3
        return BitMasks(m)

    def __len__(self):
        return self._base.shape[0] if self._index is None else self._index.shape[0]

    def nonempty(self):
        return self.tensor.flatten(1).any(dim=1)

    def crop_and_resize(self, boxes, mask_size):
        """masks.py:189-220 through the byte-tap HIP crop (no fp32 mask copy, rows addressed through the index)."""
        from ..layers import mask_crop
        if self._index is None:
            idx = torch.arange(len(boxes), device=boxes.device, dtype=torch.int32)
        else:
            idx = self._index.to(device=boxes.device, dtype=torch.int32)
        return mask_crop(self._base, boxes, idx, mask_size)

    def get_bounding_boxes(self):
        boxes = torch.zeros(self.tensor.shape[0], 4, dtype=torch.float32)
        x_any = torch.any(self.tensor, dim=1)
        y_any = torch.any(self.tensor, dim=2)
        for i in range(self.tensor.shape[0]):
            x = torch.where(x_any[i, :])[0]
            y = torch.where(y_any[i, :])[0]
            if len(x) > 0 and len(y) > 0:
                boxes[i, :] = torch.as_tensor([x[0], y[0], x[-1] + 1, y[-1] + 1], dtype=torch.float32)
        return Boxes(boxes)

    @staticmethod
    def cat(lst):
        return BitMasks(torch.cat([m.tensor for m in lst], dim=0))


def _join_field(values):
    """One field of several Instances joined along the instance axis: tensors by torch.cat, lists by concatenation, anything
    else through its own static `cat` (Boxes, BitMasks, ...)."""
    head = values[0]
    if torch.is_tensor(head):
        return torch.cat(values, dim=0)
    if isinstance(head, list):
        return [x for part in values for x in part]
    joiner = getattr(type(head), "cat", None)
    if joiner is None:
        raise ValueError("Instances.cat: no way to join fields of type %s" % type(head).__name__)
    return joiner(values)


class Instances:
    """The per-image record of the Detectron2 batched-input contract (D2/structures/instances.py): an image size plus named
    fields that all have one entry per instance.  Fields are reachable as attributes (`inst.gt_boxes`), through
    set / get / has / remove, and the record as a whole can be indexed, moved between devices and concatenated."""

    def __init__(self, image_size: Tuple[int, int], **fields: Any):
        object.__setattr__(self, "_image_size", image_size)
        object.__setattr__(self, "_fields", {})
        for name, value in fields.items():
            self.set(name, value)

    # -- field access
    @property
    def image_size(self):
        return self._image_size

    def __setattr__(self, name, value):
        if name[:1] == "_":
            object.__setattr__(self, name, value)
        else:
            self.set(name, value)

    def __getattr__(self, name):
        fields = object.__getattribute__(self, "_fields") if name != "_fields" else None
        if fields is None or name not in fields:
            raise AttributeError("Cannot find field '{}' in the given Instances!".format(name))
        return fields[name]

    def set(self, name, value):
        if self._fields and len(value) != len(self):
            raise AssertionError("field '%s' has %d entries, the Instances holds %d" % (name, len(value), len(self)))
        self._fields[name] = value

    def has(self, name):
        return name in self._fields

    def remove(self, name):
        self._fields.pop(name)

    def get(self, name):
        return self._fields[name]

    def get_fields(self):
        return self._fields

    # -- whole-record operations
    def _rebuild(self, fn):
        out = Instances(self._image_size)
        for name, value in self._fields.items():
            out.set(name, fn(value))
        return out

    def to(self, *args, **kwargs):
        return self._rebuild(lambda v: v.to(*args, **kwargs) if hasattr(v, "to") else v)

    def pin_memory(self):
        """torch.utils.data's pin thread calls this on custom batch members: worker results go up asynchronously."""
        return self._rebuild(lambda v: v.pin_memory() if hasattr(v, "pin_memory") else v)

    def __len__(self):
        if not self._fields:
            raise NotImplementedError("an Instances without fields has no length")
        return len(next(iter(self._fields.values())))

    def __getitem__(self, item):
        if isinstance(item, int) and not isinstance(item, bool):
            n = len(self)
            if not -n <= item < n:
                raise IndexError("Instances index %d out of range for %d instances" % (item, n))
            item = slice(item % n, item % n + 1)            # keep the instance axis
        return self._rebuild(lambda v: v[item])

    @staticmethod
    def cat(instance_lists):
        if not instance_lists:
            raise AssertionError("Instances.cat needs at least one Instances")
        first = instance_lists[0]
        if len(instance_lists) == 1:
            return first
        out = Instances(first.image_size)
        for name in first._fields:
            out.set(name, _join_field([inst.get(name) for inst in instance_lists]))
        return out

    def __repr__(self):
        kinds = ", ".join("%s: %s" % (k, type(v).__name__) for k, v in self._fields.items())
        return "Instances(num=%d, size=%s, fields=[%s])" % (len(self) if self._fields else 0, self._image_size, kinds)


class PatchRows:
    """The padded, normalised batch in the layout the Swin PatchEmbed consumes: rows (B, Hp*Wp, 3*4*4) bf16, one row per 4x4
    patch in (c, dy, dx) order -- written by dgx_preprocess_patches straight from the uint8 images.  `shape` is the (B,3,H,W)
    the batch tensor would have."""

    def __init__(self, rows, Hp, Wp):
        self.rows, self.Hp, self.Wp = rows, Hp, Wp
        self.shape = (rows.shape[0], 3, 4 * Hp, 4 * Wp)
        self.device = rows.device

    def to_tensor(self, dtype=torch.float32):
        B, _, H, W = self.shape
        return self.rows.view(B, self.Hp, self.Wp, 3, 4, 4).permute(0, 3, 1, 4, 2, 5).reshape(B, 3, H, W).to(dtype)


class ImageList:
    """Batch of images padded to a common, divisible size (image_list.py:59-110).  With `patch_rows` the batch exists only as
    PatchEmbed's GEMM operand; `.tensor` is then rebuilt from it on demand (bf16-rounded values)."""

    def __init__(self, tensor, image_sizes, patch_rows=None):
        self._tensor = tensor
        self.image_sizes = image_sizes
        self.patch_rows = patch_rows

    @property
    def tensor(self):
        if self._tensor is None and self.patch_rows is not None:
            self._tensor = self.patch_rows.to_tensor()
        return self._tensor

    def __len__(self):
        return len(self.image_sizes)

    def __getitem__(self, idx):
        size = self.image_sizes[idx]
        return self.tensor[idx, ..., : size[0], : size[1]]

    @property
    def device(self):
        return self.patch_rows.device if self._tensor is None and self.patch_rows is not None else self.tensor.device

    @staticmethod
    def from_tensors(tensors, size_divisibility=0, pad_value=0.0):
        assert len(tensors) > 0
        image_sizes = [(im.shape[-2], im.shape[-1]) for im in tensors]
        max_h = max(s[0] for s in image_sizes)
        max_w = max(s[1] for s in image_sizes)
        if size_divisibility > 1:
            st = size_divisibility
            max_h = (max_h + (st - 1)) // st * st
            max_w = (max_w + (st - 1)) // st * st
        if len(tensors) == 1:
            h, w = image_sizes[0]
            batched = F.pad(tensors[0], [0, max_w - w, 0, max_h - h], value=pad_value).unsqueeze_(0)
        else:
            shape = [len(tensors)] + list(tensors[0].shape[:-2]) + [max_h, max_w]
            batched = tensors[0].new_full(shape, pad_value)
            for img, pad_img in zip(tensors, batched):
                pad_img[..., : img.shape[-2], : img.shape[-1]].copy_(img)
        return ImageList(batched.contiguous(), image_sizes)


class ProposalBatch(list):
    """list[Instances] that also carries its batch-level tensors (`batch`: fixed-length proposals of the whole batch from the
    proposal generator; `train`: the sampled rows of the whole batch from label_and_sample_proposals)."""
    batch = None
    train = None
