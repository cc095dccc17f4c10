"""Box2BoxTransform.  Mirrors D2/modeling/box_regression.py:21-118."""
import math

import torch

_DEFAULT_SCALE_CLAMP = math.log(1000.0 / 16)


class Box2BoxTransform:
    def __init__(self, weights, scale_clamp=_DEFAULT_SCALE_CLAMP):
        self.weights, self.scale_clamp = tuple(weights), scale_clamp

    def get_deltas(self, src, tgt):
        sw, sh = src[:, 2] - src[:, 0], src[:, 3] - src[:, 1]
        sx, sy = src[:, 0] + 0.5 * sw, src[:, 1] + 0.5 * sh
        tw, th = tgt[:, 2] - tgt[:, 0], tgt[:, 3] - tgt[:, 1]
        tx, ty = tgt[:, 0] + 0.5 * tw, tgt[:, 1] + 0.5 * th
        wx, wy, ww, wh = self.weights
        return torch.stack((wx * (tx - sx) / sw, wy * (ty - sy) / sh, ww * torch.log(tw / sw), wh * torch.log(th / sh)), dim=1)

    def apply_deltas(self, deltas, boxes):
        deltas = deltas.float()
        boxes = boxes.to(deltas.dtype)
        w, h = boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]
        cx, cy = boxes[:, 0] + 0.5 * w, boxes[:, 1] + 0.5 * h
        wx, wy, ww, wh = self.weights
        dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
        dw = torch.clamp(deltas[:, 2::4] / ww, max=self.scale_clamp)
        dh = torch.clamp(deltas[:, 3::4] / wh, max=self.scale_clamp)
        pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
        pw, ph = torch.exp(dw) * w[:, None], torch.exp(dh) * h[:, None]
        out = torch.stack((pcx - 0.5 * pw, pcy - 0.5 * ph, pcx + 0.5 * pw, pcy + 0.5 * ph), dim=-1)
        return out.reshape(deltas.shape)
