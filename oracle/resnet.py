"""TEST INFRASTRUCTURE (oracle): CPU restatement of the R50 bottom-up of DG/divergen/modeling/backbone/timm.py:27-151.

The arithmetic lives in timm==0.4.9 (DG/requirements.txt:3), which is NOT vendored by the reference and not installed here:
`timm.models.resnet.ResNet(block=Bottleneck, layers=[3, 4, 6, 3])` -- conv1 7x7/2 pad 3 (no bias), bn1, ReLU,
MaxPool2d(3, 2, 1); four stages of bottlenecks (1x1 -> 3x3 carrying the stride -> 1x1 x4, a norm after each, ReLU after the
first two and after the residual add; 1x1-stride-s downsample + norm on the first block of a stage) -- with every norm
converted to Detectron2's FrozenBatchNorm2d (D2/layers/batch_norm.py:13-111: y = x * w * rsqrt(var + 1e-5) + (b - mean * w *
rsqrt(var + 1e-5))).  No reference test holds vectors for it: **parity unpinned** (restated from the published definition and
the call sites `CustomResNet.forward` :33-47, `TIMM.__init__` :110-134).  Plain torch ops on CPU fp32, state-dict keys as timm's."""
import torch
import torch.nn.functional as F


def frozen_bn(x, sd, prefix, eps=1e-5):
    scale = sd[prefix + ".weight"] * (sd[prefix + ".running_var"] + eps).rsqrt()
    shift = sd[prefix + ".bias"] - sd[prefix + ".running_mean"] * scale
    return x * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)


def bottleneck(x, sd, p, stride):
    y = F.relu(frozen_bn(F.conv2d(x, sd[p + ".conv1.weight"]), sd, p + ".bn1"))
    y = F.relu(frozen_bn(F.conv2d(y, sd[p + ".conv2.weight"], stride=stride, padding=1), sd, p + ".bn2"))
    y = frozen_bn(F.conv2d(y, sd[p + ".conv3.weight"]), sd, p + ".bn3")
    if p + ".downsample.0.weight" in sd:
        x = frozen_bn(F.conv2d(x, sd[p + ".downsample.0.weight"], stride=stride), sd, p + ".downsample.1")
    return F.relu(y + x)


def resnet50_features(x, sd, out_indices=(2, 3, 4), layers=(3, 4, 6, 3)):
    """x (N,3,H,W) fp32, sd = {timm key: tensor} -> the features `CustomResNet.forward` returns for out_indices."""
    x = F.relu(frozen_bn(F.conv2d(x, sd["conv1.weight"], stride=2, padding=3), sd, "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    ret = [x]
    for i, n in enumerate(layers):
        for b in range(n):
            x = bottleneck(x, sd, "layer%d.%d" % (i + 1, b), (1 if i == 0 else 2) if b == 0 else 1)
        ret.append(x)
    return [ret[i] for i in out_indices]
