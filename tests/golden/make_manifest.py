"""State-dict manifest of the reference's Swin-L CenterNet2 model (configs/DiverGen_swinL.yaml): key -> shape of every parameter
and persistent buffer, produced by instantiating the REFERENCE's own module classes (stub-import harness _refload.py) with the
constructor arguments its builders pass for that configuration, under the attribute names its meta-architecture gives them:

    CustomRCNN / GeneralizedRCNN (D2/modeling/meta_arch/rcnn.py:40-65)      backbone, proposal_generator, roi_heads
    build_swintransformer_fpn_backbone (DG/divergen/modeling/backbone/swintransformer.py)  FPN(bottom_up=SwinTransformer, ...)
    CenterNet (CN/modeling/dense_heads/centernet.py)                          centernet_head = CenterNetHead(...)
    DeticCascadeROIHeads (DG/divergen/modeling/roi_heads/detic_roi_heads.py)  box_head[k], box_predictor[k], mask_head

AUTHORING CONTAINER ONLY (reads /root/reference).  Output: tests/golden/swinL_state_manifest.json -- data (names and shapes), no
source.  What it is for: the released checkpoint (DiverGen/README.md:57) is a state dict with exactly these keys; the test
tests/test_host_logic.py::test_state_dict_matches_the_reference_manifest holds the registry-built model to it, so that
`train_net.py --eval-only MODEL.WEIGHTS <checkpoint>` cannot fail on loading.
"""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import _refload as R  # noqa: E402


def main():
    R.install()
    sw = R.ref("divergen.modeling.backbone.swintransformer")
    fpn_m = sys.modules.get("detectron2.modeling.backbone.fpn") or R.ref("detectron2.modeling.backbone.fpn")
    f5 = R.ref("centernet.modeling.backbone.fpn_p5")
    ch = R.ref("centernet.modeling.dense_heads.centernet_head")
    mh = R.ref("detectron2.modeling.roi_heads.mask_head")
    bh = R.ref("detectron2.modeling.roi_heads.box_head")
    fr = R.ref("divergen.modeling.roi_heads.detic_fast_rcnn")
    ss = sys.modules["detectron2.layers"].ShapeSpec
    NUM_CLASSES = 1453                                       # configs/DiverGen_swinL.yaml: MODEL.ROI_HEADS.NUM_CLASSES
    c = sw.size2config["L-22k-384"]
    # (the Swin constructor calls .item() on its drop-path schedule: built on the CPU, 197 M parameters; the rest on `meta`)
    swin = sw.SwinTransformer(embed_dim=c["embed_dim"], window_size=c["window_size"], depths=c["depth"], num_heads=c["num_heads"],
                              drop_path_rate=c["drop_path_rate"], out_indices=(1, 2, 3), frozen_stages=-1, use_checkpoint=False)
    with torch.device("meta"):
        fpn = fpn_m.FPN(bottom_up=swin, in_features=["swin1", "swin2", "swin3"], out_channels=256, norm="",
                        top_block=f5.LastLevelP6P7_P5(256, 256), fuse_type="sum")
        head = ch.CenterNetHead(in_channels=256, num_levels=5, num_classes=NUM_CLASSES, with_agn_hm=True, only_proposal=True,
                                norm="GN", num_cls_convs=4, num_box_convs=4, num_share_convs=0, use_deformable=False, prior_prob=0.01)
        box_heads = [bh.FastRCNNConvFCHead(ss(channels=256, height=7, width=7), conv_dims=[], fc_dims=[1024, 1024]) for _ in range(3)]
        mask = mh.MaskRCNNConvUpsampleHead(ss(channels=256, height=14, width=14), num_classes=1, conv_dims=[256] * 5, conv_norm="")
    out = {}

    def add(prefix, module):
        for k, v in module.state_dict().items():
            out[prefix + k] = list(v.shape)
    add("backbone.", fpn)
    add("proposal_generator.centernet_head.", head)
    for k, m in enumerate(box_heads):
        add("roi_heads.box_head.%d." % k, m)
    # DeticFastRCNNOutputLayers (detic_fast_rcnn.py:30-135) = FastRCNNOutputLayers(cls_score: Linear(1024, C + 1), bbox_pred:
    # Linear(1024, 4) with CLS_AGNOSTIC_BBOX_REG) + the `freq_weight` buffer (use_fed_loss; extended to num_classes at :92-98).
    # Its constructor reads the category-frequency json; the layer shapes are taken from the class's own construction on a stub path.
    import tempfile
    cats = [{"id": i + 1, "image_count": 1 + (i % 7)} for i in range(1203)]
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(cats, f)
    b2b = R.ref("detectron2.modeling.box_regression").Box2BoxTransform(weights=(10.0, 10.0, 5.0, 5.0))
    pred = fr.DeticFastRCNNOutputLayers(ss(channels=1024), box2box_transform=b2b, num_classes=NUM_CLASSES, cls_agnostic_bbox_reg=True,
                                        use_sigmoid_ce=True, use_fed_loss=True, cat_freq_path=f.name, fed_loss_freq_weight=0.5,
                                        test_score_thresh=0.0001, test_topk_per_image=300)
    os.unlink(f.name)
    for k in range(3):
        add("roi_heads.box_predictor.%d." % k, pred)
    add("roi_heads.mask_head.", mask)
    dst = os.path.join(HERE, "swinL_state_manifest.json")
    with open(dst, "w") as f:
        json.dump({"_what": "key -> shape of the reference's Swin-L CenterNet2 state dict (tests/golden/make_manifest.py)",
                   "config": "configs/DiverGen_swinL.yaml", "entries": out}, f, indent=0, sort_keys=True)
    print("wrote", dst, len(out), "entries,", sum(int(torch.tensor(v).prod()) if v else 1 for v in out.values()) / 1e6, "M elements")


if __name__ == "__main__":
    main()
