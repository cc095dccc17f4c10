"""CPU tests: samplers vs the reference's known-answer tests, checkpointer round trip + key matching."""
import math
import os

import torch


def test_repeat_factor_sampler_kat_d2t():
    # D2T/data/test_sampler.py:76-90
    from divergen_amd.data.samplers import RepeatFactorTrainingSampler
    dataset_dicts = [
        {"annotations": [{"category_id": 0}, {"category_id": 1}]},
        {"annotations": [{"category_id": 0}]},
        {"annotations": []},
    ]
    rep = RepeatFactorTrainingSampler.repeat_factors_from_category_frequency(dataset_dicts, 0.5)
    assert torch.allclose(rep, torch.tensor([math.sqrt(3 / 2), 1.0, 1.0]))


def test_training_sampler_seeded_and_sharded():
    # D2T/data/test_sampler.py:63-74: seeded stream; ranks take a strided slice of ONE stream
    from divergen_amd.data.samplers import TrainingSampler
    import itertools
    full = list(itertools.islice(TrainingSampler(5, True, seed=42, rank=0, world_size=1), 10))
    assert sorted(full[:5]) == list(range(5)) and sorted(full[5:]) == list(range(5))
    r0 = list(itertools.islice(TrainingSampler(5, True, seed=42, rank=0, world_size=2), 5))
    r1 = list(itertools.islice(TrainingSampler(5, True, seed=42, rank=1, world_size=2), 5))
    assert r0 == full[0::2] and r1 == full[1::2]


def test_inference_sampler_shards():
    # D2T/data/test_sampler.py:94-111
    from divergen_amd.data.samplers import InferenceSampler
    assert list(InferenceSampler._get_local_indices(100, 4, 0)) == list(range(25))
    sizes = [len(InferenceSampler._get_local_indices(10, 3, r)) for r in range(3)]
    assert sizes == [4, 3, 3]
    allidx = sum([list(InferenceSampler._get_local_indices(10, 3, r)) for r in range(3)], [])
    assert allidx == list(range(10))


def test_checkpointer_roundtrip_and_suffix_matching(tmp_path):
    from divergen_amd.checkpoint import DetectionCheckpointer, PeriodicCheckpointer, _match_keys
    net = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    ck = DetectionCheckpointer(net, str(tmp_path), save_to_disk=True)
    per = PeriodicCheckpointer(ck, 2, max_iter=4)
    for it in range(4):
        per.step(it)
    files = sorted(os.listdir(tmp_path))
    assert files == ["last_checkpoint", "model_0000001.pth", "model_0000003.pth", "model_final.pth"]
    net2 = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.Linear(3, 2))
    extra = DetectionCheckpointer(net2, str(tmp_path)).resume_or_load("", resume=True)
    assert extra["iteration"] == 3
    for a, b in zip(net.parameters(), net2.parameters()):
        assert torch.equal(a, b)
    m = _match_keys(["backbone.bottom_up.layers.0.blocks.0.attn.qkv.weight", "backbone.fpn_lateral3.weight"],
                    ["layers.0.blocks.0.attn.qkv.weight", "fpn_lateral3.weight", "head.weight"])
    assert m == {"layers.0.blocks.0.attn.qkv.weight": "backbone.bottom_up.layers.0.blocks.0.attn.qkv.weight",
                 "fpn_lateral3.weight": "backbone.fpn_lateral3.weight"}
