"""Training driver with the reference's CLI and loop semantics (DG/train_net.py:128-390), so that
DiverGen's launch.sh and configs/*.yaml drive it unchanged:

    python train_net.py --num-gpus N --config-file configs/DiverGen_swinL.yaml [--resume] [--eval-only] KEY VALUE ...

Loop order per iteration (train_net.py:248-304): storage.step -> loss_dict = model(data) -> EMA of the
pre-step weights -> finite check -> zero_grad/backward -> optimizer.step -> lr scalar -> scheduler.step
-> writers every 20 it -> periodic checkpoint.  MI355X-first differences: EMA + clip + AdamW are one
kernel over flat arenas; the loss dict is reduced and read on the host only when the writers fire
(no per-iteration device->host sync); gradients are all-reduced by the arena reducer (RCCL), overlapped
with backward.  `--num-gpus 0` (what launch.sh passes on a box without nvidia-smi) means all GPUs.
DATASETS.TRAIN ("synthetic",) trains on LVIS-shaped synthetic batches; a registered dataset name goes through
divergen_amd/data/build.py (json -> dataset dicts -> mapper -> repeat-factor sampler) and fails loudly when its files are absent.
"""
import datetime
import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from divergen_amd.checkpoint import DetectionCheckpointer, PeriodicCheckpointer  # noqa: E402
from divergen_amd.config import add_bsgal_config, add_centernet_config, add_divergen_config, get_cfg  # noqa: E402
from divergen_amd.data import synthetic_batch  # noqa: E402
from divergen_amd.engine import ArenaReducer, default_argument_parser, launch, total_loss  # noqa: E402
from divergen_amd.modeling import build_model  # noqa: E402
from divergen_amd.solver import build_lr_scheduler, build_optimizer  # noqa: E402
from divergen_amd.utils import comm  # noqa: E402
from divergen_amd.utils.events import CommonMetricPrinter, EventStorage, JSONWriter  # noqa: E402

logger = logging.getLogger("divergen_amd")


class ModelEma:
    """DG/divergen/ema.py surface (state_dict / load_state_dict) over the optimizer's EMA arena."""

    def __init__(self, model, optimizer):
        self.model, self.optimizer = model, optimizer

    def state_dict(self):
        return self.optimizer.ema_state_dict(self.model)

    def load_state_dict(self, sd):
        self.optimizer.load_ema_state_dict(sd)


def base_seed(cfg):
    """D2 seed_all_rng semantics (D2/utils/env.py:26-43, D2/engine/defaults.py:default_setup): SEED < 0 draws a fresh seed,
    shared by all ranks here so that rank-derived streams stay distinct and reproducible within a run."""
    return int(cfg.SEED) if cfg.SEED >= 0 else int(comm.shared_random_seed())


def build_train_loader(cfg, device, seed=None):
    """Batches with the reference's contract (list[dict] with image / instances / height / width).
    DATASETS.TRAIN ("synthetic",) -> LVIS-shaped random batches; anything else goes through divergen_amd.data.build
    (dataset dicts from the registered json, DatasetMapper / CopyPasteMapper, RepeatFactorTrainingSampler) and RAISES when
    the dataset cannot be found -- it never falls back to noise silently."""
    per_gpu = max(cfg.SOLVER.IMS_PER_BATCH // comm.get_world_size(), 1)
    seed = base_seed(cfg) if seed is None else seed
    names = tuple(cfg.DATASETS.TRAIN)
    if names != ("synthetic",):
        from divergen_amd.data.build import build_detection_train_loader
        return build_detection_train_loader(cfg, per_gpu, device, seed)
    return _synthetic_loader(cfg, per_gpu, device, seed)


def _synthetic_loader(cfg, per_gpu, device, seed):
    size = cfg.INPUT.TRAIN_SIZE
    ncls = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    it = 0
    while True:
        yield synthetic_batch(per_gpu, size, ncls, seed=(seed * 100003 + comm.get_rank() * 1009 + it) % (2 ** 63), device=device)
        it += 1


class ema_weights:
    """Context: the model computes with the optimizer's EMA weights (DG/train_net.py:62-64 evaluates `model_ema.ema`, a second
    module; here the EMA is an arena, swapped with the live weights for the duration and swapped back, bf16 shadows included)."""

    def __init__(self, optimizer):
        self.opt = optimizer if (optimizer is not None and getattr(optimizer, "ema", None) is not None) else None

    def _swap(self):
        a = self.opt.arena
        tmp = a.p.clone()
        a.p.copy_(self.opt.ema)
        self.opt.ema.copy_(tmp)
        a.sync_shadow()

    def __enter__(self):
        if self.opt is not None:
            self._swap()

    def __exit__(self, *exc):
        if self.opt is not None:
            self._swap()


def do_test(cfg, model, optimizer=None):
    """DG/train_net.py:62-126 for LVIS-type test sets: test loader (one image per batch, sharded over ranks), GPU
    post-processing + run-length encoding, results json, box / mask AP.  `optimizer` with an EMA arena: the EMA weights are
    the ones evaluated (:63-64)."""
    with ema_weights(optimizer):
        was_training = model.training
        try:
            return _do_test(cfg, model)
        finally:
            model.train(was_training)


def _do_test(cfg, model):
    from collections import OrderedDict
    from divergen_amd.data.build import build_detection_test_loader
    from divergen_amd.evaluation import LVISEvaluator, inference_on_dataset, print_csv_format
    results = OrderedDict()
    device = torch.device(cfg.MODEL.DEVICE)
    for dataset_name in cfg.DATASETS.TEST:
        loader = build_detection_test_loader(cfg, dataset_name, device)
        out_dir = os.path.join(cfg.OUTPUT_DIR, "inference_{}".format(dataset_name))
        evaluator = LVISEvaluator(dataset_name, cfg, True, out_dir, max_dets_per_image=cfg.TEST.DETECTIONS_PER_IMAGE)
        results[dataset_name] = inference_on_dataset(model, loader, evaluator)
        if comm.is_main_process():
            logger.info("Evaluation results for {} in csv format:".format(dataset_name))
            print_csv_format(results[dataset_name])
    return list(results.values())[0] if len(results) == 1 else results


def do_train(cfg, model, resume=False):
    model.train()
    optimizer = build_optimizer(cfg, model)
    scheduler = build_lr_scheduler(cfg, optimizer)
    reducer = ArenaReducer(optimizer.arena, wire_dtype=cfg.SOLVER.get("ALLREDUCE_DTYPE", "fp32"))
    # this loop back-propagates the plain sum of the loss dict once per zero_grad: the proposal generator's part may run from inside
    # the forward, ahead of the RoI heads' device->host read (meta_arch/custom_rcnn.py)
    model.early_proposal_backward = True
    # ... and every reader of the transposed weight images sits behind a join in CustomRCNN / engine.total_loss: the refresh behind the
    # optimizer step may run beside the next forward (solver.OVERLAP_TRANSPOSES); BSGAL's trial steps refresh synchronously
    from divergen_amd import solver as _solver
    overlap_before, _solver.OVERLAP_TRANSPOSES = _solver.OVERLAP_TRANSPOSES, not cfg.INPUT.get("ACTIVE_SELECT", False)
    if cfg.INPUT.get("ACTIVE_SELECT", False):
        # BSGAL (BS/train_net.py:358-557 + the selection inside its CustomRCNN.forward): the model decides per step whether the
        # pasted batch or its un-pasted original is trained on; needs the parameter arena, hence attached here
        from divergen_amd.engine.bsgal import ActiveSelector
        model.active_selector = ActiveSelector.from_config(cfg, model, optimizer.arena, model.training_losses, rank=comm.get_rank())
    kwargs = {"model_ema": ModelEma(model, optimizer)} if cfg.SOLVER.MODEL_EMA > 0 else {}
    checkpointer = DetectionCheckpointer(model, cfg.OUTPUT_DIR, optimizer=optimizer, scheduler=scheduler, **kwargs)
    if cfg.MODEL.WEIGHTS and not resume and not os.path.isfile(cfg.MODEL.WEIGHTS):
        raise FileNotFoundError("Checkpoint {} not found!".format(cfg.MODEL.WEIGHTS))      # as the reference's checkpointer
    start_iter = checkpointer.resume_or_load(cfg.MODEL.WEIGHTS if os.path.isfile(cfg.MODEL.WEIGHTS) else "",
                                             resume=resume).get("iteration", -1) + 1
    if not resume:
        start_iter = 0
    reducer.broadcast_parameters()
    optimizer.arena.sync_shadow()
    if optimizer.ema is not None and start_iter == 0:
        optimizer.ema.copy_(optimizer.arena.p)
    max_iter = cfg.SOLVER.MAX_ITER if cfg.SOLVER.TRAIN_ITER < 0 else cfg.SOLVER.TRAIN_ITER
    periodic = PeriodicCheckpointer(checkpointer, cfg.SOLVER.CHECKPOINT_PERIOD, max_iter=max_iter)
    writers = [CommonMetricPrinter(max_iter), JSONWriter(os.path.join(cfg.OUTPUT_DIR, "metrics.json"))] \
        if comm.is_main_process() else []
    device = torch.device(cfg.MODEL.DEVICE)
    loader = build_train_loader(cfg, device)
    logger.info("Starting training from iteration {}".format(start_iter))
    pending = []   # (iteration, loss_dict) kept on the device until the writers need them
    with EventStorage(start_iter) as storage:
        t_start = time.perf_counter()
        t_data = time.perf_counter()
        for data, iteration in zip(loader, range(start_iter, max_iter)):
            # the loader composites one batch ahead inside next(): `wait_s` is the part of it spent blocked on the workers
            storage.put_scalars(data_time=getattr(loader, "wait_s", time.perf_counter() - t_data))
            t_step = time.perf_counter()
            iteration = iteration + 1
            storage.step()
            optimizer.zero_grad()
            loss_dict = model(data)
            losses = total_loss(loss_dict)
            pending.append((iteration, {k: v.detach() for k, v in loss_dict.items()}))
            # DG/train_net.py:266 asserts isfinite(losses) before backward; here the flag stays on the device (no sync): a
            # non-finite loss makes the optimizer kernel skip the weights, moments and EMA of this step (found_inf), and the
            # deferred host check below runs BEFORE the periodic checkpointer can save
            bad = reducer.agree_on_skip((~torch.isfinite(losses.detach())).to(torch.int32))      # every rank skips, or none
            reducer.begin_backward()
            losses.backward()
            scale = reducer.finish()
            optimizer.step(grad_scale=scale, found_inf=bad)   # EMA of the pre-step weights + clip + AdamW, one kernel
            storage.put_scalar("lr", optimizer.param_groups[0]["lr"], smoothing_hint=False)
            storage.put_scalars(time=time.perf_counter() - t_step)
            t_data = time.perf_counter()
            scheduler.step()
            if cfg.TEST.EVAL_PERIOD > 0 and iteration % cfg.TEST.EVAL_PERIOD == 0 and iteration != max_iter:      # DG/train_net.py:294-298
                do_test(cfg, model, optimizer)
                comm.synchronize()
            # the checkpointer's own predicate (PeriodicCheckpointer.step): the deferred finite-loss check runs before it can save
            saves_now = (cfg.SOLVER.CHECKPOINT_PERIOD > 0 and (iteration + 1) % cfg.SOLVER.CHECKPOINT_PERIOD == 0) \
                or iteration >= max_iter - 1
            if (iteration - start_iter > 5 and (iteration % 20 == 0 or iteration == max_iter)) or saves_now:
                for it, ld in pending:               # one sync for 20 iterations of losses
                    red = {k: float(v) for k, v in comm.reduce_dict(ld).items()}
                    assert all(v == v and abs(v) != float("inf") for v in red.values()), red
                    if comm.is_main_process():
                        storage.put_scalars(total_loss=sum(red.values()), **red)
                pending = []
                for w in writers:
                    w.write()
            # (the EMA state dict materialises permuted copies of the box heads' first FC: only when a checkpoint is written)
            extra = {"model_ema": kwargs["model_ema"].state_dict()} if (kwargs and saves_now) else {}
            periodic.step(iteration, **extra)
        logger.info("Total training time: {}".format(str(datetime.timedelta(seconds=int(time.perf_counter() - t_start)))))
    _solver.OVERLAP_TRANSPOSES = overlap_before         # (a process-wide switch: whatever runs after this loop gets the default back)
    _solver.join_transposes()
    return optimizer


def setup(args):
    cfg = get_cfg()
    add_centernet_config(cfg)
    add_divergen_config(cfg)
    add_bsgal_config(cfg)          # BS/configs/BSGAL/*.yaml (MODEL.ACTIVE_*, INPUT.ACTIVE_SELECT) load as well
    cfg.merge_from_file(args.config_file)
    cfg.merge_from_list(args.opts)
    if "/auto" in cfg.OUTPUT_DIR:
        name = os.path.basename(args.config_file)[:-5]
        cfg.OUTPUT_DIR = cfg.OUTPUT_DIR.replace("/auto", "/{}".format(name))
    cfg.freeze()
    os.makedirs(cfg.OUTPUT_DIR, exist_ok=True)
    if comm.is_main_process():
        with open(os.path.join(cfg.OUTPUT_DIR, "config.yaml"), "w") as f:
            f.write(cfg.dump())
        logging.basicConfig(level=logging.INFO, format="[%(asctime)s %(name)s]: %(message)s",
                            handlers=[logging.StreamHandler(), logging.FileHandler(os.path.join(cfg.OUTPUT_DIR, "log.txt"))])
    torch.manual_seed(base_seed(cfg) + comm.get_rank())
    return cfg


def main(args):
    cfg = setup(args)
    model = build_model(cfg)
    if args.eval_only:
        # DG/train_net.py:340-354: with SOLVER.MODEL_EMA > 0 the checkpoint's EMA weights are the model that is evaluated
        key = "model_ema" if cfg.SOLVER.MODEL_EMA > 0 else "model"
        DetectionCheckpointer(model, save_dir=cfg.OUTPUT_DIR).resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume, model_key=key)
        return do_test(cfg, model.to(torch.device(cfg.MODEL.DEVICE)))
    optimizer = do_train(cfg, model, resume=args.resume)
    res = do_test(cfg, model, optimizer)        # DG/train_net.py:368: the EMA weights when there are any
    comm.synchronize()
    return res


if __name__ == "__main__":
    args = default_argument_parser().parse_args()
    print("Command Line Args:", args)
    launch(main, args.num_gpus, num_machines=args.num_machines, machine_rank=args.machine_rank,
           dist_url="auto" if args.num_machines == 1 else args.dist_url, args=(args,))
