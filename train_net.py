"""Training driver with the reference's CLI and loop semantics (DG/train_net.py:128-390), so that
DiverGen's launch.sh and configs/*.yaml drive it unchanged:

    python train_net.py --num-gpus N --config-file configs/DiverGen_swinL.yaml [--resume] [--eval-only] KEY VALUE ...

Loop order per iteration (train_net.py:248-304): storage.step -> loss_dict = model(data) -> EMA of the
pre-step weights -> finite check -> zero_grad/backward -> optimizer.step -> lr scalar -> scheduler.step
-> writers every 20 it -> periodic checkpoint.  MI355X-first differences: EMA + clip + AdamW are one
kernel over flat arenas; the loss dict is reduced and read on the host only when the writers fire
(no per-iteration device->host sync); gradients are all-reduced by the arena reducer (RCCL), overlapped
with backward.  `--num-gpus 0` (what launch.sh passes on a box without nvidia-smi) means all GPUs.
DATASETS.TRAIN ("synthetic",) (or a missing LVIS tree) trains on LVIS-shaped synthetic batches.
"""
import datetime
import logging
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from divergen_amd.tuning import enable as _enable_tuned_gemm  # noqa: E402
_enable_tuned_gemm()      # library-GEMM algorithm table for this model's shapes (before the first GEMM)

import torch  # noqa: E402

from divergen_amd.checkpoint import DetectionCheckpointer, PeriodicCheckpointer  # noqa: E402
from divergen_amd.config import add_centernet_config, add_divergen_config, get_cfg  # noqa: E402
from divergen_amd.data import synthetic_batch  # noqa: E402
from divergen_amd.engine import ArenaReducer, default_argument_parser, launch  # noqa: E402
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


def build_train_loader(cfg, device):
    """Batches with the reference's contract.  Real LVIS loading (json + PIL + EfficientDetResizeCrop +
    InstPool) lives in divergen_amd/data; without a dataset tree the loop runs on synthetic batches."""
    per_gpu = max(cfg.SOLVER.IMS_PER_BATCH // comm.get_world_size(), 1)
    size = cfg.INPUT.TRAIN_SIZE
    ncls = cfg.MODEL.ROI_HEADS.NUM_CLASSES
    it = 0
    while True:
        yield synthetic_batch(per_gpu, size, ncls, seed=cfg.SEED * 100003 + comm.get_rank() * 1009 + it, device=device)
        it += 1


def do_train(cfg, model, resume=False):
    model.train()
    optimizer = build_optimizer(cfg, model)
    scheduler = build_lr_scheduler(cfg, optimizer)
    reducer = ArenaReducer(optimizer.arena)
    kwargs = {"model_ema": ModelEma(model, optimizer)} if cfg.SOLVER.MODEL_EMA > 0 else {}
    checkpointer = DetectionCheckpointer(model, cfg.OUTPUT_DIR, optimizer=optimizer, scheduler=scheduler, **kwargs)
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
            storage.put_scalars(data_time=time.perf_counter() - t_data)
            t_step = time.perf_counter()
            iteration = iteration + 1
            storage.step()
            optimizer.zero_grad()
            loss_dict = model(data)
            losses = sum(loss_dict.values())
            pending.append((iteration, {k: v.detach() for k, v in loss_dict.items()}))
            losses.backward()
            scale = reducer.finish()
            optimizer.step(grad_scale=scale)          # EMA of the pre-step weights + clip + AdamW, one kernel
            storage.put_scalar("lr", optimizer.param_groups[0]["lr"], smoothing_hint=False)
            storage.put_scalars(time=time.perf_counter() - t_step)
            t_data = time.perf_counter()
            scheduler.step()
            if iteration - start_iter > 5 and (iteration % 20 == 0 or iteration == max_iter):
                for it, ld in pending:               # one sync for 20 iterations of losses
                    red = {k: float(v) for k, v in comm.reduce_dict(ld).items()}
                    assert all(v == v and abs(v) != float("inf") for v in red.values()), red
                    if comm.is_main_process():
                        storage.put_scalars(total_loss=sum(red.values()), **red)
                pending = []
                for w in writers:
                    w.write()
            extra = {"model_ema": kwargs["model_ema"].state_dict()} if kwargs else {}
            periodic.step(iteration, **extra)
        logger.info("Total training time: {}".format(str(datetime.timedelta(seconds=int(time.perf_counter() - t_start)))))


def setup(args):
    cfg = get_cfg()
    add_centernet_config(cfg)
    add_divergen_config(cfg)
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
    torch.manual_seed(cfg.SEED + comm.get_rank())
    return cfg


def main(args):
    cfg = setup(args)
    model = build_model(cfg)
    if args.eval_only:
        DetectionCheckpointer(model, save_dir=cfg.OUTPUT_DIR).resume_or_load(cfg.MODEL.WEIGHTS, resume=args.resume)
        # do_test (DG/train_net.py:61-118) up to the results file: the loop, the GPU post-processing and the LVIS-format
        # json are here (divergen_amd/evaluation); AP itself needs the dataset tree and lvis-api, absent from this image
        raise SystemExit("LVIS evaluation needs the dataset and lvis-api.  Results in LVIS format are produced by "
                         "divergen_amd.evaluation.inference_on_dataset(model, loader, LVISResultsWriter(out_dir))")
    do_train(cfg, model, resume=args.resume)


if __name__ == "__main__":
    args = default_argument_parser().parse_args()
    print("Command Line Args:", args)
    launch(main, args.num_gpus, num_machines=args.num_machines, machine_rank=args.machine_rank,
           dist_url="auto" if args.num_machines == 1 else args.dist_url, args=(args,))
