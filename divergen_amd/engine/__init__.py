from .ddp import ArenaReducer  # noqa
from .launch import launch, default_argument_parser  # noqa


def total_loss(loss_dict):
    """`sum(loss_dict.values())` of the training loops (DG/train_net.py:262, D2 SimpleTrainer) as ONE stack + ONE reduction:
    the chain of scalar adds costs a launch and an autograd node per loss, in the part of the step where the host is behind."""
    import torch
    if hasattr(loss_dict, "consumed"):          # meta_arch.custom_rcnn.EarlyLosses: part of it is already back-propagated
        if loss_dict.consumed:
            raise RuntimeError("total_loss: this loss dict was summed before; with early_proposal_backward a forward allows ONE backward "
                               "of the plain sum (the proposal generator's gradients are already in the arena)")
        loss_dict.consumed = True
    if any(v.is_cuda for v in loss_dict.values()):
        from ..solver import join_transposes
        join_transposes()                       # (solver.OVERLAP_TRANSPOSES) the backward of this sum reads the transposed weight images
    return torch.stack([v.float().reshape(()) for v in loss_dict.values()]).sum()
