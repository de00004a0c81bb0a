"""Mirror of the reference's trainer/trainer_ifnet.py training-step contract, without
PyTorch-Lightning (a third-party loop, out of scope): ``ImplicitRefinementTrainer`` keeps
``forward(batch)``, ``training_step(batch, batch_idx) -> {'loss': ...}`` and
``configure_optimizers()`` with the same semantics (trainer/trainer_ifnet.py:28-30,40-47):

    logits = ifnet(batch['input'], batch['points'])
    loss   = BCEWithLogits(logits, batch['occupancies'], reduction='none').sum(-1).mean()
    Adam(ifnet.parameters(), lr=hparams.lr)

so a Lightning ``Trainer`` (or the data-parallel loop in ..dp) can drive it unchanged.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from .. import ops
from ..model.ifnet import IFNet


class _BCELogitsSumMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        loss, dz = ops.bce_logits_sum_mean(logits.contiguous(), targets.contiguous().float(), want_grad=True)
        ctx.save_for_backward(dz)
        return loss.squeeze(0)

    @staticmethod
    def backward(ctx, g):
        (dz,) = ctx.saved_tensors
        return dz * g, None


def bce_with_logits_sum_mean(logits, targets):
    """F.binary_cross_entropy_with_logits(reduction='none').sum(-1).mean() on the HIP path."""
    return _BCELogitsSumMeanFn.apply(logits, targets)


class ImplicitRefinementTrainer(nn.Module):
    def __init__(self, kwargs=None, net_res=None):
        super().__init__()
        if kwargs is None:
            kwargs = SimpleNamespace(lr=1e-4, net_res=128)
        self.hparams = kwargs
        self.ifnet = IFNet(net_res=net_res or getattr(kwargs, "net_res", 128))

    def configure_optimizers(self):
        opt_g = torch.optim.Adam(self.ifnet.parameters(), lr=self.hparams.lr)
        return [opt_g], []

    def forward(self, batch):
        return self.ifnet(batch["input"], batch["points"])

    def training_step(self, batch, batch_idx):
        logits = self.forward(batch)
        ce_loss = bce_with_logits_sum_mean(logits, batch["occupancies"])
        return {"loss": ce_loss}
