"""FinetuneEncoderDecoder -- the fine-tuning schedule of the reference's src/thunder/callbacks.py:17-88.

The reference subclasses Lightning's BaseFinetuning: freeze the encoder's parameters before training (BatchNorm parameters stay
trainable when `train_batchnorm`; module train / eval flags are not touched), unfreeze them at `unfreeze_encoder_at_epoch` and add its parameters to the optimizer
with lr / `encoder_initial_lr_div`.  Lightning is not in this image, so the same schedule is provided as a plain object
with the same constructor and hook names; under Lightning it can be driven from a thin `pl.Callback` adapter, without it the
training loop calls `freeze_before_training` once and `finetune_function` at every epoch start."""
from __future__ import annotations

from torch import nn
from torch.optim import Optimizer

_BN = (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d)


class FinetuneEncoderDecoder:
    def __init__(self, unfreeze_encoder_at_epoch: int = 1, encoder_initial_lr_div: float = 10, train_batchnorm: bool = True):
        self.unfreeze_encoder_at_epoch = unfreeze_encoder_at_epoch
        self.encoder_initial_lr_div = encoder_initial_lr_div
        self.train_batchnorm = train_batchnorm

    def on_fit_start(self, trainer, pl_module) -> None:
        if hasattr(pl_module, "encoder") and isinstance(pl_module.encoder, nn.Module):
            return
        raise Exception("The LightningModule should have a nn.Module `encoder` attribute")

    @staticmethod
    def freeze(module: nn.Module, train_bn: bool = True) -> None:
        """BaseFinetuning.freeze (Lightning ^1.7, the version the reference pins): `requires_grad = False` on the direct parameters of
        every module; BatchNorm modules are made trainable instead when `train_bn`.  It flips NO train / eval flag: under
        `Trainer.fit` the frozen encoder keeps running in train mode (batch-statistics BatchNorm with running-stat updates, active
        Dropout) -- only its parameters stop receiving gradients."""
        for m in module.modules():
            trainable = isinstance(m, _BN) and train_bn
            for p in m.parameters(recurse=False):
                p.requires_grad = trainable

    def freeze_before_training(self, pl_module) -> None:
        self.freeze(pl_module.encoder, train_bn=self.train_batchnorm)

    def finetune_function(self, pl_module, epoch: int, optimizer: Optimizer, opt_idx: int = 0) -> None:
        if epoch != self.unfreeze_encoder_at_epoch:
            return
        # BaseFinetuning.unfreeze_and_add_param_group(encoder, optimizer, initial_denom_lr, train_bn = not self.train_batchnorm):
        # make_trainable() on every module, then the parameters that require a gradient, are not BatchNorm's unless train_bn, and are
        # not in the optimizer yet form a new group at lr / initial_denom_lr.  No train() / eval() calls here either.
        enc = pl_module.encoder
        train_bn = not self.train_batchnorm
        for m in enc.modules():
            for p in m.parameters(recurse=False):
                p.requires_grad = True
        known = {id(p) for g in optimizer.param_groups for p in g["params"]}
        new_params = []
        for m in enc.modules():
            if isinstance(m, _BN) and not train_bn:
                continue
            for p in m.parameters(recurse=False):
                if p.requires_grad and id(p) not in known:
                    known.add(id(p))
                    new_params.append(p)
        if new_params:
            optimizer.add_param_group({"params": new_params, "lr": optimizer.param_groups[0]["lr"] / self.encoder_initial_lr_div})
