"""The reference's extra optimizer / LR-scheduler registrations (host-side, once per epoch).

Mirror of ``/root/reference/graphgps/optimizer/extra_optimizers.py`` under the same registry names and call
signatures, so ``cfg.optim.optimizer`` / ``cfg.optim.scheduler`` resolve on this package alone: ``adagrad``
(:15-18), ``plateau`` (:38-41), ``reduce_on_plateau`` (:44-89), ``linear_with_warmup`` / ``cosine_with_warmup`` /
``polynomial_with_warmup`` (:92-122; the Huggingface warm-up schedules, :125-225).  ``adamW`` is the flat-arena
clip + AdamW of ``optim.py``.  Every scheduler drives ``FlatAdamW`` unchanged: it is a ``torch.optim.Optimizer``
with one param group whose ``lr`` the fused step reads from the device copy refreshed each step."""
import logging
import math

import torch.optim as optim

from .graphgym import register


def adagrad_optimizer(params, base_lr, weight_decay):
    return optim.Adagrad(params, lr=base_lr, weight_decay=weight_decay)


def plateau_scheduler(optimizer, patience, lr_decay):
    return optim.lr_scheduler.ReduceLROnPlateau(optimizer, patience=patience, factor=lr_decay)


def scheduler_reduce_on_plateau(optimizer, reduce_factor, schedule_patience, min_lr, train_mode, eval_period):
    if train_mode == 'standard':
        raise ValueError("ReduceLROnPlateau scheduler is not supported "
                         "by 'standard' graphgym training mode pipeline; "
                         "try setting config 'train.mode: custom'")
    if eval_period != 1:
        logging.warning("When config train.eval_period is not 1, the "
                        "optim.schedule_patience of ReduceLROnPlateau "
                        "may not behave as intended.")
    sched = optim.lr_scheduler.ReduceLROnPlateau(optimizer=optimizer, mode='min', factor=reduce_factor,
                                                 patience=schedule_patience, min_lr=min_lr)
    if not hasattr(sched, 'get_last_lr'):          # older torch: give it the accessor the loggers call
        sched._last_lr = [group['lr'] for group in optimizer.param_groups]
        sched.get_last_lr = lambda: sched._last_lr
    # checkpoints must not carry bound methods (reference :76-87)
    sched.state_dict = lambda: {k: v for k, v in sched.__dict__.items()
                                if k not in ('sparsifier', 'optimizer', 'get_last_lr', 'state_dict')}
    return sched


def _warmup_then(optimizer, warmup, decay, floor):
    """LambdaLR: epoch/warmup (not below ``floor``) during the warm-up epochs, ``decay(epoch)`` afterwards."""
    def factor(epoch):
        if epoch < warmup:
            return max(floor, float(epoch) / float(max(1, warmup)))
        return decay(epoch)
    return optim.lr_scheduler.LambdaLR(optimizer, factor, -1)


def linear_with_warmup_scheduler(optimizer, num_warmup_epochs, max_epoch):
    span = float(max(1, max_epoch - num_warmup_epochs))
    return _warmup_then(optimizer, num_warmup_epochs, lambda t: max(0.0, float(max_epoch - t) / span), 1e-6)


def cosine_with_warmup_scheduler(optimizer, num_warmup_epochs, max_epoch, num_cycles=0.5):
    span = float(max(1, max_epoch - num_warmup_epochs))

    def half_cosine(t):
        progress = float(t - num_warmup_epochs) / span
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))
    return _warmup_then(optimizer, num_warmup_epochs, half_cosine, 1e-6)


def polynomial_with_warmup_scheduler(optimizer, num_warmup_epochs, max_epoch, lr_end=1e-7, power=1.0):
    lr_init = optimizer.defaults["lr"]
    if not (lr_init > lr_end):
        raise ValueError(f"lr_end ({lr_end}) must be be smaller than initial lr ({lr_init})")

    def poly(t):
        if t > max_epoch:
            return lr_end / lr_init
        remaining = 1 - (t - num_warmup_epochs) / (max_epoch - num_warmup_epochs)
        return ((lr_init - lr_end) * remaining ** power + lr_end) / lr_init
    return _warmup_then(optimizer, num_warmup_epochs, poly, 0.0)


register.register_optimizer('adagrad', adagrad_optimizer, overwrite=True)
for _name, _fn in (('plateau', plateau_scheduler), ('reduce_on_plateau', scheduler_reduce_on_plateau),
                   ('linear_with_warmup', linear_with_warmup_scheduler),
                   ('cosine_with_warmup', cosine_with_warmup_scheduler),
                   ('polynomial_with_warmup', polynomial_with_warmup_scheduler)):
    register.register_scheduler(_name, _fn, overwrite=True)
