"""The caller of the hot path: one optimisation step and ``train_epoch``.

Mirror of ``graphgps/train/custom_train.py:16-47``::

    pred, true = model(batch); loss, pred_score = compute_loss(pred, true); loss.backward()
    every ``batch_accumulation`` iterations: clip_grad_norm_ -> optimizer.step() -> zero_grad()
    logger.update_stats(true, pred, loss, lr, time_used, params, dataset_name)

with three differences that are about the machine, not the arithmetic:
  * the reference's per-iteration ``loss.detach().cpu().item()`` (:42) -- a device sync that
    drains the queue every step -- is gone: losses/predictions stay on the device and the logger
    is fed once per epoch;
  * clip + AdamW are the two-launch flat-arena step of ``optim.FlatAdamW``;
  * ``TrainStep`` splits the step at the only point another GPU is involved (the gradient
    all-reduce), so both halves can be replayed from hipGraphs when the batch shape is fixed.
"""
from __future__ import annotations

import time
from typing import Callable, Optional

import torch

from .graphgym.config import cfg
from .loader import DeviceLoader
from .loss.losses import train_loss
from .optim import FlatAdamW


class TrainStep:
    """forward + loss + backward + gradient pack | [all-reduce] | clip + AdamW.

    ``exchange`` is a ``dp.FlatGradExchange`` (or None on one GPU).  ``salt`` is the device-resident
    dropout salt (``ops.enable_dropout_salt``) a captured step must advance so every replay draws
    new dropout masks."""

    def __init__(self, model: torch.nn.Module, optimizer: FlatAdamW,
                 loss_fn: Optional[Callable] = None, exchange=None, salt=None):
        if not isinstance(optimizer, FlatAdamW):
            raise TypeError("TrainStep drives optim.FlatAdamW (register_optimizer('adamW'))")
        self.model, self.opt = model, optimizer
        self.loss_fn = loss_fn or train_loss
        self.exchange = exchange
        self.salt = salt
        self._g_fb = self._g_up = None
        self._static_loss = None
        self.use_replay = True           # once captured: replay (True) or keep launching eagerly
        self.mode = "eager"

    # -- the three pieces ------------------------------------------------------------------
    def forward_backward(self, batch, zero: bool = True):
        """Returns (loss, pred_score, true); gradients are packed in the arena afterwards."""
        if self.salt is not None:
            self.salt.add_(1)
        if zero:
            self.opt.zero_grad()
        pred, true = self.model(batch)
        loss, pred_score = self.loss_fn(pred, true)
        loss.backward()
        self.opt.pack_grads()
        return loss, pred_score, true

    def reduce(self) -> None:
        if self.exchange is not None:
            self.exchange.all_reduce()

    def update(self) -> None:
        self.opt.step()

    def __call__(self, batch):
        if self._g_fb is not None and self.use_replay:
            return self.replay()
        return self.run_eager(batch)

    def run_eager(self, batch):
        loss, _, _ = self.forward_backward(batch)
        self.reduce()
        self.update()
        return loss

    # -- hipGraph replay for a fixed-shape batch ---------------------------------------------
    def capture(self, make_batch: Callable[[], object], warmup: int = 3) -> None:
        """Capture the step for batches of ONE shape (``make_batch()`` must return a fresh batch
        object over the same device tensors: a shape bucket of a loader, or the synthetic bench
        batch).  One graph when there is no exchange; two (either side of the all-reduce, which
        stays an eager RCCL call on the same stream) when there is."""
        dev = self.opt.arena.device
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(warmup):          # allocator / hipBLASLt heuristics / arena adoption
                self.forward_backward(make_batch())
                self.reduce()
                self.update()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.opt.zero_grad()
        self.opt.sync_hyper()
        split = self.exchange is not None and self.exchange.active
        g_fb = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_fb, capture_error_mode="thread_local"):
            loss, _, _ = self.forward_backward(make_batch())
            if not split:
                self.update()
        g_up = None
        if split:
            g_up = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g_up, capture_error_mode="thread_local"):
                self.update()
        torch.cuda.synchronize(dev)
        self._g_fb, self._g_up, self._static_loss = g_fb, g_up, loss
        self.mode = ("hipGraph replay: [fwd+bwd+pack] -> RCCL all-reduce -> [clip+AdamW]" if split
                     else "hipGraph replay of the whole step")

    def replay(self):
        self.opt.sync_hyper()                # LR schedulers write param_groups[0]['lr']
        self._g_fb.replay()
        if self._g_up is not None:
            self.exchange.all_reduce()
            self._g_up.replay()
        return self._static_loss


def train_epoch(logger, loader, model, optimizer, scheduler, batch_accumulation, exchange=None):
    """Drop-in for ``custom_train.train_epoch`` (custom_train.py:16-47), same arguments (+ the
    optional data-parallel ``exchange``).  ``cfg.optim.clip_grad_norm`` is applied inside the
    fused optimizer step."""
    model.train()
    if cfg.optim.clip_grad_norm:
        optimizer.param_groups[0]["max_grad_norm"] = cfg.optim.clip_grad_norm_value
    else:
        optimizer.param_groups[0]["max_grad_norm"] = None
    step = TrainStep(model, optimizer, exchange=exchange)
    optimizer.zero_grad()
    device = torch.device(cfg.accelerator)
    n_iters = len(loader)
    pending = []
    time_start = time.time()
    # next batch's H2D copies + graph index run on a copy stream behind the current step
    for it, batch in enumerate(DeviceLoader(loader, device)):
        batch.split = 'train'
        loss, pred_score, true = step.forward_backward(batch, zero=False)
        if ((it + 1) % batch_accumulation == 0) or (it + 1 == n_iters):
            step.reduce()
            step.update()
            optimizer.zero_grad()
        pending.append((true, pred_score, loss.detach(), scheduler.get_last_lr()[0],
                        time.time() - time_start))
        time_start = time.time()
    # one D2H at the end of the epoch instead of one sync per iteration
    for true, pred_score, loss, lr, dt in pending:
        if cfg.dataset.name == 'ogbg-code2':
            _true, _pred = true, pred_score
        else:
            _true = true.detach().to('cpu')
            _pred = pred_score.detach().to('cpu')
        logger.update_stats(true=_true, pred=_pred, loss=loss.cpu().item(), lr=lr, time_used=dt,
                            params=cfg.params, dataset_name=cfg.dataset.name)


@torch.no_grad()
def eval_epoch(logger, loader, model, split='val'):
    """Drop-in for ``custom_train.eval_epoch`` (custom_train.py:48-77), same arguments: eval-mode forward + loss
    for every batch of ``loader``, the logger fed once per batch.  As in ``train_epoch`` the next batch's H2D
    copies and graph index run on a copy stream behind the current forward, and the device->host reads the
    logger needs happen after the loop (one sync per epoch instead of one per batch)."""
    model.eval()
    device = torch.device(cfg.accelerator)
    pending = []
    time_start = time.time()
    for batch in DeviceLoader(loader, device):
        batch.split = split
        if cfg.gnn.head == 'inductive_edge':
            pred, true, extra_stats = model(batch)
        else:
            pred, true = model(batch)
            extra_stats = {}
        loss, pred_score = train_loss(pred, true)
        pending.append((true, pred_score, loss.detach(), time.time() - time_start, extra_stats))
        time_start = time.time()
    for true, pred_score, loss, dt, extra_stats in pending:
        if cfg.dataset.name == 'ogbg-code2':
            _true, _pred = true, pred_score
        else:
            _true, _pred = true.detach().to('cpu'), pred_score.detach().to('cpu')
        logger.update_stats(true=_true, pred=_pred, loss=loss.cpu().item(), lr=0, time_used=dt,
                            params=cfg.params, dataset_name=cfg.dataset.name, **extra_stats)
