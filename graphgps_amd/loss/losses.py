"""The registered losses of the reference and GraphGym's dispatcher over them (plain torch, after the head):
L1 / smooth-L1 (``/root/reference/graphgps/loss/l1.py:6-15``), the code2 sub-token cross-entropy
(``graphgps/loss/subtoken_prediction_loss.py:6-20``: mean over the 5 positions of CE(pred_list[i], y_arr[:, i])),
class-frequency weighted cross-entropy (``graphgps/loss/weighted_cross_entropy.py:7-30``), multilabel BCE with
NaN targets filtered (``graphgps/loss/multilabel_classification_loss.py:6-17``); ``compute_loss`` = GraphGym's
``compute_loss`` (PyG 2.2, third-party): registered losses first, then its built-in cross-entropy / MSE."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..graphgym.config import cfg
from ..graphgym.register import loss_dict, register_loss


@register_loss('l1_losses', overwrite=True)
def l1_losses(pred, true):
    if cfg.model.loss_fun == 'l1':
        return nn.L1Loss()(pred, true), pred
    if cfg.model.loss_fun == 'smoothl1':
        return nn.SmoothL1Loss()(pred, true), pred


@register_loss('subtoken_cross_entropy', overwrite=True)
def subtoken_cross_entropy(pred_list, true):
    if cfg.dataset.task_type == 'subtoken_prediction':
        if cfg.model.loss_fun != 'cross_entropy':
            raise ValueError("Only 'cross_entropy' loss_fun supported with 'subtoken_prediction' "
                             "task_type.")
        loss = 0
        for i in range(len(pred_list)):
            loss = loss + F.cross_entropy(pred_list[i].to(torch.float32), true['y_arr'][:, i])
        return loss / len(pred_list), pred_list


@register_loss('weighted_cross_entropy', overwrite=True)
def weighted_cross_entropy(pred, true):
    """Each class weighted by the fraction of nodes NOT in it (absent classes get weight 0)."""
    if cfg.model.loss_fun == 'weighted_cross_entropy':
        total = true.size(0)
        n_classes = pred.shape[1] if pred.ndim > 1 else 2
        sizes = torch.bincount(true, minlength=n_classes)[:n_classes]
        weight = (total - sizes).float() / total * (sizes > 0).float()
        if pred.ndim > 1:                                        # multiclass
            pred = F.log_softmax(pred, dim=-1)
            return F.nll_loss(pred, true, weight=weight), pred
        loss = F.binary_cross_entropy_with_logits(pred, true.float(), weight=weight[true])
        return loss, torch.sigmoid(pred)


@register_loss('multilabel_cross_entropy', overwrite=True)
def multilabel_cross_entropy(pred, true):
    if cfg.dataset.task_type == 'classification_multilabel':
        if cfg.model.loss_fun != 'cross_entropy':
            raise ValueError("Only 'cross_entropy' loss_fun supported with "
                             "'classification_multilabel' task_type.")
        labeled = true == true                                   # NaN targets are unlabeled
        return nn.BCEWithLogitsLoss()(pred[labeled], true[labeled].float()), pred


def compute_loss(pred, true):
    """GraphGym ``compute_loss`` (third-party), as called at graphgps/train/custom_train.py:29:
    squeeze the trailing dim of pred/true, try every registered loss, then the built-ins.
    (ogbg-code2 bypasses it and calls ``subtoken_cross_entropy`` directly, custom_train.py:24-25.)"""
    pred = pred.squeeze(-1) if pred.ndim > 1 else pred
    true = true.squeeze(-1) if true.ndim > 1 else true
    for fn in loss_dict.values():
        out = fn(pred, true)
        if out is not None:
            return out
    reduction = cfg.model.size_average
    if cfg.model.loss_fun == 'cross_entropy':
        if pred.ndim > 1 and true.ndim == 1:                     # multiclass
            pred = F.log_softmax(pred, dim=-1)
            return F.nll_loss(pred, true), pred
        true = true.float()                                      # binary or multilabel
        return nn.BCEWithLogitsLoss(reduction=reduction)(pred, true), torch.sigmoid(pred)
    if cfg.model.loss_fun == 'mse':
        return nn.MSELoss(reduction=reduction)(pred, true.float()), pred
    raise ValueError(f"Loss function '{cfg.model.loss_fun}' not supported")


def train_loss(pred, true):
    """The loss dispatch of the reference's train_epoch (graphgps/train/custom_train.py:24-29):
    ogbg-code2 calls the sub-token CE directly, everything else goes through compute_loss."""
    if cfg.dataset.name == 'ogbg-code2':
        return subtoken_cross_entropy(pred, true)
    return compute_loss(pred, true)
