"""Training-step pieces of the reference's VIM scripts that the benchmark configurations time
(callers of the hot path).  Plain torch; no I/O, logging or validation.

  DiceLoss          utils/losses.py:179-215      ConstraLoss   utils/losses.py:16-28
  supervised step   train_fully_supervised_2D_VIM.py:147-164 (SGD .01/.9/1e-4, 0.5*(CE+Dice), poly LR)
  semi step         train_Semi_Mamba_UNet.py:411-456 (cross pseudo supervision + ConstraLoss)
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from .dist import all_reduce_sum


def dice_loss(probs: torch.Tensor, target: torch.Tensor, n_classes: int, softmax: bool = False,
              global_batch: bool = True) -> torch.Tensor:
    """probs (B, C, H, W), target (B, 1, H, W) integer labels.  Per class 1 - (2<p,t>+eps)/(<p,p>+<t,t>+eps)
    with the sums taken over the WHOLE batch, averaged over classes (losses.py:192-215).  With
    `global_batch` the 3*C partial sums are all-reduced across ranks so a sharded batch gives the
    single-process loss."""
    if softmax:
        probs = torch.softmax(probs, dim=1)
    onehot = (target == torch.arange(n_classes, device=target.device).view(1, -1, 1, 1)).to(probs.dtype)
    if probs.shape != onehot.shape:
        raise ValueError("predict & target shape do not match")
    dims = (0, 2, 3)
    stats = torch.stack([(probs * onehot).sum(dims), (onehot * onehot).sum(dims), (probs * probs).sum(dims)])
    if global_batch:
        stats = all_reduce_sum(stats)
    inter, ysum, zsum = stats
    smooth = 1e-5
    return (1.0 - (2 * inter + smooth) / (zsum + ysum + smooth)).mean()


def constra_loss(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Global-average-pool, L2-normalise over channels, mean squared difference (losses.py:16-28)."""
    pa = F.normalize(a.mean(dim=(2, 3)), p=2, dim=1)
    pb = F.normalize(b.mean(dim=(2, 3)), p=2, dim=1)
    return ((pa - pb) ** 2).mean()


def poly_lr(base_lr: float, it: int, max_it: int) -> float:
    return base_lr * (1.0 - it / max_it) ** 0.9


def make_sgd(params, lr=0.01):
    return torch.optim.SGD(params, lr=lr, momentum=0.9, weight_decay=1e-4)


def supervised_step(model, optimizer, images, labels, n_classes, it, max_it, base_lr=0.01, autocast_bf16=False):
    """One iteration of train_fully_supervised_2D_VIM.py:150-164.  Returns the (detached) loss tensor; no
    .item() here: the reference syncs every iteration for tensorboard (:172-174), the caller decides."""
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast_bf16):
        out = model(images)
    out = out.float()
    loss_ce = F.cross_entropy(out, labels.long())
    loss_dice = dice_loss(torch.softmax(out, dim=1), labels.unsqueeze(1), n_classes)
    loss = 0.5 * (loss_dice + loss_ce)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    lr = poly_lr(base_lr, it, max_it)
    for g in optimizer.param_groups:
        g["lr"] = lr
    return loss.detach()


def semi_step(model1, model2, opt1, opt2, images, labels, labeled_bs, n_classes, it, max_it, consistency_weight,
              base_lr=0.01):
    """Cross pseudo supervision between two networks + ConstraLoss (train_Semi_Mamba_UNet.py:411-456)."""
    o1, o2 = model1(images).float(), model2(images).float()
    s1, s2 = torch.softmax(o1, dim=1), torch.softmax(o2, dim=1)
    lab = labels[:labeled_bs]
    sup1 = 0.5 * (F.cross_entropy(o1[:labeled_bs], lab.long()) + dice_loss(s1[:labeled_bs], lab.unsqueeze(1), n_classes))
    sup2 = 0.5 * (F.cross_entropy(o2[:labeled_bs], lab.long()) + dice_loss(s2[:labeled_bs], lab.unsqueeze(1), n_classes))
    pseudo1 = torch.argmax(s1[labeled_bs:].detach(), dim=1)
    pseudo2 = torch.argmax(s2[labeled_bs:].detach(), dim=1)
    ps1 = dice_loss(s1[labeled_bs:], pseudo2.unsqueeze(1), n_classes)
    ps2 = dice_loss(s2[labeled_bs:], pseudo1.unsqueeze(1), n_classes)
    con = constra_loss(o1, o2)
    loss = (sup1 + consistency_weight * ps1 + 0.5 * con) + (sup2 + consistency_weight * ps2 + 0.5 * con)
    opt1.zero_grad(set_to_none=True)
    opt2.zero_grad(set_to_none=True)
    loss.backward()
    opt1.step()
    opt2.step()
    lr = poly_lr(base_lr, it, max_it)
    for opt in (opt1, opt2):
        for g in opt.param_groups:
            g["lr"] = lr
    return loss.detach()
