"""Training-step pieces of the reference's VIM scripts that the benchmark configurations time
(callers of the hot path).  Plain torch; no I/O, logging or validation.

  DiceLoss          utils/losses.py:179-215      ConstraLoss   utils/losses.py:16-28
  supervised step   train_fully_supervised_2D_VIM.py:147-164 (SGD .01/.9/1e-4, 0.5*(CE+Dice), poly LR)
  semi step         train_Semi_Mamba_UNet.py:411-456 (cross pseudo supervision + ConstraLoss)
"""
from __future__ import annotations

import os as _os

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


class CapturableSGD:
    """torch.optim.SGD(momentum, weight_decay; dampening 0, no nesterov) with the learning rate in a 0-dim device
    tensor and no host read of it, so the update can sit inside a captured HIP graph (torch's foreach SGD calls
    .item() on a tensor lr).  Same arithmetic: d = g + wd*p; buf = d (first step) or mu*buf + d; p -= lr*buf."""

    def __init__(self, params, lr, momentum=0.9, weight_decay=1e-4, device=None):
        self.params = [p for p in params if p.requires_grad]
        device = device or self.params[0].device
        self.param_groups = [{"lr": torch.tensor(float(lr), device=device), "momentum": momentum,
                              "weight_decay": weight_decay, "params": self.params}]
        self.bufs = None

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    @torch.no_grad()
    def step(self):
        g = self.param_groups[0]
        ps = [p for p in self.params if p.grad is not None]
        d = torch._foreach_add([p.grad for p in ps], ps, alpha=g["weight_decay"])
        if self.bufs is None:
            self.bufs = {id(p): t.clone() for p, t in zip(ps, d)}
            bufs = [self.bufs[id(p)] for p in ps]
        else:
            bufs = [self.bufs[id(p)] for p in ps]
            torch._foreach_mul_(bufs, g["momentum"])
            torch._foreach_add_(bufs, d)
        torch._foreach_sub_(ps, torch._foreach_mul(bufs, g["lr"]))


def make_sgd(params, lr=0.01, tensor_lr_device=None):
    """SGD of train_fully_supervised_2D_VIM.py:117-118.  With `tensor_lr_device` the learning rate lives in a
    0-dim device tensor so that a HIP-graph-captured step (GraphedSupervisedStep) sees the poly-LR updates."""
    if tensor_lr_device is not None:
        return CapturableSGD(params, lr, momentum=0.9, weight_decay=1e-4, device=tensor_lr_device)
    params = list(params)
    # same update rule; on GPU parameters torch's fused implementation does it in one multi-tensor pass instead of three
    fused = bool(params) and all(p.is_cuda for p in params) and not _os.environ.get("SS2D_NO_FUSED_SGD")
    return torch.optim.SGD(params, lr=lr, momentum=0.9, weight_decay=1e-4, fused=fused)


def set_lr(optimizer, lr: float):
    for g in optimizer.param_groups:
        if isinstance(g["lr"], torch.Tensor):
            g["lr"].fill_(lr)
        else:
            g["lr"] = lr


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
    set_lr(optimizer, poly_lr(base_lr, it, max_it))
    return loss.detach()


class GraphedSupervisedStep:
    """The whole supervised iteration (forward, loss, backward, SGD update) captured once in a HIP graph and
    replayed (one launch per step instead of ~1150).  Single process / single GPU (DDP's bucketed all-reduce is not
    captured here).  Inputs are copied into static buffers; the learning rate must be a device tensor
    (make_sgd(..., tensor_lr_device=dev)).

    NOT USABLE on the ROCm 7.0 / PyTorch 2.10 stack of this image, and refused unless SS2D_ALLOW_HIP_GRAPH=1: from
    the second replay on, torch's multi-block reduction kernels (`x.sum((0, 2, 3))` of the Dice statistics, the row
    sums of bias gradients) return wrong sums -- the first of several reductions in a graph is right, the later
    ones are not (scratch/graph_dice.py isolates it: no kernel of this package involved; every op of this package
    replays bit-exactly, scratch/graph_ops.py).  The training loss then drifts and goes NaN within ~6 replays.
    The eager step is GPU-bound since the fusions (26.8 ms eager vs 26.7 ms replayed), so nothing is lost."""

    def __init__(self, model, optimizer, images, labels, n_classes, max_it, base_lr=0.01, autocast_bf16=False, warmup=3):
        import os
        if not os.environ.get("SS2D_ALLOW_HIP_GRAPH"):
            raise RuntimeError("GraphedSupervisedStep: HIP-graph replay of torch's reduction kernels returns wrong sums on "
                               "this ROCm/PyTorch stack (see the class docstring); use supervised_step, or set "
                               "SS2D_ALLOW_HIP_GRAPH=1 to experiment")
        self.opt, self.max_it, self.base_lr = optimizer, max_it, base_lr
        self.images, self.labels = images.clone(), labels.clone()
        args = (model, optimizer, self.images, self.labels, n_classes)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for i in range(warmup):   # allocator, MIOpen / hipBLASLt heuristics and autograd warm up eagerly
                supervised_step(*args, i, max_it, base_lr, autocast_bf16)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = supervised_step(*args, warmup, max_it, base_lr, autocast_bf16)

    def __call__(self, images, labels, it):
        if images is not self.images:
            self.images.copy_(images)
            self.labels.copy_(labels)
        self.graph.replay()
        set_lr(self.opt, poly_lr(self.base_lr, it, self.max_it))
        return self.loss


class GraphedForward:
    """Eval-mode forward captured once in a HIP graph and replayed: at batch 1 the forward is ~500 launches and
    host-bound (6.0 ms for a 224x224 slice), a replay is one launch.  Unlike the training step, the forward contains
    none of torch's multi-block reductions, and every kernel of this package replays bit-exactly; the constructor
    nevertheless replays twice and checks both results against the eager forward before it returns."""

    def __init__(self, model, example, autocast_bf16=False, warmup=2):
        self.inp = example.clone()
        model.eval()

        def fwd():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast_bf16):
                return model(self.inp)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                ref = fwd()
        torch.cuda.current_stream().wait_stream(side)
        ref = ref.clone()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = fwd()
        for _ in range(2):
            self.out.zero_()
            self.graph.replay()
            if not torch.equal(self.out, ref):
                raise RuntimeError("GraphedForward: a replay does not reproduce the eager forward on this stack")

    def __call__(self, x):
        if x is not self.inp:
            self.inp.copy_(x)
        self.graph.replay()
        return self.out


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
    for opt in (opt1, opt2):
        set_lr(opt, poly_lr(base_lr, it, max_it))
    return loss.detach()
