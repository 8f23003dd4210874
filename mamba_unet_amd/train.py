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
from .ss2d_ops import seg_loss_stats


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


def ce_dice_from_stats(stats: torch.Tensor, n_pixels: int, n_classes: int, global_batch: bool = True):
    """(mean cross entropy, Dice loss) from the (3C+1) statistics of `seg_loss_stats`: the same numbers as
    nn.CrossEntropyLoss()(logits, y) and DiceLoss(C)(softmax(logits), y) (losses.py:192-215), of which only 13 (C = 4)
    touch torch.  With `global_batch` the Dice sums are all-reduced across ranks (as in `dice_loss`); the cross entropy
    stays the rank's own mean, which DistributedDataParallel's gradient average turns into the global mean."""
    C = n_classes
    ce = stats[3 * C] / n_pixels
    if global_batch:
        stats = all_reduce_sum(stats)
    inter, ysum, zsum = stats[:C], stats[C:2 * C], stats[2 * C:3 * C]
    smooth = 1e-5
    return ce, (1.0 - (2 * inter + smooth) / (zsum + ysum + smooth)).mean()


def ce_dice_loss(logits: torch.Tensor, labels: torch.Tensor, n_classes: int, global_batch: bool = True):
    """(CE, Dice) of `logits` (B, C, H, W) against integer `labels` (B, H, W) through the fused statistics kernel."""
    stats = seg_loss_stats(logits, labels, n_classes)
    return ce_dice_from_stats(stats, logits.numel() // n_classes, n_classes, global_batch)


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
        self.bufs = {}
        self._post_hooks = {}

    def register_step_post_hook(self, hook):
        """hook(optimizer, args, kwargs) after every step(), as torch.optim.Optimizer has it; returns a handle with .remove()."""
        key = object()
        self._post_hooks[key] = hook

        class _Handle:
            def remove(_self):
                self._post_hooks.pop(key, None)
        return _Handle()

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
        # momentum buffers are created per parameter at its first gradient, as torch.optim.SGD does
        old = [(p, t) for p, t in zip(ps, d) if id(p) in self.bufs]
        for p, t in zip(ps, d):
            if id(p) not in self.bufs:
                self.bufs[id(p)] = t.clone()
        if old:
            ob = [self.bufs[id(p)] for p, _ in old]
            torch._foreach_mul_(ob, g["momentum"])
            torch._foreach_add_(ob, [t for _, t in old])
        bufs = [self.bufs[id(p)] for p in ps]
        torch._foreach_sub_(ps, torch._foreach_mul(bufs, g["lr"]))
        for hook in list(self._post_hooks.values()):
            hook(self, (), {})

    def state_dict(self):
        """Momentum buffers by parameter position (the order of the constructor's `params`) and the learning rate."""
        return {"lr": float(self.param_groups[0]["lr"]), "momentum_buffers": {i: self.bufs[id(p)].clone()
                                                                              for i, p in enumerate(self.params) if id(p) in self.bufs}}

    def load_state_dict(self, sd):
        self.param_groups[0]["lr"].fill_(sd["lr"])
        self.bufs = {id(self.params[int(i)]): t.clone().to(self.params[int(i)].device) for i, t in sd["momentum_buffers"].items()}


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
    loss_ce, loss_dice = ce_dice_loss(out, labels, n_classes)     # softmax, CE and the Dice sums: one pass over the logits
    loss = 0.5 * (loss_dice + loss_ce)
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    set_lr(optimizer, poly_lr(base_lr, it, max_it))
    return loss.detach()


class WeightCopies:
    """bf16 working copies of a model's Linear weights for autocast training (fp32 master weights stay the parameters):
    `WeightCopies(model, optimizer)` makes the copies, refreshes them after every `optimizer.step()` (a step post-hook:
    ONE multi-tensor cast instead of a cast kernel per Linear per step, 42 of them in MambaUNet) and registers them with
    ss2d_ops.cast_weight, which the Linear functions of this package consult.  A copy is ignored as soon as its
    parameter's version counter moves without a refresh, so `load_state_dict` or a second optimizer cannot make it stale;
    an update through `.data` (the EMA idiom of the reference's mean-teacher scripts) does not move the counter -- do not
    attach WeightCopies to a model updated that way, or call refresh() after the update.  close() unregisters."""

    def __init__(self, model, optimizer=None, dtype=torch.bfloat16):
        seen, self.params = set(), []
        for m in model.modules():
            if type(m) is torch.nn.Linear and m.weight.dtype == torch.float32 and id(m.weight) not in seen:
                seen.add(id(m.weight))
                self.params.append(m.weight)
        self.copies = [torch.empty_like(p, dtype=dtype) for p in self.params]
        self.hook = optimizer.register_step_post_hook(lambda *_: self.refresh()) if optimizer is not None else None
        self.refresh()

    def refresh(self):
        import weakref
        from . import ss2d_ops
        if not self.params:
            return
        with torch.no_grad():
            torch._foreach_copy_(self.copies, self.params)
        for p, c in zip(self.params, self.copies):
            ss2d_ops._WORKING_COPIES[id(p)] = (weakref.ref(p), c, p._version)

    def close(self):
        from . import ss2d_ops
        if self.hook is not None:
            self.hook.remove()
            self.hook = None
        for p in self.params:
            ss2d_ops._WORKING_COPIES.pop(id(p), None)
        self.params, self.copies = [], []


class GraphedSupervisedStep:
    """The whole supervised iteration (forward, loss, backward, SGD update) captured once in a HIP graph and
    replayed (one launch per step instead of ~1150).  Single process / single GPU (DDP's bucketed all-reduce is not
    captured here).  Inputs are copied into static buffers; the learning rate must be a device tensor
    (make_sgd(..., tensor_lr_device=dev)).

    History: on this ROCm 7.0 / PyTorch 2.10 stack torch's multi-block reduction kernels return wrong sums from the
    second replay of a captured graph on (tools/graph_dice_repro.py shows it with torch ops alone: the second and third
    `x.sum((0, 2, 3))` of a graph go wrong).  The step used to contain two kinds of them -- the Dice statistics and the
    bias-gradient row sums of the skip-connection Linears -- and its replays drifted to NaN.  Both are kernels of this
    package now (ss2d_seg_loss_stats, ss2d_colsum), and ten replays reproduce ten eager steps to the last bit of the
    loss (tests/test_configs_gpu.py::test_graphed_supervised_step_matches_eager, tools/graph_vs_eager.py)."""

    def __init__(self, model, optimizer, images, labels, n_classes, max_it, base_lr=0.01, autocast_bf16=False, warmup=3):
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


def semi_loss(o1, o2, labels, labeled_bs, n_classes, consistency_weight):
    """The loss of train_Semi_Mamba_UNet.py:414-446 from the two networks' logits: supervised 0.5*(CE + Dice) on the
    labelled slices, Dice of each network's softmax against the OTHER network's argmax on the unlabelled slices
    (argmax of the softmax = argmax of the logits), and ConstraLoss of the logits, counted once for each network."""
    lab = labels[:labeled_bs]
    pseudo1 = torch.argmax(o1[labeled_bs:].detach(), dim=1)
    pseudo2 = torch.argmax(o2[labeled_bs:].detach(), dim=1)
    C = n_classes
    # the four statistics vectors travel in ONE all-reduce (forward) and one in backward instead of four each; the
    # cross entropies are means over the rank's own pixels (DDP's gradient average makes them global), taken before it
    local = torch.stack([seg_loss_stats(o1[:labeled_bs], lab, C), seg_loss_stats(o2[:labeled_bs], lab, C),
                         seg_loss_stats(o1[labeled_bs:], pseudo2, C), seg_loss_stats(o2[labeled_bs:], pseudo1, C)])
    n_lab = o1[:labeled_bs].numel() // C
    ce1, ce2 = local[0, 3 * C] / n_lab, local[1, 3 * C] / n_lab
    stats = all_reduce_sum(local)
    inter, ysum, zsum = stats[:, :C], stats[:, C:2 * C], stats[:, 2 * C:3 * C]
    d1, d2, ps1, ps2 = (1.0 - (2 * inter + 1e-5) / (zsum + ysum + 1e-5)).mean(dim=1)
    con = constra_loss(o1.float(), o2.float())
    return (0.5 * (ce1 + d1) + consistency_weight * ps1 + 0.5 * con) + (0.5 * (ce2 + d2) + consistency_weight * ps2 + 0.5 * con)


def semi_step(model1, model2, opt1, opt2, images, labels, labeled_bs, n_classes, it, max_it, consistency_weight,
              base_lr=0.01, autocast_bf16=False):
    """Cross pseudo supervision between two networks + ConstraLoss (train_Semi_Mamba_UNet.py:411-463).  `it` is the
    0-based iteration: the reference increments iter_num BEFORE it evaluates the poly schedule (:458-461), so the
    learning rate set here is poly_lr(it + 1) (the supervised script evaluates it before the increment)."""
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast_bf16):
        o1, o2 = model1(images), model2(images)
    loss = semi_loss(o1, o2, labels, labeled_bs, n_classes, consistency_weight)
    opt1.zero_grad(set_to_none=True)
    opt2.zero_grad(set_to_none=True)
    loss.backward()
    opt1.step()
    opt2.step()
    for opt in (opt1, opt2):
        set_lr(opt, poly_lr(base_lr, it + 1, max_it))
    return loss.detach()
