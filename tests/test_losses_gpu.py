"""Fused segmentation-loss statistics (ss2d_seg_loss_stats / _bwd): against an fp64 torch composition of the same
sums, and -- through train.ce_dice_loss / train.semi_loss -- against the loss values and logit gradients that the
REFERENCE's own loss code produced (tests/golden/losses.npz, made by make_golden.py from code/utils/losses.py composed
as in train_fully_supervised_2D_VIM.py:153-157 and train_Semi_Mamba_UNet.py:414-446)."""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, rel_l2

pytestmark = pytest.mark.gpu


def _stats_ref(logits, labels, C):
    p = torch.softmax(logits.double(), 1)
    oh = (labels.unsqueeze(1) == torch.arange(C).view(1, -1, 1, 1)).double()
    ce = -(torch.log_softmax(logits.double(), 1) * oh).sum()
    return torch.cat([(p * oh).sum((0, 2, 3)), oh.sum((0, 2, 3)), (p * p).sum((0, 2, 3)), ce.view(1)])


@pytest.mark.parametrize("B,C,H,W,dtype,ldt", [(3, 4, 16, 16, torch.float32, torch.uint8), (2, 9, 12, 12, torch.float32, torch.int64),
                                               (2, 3, 5, 7, torch.float32, torch.int32), (1, 16, 8, 8, torch.float32, torch.int64),
                                               (24, 4, 224, 224, torch.float32, torch.uint8), (2, 9, 64, 64, torch.bfloat16, torch.uint8)])
def test_seg_loss_stats_fwd_bwd(B, C, H, W, dtype, ldt):
    from mamba_unet_amd.ss2d_ops import seg_loss_stats
    gen = torch.Generator().manual_seed(B * 100 + C)
    logits = (2.0 * torch.randn(B, C, H, W, generator=gen)).to(dtype).float()
    labels = torch.randint(0, C, (B, H, W), generator=gen)
    gs = torch.randn(3 * C + 1, generator=gen)
    lr = logits.clone().double().requires_grad_(True)
    sr = _stats_ref(lr, labels, C)
    (sr * gs.double()).sum().backward()
    ld = logits.cuda().to(dtype).requires_grad_(True)
    s = seg_loss_stats(ld, labels.to(ldt).cuda(), C)
    assert s.dtype == torch.float32 and s.shape == (3 * C + 1,)
    (s * gs.cuda()).sum().backward()
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert rel_l2(s.detach().cpu(), sr.detach()) < 1e-5            # fp32 sums of values computed from the same stored logits
    assert torch.equal(s[C:2 * C].cpu(), sr[C:2 * C].float())      # label counts are exact
    assert ld.grad.dtype == dtype and rel_l2(ld.grad.float().cpu(), lr.grad) < tol
    # deterministic: fixed summation order, no atomics
    s2 = seg_loss_stats(ld.detach(), labels.to(ldt).cuda(), C)
    assert torch.equal(s, s2)


def test_seg_loss_rejects_bad_arguments():
    from mamba_unet_amd.ss2d_ops import seg_loss_stats
    x = torch.randn(2, 4, 8, 8, device="cuda")
    with pytest.raises(ValueError):
        seg_loss_stats(x, torch.zeros(2, 8, 4, dtype=torch.long, device="cuda"), 4)      # label count mismatch
    with pytest.raises(ValueError):
        seg_loss_stats(torch.randn(1, 17, 4, 4, device="cuda"), torch.zeros(1, 4, 4, dtype=torch.long, device="cuda"), 17)  # > 16 classes
    with pytest.raises(RuntimeError):
        seg_loss_stats(x.cpu(), torch.zeros(2, 8, 8, dtype=torch.long), 4)               # no CPU path


@pytest.mark.parametrize("tag", ["sup4", "sup9"])
def test_supervised_loss_matches_reference_losses(tag):
    from mamba_unet_amd.train import ce_dice_loss
    g = np.load(os.path.join(GOLDEN, "losses.npz"))
    logits = torch.from_numpy(g[tag + ".logits"]).cuda().requires_grad_(True)
    labels = torch.from_numpy(g[tag + ".labels"]).cuda()
    ce, dice = ce_dice_loss(logits, labels, logits.shape[1])
    loss = 0.5 * (dice + ce)
    loss.backward()
    assert abs(float(loss.detach()) - float(g[tag + ".loss"])) < 1e-5 * abs(float(g[tag + ".loss"]))
    assert rel_l2(logits.grad.cpu(), g[tag + ".dlogits"]) < 1e-5


def test_semi_loss_matches_reference_composition():
    from mamba_unet_amd.train import semi_loss
    g = np.load(os.path.join(GOLDEN, "losses.npz"))
    o1 = torch.from_numpy(g["semi.o1"]).cuda().requires_grad_(True)
    o2 = torch.from_numpy(g["semi.o2"]).cuda().requires_grad_(True)
    labels = torch.from_numpy(g["semi.labels"]).cuda()
    loss = semi_loss(o1, o2, labels, int(g["semi.labeled_bs"]), 4, float(g["semi.consistency_weight"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["semi.loss"])) < 1e-5 * abs(float(g["semi.loss"]))
    assert rel_l2(o1.grad.cpu(), g["semi.do1"]) < 1e-5 and rel_l2(o2.grad.cpu(), g["semi.do2"]) < 1e-5
