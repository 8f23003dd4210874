"""CPU tests of the host side: C-ABI library loads and exports every declared symbol, module
construction reproduces the reference's init for a given seed, state-dict keys match."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT


def test_library_exports_every_symbol_declared_in_header():
    from mamba_unet_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "ss2d.h")).read()
    declared = set(re.findall(r"\b(ss2d_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTED_SYMBOLS), declared ^ set(_lib.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    loaded = _lib.load()  # argtypes set, ABI version checked; no compute call without a GPU
    assert loaded.ss2d_abi_version() == 1
    assert loaded.ss2d_scan_num_chunks(3136) == -(-3136 // 128) or loaded.ss2d_scan_num_chunks(3136) > 0
    assert loaded.ss2d_scan_ckpt_bytes(2, 8, 100, 16) == 2 * 8 * loaded.ss2d_scan_num_chunks(100) * 16 * 4


def test_argument_validation_without_gpu():
    from mamba_unet_amd import _lib
    lib = _lib.load()
    rc = lib.ss2d_selective_scan_fwd(0, None, None, None, None, None, None, None, None, None, None, None,
                                     1, 8, 16, 16, 4, 1, None)
    assert rc < 0 and b"null" in lib.ss2d_last_error()
    rc = lib.ss2d_selective_scan_fwd(5, None, None, None, None, None, None, None, None, None, None, None,
                                     1, 8, 16, 16, 4, 1, None)
    assert rc < 0 and b"dtype" in lib.ss2d_last_error()


def test_cpu_tensors_are_rejected_loudly():
    from mamba_unet_amd import cross_scan, selective_scan_fn
    with pytest.raises(RuntimeError):
        cross_scan(torch.zeros(1, 2, 3, 3))
    with pytest.raises(RuntimeError):
        selective_scan_fn(torch.zeros(1, 4, 8), torch.zeros(1, 4, 8), -torch.ones(4, 2), torch.zeros(1, 2, 8),
                          torch.zeros(1, 2, 8))


def test_vssm_small_init_matches_reference_for_same_seed():
    """Same seed -> same parameters as the reference's VSSM (module order and RNG consumption mirrored)."""
    from mamba_unet_amd import VSSM
    g = np.load(os.path.join(GOLDEN, "vssm_small.npz"))
    torch.manual_seed(1337)
    m = VSSM(patch_size=4, in_chans=3, num_classes=4, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128], drop_path_rate=0.2)
    sd = m.state_dict()
    ref_keys = [k[3:] for k in g.files if k.startswith("sd.")]
    assert list(sd.keys()) == ref_keys
    for k in ref_keys:
        assert np.array_equal(sd[k].numpy(), g["sd." + k]), k


def test_vssm_tiny_init_summary_matches_reference():
    from mamba_unet_amd import VSSM
    g = np.load(os.path.join(GOLDEN, "vssm_tiny_init_summary.npz"))
    torch.manual_seed(1337)
    m = VSSM(patch_size=4, in_chans=3, num_classes=4, depths=[2, 2, 2, 2], drop_path_rate=0.2)
    assert sum(p.numel() for p in m.parameters()) == int(g["n_params"]) == 19121664
    sd = m.state_dict()
    names = sorted(sd)
    assert names == list(g["names"])
    sums = np.array([float(sd[k].double().sum()) for k in names])
    abss = np.array([float(sd[k].double().abs().sum()) for k in names])
    assert np.allclose(sums, g["sums"], rtol=0, atol=1e-9) and np.allclose(abss, g["abs_sums"], rtol=0, atol=1e-9)


def test_mambaunet_wrapper_and_state_dict_roundtrip(tmp_path):
    from mamba_unet_amd import MambaUnet, MambaUnetConfig
    cfg = MambaUnetConfig(depths=[1, 1, 1, 1])
    torch.manual_seed(0)
    a = MambaUnet(cfg, img_size=64, num_classes=4)
    # a "pretrained" checkpoint in the upstream format: {"model": encoder-only state dict}
    enc = {k: v for k, v in a.mamba_unet.state_dict().items() if k.startswith(("patch_embed", "layers."))}
    path = os.path.join(tmp_path, "ckpt.pth")
    torch.save({"model": enc}, path)
    torch.manual_seed(1)
    b = MambaUnet(MambaUnetConfig(depths=[1, 1, 1, 1], pretrain_ckpt=path), img_size=64, num_classes=4)
    b.load_from()
    sa, sb = a.mamba_unet.state_dict(), b.mamba_unet.state_dict()
    for k in enc:
        assert torch.equal(sa[k], sb[k])
    # encoder stage N also seeds decoder stage 3-N where shapes agree (vision_mamba.py:69-80)
    k_up = "layers_up.1.blocks.0.self_attention.A_logs"
    assert torch.equal(sb[k_up], sa["layers.2.blocks.0.self_attention.A_logs"])


def test_capturable_sgd_matches_torch_sgd():
    """CapturableSGD (tensor lr, graph-safe) == torch.optim.SGD(momentum .9, wd 1e-4) over several steps."""
    import torch
    from mamba_unet_amd.train import CapturableSGD, set_lr
    torch.manual_seed(0)
    a = [torch.randn(5, 3, requires_grad=True), torch.randn(7, requires_grad=True)]
    b = [t.detach().clone().requires_grad_(True) for t in a]
    o1 = torch.optim.SGD(a, lr=0.01, momentum=0.9, weight_decay=1e-4)
    o2 = CapturableSGD(b, lr=0.01, momentum=0.9, weight_decay=1e-4)
    for it in range(4):
        for ps, o in ((a, o1), (b, o2)):
            o.zero_grad(set_to_none=True)
            (sum((p ** 2).sum() * (k + 1) for k, p in enumerate(ps)) + ps[0].sin().sum()).backward()
            o.step()
            set_lr(o, 0.01 * (1 - it / 10) ** 0.9)
    for x, y in zip(a, b):
        assert torch.allclose(x, y, rtol=1e-6, atol=1e-7)


def test_unet_model1_matches_reference_keys_and_output():
    """UNet (model1 of the semi configuration): same state-dict keys / shapes and, from the same seed, the same
    eval-mode output as the reference's networks/unet.py (tests/golden/unet_ref.npz)."""
    import numpy as np
    from mamba_unet_amd.unet import UNet
    g = np.load(os.path.join(GOLDEN, "unet_ref.npz"))
    torch.manual_seed(5)
    m = UNet(in_chns=1, class_num=4).eval()
    sd = m.state_dict()
    assert list(sd) == list(g["names"]) and [str(tuple(v.shape)) for v in sd.values()] == list(g["shapes"])
    with torch.no_grad():
        y = m(torch.from_numpy(g["x"]))
    assert torch.allclose(y, torch.from_numpy(g["y"]), atol=1e-6)


def test_ce_dice_from_stats_reproduces_reference_losses():
    """The 13-number half of the fused loss (train.ce_dice_from_stats) on statistics summed here in torch, against the
    loss values of the reference's own DiceLoss + CrossEntropyLoss (tests/golden/losses.npz)."""
    import numpy as np
    from mamba_unet_amd.train import ce_dice_from_stats
    g = np.load(os.path.join(GOLDEN, "losses.npz"))
    for tag in ("sup4", "sup9"):
        logits, labels = torch.from_numpy(g[tag + ".logits"]).double(), torch.from_numpy(g[tag + ".labels"]).long()
        C = logits.shape[1]
        p = torch.softmax(logits, 1)
        oh = (labels.unsqueeze(1) == torch.arange(C).view(1, -1, 1, 1)).double()
        ce = -(torch.log_softmax(logits, 1) * oh).sum()
        stats = torch.cat([(p * oh).sum((0, 2, 3)), oh.sum((0, 2, 3)), (p * p).sum((0, 2, 3)), ce.view(1)])
        ce_m, dice = ce_dice_from_stats(stats, logits.numel() // C, C, global_batch=False)
        assert abs(float(0.5 * (ce_m + dice)) - float(g[tag + ".loss"])) < 1e-6


def test_weight_copies_follow_the_optimizer_and_never_go_stale():
    """train.WeightCopies: the bf16 working copy is what ss2d_ops.cast_weight returns right after a refresh, is refreshed by
    the optimizer's step hook, and is ignored (fresh cast instead) after any other in-place update of the parameter."""
    import torch
    from mamba_unet_amd import ss2d_ops
    from mamba_unet_amd.train import WeightCopies
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 8, bias=False), torch.nn.ReLU(), torch.nn.Linear(8, 4))
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    wc = WeightCopies(net, opt)
    w = net[0].weight
    c0 = ss2d_ops.cast_weight(w, torch.bfloat16)
    assert c0 is wc.copies[0] and torch.equal(c0, w.detach().to(torch.bfloat16))
    assert ss2d_ops.cast_weight(w, torch.float32) is w
    net(torch.randn(5, 16)).sum().backward()
    opt.step()                                                  # hook: refreshed in place
    c1 = ss2d_ops.cast_weight(w, torch.bfloat16)
    assert c1 is wc.copies[0] and torch.equal(c1, w.detach().to(torch.bfloat16))
    with torch.no_grad():
        w.mul_(2.0)                                             # an update the hook does not see: the copy must not be used
    c2 = ss2d_ops.cast_weight(w, torch.bfloat16)
    assert c2 is not wc.copies[0] and torch.equal(c2, w.detach().to(torch.bfloat16))
    wc.refresh()
    assert ss2d_ops.cast_weight(w, torch.bfloat16) is wc.copies[0]
    wc.close()
    assert ss2d_ops.cast_weight(w, torch.bfloat16) is not c0 and id(w) not in ss2d_ops._WORKING_COPIES
