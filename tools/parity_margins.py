#!/usr/bin/env python3
"""Measured errors behind the model-level parity gates (tests/test_ss2d_gpu.py, tests/test_configs_gpu.py): the SS2D and VSSM
fixtures made from the imported reference (tests/golden/make_golden.py), fp32.  The gates sit at ~10x these figures."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_l2, rel_max  # noqa: E402
from mamba_unet_amd import SS2D, VSSM  # noqa: E402
GOLDEN = os.path.join(ROOT, "tests", "golden")
for name in ["ss2d_2x8x5x7_s0.npz", "ss2d_1x16x8x8_s1.npz", "ss2d_2x96x14x14_s0.npz"]:
    g = np.load(os.path.join(GOLDEN, name))
    m = SS2D(d_model=g["x"].shape[-1])
    m.load_state_dict({k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("sd.")})
    m = m.cuda()
    x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
    core = m.forward_corev0(torch.from_numpy(g["core_in"]).cuda())
    y = m(x)
    (y * torch.from_numpy(g["g"]).cuda()).sum().backward()
    worst = max((rel_l2(p.grad.cpu(), g["grad." + k]), k) for k, p in m.named_parameters())
    print("%-24s core max %.1e  y max %.1e l2 %.1e  dx l2 %.1e  worst parameter gradient l2 %.1e (%s)" % (
        name, rel_max(core.detach().cpu(), g["core_out"]), rel_max(y.detach().cpu(), g["y"]), rel_l2(y.detach().cpu(), g["y"]),
        rel_l2(x.grad.cpu(), g["dx"]), worst[0], worst[1]))
sd = np.load(os.path.join(GOLDEN, "vssm_small.npz"))
g = np.load(os.path.join(GOLDEN, "vssm_small_grads.npz"))
m = VSSM(patch_size=4, in_chans=3, num_classes=4, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128], drop_path_rate=0.2)
m.load_state_dict({k[3:]: torch.from_numpy(sd[k]) for k in sd.files if k.startswith("sd.")})
m = m.cuda().eval()
with torch.no_grad():
    y0 = m(torch.from_numpy(sd["x"]).cuda())
print("vssm_small logits        max %.1e l2 %.1e" % (rel_max(y0.cpu(), sd["logits"]), rel_l2(y0.cpu(), sd["logits"])))
x = torch.from_numpy(g["x"]).cuda().requires_grad_(True)
y = m(x)
(y * torch.from_numpy(g["g"]).cuda()).sum().backward()
worst = max((rel_l2(p.grad.cpu(), g["grad." + k]), k) for k, p in m.named_parameters())
print("vssm_small grads         logits l2 %.1e  dx l2 %.1e  worst parameter gradient l2 %.1e (%s)" % (
    rel_l2(y.detach().cpu(), g["logits"]), rel_l2(x.grad.cpu(), g["dx"]), worst[0], worst[1]))
from mamba_unet_amd import MambaUnet  # noqa: E402
g = np.load(os.path.join(GOLDEN, "config0_vssm_tiny_224.npz"))
torch.manual_seed(1337)
net = MambaUnet(img_size=224, num_classes=4).cuda().eval()
with torch.no_grad():
    y = net(torch.from_numpy(g["x"]).cuda())
print("config0 (tiny, 224) logits max %.1e l2 %.1e" % (rel_max(y.cpu(), g["logits"]), rel_l2(y.cpu(), g["logits"])))
