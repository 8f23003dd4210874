#!/usr/bin/env python3
"""Host side of the eager bf16 training step: wall time per step against the time the Python thread needs to ISSUE a step
(no synchronisation inside the loop; the queue is drained before and after).  When the issue time approaches the step
time the step is host-bound and N ranks on one node compete for cores."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_unet_amd import SS2D, MambaUnet
from mamba_unet_amd.train import WeightCopies, make_sgd, supervised_step
SS2D.bf16_scan_io = True
torch.manual_seed(0)
net = MambaUnet(img_size=224, num_classes=4).cuda().train()
opt = make_sgd(net.parameters())
copies = WeightCopies(net, opt)
x = torch.randn(24, 1, 224, 224, device="cuda"); y = torch.randint(0, 4, (24, 224, 224), device="cuda")
for it in range(8):
    supervised_step(net, opt, x, y, 4, it, 1000, autocast_bf16=True)
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter(); c0 = time.process_time()
for it in range(n):
    supervised_step(net, opt, x, y, 4, it, 1000, autocast_bf16=True)
t1 = time.perf_counter(); c1 = time.process_time()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("issue %.2f ms/step (cpu time %.2f ms/step), wall %.2f ms/step" % ((t1 - t0) / n * 1e3, (c1 - c0) / n * 1e3, (t2 - t0) / n * 1e3))
