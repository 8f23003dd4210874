import os, sys, time, cProfile, pstats
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
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
pr = cProfile.Profile()
pr.enable()
for it in range(10):
    supervised_step(net, opt, x, y, 4, it, 1000, autocast_bf16=True)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
