import sys, os, copy, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mamba_unet_amd import MambaUnet, SS2D
from mamba_unet_amd.train import GraphedSupervisedStep, make_sgd, supervised_step
torch.manual_seed(1337)
net1 = MambaUnet(img_size=224, num_classes=4).cuda().train()
for m in net1.modules():
    if type(m).__name__ == "DropPath": m.drop_prob = 0.0
net2 = copy.deepcopy(net1)
torch.manual_seed(1337)
images = torch.rand(24, 1, 224, 224, device="cuda"); labels = torch.randint(0, 4, (24, 224, 224), device="cuda", dtype=torch.uint8)
opt1 = make_sgd(net1.parameters(), tensor_lr_device=torch.device("cuda"))
g = GraphedSupervisedStep(net1, opt1, images, labels, 4, 10000)
opt2 = make_sgd(net2.parameters(), tensor_lr_device=torch.device("cuda"))
for i in range(3): l2 = supervised_step(net2, opt2, images, labels, 4, i, 10000)
torch.cuda.synchronize()
d0 = max(float((a - b).abs().max()) for a, b in zip(net1.parameters(), net2.parameters()))
print("after warmup: max param diff", d0, "eager loss", float(l2))
nreplay = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for k in range(nreplay):
    l1 = g(g.images, g.labels, 3 + k)
    l2 = supervised_step(net2, opt2, images, labels, 4, 3 + k, 10000)
    torch.cuda.synchronize()
    rows = []
    for (n, a), b in zip(net1.named_parameters(), net2.parameters()):
        den = float(b.abs().max()) + 1e-12
        rows.append((float((a - b).abs().max()) / den, n))
    rows.sort(reverse=True)
    print("replay", k, "graph loss", float(l1), "eager loss", float(l2), "worst params:", [(round(r, 5), n) for r, n in rows[:6]], flush=True)
    # gradient comparison
    rows = []
    for (n, a), b in zip(net1.named_parameters(), net2.parameters()):
        if a.grad is None or b.grad is None: continue
        den = float(b.grad.abs().max()) + 1e-20
        rows.append((float((a.grad - b.grad).abs().max()) / den, n))
    rows.sort(reverse=True)
    print("   worst grads:", [(round(r, 4), n) for r, n in rows[:8]], flush=True)
