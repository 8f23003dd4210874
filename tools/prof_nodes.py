"""Host time inside this package's autograd nodes, forward and backward, per node class, for the eager bf16 training
step.  The backward runs on the autograd engine's device thread, which cProfile on the main thread does not see; here
every `forward` / `backward` staticmethod of the package's Function classes is wrapped in a perf_counter pair.

    python3 tools/prof_nodes.py
"""
import collections
import inspect
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import mamba_unet_amd                                                     # noqa: E402
from mamba_unet_amd import SS2D, MambaUnet, selective_scan_interface, ss2d_ops, train   # noqa: E402
from mamba_unet_amd.train import WeightCopies, make_sgd, supervised_step  # noqa: E402

ACC = collections.defaultdict(lambda: [0.0, 0])


def wrap(cls, name):
    fn = getattr(cls, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = ACC[f"{cls.__name__}.{name}"]
            e[0] += time.perf_counter() - t0
            e[1] += 1
    setattr(cls, name, staticmethod(timed))


for mod in (ss2d_ops, selective_scan_interface, train, mamba_unet_amd.dist):
    for _, cls in inspect.getmembers(mod, inspect.isclass):
        if issubclass(cls, torch.autograd.Function) and cls is not torch.autograd.Function and cls.__module__ == mod.__name__:
            wrap(cls, "forward")
            wrap(cls, "backward")

SS2D.bf16_scan_io = True
torch.manual_seed(0)
net = MambaUnet(img_size=224, num_classes=4).cuda().train()
opt = make_sgd(net.parameters())
copies = WeightCopies(net, opt)
x = torch.randn(24, 1, 224, 224, device="cuda")
y = torch.randint(0, 4, (24, 224, 224), device="cuda")
for it in range(8):
    supervised_step(net, opt, x, y, 4, it, 1000, autocast_bf16=True)
torch.cuda.synchronize()
ACC.clear()
STEPS = 20
t0 = time.perf_counter()
for it in range(STEPS):
    supervised_step(net, opt, x, y, 4, it, 1000, autocast_bf16=True)
issue = time.perf_counter() - t0
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"per step: wall {wall / STEPS * 1e3:.2f} ms, host issue {issue / STEPS * 1e3:.2f} ms (with the timers' own cost)")
tot_f = sum(v[0] for k, v in ACC.items() if k.endswith("forward")) / STEPS * 1e3
tot_b = sum(v[0] for k, v in ACC.items() if k.endswith("backward")) / STEPS * 1e3
print(f"inside node forwards {tot_f:.2f} ms, inside node backwards {tot_b:.2f} ms per step")
for k, v in sorted(ACC.items(), key=lambda kv: -kv[1][0]):
    print(f"  {k:44s} {v[0] / STEPS * 1e3:7.3f} ms/step  {v[1] / STEPS:6.1f} calls/step  {v[0] / v[1] * 1e6:7.1f} us/call")
