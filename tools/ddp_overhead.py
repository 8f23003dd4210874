"""What DistributedDataParallel costs the eager training step on ONE GPU (an RCCL world of one rank): the hooks, bucket
views and per-bucket all-reduce launches are all there, only the exchange itself is trivial.  The difference to the bare
step is host work plus a few copy-sized kernels: what every rank of an N-GPU run pays before any xGMI traffic.

    python3 tools/ddp_overhead.py [--steps 40] [--warmup 10] [--dtype bf16|fp32]
"""
import argparse
import json
import os
import socket
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(step, steps, warmup):
    for it in range(warmup):
        step(it)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for it in range(steps):
        step(warmup + it)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, t_issue / steps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", choices=["fp32", "bf16"], default="bf16")
    ap.add_argument("--batch", type=int, default=24)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)

    from mamba_unet_amd import SS2D, MambaUnet
    from mamba_unet_amd.dist import wrap_ddp
    from mamba_unet_amd.train import WeightCopies, make_sgd, supervised_step
    bf16 = args.dtype == "bf16"
    SS2D.bf16_scan_io = bf16
    torch.manual_seed(1337)
    images = torch.rand(args.batch, 1, 224, 224, device=dev)
    labels = torch.randint(0, 4, (args.batch, 224, 224), device=dev, dtype=torch.uint8)
    out = {}
    for name in ("bare", "ddp", "bare_again"):
        torch.manual_seed(1337)
        net = MambaUnet(img_size=224, num_classes=4).to(dev).train()
        if name == "ddp":
            net = wrap_ddp(net, 0, force=True)
        opt = make_sgd(net.parameters())
        copies = WeightCopies(net, opt) if bf16 else None

        def step(it):
            return supervised_step(net, opt, images, labels, 4, it, 10000, autocast_bf16=bf16)

        wall, issue = timed(step, args.steps, args.warmup)
        out[name] = {"ms_per_step": round(wall, 3), "host_issue_ms": round(issue, 3)}
        if copies is not None:
            copies.close()
        del net, opt, copies
        torch.cuda.empty_cache()
    out["ddp_minus_bare_ms"] = round(out["ddp"]["ms_per_step"] - min(out["bare"]["ms_per_step"], out["bare_again"]["ms_per_step"]), 3)
    print(json.dumps(out))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
