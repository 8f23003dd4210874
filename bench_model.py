#!/usr/bin/env python3
"""bench_model.py -- MambaUNet slices/s (BASELINE.json configs[2], [3], [4]; SURVEY.md 8d "Configs 3/5 inputs").

The second half of BASELINE.json's metric: a caller-level measurement of the hot path inside the whole
network.  bench.py stays the headline (scan GB/s on configs[1]); this file reports

    --mode train   full training step of train_fully_supervised_2D_VIM.py:150-164
                   (fwd, softmax, 0.5*(CE+Dice), zero_grad, backward, SGD step, poly LR)
    --mode infer   eval-mode forward under no_grad
    --mode semi    dual-network cross pseudo supervision + ConstraLoss (train_Semi_Mamba_UNet.py:411-456)

on synthetic slices (`rand(B,1,S,S)`, `randint` labels), seed 1337, random init (no checkpoint offline).
One process per GPU; under torch.distributed.run the model is wrapped in DDP (RCCL all-reduce of the
76.5 MB gradient overlapped with backward), batch 24 per GPU (weak scaling).

    python bench_model.py [--mode train] [--size 224] [--classes 4] [--batch 24] [--dtype fp32|bf16]
                          [--gpus N] [--steps K] [--warmup W]
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["train", "infer", "semi"], default="train")
    ap.add_argument("--model1", choices=["unet", "vim"], default="unet", help="semi mode: the first network (reference default: unet)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--classes", type=int, default=4)
    ap.add_argument("--batch", type=int, default=24, help="slices per GPU")
    ap.add_argument("--dtype", choices=["fp32", "bf16", "bf16-fp32scan"], default="fp32",
                    help="bf16: autocast + bf16 kernel I/O; bf16-fp32scan: autocast with the reference's fp32 casts around the scan")
    ap.add_argument("--literal-directions", action="store_true",
                    help="A/B: run cross_scan / scan / cross_merge literally (4 materialised directions) instead of the fused-direction path")
    ap.add_argument("--graph", action="store_true",
                    help="1 GPU.  infer: replay the forward from a HIP graph (checked against eager at capture).  train: "
                         "the whole step (forward, loss, backward, SGD) captured once and replayed")
    ap.add_argument("--no-weight-copies", action="store_true",
                    help="A/B switch: cast every Linear weight to bf16 in every step instead of refreshing working copies once per optimizer step")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dry", choices=["gloo"], default=None, help="CPU dry run of the launcher / rank plumbing (no GPU, no kernels)")
    return ap


def main():
    args = parser().parse_args()
    # `--gpus N > 1` given to a bare process: become the launcher of N fresh ranks before any GPU call (launch.py)
    from mamba_unet_amd import launch
    launch.maybe_launch(os.path.abspath(__file__), args.gpus, sys.argv[1:], dry_backend=args.dry)
    if args.dry:
        rec = launch.dry_rank(f"MambaUNet {args.mode} slices/s @{args.size}x{args.size}")
        if rec["rank"] == 0:
            print(json.dumps(rec))
        return
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(minutes=5))
    line = run(args, rank, world, local, dist)
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def run(args, rank, world, local, dist):
    """One measurement (bench.py calls this too, for the model-level half of BASELINE.json's metric).  Returns the JSON
    record on rank 0, None elsewhere.  `dist` is the initialised torch.distributed module or None."""
    dev = torch.device("cuda", local)
    torch.cuda.reset_peak_memory_stats(dev)
    from mamba_unet_amd import MambaUnet
    from mamba_unet_amd.dist import wrap_ddp
    from mamba_unet_amd.train import GraphedSupervisedStep, WeightCopies, make_sgd, semi_step, supervised_step

    bf16 = args.dtype != "fp32"
    from mamba_unet_amd import SS2D
    SS2D.bf16_scan_io = args.dtype == "bf16"
    SS2D.fused_directions = not args.literal_directions
    torch.manual_seed(1337)                       # train_fully_supervised_2D_VIM.py:243-246
    if args.mode == "semi" and args.model1 == "unet":     # train_Semi_Mamba_UNet.py:302-303 (default model1) + :325 (model2)
        from mamba_unet_amd.unet import UNet
        nets = [UNet(in_chns=1, class_num=args.classes).to(dev).native_layout(), MambaUnet(img_size=args.size, num_classes=args.classes).to(dev)]
    else:
        nets = [MambaUnet(img_size=args.size, num_classes=args.classes).to(dev) for _ in range(2 if args.mode == "semi" else 1)]
    n_params = sum(p.numel() for p in nets[-1].parameters())
    torch.manual_seed(1337 + rank)                # per-rank data and DropPath streams
    images = torch.rand(args.batch, 1, args.size, args.size, device=dev)
    labels = torch.randint(0, args.classes, (args.batch, args.size, args.size), device=dev, dtype=torch.uint8)
    max_it = 10000

    if args.mode == "infer":
        net = nets[0].eval()

        if args.graph and world == 1:      # forward-only capture: verified against eager inside GraphedForward
            from mamba_unet_amd.train import GraphedForward
            gfwd = GraphedForward(net, images, autocast_bf16=bf16)

            def step(it):
                return gfwd(gfwd.inp)
        else:
            def step(it):
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=bf16):
                    return net(images)
    elif args.mode == "train":
        net = wrap_ddp(nets[0].train(), local)
        if args.graph and world == 1:
            opt = make_sgd(net.parameters(), tensor_lr_device=dev)
            copies = WeightCopies(net, opt) if (bf16 and not args.no_weight_copies) else None     # noqa: F841 (kept alive)
            gstep = GraphedSupervisedStep(net, opt, images, labels, args.classes, max_it, autocast_bf16=bf16)

            def step(it):
                return gstep(gstep.images, gstep.labels, it)
        else:
            opt = make_sgd(net.parameters())
            copies = WeightCopies(net, opt) if (bf16 and not args.no_weight_copies) else None     # noqa: F841 (kept alive)

            def step(it):
                return supervised_step(net, opt, images, labels, args.classes, it, max_it, autocast_bf16=bf16)
    else:
        n1, n2 = (wrap_ddp(n.train(), local) for n in nets)
        o1, o2 = make_sgd(n1.parameters()), make_sgd(n2.parameters())
        copies = [WeightCopies(n, o) for n, o in ((n1, o1), (n2, o2))] if (bf16 and not args.no_weight_copies) else None     # noqa: F841

        def step(it):
            return semi_step(n1, n2, o1, o2, images, labels, args.batch // 2, args.classes, it, max_it, 0.1, autocast_bf16=bf16)

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for it in range(args.warmup):
        step(it)
    sync_all()
    t0 = time.perf_counter()
    for it in range(args.steps):
        last = step(args.warmup + it)
    sync_all()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    finite = bool(torch.isfinite(last.float()).all())

    from mamba_unet_amd import SS2D as _SS2D
    _SS2D.bf16_scan_io, _SS2D.fused_directions = False, True          # class-wide switches back to their defaults
    cp = locals().get("copies")
    for c in (cp if isinstance(cp, list) else [cp] if cp is not None else []):
        c.close()                                                      # bench.py runs several legs in one process
    if rank == 0:
        if not finite:
            print("bench_model.py: WARNING: the last step's output is not finite -- this run is not a valid measurement", file=sys.stderr)
        return ({
            "metric": f"MambaUNet {args.mode} slices/s @{args.size}x{args.size}",
            "value": round(world * args.batch * args.steps / elapsed, 2), "unit": "slices/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"fp32": "f32", "bf16": "bf16 autocast + bf16 kernel I/O (fp32 scan arithmetic/state, fp32 master weights)",
                      "bf16-fp32scan": "bf16 autocast, fp32 scan I/O, fp32 master weights"}[args.dtype], "data": "synthetic",
            "config": {"workload": f"MambaUNet (VSSM tiny, depths 2-2-2-2, {n_params} params) {args.mode}, batch "
                                   f"{args.batch} per GPU, {args.size}x{args.size}, {args.classes} classes",
                       "parallelism": f"DDP x{world} (batch-sharded slices, RCCL gradient all-reduce)"},
            "hip_graph": bool(args.graph and world == 1 and args.mode in ("train", "infer")),
            "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 1e9, 2), "finite": finite, "valid": finite})
    return None


if __name__ == "__main__":
    main()
