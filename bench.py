#!/usr/bin/env python3
"""bench.py -- SS2D selective-scan fwd+bwd microbench (BASELINE.json configs[1]) on N MI355X.

A "step" = one forward + one backward pass of the selective scan over one batch of synthetic input
(B=24, K=4, D=192 -> K*D=768, L=3136, d_state=16, fp32), called through the C ABI of
libss2d_hip.so with every input already resident in HBM.  The step includes the zero-fill of the
atomically accumulated gradients (dA, dB, dC, dD, ddelta_bias).

Multi-GPU: the path shards over the batch of slices with no data-path collective, so
every rank runs its own batch (weak scaling); value = bytes of all ranks / max-over-ranks time.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A bare `python bench.py --gpus N` (no RANK / WORLD_SIZE in the environment) is the launcher: before any GPU call it
starts N fresh rank processes through torch.distributed.run (mamba_unet_amd/launch.py), relays rank 0's line and
leaves with their exit status; it refuses to run when fewer than N GPUs are visible.  `--dry gloo` runs the same
launcher / rank plumbing on CPU (no GPU, no kernels): the CPU test suite uses it.

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

BATCH, K, D, L, N = 24, 4, 192, 3136, 16
KD = K * D
CPU_SLICES = 24         # bounded CPU sample: the whole config-2 batch once, ~10 s of host time
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# HBM bytes per launch and the VALU-busy share come from the PMC counters, which cannot be collected inside this
# process: tools/profile_round.sh runs separate rocprofv3 --pmc passes of this very command and
# tools/summarize_pmc.py writes profiles/rNN/bench_pmc.json (2 x FETCH_SIZE x 1024 -- gfx950 tallies a 128-byte read
# request as 64 bytes -- + WRITE_SIZE x 1024, averaged over the dispatches) together with a hash of the scan kernel
# sources it was taken with.  The newest such file is read here; `sources_match` says whether the kernels have
# changed since (then the figures are stale and the round's profile has to be retaken).
SCAN_SOURCES = ("ss2d_scan_fwd.hip", "ss2d_scan_bwd.hip", "ss2d_scan_bwd_sr.hip", "ss2d_scan.h", "ss2d_common.h")


def scan_sources_hash():
    import hashlib
    h = hashlib.sha256()
    for f in SCAN_SOURCES:
        with open(os.path.join(ROOT, "mamba_unet_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def load_pmc():
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "bench_pmc.json")),
                   key=lambda f: int(re.search(r"r(\d+)", os.path.basename(os.path.dirname(f))).group(1)))
    if not files:
        return None
    with open(files[-1]) as fh:
        d = json.load(fh)
    d["file"] = os.path.relpath(files[-1], ROOT)
    d["sources_match"] = d.get("sources_hash") == scan_sources_hash()
    return d


def algorithmic_bytes(batch):
    """per call, fp32."""
    s = 4
    fwd = s * (3 * batch * KD * L + 2 * batch * K * N * L) + 4 * (KD * N + 2 * KD)
    bwd = s * (5 * batch * KD * L + 4 * batch * K * N * L) + 4 * 2 * (KD * N + 2 * KD)
    return fwd, bwd


PREWARM_STEPS = 25      # untimed steps ahead of the warm-up (clock ramp), see main()

# MambaUNet legs appended to the line: bench_model.py arguments + algorithmic scan MB per slice fwd+bwd (SURVEY.md 8d;
# forward only = 732/1965 of it; the 512x512 network has 5.22x the tokens of the 224x224 one)
_T = ["--dtype", "bf16", "--batch", "24"]
MODEL_LEGS = {
    "train224": (["--mode", "train", *_T, "--steps", "20", "--warmup", "5"], 584.2),                      # configs[2]
    "infer224": (["--mode", "infer", *_T, "--steps", "20", "--warmup", "5"], 584.2 * 732.2 / 1965.5),
    "train512": (["--mode", "train", *_T, "--size", "512", "--classes", "9", "--steps", "10", "--warmup", "3"],
                 584.2 * (512 * 512) / (224 * 224)),                                                     # configs[4]
    "semi224": (["--mode", "semi", *_T, "--steps", "10", "--warmup", "3"], None),                          # configs[3]
}


def cpu_baseline():
    """The reference's CPU path (sequential selective_scan_ref in PyTorch, restated in oracle/) timed
    on this host: forward + autograd backward on CPU_SLICES slices of the same workload."""
    from oracle.selective_scan_ref import selective_scan_ref      # the checker: this leg and the parity block only
    from mamba_unet_amd.synthetic import make_scan_inputs
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    nb = CPU_SLICES
    inp = make_scan_inputs(nb, KD, L, dstate=N, ngroups=K, seed=0)
    leaves = {k: v.clone().requires_grad_(True) for k, v in inp.items()}
    dout = torch.randn(nb, KD, L, generator=torch.Generator().manual_seed(1))
    t0 = time.perf_counter()
    y = selective_scan_ref(leaves["u"], leaves["delta"], leaves["A"], leaves["B"], leaves["C"], leaves["D"], None,
                           leaves["delta_bias"], True)
    y.backward(dout)
    dt = time.perf_counter() - t0
    fwd, bwd = algorithmic_bytes(nb)
    return {"value": round((fwd + bwd) / dt / 1e9, 4), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": f"{nb} of {BATCH} slices (B={nb}, K*D={KD}, L={L}, N={N}) fwd + autograd bwd of "
                      f"oracle.selective_scan_ref (the time-step `unbind` form of the reference's loop, which spares "
                      f"autograd the O(L^2) select_backward of the literal x[:, :, i] indexing), {dt:.2f} s",
            "seconds": round(dt, 3)}, (inp, y.detach(), leaves)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--atomic-dbc", action="store_true", help="dB/dC through float atomics (no workspace): the round-1 form")
    ap.add_argument("--no-model", action="store_true", help="skip the MambaUNet legs (second half of BASELINE.json's metric)")
    ap.add_argument("--model-legs", default="train224,infer224,train512,semi224",
                    help="comma list of the MambaUNet legs appended outside the timed region (see MODEL_LEGS)")
    ap.add_argument("--dry", choices=["gloo"], default=None, help="CPU dry run of the launcher / rank plumbing (no GPU, no kernels)")
    args = ap.parse_args()

    # ---- launcher: `--gpus N > 1` given to a bare process -> N fresh ranks, started before this process touches the GPU
    from mamba_unet_amd import launch
    launch.maybe_launch(os.path.abspath(__file__), args.gpus, sys.argv[1:], dry_backend=args.dry)
    if args.dry:
        rec = launch.dry_rank("SS2D selective-scan fwd+bwd achieved HBM GB/s (algorithmic bytes / time)")
        if rec["rank"] == 0:
            print(json.dumps(rec))
        return

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and rank == 0:
        print(f"bench.py: note: --gpus {args.gpus} but the launcher's WORLD_SIZE is {world}; reporting n_gpus={world}", file=sys.stderr)
    torch.cuda.set_device(local)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        # a rank that dies leaves the others in a collective: a bounded timeout turns that into an error, not a hang
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(minutes=5))

    from mamba_unet_amd import _lib
    from mamba_unet_amd.synthetic import make_scan_inputs
    lib = _lib.load()
    dev = torch.device("cuda", local)

    inp = {k: v.to(dev) for k, v in make_scan_inputs(BATCH, KD, L, dstate=N, ngroups=K, seed=rank).items()}
    dout = torch.randn(BATCH, KD, L, generator=torch.Generator().manual_seed(1 + rank)).to(dev)
    out = torch.empty_like(inp["u"])
    ckpt = torch.empty(lib.ss2d_scan_ckpt_bytes(BATCH, KD, L, N) // 4, device=dev)
    du, ddelta = torch.empty_like(out), torch.empty_like(out)
    # one flat buffer for the atomically accumulated gradients -> a single memset per step.  With the workspace the
    # library writes dB / dC itself (partial tiles + one reduction pass), so only dA, dD, dbias are zero-filled.
    ws_bytes = 0 if args.atomic_dbc else int(lib.ss2d_scan_bwd_workspace_bytes(BATCH, KD, L, N, K))
    ws = torch.empty(ws_bytes // 4, device=dev) if ws_bytes else None
    sizes = [KD * N, KD, KD] + ([] if ws_bytes else [BATCH * K * N * L, BATCH * K * N * L])
    flat = torch.empty(sum(sizes), device=dev)
    if ws_bytes:
        dA, dD, dbias = torch.split(flat, sizes)
        dB, dC = torch.empty(BATCH * K * N * L, device=dev), torch.empty(BATCH * K * N * L, device=dev)
    else:
        dA, dD, dbias, dB, dC = torch.split(flat, sizes)
    P = _lib.ptr
    stream = torch.cuda.current_stream().cuda_stream

    def fwd():
        _lib.check(lib.ss2d_selective_scan_fwd(0, P(inp["u"]), P(inp["delta"]), P(inp["A"]), P(inp["B"]), P(inp["C"]),
                                               P(inp["D"]), None, P(inp["delta_bias"]), P(out), None, P(ckpt),
                                               BATCH, KD, L, N, K, 1, stream), "fwd")

    def bwd():
        _lib.check(lib.ss2d_selective_scan_dir_bwd_ws(0, P(inp["u"]), P(inp["delta"]), P(inp["A"]), P(inp["B"]), P(inp["C"]),
                                                      P(inp["D"]), None, P(inp["delta_bias"]), P(dout), P(ckpt), P(du),
                                                      P(ddelta), P(dA), P(dB), P(dC), P(dD), None, P(dbias),
                                                      BATCH, KD, L, N, K, 1, K, 0, 0, P(ws), ws_bytes, stream), "bwd")

    def step(ev=None):
        if ev:
            ev[0].record()
        fwd()
        if ev:
            ev[1].record()
        flat.zero_()
        if ev:
            ev[2].record()
        bwd()
        if ev:
            ev[3].record()

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # The chip needs ~15 ms of sustained load before it runs at its steady clock (20 timed steps after 5 warm-up steps read
    # 7 % lower than after 30): a fixed pre-warm of untimed steps ahead of the W warm-up steps, reported as `prewarm_steps`.
    for _ in range(PREWARM_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
    sync_all()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(evs[i])
    sync_all()
    elapsed = time.perf_counter() - t0
    rccl_world_seen = 1
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        ones = torch.ones(1, device=dev)          # proof on the record that `world` ranks took part (RCCL sum of ones)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        rccl_world_seen = int(ones.item())

    fwd_us = sum(e[0].elapsed_time(e[1]) for e in evs) / args.steps * 1e3
    bwd_us = sum(e[2].elapsed_time(e[3]) for e in evs) / args.steps * 1e3
    fb, bb = algorithmic_bytes(BATCH)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * (fb + bb) / (elapsed / args.steps) / 1e9

    # ---- second half of BASELINE.json's metric: MambaUNet slices/s (configs[2], [3], [4] + the eval forward), batch 24
    # per GPU, bf16, DDP over RCCL when N > 1; outside the timed region of the headline value, same barrier +
    # max-over-ranks timing.  A leg that fails is reported, not hidden, and must fail on EVERY rank together: the ranks
    # agree on a success flag before the next leg enters its collectives.
    models = {}
    if not args.no_model:
        import bench_model
        for leg in [x for x in args.model_legs.split(",") if x]:
            margv, scan_mb_per_slice = MODEL_LEGS[leg]
            margs = bench_model.parser().parse_args(margv)
            rec, ok = None, 1
            try:
                rec = bench_model.run(margs, rank, world, local, dist)
            except Exception as e:
                rec, ok = {"error": f"rank {rank}: {type(e).__name__}: {e}"[:300]}, 0
            if dist is not None:
                flag = torch.tensor([ok], device=dev, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 0:
                    if ok:
                        rec = {"error": "another rank failed in this leg"}
                    elif rank != 0:
                        print(f"bench.py: model leg {leg}: {rec['error']}", file=sys.stderr)
            if rank == 0 and rec is not None and "error" not in rec:
                rec = {k: rec[k] for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "steps", "warmup", "dtype", "config", "peak_mem_GB", "valid")}
                if scan_mb_per_slice:
                    # SURVEY.md 8(d) "whole-model algorithmic scan bytes" per slice (fp32 figure; bf16 I/O moves half of it)
                    gbs = 0.5 * scan_mb_per_slice * 1e6 * margs.batch / (rec["ms_per_step"] * 1e-3) / 1e9
                    rec["scan_algorithmic_GBs_per_gpu"] = round(gbs, 1)
                    rec["frac_of_hbm_peak_scan_bytes_only"] = round(gbs / HBM_PEAK_GBS, 4)
            models[leg] = rec
            torch.cuda.empty_cache()
    model = models.get("train224")

    if rank == 0:
        # parity spot check of what was just timed (slice 0) + CPU baseline, N=1 only
        cpu = None
        parity = None
        if world == 1 and not args.no_cpu_baseline:
            cpu, (cinp, cy, leaves) = cpu_baseline()
            from mamba_unet_amd import selective_scan_fn
            gi = {k: v.to(dev).requires_grad_(True) for k, v in cinp.items()}
            gy = selective_scan_fn(gi["u"], gi["delta"], gi["A"], gi["B"], gi["C"], gi["D"], None, gi["delta_bias"], True)
            gdout = torch.randn(CPU_SLICES, KD, L, generator=torch.Generator().manual_seed(1))
            gy.backward(gdout.to(dev))
            rel = lambda a, b: float((a.double().cpu() - b.double()).norm() / b.double().norm())  # noqa: E731
            parity = {"out": rel(gy.detach(), cy), **{"d" + k: rel(gi[k].grad, leaves[k].grad) for k in gi}}
        # the backward is two launches: scan_bwd_sr_kernel + the small dB/dC partial-tile reduction that follows it
        dominant_us, dominant_bytes, dominant = (bwd_us, bb, "scan_bwd_sr_kernel<float> (+ scan_bwd_sr_reduce_kernel)") \
            if bwd_us >= fwd_us else (fwd_us, fb, "scan_fwd_kernel<float>")
        achieved = dominant_bytes / (dominant_us * 1e-6) / 1e9
        pmc = load_pmc()
        pk = "scan_bwd" if bwd_us >= fwd_us else "scan_fwd"
        traffic = pmc["traffic_bytes"].get(pk) if pmc else None
        valu = None
        if pmc and pmc.get("valu"):
            # the scan is bound by the vector ALU, not by HBM: busy share of the SIMDs at the clock the kernel ran at
            valu = {"bound": "valu", "kernel": pk, **pmc["valu"].get(pk, {}), "source": pmc["file"], "sources_match": pmc["sources_match"]}
        line = {
            "metric": "SS2D selective-scan fwd+bwd achieved HBM GB/s (algorithmic bytes / time)",
            "value": round(value, 2), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": PREWARM_STEPS,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: SS2D selective-scan microbench B=24 K=4 D=192 L=3136 "
                                   "d_state=16 fp32 fwd+bwd per GPU", "batch_per_gpu": BATCH, "KD": KD, "L": L,
                       "d_state": N, "parallelism": f"batch-sharded x{world}, no collective"},
            "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": (pmc["file"] + (" (kernel sources unchanged since)" if pmc["sources_match"] else
                                                            " (STALE: kernel sources changed since)")) if pmc else None,
                         "algorithmic_bytes_per_launch": dominant_bytes, "avg_launch_us": round(dominant_us, 2),
                         "limiter": "valu (see roofline_valu): the vector ALU is what is busy; measured HBM traffic / algorithmic bytes = "
                                    + (f"{traffic / dominant_bytes:.2f}x" if traffic else "not measured")},
            "roofline_valu": valu, "rccl_world_seen": rccl_world_seen,
            "frac_of_hbm_peak_fwd_bwd": round(value / world / HBM_PEAK_GBS, 4),
            "kernels_us": {"scan_fwd": round(fwd_us, 2), "scan_bwd": round(bwd_us, 2),
                           "fwd_GBs": round(fb / fwd_us / 1e3, 1), "bwd_GBs": round(bb / bwd_us / 1e3, 1)},
            "cpu_baseline": cpu, "parity_rel_l2_vs_cpu_oracle": parity, "mambaunet_train": model,
            "mambaunet_infer_224": models.get("infer224"), "mambaunet_train_512": models.get("train512"),
            "semi_224": models.get("semi224"),
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
