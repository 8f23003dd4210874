"""`--gpus N` launcher: one FRESH child process per GPU through torch.distributed.run, started before the
parent has made any GPU call (a process that has initialised the GPU must never be replaced or forked into
workers), rank 0's output relayed, the children's exit status propagated.

The reference has no working multi-process path: `utils/util.py:236-240` defines an `init_process_group("nccl")`
helper nobody calls and `train_Semi_Mamba_UNet_multi.py:287-289` wraps the nets in single-process
`nn.DataParallel`.  Here the contract is the usual one: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
environment mean "I am a rank" (the driver's own `python -m torch.distributed.run ... bench.py --gpus N`
lands there directly); their absence together with `--gpus N > 1` means "I am the launcher"."""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional


def is_rank_process() -> bool:
    """True when a launcher (ours, the driver's or torchrun) already put this process into a world."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def visible_gpus() -> int:
    """Device count without initialising the GPU runtime in this process (torch counts through the driver's
    sysfs / amdsmi path on this image; no context, no HIP call)."""
    import torch
    return int(torch.cuda.device_count())


def launch_ranks(script: str, argv: List[str], n: int, dry_backend: Optional[str] = None) -> int:
    """Run `script argv` as `n` ranks on this host; returns the exit status to leave with.

    `dry_backend="gloo"`: the ranks run their CPU dry mode (no GPU needed), which is how the CPU test suite covers
    this function.  Otherwise `n` GPUs must be visible: measuring fewer ranks than asked for is refused loudly."""
    if n < 1:
        print(f"{os.path.basename(script)}: --gpus must be >= 1 (got {n})", file=sys.stderr)
        return 2
    if dry_backend is None:
        have = visible_gpus()
        if have < n:
            print(f"{os.path.basename(script)}: --gpus {n} asked for, but only {have} GPU(s) are visible: refusing to "
                  f"measure fewer ranks than requested", file=sys.stderr)
            return 3
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")    # dmabuf IPC: RCCL needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 1)) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, *argv]
    proc = subprocess.run(cmd, env=env)
    if proc.returncode != 0:
        print(f"{os.path.basename(script)}: the {n}-rank run failed (exit {proc.returncode})", file=sys.stderr)
    return proc.returncode


def maybe_launch(script: str, gpus: int, argv: List[str], dry_backend: Optional[str] = None) -> None:
    """Call first thing in main(), before any GPU call: when `--gpus N > 1` was given to a bare process, become
    the launcher and leave with the ranks' status; in a rank process (or N = 1) return and let main() go on."""
    if gpus > 1 and not is_rank_process():
        sys.stdout.flush()
        sys.exit(launch_ranks(script, argv, gpus, dry_backend))


def dry_rank(metric: str) -> dict:
    """CPU dry mode of a rank (`--dry gloo`): joins the world over gloo, proves the ranks took part with a sum of
    ones (what the real run reports as `rccl_world_seen`) and returns the JSON record rank 0 prints."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    seen = 1
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo")
        ones = torch.ones(1)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        seen = int(ones.item())
        dist.barrier()
        dist.destroy_process_group()
    return {"metric": metric, "dry": True, "backend": "gloo", "n_gpus": world, "rccl_world_seen": seen, "rank": rank}
