"""world_size-2 gloo tests (CPU) of the N>1 host logic: batch sharding, max-over-ranks timing, and the
batch-global Dice loss whose sharded form must reproduce the single-process loss AND gradients under DDP
averaging."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from mamba_unet_amd.dist import init_from_env, max_over_ranks, shard_range, wrap_ddp
    from mamba_unet_amd.train import dice_loss
    torch.set_num_threads(1)
    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    # sharding covers the batch exactly once
    sizes = [len(shard_range(7, i, w)) for i in range(w)]
    assert sum(sizes) == 7 and max(sizes) - min(sizes) <= 1
    assert max_over_ranks(float(rank + 1)) == float(world)

    torch.manual_seed(0)
    net = torch.nn.Conv2d(1, 4, 3, padding=1)
    x = torch.rand(6, 1, 8, 8)
    y = torch.randint(0, 4, (6, 1, 8, 8))
    # single-process reference on the whole batch
    ref = torch.nn.Conv2d(1, 4, 3, padding=1)
    ref.load_state_dict(net.state_dict())
    full = dice_loss(torch.softmax(ref(x), 1), y, 4, global_batch=False)
    full.backward()
    # sharded: each rank sees its slices, statistics are all-reduced, DDP averages the gradients
    ddp = wrap_ddp(net)
    idx = list(shard_range(6, r, w))
    loss = dice_loss(torch.softmax(ddp(x[idx]), 1), y[idx], 4, global_batch=True)
    loss.backward()
    ok_loss = torch.allclose(loss.detach(), full.detach(), atol=1e-6)
    ok_grad = all(torch.allclose(p.grad, q.grad, atol=1e-6) for p, q in zip(net.parameters(), ref.parameters()))
    ret[rank] = (bool(ok_loss), bool(ok_grad))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_global_dice():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert dict(ret) == {0: (True, True), 1: (True, True)}


def test_dice_and_constra_match_reference_formulas():
    from mamba_unet_amd.train import constra_loss, dice_loss
    torch.manual_seed(1)
    p = torch.softmax(torch.randn(3, 4, 5, 6), 1)
    t = torch.randint(0, 4, (3, 1, 5, 6))
    # literal per-class loop of utils/losses.py:192-215
    want = 0.0
    for i in range(4):
        ti = (t[:, 0] == i).float()
        inter, ysum, zsum = (p[:, i] * ti).sum(), (ti * ti).sum(), (p[:, i] * p[:, i]).sum()
        want = want + (1 - (2 * inter + 1e-5) / (zsum + ysum + 1e-5))
    assert torch.allclose(dice_loss(p, t, 4, global_batch=False), want / 4, atol=1e-6)
    a, b = torch.randn(2, 4, 5, 5), torch.randn(2, 4, 5, 5)
    pa = torch.nn.functional.normalize(torch.nn.AdaptiveAvgPool2d(1)(a).view(2, -1), p=2, dim=1)
    pb = torch.nn.functional.normalize(torch.nn.AdaptiveAvgPool2d(1)(b).view(2, -1), p=2, dim=1)
    assert torch.allclose(constra_loss(a, b), ((pa - pb) ** 2).mean(), atol=1e-7)


# ----------------------------------------------------------------------------- the `--gpus N` launcher (CPU dry mode)
@pytest.mark.parametrize("script", ["bench.py", "bench_model.py"])
def test_gpus_flag_launches_fresh_ranks(script):
    """`python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment must start two rank processes
    (mamba_unet_amd/launch.py) and relay rank 0's line; the dry mode runs the same plumbing over gloo without a GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, script), "--gpus", "2", "--dry", "gloo"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout                       # ONE JSON line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["rccl_world_seen"] == 2 and rec["dry"] is True


def test_gpus_flag_refuses_more_ranks_than_gpus():
    """No GPU in the CPU test container: a real `--gpus 2` must leave non-zero with a clear message, not measure one rank."""
    import subprocess
    import sys
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs visible")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "refusing" in p.stderr and not p.stdout.strip()


def test_launcher_propagates_child_failure(tmp_path):
    from mamba_unet_amd import launch
    bad = tmp_path / "bad_rank.py"
    bad.write_text("import os, sys\nsys.exit(7 if os.environ['RANK'] == '1' else 0)\n")
    assert launch.launch_ranks(str(bad), [], 2, dry_backend="gloo") != 0
    good = tmp_path / "good_rank.py"
    good.write_text("import os\nassert os.environ['WORLD_SIZE'] == '2'\n")
    assert launch.launch_ranks(str(good), [], 2, dry_backend="gloo") == 0


# ----------------------------------------------------------------------------- semi loss: one stacked all-reduce
def _torch_seg_loss_stats(logits, labels, n_classes):
    """Plain-torch stand-in for the HIP statistics kernel (same 3C+1 numbers), so the host logic of semi_loss --
    the stacking, the single all-reduce, the per-rank cross entropy -- runs on CPU ranks."""
    p = torch.softmax(logits.float(), dim=1)
    if labels.dim() == 4:
        labels = labels[:, 0]
    oh = torch.nn.functional.one_hot(labels.long(), n_classes).permute(0, 3, 1, 2).float()
    ce = -(torch.log_softmax(logits.float(), dim=1) * oh).sum()
    return torch.cat([(p * oh).sum((0, 2, 3)), oh.sum((0, 2, 3)), (p * p).sum((0, 2, 3)), ce.view(1)])


def _semi_worker(rank, world, port, ret):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    from mamba_unet_amd import train
    from mamba_unet_amd.dist import init_from_env, wrap_ddp
    torch.set_num_threads(1)
    init_from_env("gloo")
    train.seg_loss_stats = _torch_seg_loss_stats
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda *a, **k: (calls.append(1), real(*a, **k))[1]

    torch.manual_seed(0)
    nets = [torch.nn.Conv2d(1, 4, 3, padding=1) for _ in range(2)]
    refs = [torch.nn.Conv2d(1, 4, 3, padding=1) for _ in range(2)]
    for n, r in zip(nets, refs):
        r.load_state_dict(n.state_dict())
    x, y = torch.rand(8, 1, 6, 6), torch.randint(0, 4, (8, 6, 6))
    # single process: 4 labelled + 4 unlabelled slices (the collective switched off: this process IS in a world of two)
    keep, train.all_reduce_sum = train.all_reduce_sum, (lambda t: t)
    full = train.semi_loss(refs[0](x), refs[1](x), y, 4, 4, 0.3)
    train.all_reduce_sum = keep
    n_single = len(calls)
    full.backward()
    # two ranks: 2 labelled + 2 unlabelled slices each
    idx = [0, 1, 4, 5] if rank == 0 else [2, 3, 6, 7]
    d1, d2 = wrap_ddp(nets[0]), wrap_ddp(nets[1])
    calls.clear()
    loss = train.semi_loss(d1(x[idx]), d2(x[idx]), y[idx], 2, 4, 0.3)
    n_fwd = len(calls)
    loss.backward()
    # pseudo labels come from the argmax of identical networks on the same slices, the Dice sums are global, the cross
    # entropies are per-rank means whose average over ranks is the global mean -> DDP's averaged gradient is the single one
    ok_grad = all(torch.allclose(p.grad, q.grad, atol=2e-6) for n, r in zip(nets, refs) for p, q in zip(n.parameters(), r.parameters()))
    lsum = loss.detach().clone()
    real(lsum)
    ok_loss = torch.allclose(lsum / world, full.detach(), atol=1e-6)
    ret[rank] = (n_single, n_fwd, bool(ok_grad), bool(ok_loss))
    dist.destroy_process_group()


def test_semi_loss_single_stacked_all_reduce_matches_single_process():
    world = 2
    ret = mp.Manager().dict()
    mp.spawn(_semi_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    # world 1: no collective at all; world 2: exactly ONE all-reduce in the forward of the loss
    assert dict(ret) == {0: (0, 1, True, True), 1: (0, 1, True, True)}
