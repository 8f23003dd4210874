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
