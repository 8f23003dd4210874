"""DistributedDataParallel around the REAL networks on one GPU: a world of one rank over RCCL (backend "nccl").

The CPU suite covers the sharding logic with gloo ranks, but this package has no CPU path, so there DDP can only wrap a
stand-in.  Here the production wrapper (`dist.wrap_ddp`: static_graph, gradient_as_bucket_view) meets this package's
autograd nodes -- holder-passing gradients (GradSlots / GradHalves), gradients that are views of the zero arena,
HIP-kernel Functions -- before the driver's 8-GPU run does: three training steps of a wrapped network must reproduce
three steps of an unwrapped copy (reference step: train_fully_supervised_2D_VIM.py:147-164; the reference's own
multi-GPU attempts: utils/util.py:236-240, train_Semi_Mamba_UNet_multi.py:287-289)."""
import copy
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world_of_one():
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.fixture(autouse=True)
def _restore_switches():
    from mamba_unet_amd import SS2D
    old = SS2D.bf16_scan_io
    yield
    SS2D.bf16_scan_io = old


def _deterministic(net):
    for m in net.modules():                 # DropPath / Dropout draw from the generator: the two copies would see other masks
        if hasattr(m, "drop_prob"):
            m.drop_prob = 0.0
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    return net


def _batch(seed, n=4, size=224, classes=4):
    gen = torch.Generator().manual_seed(seed)
    return (torch.rand(n, 1, size, size, generator=gen).cuda(),
            torch.randint(0, classes, (n, size, size), generator=gen).to(torch.uint8).cuda())


@pytest.mark.parametrize("bf16", [False, True])
def test_ddp_wrapped_mambaunet_supervised_steps_equal_unwrapped(world_of_one, bf16):
    from torch.nn.parallel import DistributedDataParallel as DDP
    from mamba_unet_amd import SS2D, MambaUnet
    from mamba_unet_amd.dist import wrap_ddp
    from mamba_unet_amd.train import make_sgd, supervised_step
    SS2D.bf16_scan_io = bf16
    torch.manual_seed(1337)
    bare = _deterministic(MambaUnet(img_size=224, num_classes=4).cuda().train())
    twin = copy.deepcopy(bare)
    assert wrap_ddp(twin, 0) is twin                    # a world of one stays bare unless forced
    ddp = wrap_ddp(twin, 0, force=True)
    assert isinstance(ddp, DDP) and ddp.static_graph and ddp.gradient_as_bucket_view
    images, labels = _batch(21)
    o1, o2 = make_sgd(bare.parameters()), make_sgd(ddp.parameters())
    for it in range(3):
        l1 = float(supervised_step(bare, o1, images, labels, 4, it, 10000, autocast_bf16=bf16))
        l2 = float(supervised_step(ddp, o2, images, labels, 4, it, 10000, autocast_bf16=bf16))
        # fp32: same kernels, DDP averages over one rank; the weight-gradient float atomics move the last bits only.
        # bf16: two EAGER runs already differ by ~2e-5 (tests/test_configs_gpu.py::test_graphed_supervised_step_matches_eager)
        assert abs(l1 - l2) <= (1e-3 if bf16 else 1e-6) * abs(l1), (it, l1, l2)
    # every parameter took part in every step: gradients present, finite, living in DDP's buckets (views), and equal
    # bf16: the float atomics of the bf16 weight-gradient kernels make two EAGER runs of the bare network differ too, and
    # three steps amplify it -- measured over four trials of bare against bare: worst single parameter (an x_proj weight)
    # 2.4e-2 ... 6.1e-2, all gradients together 2.5e-3, the same figures with and without the wrapper.  The gates are ~5x / 4x
    # those; the fp32 gate is the sharp one.
    worst, num, den2 = 0.0, 0.0, 0.0
    for (k, p), q in zip(bare.named_parameters(), twin.parameters()):
        assert q.grad is not None and torch.isfinite(q.grad).all(), k
        d, den = float((p.grad - q.grad).norm()), float(p.grad.norm())
        worst = max(worst, d / (den if den > 0 else 1.0))
        num, den2 = num + d * d, den2 + den * den
    assert worst < (3e-1 if bf16 else 1e-4), worst
    assert (num / den2) ** 0.5 < (1e-2 if bf16 else 1e-4), (num / den2) ** 0.5
    wmax = max(float((p.detach() - q.detach()).abs().max()) for p, q in zip(bare.parameters(), twin.parameters()))
    assert wmax < (1e-3 if bf16 else 1e-6), wmax


def test_ddp_wrapped_semi_step_two_networks_equal_unwrapped(world_of_one):
    """configs[3] under DDP: UNet (BatchNorm buffers broadcast by DDP) + MambaUnet, ONE backward through both wrappers."""
    from mamba_unet_amd import MambaUnet
    from mamba_unet_amd.dist import wrap_ddp
    from mamba_unet_amd.train import make_sgd, semi_step
    from mamba_unet_amd.unet import UNet
    torch.manual_seed(1337)
    a1 = _deterministic(UNet(in_chns=1, class_num=4).cuda().native_layout().train())
    a2 = _deterministic(MambaUnet(img_size=224, num_classes=4).cuda().train())
    b1, b2 = copy.deepcopy(a1), copy.deepcopy(a2)
    d1, d2 = wrap_ddp(b1, 0, force=True), wrap_ddp(b2, 0, force=True)
    images, labels = _batch(22, n=4)
    oa = [make_sgd(a1.parameters()), make_sgd(a2.parameters())]
    ob = [make_sgd(d1.parameters()), make_sgd(d2.parameters())]
    for it in range(3):
        la = float(semi_step(a1, a2, *oa, images, labels, 2, 4, it, 30000, 0.1))
        lb = float(semi_step(d1, d2, *ob, images, labels, 2, 4, it, 30000, 0.1))
        # pseudo labels are an argmax: a last-bit difference can flip a pixel, so the gate is looser than the supervised one
        assert abs(la - lb) <= 1e-4 * abs(la), (it, la, lb)
    for net in (b1, b2):
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in net.parameters())
