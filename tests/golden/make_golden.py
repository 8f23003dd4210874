"""Generates tests/golden/*.npz by RUNNING THE REFERENCE's own code in the authoring container.

Run from the repo root:   python tests/golden/make_golden.py
Needs /root/reference (absent on the GPU box -- only the .npz fixtures travel).

What is imported from the reference: code/networks/mamba_sys.py (SS2D, VSSBlock, VSSM), loaded by
path.  Its imports of `timm.models.layers` and `fvcore.nn` (mamba_sys.py:12-13) are absent here
and are replaced by stub modules (DropPath = standard stochastic depth, identity in eval;
trunc_normal_ = torch.nn.init.trunc_normal_; fvcore names = None; they are used only by
VSSM.flops()).  The third-party `mamba_ssm` scan is absent too (mamba_sys.py:17-20 swallows the
ImportError); oracle.selective_scan_ref (the restatement of the text embedded at
mamba_sys.py:62-144) is assigned to the module global `selective_scan_fn` that
SS2D.forward_corev0 looks up at call time (mamba_sys.py:397).

The fixtures are DATA (inputs, parameters, outputs, gradients, index tables); no reference source
text is stored.
"""
import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.selective_scan_ref import selective_scan_ref  # noqa: E402

REF = "/root/reference/code/networks/mamba_sys.py"


def load_reference():
    class DropPath(nn.Module):
        def __init__(self, drop_prob=0.0):
            super().__init__()
            self.drop_prob = drop_prob

        def forward(self, x):
            if self.drop_prob == 0.0 or not self.training:
                return x
            keep = 1 - self.drop_prob
            mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
            return x * mask / keep

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")
    timm_layers.DropPath = DropPath
    timm_layers.trunc_normal_ = torch.nn.init.trunc_normal_
    timm.models = timm_models
    timm_models.layers = timm_layers
    fv = types.ModuleType("fvcore")
    fvnn = types.ModuleType("fvcore.nn")
    for name in ("FlopCountAnalysis", "flop_count_str", "flop_count", "parameter_count"):
        setattr(fvnn, name, None)
    fv.nn = fvnn
    sys.modules.update({"timm": timm, "timm.models": timm_models, "timm.models.layers": timm_layers,
                        "fvcore": fv, "fvcore.nn": fvnn})
    spec = importlib.util.spec_from_file_location("ref_mamba_sys", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def np_(t):
    return t.detach().cpu().numpy()


def ss2d_case(ref, name, B, C, H, W, seed, with_scan_args):
    torch.manual_seed(seed)
    m = ref.SS2D(d_model=C).float()
    # perturb the deterministic inits (A_logs = log(1..N), Ds = 1, LayerNorm = identity) so that a
    # wrong channel/direction index cannot hide behind identical rows
    with torch.no_grad():
        m.A_logs.add_(0.1 * torch.randn_like(m.A_logs))
        m.Ds.add_(0.1 * torch.randn_like(m.Ds))
        m.out_norm.weight.add_(0.1 * torch.randn_like(m.out_norm.weight))
        m.out_norm.bias.add_(0.1 * torch.randn_like(m.out_norm.bias))
    x = torch.randn(B, H, W, C, requires_grad=True)
    g = torch.randn(B, H, W, C)

    captured = {}

    def spy(u, delta, A, Bm, Cm, D=None, z=None, delta_bias=None, delta_softplus=False,
            return_last_state=False):
        out = selective_scan_ref(u, delta, A, Bm, Cm, D, z, delta_bias, delta_softplus, return_last_state)
        captured.update(u=u, delta=delta, A=A, B=Bm, C=Cm, D=D, delta_bias=delta_bias, out_y=out,
                        delta_softplus=delta_softplus, z_is_none=z is None)
        return out

    ref.selective_scan_fn = spy
    # forward_corev0 alone (input is the conv output), then the full forward
    xz = m.in_proj(x)
    xi, _ = xz.chunk(2, dim=-1)
    xi = m.act(m.conv2d(xi.permute(0, 3, 1, 2).contiguous()))
    core_out = m.forward_corev0(xi)
    y = m(x)
    (y * g).sum().backward()

    out = {"x": np_(x), "g": np_(g), "y": np_(y), "dx": np_(x.grad),
           "core_in": np_(xi), "core_out": np_(core_out)}
    for k, v in m.state_dict().items():
        out["sd." + k] = np_(v)
    for k, p in m.named_parameters():
        out["grad." + k] = np_(p.grad)
    assert captured["delta_softplus"] is True and captured["z_is_none"]
    if with_scan_args:
        for k in ("u", "delta", "A", "B", "C", "D", "delta_bias", "out_y"):
            out["scan." + k] = np_(captured[k])
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, {k: v.shape for k, v in out.items() if not k.startswith(("sd.", "grad."))})


def index_tables(ref):
    """Cross-scan / cross-merge tables measured from the reference's own forward_corev0
    (mamba_sys.py:396-461): an arange image goes in, the scan spy records the `xs` it receives
    (-> src table) and returns a tagged out_y whose merged image is read back (-> dst table);
    out_norm is replaced by Identity so the merged values come out untouched."""
    out = {}
    m = ref.SS2D(d_model=4).float()      # d_inner = 8
    m.out_norm = nn.Identity()
    D = m.d_inner
    for (H, W) in [(3, 5), (5, 3), (7, 7), (4, 6), (56, 56), (128, 128)]:
        L = H * W
        x = torch.zeros(1, D, H, W)
        x[0, 0] = torch.arange(L, dtype=torch.float32).view(H, W)
        src = np.empty((4, L), dtype=np.int64)
        dst = np.empty((4, L), dtype=np.int64)
        for k in range(4):
            def spy(u, delta, A, Bm, Cm, Dv=None, z=None, delta_bias=None, delta_softplus=False,
                    return_last_state=False, _k=k):
                src[_k] = u[0, _k * D].round().to(torch.int64).numpy()
                o = torch.zeros_like(u)
                o[0, _k * D] = torch.arange(1, L + 1, dtype=torch.float32)
                return o
            ref.selective_scan_fn = spy
            with torch.no_grad():
                y = m.forward_corev0(x)                       # (1, H, W, D)
            dst[k] = y[0, :, :, 0].reshape(L).round().to(torch.int64).numpy() - 1
        if L <= 64:
            out[f"src_{H}x{W}"] = src
            out[f"dst_{H}x{W}"] = dst
        else:  # big tables: hashes only (regenerable from the closed forms in oracle/ss2d_ref.py)
            out[f"src_{H}x{W}_sha256"] = np.frombuffer(hashlib.sha256(src.tobytes()).digest(), dtype=np.uint8)
            out[f"dst_{H}x{W}_sha256"] = np.frombuffer(hashlib.sha256(dst.tobytes()).digest(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "index_tables.npz"), **out)
    print("index_tables", list(out))


def vssm_case(ref):
    """A small VSSM (same code path as vmamba_tiny, narrower) in eval mode: full state dict,
    input and logits; plus the parameter count and a checksum of the seed-1337 default-width model's
    init (train_fully_supervised_2D_VIM.py:243-246 seeds 1337)."""
    ref.selective_scan_fn = selective_scan_ref
    torch.manual_seed(1337)
    m = ref.VSSM(patch_size=4, in_chans=3, num_classes=4, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128],
                 drop_path_rate=0.2).eval()
    x = torch.rand(1, 3, 32, 32)
    with torch.no_grad():
        y = m(x)
    out = {"x": np_(x), "logits": np_(y)}
    for k, v in m.state_dict().items():
        out["sd." + k] = np_(v)
    np.savez_compressed(os.path.join(HERE, "vssm_small.npz"), **out)
    print("vssm_small", y.shape, sum(p.numel() for p in m.parameters()))

    torch.manual_seed(1337)
    big = ref.VSSM(patch_size=4, in_chans=3, num_classes=4, depths=[2, 2, 2, 2], drop_path_rate=0.2)
    sd = big.state_dict()
    names = sorted(sd)
    summary = {"n_params": np.array(sum(p.numel() for p in big.parameters()), dtype=np.int64),
               "names": np.array(names),
               "sums": np.array([float(sd[k].double().sum()) for k in names]),
               "abs_sums": np.array([float(sd[k].double().abs().sum()) for k in names]),
               "numels": np.array([sd[k].numel() for k in names], dtype=np.int64)}
    np.savez_compressed(os.path.join(HERE, "vssm_tiny_init_summary.npz"), **summary)
    print("vssm_tiny params", int(summary["n_params"]))


def config0_case(ref):
    """BASELINE.json configs[0]: the reference's own CPU-runnable case -- VSSM tiny (depths 2-2-2-2, dims 96..768,
    seed 1337 as train_fully_supervised_2D_VIM.py:243-246), eval mode, one synthetic 1x1x224x224 slice repeated to
    3 channels (vision_mamba.py:43-44), 4 classes, sequential selective_scan_ref path.  The model is NOT stored: the
    same seed reproduces its initialisation bit for bit (vssm_tiny_init_summary.npz pins that)."""
    ref.selective_scan_fn = selective_scan_ref
    torch.manual_seed(1337)
    m = ref.VSSM(patch_size=4, in_chans=3, num_classes=4, depths=[2, 2, 2, 2], drop_path_rate=0.2).eval()
    x = torch.rand(1, 1, 224, 224, generator=torch.Generator().manual_seed(224))
    with torch.no_grad():
        y = m(x.repeat(1, 3, 1, 1))
    np.savez_compressed(os.path.join(HERE, "config0_vssm_tiny_224.npz"), x=np_(x), logits=np_(y))
    print("config0", y.shape, float(y.abs().mean()))


def vssm_grad_case(ref):
    """Gradients of the whole network: the small VSSM of vssm_case (same seed and constructor, so vssm_small.npz holds
    its state dict), eval mode (DropPath inert), batch 2, loss = sum(logits * g): logits, dx and the gradient of
    EVERY parameter as the reference's autograd computes them on top of the sequential scan."""
    ref.selective_scan_fn = selective_scan_ref
    torch.manual_seed(1337)
    m = ref.VSSM(patch_size=4, in_chans=3, num_classes=4, depths=[1, 1, 1, 1], dims=[16, 32, 64, 128],
                 drop_path_rate=0.2).eval()
    gen = torch.Generator().manual_seed(7)
    x = torch.rand(2, 3, 32, 32, generator=gen).requires_grad_(True)
    y = m(x)
    g = torch.randn(y.shape, generator=gen)
    (y * g).sum().backward()
    out = {"x": np_(x), "g": np_(g), "logits": np_(y), "dx": np_(x.grad)}
    for k, p in m.named_parameters():
        out["grad." + k] = np_(p.grad)
    np.savez_compressed(os.path.join(HERE, "vssm_small_grads.npz"), **out)
    print("vssm_small_grads", y.shape, len(out))


def unet_case():
    """The reference's convolutional UNet (code/networks/unet.py:304-321; model1 of train_Semi_Mamba_UNet.py:302-303):
    state-dict keys and shapes, and the eval-mode output of the seed-5 model on a fixed input (same seed -> same init)."""
    spec = importlib.util.spec_from_file_location("ref_unet", "/root/reference/code/networks/unet.py")
    U = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(U)
    torch.manual_seed(5)
    m = U.UNet(in_chns=1, class_num=4).eval()
    x = torch.rand(2, 1, 32, 32, generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        y = m(x)
    sd = m.state_dict()
    np.savez_compressed(os.path.join(HERE, "unet_ref.npz"), x=np_(x), y=np_(y), names=np.array(list(sd)),
                        shapes=np.array([str(tuple(v.shape)) for v in sd.values()]))
    print("unet_ref", y.shape, len(sd))


def losses_case():
    """The reference's own loss code (code/utils/losses.py: DiceLoss, ConstraLoss) composed exactly as its scripts do:
    supervised  0.5 * (CE + Dice(softmax))            train_fully_supervised_2D_VIM.py:153-157
    semi        cross pseudo supervision + ConstraLoss train_Semi_Mamba_UNet.py:414-446
    on fixed random logits: loss values and the gradients w.r.t. the logits."""
    spec = importlib.util.spec_from_file_location("ref_losses", "/root/reference/code/utils/losses.py")
    L = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(L)
    out = {}
    gen = torch.Generator().manual_seed(11)
    for tag, (B, C, S) in {"sup4": (3, 4, 16), "sup9": (2, 9, 12)}.items():
        logits = (2.0 * torch.randn(B, C, S, S, generator=gen)).requires_grad_(True)
        labels = torch.randint(0, C, (B, S, S), generator=gen)
        ce, dice = nn.CrossEntropyLoss(), L.DiceLoss(C)
        soft = torch.softmax(logits, dim=1)
        loss = 0.5 * (dice(soft, labels.unsqueeze(1)) + ce(logits, labels.long()))
        loss.backward()
        out.update({f"{tag}.logits": np_(logits), f"{tag}.labels": labels.numpy().astype(np.uint8), f"{tag}.loss": np_(loss),
                    f"{tag}.dlogits": np_(logits.grad)})
    B, C, S, lb, cw = 4, 4, 16, 2, 0.37
    o1 = (2.0 * torch.randn(B, C, S, S, generator=gen)).requires_grad_(True)
    o2 = (2.0 * torch.randn(B, C, S, S, generator=gen)).requires_grad_(True)
    labels = torch.randint(0, C, (B, S, S), generator=gen)
    ce, dice = nn.CrossEntropyLoss(), L.DiceLoss(C)
    s1, s2 = torch.softmax(o1, dim=1), torch.softmax(o2, dim=1)
    loss1 = 0.5 * (ce(o1[:lb], labels[:lb].long()) + dice(s1[:lb], labels[:lb].unsqueeze(1)))
    loss2 = 0.5 * (ce(o2[:lb], labels[:lb].long()) + dice(s2[:lb], labels[:lb].unsqueeze(1)))
    p1 = torch.argmax(s1[lb:].detach(), dim=1, keepdim=False)
    p2 = torch.argmax(s2[lb:].detach(), dim=1, keepdim=False)
    ps1, ps2 = dice(s1[lb:], p2.unsqueeze(1)), dice(s2[lb:], p1.unsqueeze(1))
    con1 = L.ConstraLoss(o1, o2)
    loss = (loss1 + cw * ps1 + 0.5 * con1) + (loss2 + cw * ps2 + 0.5 * con1)
    loss.backward()
    out.update({"semi.o1": np_(o1), "semi.o2": np_(o2), "semi.labels": labels.numpy().astype(np.uint8), "semi.loss": np_(loss),
                "semi.do1": np_(o1.grad), "semi.do2": np_(o2.grad), "semi.labeled_bs": np.array(lb), "semi.consistency_weight": np.array(cw)})
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses", {k: float(v) for k, v in out.items() if k.endswith(".loss")})


if __name__ == "__main__":
    import sys
    torch.set_num_threads(8)
    if len(sys.argv) > 1 and sys.argv[1] == "losses":
        losses_case()
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "unet":
        unet_case()
        sys.exit(0)
    ref = load_reference()
    if len(sys.argv) > 1 and sys.argv[1] == "config0":     # add this fixture without rewriting the others
        config0_case(ref)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "vssm_grads":
        vssm_grad_case(ref)
        sys.exit(0)
    index_tables(ref)
    ss2d_case(ref, "ss2d_2x8x5x7_s0.npz", 2, 8, 5, 7, seed=0, with_scan_args=True)
    ss2d_case(ref, "ss2d_1x16x8x8_s1.npz", 1, 16, 8, 8, seed=1, with_scan_args=True)
    ss2d_case(ref, "ss2d_2x96x14x14_s0.npz", 2, 96, 14, 14, seed=0, with_scan_args=False)
    vssm_case(ref)
    config0_case(ref)
    vssm_grad_case(ref)
    losses_case()
    unet_case()
