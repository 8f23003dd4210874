import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
torch.manual_seed(0)
probs = torch.softmax(torch.randn(24, 4, 224, 224, device="cuda"), dim=1).requires_grad_(True)
labels = torch.randint(0, 4, (24, 224, 224), device="cuda", dtype=torch.uint8)
def run(with_grad):
    target = labels.unsqueeze(1)
    ar = torch.arange(4, device=target.device).view(1, -1, 1, 1)
    eq = target == ar
    onehot = eq.to(probs.dtype)
    a = (probs * onehot).sum((0, 2, 3)); b = (onehot * onehot).sum((0, 2, 3)); c = (probs * probs).sum((0, 2, 3))
    stats = torch.stack([a, b, c])
    inter, ysum, zsum = stats
    loss = (1.0 - (2 * inter + 1e-5) / (zsum + ysum + 1e-5)).mean()
    res = {"arange": ar.float(), "onehot_sum": onehot.sum().reshape(1), "a": a, "b": b, "c": c, "loss": loss.reshape(1)}
    if with_grad:
        res["grad"] = torch.autograd.grad(loss, probs)[0]
    return {k: v.detach() for k, v in res.items()}
for with_grad in (False, True):
    ref = {k: v.clone() for k, v in run(with_grad).items()}
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s): run(with_grad)
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g): out = run(with_grad)
    for k in range(3):
        g.replay(); torch.cuda.synchronize()
        print("with_grad", with_grad, "replay", k, {n: "%.2e" % float((out[n].float() - ref[n].float()).abs().max()) for n in out}, flush=True)
