"""What a plain streaming pass reaches on this GPU with cold operands: torch's bf16 copy over buffers rotated through more
than 1 GB (nothing survives in the 256 MB memory-side cache), per-copy time by HIP events.  The yardstick for the
memory-bound kernels of the training step: 4.5 TB/s at 29 MB per operand, 5.3 TB/s at 231 MB."""
import torch

for mb in (29, 58, 116, 231):
    n = mb * 1000 * 1000 // 2
    nbuf = max(2, int(1200 / mb))
    xs = [torch.randn(n, device="cuda").bfloat16() for _ in range(nbuf)]
    ys = [torch.empty_like(xs[0]) for _ in range(nbuf)]
    for i in range(nbuf):
        ys[i].copy_(xs[i])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3
    a.record()
    for _ in range(reps):
        for i in range(nbuf):
            ys[i].copy_(xs[i])
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / (reps * nbuf) * 1e3
    print("%4d MB per operand: copy %7.1f us = %5.2f TB/s (read + write)" % (mb, us, 2 * mb / us))
