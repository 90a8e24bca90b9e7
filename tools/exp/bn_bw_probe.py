"""How far are the BatchNorm passes from what a plain elementwise kernel of the same traffic reaches on this box?
bn_bwd_apply_p16 (reads g, y; writes dy: 12 B / element) against torch's z = x + y (12 B / element) and a copy (8 B / element);
bn_apply_p16 (reads y, writes z: 8 B / element) against the copy."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from textreid_amd import ops

dev = torch.device("cuda:0")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3  # us


for (B, H, W, C) in ((128, 96, 32, 256), (128, 48, 16, 512), (128, 24, 8, 1024), (128, 96, 32, 64), (128, 24, 8, 2048)):
    n = B * H * W * C
    x = torch.randn(B, H, W, C, device=dev)
    y = torch.randn(B, H, W, C, device=dev)
    z = torch.empty_like(x)
    t_add = timeit(lambda: torch.add(x, y, out=z))
    t_cpy = timeit(lambda: z.copy_(x))
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    y2, part = ops.conv1x1(x.reshape(B, H, W, C), torch.eye(C, device=dev), stats=True, prec=0) if C <= 256 else (x, None)
    if part is None:
        # statistics through a plain pass (only the timing of the apply kernels matters here)
        m, v = x.reshape(-1, C).mean(0), x.reshape(-1, C).var(0, unbiased=False)

        class St:
            pass

        st = St()
        st.mean, st.invstd = m.contiguous(), (1 / torch.sqrt(v + 1e-5)).contiguous()
        st.scale, st.shift = (gamma * st.invstd).contiguous(), (beta - gamma * st.invstd * m).contiguous()
        y2 = x
    else:
        st = ops.bn_finalize(part, B * H * W, gamma, beta, None, None)
    g = y
    t_bwd = timeit(lambda: ops.bn_bwd_p16(g, y2, st, 1))
    bound = ops.amax(y2)
    t_fwd = timeit(lambda: ops.bn_apply_p16(y2, st, bound, relu=True))
    t_f32 = timeit(lambda: ops.bn_apply(y2, st, relu=True))
    print("[%d,%d,%d,%d] %6.1f MB/tensor | add %6.1f us %5.2f TB/s | copy %6.1f us %5.2f TB/s | bn_bwd (reduce+final+apply, 20 B/el) %6.1f us %5.2f TB/s | bn_apply (8 B/el) P16 out %6.1f us %5.2f TB/s, fp32 out %6.1f us %5.2f TB/s"
          % (B, H, W, C, n * 4 / 1e6, t_add, 12 * n / t_add / 1e6, t_cpy, 8 * n / t_cpy / 1e6, t_bwd, 20 * n / t_bwd / 1e6, t_fwd, 8 * n / t_fwd / 1e6,
             t_f32, 8 * n / t_f32 / 1e6))
