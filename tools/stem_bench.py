"""GPU box: the stem's convolutions at the benchmarked size (B = 128, 384x128 input -> 192x64 maps), old kernels against
the bandwidth-shaped ones of csrc/stem_conv.hip.  Prints microseconds per launch and the HBM rate of the ALGORITHMIC
bytes (input once + output once)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from textreid_amd import ops  # noqa: E402

dev = torch.device("cuda")
B, H, W = 128, 192, 64
M = B * H * W
g = torch.Generator().manual_seed(0)


def t(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def line(name, us, nbytes):
    print("%-58s %8.1f us   %6.2f TB/s of %6.0f MB" % (name, us, nbytes / us / 1e6, nbytes / 1e6), flush=True)


# ---- conv1
img = torch.randn(B, 3, 384, 128, generator=g).to(dev)
w1 = (torch.randn(32, 3, 3, 3, generator=g) * 0.3).to(dev)
w1p = torch.zeros(32, 28, device=dev)
w1p[:, :27] = w1.reshape(32, 27)


def old_conv1():
    col, Ho, Wo = ops.stem_im2col(img)
    return ops.conv1x1(col, w1p, stats=True)


by1 = img.numel() * 4 + M * 32 * 4
line("conv1 3->32 s2: im2col + GEMM (old)", t(old_conv1), by1)
line("conv1 3->32 s2: direct fp32-MFMA kernel", t(lambda: ops.stem_conv1(img, w1)), by1)

dy1 = torch.randn(B, H, W, 32, generator=g).to(dev)
line("wgrad conv1: im2col + split GEMM [32, 28] (old)", t(lambda: ops.conv1x1_wgrad(dy1, ops.stem_im2col(img)[0]), reps=5), by1)
line("wgrad conv1: direct fp32-MFMA kernel + slab fold", t(lambda: ops.stem_conv1_wgrad(img, dy1), reps=5), by1)
del dy1

# ---- 3x3 convs
for (C, N, tag) in ((32, 32, "conv2 / dgrad conv2"), (32, 64, "conv3"), (64, 32, "dgrad conv3")):
    x = torch.relu(torch.randn(B, H, W, C, generator=g)).to(dev)
    w = (torch.randn(N, 9 * C, generator=g) * 0.1).to(dev)
    ax, aw = ops.amax(x), ops.amax(w)
    xp, wp = ops.p16_pack(x, ax), ops.p16_pack(w, aw)
    nb = M * (C + N) * 4
    line("%s %d->%d: on-the-fly split GEMM (old)" % (tag, C, N), t(lambda: ops.conv3x3(x, w, stats=True, prec=16, aa=ax, ba=aw)), nb)
    line("%s %d->%d: P16 implicit GEMM" % (tag, C, N), t(lambda: ops.conv_p16(xp, wp, conv3=True)), nb)
    line("%s %d->%d: ring-of-rows kernel, with BatchNorm partials" % (tag, C, N), t(lambda: ops.conv3x3_halo_p16(xp, wp)), nb)
    line("%s %d->%d: ring-of-rows kernel, no partials" % (tag, C, N), t(lambda: ops.conv3x3_halo_p16(xp, wp, stats=False)), nb)
    del x, xp

# ---- weight gradients
for (C, N, tag) in ((32, 32, "wgrad conv2"), (32, 64, "wgrad conv3")):
    x = torch.relu(torch.randn(B, H, W, C, generator=g)).to(dev)
    dy = torch.randn(B, H, W, N, generator=g).to(dev)
    ax, ad = ops.amax(x), ops.amax(dy)
    xp, dp = ops.p16_pack(x, ax), ops.p16_pack(dy, ad)
    nb = M * (C + N) * 4
    line("%s: dy[M,%d]^T x gather(x[M,%d]) library default (old: %s)" % (tag, N, C, "exact fp32" if N == 32 else "split"),
         t(lambda: ops.conv3x3_wgrad(dy, x) if N == 32 else ops.conv3x3_wgrad(dy, x, prec=16, aa=ad, ba=ax), reps=5), nb)
    line("%s: transposing P16 kernel" % tag, t(lambda: ops.wgrad_p16(dp, xp, conv=(H, W, C)), reps=5), nb)
    line("%s: ring-of-rows kernel + slab fold" % tag, t(lambda: ops.conv3x3_wgrad_halo_p16(dp, xp), reps=5), nb)
    del x, dy, xp, dp
x = torch.relu(torch.randn(B, 96, 32, 64, generator=g)).to(dev)
dy = torch.randn(B, 96, 32, 64, generator=g).to(dev)
xp, dp = ops.p16_pack(x), ops.p16_pack(dy)
nb = B * 96 * 32 * 128 * 4
line("wgrad layer1 conv2 64->64 (96x32): transposing P16 kernel", t(lambda: ops.wgrad_p16(dp, xp, conv=(96, 32, 64)), reps=5), nb)
line("wgrad layer1 conv2 64->64 (96x32): ring-of-rows kernel + slab fold", t(lambda: ops.conv3x3_wgrad_halo_p16(dp, xp), reps=5), nb)
del x, dy, xp, dp

# ---- the BatchNorm passes of the stem (for the byte budget)
y = torch.randn(B, H, W, 32, generator=g).to(dev)
st = ops.BNState(32, y)
for tns in (st.mean, st.shift):
    tns.zero_()
st.invstd.fill_(1.0)
st.scale.fill_(1.0)
bound = ops.amax(y)
line("bn_apply fp32 -> fp32 [M,32]", t(lambda: ops.bn_apply(y, st, relu=True)), 2 * M * 32 * 4)
line("bn_apply fp32 -> P16 [M,32]", t(lambda: ops.bn_apply_p16(y, st, bound, relu=True)), 2 * M * 32 * 4)
y3 = torch.randn(B, H, W, 64, generator=g).to(dev)
st3 = ops.BNState(64, y3)
for tns in (st3.mean, st3.shift):
    tns.zero_()
st3.invstd.fill_(1.0)
st3.scale.fill_(1.0)
line("bn_apply_pool2 fp32 -> fp32 [M,64] -> [M/4,64]", t(lambda: ops.bn_apply_pool2(y3, st3, relu=True)), M * 64 * 4 * 1.25)
line("bn_apply_pool2 fp32 -> P16", t(lambda: ops.bn_apply_pool2_p16(y3, st3, ops.amax(y3), relu=True)), M * 64 * 4 * 1.25)
gq = torch.randn(B, H // 2, W // 2, 64, generator=g).to(dev)
line("bn_bwd (pooled, relu) [M,64]: reduce + apply -> fp32", t(lambda: ops.bn_bwd(gq, y3, st3, None, 1, pooled=True)), M * 64 * 4 * 3.5)
line("bn_bwd (pooled, relu) [M,64]: reduce_bound + apply -> P16", t(lambda: ops.bn_bwd_p16(gq, y3, st3, 1, pooled=True)), M * 64 * 4 * 3.5)
g2 = torch.randn(B, H, W, 32, generator=g).to(dev)
line("bn_bwd (relu) [M,32]: reduce + apply -> fp32", t(lambda: ops.bn_bwd(g2, y, st, None, 1)), M * 32 * 4 * 5)
line("bn_bwd (relu) [M,32]: reduce_bound + apply -> P16", t(lambda: ops.bn_bwd_p16(g2, y, st, 1)), M * 32 * 4 * 5)
