"""GPU box: layer1's conv2 (64 -> 64 channels, 96x32 maps, B = 128): implicit-GEMM tile kernel against the ring-of-rows kernel."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textreid_amd import ops
dev = torch.device("cuda")
B, H, W, C = 128, 96, 32, 64
x = torch.relu(torch.randn(B, H, W, C, device=dev))
w = torch.randn(C, 9 * C, device=dev) * 0.1
xp, wp = ops.p16_pack(x), ops.p16_pack(w)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for halo in (False, True, False, True):
    ops.USE_HALO_BLOCKS = halo
    print("halo" if halo else "tile", "with partials %.1f us" % t(lambda: ops.conv_p16(xp, wp, conv3=True)), "without %.1f us" % t(lambda: ops.conv_p16(xp, wp, conv3=True, stats=False)), flush=True)
for cpi in (2, 3, 4, 6, 8, 12, 16, 24, 48):
    print("halo chunks per image %2d: %.1f us" % (cpi, t(lambda: ops.conv3x3_halo_p16(xp, wp, chunks_per_image=cpi))), flush=True)
