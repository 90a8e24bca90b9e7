import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textreid_amd import ops
torch.manual_seed(0)
B, C, H, W, N = 2, 64, 96, 32, 64
x, w = torch.randn(B, C, H, W), torch.randn(N, C, 3, 3) * 0.1
y_ref = F.conv2d(x, w, padding=1).permute(0, 2, 3, 1)
xp = ops.p16_pack(x.permute(0, 2, 3, 1).contiguous().cuda())
wp = ops.p16_pack(w.permute(0, 2, 3, 1).reshape(N, 9 * C).contiguous().cuda())
for cpi in (0, 48, 12, 1):
    y = ops.conv3x3_halo_p16(xp, wp, stats=False, chunks_per_image=cpi).cpu()
    bad = ((y - y_ref).abs() > 1e-3).nonzero()
    print("cpi", cpi, "bad", bad.shape[0], "of", y.numel())
    if bad.shape[0]:
        print(" b", bad[:, 0].unique().tolist()[:10], "y", bad[:, 1].unique().tolist()[:20], "x", bad[:, 2].unique().tolist()[:40], "n", bad[:, 3].unique().tolist()[:70])
        print(bad[:12].tolist())
