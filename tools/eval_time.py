import sys, time, torch
sys.path.insert(0, '/root/repo')
import oracle.visual as OV
from textreid_amd.backbones.m_resnet import ModifiedResNet
spec = OV.RN50
m = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width).cuda().eval()
x = torch.randn(128, 3, 384, 128, device='cuda')
with torch.no_grad():
    for fold in (False, True, False, True):
        m.fold_eval_bn = fold
        m(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): m(x)
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("fold", fold, "%.2f ms/batch (host enqueue %.2f)  %.0f imgs/s" % (dt * 100, th * 100, 1280 / dt))
