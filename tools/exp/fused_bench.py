"""GPU box: conv3 + bn3 + identity + ReLU of the identity blocks of layer1 / layer2 at B = 128: three kernels (GEMM, finalize,
apply) against statistics-only pass + finalize + fused pass; us per launch."""
import os, sys
import torch as T
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textreid_amd import ops
dev = T.device("cuda")
def t(fn, reps=20):
    fn(); T.cuda.synchronize()
    e0, e1 = T.cuda.Event(enable_timing=True), T.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); T.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for (M, N, K) in ((393216, 256, 64), (98304, 512, 128), (24576, 1024, 256)):
    x, w, ident = T.relu(T.randn(M, K, device=dev)), T.randn(N, K, device=dev) * 0.2, T.relu(T.randn(M, N, device=dev))
    xp, wp, ip = ops.p16_pack(x), ops.p16_pack(w), ops.p16_pack(ident)
    del x, ident
    g, b = T.ones(N, device=dev), T.zeros(N, device=dev)
    y, st = ops.conv_p16(xp, wp)
    bound = ops.amax_slot(dev)
    fin = ops.bn_finalize_minmax(st, M, g, b, None, None, False, bound)
    print("M %d N %d K %d" % (M, N, K))
    print("  GEMM with partials, y stored      %7.1f us" % t(lambda: ops.conv_p16(xp, wp)))
    print("  bn_apply + identity + mask        %7.1f us" % t(lambda: ops.bn_apply_p16(y, fin, bound, relu=True, res=ip, bound_res=ip.amax, want_mask=True)))
    print("  bn_apply + identity (no mask)     %7.1f us" % t(lambda: ops.bn_apply_p16(y, fin, bound, relu=True, res=ip, bound_res=ip.amax)))
    print("  statistics-only pass              %7.1f us" % t(lambda: ops.conv1x1_stats_p16(xp, wp)))
    print("  fused pass, y kept, mask          %7.1f us" % t(lambda: ops.conv1x1_bn_res_p16(xp, wp, fin, bound, ip, want_mask=True, keep_y=True)))
    print("  fused pass, no y, no mask         %7.1f us" % t(lambda: ops.conv1x1_bn_res_p16(xp, wp, fin, bound, ip)))
