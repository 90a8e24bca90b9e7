"""GPU box: the eval-mode image encoder on TWO streams (two independent half batches co-running: one's HBM-bound layer1 / stem
kernels under the other's MFMA-bound convolutions, partly empty grids filled) against one stream at the full batch.
usage: python tools/exp/eval_two_streams.py [total batch ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle.visual as OV
from textreid_amd.backbones.m_resnet import ModifiedResNet

spec = OV.RN50
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [256, 512]
torch.manual_seed(0)
m = ModifiedResNet(list(spec.layers), spec.output_dim, spec.heads, spec.last_stride, (spec.height, spec.in_width), spec.width).cuda()
with torch.no_grad():
    m.train()
    for _ in range(3):
        m(torch.randn(32, 3, 384, 128, device="cuda"))
    m.eval()
    for B in sizes:
        x = torch.randn(B, 3, 384, 128, device="cuda")
        ref = m(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            m(x)
        torch.cuda.synchronize()
        one = 10 * B / (time.perf_counter() - t0)
        for parts in (2, 4):
            xs = [c.contiguous() for c in x.chunk(parts)]
            streams = [torch.cuda.Stream() for _ in range(parts)]
            main = torch.cuda.current_stream()

            def run():
                outs = []
                for s, xc in zip(streams, xs):
                    s.wait_stream(main)
                    with torch.cuda.stream(s):
                        outs.append(m(xc))
                for s in streams:
                    main.wait_stream(s)
                return torch.cat(outs)

            got = run()
            torch.cuda.synchronize()
            err = float((got - ref).abs().max() / ref.abs().max())
            t0 = time.perf_counter()
            for _ in range(10):
                run()
            th = time.perf_counter() - t0
            torch.cuda.synchronize()
            two = 10 * B / (time.perf_counter() - t0)
            print("B %4d: one stream %7.0f imgs/s | %d streams x %3d %7.0f imgs/s (%+.1f %%), host enqueue %.1f ms / round, max rel diff %.1e"
                  % (B, one, parts, B // parts, two, 100 * (two / one - 1), th * 100, err), flush=True)
