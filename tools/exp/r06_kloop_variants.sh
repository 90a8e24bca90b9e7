#!/bin/bash
# r06c: 192x128 one-workgroup-per-CU tiles (variants 15/16) against the 128x128 loops; one workgroup per CU forced on the 128x128
# loops (TRID_GEMM_LDS_PAD=96); weight gradients with and without the software pipeline
O=gpurun_out/r06c; mkdir -p $O
python tools/kloop_bench.py 3 12 15 16 > $O/kloop.txt 2>&1
TRID_GEMM_LDS_PAD=96 python tools/kloop_bench.py 3 12 > $O/kloop_one_wg_per_cu.txt 2>&1
TRID_WGRAD_SP=0 python tools/kloop_bench.py --wgrad > $O/wgrad_sp0.txt 2>&1
TRID_WGRAD_SP=1 python tools/kloop_bench.py --wgrad > $O/wgrad_sp1.txt 2>&1
tail -4 $O/kloop.txt; tail -2 $O/wgrad_sp0.txt $O/wgrad_sp1.txt
