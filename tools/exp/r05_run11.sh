#!/bin/bash
# round 5, run 11: full GPU suite + bench after the stream-kernel changes (compiler-visible wait ahead of the loop; retrieval: staged
# candidates, late scan, DMA spread, segments)
mkdir -p gpurun_out/r05_run11
O=gpurun_out/r05_run11
python -m pytest tests -x -q -m gpu > $O/tests.txt 2>&1; tail -3 $O/tests.txt
python bench.py > $O/bench.json 2> $O/bench.err; cat $O/bench.json
bash tools/exp/r05_sq_retrieval.sh > /dev/null 2>&1; cp gpurun_out/sq_retrieval.txt $O/sq_retrieval.txt; cat $O/sq_retrieval.txt
