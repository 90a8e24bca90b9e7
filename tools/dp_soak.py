#!/usr/bin/env python3
"""Data-parallel soak: `engine.trainer.do_train` on W ranks that share the visible GPU over gloo (tests/dp_gpu_worker.py's loop,
lengthened): ranks start from DIFFERENT seeds, are made replicas by the initial broadcast, train N iterations on their shards -
two eager steps, then the recorded step replayed in segments around its collectives - with the replica digest checked on EVERY
iteration and a sha-256 of the whole state compared at the end.
GPU box:  python tools/dp_soak.py [iterations] [ranks]"""
import os, socket, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
procs = []
t0 = time.perf_counter()
for r in range(W):
    env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(W), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               OMP_NUM_THREADS="8", HSA_ENABLE_IPC_MODE_LEGACY="0", TRID_DIST_BACKEND="gloo", TRID_DP_TRAINER="1",
               TRID_DP_TRAINER_CAPTURE="1", TRID_DP_TRAINER_ITERS=str(N))
    procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_gpu_worker.py")], env=env,
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
try:
    outs = [p.communicate(timeout=1500)[0] for p in procs]
finally:
    for p in procs:  # (exactly the processes started here)
        if p.poll() is None:
            p.kill()
rc = max(p.returncode for p in procs)
keep = [ln[:200] for ln in outs[0].splitlines() if ln.startswith("DP_")]
print("\n".join(keep[-12:]))
print("ranks %d, iterations %d, exit %d, %.1f s" % (W, N, rc, time.perf_counter() - t0))
if rc != 0:
    print(outs[0][-3000:])
sys.exit(rc)
