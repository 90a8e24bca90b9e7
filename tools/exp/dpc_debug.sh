cd $GRAFT_REPO_ROOT
export RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 TRID_DIST_BACKEND=nccl TRID_DP_FORCE=1 TRID_DP_CAPTURED=1
timeout 300 python -X faulthandler tests/dp_gpu_worker.py 2>&1 | grep -v "^DP_ERRS" | tail -60
