cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 2400 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
