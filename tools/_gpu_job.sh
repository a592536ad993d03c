cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python3 tools/prof_cart.py f64 8 2>&1 | tail -1 | tee -a gpurun_out/cart_ab4.jsonl
timeout 300 python3 tools/c5_error_growth.py 2>&1 | tail -60 > gpurun_out/c5_growth.log
timeout 3000 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
