cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_control.py tests/test_gpu_examples.py -x -q -m gpu 2>&1 | tail -15
