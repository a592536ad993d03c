cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ex
timeout 900 python examples/example_pandas_Jointspace.py --steps 7000 --device-episode > gpurun_out/ex/jointspace.json 2> gpurun_out/ex/jointspace.err
timeout 900 python examples/example_pandas_cartesian.py --steps 7000 > gpurun_out/ex/cartesian.json 2> gpurun_out/ex/cartesian.err
timeout 600 python examples/evaluation/evaluate_horizon.py --steps 100 --out gpurun_out/ex/results_horizon > gpurun_out/ex/horizon.json 2> gpurun_out/ex/horizon.err
timeout 1500 python examples/evaluation/evaluate_random_dynamic_scenarios.py --runs 2 --steps 7000 > gpurun_out/ex/random.txt 2> gpurun_out/ex/random.err
tail -c 600 gpurun_out/ex/jointspace.json; tail -c 400 gpurun_out/ex/cartesian.json; tail -c 300 gpurun_out/ex/horizon.json; head -5 gpurun_out/ex/random.txt
