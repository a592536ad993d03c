# Runs ON THE GPU BOX: every example driver and evaluation script under its reference entry point, results as JSON under
# gpurun_out/ex/ (tools/summarize_examples.py condenses them into the record kept in profiles/).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/ex
timeout 900 python3 examples/example_pandas_Jointspace.py --steps 7000 > gpurun_out/ex/jointspace.json 2> gpurun_out/ex/jointspace.err
timeout 900 python3 examples/example_pandas_cartesian.py --steps 7000 > gpurun_out/ex/cartesian.json 2> gpurun_out/ex/cartesian.err
timeout 900 python3 examples/example_pandas_Jointspace.py --steps 7000 --scenes 64 > gpurun_out/ex/jointspace_64scenes.json 2> gpurun_out/ex/jointspace_64.err
timeout 600 python3 examples/example_pointmasses_static.py > gpurun_out/ex/point_static.json 2> gpurun_out/ex/point_static.err
timeout 600 python3 examples/example_pointmasses_dynamic.py > gpurun_out/ex/point_dynamic.json 2> gpurun_out/ex/point_dynamic.err
timeout 600 python3 examples/evaluation/evaluate_horizon.py --steps 100 --out gpurun_out/ex/results_horizon > gpurun_out/ex/horizon.json 2> gpurun_out/ex/horizon.err
timeout 1500 python3 examples/evaluation/evaluate_random_dynamic_scenarios.py --runs 16 --steps 7000 > gpurun_out/ex/random.txt 2> gpurun_out/ex/random.err
timeout 1500 python3 examples/evaluation/evaluate_random_dynamic_scenarios.py --device --scenarios 512 --steps 4000 --blocks 2 > gpurun_out/ex/random_device.json 2> gpurun_out/ex/random_device.err
python3 tools/summarize_examples.py > gpurun_out/ex/summary.json
tail -c 1500 gpurun_out/ex/summary.json
