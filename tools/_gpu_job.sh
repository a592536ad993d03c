cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 bench.py > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err; echo rc=$?
python3 - <<'PY'
import json
r=json.load(open('gpurun_out/bench_r04.json'))
print({k:r[k] for k in ('value','ms_per_step','rollout_kernel_ms','effective_clock_ghz')})
print(r['roofline']['frac'], r['roofline'].get('frac_at_measured_clock'), r['roofline']['roofline_inputs_stale'])
for k,v in r.get('configs',{}).items(): print(k, v.get('kernel_ms'), v.get('units_per_s'), v.get('parity_spot_check',{}).get('max_rel_err'), {x:v['roofline'].get(x) for x in ('bound','frac','traffic','roofline_inputs_stale')})
print(r['cpu_baseline']['value'], r['cpu_baseline']['cores'], r['single_scenario'])
PY
