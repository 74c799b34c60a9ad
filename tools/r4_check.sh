#!/bin/bash
# round 4: RCCL world-of-one tests, option rejection, default bench line with the hs block and the CPU baseline
set -o pipefail
O=gpurun_out/r4d; mkdir -p $O
python -m pytest tests/test_rccl_gpu.py tests/test_gpu_parity.py -x -q -m gpu -k "rccl or launcher or open_errors or both_range or soak" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 900 python bench.py > $O/bench_default.json 2> $O/bench_default.log || { tail -30 $O/bench_default.log; exit 1; }
tail -4 $O/bench_default.log
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4d/bench_default.json').read().strip().splitlines()[-1])
print('value', d['value']/1e9, 'ms', d['ms_per_step'], 'wall', d['bench_wall_s'])
r=d['roofline']; print({k:r[k] for k in ('kernel','frac','avg_launch_ms','kernel_ms_per_step_one_stream','dominant_kernel_ms_per_step_one_stream','ms_per_step','overlap_gain','traffic_source')})
c=d['cpu_baseline']; print({k:c[k] for k in ('value','cores','threads','cpu_quota','cpus_visible','positions_per_s_per_granted_cpu')}, [(x['threads'], x['value']) for x in c['runs']])
h=d['hs']; print('hs', h['value']/1e9, h['roofline']['kernel'], h['roofline']['frac'], {k:v['value']/1e9 for k,v in h['list_mode'].items()}, h['pipeline']['resolve'])
print('configs1', d['configs1']['value']/1e9, 'e2e', d['end_to_end']['cli_search_s'], d['end_to_end']['cli_search_s_second_call'])
PY
