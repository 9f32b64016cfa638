cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -q -m gpu --timeout 600 -p no:cacheprovider -x 2>&1 | tail -3
timeout 600 python bench.py --group --gpus 1 --steps 3 --warmup 1 > gpurun_out/bench_group1.json 2> gpurun_out/bench_group1.err; echo "group rc=$?"; python -c "
import json;d=json.loads(open('gpurun_out/bench_group1.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'],d['hbm_budget_per_gpu']['total_GB_per_gpu'])"
