cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/s2_m
for g in 1 2 3; do
  dv=$(python3 -c "print(','.join(['0']*$g))")
  timeout -k 10 300 python bench.py --gpus $g --devices $dv --steps 3 --warmup 1 --cpu-seconds 0 --other-configs off --projection off 2>gpurun_out/s2_m/err_$g.log | tail -1 > gpurun_out/s2_m/bench_$g.json
  python3 -c "
import json; d=json.loads(open('gpurun_out/s2_m/bench_$g.json').read().strip().splitlines()[-1]); print($g, d['value'], d['ms_per_step'], d.get('multi_gpu',{}).get('render_ms_per_replica'), d['config'].get('spp_per_pass'))"
done
