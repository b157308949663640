#!/bin/bash
# after a tools/profile_r06.sh job has been merged back: copy what is judged from gpurun_out/r06 (scratch) into profiles/ (tracked)
R=$(cd "$(dirname "$0")/.." && pwd); O=$R/gpurun_out/r06; P=$R/profiles
cp $O/r06_spmm_pmc.json $P/r06_spmm_pmc.json
for f in bench.json bench_config3.json bench_rmat_10M_200M.json; do [ -f $O/$f ] && cp $O/$f $P/r06_$f; done
[ -f $O/bench_kernel_stats.csv ] && cp $O/bench_kernel_stats.csv $P/r06_bench_kernel_stats.csv
[ -f $O/config3_kernel_stats.csv ] && cp $O/config3_kernel_stats.csv $P/r06_bench_config3_kernel_stats.csv
[ -f $O/bench_under_rocprof.txt ] && cp $O/bench_under_rocprof.txt $P/r06_bench_under_rocprof.json
[ -f $O/config3_under_rocprof.txt ] && cp $O/config3_under_rocprof.txt $P/r06_bench_config3_under_rocprof.json
python3 - <<'PY'
import json, hashlib, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) if '__file__' in globals() else os.getcwd()
z = json.load(open('profiles/r06_spmm_pmc.json'))
h = hashlib.sha256(open('gcn-drug-repurposing_amd/csrc/spmm.hip', 'rb').read()).hexdigest()
print('counters taken with spmm.hip', z['source_hash']['spmm.hip'][:16], '| tree:', h[:16], '| match:', z['source_hash']['spmm.hip'] == h)
for f in ('profiles/r06_bench.json', 'profiles/r06_bench_config3.json', 'profiles/r06_bench_rmat_10M_200M.json'):
    b = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(b['ms_per_step'], 4), 'lazy', b.get('lazy_top', {}).get('ms_per_step'), b.get('lazy_top', {}).get('ms_per_step_with_layer1_kept'),
          {k: (round(b[k]['frac'], 3), b[k].get('traffic_over_alg') and round(b[k]['traffic_over_alg'], 2)) for k in b if k.startswith('roofline') and b[k]})
PY
