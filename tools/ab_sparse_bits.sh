#!/bin/bash
# from which graph size do the bitmaps of the sparsity-aware backward hops pay?  (knob sparse_bits_rows, default 500000)
for w in rmat:60000:1200000 rmat:120000:2400000 rmat:250000:5000000 rmat:450000:9000000; do for v in 500000 1 500000 1; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --set sparse_bits_rows=$v 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernel_us']; print('$w sparse_bits_rows=$v', round(r['ms_per_step'],4), round(r['long_run']['ms_per_step'],4), {x:round(k[x],1) for x in ('spmm_bwd1','spmm_bwd2')})"; done; done
