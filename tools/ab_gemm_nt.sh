#!/bin/bash
# feature-tile width of the projections (knob gemm_nt_cap, in 16-feature units) with the XCD-aware numbering, inside the step
for w in whole_graph whole_graph_pathway; do for v in 0 4 2 0 4; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --set gemm_nt_cap=$v 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernel_us']; print('$w gemm_nt_cap=$v', round(r['ms_per_step'],4), round(r['long_run']['ms_per_step'],4), {x:round(k[x],1) for x in ('dense_fwd','dgrad')})"; done; done
