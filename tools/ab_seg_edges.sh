#!/bin/bash
# entries per SpMM segment (knob spmm_seg_edges, default 32) on the round-2 graphs, inside the step
for w in whole_graph whole_graph_pathway; do for v in 32 16 24 48 64 32; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --no-lazy-top --set spmm_seg_edges=$v 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernel_us']; print('$w seg_edges=$v', round(r['ms_per_step'],4), round(r['long_run']['ms_per_step'],4), {x:round(k[x],1) for x in k if x.startswith('spmm')})"; done; done
