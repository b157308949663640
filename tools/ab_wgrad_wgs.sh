#!/bin/bash
# weight-gradient launch size (knob wgrad_wgs) inside the step, config 2 and config 3
for w in whole_graph whole_graph_pathway; do for v in 256 384 512 256 512; do python bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline --set wgrad_wgs=$v 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); k=r['kernel_us']; print('$w wgrad_wgs=$v', round(r['ms_per_step'],4), round(r['long_run']['ms_per_step'],4), {x:round(k[x],1) for x in ('wgrad','adam')})"; done; done
