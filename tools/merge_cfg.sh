#!/bin/bash
for cfg in 0 1 2 3 4 5 6; do
  DC3HIP_MERGE_CFG=$cfg python bench.py --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('cfg $cfg', round(d['ms_per_step'],2), d['verify'], 'merge', d['roofline_path']['phase_ms']['merge'])"
done
