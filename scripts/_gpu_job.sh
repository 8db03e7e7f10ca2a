#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for r in default 256 64 16 8 2; do
  if [ $r = default ]; then unset ARTEMIS_STAGE2D_RGRID; else export ARTEMIS_STAGE2D_RGRID=$r; fi
  timeout 300 python bench.py --workload ssheet_dust --n 1024 --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('rgrid $r', d['value'], d['roofline']['launch_ms'], d['roofline']['frac'])"
done
unset ARTEMIS_STAGE2D_RGRID
timeout 300 python bench.py --workload ssheet_dust --n 4096 --no-cpu-baseline --steps 50 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('4096', d['value'], d['roofline']['launch_ms'], d['roofline']['frac'])"
timeout 600 python -m pytest tests/test_parity_stage_general.py -q -m gpu -k "row or 2d or stage2d or vanishing" 2>&1 | tail -2
