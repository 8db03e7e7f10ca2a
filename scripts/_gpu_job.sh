#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
for m in 2 1; do
timeout 600 python bench.py --loopback --overlap-mode $m --steps 50 --no-cpu-baseline 2>gpurun_out/lb_$m.err | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('loopback mode $m', d['value'], d['ms_per_step'], d['config'].get('overlap_mode'), d['config'].get('overlap_wait_timeout'), d['config'].get('transport','')[:60])"
tail -2 gpurun_out/lb_$m.err | cut -c1-200
done
ARTEMIS_FORCE_OVERLAP=1 timeout 600 python bench.py --blocks-per-gpu 2 --overlap-mode 2 --steps 50 --no-cpu-baseline --no-dropin 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2 blocks forced overlap', d['value'], d['ms_per_step'], d['config'].get('overlap_mode'))"
timeout 600 python bench.py --blocks-per-gpu 2 --steps 50 --no-cpu-baseline --no-dropin 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('2 blocks', d['value'], d['ms_per_step'], d['config'].get('overlap_mode'))"
