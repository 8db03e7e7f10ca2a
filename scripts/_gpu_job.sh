cd /root/repo
timeout 300 python bench.py --no-cpu-baseline --no-dropin --loopback --steps 20 > gpurun_out/o1.txt 2> gpurun_out/e1.txt; wc -l gpurun_out/o1.txt; cut -c1-80 gpurun_out/o1.txt; grep -c "RCCL version" gpurun_out/e1.txt
timeout 600 python bench.py --steps 20 > gpurun_out/o2.txt 2> gpurun_out/e2.txt; wc -l gpurun_out/o2.txt; python -c "
import json; d=json.loads(open('gpurun_out/o2.txt').read()); print(d['value'], d['cpu_baseline']['value'], d['roofline']['frac'])"
