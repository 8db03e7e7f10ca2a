# scratch: the command of the most recent ad-hoc `gpurun -- 'bash scripts/_gpu_job.sh'` call (see scripts/final_gpu_job.sh
# for the evidence run)
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/r03a_tests.txt
timeout 600 python bench.py --steps 100 > gpurun_out/r03a_bench_line.json 2> gpurun_out/r03a_bench.err; cut -c1-600 gpurun_out/r03a_bench_line.json
