# scratch: the command of the most recent ad-hoc `gpurun -- 'bash scripts/_gpu_job.sh'` call (see scripts/final_gpu_job.sh
# for the evidence run)
cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
