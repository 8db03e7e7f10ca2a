cd /root/repo
export TMPDIR=/tmp
timeout 600 python bench.py --loopback --steps 50 --no-cpu-baseline --no-dropin > gpurun_out/loop.json 2> gpurun_out/loop.err; echo rc=$?; cut -c1-1000 gpurun_out/loop.json; tail -5 gpurun_out/loop.err
