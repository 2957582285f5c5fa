cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|^E  " gpurun_out/t.log | head -20
timeout 600 python bench.py --steps 50 --warmup 5 > gpurun_out/b_single.json 2> gpurun_out/b_single.err; tail -2 gpurun_out/b_single.err
