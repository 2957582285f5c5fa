cd /root/repo
mkdir -p gpurun_out
timeout 1500 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/t.log 2>&1; grep -E "passed|failed|^E  " gpurun_out/t.log | head -20
timeout 600 python bench.py --steps 50 --warmup 5 --force-dist --no-cpu > gpurun_out/b_cat1.json 2> gpurun_out/b_cat1.err; tail -2 gpurun_out/b_cat1.err
timeout 900 python bench.py --steps 50 --warmup 5 --force-dist --emulate 8:3 > gpurun_out/b_cat8.json 2> gpurun_out/b_cat8.err; tail -2 gpurun_out/b_cat8.err
