cd /root/repo
mkdir -p gpurun_out
timeout 900 python tools/wallclock_config2.py > gpurun_out/wall.json 2> gpurun_out/wall.err; cat gpurun_out/wall.json
timeout 900 python tools/_prof_cli.py 2> gpurun_out/cliprof.txt >/dev/null
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
