cd /tmp; export TMPDIR=/tmp
R=/root/repo
mkdir -p $R/gpurun_out/prof4
cd $R
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof4 -o s -- python bench.py --steps 20 --warmup 5 --no-cpu > gpurun_out/b_prof.json 2> gpurun_out/b_prof.err
