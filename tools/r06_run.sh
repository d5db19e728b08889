export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 600 python bench.py --timed-only > $O/bench_timed.json 2> $O/bench_timed.err; tail -c 1500 $O/bench_timed.json
rm -rf /tmp/zt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/zt -o k -- python $PYTHONPATH/bench.py --timed-only --steps 8 --warmup 6 ) > $O/zt.log 2>&1
f=$(find /tmp/zt -name "*kernel_trace.csv" | head -1)
head -1 $f
python tools/zero_users.py $f > $O/zero_users.txt 2>&1; cat $O/zero_users.txt
