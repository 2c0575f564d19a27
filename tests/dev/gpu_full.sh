# full GPU validation: test suite, smoke, default bench (with the CPU baseline and the secondary workloads)
timeout 1400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
mkdir -p gpurun_out
( time timeout 900 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err ) 2>&1 | grep real
tail -c 6000 gpurun_out/bench_full.json
