mkdir -p gpurun_out/c1
(timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/c1/pytest.log 2>&1; echo "pytest exit $?" >> gpurun_out/c1/pytest.log)
timeout 300 tools/micro/bin/pipes_bench > gpurun_out/c1/pipes.txt 2>&1
timeout 60 tools/micro/bin/pipes_bench unaligned > gpurun_out/c1/unaligned.txt 2>&1
timeout 600 python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err
tail -5 gpurun_out/c1/pytest.log; cat gpurun_out/c1/unaligned.txt; tail -c 1500 gpurun_out/c1/bench.json
