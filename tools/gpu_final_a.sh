set -x
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r03_smoke.log 2>&1; tail -2 gpurun_out/r03_smoke.log
python -m pytest tests -x -q -m gpu > gpurun_out/r03_tests_full.log 2>&1; echo "rc=$?" >> gpurun_out/r03_tests_full.log; tail -3 gpurun_out/r03_tests_full.log
python tools/bench_ingest.py 48 > gpurun_out/r03_ingest.json 2> gpurun_out/r03_ingest.err; cat gpurun_out/r03_ingest.json
