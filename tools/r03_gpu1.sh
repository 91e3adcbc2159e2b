set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_a_gputests.log 2>&1 || { tail -40 gpurun_out/r03_a_gputests.log; exit 1; }
tail -3 gpurun_out/r03_a_gputests.log
python bench.py > gpurun_out/r03_a_bench.json 2> gpurun_out/r03_a_bench.err
cat gpurun_out/r03_a_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['us_per_iteration'], d['ms_per_step'])"
timeout -k 10 300 python tools/time_fortran.py 1000000 > gpurun_out/r03_a_fortran_timing.txt 2>&1 || true
cat gpurun_out/r03_a_fortran_timing.txt
timeout -k 10 500 python tools/soak_production.py > gpurun_out/r03_a_soak_production.txt 2>&1 || true
tail -6 gpurun_out/r03_a_soak_production.txt
