set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r03_i_gputests.log 2>&1 || { tail -60 gpurun_out/r03_i_gputests.log; exit 1; }
tail -3 gpurun_out/r03_i_gputests.log
( timeout -k 10 500 python tools/stress_rejections.py 20000 > gpurun_out/r03_i_stress_rejections.txt 2>&1; echo "stress rc $?" )
tail -2 gpurun_out/r03_i_stress_rejections.txt
bash tools/shape_table.sh > gpurun_out/r03_i_shapes.txt 2>&1
cat gpurun_out/r03_i_shapes.txt
