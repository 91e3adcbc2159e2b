# Run ON THE GPU BOX (through gpurun): everything profiles/<tag>_* of a round comes from -- make_profiles.sh, the lock-step bench line,
# the shape table and the stress runs.  Edit T below.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T=r02_x   # <- the tag of the snapshot
bash tools/make_profiles.sh $T > gpurun_out/${T}_make_profiles.log 2>&1 || { tail -20 gpurun_out/${T}_make_profiles.log; exit 1; }
echo "profiles done"
python3 bench.py --force-lockstep --no-cpu-baseline > gpurun_out/${T}_bench_lockstep.json 2>/dev/null
bash tools/shape_table.sh > gpurun_out/${T}_shapes.txt 2>&1
cat gpurun_out/${T}_shapes.txt
( timeout -k 10 600 python tools/stress_rejections.py 20000 > gpurun_out/${T}_stress_rejections.txt 2>&1; echo "stress rc $?" )
tail -1 gpurun_out/${T}_stress_rejections.txt
( timeout -k 10 300 python tools/stress_rejections.py 8000 4:6 0.4,1.0 7 > gpurun_out/${T}_stress_default_steps.txt 2>&1; echo "stress default rc $?" )
tail -1 gpurun_out/${T}_stress_default_steps.txt
( timeout -k 10 400 python tools/stress_variants.py 6000 > gpurun_out/${T}_stress_variants.txt 2>&1; echo "variants rc $?" )
tail -2 gpurun_out/${T}_stress_variants.txt
rm -rf gpurun_out/prof_${T}_stats gpurun_out/prof_${T}_fetch gpurun_out/prof_${T}_write
if [ -f hypotremormcmc_amd/lib/libhtm_hip_stamps.so ]; then
  HTM_LIB=hypotremormcmc_amd/lib/libhtm_hip_stamps.so timeout -k 10 200 python tools/stamps.py 8 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_stamps.txt || true
fi
timeout -k 10 200 python tools/bench_regress.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_regress.txt || true
cat gpurun_out/${T}_regress.txt
