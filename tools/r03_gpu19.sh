set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HTM_LIB=hypotremormcmc_amd/lib/libhtm_mfma.so
python - <<'PY'
import ctypes as C, numpy as np
from hypotremormcmc_amd import _lib
lib = _lib.load()
rng = np.random.default_rng(3)
x = rng.standard_normal(64 * 1000) * np.exp(rng.uniform(-20, 20, 64 * 1000))
y = np.empty_like(x)
_lib.check(lib.htm_selftest_math(0, 4, x.ctypes.data_as(_lib.dp), y.ctypes.data_as(_lib.dp), C.c_int(len(x))))
X = x.reshape(-1, 64); Y = y.reshape(-1, 64)
assert (Y == Y[:, :1]).all(), "lanes disagree"
S = ((X[:, 0:16] + X[:, 16:32]) + X[:, 32:48]) + X[:, 48:64]
G = (S[:, 0:4] + S[:, 4:8]) + (S[:, 8:12] + S[:, 12:16])
T = ((G[:, 0] + G[:, 1]) + G[:, 2]) + G[:, 3]
ex = np.array([float(np.sum(r.astype(np.longdouble))) for r in X])
print("bit-equal to the assumed association:", int((T == Y[:, 0]).sum()), "of", len(T))
print("max rel err vs long double / sum|x|:", float(np.max(np.abs(Y[:, 0] - ex) / np.abs(X).sum(1))))
print("selftest rc", lib.htm_selftest(0))
PY
unset HTM_LIB
ROUNDS=2 BENCH_ARGS="--steps 6 --warmup 2 --iters-per-step 16384" bash tools/ab.sh hypotremormcmc_amd/lib/libhtm_hip.so hypotremormcmc_amd/lib/libhtm_mfma.so 2>&1 | tee gpurun_out/r03_ab3.txt
for L in hypotremormcmc_amd/lib/libhtm_hip.so hypotremormcmc_amd/lib/libhtm_mfma.so; do
HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --steps 4 --warmup 1 --iters-per-step 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 fp64', '$L'[-12:], '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --events 10000 --stations 128 --chains 16 --forward-precision fp32 --steps 4 --warmup 1 --iters-per-step 1024 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c5 fp32', '$L'[-12:], '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']))"
HTM_LIB=$L timeout -k 10 120 python bench.py --no-cpu-baseline --events 1000 --stations 64 --steps 4 --warmup 1 --iters-per-step 8192 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('1000x64', '$L'[-12:], '%.0f steps/s %.3f us' % (d['value'], d['config']['us_per_iteration']), {k:v for k,v in d['config'].items() if 'full' in k})"
done
HTM_LIB=hypotremormcmc_amd/lib/libhtm_mfma.so timeout -k 10 900 python -m pytest tests/test_gpu_chains.py tests/test_gpu_forward.py -m gpu -x -q > gpurun_out/r03_o_tests.log 2>&1 || { tail -40 gpurun_out/r03_o_tests.log; exit 1; }
tail -2 gpurun_out/r03_o_tests.log
