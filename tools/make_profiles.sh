#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 summaries of the default bench for profiles/.
#   tools/make_profiles.sh <tag>        e.g. r01_c  -> gpurun_out/<tag>_{kernel_stats.csv,pmc_summary.txt,bench.json}
# Counter passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md, HBM section).
set -e
TAG=${1:-r02_x}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf "$OUT/prof_${TAG}_stats" "$OUT/prof_${TAG}_fetch" "$OUT/prof_${TAG}_write"
rocprofv3 --kernel-trace --stats -d "$OUT/prof_${TAG}_stats" -o st --output-format csv -- python3 "$ROOT/bench.py" --no-cpu-baseline \
    > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/prof_${TAG}_stats.log"
echo "stats pass done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/prof_${TAG}_fetch" -o pf --output-format csv -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --iters-per-step 1024 --no-cpu-baseline \
    > /dev/null 2> "$OUT/prof_${TAG}_fetch.log"
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/prof_${TAG}_write" -o pw --output-format csv -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --iters-per-step 1024 --no-cpu-baseline \
    > /dev/null 2> "$OUT/prof_${TAG}_write.log"
echo "write pass done"
python3 "$ROOT/bench.py" > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
st = glob.glob(os.path.join(out, f"prof_{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)
if st:
    open(os.path.join(out, f"{tag}_kernel_stats.csv"), "w").write(open(st[0]).read())
lines = [f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 4 --warmup 1 --iters-per-step 1024 --no-cpu-baseline",
         "# per dispatch, KB as rocprofv3 reports them (raw; gfx950: FETCH_SIZE counts 64 B per 128-B request of wide coalesced reads)",
         "%-34s %-11s %7s %14s %12s %14s" % ("kernel", "counter", "count", "mean", "min", "max")]
for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    agg = collections.defaultdict(list)
    for fn in glob.glob(os.path.join(out, f"prof_{tag}_{sub}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if row.get("Counter_Name") == cname:
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[k].append(float(row["Counter_Value"]))
    for k in sorted(agg):
        v = agg[k]
        lines.append("%-34s %-11s %7d %14.3f %12.3f %14.3f" % (k[:34], cname, len(v), sum(v) / len(v), min(v), max(v)))
open(os.path.join(out, f"{tag}_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
# k_mcmc launch by launch from the kernel trace, beside what bench.py measured with HIP events in the same run
tr = glob.glob(os.path.join(out, f"prof_{tag}_stats", "**", "*kernel_trace.csv"), recursive=True)
try:
    import json as _j
    b = _j.load(open(os.path.join(out, f"{tag}_bench_under_rocprof.json")))["roofline"]
    dur = [(int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e6 for x in csv.DictReader(open(tr[0])) if "k_mcmc" in x["Kernel_Name"]]
    open(os.path.join(out, f"{tag}_kernel_launches.txt"), "w").write(
        "# k_mcmc launches of `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu-baseline` (ms, in launch order)\n"
        "# the first launches are the untimed warm-up, the last %d the timed region\n" % b["launches"] +
        " ".join("%.3f" % x for x in dur) + "\n" +
        "timed launches, rocprofv3 trace: %.3f ms   bench.py (HIP events on the kernels' stream, same run): %.3f ms\n"
        % (sum(dur[-b["launches"]:]), b["avg_launch_us"] * b["launches"] / 1e3))
except Exception as e:
    print("kernel_launches:", e)
# the k_mcmc rows per iteration: what bench.py reports as roofline.traffic
import json
tot = {}
nl = 0
for sub, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    v = []
    for fn in glob.glob(os.path.join(out, f"prof_{tag}_{sub}", "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(fn)):
            if row.get("Counter_Name") == cname and "k_mcmc" in row["Kernel_Name"]:
                v.append(float(row["Counter_Value"]))
    tot[sub] = sum(v)
    nl = len(v)
iters = 5 * 1024
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace), python3 bench.py --steps 4 --warmup 1 --iters-per-step 1024 --no-cpu-baseline",
           "kernel": "htm::k_mcmc (the launches of the run)", "iterations": iters, "launches": nl,
           "fetch_kb_raw_total": tot["fetch"], "write_kb_total": tot["write"],
           "fetch_bytes_per_iteration_raw": tot["fetch"] * 1024 / iters, "write_bytes_per_iteration": tot["write"] * 1024 / iters,
           "note": "raw counter values; gfx950 FETCH_SIZE counts 64 B per 128-B request for wide coalesced reads (x2 at most); 8-byte-per-lane access widths are uncalibrated"},
          open(os.path.join(out, f"{tag}_traffic.json"), "w"), indent=1)
print("\n".join(lines))
PY
cat "$OUT/${TAG}_kernel_stats.csv"
cat "$OUT/${TAG}_bench.json"
