#!/bin/bash
# The driver's N = 1 bench line and the rocprofv3 --kernel-trace --stats summary of the same command (profiles/rNN/).
# usage: bench_profile.sh TAG      -> gpurun_out/bench_TAG.json, gpurun_out/bench_TAG_kernel_stats.csv
TAG=${1:-r03}
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
python3 bench.py --steps 8 --warmup 2 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -5 gpurun_out/bench_$TAG.err; exit 1; }
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$TAG -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/bench_${TAG}_profiled.json 2> gpurun_out/bench_${TAG}_profiled.err
DB=$(find gpurun_out/prof_$TAG -name "*_results.db" | head -1)
python3 profiles/tools/kernel_stats_from_db.py "$DB" gpurun_out/bench_${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
python3 - <<PY
import json
d = json.load(open("gpurun_out/bench_$TAG.json"))
print({k: d[k] for k in ("value", "ms_per_step", "n_gpus")}, d["gmres_call"]["value"], d["config"]["setup_seconds"])
r = d["roofline"]
print(r["kernel"], r["achieved"], r["frac"], r["traffic"], r["measured_streams"], r.get("reference_ordering"))
print({k: (v["achieved"], v["kernel"]) for k, v in r["single_operators"].items()})
PY
