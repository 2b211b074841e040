#!/bin/bash
# A/B of the deferred-level path on the headline workload (and the full bench line with it on).
mkdir -p gpurun_out/r03
HGVAE_DEFER=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline > gpurun_out/r03/b2_plain.json 2>/dev/null || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r03/b2_defer.json 2>gpurun_out/r03/b2_defer.err || exit 1
HGVAE_DEFER=0 timeout -k 10 200 python bench.py --steps 20 --warmup 5 --headline-only --no-cpu-baseline > gpurun_out/r03/b2_plain2.json 2>/dev/null || exit 1
python - <<PY
import json
for n in ("b2_plain","b2_defer","b2_plain2"):
    d=json.load(open("gpurun_out/r03/%s.json"%n)); print(n, d["ms_per_step"], d["loss"])
d=json.load(open("gpurun_out/r03/b2_defer.json"))
for w in d["workloads"]: print(w["name"], w["ms_per_step"], w.get("captured_ms_per_step"))
PY
