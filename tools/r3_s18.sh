#!/bin/bash
# one chunk of up to 2^31 hits instead of 2^30: north-star size, same HSP list?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s18
for cap in 1073741824 2147483648; do
  LZGPU_HIT_CAPACITY=$cap timeout 900 python bench.py --north-star --steps 1 --warmup 0 --no-cpu-baseline --no-cli --no-gapped > gpurun_out/s18/c$cap.json 2> gpurun_out/s18/c$cap.err
  C=$cap python - <<'PY'
import json, os
c = os.environ["C"]
try:
    d = json.loads(open(f"gpurun_out/s18/c{c}.json").read().strip().splitlines()[-1])
    print("cap", c, "ms", round(d["ms_per_step"], 1), "hsps", d["hsps"], d["parity"].get("hsp_sha", "")[:16], {k: round(v) for k, v in d["kernel_ms_per_step"].items() if v > 50})
except Exception as e:
    print("cap", c, "failed", e, open(f"gpurun_out/s18/c{c}.err").read()[-600:])
PY
done
