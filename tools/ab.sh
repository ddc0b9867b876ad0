#!/bin/bash
# A/B of environment settings on the seed bench: bash tools/ab.sh "VAR=1 OTHER=2" "VAR=0" ...   (one bench line each)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  env $cfg python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-gapped > /tmp/o.json 2>/tmp/e.txt
  CFG="$cfg" python - <<'PY'
import json, os
try:
    d = json.load(open('/tmp/o.json'))
    print(os.environ["CFG"], "|", round(d["ms_per_step"], 1), d["parity"]["hsp_sha_ok"], {k: round(v, 1) for k, v in d["kernel_ms_per_step"].items() if v > 4})
except Exception as e:
    print(os.environ["CFG"], "| failed", e, open('/tmp/e.txt').read()[-400:])
PY
done
