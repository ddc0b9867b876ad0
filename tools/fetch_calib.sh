#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of gfx950 against known byte counts, per access pattern (tools/ub/fetch_calib.hip; VERDICT r4 #3b).
# usage (GPU box): bash tools/fetch_calib.sh <outdir>   -> <outdir>/fetch_size_calibration.txt (+ .json: counted / known per kernel)
set -u
O=${1:-gpurun_out/calib}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/ub/fetch_calib.hip 2> $O/build.err || { cat $O/build.err; exit 1; }
/tmp/fetch_calib > $O/known.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -- /tmp/fetch_calib > /dev/null 2> $O/err_f.txt
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -- /tmp/fetch_calib > /dev/null 2> $O/err_w.txt
python tools/pmc_fetch_write.py "$O/f/**/*counter_collection.csv" "$O/w/**/*counter_collection.csv" > $O/pmc.csv
python - "$O" <<'PY'
import sys, json
O = sys.argv[1]
known = {l.split()[1]: int(l.split()[2]) for l in open(O + "/known.txt") if l.startswith("known_bytes")}
rows = [l.strip().split(",") for l in open(O + "/pmc.csv").read().split("\n")[1:] if l.strip()]
out = {}
lines = ["# tools/fetch_calib.sh: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace) of tools/ub/fetch_calib.hip on gfx950;",
         "# counter units KiB; 'known' = bytes that move by construction per launch; factor = known / counted (multiply a counted value by it)",
         "kernel,known_bytes,FETCH_SIZE_bytes,WRITE_SIZE_bytes,avg_ms,factor_fetch,factor_write,GB_per_s_known"]
for k, n, f, w, ms in rows:
    if k not in known: continue
    fb, wb = float(f) * 1024, float(w) * 1024
    ff = known[k] / fb if fb > 0 and "store" not in k else None
    fw = known[k] / wb if wb > 0 and "store" in k else None
    out[k] = {"known_bytes": known[k], "fetch_bytes": fb, "write_bytes": wb, "avg_ms": float(ms), "factor_fetch": ff, "factor_write": fw}
    lines.append("%s,%d,%.0f,%.0f,%s,%s,%s,%.0f" % (k, known[k], fb, wb, ms, "%.3f" % ff if ff else "", "%.3f" % fw if fw else "", known[k] / (float(ms) * 1e-3) / 1e9))
open(O + "/fetch_size_calibration.txt", "w").write("\n".join(lines) + "\n")
json.dump(out, open(O + "/fetch_size_calibration.json", "w"), indent=1)
print("\n".join(lines))
PY
rm -rf $O/f $O/w
