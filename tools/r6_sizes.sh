cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for TQ in "50000000 50000000" "4000000 50000000" "1000000 50000000" "50000000 4000000" "12000000 50000000"; do set -- $TQ
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-cli --no-north-star --no-content --no-gapped --no-pmc --tlen $1 --qlen $2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); h=d['counters_per_step']['raw_hits']; k=d['kernel_ms_per_step']
print('$1 x $2 hits %.3g' % h, 'scan %.2f ms = %.2f ps/hit' % (k['k_scan_hits'], k['k_scan_hits']*1e9/h), 'bp/hit %.1f' % (d['counters_per_step']['bp_extended']/h), {kk: round(v*1e9/h,2) for kk,v in k.items() if v*1e9/h>0.5})"
done
