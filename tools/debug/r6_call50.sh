cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call50; mkdir -p $O
for t in 2048 1280 2560 3840 5120; do
WITW_TOPK_BLOCKS=$t rocprofv3 --kernel-trace --stats -o p --output-format csv -d $O/prof$t -- python3 bench.py --mode retrieval --match dft --steps 2 --warmup 1 --detail-out $O/d.json > $O/b$t.json 2> $O/prof$t.log
python3 - <<PY
import csv, json
d=json.loads(open('$O/b$t.json').read().strip().splitlines()[-1])
rows=list(csv.DictReader(open('$O/prof$t/p_kernel_stats.csv')))
tk=[r for r in rows if 'topk_kernel' in r['Name']][0]; mg=[r for r in rows if 'topk_merge' in r['Name']][0]
print('target blocks $t: pass %.2f ms, topk_kernel total %.2f ms (max %.3f), merge total %.2f ms' % (d['ms_per_step'], float(tk['TotalDurationNs'])/1e6, float(tk['MaxNs'])/1e6, float(mg['TotalDurationNs'])/1e6))
PY
done
