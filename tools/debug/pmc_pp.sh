#!/bin/bash
# PMC passes over the two weight-resident bf16 kernels (run through gpurun): wave-cycle breakdown, LDS conflicts, matrix-pipe busy
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES"; do
  d=$R/gpurun_out/pmc_pp_$(echo $set | cut -d' ' -f1)
  rm -rf $d
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $d -- python3 $R/tools/bench_f2_wres.py > $d.log 2>&1
  f=$(find $d -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'][:60]
    if 'first2' not in k and 'wres' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, d in acc.items():
    print(k)
    for c, v in d.items(): print('   %-28s %.5g per launch' % (c, v / n[(k, c)]))
PY
done
