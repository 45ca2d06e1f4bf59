cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call16; mkdir -p $O
for blk in train_step_fp32 config4_semantic_bf16 config4_semantic_bf16_train train_step_bf16 config1_baseline; do
WITW_SIDES_ONLY=$blk,batch_sweep timeout -k 10 300 python3 bench.py --mode sides --steps 5 --no-cpu-baseline --detail-out $O/s_$blk.json > /dev/null 2> $O/s_$blk.err
python3 -c "
import json; d=json.load(open('$O/s_$blk.json'))
print('$blk + sweep:', [(p['pairs_per_gpu'], p['value']) for p in d['batch_sweep']['points'] if p['precision']=='bf16'])"
done
