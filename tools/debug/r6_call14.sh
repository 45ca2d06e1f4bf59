cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call14; mkdir -p $O
timeout -k 10 300 python3 -m pytest tests/test_jpeg_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest.log | cut -c1-300
for i in 1 2; do
WITW_SIDES_ONLY=batch_sweep timeout -k 10 300 python3 bench.py --mode sides --steps 5 --no-cpu-baseline --detail-out $O/s$i.json > /dev/null 2> $O/s$i.err
python3 -c "
import json; d=json.load(open('$O/s$i.json'))
print('only sweep:', [(p['precision'], p['pairs_per_gpu'], p['value'], p.get('graph_replay',{}).get('value')) for p in d['batch_sweep']['points'] if p['precision']=='bf16'])"
done
WITW_SIDES_ONLY=config5_retrieval,config5_retrieval_direct,hbm_kernels,batch_sweep timeout -k 10 300 python3 bench.py --mode sides --steps 5 --no-cpu-baseline --detail-out $O/s3.json > /dev/null 2> $O/s3.err
python3 -c "
import json; d=json.load(open('$O/s3.json'))
print('retr+hbm+sweep:', [(p['precision'], p['pairs_per_gpu'], p['value'], p.get('graph_replay',{}).get('value')) for p in d['batch_sweep']['points'] if p['precision']=='bf16'])"
WITW_SIDES_ONLY=train_step_fp32,config4_semantic_bf16,config4_semantic_bf16_train,train_step_bf16,config1_baseline,batch_sweep timeout -k 10 300 python3 bench.py --mode sides --steps 5 --no-cpu-baseline --detail-out $O/s4.json > /dev/null 2> $O/s4.err
python3 -c "
import json; d=json.load(open('$O/s4.json'))
print('trains+sweep:', [(p['precision'], p['pairs_per_gpu'], p['value'], p.get('graph_replay',{}).get('value')) for p in d['batch_sweep']['points'] if p['precision']=='bf16'])"
