"""The one stdout line of bench.py stays within the driver's capture (round 4's 24 KB line left BENCH_r04.parsed
null). CPU only: full records of the shapes the modes produce -- round 4's committed 24 KB default record, an 8-rank training
record with every collectives field filled, the sweep -- are compacted and must fit LINE_BUDGET with the headline keys intact."""
import io
import json
import os
import sys
import types
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

HEADLINE = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype',
            'data', 'config', 'recall', 'loss', 'roofline', 'collectives', 'guards', 'cpu_baseline')


def _emit(full, tmp_path, **kw):
    a = types.SimpleNamespace(detail_out=str(tmp_path / 'detail.json'), mode=kw.get('mode', 'infer'), single_device=kw.get('single_device', False))
    so, se = io.StringIO(), io.StringIO()
    with redirect_stdout(so), redirect_stderr(se):
        rc = bench.emit(full, a)
    lines = [ln for ln in so.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1
    return lines[0], json.loads(lines[0]), json.load(open(a.detail_out)), rc, se.getvalue()


def _r04_full():
    rec = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench.json')))
    full = {k: rec[k] for k in HEADLINE}
    full['_side_full'] = {k: v for k, v in rec.items() if k not in HEADLINE}
    return rec, full


def test_default_record_of_round4_fits_the_line_budget(tmp_path):
    rec, full = _r04_full()
    assert len(json.dumps(rec)) > 20000                         # the record that broke the driver's parse
    text, line, detail, rc, err = _emit(full, tmp_path)
    assert rc is None and len(text) < bench.LINE_BUDGET == 6144
    assert 'shed_to_fit' not in line                            # nothing had to be dropped to get there
    for k in HEADLINE:
        assert k in line, k
    assert line['value'] == rec['value'] and line['ms_per_step'] == rec['ms_per_step'] and line['config']['pairs_per_gpu'] == 128
    r = line['roofline']
    assert r['bound'] == 'mfma' and r['frac'] == rec['roofline']['frac'] and r['peak'] == 157.3 and 'traffic' in r and r['kernel'].startswith('conv3x3_nhwc_f32')
    c = line['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == rec['cpu_baseline']['cores'] and c['value'] == rec['cpu_baseline']['value'] and c['sample']
    # one compact entry per side config: value, ms_per_step, dtype, dominant kernel + fraction
    for name in ('train_step_fp32', 'config4_semantic_bf16', 'config1_baseline', 'config5_retrieval', 'config5_retrieval_direct'):
        sblk = line['side'][name]
        assert sblk['value'] == rec[name]['value'] and sblk['ms_per_step'] == rec[name]['ms_per_step'] and sblk['frac'] == rec[name]['roofline']['frac'], name
    assert len(line['side']['batch_sweep']['points']) == 8
    # ... and the full blocks are in the detail record (file + stderr), untouched
    assert detail['batch_sweep'] == rec['batch_sweep'] and detail['e2e_data_path_bf16'] == rec['e2e_data_path_bf16']
    assert detail['roofline'] == rec['roofline'] and 'BENCH_DETAIL ' in err
    assert line['detail'].endswith('detail.json')


def _world8_train():
    phases = {n: 1.2345 for n in ('preprocess', 'encoders_forward', 'overhead_all_gather', 'slab_match', 'diagonal_all_gather',
                                  'loss_partial_all_reduce', 'backward_incl_its_collectives', 'row_sigmoid_all_reduce', 'slab_match_backward',
                                  'overhead_grad_reduce_scatter', 'grad_bucket0_all_reduce_issue_to_joined',
                                  'grad_bucket1_all_reduce_issue_to_joined', 'reducer_wait_stall', 'adam')}
    mb = {n: {'bytes': 28945728, 'us': 123.4, 'algbw_GBps': 234.56, 'xgmi_direct_bound_us': 47.3, 'frac_of_direct_bound': 0.3833,
              'xgmi_ring_bound_us': 331.1} for n in ('all_gather_overhead_embeddings', 'reduce_scatter_overhead_grads',
                                                     'all_reduce_weight_grads_one_encoder')}
    mb.update(overlap_hidden_ms=0.4321, link_GBps=153.0, how='x' * 120)
    return {'metric': 'image-pairs/sec (training step)', 'value': 9876.54, 'unit': 'pairs/s', 'n_gpus': 8, 'steps': 20, 'warmup': 5,
            'ms_per_step': 103.678, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'w' * 400, 'pairs_per_gpu': 128, 'global_batch': 1024, 'ground_raw': '3x224x224', 'overhead_raw': '3x512x512',
                       'parallelism': 'dp8 (overhead-embedding all-gather, global-batch loss from column slabs)'},
            'recall': {'top1_pct': 0.1, 'top5_pct': 0.5, 'N': 1024}, 'loss': 0.693, 'recall_note': 'n' * 200,
            'roofline': {'bound': 'mfma', 'kernel': 'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>', 'achieved': 147.0, 'peak': 157.3, 'unit': 'TFLOP/s',
                         'frac': 0.9345, 'traffic': 998000000, 'traffic_source': {'file': 'profiles/traffic.json', 'how': 'h' * 200, 'stale': False,
                                                                                 'kernel_sources_sha16': {'a': 'b' * 16}},
                         'launches': 240, 'avg_launch_ms': 3.14, 'avg_launch_gflop': 463.86, 'all_conv_launches_tflops': 141.0, 'whole_step_frac': 0.88},
            'collectives': {'backend': 'nccl (RCCL)', 'world': 8, 'rccl_version': '2.22.3', 'ranks_seen': list(range(8)), 'all_reduce_of_ones': 8.0,
                            'devices': [{'rank': r, 'local_device': r, 'pid': 1000 + r, 'device_name': 'AMD Instinct MI355X',
                                         'pci_bus_id': '0000:%02x:00.0' % (5 + 16 * r)} for r in range(8)], 'distinct_devices': 8,
                            'per_phase_ms': {'rank0': phases, 'max_over_ranks': phases, 'ms_per_step': 103.678, 'how': 'h' * 400},
                            'per_step': 'p' * 250, 'microbench': mb},
            'guards': {'tripped': [], 'forced_by_env': [], 'detail': {}, 'bf16_16x16x32_kernel_on': True, 'bf16_weight_resident_kernel_on': True,
                       'what': 'w' * 300}}


def test_eight_rank_training_record_fits_and_keeps_the_comms_roofline(tmp_path):
    full = _world8_train()
    text, line, detail, rc, err = _emit(full, tmp_path, mode='train')
    assert rc is None and len(text) < bench.LINE_BUDGET and 'shed_to_fit' not in line
    c = line['collectives']
    assert c['world'] == 8 and c['ranks_seen'] == 8 and c['distinct_devices'] == 8 and len(c['devices']) == 8 and c['rccl_version']
    assert c['microbench']['all_reduce_weight_grads_one_encoder']['frac_of_direct_bound'] == 0.3833 and c['microbench']['overlap_hidden_ms'] == 0.4321
    assert c['per_phase_ms_max_over_ranks']['reducer_wait_stall'] == 1.2345
    assert detail['collectives']['per_phase_ms']['rank0'] == full['collectives']['per_phase_ms']['rank0']


@pytest.mark.parametrize('field,value', [('all_reduce_of_ones', 7.0), ('distinct_devices', 4), ('ranks_seen', [0, 1, 2])])
def test_a_line_whose_process_group_is_not_n_ranks_on_n_devices_is_refused(tmp_path, field, value):
    full = _world8_train()
    full['collectives'][field] = value
    text, line, detail, rc, err = _emit(full, tmp_path, mode='train')
    assert rc == 3 and json.loads(text)['n_gpus'] == 8          # printed, then refused through the exit status
    if field == 'distinct_devices':                             # the one-GPU rehearsal (--single-device) is exempt from the device count only
        assert _emit(full, tmp_path, mode='train', single_device=True)[3] is None


def test_an_oversized_record_sheds_optional_parts_not_the_headline(tmp_path):
    rec, full = _r04_full()
    full['_side_full'].update({'extra_%d' % i: {'value': 1.0, 'unit': 'pairs/s', 'ms_per_step': 1.0, 'dtype': 'f32',
                                               'roofline': {'kernel': 'k' * 80, 'frac': 0.5, 'bound': 'mfma'}} for i in range(40)})
    text, line, detail, rc, err = _emit(full, tmp_path)
    assert len(text) < bench.LINE_BUDGET and line['shed_to_fit']
    for k in ('value', 'ms_per_step', 'roofline', 'cpu_baseline', 'config'):
        assert k in line


def test_sweep_record_is_one_row_per_point(tmp_path):
    sweep = json.load(open(os.path.join(ROOT, 'profiles', 'r04_bench_sweep.json')))
    text, line, detail, rc, err = _emit(sweep, tmp_path, mode='sweep')
    assert len(text) < bench.LINE_BUDGET and len(line['points']) == 8 and len(line['points'][0]) == len(line['points_columns'])
    assert 'kernels' in detail['points'][0]
