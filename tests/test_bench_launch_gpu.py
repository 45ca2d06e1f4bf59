"""`python bench.py --gpus N` as the driver runs it (no launcher around it): the parent starts N rank processes before it
has touched the GPU and relays rank 0's one JSON line. Rehearsed on the one GPU of the test box with two ranks sharing
cuda:0 over gloo (`--backend gloo --single-device`); on an 8-GPU node the same command line runs one rank per GPU over RCCL.
Also the two-rank cvig_semantic driver (every rank on its own LOCAL_RANK device; the ranks share one device object between
the module-level transforms and the encoders)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from witw_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_BUDGET = 6144        # bench.LINE_BUDGET: the driver's record keeps the last 8 KB of stdout (round 4's 24 KB line was lost)


def _run_bench(*args, timeout=900):
    """-> the ONE stdout line (asserted to fit the budget), with the full record of --detail-out under '_detail'"""
    import tempfile
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    fd, detail = tempfile.mkstemp(prefix='witw_bench_detail_', suffix='.json')
    os.close(fd)
    try:
        p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--detail-out', detail] + list(args), env=env,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
        assert p.returncode == 0, p.stderr[-3000:]
        every = [ln for ln in p.stdout.splitlines() if ln.strip()]
        lines = [ln for ln in every if ln.startswith('{')]
        # exactly ONE line on stdout (RCCL prints a version banner on stdout when its first communicator is created: bench.py points
        # file descriptor 1 at stderr while that happens)
        assert len(lines) == 1 and every == lines, p.stdout
        assert len(lines[0]) < LINE_BUDGET, len(lines[0])     # ... that the driver's capture holds whole
        out = json.loads(lines[0])
        out['_detail'] = json.load(open(detail))
        assert 'BENCH_DETAIL ' in p.stderr                    # the full record also goes to stderr
    finally:
        os.remove(detail)
    return out


def test_bench_two_ranks_self_launched_infer():
    one = _run_bench('--gpus', '1', '--steps', '1', '--warmup', '1', '--batch', '8', '--no-cpu-baseline', '--no-side-blocks')
    two = _run_bench('--gpus', '2', '--backend', 'gloo', '--single-device', '--steps', '1', '--warmup', '1', '--batch', '8',
                     '--no-cpu-baseline')
    assert two['n_gpus'] == 2 and two['config']['global_batch'] == 16 and two['config']['pairs_per_gpu'] == 8
    assert two['recall']['N'] == 16 and two['unit'] == 'pairs/s' and two['value'] > 0 and two['scaling'] == 'weak'
    assert one['n_gpus'] == 1 and one['recall']['N'] == 8
    assert np.isfinite(two['loss']) and two['roofline']['all_conv_launches_tflops'] > 0
    # the line says what its process group was, from every rank
    c = two['collectives']
    assert c['world'] == 2 and c['ranks_seen'] == 2 and c['backend'] == 'gloo' and c['all_reduce_of_ones'] == 2.0 and len(c['devices']) == 2
    cd = two['_detail']['collectives']
    assert cd['ranks_seen'] == [0, 1] and [d['rank'] for d in cd['devices']] == [0, 1] and cd['devices'][0]['pid'] != cd['devices'][1]['pid']
    assert one['collectives']['world'] == 1 and one['collectives']['ranks_seen'] == 1
    # ... and where the step's device time went, phase by phase (HIP events), from every rank: the collectives' own brackets
    pp = cd['per_phase_ms']
    for name in ('preprocess', 'encoders_forward', 'overhead_all_gather', 'slab_match', 'diagonal_all_gather', 'loss_partial_all_reduce'):
        assert pp['rank0'][name] > 0 and pp['max_over_ranks'][name] >= pp['rank0'][name], (name, pp)
        assert c['per_phase_ms_max_over_ranks'][name] == pp['max_over_ranks'][name]
    assert sum(pp['rank0'].values()) <= 1.05 * pp['ms_per_step'] + 0.5          # phases are disjoint parts of the step
    assert set(one['_detail']['collectives']['per_phase_ms']['rank0']) == set(pp['rank0'])
    # ... and the comms roofline: the step's three payloads timed standalone against the xGMI bound (plumbing rehearsed over gloo)
    mb = c['microbench']
    assert mb['all_gather_overhead_embeddings']['bytes'] == 8 * 4096 * 4 and mb['reduce_scatter_overhead_grads']['bytes'] == 2 * 8 * 4096 * 4
    assert mb['all_reduce_weight_grads_one_encoder']['bytes'] == 7236432 * 4 and mb['all_reduce_weight_grads_one_encoder']['per_step'] == 2
    for name in ('all_gather_overhead_embeddings', 'reduce_scatter_overhead_grads', 'all_reduce_weight_grads_one_encoder'):
        m = mb[name]
        # (over gloo on a loaded host a 128 KB all-gather can take tens of ms: the rounded figures may be tiny, never negative)
        assert m['us'] > 0 and m['algbw_GBps'] >= 0 and 0 <= m['frac_of_direct_bound'] < 1.0 and m['xgmi_ring_bound_us'] >= m['xgmi_direct_bound_us']
    assert 'microbench' not in one['collectives']
    # ... and BASELINE configs[2] as what it is, a DDP TRAINING config (round 6): behind the inference headline the N > 1 line runs the
    # training step in the same process group -- all-gather, reduce-scatter and the two bucket all-reduces inside a real step
    ts = two['train_step']
    assert ts['value'] > 0 and ts['ms_per_step'] > 0 and ts['steps'] == 1 and np.isfinite(ts['loss']) and 'configs[2]' in ts['baseline_config']
    for name in ('encoders_forward', 'overhead_all_gather', 'slab_match', 'backward_incl_its_collectives', 'overhead_grad_reduce_scatter',
                 'grad_bucket0_all_reduce_issue_to_joined', 'grad_bucket1_all_reduce_issue_to_joined', 'reducer_wait_stall', 'adam'):
        assert ts['per_phase_ms_max_over_ranks'][name] > 0, (name, ts)
    assert abs(ts['overlap_hidden_ms'] - (ts['bucket_inflight_ms'] - ts['reducer_wait_stall_ms'])) < 1e-3
    assert abs(mb['overlap_hidden_ms'] - ts['overlap_hidden_ms']) < 1e-3          # the microbench block quotes the TRAINING step's overlap
    td = two['_detail']['train_step_detail']
    assert len(td['per_phase_ms']['every_rank']) == 2 and 'training step' in td['metric']
    assert 'train_step' not in one
    # ... and which register-allocation guards were active (none may have tripped on the validated toolchain)
    for line in (one, two):
        g = line['guards']
        assert g['tripped'] == [] and g['bf16_16x16x32_kernel_on'] is True and g['bf16_weight_resident_kernel_on'] is True, g


def test_bench_one_rank_under_the_launcher_equals_the_plain_run():
    """The driver's N > 1 entry (python -m torch.distributed.run ... bench.py --gpus N) with N = 1: one rank, no process group,
    the same step and the same numbers as `python bench.py --gpus 1`."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    args = ['--gpus', '1', '--steps', '2', '--warmup', '1', '--batch', '16', '--no-cpu-baseline', '--no-side-blocks']
    p = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py')] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout
    launched = json.loads(lines[0])
    plain = _run_bench(*args)
    assert launched['n_gpus'] == plain['n_gpus'] == 1 and launched['collectives']['world'] == 1
    assert launched['loss'] == plain['loss'] and launched['recall'] == plain['recall']           # the same step, bit for bit
    assert 0.5 < launched['value'] / plain['value'] < 2.0


def test_bench_two_ranks_self_launched_train_and_retrieval():
    tr = _run_bench('--gpus', '2', '--backend', 'gloo', '--single-device', '--mode', 'train', '--steps', '1', '--warmup', '1',
                    '--batch', '8')
    assert tr['n_gpus'] == 2 and 'training step' in tr['metric'] and np.isfinite(tr['loss'])
    pp = tr['_detail']['collectives']['per_phase_ms']
    mb = tr['collectives']['microbench']
    stall = pp['max_over_ranks']['reducer_wait_stall']
    assert abs(mb['overlap_hidden_ms'] - (pp['rank0']['grad_bucket0_all_reduce_issue_to_joined'] + pp['rank0']['grad_bucket1_all_reduce_issue_to_joined']
                                          - pp['rank0']['reducer_wait_stall'])) < 1e-3 and stall > 0
    for name in ('encoders_forward', 'overhead_all_gather', 'slab_match', 'diagonal_all_gather', 'loss_partial_all_reduce',
                 'backward_incl_its_collectives', 'row_sigmoid_all_reduce', 'slab_match_backward', 'overhead_grad_reduce_scatter',
                 'grad_bucket0_all_reduce_issue_to_joined', 'grad_bucket1_all_reduce_issue_to_joined', 'reducer_wait_stall', 'adam'):
        assert pp['rank0'][name] > 0 and pp['max_over_ranks'][name] >= pp['rank0'][name], (name, pp)
    assert tr['guards']['tripped'] == []
    rt = _run_bench('--gpus', '2', '--backend', 'gloo', '--single-device', '--mode', 'retrieval', '--gallery', '2000',
                    '--queries', '100', '--steps', '1', '--warmup', '1')
    assert rt['n_gpus'] == 2 and rt['recall']['N'] == 4000
    one = _run_bench('--gpus', '1', '--mode', 'retrieval', '--gallery', '2000', '--queries', '100', '--steps', '1', '--warmup', '1')
    # the planted match of every query is in rank 0's shard and is found from both layouts
    assert rt['recall']['topk_first_is_true_pct'] == one['recall']['topk_first_is_true_pct'] >= 90.0
    rd = _run_bench('--gpus', '2', '--backend', 'gloo', '--single-device', '--mode', 'retrieval', '--match', 'dft', '--gallery',
                    '2000', '--queries', '100', '--steps', '1', '--warmup', '1')
    assert rd['recall'] == rt['recall']


def test_bench_wrong_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE='3', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2'], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and 'WORLD_SIZE' in p.stderr


def _semantic_worker(rank, world, port, root, out_q):
    import torch.distributed as dist
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    os.environ['LOCAL_RANK'] = '0'                      # both ranks share the one GPU of the test box
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from witw_amd import cvig_fov, cvig_semantic
        assert cvig_semantic.device == cvig_fov.device          # ONE device per process for transforms, encoders, collectives
        os.chdir(root)
        csv = os.path.join(root, 'scenes.csv')
        best = cvig_semantic.train(dataset='witw', fov=70, val_quantity=4, batch_size=2, num_workers=0, num_epochs=1, csv_path=csv,
                                   seed=3)
        cvig_semantic.Globals.test_random_orientation = False
        table = cvig_semantic.test(dataset='witw', fov=70, batch_size=4, num_workers=0, csv_path=csv)
        out_q.put((rank, best, table))
    finally:
        dist.destroy_process_group()


def test_semantic_driver_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    from tests.test_drivers2_gpu import _write_semantic_dataset
    from tests.test_parallel_gpu import _free_port
    root = str(tmp_path)
    _write_semantic_dataset(root, 12)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_semantic_worker, args=(r, 2, port, root, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] is not None and np.isfinite(res[0][1]) and res[0][1] == res[1][1]
    assert res[0][2] == res[1][2]
    sd = torch.load(os.path.join(root, 'weights', 'fov_70_surface_best.pth'))
    assert sd['model.features.0.weight'].shape == (64, 5, 3, 3)


def test_bench_collectives_block_through_rccl_in_a_world_of_one():
    """`--pg-of-one`: the bench's collectives block and microbench inside a real process group on backend nccl = RCCL (one rank: a second
    one cannot share the test box's GPU) -- the RCCL-only branches of the N > 1 line (device_id init, RCCL version, all_gather_into_tensor,
    the counted all-reduce) execute on hardware before the first multi-GPU run does."""
    one = _run_bench('--gpus', '1', '--pg-of-one', '--steps', '1', '--warmup', '1', '--batch', '8', '--no-cpu-baseline', '--no-side-blocks')
    c = one['collectives']
    assert c['backend'] == 'nccl (RCCL)' and c['world'] == 1 and c['rccl_version'] and c['all_reduce_of_ones'] == 1.0 and c['ranks_seen'] == 1
    mb = c['microbench']
    for name in ('all_gather_overhead_embeddings', 'all_reduce_weight_grads_one_encoder'):
        assert mb[name]['us'] > 0 and mb[name]['bytes'] > 0
    assert one['n_gpus'] == 1 and one['recall']['N'] == 8 and np.isfinite(one['loss'])
    # the training step of the N > 1 line ran through RCCL too (one rank: every collective is the identity)
    ts = one['train_step']
    assert ts['value'] > 0 and ts['per_phase_ms_max_over_ranks']['adam'] > 0 and 'overlap_hidden_ms' in ts


def test_bench_two_ranks_with_the_cpu_baseline_behind_the_group():
    """N > 1 with the cpu_baseline on: rank 0 times the CPU port AFTER every rank has left the process group (no rank sits in a
    collective meanwhile); the line carries the three legs of BASELINE.md section 4."""
    two = _run_bench('--gpus', '2', '--backend', 'gloo', '--single-device', '--steps', '1', '--warmup', '1', '--batch', '4',
                     '--cpu-pairs', '2', '--no-microbench')
    c = two['cpu_baseline']
    assert c['kind'] == 'port' and c['passes'] == 3 and len(c['pass_pairs_per_s']) == 3 and c['value'] > 0
    assert c['train_pairs_per_s'] > 0 and c['rank_ms_per_query'] > 0
    assert two['n_gpus'] == 2 and two['collectives']['ranks_seen'] == 2 and two['train_step']['value'] > 0
