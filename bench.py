#!/usr/bin/env python
"""Headline benchmark: image-pairs/sec of the cvig_fov embedding + similarity hot path.

    python bench.py [--gpus N --steps K --warmup W]        (N>1 without a launcher: starts its own N ranks; under
                                                            torch.distributed.run it is one of them)

One step = one pass of the hot path over one synthetic batch that is already resident in HBM:
  raw ground 3x224x224 + raw overhead 3x512x512 (uint8-valued fp32)
    -> Resize + ImageNormalization + PolarTransform            (A1-A3, model/cvig_fov.py:100-209)
    -> surface / overhead FOV_DSM encoders, eval mode, fp32     (A4-A6, :248-294)
    -> [N>1] all-gather of the overhead embeddings over RCCL    (global negatives)
    -> fused correlation/argmax/crop/chord distance of ALL overheads vs the local surfaces, global-batch
       soft-margin triplet loss (diagonal all-gather + scalar all-reduce), rank counts for the local
       queries                                                  (A7-A10, A12, :297-382, :543-552)
Per-GPU batch is fixed (weak scaling); value = global pairs / max-over-ranks step time.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, dense fp32 MFMA
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same table, dense bf16 MFMA (no sparsity)
DOMINANT = (128, 1, False, 8)  # conv3x3_nhwc_f32_kernel<128,1,false,8,0,9> (TN, SH, POOL, NW, GEO, TAPS): layers 5,10,12,17,19,21 at B=128


def make_inputs(cvig_fov, ops, synth, batch, fov, seed, device, channels=3):
    """Synthetic raw pairs with planted matches: the overhead is uint8 noise; its ground
    image is the polar view of that overhead, rolled by a seeded shift, plus noise, at raw size.
    channels=5 (cvig_semantic): bands 3,4 are the semantic channels, uniform in [0,1] (they are not divided
    by 255, model/cvig_semantic.py:172-176)."""
    ov_raw = torch.from_numpy(synth.images_u8(seed, 1, (batch, channels, 512, 512))).to(device)
    if channels > 3:
        ov_raw[:, 3:] /= 255.0
    ws = int(fov / 360 * 512)
    small = ops.resize_bilinear(ov_raw, (256, 256))
    polar = ops.polar_transform(small)                                   # [B,3,128,512], 0..255 scale
    g = np.random.Generator(np.random.Philox(key=[seed, 77]))
    shifts = g.integers(0, 512, size=batch)
    rolled = torch.stack([torch.roll(polar[i], -int(shifts[i]), dims=2)[:, :, :ws] for i in range(batch)])
    noise = torch.from_numpy(synth.images_u8(seed, 2, (batch, channels, 128, ws))).to(device)
    if channels > 3:
        noise[:, 3:] /= 255.0
    ground = (0.7 * rolled + 0.3 * noise).contiguous()
    ground_raw = ops.resize_bilinear(ground, (224, 224))
    ground_raw[:, :3] = ground_raw[:, :3].round().clamp(0, 255)
    return ground_raw.contiguous(), ov_raw


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N fresh rank processes (one per GPU) through
    torch.distributed.run and relay rank 0's JSON line. Runs BEFORE this process has made any GPU call, and the ranks are
    children, not an exec of this process. -> exit status (0 only if every rank exited 0 and one JSON line came back)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')     # dmabuf IPC: RCCL's intra-node transport needs it on this driver
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith('{') and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    return p.returncode if p.returncode != 0 else (0 if line is not None else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=128, help='pairs per GPU (BASELINE.json configs[1]: bs=128)')
    ap.add_argument('--fov', type=int, default=360)
    ap.add_argument('--mode', choices=['infer', 'train', 'retrieval'], default='infer',
                    help='infer (headline): embedding + similarity; train: the full step of model/cvig_fov.py:444-461; '
                         'retrieval: BASELINE config 5, --gallery rows per GPU x --queries, ranks + top-k')
    ap.add_argument('--gallery', type=int, default=125000, help='retrieval: gallery rows PER GPU (1M / 8)')
    ap.add_argument('--queries', type=int, default=10000, help='retrieval: ground queries (replicated)')
    ap.add_argument('--topk', type=int, default=10)
    ap.add_argument('--match', choices=['direct', 'dft'], default='direct',
                    help="retrieval: 'dft' = orientation search through the 64-point row spectra (21k instead of 524k FLOP per "
                         "pair, same fp32 MFMA; scores equal to fp32 rounding)")
    ap.add_argument('--precision', choices=['fp32', 'bf16', 'fp16x3'], default='fp32',
                    help='fp32 (headline, BASELINE configs[1]) or the bf16 MFMA inference path (configs[3] arithmetic)')
    ap.add_argument('--model', choices=['fov', 'semantic'], default='fov',
                    help='fov = cvig_fov (3-channel, BASELINE configs[1]); semantic = cvig_semantic (5-channel first conv, '
                         'configs[3]: run it with --precision bf16)')
    ap.add_argument('--graph', action='store_true',
                    help='inference, one GPU: capture the whole step in a hipGraph (parallel.CapturedStep) and time replays; '
                         'pays off where the step is launch-bound (small --batch, bf16)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-side-blocks', action='store_true', help='only the headline measurement (no blocks for the other BASELINE configs)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL); gloo only for 1-GPU self-tests')
    ap.add_argument('--single-device', action='store_true', help='self-test: put every rank on cuda:0')
    ap.add_argument('--cpu-pairs', type=int, default=8)
    a = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if 'WORLD_SIZE' not in os.environ and a.gpus > 1:
        # plain `python bench.py --gpus N` (the reference's multi-GPU construct is one command too, nn.DataParallel inside
        # one python invocation, model/cvig_baseline.py:339-343): this process has not touched the GPU and never will
        sys.exit(launch_ranks(a.gpus))
    if world != a.gpus:
        sys.exit('bench.py --gpus %d was started with WORLD_SIZE=%d' % (a.gpus, world))
    if a.single_device:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(a.backend)

    from witw_amd import _lib, cvig_fov, ops, synth, parallel
    _lib.check(_lib.load().witw_device_check(local), 'witw_device_check')

    if a.mode == 'retrieval':
        return retrieval(a, rank, world, device, cvig_fov, ops)

    B = a.batch
    seed = 1234
    semantic = a.model == 'semantic'
    channels = 5 if semantic else 3
    if semantic:
        from witw_amd import cvig_semantic as model_mod
    else:
        model_mod = cvig_fov
    wts = synth.fov_dsm_weights(seed, in_channels=channels)
    surface_encoder = model_mod.FOV_DSM(circ_padding=False, weights=wts).to(device)
    overhead_encoder = model_mod.FOV_DSM(circ_padding=True, weights=wts).to(device)
    train = a.mode == 'train'
    bf16 = a.precision == 'bf16'
    f16x3 = a.precision == 'fp16x3'      # fp32-grade products as fp16 hi/lo triples on the fp16 MFMA (inference)
    if bf16 and train:
        surface_encoder.precision = overhead_encoder.precision = 'bf16'     # mixed-precision step, fp32 master weights
    if f16x3 and train:
        surface_encoder.precision = overhead_encoder.precision = 'fp16x3'   # forward, dgrad and wgrad on fp16x3
    surface_encoder.train(train)
    overhead_encoder.train(train)
    all_params = list(surface_encoder.parameters()) + list(overhead_encoder.parameters())
    optimizer = cvig_fov.Adam(all_params, lr=1.E-5) if train else None
    reducer = parallel.OverlappedGradReducer([surface_encoder, overhead_encoder]) if train else None
    ground_raw, ov_raw = make_inputs(cvig_fov, ops, synth, B, a.fov, seed + rank, device, channels)
    ws = int(a.fov / 360 * 512)
    mean, std = model_mod.Globals.img_mean, model_mod.Globals.img_std
    ndiv = 3 if semantic else None      # only the RGB bands are /255 (model/cvig_semantic.py:172-176)

    def train_step():
        with torch.no_grad():
            surface = ops.resize_bilinear(ground_raw, (128, ws), mean, std, ndiv)
            overhead = ops.resize_bilinear(ov_raw, (256, 256), mean, std, ndiv)
            polar = ops.polar_transform(overhead)
        su = surface_encoder(surface)
        ov = overhead_encoder(polar)
        loss, ori, d = cvig_fov.sharded_match_loss(ov, su)       # global-batch loss from this rank's [B_global, B] slab
        optimizer.zero_grad()
        loss.backward()          # each encoder's gradient all-reduce starts as soon as its backward node has run
        reducer.wait()
        optimizer.step()
        with torch.no_grad():
            ranks = ops.rank_count(d, rank * B)
        return loss.detach(), ranks, ori

    def infer_step():
        with torch.no_grad():
            surface = ops.resize_bilinear(ground_raw, (128, ws), mean, std, ndiv)
            overhead = ops.resize_bilinear(ov_raw, (256, 256), mean, std, ndiv)
            polar = ops.polar_transform(overhead)
            if f16x3:
                su, ov = surface_encoder.forward_f16x3(surface), overhead_encoder.forward_f16x3(polar)
            else:
                su = surface_encoder.forward_bf16(surface) if bf16 else surface_encoder(surface)
                ov = overhead_encoder.forward_bf16(polar) if bf16 else overhead_encoder(polar)
            ov_all = parallel._all_gather_cat(ov) if world > 1 else ov     # global gallery; surfaces stay local
            loss, ranks, ori, d = cvig_fov.evaluate_global_batch(ov_all, su, rank * B)
        return loss, ranks, ori

    step = train_step if train else infer_step
    if a.graph:
        if train or world > 1:
            sys.exit('--graph captures the single-GPU inference step only (no collectives, no optimizer)')

        def graph_body(g_raw, o_raw):
            with torch.no_grad():
                surface = ops.resize_bilinear(g_raw, (128, ws), mean, std, ndiv)
                polar = ops.polar_transform(ops.resize_bilinear(o_raw, (256, 256), mean, std, ndiv))
                su = surface_encoder.forward_bf16(surface) if bf16 else surface_encoder(surface)
                ov = overhead_encoder.forward_bf16(polar) if bf16 else overhead_encoder(polar)
                return cvig_fov.evaluate_global_batch(ov, su, 0)[:3]
        captured = parallel.CapturedStep(graph_body, [ground_raw, ov_raw])
        step = lambda: captured(ground_raw, ov_raw)      # noqa: E731  (input copy + one hipGraphLaunch)
    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        loss, ranks, ori = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rk = [torch.empty_like(ranks) for _ in range(world)]
        dist.all_gather(rk, ranks)
        ranks = torch.cat(rk)
    if f16x3 and ops.f16x3_overflowed(device):
        sys.exit('fp16x3: an activation left the fp16 range; the run is invalid')
    ranks_h = ranks.cpu().numpy().astype(np.int64)
    pairs = B * world * a.steps
    value = pairs / dt

    # ---- live roofline of the dominant kernel (HIP events on the launch stream, timed region only)
    dominant = ('bf16', 128, 1, False) if bf16 else ('f16x3', 128, 1, False) if f16x3 else DOMINANT
    # fp16x3 executes 28 fp16 MFMAs (K=16) per 16 input channels and 9 taps where a plain fp16 conv needs 9: its bound in
    # fp32-equivalent FLOP/s is the dense fp16 MFMA peak x 9/28
    peak = PEAK_BF16_MFMA_TFLOPS if bf16 else round(PEAK_BF16_MFMA_TFLOPS * 9 / 28, 1) if f16x3 else PEAK_F32_MFMA_TFLOPS
    kname = 'conv3x3_nhwc_bf16_kernel<128,1,false,8>' if bf16 else 'conv3x3_nhwc_f16x3_kernel<128,1,false,8>' if f16x3 else \
        'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>'
    dom = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v == dominant]
    allc = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v[0] not in ('match', 'wgrad_bf16', 'wgrad_f16x3')]
    wg = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v[0] == 'wgrad_bf16']
    dom_fl = sum(f for f, _ in dom) / max(1, len(dom))
    dom_ms = sum(m for _, m in dom) / max(1, len(dom))
    achieved = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
    conv_tf = sum(f for f, _ in allc) / (sum(m for _, m in allc) * 1e-3) / 1e12 if allc else 0.0
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath) and a.fov == 360:      # the PMC passes were taken at fov 360 (other widths change the launch mix)
        try:
            tag = ('_bf16_train' if bf16 else '_train') if train else ''     # train modes average forward + dgrad launches
            traffic = json.load(open(tpath)).get('%s_bytes_per_launch_B%d%s' % (kname, B, tag))
        except Exception:
            traffic = None

    mname = 'cvig_semantic (5-channel)' if semantic else 'cvig_fov'
    out = {
        'metric': 'image-pairs/sec (embedding+similarity)' if not train else 'image-pairs/sec (training step)', 'value': round(value, 2), 'unit': 'pairs/s',
        'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3),
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'bf16' if bf16 else ('f16x3 forward, dgrad, wgrad (fp16 hi+lo operands, fp32 accumulate; fp32 gradients and Adam)' if train else
                                     'f16x3 (fp16 hi+lo operands, 3 fp16 MFMAs per fp32-equivalent product, fp32 accumulate)') if f16x3 else 'f32',
        'data': 'synthetic',
        'config': {'workload': ('%s fov=%d eval%s: resize+normalize+polar -> 2x FOV_DSM (VGG16[:23]+3 conv) -> '
                                'fused match + soft-margin triplet loss + rank counts' % (mname, a.fov, ' [bf16 MFMA encoders, fp32 accumulate; matching fp32]' if bf16 else ' [fp16x3 encoders: fp32-grade products on the fp16 MFMA; matching fp32]' if f16x3 else '')) if not train else
                               ('%s fov=%d TRAIN step: resize+normalize+polar -> 2x FOV_DSM fwd (Dropout2d) -> match + '
                                'triplet loss -> backward (%s) -> grad all-reduce -> Adam'
                                % (mname + (' [bf16 MFMA fwd/dgrad/wgrad, fp32 accumulate + master weights]' if bf16 else
                                            ' [forward, dgrad and wgrad on fp16x3 (fp32-grade products on the fp16 MFMA), fp32 gradients / Adam]' if f16x3 else ''), a.fov,
                                   'dgrad L2-27, max-pool scatter, wgrad L0 + L17-27' if semantic else 'dgrad L19-27, wgrad L17-27')),
                   'pairs_per_gpu': B, 'global_batch': B * world, 'ground_raw': '%dx224x224' % channels,
                   'overhead_raw': '%dx512x512' % channels,
                   'parallelism': 'dp%d (overhead-embedding all-gather, global-batch loss from column slabs)' % world,
                   **({'launch': 'whole step replayed as one hipGraph'} if a.graph else {})},
        'recall': {'top1_pct': float(np.mean(ranks_h <= 1) * 100), 'top5_pct': float(np.mean(ranks_h <= 5) * 100),
                   'N': int(len(ranks_h))},
        'loss': float(loss.item()),
        **({'recall_note': 'train mode: the two encoders draw independent Dropout2d masks (reference :287-288) on random-init '
                           'weights, so the in-step recall is near chance; the eval-mode recall is the inference bench line'}
           if train else {}),
        'roofline': {'bound': 'mfma', 'kernel': kname, 'achieved': round(achieved, 2),
                     'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4),
                     'traffic': traffic, 'launches': len(dom), 'avg_launch_ms': round(dom_ms, 4),
                     'avg_launch_gflop': round(dom_fl / 1e9, 2), 'all_conv_launches_tflops': round(conv_tf, 2),
                     **({'wgrad_bf16_tflops_incl_layout_passes': round(sum(f for f, _ in wg) / (sum(m for _, m in wg) * 1e-3) / 1e12, 2)}
                        if wg else {}),
                     **({'note': 'achieved = fp32-equivalent FLOP/s (2*Cin*Cout*9 per output); peak = dense fp16 MFMA peak x 9/28 '
                                 '(the split arithmetic issues 28 MFMAs where a plain fp16 conv issues 9); the fp32 MFMA peak is 157.3'}
                        if f16x3 else {})},
    }

    if rank == 0 and world == 1 and not a.no_cpu_baseline and not train and not bf16 and not f16x3:
        out['cpu_baseline'] = cpu_baseline(a, ground_raw, ov_raw, wts, ws, step, semantic)
    if world == 1 and not train and not bf16 and not f16x3 and not a.graph and not a.no_side_blocks:
        # the same step with fp32-grade products from fp16 hi/lo pairs on the fp16 MFMA (--precision fp16x3), reported beside
        # the headline, never as `value`: same inputs, same weights, embeddings compared with the exact-fp32 kernels'
        def step3():
            with torch.no_grad():
                surface = ops.resize_bilinear(ground_raw, (128, ws), mean, std, ndiv)
                polar = ops.polar_transform(ops.resize_bilinear(ov_raw, (256, 256), mean, std, ndiv))
                su3, ov3 = surface_encoder.forward_f16x3(surface), overhead_encoder.forward_f16x3(polar)
                return cvig_fov.evaluate_global_batch(ov3, su3, 0) + (su3, ov3, surface, polar)
        for _ in range(2):
            r3 = step3()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        for _ in range(a.steps):
            r3 = step3()
        torch.cuda.synchronize()
        dt3 = (time.perf_counter() - t3) / a.steps
        with torch.no_grad():
            su32, ov32 = surface_encoder(r3[6]), overhead_encoder(r3[7])
        diff = max(float((r3[4] - su32).abs().max()), float((r3[5] - ov32).abs().max()))
        out['fp32_grade_on_fp16_mfma'] = {
            'precision': 'fp16x3', 'value': round(B / dt3, 2), 'unit': 'pairs/s', 'ms_per_step': round(dt3 * 1e3, 3),
            'max_abs_embedding_diff_vs_f32_kernels': diff, 'loss': float(r3[0].item()),
            'recall': {'top1_pct': float((r3[1] <= 1).float().mean().item() * 100), 'top5_pct': float((r3[1] <= 5).float().mean().item() * 100)},
            'ranks_differing_from_f32_step': int((r3[1] != ranks).sum().item()), 'overflow': bool(ops.f16x3_overflowed(device)),
            'note': 'operands carried as fp16 hi + lo, products hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16, fp32 accumulate; '
                    'held to the reference goldens at the same 1e-4 as the f32 kernels (tests/test_f16x3_gpu.py)'}
    if world == 1 and not train and not bf16 and not f16x3 and not a.graph and not semantic and a.fov == 360 and not a.no_side_blocks:
        out['config5_retrieval'] = retrieval_summary(device, cvig_fov, ops)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def retrieval_summary(device, cvig_fov, ops, G=125000, Q=10000, k=10):
    """BASELINE config 5's per-GPU share (125,000 gallery rows x 10,000 queries, ranks + top-10) measured beside the headline
    with the spectral orientation search: one warm-up and one timed pass (`--mode retrieval [--match dft]` is the full line,
    the direct-sum pass takes 4.6 s)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(4321)
    gallery = torch.randn((G, 16, 4, 64), generator=gen, device=device)
    shifts = torch.randint(0, 64, (Q,), generator=gen, device=device)
    col = (torch.arange(64, device=device)[None, :] + shifts[:, None]) % 64
    queries = torch.gather(gallery[:Q], 3, col[:, None, None, :].expand(-1, 16, 4, -1)) \
        + 10.0 * torch.randn((Q, 16, 4, 64), generator=gen, device=device)
    cvig_fov.retrieve(gallery, queries, k=k, method='dft')
    torch.cuda.synchronize()
    ops.PROFILE = []
    t0 = time.perf_counter()
    ranks_h, vals, idx = cvig_fov.retrieve(gallery, queries, k=k, method='dft')
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    m = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v[0] == 'match_dft']
    tf = sum(f for f, _ in m) / (sum(t for _, t in m) * 1e-3) / 1e12 if m else 0.0
    st = cvig_fov.retrieve.last_stats
    return {'workload': 'gallery retrieval: %d overhead embeddings x %d ground queries, fov=360, ranks + top-%d' % (G, Q, k),
            'match': 'dft (witw_match_fwd_dft: orientation search through 64-point row spectra, 21,120 FLOP per pair)',
            'value': round(float(G) * Q / dt, 1), 'unit': 'pairs/s', 'ms_per_step': round(dt * 1e3, 3),
            'recall': {'top1_pct': float(np.mean(ranks_h <= 1) * 100), 'top10_pct': float(np.mean(ranks_h <= 10) * 100), 'N': G},
            'match_kernel_tflops': round(tf, 2), 'match_kernel_frac_of_f32_mfma_peak': round(tf / PEAK_F32_MFMA_TFLOPS, 4),
            'index_exact': rescore_block(st),
            'direct_sum_pairs_per_s_in_profiles_r01': 2.73e8,
            'note': 'the direct-sum pass (witw_match_fwd, 524,288 FLOP per pair at 0.91 of the fp32 MFMA peak) is '
                    'profiles/r01_bench_retrieval.json; same recall figures'}


def rescore_block(st):
    """What the index-exact contract of retrieve(method='dft') cost in the timed pass: pairs re-scored with the direct
    arithmetic (witw_match_pairs) per million pairs of the pass, and queries that took the direct pass outright."""
    n = st['rescored_rank'] + st['rescored_topk'] + st['rescored_true'] + st['rescored_orientation']
    return {'rescored_pairs': int(n), 'rescored_per_million': round(n / max(1.0, st['pairs']) * 1e6, 2),
            'fallback_queries': int(st['fallback_queries']), 'distance_eps': st['eps'],
            'contract': 'ranks and top-k indices equal the direct pass (tests/test_match_dft_gpu.py)'}


def retrieval(a, rank, world, device, cvig_fov, ops):
    """BASELINE config 5 (gallery retrieval): every rank holds --gallery overhead embeddings [16,4,64] (weak
    scaling: 8 x 125k = 1M rows), the --queries ground embeddings are replicated; one step = fused orientation
    search + chord distance of every (gallery row, query) pair (524,288 FLOP each at fov 360), rank counts
    against the true match and the k nearest rows, merged over ranks (A7-A9, A12; model/cvig_fov.py:543-552)."""
    G, Q, k = a.gallery, a.queries, a.topk
    we = int(a.fov / 360 * 512) // 8
    gen = torch.Generator(device=device)
    gen.manual_seed(4321 + rank)
    gallery = torch.randn((G, 16, 4, 64), generator=gen, device=device)
    # query q = gallery row q (global numbering, rank-major) rolled by a per-query shift, cropped to the FoV, plus noise
    queries = torch.zeros((Q, 16, 4, we), device=device)
    lo, hi = rank * G, min(Q, (rank + 1) * G)
    if hi > lo:
        shifts = torch.randint(0, 64, (hi - lo,), generator=gen, device=device)
        col = (torch.arange(we, device=device)[None, :] + shifts[:, None]) % 64                      # [n, we]
        rows = gallery[lo - rank * G:hi - rank * G]
        queries[lo:hi] = torch.gather(rows, 3, col[:, None, None, :].expand(-1, 16, 4, -1)) \
            + 10.0 * torch.randn((hi - lo, 16, 4, we), generator=gen, device=device)
    if world > 1:
        dist.all_reduce(queries)

    def step():
        return cvig_fov.retrieve(gallery, queries, k=k, shard_begin=rank * G, method=a.match)

    for _ in range(a.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ranks_h, vals, idx = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    m = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v[0] in ('match', 'match_dft')]
    m_fl = sum(f for f, _ in m) / max(1, len(m))
    m_ms = sum(t for _, t in m) / max(1, len(m))
    achieved = m_fl / (m_ms * 1e-3) / 1e12 if m_ms > 0 else 0.0
    top1_hit = float((idx[:, 0].cpu().numpy() == np.arange(Q)).mean() * 100)
    out = {
        'metric': 'query-gallery pairs/sec (orientation search + distance + rank + top-%d)' % k,
        'value': round(float(G) * world * Q * a.steps / dt, 1), 'unit': 'pairs/s', 'n_gpus': world, 'steps': a.steps,
        'warmup': a.warmup, 'ms_per_step': round(dt / a.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'gallery retrieval: %d overhead embeddings per GPU (%d total) x %d ground queries, fov=%d, '
                               'ranks + top-%d' % (G, G * world, Q, a.fov, k),
                   'match': a.match,
                   'parallelism': 'gallery rows sharded over %d rank(s); all-reduce of true distances and rank counts, '
                                  'all-gather + merge of top-k candidates' % world},
        'queries_per_sec': round(Q * a.steps / dt, 1),
        'recall': {'top1_pct': float(np.mean(ranks_h <= 1) * 100), 'top5_pct': float(np.mean(ranks_h <= 5) * 100),
                   'top10_pct': float(np.mean(ranks_h <= 10) * 100), 'N': int(G * world), 'topk_first_is_true_pct': top1_hit},
        'roofline': {'bound': 'mfma', 'kernel': 'witw_match_fwd_dft launch (match_dft_kernel + norm / table kernels), FLOP of the '
                                                'spectral form: 21,120 per pair' if a.match == 'dft' else
                                                'witw_match_fwd launch (match_kernel_w64 + 2 norm kernels)',
                     'achieved': round(achieved, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                     'frac': round(achieved / PEAK_F32_MFMA_TFLOPS, 4), 'traffic': None, 'launches': len(m),
                     'avg_launch_ms': round(m_ms, 3), 'avg_launch_gflop': round(m_fl / 1e9, 1)},
    }
    if a.match == 'dft':
        out['index_exact'] = rescore_block(cvig_fov.retrieve.last_stats)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(a, ground_raw, ov_raw, wts, ws, gpu_step, semantic=False):
    """The oracle (CPU restatement of the reference, kind 'port') timed on this host on a bounded
    sample of the same workload, plus a parity check of the GPU step against it on that sample."""
    from oracle import cvig_fov_oracle as O
    n = min(a.cpu_pairs, ground_raw.shape[0])
    g = ground_raw[:n].cpu()
    o = ov_raw[:n].cpu()
    w = {k: (torch.from_numpy(v[0]), torch.from_numpy(v[1])) for k, v in wts.items()}
    threads = torch.get_num_threads()
    norm = O.image_normalization_semantic if semantic else O.image_normalization

    def cpu_step():
        with torch.no_grad():
            su_in, ov_in = [], []
            for i in range(n):
                s, ov = O.resize_pair(g[i], o[i], fov=a.fov, panorama=False)
                su_in.append(norm(s))
                ov_in.append(O.polar_transform(norm(ov)))
            su = O.fov_dsm_forward(torch.stack(su_in), w, False)
            ov = O.fov_dsm_forward(torch.stack(ov_in), w, True)
            ori, d = O.match(ov, su)
            loss = O.triplet_loss(d)
            ranks = (d <= torch.diagonal(d)[None, :]).sum(0)
        return su, ov, ori, d, loss, ranks

    cpu_step()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        su, ov, ori, d, loss, ranks = cpu_step()
        times.append(time.perf_counter() - t0)
    med = sorted(times)[1]
    cpu_model = ''
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                cpu_model = line.split(':', 1)[1].strip()
                break
    except OSError:
        pass
    return {'value': round(n / med, 3), 'unit': 'pairs/s', 'cores': threads, 'kind': 'port', 'cpu': cpu_model,
            'sample': '%d pairs of the same synthetic batch, full step (transforms+encoders+match+loss+ranks), '
                      'median of 3 after 1 warm-up, torch %s CPU ops' % (n, torch.__version__)}


if __name__ == '__main__':
    main()
