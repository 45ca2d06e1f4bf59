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
Prints ONE JSON line on rank 0. At N=1 the line also carries one compact block per other single-GPU BASELINE config
(config 1 cvig_baseline, the config-2 training step, config 4 cvig_semantic on the bf16 MFMA, config 5 retrieval), each
measured in this run with its dominant kernel's roofline; `--no-side-blocks` leaves them out.
"""
import argparse
import hashlib
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
PEAK_HBM_GBS = 8000.0           # same guide: HBM3E ~8 TB/s
LINE_BUDGET = 6144             # bytes of the one stdout line (the driver's record keeps the last 8 KB of stdout)
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


def sha16(path):
    try:
        return hashlib.sha256(open(path, 'rb').read()).hexdigest()[:16]
    except OSError:
        return None


def traffic_of(kname, batch, tag):
    """HBM/fabric bytes per launch of the dominant kernel from the committed PMC passes (profiles/traffic.json: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs, corrected as MI355X_MICROARCH.md prescribes). Counters cannot be read
    inside a plain run, so the value is tied to the kernel SOURCES it was taken on: if those csrc files changed since, the
    figure is withheld (null) and `stale` says so. -> (bytes or None, traffic_source dict)."""
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    src = {'file': 'profiles/traffic.json', 'how': 'rocprofv3 --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes), tools/make_traffic.py'}
    try:
        t = json.load(open(tpath))
    except Exception:
        return None, dict(src, stale=True, reason='no traffic.json')
    recorded = t.get('_kernel_sources_sha16', {})
    current = {f: sha16(os.path.join(ROOT, 'witw_amd', 'csrc', f)) for f in recorded}
    stale = (not recorded) or any(current[f] != recorded[f] for f in recorded)
    src.update(kernel_sources_sha16=recorded, stale=bool(stale), libwitw_hip_sha16=sha16(os.path.join(ROOT, 'witw_amd', 'libwitw_hip.so')))
    if stale:
        return None, src
    return t.get('%s_bytes_per_launch_B%d%s' % (kname, batch, tag)), src


def rocprof_avg_of(kname, tag):
    """Average duration of `kname` in the committed rocprofv3 --kernel-trace --stats table of bench mode `tag`
    (profiles/rocprof_kernel_avg.json, tools/make_rocprof_avg.py) -> (ms, csv path) or (None, None); withheld when any kernel
    source changed since the table was taken (same rule as traffic_of)."""
    try:
        t = json.load(open(os.path.join(ROOT, 'profiles', 'rocprof_kernel_avg.json')))
    except Exception:
        return None, None
    recorded = t.get('_kernel_sources_sha16', {})
    if not recorded or any(sha16(os.path.join(ROOT, 'witw_amd', 'csrc', f)) != h for f, h in recorded.items()):
        return None, None
    blk = t.get(tag) or {}
    k = (blk.get('kernels') or {}).get(kname.replace(' ', ''))
    return (k['avg_ms'], blk.get('csv')) if k else (None, None)


class StepBench(object):
    """One configuration of the cvig_fov / cvig_semantic step (model, mode, precision, batch, fov): builds the synthetic batch
    and the two encoders, times K steps between barrier + synchronize pairs, and derives the live roofline of the dominant
    kernel from HIP events recorded around every launch on the launch stream (ops.PROFILE)."""

    def __init__(self, model, mode, precision, batch, fov, rank, world, device, graph=False, share=None, pair=True):
        """share: another StepBench of the same model / fov whose weights and first `batch` input pairs are reused (batch sweep)"""
        from witw_amd import cvig_fov, ops, synth, parallel
        self.cvig_fov, self.ops, self.synth, self.parallel = cvig_fov, ops, synth, parallel
        self.model, self.mode, self.precision, self.B, self.fov = model, mode, precision, batch, fov
        self.rank, self.world, self.device, self.graph = rank, world, device, graph
        self.semantic = model == 'semantic'
        self.channels = 5 if self.semantic else 3
        if self.semantic:
            from witw_amd import cvig_semantic as model_mod
        else:
            model_mod = cvig_fov
        self.model_mod = model_mod
        seed = 1234
        self.train = mode == 'train'
        self.bf16, self.f16x3 = precision == 'bf16', precision == 'fp16x3'
        self.wts = share.wts if share is not None else synth.fov_dsm_weights(seed, in_channels=self.channels)
        self.se = model_mod.FOV_DSM(circ_padding=False, weights=self.wts).to(device)
        self.oe = model_mod.FOV_DSM(circ_padding=True, weights=self.wts).to(device)
        if self.bf16 and self.train:
            self.se.precision = self.oe.precision = 'bf16'        # mixed-precision step, fp32 master weights
        if self.f16x3 and self.train:
            self.se.precision = self.oe.precision = 'fp16x3'      # forward, dgrad and wgrad on fp16x3
        self.se.train(self.train)
        self.oe.train(self.train)
        params = list(self.se.parameters()) + list(self.oe.parameters())
        self.optimizer = cvig_fov.Adam(params, lr=1.E-5) if self.train else None
        self.reducer = parallel.OverlappedGradReducer([self.se, self.oe]) if self.train else None
        if share is not None:
            assert share.channels == self.channels and share.fov == fov and share.B >= batch
            self.ground_raw, self.ov_raw = share.ground_raw[:batch].contiguous(), share.ov_raw[:batch].contiguous()
        else:
            self.ground_raw, self.ov_raw = make_inputs(cvig_fov, ops, synth, batch, fov, seed + rank, device, self.channels)
        self.ws = int(fov / 360 * 512)
        self.mean, self.std = model_mod.Globals.img_mean, model_mod.Globals.img_std
        self.ndiv = 3 if self.semantic else None      # only the RGB bands are /255 (model/cvig_semantic.py:172-176)
        self.step = self.train_step if self.train else self.infer_step
        self.pair = None
        if pair and not self.train and precision in ('fp32', 'bf16') and batch <= 64:
            self.se.precision = self.oe.precision = precision
            self.pair = cvig_fov.PairEmbedder(self.se, self.oe)
        if graph:
            if self.train or world > 1:
                sys.exit('--graph captures the single-GPU inference step only (no collectives, no optimizer)')
            captured = parallel.CapturedStep(self.graph_body, [self.ground_raw, self.ov_raw])
            self.step = lambda: captured(self.ground_raw, self.ov_raw)      # input copy + one hipGraphLaunch

    def preprocess(self, g_raw, o_raw):
        ops = self.ops
        surface = ops.resize_bilinear(g_raw, (128, self.ws), self.mean, self.std, self.ndiv)
        polar = ops.polar_from_raw(o_raw, mean=self.mean, std=self.std, n_div255=self.ndiv)      # resize + normalise + polar: one launch
        return surface, polar

    def embed(self, surface, polar, precision=None):
        p = precision or self.precision
        if self.pair is not None and precision is None:
            # the drivers' path at the reference's default batch sizes (cvig_fov.PairEmbedder: both encoders at once on two streams,
            # the bf16 pair as one hipGraph); inside a whole-step capture only the two-stream form
            if torch.cuda.is_current_stream_capturing() or self.graph:
                return (self.pair._dual if self.pair.dual_for(surface.shape[0]) else self.pair._plain)(surface, polar)
            return self.pair(surface, polar)
        if p == 'fp16x3':
            return self.se.forward_f16x3(surface), self.oe.forward_f16x3(polar)
        if p == 'bf16':
            return self.se.forward_bf16(surface), self.oe.forward_bf16(polar)
        return self.se(surface), self.oe(polar)

    def train_step(self):
        phase = self.parallel.phase
        with torch.no_grad(), phase('preprocess'):
            surface, polar = self.preprocess(self.ground_raw, self.ov_raw)
        with phase('encoders_forward'):
            su = self.se(surface)
            ov = self.oe(polar)
        loss, ori, d = self.cvig_fov.sharded_match_loss(ov, su)     # global-batch loss from this rank's [B_global, B] slab
        self.optimizer.zero_grad()
        with phase('backward_incl_its_collectives'):
            loss.backward()      # each encoder's gradient all-reduce starts as soon as its backward node has run
        self.reducer.wait()
        with phase('adam'):
            self.optimizer.step()
        with torch.no_grad():
            ranks = self.ops.rank_count(d, self.rank * self.B)
        return loss.detach(), ranks, ori

    def infer_step(self):
        phase = self.parallel.phase
        with torch.no_grad():
            with phase('preprocess'):
                surface, polar = self.preprocess(self.ground_raw, self.ov_raw)
            with phase('encoders_forward'):
                su, ov = self.embed(surface, polar)
            with phase('overhead_all_gather'):
                ov_all = self.parallel._all_gather_cat(ov) if self.world > 1 else ov     # global gallery; surfaces stay local
            loss, ranks, ori, d = self.cvig_fov.evaluate_global_batch(ov_all, su, self.rank * self.B)
        return loss, ranks, ori

    def graph_body(self, g_raw, o_raw):
        with torch.no_grad():
            surface, polar = self.preprocess(g_raw, o_raw)
            su, ov = self.embed(surface, polar)
            return self.cvig_fov.evaluate_global_batch(ov, su, 0)[:3]

    def run(self, steps, warmup):
        ops, world, device = self.ops, self.world, self.device
        for _ in range(warmup):
            self.step()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        ops.PROFILE = []
        # per-phase device time of the step (HIP events on the launch stream, witw_amd/parallel.py PhaseTimer): what explains an
        # N > 1 line -- the collectives' brackets, the stall of the gradient join -- and costs ~15 event records per step
        timer = self.parallel.PHASES = None if self.graph else self.parallel.PhaseTimer()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, ranks, ori = self.step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        prof, ops.PROFILE = ops.PROFILE, None
        self.parallel.PHASES = None
        self.phases = timer.summary(steps) if timer is not None else {}
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            rk = [torch.empty_like(ranks) for _ in range(world)]
            dist.all_gather(rk, ranks)
            ranks = torch.cat(rk)
        if self.f16x3 and ops.f16x3_overflowed(device):
            sys.exit('fp16x3: an activation left the fp16 range; the run is invalid')
        self.dt, self.steps, self.warmup = dt, steps, warmup
        self.loss, self.ranks = loss, ranks
        self.ranks_h = ranks.cpu().numpy().astype(np.int64)
        self.value = self.B * world * steps / dt
        self.ms = dt / steps * 1e3
        self.roofline = self._roofline(prof)
        if self.fov == 360 and not self.f16x3:
            # algorithmic conv FLOP of the WHOLE step (cvig_fov 75.51 GFLOP per pair eval, 124.15 train; cvig_semantic
            # 2 x 18.953 GMAC eval, + 2 x 25.6 GMAC backward) over the wall-clock step, against the same peak: what the step as a
            # whole reaches, launch gaps and every non-dominant kernel included
            gf = ((2 * 2 * 18.953 + (2 * 2 * 25.6 if self.train else 0.0)) if self.semantic else (124.15 if self.train else 75.51))
            self.roofline['whole_step_frac'] = round(gf * 1e9 * self.B / (self.ms * 1e-3) / 1e12 / self.roofline['peak'], 4)
        return self

    def _roofline(self, prof):
        """Live roofline of the dominant kernel (HIP events on the launch stream, timed region only)."""
        bf16, f16x3 = self.bf16, self.f16x3
        dominant = ('bf16', 128, 1, False) if bf16 else ('f16x3', 128, 1, False) if f16x3 else DOMINANT
        # fp16x3 executes 28 fp16 MFMAs (K=16) per 16 input channels and 9 taps where a plain fp16 conv needs 9: its bound in
        # fp32-equivalent FLOP/s is the dense fp16 MFMA peak x 9/28
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else round(PEAK_BF16_MFMA_TFLOPS * 9 / 28, 1) if f16x3 else PEAK_F32_MFMA_TFLOPS
        # the bf16 inference forward runs its large layers on the 16x16x32 kernel (ops.bf16_mfma16); a bf16 training step mixes it
        # (frozen forward layers) with the 32x32x16 kernel (Dropout2d layers, dgrad) under the same launch class
        s16 = bf16 and not self.train and self.ops.bf16_mfma16()
        s16t = bf16 and self.train and self.ops.bf16_mfma16()       # training: the same launch class on the TRAIN instantiation
        kname = ('conv3x3_bf16_s16_kernel<false,false>' if s16 else
                 'conv3x3_bf16_s16_kernel<false,true>' if s16t else      # Dropout2d forwards + dgrad launches (<false,false>: frozen layers)
                 'conv3x3_nhwc_bf16_kernel<128,1,false,8>') if bf16 else \
            'conv3x3_nhwc_f16x3_kernel<128,1,false,8>' if f16x3 else 'conv3x3_nhwc_f32_kernel<128,1,false,8,0,9>'
        dom = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v == dominant]
        allc = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v[0] not in ('match', 'wgrad_bf16', 'wgrad_f16x3')]
        wg = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v[0] == 'wgrad_bf16']
        wres = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v[0] == 'bf16_wres']      # layer 5 (64 input channels)
        dom_fl = sum(f for f, _ in dom) / max(1, len(dom))
        dom_ms = sum(m for _, m in dom) / max(1, len(dom))
        achieved = dom_fl / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        conv_tf = sum(f for f, _ in allc) / (sum(m for _, m in allc) * 1e-3) / 1e12 if allc else 0.0
        traffic, tsrc = None, None
        if self.fov == 360:      # the PMC passes were taken at fov 360 (other widths change the launch mix)
            tag = ('_bf16_train' if bf16 else '_train') if self.train else ''     # train modes average forward + dgrad launches
            traffic, tsrc = traffic_of(kname, self.B, tag)
        # the same fraction on rocprofv3's clock: the committed --kernel-trace --stats table of this bench mode (same kernel sources),
        # same FLOP per launch. Events bracket the launch on the stream, rocprof times the dispatch: they differ by 0.5 % on the fp32
        # kernel and by up to 5 % on the bf16 ones, so the line carries both.
        mode_tag = ('sem_' if self.semantic else '') + ('bf16' if bf16 else 'f16x3' if f16x3 else '') + ('_train' if self.train else '')
        mode_tag = mode_tag.strip('_') or 'infer'
        rp_ms, rp_csv = rocprof_avg_of(kname, mode_tag) if self.fov == 360 and self.B == 128 else (None, None)
        clocks = {'frac_events': round(achieved / peak, 4)}
        if rp_ms:
            clocks.update(frac_rocprof=round(dom_fl / (rp_ms * 1e-3) / 1e12 / peak, 4), rocprof_avg_launch_ms=rp_ms, rocprof_csv=rp_csv)
        return {'bound': 'mfma', 'kernel': kname, 'achieved': round(achieved, 2),
                'peak': peak, 'unit': 'TFLOP/s', 'frac': round(achieved / peak, 4), **clocks,
                'traffic': traffic, **({'traffic_source': tsrc} if tsrc else {}),
                'launches': len(dom), 'avg_launch_ms': round(dom_ms, 4),
                'avg_launch_gflop': round(dom_fl / 1e9, 2), 'all_conv_launches_tflops': round(conv_tf, 2),
                **({'layer5_weight_resident_kernel': {'kernel': 'conv3x3_bf16_wres_kernel', 'launches': len(wres),
                                                      'avg_launch_ms': round(sum(m for _, m in wres) / len(wres), 4),
                                                      'tflops': round(sum(f for f, _ in wres) / (sum(m for _, m in wres) * 1e-3) / 1e12, 2),
                                                      'note': 'not part of the dominant kernel\'s launches: 64 -> 128 channels, K = 576, the filter block stays in LDS'}}
                   if wres else {}),
                **({'wgrad_bf16_tflops_incl_layout_passes': round(sum(f for f, _ in wg) / (sum(m for _, m in wg) * 1e-3) / 1e12, 2)}
                   if wg else {}),
                **({'note': 'achieved = fp32-equivalent FLOP/s (2*Cin*Cout*9 per output); peak = dense fp16 MFMA peak x 9/28 '
                            '(the split arithmetic issues 28 MFMAs where a plain fp16 conv issues 9); the fp32 MFMA peak is 157.3'}
                   if f16x3 else {})}

    def dtype(self):
        if self.bf16:
            return 'bf16'
        if self.f16x3:
            return ('f16x3 forward, dgrad, wgrad (fp16 hi+lo operands, fp32 accumulate; fp32 gradients and Adam)' if self.train else
                    'f16x3 (fp16 hi+lo operands, 3 fp16 MFMAs per fp32-equivalent product, fp32 accumulate)')
        return 'f32'

    def workload(self):
        mname = 'cvig_semantic (5-channel)' if self.semantic else 'cvig_fov'
        if not self.train:
            return ('%s fov=%d eval%s: resize+normalize+polar -> 2x FOV_DSM (VGG16[:23]+3 conv) -> fused match + soft-margin '
                    'triplet loss + rank counts' % (mname, self.fov, ' [bf16 MFMA encoders, fp32 accumulate; matching fp32]' if self.bf16 else
                                                   ' [fp16x3 encoders: fp32-grade products on the fp16 MFMA; matching fp32]' if self.f16x3 else ''))
        return ('%s fov=%d TRAIN step: resize+normalize+polar -> 2x FOV_DSM fwd (Dropout2d) -> match + triplet loss -> backward (%s) '
                '-> grad all-reduce -> Adam' % (mname + (' [bf16 MFMA fwd/dgrad/wgrad, fp32 accumulate + master weights]' if self.bf16 else
                                                         ' [forward, dgrad and wgrad on fp16x3 (fp32-grade products on the fp16 MFMA), fp32 gradients / Adam]'
                                                         if self.f16x3 else ''), self.fov,
                                                'dgrad L2-27, max-pool scatter, wgrad L0 + L17-27' if self.semantic else 'dgrad L19-27, wgrad L17-27'))

    def line(self):
        rk = self.ranks_h
        return {
            'metric': 'image-pairs/sec (embedding+similarity)' if not self.train else 'image-pairs/sec (training step)',
            'value': round(self.value, 2), 'unit': 'pairs/s',
            'n_gpus': self.world, 'steps': self.steps, 'warmup': self.warmup, 'ms_per_step': round(self.ms, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': self.dtype(), 'data': 'synthetic',
            'config': {'workload': self.workload(), 'pairs_per_gpu': self.B, 'global_batch': self.B * self.world,
                       'ground_raw': '%dx224x224' % self.channels, 'overhead_raw': '%dx512x512' % self.channels,
                       'parallelism': 'dp%d (overhead-embedding all-gather, global-batch loss from column slabs)' % self.world,
                       **({'launch': 'whole step replayed as one hipGraph'} if self.graph else {})},
            'recall': {'top1_pct': float(np.mean(rk <= 1) * 100), 'top5_pct': float(np.mean(rk <= 5) * 100), 'N': int(len(rk))},
            'loss': float(self.loss.item()),
            **({'recall_note': 'train mode: the two encoders draw independent Dropout2d masks (reference :287-288) on random-init '
                               'weights, so the in-step recall is near chance; the eval-mode recall is the inference bench line'}
               if self.train else {}),
            'roofline': self.roofline,
        }

    def block(self, baseline_config, parity):
        """Compact form for a side block of the headline line."""
        r = self.roofline
        return {'baseline_config': baseline_config, 'workload': self.workload(), 'pairs_per_gpu': self.B,
                'value': round(self.value, 2), 'unit': 'pairs/s', 'ms_per_step': round(self.ms, 3), 'steps': self.steps,
                'dtype': self.dtype(), 'loss': float(self.loss.item()),
                'recall': {'top1_pct': float(np.mean(self.ranks_h <= 1) * 100), 'top5_pct': float(np.mean(self.ranks_h <= 5) * 100)},
                'roofline': {k: r[k] for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'frac_events', 'frac_rocprof', 'launches', 'avg_launch_ms',
                                               'all_conv_launches_tflops', 'whole_step_frac', 'wgrad_bf16_tflops_incl_layout_passes') if k in r},
                'parity': parity}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=128, help='pairs per GPU (BASELINE.json configs[1]: bs=128)')
    ap.add_argument('--fov', type=int, default=360)
    ap.add_argument('--mode', choices=['infer', 'train', 'retrieval', 'baseline', 'e2e', 'sweep', 'sides'], default='infer',
                    help='infer (headline): embedding + similarity; train: the full step of model/cvig_fov.py:444-461; '
                         'retrieval: BASELINE config 5, --gallery rows per GPU x --queries, ranks + top-k; baseline: BASELINE '
                         'config 1, cvig_baseline 32 pairs; e2e: disk -> embeddings through ImagePairDataset + DataLoader workers')
    ap.add_argument('--gallery', type=int, default=125000, help='retrieval: gallery rows PER GPU (1M / 8)')
    ap.add_argument('--queries', type=int, default=10000, help='retrieval: ground queries (replicated)')
    ap.add_argument('--topk', type=int, default=10)
    ap.add_argument('--match', choices=['direct', 'dft'], default='direct',
                    help="retrieval: 'dft' = orientation search through the 64-point row spectra (21k instead of 524k FLOP per "
                         "pair, same fp32 MFMA) with every rounding-level decision re-made on exact distances")
    ap.add_argument('--precision', choices=['fp32', 'bf16', 'fp16x3'], default='fp32',
                    help='fp32 (headline, BASELINE configs[1]) or the bf16 MFMA inference path (configs[3] arithmetic)')
    ap.add_argument('--model', choices=['fov', 'semantic'], default='fov',
                    help='fov = cvig_fov (3-channel, BASELINE configs[1]); semantic = cvig_semantic (5-channel first conv, '
                         'configs[3]: run it with --precision bf16)')
    ap.add_argument('--graph', action='store_true',
                    help='inference, one GPU: capture the whole step in a hipGraph (parallel.CapturedStep) and time replays; '
                         'pays off where the step is launch-bound (small --batch, bf16)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--detail-out', default=None,
                    help='where the FULL record goes (default bench_detail.json beside bench.py): stdout carries one line of at most '
                         '%d bytes -- headline, roofline, cpu_baseline and one compact entry per side config' % LINE_BUDGET)
    ap.add_argument('--no-microbench', action='store_true', help='N > 1: skip the standalone timing of the step\'s collectives')
    ap.add_argument('--no-train-step', action='store_true', help='N > 1: skip the training step (BASELINE configs[2]) run behind the inference headline')
    ap.add_argument('--no-side-blocks', action='store_true', help='only the headline measurement (no blocks for the other BASELINE configs)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL); gloo only for 1-GPU self-tests')
    ap.add_argument('--single-device', action='store_true', help='self-test: put every rank on cuda:0')
    ap.add_argument('--pg-of-one', action='store_true',
                    help='self-test on a one-GPU box: one rank, but inside a REAL process group of world size 1 (backend nccl = RCCL): the '
                         'collectives block and its microbench run through RCCL (every collective is then the identity)')
    ap.add_argument('--cpu-pairs', type=int, default=32, help='pairs of the batch the CPU oracle is timed on (cpu_baseline)')
    ap.add_argument('--e2e-pairs', type=int, default=4096)
    ap.add_argument('--decode', choices=['device', 'host'], default='device',
                    help="e2e: 'device' = DataLoader workers entropy-decode the JPEG files, the GPU does dequantisation / IDCT / upsampling / "
                         "colour conversion (byte-identical to Pillow); 'host' = Pillow in the workers (the reference's arrangement)")
    ap.add_argument('--e2e-dir', default=None, help='e2e: directory of the synthetic JPEG data set (kept; files already there are reused)')
    ap.add_argument('--jpeg-restart-rows', type=int, default=0,
                    help='e2e: write the synthetic JPEG files with a restart marker every N MCU rows (0: none, as Pillow / libjpeg write by '
                         'default); files with restart markers are entropy-decoded on the GPU, one thread per restart interval')
    ap.add_argument('--device-entropy', choices=['off', 'restart', 'all'], default=None,
                    help="e2e: which JPEG files are Huffman-decoded on the GPU (witw_amd/jpeg.py DEVICE_ENTROPY; default 'all': files with "
                         "restart markers one thread per interval, the others by the self-synchronising decode; 'restart': only the former; "
                         "'off': Huffman decoding in the loader workers)")
    ap.add_argument('--jpeg-restart-blocks', type=int, default=0, help='e2e: ... or a restart marker every N MCUs (jpegtran -restart NB)')
    ap.add_argument('--no-decode-scaling', action='store_true', help='e2e: skip the host entropy-decode scaling sweep')
    ap.add_argument('--decode-scaling-seconds', type=float, default=1.0)
    ap.add_argument('--no-ring', action='store_true', help='e2e: torch DataLoader staging (shared-memory pickling + pin_memory thread) instead of ring.PinnedRing')
    ap.add_argument('--workers', type=int, default=12, help='e2e: DataLoader workers (reference: 12, model/cvig_fov.py:402)')
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
    # RCCL prints a version banner ('RCCL version : ...', 5 lines) on STDOUT when its first communicator is created: that happens
    # inside this block (init with device_id + one warm-up collective), with file descriptor 1 pointed at stderr meanwhile, so that
    # rank 0's stdout carries the ONE JSON line and nothing else
    sys.stdout.flush()
    fd_out = os.dup(1)
    os.dup2(2, 1)
    try:
        init_group(a, world, device)
    finally:
        sys.stdout.flush()
        os.dup2(fd_out, 1)
        os.close(fd_out)

    from witw_amd import _lib, cvig_fov, ops
    _lib.check(_lib.load().witw_device_check(local), 'witw_device_check')
    run_mode(a, rank, world, local, device, _lib, cvig_fov, ops)


def init_group(a, world, device):
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if a.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(a.backend)
    elif a.pg_of_one:
        import socket
        sk = socket.socket()
        sk.bind(('127.0.0.1', 0))
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', str(sk.getsockname()[1]))
        sk.close()
        if a.backend == 'nccl':
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=device)
        else:
            dist.init_process_group(a.backend, rank=0, world_size=1)
    if dist.is_available() and dist.is_initialized():
        warm = torch.ones(1, device=device)
        dist.all_reduce(warm)                       # creates the communicator (and prints RCCL's banner) here
        torch.cuda.synchronize()


def run_mode(a, rank, world, local, device, _lib, cvig_fov, ops):

    if a.mode == 'retrieval':
        out = retrieval(a, rank, world, device, cvig_fov, ops)
    elif a.mode == 'baseline':
        out = baseline_bench(a, device, full=not a.no_cpu_baseline)
    elif a.mode == 'e2e':
        out = e2e_bench(a, device)
    elif a.mode == 'sweep':
        out = batch_sweep(a, rank, world, device, ops)
    elif a.mode == 'sides':
        out = side_blocks(a, rank, world, device, cvig_fov, ops)
    else:
        out = step_line(a, rank, world, device, cvig_fov, ops)
    bad = None
    after = out.pop('_after_group', None) if isinstance(out, dict) else None
    if after is not None:
        # every collective of the line is behind us: the group goes first, so that no rank sits in a collective (or its watchdog)
        # while rank 0 spends ~40 s on the CPU port
        dist.barrier()
        dist.destroy_process_group()
        try:
            after()
        except Exception as e:
            out['cpu_baseline'] = {'error': '%s: %s' % (type(e).__name__, str(e)[:200])}
    if rank == 0:
        bad = emit(out, a)
    if (world > 1 or a.pg_of_one) and dist.is_initialized():
        dist.destroy_process_group()
    if bad:
        sys.exit(bad)


def stamp(what, t0=[None]):
    """progress + where the run's wall time goes, on stderr (the driver's clock sees all of it)"""
    now = time.perf_counter()
    if t0[0] is None:
        t0[0] = now
    sys.stderr.write('[bench %7.1f s] %s\n' % (now - t0[0], what))
    sys.stderr.flush()


def step_line(a, rank, world, device, cvig_fov, ops):
    """The headline line. Order: the headline measurement, then its cpu_baseline, and only then -- in a CHILD
    process whose failure cannot lose either -- the blocks of the other single-GPU BASELINE configs. stdout gets the compact line
    (<= LINE_BUDGET bytes), the full record goes to --detail-out and stderr."""
    stamp('start (%s %s %s B=%d, %d rank(s))' % (a.model, a.mode, a.precision, a.batch, world))
    sb = StepBench(a.model, a.mode, a.precision, a.batch, a.fov, rank, world, device, a.graph).run(a.steps, a.warmup)
    out = sb.line()
    stamp('headline: %.1f pairs/s, %.3f ms per step' % (sb.value, sb.ms))
    out['collectives'] = collectives_info(a, rank, world, device, sb.phases, sb.ms)          # every rank takes part; rank 0 prints
    grouped = world > 1 or a.pg_of_one
    headline = a.mode == 'infer' and a.precision == 'fp32' and not a.graph
    cpu_args = (sb.ground_raw[:a.cpu_pairs].cpu(), sb.ov_raw[:a.cpu_pairs].cpu(), sb.wts, sb.semantic) if headline and rank == 0 else None
    phases = sb.phases
    del sb
    torch.cuda.empty_cache()
    if grouped and a.mode == 'infer' and not a.graph and not a.no_train_step:
        # BASELINE configs[2] is a DDP TRAINING config (model/cvig_fov.py:444-461; nn.DataParallel, model/cvig_baseline.py:339-343): the
        # N > 1 line therefore also runs the training step in the same process group -- the overhead-embedding all-gather, the
        # reduce-scatter of their gradients and the two asynchronous bucket all-reduces inside a real step, not only standalone
        k = max(1, min(a.steps, 5))
        tb = StepBench(a.model, 'train', a.precision, a.batch, a.fov, rank, world, device).run(k, max(1, min(a.warmup, 2)))
        out['train_step'], out['train_step_detail'] = train_step_entry(a, tb, rank, world, device)
        phases = tb.phases
        stamp('train step in the same group: %.1f pairs/s, %.3f ms per step' % (tb.value, tb.ms))
        del tb
        torch.cuda.empty_cache()
    if grouped and not a.no_microbench:
        out['collectives']['microbench'] = collectives_microbench(a, rank, world, device, phases)
        stamp('collectives microbench')
    out['guards'] = guards_block(ops)
    side = world == 1 and headline and not a.no_side_blocks and a.model == 'fov' and a.fov == 360 and a.batch == 128
    if headline and not a.no_cpu_baseline:
        def cpu_leg():
            if rank == 0:
                out['cpu_baseline'] = cpu_baseline(a, *cpu_args)
                stamp('cpu_baseline: %.2f pairs/s on %d threads' % (out['cpu_baseline']['value'], out['cpu_baseline']['cores']))
        if world > 1:
            out['_after_group'] = cpu_leg      # N > 1: rank 0 times the CPU port once the process group is gone (every rank takes this
        else:                                  # branch, so that all of them leave the group together and none waits on rank 0)
            cpu_leg()
    if side:
        out['_side_full'] = sides_child(a)       # this process is idle on the GPU meanwhile (its memory is released)
        stamp('side blocks done')
    return out


def train_step_entry(a, tb, rank, world, device):
    """The training step measured inside the N > 1 line -> (compact entry for the stdout line, full block for the detail record)"""
    allp = [tb.phases]
    if world > 1:
        allp = [None] * world
        dist.all_gather_object(allp, tb.phases)
    mx = {n: round(max(p.get(n, 0.0) for p in allp), 3) for n in tb.phases}
    inflight = sum(v for k, v in tb.phases.items() if k.endswith('_all_reduce_issue_to_joined'))
    stall = tb.phases.get('reducer_wait_stall', 0.0)
    r = tb.roofline
    entry = {'baseline_config': 'configs[2]: DDP training step, %d pairs per GPU, global batch %d' % (tb.B, tb.B * world),
             'value': round(tb.value, 2), 'unit': 'pairs/s', 'ms_per_step': round(tb.ms, 3), 'steps': tb.steps, 'warmup': tb.warmup,
             'dtype': tb.dtype(), 'loss': float(tb.loss.item()),
             'per_phase_ms_max_over_ranks': mx,
             'bucket_inflight_ms': round(inflight, 4), 'reducer_wait_stall_ms': round(stall, 4), 'overlap_hidden_ms': round(inflight - stall, 4),
             'frac': r.get('frac'), 'whole_step_frac': r.get('whole_step_frac')}
    full = dict(tb.line(), per_phase_ms={'rank0': tb.phases, 'max_over_ranks': mx, 'every_rank': allp})
    return entry, full


SIDE_ORDER = ('train_step_fp32', 'config4_semantic_bf16', 'config4_semantic_bf16_train', 'train_step_bf16', 'config1_baseline', 'config5_retrieval', 'config5_retrieval_direct',
              'fp32_grade_on_fp16_mfma', 'hbm_kernels', 'batch_sweep', 'e2e_data_path', 'e2e_data_path_bf16', 'e2e_data_path_bf16_device_entropy_all', 'e2e_data_path_bf16_device_entropy')


def side_blocks(a, rank, world, device, cvig_fov, ops):
    """`--mode sides`: one block per other single-GPU BASELINE config (config 1 cvig_baseline, the config-2 training step in fp32 and
    bf16, config 4 cvig_semantic on the bf16 MFMA, config 5 retrieval), the fp16x3 / HBM-kernel / batch-size / data-path blocks --
    each measured here with its dominant kernel's live roofline. Run by the headline as a child; its stdout line is the full dict
    (it is NOT bound by LINE_BUDGET: --detail-out takes it), and the dict is re-written to --detail-out after every block, so a
    block that dies costs only itself."""
    out = {}
    k = max(2, min(a.steps, 5))

    def done(name, blk):
        out[name] = blk
        stamp('side block %s' % name)
        if a.detail_out:
            tmp = a.detail_out + '.tmp'
            with open(tmp, 'w') as f:
                json.dump(out, f)
            os.replace(tmp, a.detail_out)

    only = [n for n in os.environ.get('WITW_SIDES_ONLY', '').split(',') if n]      # diagnostic: run only these side blocks

    def guarded(name, fn):
        if only and name not in only:
            return
        try:
            done(name, fn())
        except Exception as e:       # a failed block is reported as such; the others still run
            import traceback
            traceback.print_exc(file=sys.stderr)
            done(name, {'error': '%s: %s' % (type(e).__name__, str(e)[:300])})
        torch.cuda.empty_cache()

    def train_fp32():
        t = StepBench('fov', 'train', 'fp32', a.batch, a.fov, rank, world, device).run(k, 2)
        return t.block('configs[1] shape (cvig_fov bs=128, fp32), the training step of model/cvig_fov.py:444-461',
                       'every trainable gradient within 1e-4 of its norm of the reference run with reconciled ReLU gates, '
                       'Adam update within 1e-3 lr (tests/test_backward_gpu.py::test_training_step_matches_reference_golden)')

    def train_bf16():
        t = StepBench('fov', 'train', 'bf16', a.batch, a.fov, rank, world, device).run(k, 2)
        return t.block('configs[1] shape on the configs[3] arithmetic: the cvig_fov training step with bf16 MFMA forward / dgrad / wgrad, '
                       'fp32 accumulate, fp32 master weights and Adam', 'tests/test_bf16_train_gpu.py (wgrad vs autograd, step vs the fp32 path)')

    def semantic_bf16():
        s = StepBench('semantic', 'infer', 'bf16', a.batch, a.fov, rank, world, device).run(k, 2)
        blk = s.block('configs[3]: cvig_semantic, bf16 MFMA, 1 GPU', None)
        with torch.no_grad():       # accuracy of this very step against the exact-fp32 kernels on the same inputs
            surface, polar = s.preprocess(s.ground_raw, s.ov_raw)
            su_b, ov_b = s.embed(surface, polar)
            su_f, ov_f = s.embed(surface, polar, 'fp32')
            rel = max(float((su_b - su_f).norm() / su_f.norm()), float((ov_b - ov_f).norm() / ov_f.norm()))
            _l, r32, _o, _d = cvig_fov.evaluate_global_batch(ov_f, su_f, 0)
        blk['parity'] = {'embedding_rel_l2_vs_f32_kernels': rel, 'stated_tolerance': 5e-2,
                         'top1_pct_f32_kernels': float((r32 <= 1).float().mean().item() * 100),
                         'test': 'tests/test_bf16_gpu.py (vs CPU emulation of bf16 storage 1e-2, vs fp32 reference goldens 5e-2 of the norm)'}
        return blk

    def fp32_blocks():
        sb = StepBench('fov', 'infer', 'fp32', a.batch, a.fov, rank, world, device).run(2, 1)
        done('fp32_grade_on_fp16_mfma', fp16x3_block(sb, k))
        return hbm_block(sb)

    def e2e(extra):
        e = e2e_child(a, extra)
        if 'error' in e:
            return e
        keys = ('metric', 'value', 'unit', 'dtype', 'jpeg_decode', 'pcie_bytes_per_pair', 'staging', 'steady_state_pairs_per_s', 'pipeline_fill_s', 'stage_pairs_per_s',
                'limiting_stage', 'overlap_efficiency_steady_state', 'overlap_efficiency_raw', 'gpu_stage_serialised_pairs_per_s',
                'host_decode_pairs_per_s_per_core', 'host_decode_scaling')
        blk = {kk: e[kk] for kk in keys if kk in e}
        blk['workload'] = (e.get('config') or {}).get('workload')
        return blk

    def semantic_bf16_train():
        t = StepBench('semantic', 'train', 'bf16', a.batch, a.fov, rank, world, device).run(k, 2)
        return t.block('configs[3] training: cvig_semantic (layer 0 trains, model/cvig_semantic.py:301-309) on the bf16 MFMA -- forward, data '
                       'gradient through all 13 layers, weight gradients of layers 0 and 17-27, fp32 master weights and Adam',
                       'tests/test_bf16_train_gpu.py (semantic cases: vs the fp32 step; fused first-two-layers forward bitwise the unfused one)')

    guarded('train_step_fp32', train_fp32)
    guarded('config4_semantic_bf16', semantic_bf16)
    guarded('config4_semantic_bf16_train', semantic_bf16_train)
    guarded('train_step_bf16', train_bf16)
    cpu_leg = [None]

    def baseline():      # its CPU leg runs at the very end, behind the data-path blocks (their loaders compete for the same cores)
        if a.no_cpu_baseline:
            return baseline_bench(a, device, full=False)
        blk, cpu_leg[0] = baseline_bench(a, device, full=True, defer_cpu=True)
        return blk
    guarded('config1_baseline', baseline)
    guarded('config5_retrieval', lambda: retrieval_block(device, cvig_fov, ops, 125000, 10000, 10, 'dft'))
    guarded('config5_retrieval_direct', lambda: retrieval_block(device, cvig_fov, ops, 125000, 1024, 10, 'direct'))
    guarded('hbm_kernels', fp32_blocks)
    guarded('batch_sweep', lambda: batch_sweep(a, rank, world, device, ops, compact=True))
    # The data-path blocks run as CHILD processes (`bench.py --mode e2e ...`): 16 loader workers forked from a process that has built
    # every other block measured a fifth slower with every stage on its own unchanged; a driver's train() / test() is a fresh process.
    # fp32 encoders: the GPU is the limiting stage; bf16 encoders (configs[3] arithmetic) need 8x the images per second
    import shutil
    import tempfile
    jpegs = tempfile.mkdtemp(prefix='witw_e2e_')          # ONE synthetic data set for both blocks; the host-decode scaling sweep once
    try:
        # (the files carry no restart markers, as Pillow / libjpeg write them. fp32 block: the library's default data path -- Huffman
        # decoding on the GPU, self-synchronising; bf16: the host-Huffman path with 16 workers, then the device path with FOUR)
        guarded('e2e_data_path', lambda: e2e(['--e2e-pairs', '2048', '--e2e-dir', jpegs, '--no-decode-scaling']))
        guarded('e2e_data_path_bf16', lambda: e2e(['--e2e-pairs', '8192', '--workers', '16', '--precision', 'bf16', '--e2e-dir', jpegs,
                                                   '--device-entropy', 'off', '--decode-scaling-seconds', '0.5']))
        guarded('e2e_data_path_bf16_device_entropy_all', lambda: e2e(['--e2e-pairs', '8192', '--workers', '4', '--precision', 'bf16', '--e2e-dir', jpegs,
                                                                      '--device-entropy', 'all']))
    finally:
        shutil.rmtree(jpegs, ignore_errors=True)
    # the same bf16 pass on files that carry restart markers (one GPU thread per restart interval instead of the self-synchronising
    # decode): FOUR workers (marker scan + packing) feed it -- the data path no longer scales with the host's cores
    jpegs = tempfile.mkdtemp(prefix='witw_e2e_rst_')
    try:
        guarded('e2e_data_path_bf16_device_entropy', lambda: e2e(['--e2e-pairs', '8192', '--workers', '4', '--precision', 'bf16', '--e2e-dir', jpegs,
                                                                  '--jpeg-restart-blocks', '2']))
    finally:
        shutil.rmtree(jpegs, ignore_errors=True)
    if cpu_leg[0] is not None:
        try:
            cpu_leg[0]()                         # fills config1_baseline's cpu_baseline + parity in place
            done('config1_baseline', out['config1_baseline'])
        except Exception as e:
            out['config1_baseline']['cpu_baseline'] = {'error': str(e)[:200]}
    return out


def child_env():
    """environment of a bench child: no launcher variables, and no profiler pre-load (a child under `rocprofv3 -- python3 bench.py`
    would otherwise be measured with the tool attached and write into the parent's output directory)"""
    drop = ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT', 'LD_PRELOAD')
    return {k: v for k, v in os.environ.items() if k not in drop and not k.startswith('ROCP') and not k.startswith('ROCPROF')}


def run_child(cmd, timeout):
    """cmd in its own session; on timeout the whole process group (loader workers, pools) is killed. -> (rc or None, stdout, stderr tail)"""
    import signal
    import subprocess
    p = subprocess.Popen(cmd, env=child_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        so, se = p.communicate(timeout=timeout)
        return p.returncode, so, se
    except subprocess.TimeoutExpired:
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except OSError:
            pass
        so, se = p.communicate()
        return None, so, se


def sides_child(a):
    """`bench.py --mode sides` as a child -> {block: dict}; whatever it had finished when it failed or ran out of time is kept"""
    import tempfile
    fd, path = tempfile.mkstemp(prefix='witw_sides_', suffix='.json')
    os.close(fd)
    cmd = [sys.executable, os.path.abspath(__file__), '--mode', 'sides', '--batch', str(a.batch), '--fov', str(a.fov), '--steps', str(a.steps),
           '--detail-out', path] + (['--no-cpu-baseline'] if a.no_cpu_baseline else [])
    rc, so, se = run_child(cmd, 600)
    sys.stderr.write(se)
    blocks = {}
    try:
        blocks = json.load(open(path))
    except Exception:
        pass
    for f in (path, path + '.tmp'):
        if os.path.exists(f):
            os.remove(f)
    if rc != 0:
        blocks['_child'] = {'error': 'timeout' if rc is None else 'exit status %d' % rc, 'stderr_tail': se[-400:]}
    return blocks


def short(s, n=120):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + '~'


def compact_side(name, b):
    """one side block -> the few keys the line carries: value, ms_per_step, dtype, the dominant kernel and its roofline fraction"""
    if not isinstance(b, dict):
        return None
    if 'error' in b:
        return {'error': short(b['error'], 80)}
    c = {}
    for k in ('value', 'unit', 'ms_per_step', 'dtype'):
        if k in b:
            c[k] = b[k]
    r = b.get('roofline')
    if isinstance(r, dict):
        c['kernel'] = short(r.get('kernel', ''), 48)
        c['frac'] = r.get('frac')
        if r.get('frac_rocprof') is not None:
            c['frac_rocprof'] = r['frac_rocprof']
        if r.get('bound') and r['bound'] != 'mfma':
            c['bound'] = r['bound']
    if name == 'config1_baseline':
        if isinstance(b.get('train_step'), dict):
            c['train_ms_per_step'] = b['train_step'].get('ms_per_step')
        if isinstance(b.get('cpu_baseline'), dict):
            c['cpu_pairs_per_s'] = b['cpu_baseline'].get('value')
            c['cpu_cores'] = b['cpu_baseline'].get('cores')
    if (name.startswith('train_step') or name.endswith('_train')) and isinstance(r, dict) and 'wgrad_bf16_tflops_incl_layout_passes' in r:
        c['wgrad_tflops'] = r['wgrad_bf16_tflops_incl_layout_passes']
    if name == 'hbm_kernels':
        c = {'unit': 'frac of 8 TB/s', 'kernels': {short(k.split(' ')[0], 40): v['frac_of_hbm_peak'] for k, v in b.get('kernels', {}).items()}}
    if name == 'batch_sweep':
        c = {'unit': 'pairs/s', 'points': {'%s_B%d' % (p_['precision'], p_['pairs_per_gpu']):
                                           [p_.get('plain_one_stream_eager', {}).get('value', p_['value']),
                                            p_['value'] if 'plain_one_stream_eager' in p_ else None,
                                            p_.get('graph_replay', {}).get('value')] for p_ in b.get('points', [])},
             'what': '[plain step: one stream, eager; PairEmbedder (what test() runs at B <= 64: two streams, bf16 pair as a hipGraph) or null; '
                     'whole step as one hipGraph or null]'}
    if name.startswith('e2e'):
        for k in ('steady_state_pairs_per_s', 'overlap_efficiency_steady_state'):
            if k in b:
                c[k] = b[k]
        if 'limiting_stage' in b:
            c['limiting_stage'] = short(b['limiting_stage'], 40)
    if name == 'fp32_grade_on_fp16_mfma':
        c['max_abs_embedding_diff_vs_f32_kernels'] = b.get('max_abs_embedding_diff_vs_f32_kernels')
    if name == 'config5_retrieval' and isinstance(b.get('index_exact'), dict):
        c['rescored_per_million'] = b['index_exact'].get('rescored_per_million')
    return c


def compact_line(full):
    """The stdout line: headline keys in full, everything verbose trimmed, one compact entry per side config."""
    line = {}
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data'):
        if k in full:
            line[k] = full[k] if k != 'dtype' else short(full[k], 40)
    if isinstance(full.get('config'), dict):
        line['config'] = {k: (short(v, 200) if isinstance(v, str) else v) for k, v in full['config'].items()}
    for k in ('queries_per_sec', 'recall', 'loss', 'index_exact'):
        if k in full:
            line[k] = full[k]
    r = full.get('roofline')
    if isinstance(r, dict):
        line['roofline'] = {k: (short(r[k], 100) if isinstance(r[k], str) else r[k]) for k in
                            ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'frac_events', 'frac_rocprof', 'rocprof_avg_launch_ms', 'traffic',
                             'launches', 'avg_launch_ms', 'avg_launch_gflop', 'all_conv_launches_tflops', 'whole_step_frac',
                             'wgrad_bf16_tflops_incl_layout_passes') if k in r}
        ts = r.get('traffic_source')
        if isinstance(ts, dict):      # `traffic` is NOT measured in this run: the committed PMC passes, tied to the kernel sources' hashes
            line['roofline']['traffic_source'] = {'file': ts.get('file'), 'stale': ts.get('stale'),
                                                  'what': 'lookup: committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, not this run'}
    c = full.get('cpu_baseline')
    if isinstance(c, dict):
        line['cpu_baseline'] = {k: (short(c[k], 160) if isinstance(c[k], str) else c[k]) for k in
                                ('value', 'unit', 'cores', 'kind', 'cpu', 'passes', 'pass_pairs_per_s', 'train_pairs_per_s', 'rank_ms_per_query',
                                 'sample') if k in c}
    g = full.get('guards')
    if isinstance(g, dict):
        line['guards'] = {k: g[k] for k in ('tripped', 'forced_by_env', 'bf16_16x16x32_kernel_on', 'bf16_weight_resident_kernel_on') if k in g}
    c = full.get('collectives')
    if isinstance(c, dict):
        cc = {k: c[k] for k in ('backend', 'world', 'rccl_version', 'all_reduce_of_ones', 'distinct_devices', 'microbench') if k in c}
        cc['ranks_seen'] = len(c.get('ranks_seen', []))
        cc['devices'] = ['%d:%s' % (d['rank'], d.get('pci_bus_id') or d.get('local_device')) for d in c.get('devices', [])]
        pp = c.get('per_phase_ms')
        if isinstance(pp, dict):
            cc['per_phase_ms_max_over_ranks'] = pp.get('max_over_ranks')
        line['collectives'] = cc
    sides = full.get('_side_full')
    if isinstance(sides, dict):
        line['side'] = {}
        for name in list(SIDE_ORDER) + [k for k in sides if k not in SIDE_ORDER]:
            if name in sides:
                cs = compact_side(name, sides[name])
                if cs is not None:
                    line['side'][name] = cs
    if isinstance(full.get('points'), list):        # --mode sweep: one row per point, the per-kernel tables stay in the detail record
        line['points'] = [[p_['precision'], p_['pairs_per_gpu'], p_['value'], p_.get('all_conv_launches_frac'), p_.get('all_conv_frac_vs_B128'),
                           p_.get('graph_replay', {}).get('value')] for p_ in full['points']]
        line['points_columns'] = ['precision', 'pairs_per_gpu', 'pairs_per_s', 'all_conv_launches_frac', 'all_conv_frac_vs_B128', 'graph_replay_pairs_per_s']
    for k in ('what', 'stage_pairs_per_s', 'limiting_stage', 'steady_state_pairs_per_s', 'overlap_efficiency_steady_state',
              'jpeg_decode', 'train_step', 'parity', 'error'):          # the smaller modes' own keys (sweep, e2e, baseline)
        if k in full and k not in line:
            line[k] = full[k] if not isinstance(full[k], str) else short(full[k], 200)
    return line


def emit(full, a):
    """Rank 0: full record -> --detail-out (default bench_detail.json beside bench.py) and stderr; compact line (< LINE_BUDGET bytes, or
    optional parts are shed until it is) -> stdout, exactly one line. -> exit status to raise (None = fine)."""
    path = a.detail_out or os.path.join(ROOT, 'bench_detail.json')
    if a.mode == 'sides':                     # the child of a headline run: its parent compacts; the file was written block by block
        print(json.dumps(full), flush=True)
        return None
    rec = dict(full)
    sides = rec.pop('_side_full', None)
    if sides:
        rec.update(sides)
    blob = json.dumps(rec)
    try:
        with open(path, 'w') as f:
            f.write(blob + '\n')
    except OSError as e:
        sys.stderr.write('bench: could not write %s (%s)\n' % (path, e))
        path = None
    sys.stderr.write('BENCH_DETAIL ' + blob + '\n')
    sys.stderr.flush()
    line = compact_line(full)
    line['detail'] = (os.path.relpath(path, ROOT) if path else None)
    text = json.dumps(line, separators=(',', ':'))
    for shed in ('side.batch_sweep', 'side.hbm_kernels', 'collectives.per_phase_ms_max_over_ranks', 'side', 'collectives.devices',
                 'cpu_baseline.sample', 'train_step.baseline_config', 'config.workload', 'train_step.per_phase_ms_max_over_ranks'):
        if len(text) < LINE_BUDGET:
            break
        d, keys = line, shed.split('.')
        for k in keys[:-1]:
            d = d.get(k, {})
        if keys[-1] in d:
            del d[keys[-1]]
            line.setdefault('shed_to_fit', []).append(shed)
        text = json.dumps(line, separators=(',', ':'))
    if len(text) >= LINE_BUDGET:
        sys.stderr.write('bench: line of %d bytes exceeds the budget of %d\n' % (len(text), LINE_BUDGET))
    print(text, flush=True)
    c = full.get('collectives') or {}
    if c.get('world', 1) > 1:
        # an N-GPU line that did not really run on N ranks / N devices is refused (exit status 3), after it has been printed
        if c.get('all_reduce_of_ones') != float(c['world']) or len(c.get('ranks_seen', [])) != c['world']:
            sys.stderr.write('bench: the process group did not count %d ranks\n' % c['world'])
            return 3
        if not a.single_device and c.get('distinct_devices') != c['world']:
            sys.stderr.write('bench: %d ranks on %s distinct devices\n' % (c['world'], c.get('distinct_devices')))
            return 3
    return None


def batch_sweep(a, rank, world, device, ops, compact=False):
    """The reference's own default operating points (train / test batch_size 64: model/cvig_fov.py:385,490; cvig_semantic 32:
    model/cvig_semantic.py:416; cvig_baseline 16) beside the B = 128 the headline is quoted on: the cvig_fov eval step at B = 16, 32, 64,
    128 in fp32 and bf16, eager and -- where the step is launch-bound -- replayed as one hipGraph. Per point: pairs/s, and for the
    conv kernel instantiation that takes the most time at that batch its FLOP/s against the MFMA peak (HIP events around every
    launch, by kernel name), so that a kernel-selection cliff shows as a drop of `frac` against B = 128. compact (the default run's
    side block): no per-kernel table; one synthetic batch of 128 pairs is built once and its first B pairs are every point's input."""
    out = {'what': 'cvig_fov fov=360 eval step (the headline workload) at the reference\'s default batch sizes', 'points': []}
    base = None
    for precision in ('fp32', 'bf16'):
        peak = PEAK_BF16_MFMA_TFLOPS if precision == 'bf16' else PEAK_F32_MFMA_TFLOPS
        for B in (128, 64, 32, 16):
            # the plain step (one stream, eager) gives the per-kernel table: HIP events around every launch, by instantiation
            sb = StepBench('fov', 'infer', precision, B, a.fov, rank, world, device, share=base, pair=False)
            base = base or sb
            k = 5 if precision == 'fp32' else 10
            # (a timed window of 10-20 ms at the small batches: one host hiccup of 60 ms read as a fifth of the rate in two committed
            # records, at a different point each time -- below B = 128 every figure is the better of two windows)
            reps = 1 if B >= 128 else 2
            byk = None
            for _rep in range(reps):
                ops.PROFILE_BY_KERNEL = {}
                before = getattr(sb, 'ms', None)
                sb.run(k, 2)
                got, ops.PROFILE_BY_KERNEL = ops.PROFILE_BY_KERNEL, None
                if before is None or sb.ms <= before:
                    byk, best = got, (sb.value, sb.ms, sb.roofline)
            sb.value, sb.ms, sb.roofline = best
            agg = {n: (sum(f for f, _, _ in v), sum(e0.elapsed_time(e1) for _, e0, e1 in v), len(v)) for n, v in byk.items()}
            top = max(agg, key=lambda n: agg[n][1])
            conv_ms = sum(v[1] for v in agg.values()) / k
            pt = {'precision': precision, 'pairs_per_gpu': B, 'value': round(sb.value, 1), 'unit': 'pairs/s', 'ms_per_step': round(sb.ms, 3),
                  'dominant_kernel': top, 'dominant_kernel_tflops': round(agg[top][0] / (agg[top][1] * 1e-3) / 1e12, 1),
                  'dominant_kernel_frac': round(agg[top][0] / (agg[top][1] * 1e-3) / 1e12 / peak, 4),
                  'dominant_kernel_share_of_conv_time': round(agg[top][1] / max(1e-9, sum(v[1] for v in agg.values())), 3),
                  'all_conv_launches_tflops': sb.roofline['all_conv_launches_tflops'],
                  'all_conv_launches_frac': round(sb.roofline['all_conv_launches_tflops'] / peak, 4),
                  'conv_launches_ms_per_step': round(conv_ms, 3)}
            if not compact:
                pt['kernels'] = {n: {'launches_per_step': agg[n][2] // k, 'ms_per_step': round(agg[n][1] / k, 4),
                                     'tflops': round(agg[n][0] / (agg[n][1] * 1e-3) / 1e12, 1)} for n in sorted(agg, key=lambda n: -agg[n][1])}
            if B <= 64:
                # what test() / the validation phase of train() run at this batch (cvig_fov.PairEmbedder: the two encoders on two
                # streams, the bf16 pair as one hipGraph): THIS is `value`; the plain step stays listed beside it
                dr = StepBench('fov', 'infer', precision, B, a.fov, rank, world, device, share=base, pair=True).run(k, 3)
                first = (dr.value, dr.ms)
                dr.run(k, 1)
                if first[1] < dr.ms:
                    dr.value, dr.ms = first
                pt['plain_one_stream_eager'] = {'value': pt['value'], 'ms_per_step': pt['ms_per_step']}
                pt['value'], pt['ms_per_step'] = round(dr.value, 1), round(dr.ms, 3)
                pt['pair_embedder'] = dict(dr.pair.stats)
                del dr
            if sb.ms - conv_ms > 0.15 * sb.ms:        # a sixth of the step is not conv kernels: launch gaps matter -> one hipGraph
                g = StepBench('fov', 'infer', precision, B, a.fov, rank, world, device, graph=True, share=base).run(k, 2)
                first = (g.value, g.ms)
                g.run(k, 1)
                if first[1] < g.ms:
                    g.value, g.ms = first
                pt['graph_replay'] = {'value': round(g.value, 1), 'ms_per_step': round(g.ms, 3)}
                del g
            out['points'].append(pt)
            if sb is not base:
                del sb
            torch.cuda.empty_cache()
    ref = {p['precision']: p for p in out['points'] if p['pairs_per_gpu'] == 128}
    for p in out['points']:
        p['all_conv_frac_vs_B128'] = round(p['all_conv_launches_frac'] / max(1e-9, ref[p['precision']]['all_conv_launches_frac']), 3)
    out['points'].sort(key=lambda p: (p['precision'] != 'fp32', p['pairs_per_gpu']))
    return out


def hbm_block(sb):
    """The HBM-bound kernels of the step, timed live with events on the stream they run on (20 launches each): algorithmic
    bytes (read + written once) / duration against the HBM peak. They are 2 % of the step; the dominant kernel's roofline is
    the MFMA one above."""
    ops = sb.ops
    B = sb.B
    with torch.no_grad():
        ov = ops.resize_bilinear(sb.ov_raw, (256, 256), sb.mean, sb.std, sb.ndiv)
        polar = ops.polar_transform(ov)
        packed = sb.oe._pack_first(False)

        def timed(fn, n=20):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / n * 1e-3
        c = polar.shape[1]
        rows = {
            'polar_from_raw_kernel (overhead side: resize 512 -> 256 + normalise + polar transform, one launch: what the step runs)': (
                timed(lambda: ops.polar_from_raw(sb.ov_raw, mean=sb.mean, std=sb.std, n_div255=sb.ndiv)),
                sb.ov_raw.numel() * 4 + polar.numel() * 4, 'raw %d x %d x 512 x 512 fp32 in, %d x %d x 128 x 512 out; the 256 x 256 image is never written' % (B, c, B, c)),
            'resize_bilinear_norm_kernel (ground side 224 -> 128 x 512)': (timed(lambda: ops.resize_bilinear(sb.ground_raw, (128, sb.ws), sb.mean, sb.std, sb.ndiv)),
                                                                          sb.ground_raw.numel() * 4 + B * c * 128 * sb.ws * 4, 'raw 224 x 224 in, 128 x %d out' % sb.ws),
            'resize_then_polar_two_launches (the former overhead side: resize_bilinear_norm_kernel 512 -> 256, then polar_kernel)': (
                timed(lambda: ops.polar_transform(ops.resize_bilinear(sb.ov_raw, (256, 256), sb.mean, sb.std, sb.ndiv))),
                sb.ov_raw.numel() * 4 + polar.numel() * 4, 'same algorithmic bytes (raw in, polar out) as the fused launch'),
            'conv3x3_first_persist_kernel (3 -> 64 channels, NCHW in, NHWC out; round 4: two persistent workgroups per CU, the next tile prefetched)': (timed(lambda: ops.conv3x3_first_fwd(polar, packed, circular=True, relu=True)),
                                                                           polar.numel() * 4 + B * 128 * 512 * 64 * 4, '2.1 GB written per launch'),
        }
    return {'note': 'algorithmic bytes / HIP-event duration, 20 launches each, peak %.0f GB/s' % PEAK_HBM_GBS,
            'kernels': {k: {'ms': round(t * 1e3, 4), 'GBps': round(nb / t / 1e9, 1), 'frac_of_hbm_peak': round(nb / t / 1e9 / PEAK_HBM_GBS, 3), 'what': w}
                        for k, (t, nb, w) in rows.items()}}


def fp16x3_block(sb, steps):
    """The same step with fp32-grade products from fp16 hi/lo pairs on the fp16 MFMA (--precision fp16x3), reported beside the
    headline, never as `value`: same inputs, same weights, embeddings compared with the exact-fp32 kernels'."""
    ops, cvig_fov = sb.ops, sb.cvig_fov

    def step3():
        with torch.no_grad():
            surface, polar = sb.preprocess(sb.ground_raw, sb.ov_raw)
            su3, ov3 = sb.embed(surface, polar, 'fp16x3')
            return cvig_fov.evaluate_global_batch(ov3, su3, 0) + (su3, ov3, surface, polar)
    for _ in range(2):
        r3 = step3()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    for _ in range(steps):
        r3 = step3()
    torch.cuda.synchronize()
    dt3 = (time.perf_counter() - t3) / steps
    with torch.no_grad():
        su32, ov32 = sb.se(r3[6]), sb.oe(r3[7])
    diff = max(float((r3[4] - su32).abs().max()), float((r3[5] - ov32).abs().max()))
    return {'precision': 'fp16x3', 'value': round(sb.B / dt3, 2), 'unit': 'pairs/s', 'ms_per_step': round(dt3 * 1e3, 3),
            'max_abs_embedding_diff_vs_f32_kernels': diff, 'loss': float(r3[0].item()),
            'recall': {'top1_pct': float((r3[1] <= 1).float().mean().item() * 100), 'top5_pct': float((r3[1] <= 5).float().mean().item() * 100)},
            'ranks_differing_from_f32_step': int((r3[1] != sb.ranks).sum().item()), 'overflow': bool(ops.f16x3_overflowed(sb.device)),
            'note': 'operands carried as fp16 hi + lo, products hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_f16, fp32 accumulate; '
                    'held to the reference goldens at the same 1e-4 as the f32 kernels (tests/test_f16x3_gpu.py)'}


def baseline_bench(a, device, full=True, B=32, defer_cpu=False):
    """BASELINE config 1: cvig_baseline, 32 pairs, ground 500x500 (SurfaceResize('witw'), model/cvig_baseline.py:219-221) /
    overhead 512x512, eval step = 2 encoders (7 x [Conv2d(4,2) -> LeakyReLU -> BatchNorm2d], 3 GeM pools) -> exhaustive
    minibatch triplet loss -> Euclidean rank counts (:228-315, :454-466). BASELINE calls this config CPU plumbing, so the
    block carries its own cpu_baseline: the oracle on the same 32 pairs."""
    from witw_amd import cvig_baseline as cb, ops, synth
    seed = 4242
    xs = torch.from_numpy(synth.images_u8(seed, 1, (B, 3, 500, 500))).to(device)
    xo = torch.from_numpy(synth.images_u8(seed, 2, (B, 3, 512, 512))).to(device)
    # planted matches: the ground image is the (resized) overhead plus noise, re-quantised to 0..255
    xs = (torch.nn.functional.interpolate(xo, size=(500, 500), mode='bilinear', align_corners=False) * 0.7 + xs * 0.3).round().contiguous()
    prm_s, prm_o = synth.baseline_params(seed), synth.baseline_params(seed + 1)
    se, oe = cb.SurfaceEncoder().to(device).eval(), cb.OverheadEncoder().to(device).eval()
    for enc, prm in ((se, prm_s), (oe, prm_o)):
        with torch.no_grad():
            for i, q in enumerate(prm, 1):
                getattr(enc, 'conv%d' % i).weight.copy_(torch.from_numpy(q['w']))
                getattr(enc, 'conv%d' % i).bias.copy_(torch.from_numpy(q['b']))
                bn = getattr(enc, 'bn%d' % i)
                bn.weight.copy_(torch.from_numpy(q['gamma']))
                bn.bias.copy_(torch.from_numpy(q['beta']))
                bn.running_mean.copy_(torch.from_numpy(q['mean']))
                bn.running_var.copy_(torch.from_numpy(q['var']))

    def step():
        with torch.no_grad():
            es, eo = se(xs), oe(xo)
            loss = cb.exhaustive_minibatch_triplet_loss(es, eo)
            D = ops.pairwise_sqdist(eo, es, take_sqrt=True)
            return loss, ops.rank_count(D, 0), es, eo
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ops.PROFILE = []
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        loss, ranks, es, eo = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    prof, ops.PROFILE = ops.PROFILE, None
    conv = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v[0] not in ('match', 'match_dft')]
    conv_ms = sum(m for _, m in conv)
    tf = sum(f for f, _ in conv) / (conv_ms * 1e-3) / 1e12 if conv else 0.0
    out = {'baseline_config': 'configs[0]: cvig_baseline, 32 pairs (BASELINE runs it CPU-only; here the same step on the GPU beside the CPU port)',
           'metric': 'image-pairs/sec (embedding+similarity)',
           'workload': 'cvig_baseline eval: 32 pairs, ground 3x500x500 + overhead 3x512x512 -> 2 encoders (7 conv4x4/2 + LeakyReLU + '
                       'BatchNorm, 3 GeM pools) -> exhaustive triplet loss -> Euclidean ranks', 'pairs_per_gpu': B,
           'value': round(B / ms * 1e3, 2), 'unit': 'pairs/s', 'ms_per_step': round(ms, 3), 'steps': n, 'dtype': 'f32', 'loss': float(loss.item()),
           'n_gpus': 1, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'data': 'synthetic',
           'recall': {'top1_pct': float((ranks <= 1).float().mean().item() * 100)},
           'roofline': {'bound': 'mfma', 'kernel': 'conv3x3_nhwc_f32_kernel<...,TAPS=4> (Conv2d(4,2,0) as space-to-depth(2) + the 2x2 live taps) x 12 + conv4x4s2_first_kernel (block 1 from the raw image) x 2: all conv launches of a step',
                        'achieved': round(tf, 2), 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(tf / PEAK_F32_MFMA_TFLOPS, 4),
                        'launches': len(conv), 'avg_launch_ms': round(conv_ms / max(1, len(conv)), 4),
                        'note': 'algorithmic 2*Cin*Cout*16*Ho*Wo FLOP per launch; the conv launches take %.2f of the %.2f ms step' % (conv_ms / n, ms)}}
    # the training step of the same batch (train-mode BatchNorm, backward through both encoders, Adam), model/cvig_baseline.py:373-387
    from witw_amd import cvig_fov
    se.train()
    oe.train()
    opt = cvig_fov.Adam(list(se.parameters()) + list(oe.parameters()), lr=1e-5)

    def train_step():
        es_t, eo_t = se(xs), oe(xo)
        l = cb.exhaustive_minibatch_triplet_loss(es_t, eo_t)
        opt.zero_grad()
        l.backward()
        opt.step()
        return l
    first = float(train_step().item())
    for _ in range(2):
        train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tl = train_step()
    torch.cuda.synchronize()
    tms = (time.perf_counter() - t0) / n * 1e3
    out['train_step'] = {'ms_per_step': round(tms, 3), 'value': round(B / tms * 1e3, 2), 'unit': 'pairs/s', 'steps': n, 'loss_first_step': first,
                         'loss_last_step': float(tl.item()),       # one fixed batch of 32 random pairs is fitted within a few Adam steps (hinge -> 0)
                         'what': 'forward with batch statistics + backward of both encoders + Adam on the same 32 pairs (round 2: 49.3 ms = 650 pairs/s); '
                                 'parity: tests/test_baseline_gpu.py against tests/golden/baseline_train.npz'}
    se.eval()
    oe.eval()

    def cpu_leg():
        """the CPU port on the same 32 pairs + the in-bench parity check; defer_cpu: run by the caller at the very end of the line
        (128 OpenMP threads started in this process in front of the e2e blocks cost their loaders a fifth of their rate)"""
        from oracle import cvig_baseline_oracle as OB
        prm = [[dict((k, torch.from_numpy(np.asarray(v))) for k, v in q.items()) for q in p] for p in (prm_s, prm_o)]
        xs_c, xo_c = xs.cpu(), xo.cpu()
        es_h, eo_h, ranks_h = es.cpu(), eo.cpu(), ranks.cpu()

        def cpu_step():
            with torch.no_grad():
                es_r, eo_r = OB.encoder_forward(xs_c, prm[0]), OB.encoder_forward(xo_c, prm[1])
                return es_r, eo_r, OB.exhaustive_minibatch_triplet_loss(es_r, eo_r), OB.ranks(eo_r, es_r)
        es_r, eo_r, loss_r, ranks_r = cpu_step()
        out['cpu_baseline'] = cpu_thread_sweep(cpu_step, B, 'the same 32 pairs, full eval step')
        out['parity'] = {'max_abs_embedding_diff_vs_oracle': max(float((es_h - es_r).abs().max()), float((eo_h - eo_r).abs().max())),
                         'loss_abs_diff_vs_oracle': abs(loss_h - float(loss_r)),
                         'ranks_equal_oracle': bool(np.array_equal(ranks_h.numpy().astype(np.int64), np.asarray(ranks_r).astype(np.int64))),
                         'tolerance': 1e-4}
    loss_h = float(loss.item())
    if full and defer_cpu:
        return out, cpu_leg
    if full:
        cpu_leg()
    else:
        out['parity'] = 'embeddings 1e-4 / ranks bit-exact vs the reference goldens (tests/test_baseline_gpu.py)'
    return out


def _retrieval_data(device, G, Q, we, rank=0):
    """Gallery N(0,1) rows; query q = gallery row q (global numbering, rank-major) rolled by a per-query shift, cropped to the
    FoV, plus noise (planted matches)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(4321 + rank)
    gallery = torch.randn((G, 16, 4, 64), generator=gen, device=device)
    queries = torch.zeros((Q, 16, 4, we), device=device)
    lo, hi = rank * G, min(Q, (rank + 1) * G)
    if hi > lo:
        shifts = torch.randint(0, 64, (hi - lo,), generator=gen, device=device)
        col = (torch.arange(we, device=device)[None, :] + shifts[:, None]) % 64                      # [n, we]
        rows = gallery[lo - rank * G:hi - rank * G]
        queries[lo:hi] = torch.gather(rows, 3, col[:, None, None, :].expand(-1, 16, 4, -1)) \
            + 10.0 * torch.randn((hi - lo, 16, 4, we), generator=gen, device=device)
    return gallery, queries


def retrieval_block(device, cvig_fov, ops, G, Q, k, method):
    """BASELINE config 5's per-GPU share (G gallery rows x Q queries, ranks + top-k) measured beside the headline: one warm-up
    and one timed pass. 'dft': the spectral orientation search with the index-exact re-scoring; 'direct': the direct-sum
    kernel (524,288 FLOP per pair) on a bounded number of queries so that the block stays under a second."""
    gallery, queries = _retrieval_data(device, G, Q, 64)
    cvig_fov.retrieve(gallery, queries, k=k, method=method)
    torch.cuda.synchronize()
    ops.PROFILE = []
    t0 = time.perf_counter()
    ranks_h, vals, idx = cvig_fov.retrieve(gallery, queries, k=k, method=method)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    kind = 'match_dft' if method == 'dft' else 'match'
    m = [(fl, e0.elapsed_time(e1)) for (v, fl, e0, e1) in prof if v[0] == kind]
    tf = sum(f for f, _ in m) / (sum(t for _, t in m) * 1e-3) / 1e12 if m else 0.0
    out = {'baseline_config': 'configs[4] per-GPU share: gallery retrieval, 1M / 8 overhead embeddings per GPU',
           'workload': 'gallery retrieval: %d overhead embeddings x %d ground queries, fov=360, ranks + top-%d' % (G, Q, k),
           'match': 'dft (witw_match_fwd_dft: orientation search through 64-point row spectra, 21,120 FLOP per pair; rounding-level '
                    'decisions re-made on witw_match_pairs distances)' if method == 'dft' else
                    'direct (witw_match_fwd: 524,288 FLOP per pair)',
           'value': round(float(G) * Q / dt, 1), 'unit': 'pairs/s', 'ms_per_step': round(dt * 1e3, 3), 'steps': 1, 'dtype': 'f32',
           'recall': {'top1_pct': float(np.mean(ranks_h <= 1) * 100), 'top10_pct': float(np.mean(ranks_h <= 10) * 100), 'N': G},
           'roofline': {'bound': 'mfma', 'kernel': 'match_dft_kernel (+ norm / table kernels of the launch)' if method == 'dft' else
                        'match_kernel_w64 (+ 2 norm kernels of the launch)', 'achieved': round(tf, 2), 'peak': PEAK_F32_MFMA_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': round(tf / PEAK_F32_MFMA_TFLOPS, 4), 'launches': len(m),
                        'avg_launch_ms': round(sum(t for _, t in m) / max(1, len(m)), 3),
                        'flop_per_pair': 21120 if method == 'dft' else 524288}}
    if method == 'dft':
        out['index_exact'] = rescore_block(cvig_fov.retrieve.last_stats)
    else:
        out['parity'] = 'orientation / ranks bit-exact, distances 1e-5 vs the reference goldens (tests/test_match_gpu.py)'
    return out


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
    gallery, queries = _retrieval_data(device, G, Q, we, rank)
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
    return out


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return ''


def cpu_baseline(a, g, o, wts, semantic):
    """The oracle (CPU restatement of the reference, kind 'port') timed on this host on a bounded sample of the same workload, by the
    protocol of BASELINE.md section 4: same seeded inputs and weights as the GPU run, 1 warm-up + the MEDIAN of 3 timed passes, three legs:
      (i)   `value`: the full eval step (transforms + 2 encoders + match + loss + ranks) on the first --cpu-pairs pairs (default 32, the
            batch SURVEY's CPU anchor and BASELINE configs[0] use) of the batch the GPU step ran on;
      (ii)  `train_pairs_per_s`: the full training step (forward with Dropout2d -> match -> loss -> backward -> Adam, the loop body of
            model/cvig_fov.py:439-465 = O.train_step) on 8 of those pairs (the survey's anchor batch);
      (iii) `rank_ms_per_query`: the loop body of test() (model/cvig_fov.py:543-558 = O.ranks) against a gallery of N = 1000.
    Thread counts and what is reported: cpu_thread_sweep."""
    from oracle import cvig_fov_oracle as O
    n = g.shape[0]
    w = {k: (torch.from_numpy(v[0]), torch.from_numpy(v[1])) for k, v in wts.items()}
    norm = O.image_normalization_semantic if semantic else O.image_normalization

    def transforms(m):
        su_in, ov_in = [], []
        for i in range(m):
            s, ov = O.resize_pair(g[i], o[i], fov=a.fov, panorama=False)
            su_in.append(norm(s))
            ov_in.append(O.polar_transform(norm(ov)))
        return torch.stack(su_in), torch.stack(ov_in)

    def cpu_step(m=n):
        with torch.no_grad():
            su_in, ov_in = transforms(m)
            su = O.fov_dsm_forward(su_in, w, False)
            ov = O.fov_dsm_forward(ov_in, w, True)
            ori, d = O.match(ov, su)
            loss = O.triplet_loss(d)
            ranks = (d <= torch.diagonal(d)[None, :]).sum(0)
        return su, ov, ori, d, loss, ranks

    out = cpu_thread_sweep(cpu_step, n, '%d pairs of the same synthetic batch, full step (transforms+encoders+match+loss+ranks)' % n,
                           all_threads_fn=lambda: cpu_step(min(n, 16)), all_threads_units=min(n, 16))
    threads = out['cores']
    all_threads = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        # leg (ii): the training step on 8 pairs; fresh weight copies per pass (Adam updates them in place)
        nt = min(n, 8)
        with torch.no_grad():
            su_in, ov_in = transforms(nt)
        trainable = ((0,) + O.TRAINABLE) if semantic else None

        def train_pass():
            ws = {k: (v[0].clone(), v[1].clone()) for k, v in w.items()}
            wo = {k: (v[0].clone(), v[1].clone()) for k, v in w.items()}
            t0 = time.perf_counter()
            O.train_step(su_in, ov_in, ws, wo, trainable=trainable)
            return time.perf_counter() - t0
        train_pass()
        tt = sorted(train_pass() for _ in range(3))
        out['train_pairs_per_s'] = round(nt / tt[1], 3)
        out['train_sample'] = '%d pairs, O.train_step (fwd+match+loss+bwd+Adam), 1 warm-up + median of 3: %s s' % (nt, [round(t, 3) for t in tt])
        # leg (iii): the ranking loop body per query against N = 1000 gallery rows (synthetic N(0,1) embeddings, fov 360 -> W_e = 64)
        gen = torch.Generator().manual_seed(4321)
        we = int(a.fov / 360 * 512) // 8
        gal = torch.randn((1000, 16, 4, 64), generator=gen)
        qry = torch.randn((1000, 16, 4, we), generator=gen)
        nq = 8

        def rank_pass():
            t0 = time.perf_counter()
            with torch.no_grad():
                for idx in range(nq):            # O.ranks' loop body, model/cvig_fov.py:545-552, for the first nq queries
                    _, d = O.match(gal, qry[idx:idx + 1])
                    d = torch.squeeze(d, 1)
                    torch.sum(torch.le(d, d[idx])).item()
            return (time.perf_counter() - t0) / nq
        rank_pass()
        rt = sorted(rank_pass() for _ in range(3))
        out['rank_ms_per_query'] = round(rt[1] * 1e3, 3)
        out['rank_sample'] = 'N = 1000 gallery, %d queries per pass, 1 warm-up + median of 3: %s ms/query' % (nq, [round(t * 1e3, 2) for t in rt])
    finally:
        torch.set_num_threads(all_threads)
    return out


def cpu_thread_sweep(fn, units, sample, all_threads_fn=None, all_threads_units=None):
    """ONE convention for every cpu_baseline of the line: the CPU port is timed at 16 threads -- this box's CPU share per GPU (or every
    core torch sees, if fewer; the survey container had 8). Rounds 2-5 also timed 8 threads and all 128: 16
    won every time (r04: 5.7 / 7.5 / 3.6 pairs/s at 8 / 16 / 128; r05: 6.8 / 2.4 at 16 / 128 -- oversubscribed oneDNN convolutions).
    One warm-up call, then the MEDIAN of 3 timed calls ('1 warm-up + median of >= 3'; rounds 1-5 timed one pass and
    spread 12 % across records). all_threads_fn: one more timed call with every visible thread (reported once per run as
    `all_threads`, in the detail record) on a smaller sample."""
    all_threads = torch.get_num_threads()
    threads = min(16, all_threads)
    torch.set_num_threads(threads)
    fn()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        times.append(time.perf_counter() - t0)
    times.sort()
    value = round(units / times[1], 3)
    extra = {}
    if all_threads_fn is not None and all_threads > threads:
        torch.set_num_threads(all_threads)
        t0 = time.perf_counter()
        all_threads_fn()
        extra['all_threads'] = {'threads': all_threads, 'pairs_per_s': round((all_threads_units or units) / (time.perf_counter() - t0), 3),
                                'sample': '%d pairs, one pass, no warm-up at this thread count' % (all_threads_units or units)}
    torch.set_num_threads(all_threads)
    return {'value': value, 'unit': 'pairs/s', 'cores': threads, 'kind': 'port', 'cpu': cpu_model(), 'passes': 3,
            'pass_pairs_per_s': [round(units / t, 3) for t in times],
            'pairs_per_s_by_threads': {str(threads): value},
            'threads_convention': '%d threads (CPU share of one GPU on this box; %d visible)' % (threads, all_threads),
            'sample': sample + '; 1 warm-up + median of 3 timed passes, torch %s CPU ops (oneDNN / BLAS as built)' % torch.__version__,
            **extra}


def guards_block(ops):
    """Which hand-scheduled kernels this run used and which fell back (witw_amd/build.py's register-allocation guards, decided by
    _lib.load() after the build): a tripped guard is a 6-30 % slower bf16 path, so the line says so itself."""
    from witw_amd import _lib
    g = _lib.guards()
    return {'tripped': sorted(k for k, v in g.items() if not v['hand_scheduled_kernel']),
            'forced_by_env': sorted(k for k, v in g.items() if v['forced']),
            'detail': {k: v['detail'] for k, v in g.items() if v['detail']},
            'bf16_16x16x32_kernel_on': bool(ops.bf16_mfma16()), 'bf16_weight_resident_kernel_on': bool(_lib.load().witw_conv3x3_bf16_wres(-1)),
            'what': 's16: conv3x3_bf16_s16_kernel (else the 32x32x16 kernel); wres: conv3x3_bf16_wres_kernel for 64-input-channel layers '
                    '(else the tiled kernels); first2: conv_first2_bf16_kernel (else layers 0 and 2 as two launches)'}


def collectives_info(a, rank, world, device, phases=None, step_ms=None):
    """What the process group of this run really is, gathered from EVERY rank (so that an N-GPU line is self-evidencing):
    backend, world size, RCCL version, which ranks answered and on which device each one ran; per_phase_ms: device time per step
    of each phase of the step on rank 0 and the largest value over the ranks (HIP events on the launch stream)."""
    mine = {'rank': rank, 'local_device': int(device.index or 0), 'pid': os.getpid(),
            'device_name': torch.cuda.get_device_name(device), 'pci_bus_id': None}
    per_phase = None
    if phases is not None:
        allp = [phases]
        if world > 1:
            allp = [None] * world
            dist.all_gather_object(allp, phases)
        names = list(phases)
        per_phase = {'rank0': phases, 'max_over_ranks': {n: max(p.get(n, 0.0) for p in allp) for n in names},
                     'ms_per_step': None if step_ms is None else round(step_ms, 3),
                     'how': 'HIP events on the launch stream around each phase, summed per step and averaged over the timed steps; a blocking '
                            'collective\'s bracket includes its two stream hand-overs; grad_bucket*_issue_to_joined spans the other encoder\'s '
                            'backward (overlap window), reducer_wait_stall is what the compute stream loses to the join'}
    try:
        mine['pci_bus_id'] = torch.cuda.get_device_properties(device).pci_bus_id
    except Exception:
        pass
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return {'backend': None, 'world': 1, 'rccl_version': None, 'ranks_seen': [0], 'devices': [mine],
                'note': 'one rank: no process group, no collective on the path', **({'per_phase_ms': per_phase} if per_phase else {})}
    seen = [None] * world
    dist.all_gather_object(seen, mine)
    probe = torch.ones(1, device=device)
    dist.all_reduce(probe)                      # one real collective through the backend: must count the ranks
    ver = None
    if dist.get_backend() == 'nccl':
        try:
            ver = '.'.join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = 'unknown'
    return {'backend': dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else ''), 'world': dist.get_world_size(),
            'rccl_version': ver, 'ranks_seen': sorted(d['rank'] for d in seen), 'all_reduce_of_ones': float(probe.item()),
            'devices': sorted(seen, key=lambda d: d['rank']),
            'distinct_devices': len({(d['pci_bus_id'], d['local_device']) for d in seen}),
            **({'per_phase_ms': per_phase} if per_phase else {}),
            'per_step': 'all-gather of the overhead embeddings (2 MiB per rank at 128 pairs), all-gather of the diagonal, scalar loss '
                        'all-reduce' + ('; reduce-scatter of overhead-embedding gradients, all-reduce(SUM) of 2 x 7.24 M weight gradients'
                                        if a.mode == 'train' else '')}


XGMI_LINK_GBS = 153.0      # per point-to-point xGMI link (7 per GPU); SURVEY.md section 5 prices the collectives against it


def collectives_microbench(a, rank, world, device, phases=None, iters=10):
    """The three payloads the step really sends, each timed STANDALONE (nothing else on the device; HIP events on the launch
    stream around `iters` back-to-back blocking calls, after 3 warm-up calls; max over ranks) -> microseconds per call, algorithm
    bandwidth (payload bytes / time) and the fraction of the xGMI bound of a DIRECT (fully connected, all 7 links at once)
    schedule, SURVEY.md section 5:
      all-gather of the overhead embeddings  b x 16 KiB per rank    bound = payload_per_rank / link          (2 MiB -> ~14 us)
      reduce-scatter of their gradients      world x b x 16 KiB     bound = (total / world) / link          (16 MiB -> ~14 us)
      all-reduce (SUM) of one encoder's weight gradients, 7,236,432 fp32, x 2 encoders
                                                                     bound = 2 phases x (bytes / world) / link   (2 x 29 MB -> ~95 us)
    A ring schedule is bound by ONE link for (world - 1) hops: its figure is listed beside the direct one. In a training run
    `overlap_hidden_ms` = what the asynchronous bucket all-reduces spent in flight (issue -> joined) minus what the compute stream
    lost to the join (reducer_wait_stall): communication time that backward kernels covered. Reference semantics of the exchange:
    nn.DataParallel, model/cvig_baseline.py:339-343; global-batch normaliser model/cvig_fov.py:380."""
    from witw_amd import parallel
    E = 16 * 4 * 64
    b = a.batch
    n_grad = 7236432
    ov = torch.randn((b, E), device=device)
    gall = torch.randn((world * b, E), device=device)
    bucket = torch.randn((n_grad,), device=device)

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / iters * 1e3], device=device, dtype=torch.float64)       # us per call
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    link = XGMI_LINK_GBS * 1e9
    rows = (('all_gather_overhead_embeddings', lambda: parallel._all_gather_cat(ov), ov.numel() * 4, ov.numel() * 4 / link,
             ov.numel() * 4 * (world - 1) / link),
            ('reduce_scatter_overhead_grads', lambda: parallel.reduce_scatter_rows(gall, b), gall.numel() * 4, gall.numel() * 4 / world / link,
             gall.numel() * 4 / world * (world - 1) / link),
            ('all_reduce_weight_grads_one_encoder', lambda: dist.all_reduce(bucket, op=dist.ReduceOp.SUM), n_grad * 4,
             2 * n_grad * 4 / world / link, 2 * n_grad * 4 / world * (world - 1) / link))
    out = {}
    for name, fn, nbytes, direct_s, ring_s in rows:
        us = timed(fn)
        out[name] = {'bytes': int(nbytes), 'us': round(us, 1), 'algbw_GBps': round(nbytes / (us * 1e-6) / 1e9, 4),
                     'xgmi_direct_bound_us': round(direct_s * 1e6, 1), 'frac_of_direct_bound': round(direct_s * 1e6 / us, 6),
                     'xgmi_ring_bound_us': round(ring_s * 1e6, 1)}
    out['all_reduce_weight_grads_one_encoder']['per_step'] = 2
    if phases and 'reducer_wait_stall' in phases:
        inflight = sum(v for k, v in phases.items() if k.endswith('_all_reduce_issue_to_joined'))
        out['overlap_hidden_ms'] = round(inflight - phases['reducer_wait_stall'], 4)
    out['link_GBps'] = XGMI_LINK_GBS
    out['how'] = 'standalone, %d calls after 3 warm-ups, HIP events, max over ranks; bounds: SURVEY.md section 5' % iters
    return out


def e2e_child(a, extra):
    """`bench.py --mode e2e <extra>` as a child process on the same GPU (this process is idle meanwhile) -> its FULL record (the
    child's stdout carries only the compact line: the record comes from a --detail-out file of its own, stdout is the fallback),
    or {'error': ...}: a timeout kills the child's whole process group (loader workers, decode pool) and is reported, not dropped"""
    import tempfile
    fd, path = tempfile.mkstemp(prefix='witw_e2e_detail_', suffix='.json')
    os.close(fd)
    os.remove(path)
    cmd = [sys.executable, os.path.abspath(__file__), '--mode', 'e2e', '--batch', str(a.batch), '--fov', str(a.fov), '--detail-out', path] + list(extra)
    try:
        rc, so, se = run_child(cmd, 300)
        if rc is None:
            return {'error': 'timeout after 300 s (process group killed)'}
        try:
            return json.load(open(path))
        except (OSError, ValueError):
            pass
        for ln in so.splitlines():
            if ln.startswith('{') and '"metric"' in ln:
                return json.loads(ln)
        sys.stderr.write('e2e child failed (%d): %s\n' % (rc, se[-800:]))
        return {'error': 'exit status %d: %s' % (rc, se[-200:])}
    finally:
        if os.path.exists(path):
            os.remove(path)


def e2e_bench(a, device):
    from witw_amd import e2e
    return e2e.bench(a, device)


if __name__ == '__main__':
    main()
