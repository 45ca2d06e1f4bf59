"""Multi-GPU plumbing: one process per GPU, torch.distributed over RCCL (backend name 'nccl').

The encoder has no BatchNorm, so samples are independent and the minibatch shards across ranks
(SURVEY.md §8e). The loss couples the GLOBAL batch (nn.DataParallel semantics of the reference,
model/cvig_baseline.py:339-343; normaliser 2B(B-1) with B = global batch, model/cvig_fov.py:380):
  1. the overhead embeddings are all-gathered (2 MiB per rank at 128 pairs/rank),
  2. every rank evaluates match + loss for its column slab [B_global, b_local] (cvig_fov.sharded_match_loss);
     the diagonal, the loss partial and the row sigmoid sums are exchanged (B floats each), the
     overhead-embedding gradients are reduce-scattered to their owners (reduce_scatter_rows);
     all_gather_embeddings keeps the simpler replicated form (every rank evaluates the full matrix, backward
     = local slice),
  3. weight gradients (57.9 MB for both encoders) are summed with one all-reduce over a flat bucket.
Retrieval (gallery >> queries) shards the gallery rows instead: rank counts are summed with an
all-reduce of int32[queries].

Everything here is backend-agnostic host logic (tested on CPU with gloo, world_size 2).
"""
import torch
import torch.distributed as dist


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class PhaseTimer:
    """Per-phase device time of a step, for the bench line's `collectives.per_phase_ms`: HIP events recorded on the CURRENT
    stream in front of and behind each named phase. A blocking torch.distributed collective makes the current stream wait for
    the communication stream, so the bracket holds the collective plus its two stream hand-overs; for the asynchronous gradient
    all-reduce the bracket runs from the launch to the point where wait() has joined it (an upper bound of the collective's own
    duration: the other encoder's backward kernels run inside it) and `reducer_wait_stall` is what the compute stream really
    loses. Off (PHASES is None) everywhere but in bench.py. Every event of one timer is recorded on ONE stream -- the stream
    that was current when the timer was made -- whichever thread calls begin() / end() (an autograd hook thread's current stream
    need not be the step's; elapsed_time across streams means nothing). A timer made in a process that has no initialised GPU
    (the gloo tests of the host logic) is disabled: begin() returns None, nothing touches the device."""

    def __init__(self):
        self.rec = {}          # name -> [(start event, end event)]
        self.order = []
        self.enabled = torch.cuda.is_available() and torch.cuda.is_initialized()
        self.stream = torch.cuda.current_stream() if self.enabled else None

    def begin(self):
        if not self.enabled:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(self.stream)
        return e

    def end(self, name, e0):
        if e0 is None or not self.enabled:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(self.stream)
        if name not in self.rec:
            self.rec[name] = []
            self.order.append(name)
        self.rec[name].append((e0, e1))

    def summary(self, steps):
        """-> {phase: ms per step} (call after a device synchronize); a phase that runs k times per step is summed"""
        return {n: round(sum(a.elapsed_time(b) for a, b in self.rec[n]) / max(1, steps), 4) for n in self.order}


PHASES = None       # bench.py: parallel.PHASES = parallel.PhaseTimer() for the timed region


class phase:
    """with phase('name'): ... -- a no-op unless PHASES is set"""
    __slots__ = ('name', 'e0')

    def __init__(self, name):
        self.name, self.e0 = name, None

    def __enter__(self):
        if PHASES is not None:
            self.e0 = PHASES.begin()
        return self

    def __exit__(self, *exc):
        if self.e0 is not None and PHASES is not None:
            PHASES.end(self.name, self.e0)
        return False


def _all_gather_cat(t):
    n = world()
    t = t.contiguous()
    out = torch.empty((n * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    if dist.get_backend() == 'gloo':
        parts = [torch.empty_like(t) for _ in range(n)]
        dist.all_gather(parts, t)
        torch.cat(parts, 0, out=out)
    else:
        dist.all_gather_into_tensor(out, t)
    return out


class _AllGather(torch.autograd.Function):
    """cat over ranks; backward keeps the local slice (each rank holds the full global loss)."""

    @staticmethod
    def forward(ctx, t):
        ctx.b = t.shape[0]
        return _all_gather_cat(t)

    @staticmethod
    def backward(ctx, g):
        r = rank()
        return g[r * ctx.b:(r + 1) * ctx.b].contiguous()


def all_gather_embeddings(surface_embed, overhead_embed):
    """[b,16,4,We],[b,16,4,64] per rank -> global [B,...] tensors in rank order. Differentiable."""
    if world() == 1:
        return surface_embed, overhead_embed
    if torch.is_grad_enabled() and (surface_embed.requires_grad or overhead_embed.requires_grad):
        return _AllGather.apply(surface_embed), _AllGather.apply(overhead_embed)
    return _all_gather_cat(surface_embed), _all_gather_cat(overhead_embed)


def all_reduce_grads(params):
    """SUM (not mean: the loss is already normalised by the global batch) of every existing .grad
    through ONE flat bucket; parameters without a gradient (frozen layers) are skipped on all ranks
    alike. Returns the number of floats reduced."""
    grads = [p.grad for p in params if p.grad is not None]
    if world() == 1 or not grads:
        return sum(g.numel() for g in grads)
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off:off + n].view_as(g))
        off += n
    return off


class GradBucket:
    """The gradients of one module in ONE persistent flat fp32 buffer: every trainable parameter's .grad is a view into it for
    the life of the bucket, so the all-reduce runs on the buffer itself -- no torch.cat into a staging tensor, no copy back.
    zero() (called through the optimizer's zero_grad) clears the buffer and marks it `fresh`; a backward that computes all of
    the module's weight gradients itself (cvig_fov._EncoderFn) then WRITES them straight into the views and calls notify(),
    any other backward goes through autograd, which accumulates into the views in place."""

    def __init__(self, params, on_ready=None):
        self.params = list(params)
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()
            self.views.append(v)
            p.grad = v
            p._witw_grad_view = v
            p._witw_bucket = self
        self.fresh = True
        self.on_ready = on_ready
        self.arrived = 0            # gradients accumulated through autograd since zero() / the last reduction
        self.nodes = 0              # encoder autograd nodes built since zero() (cvig_fov._EncoderFn.forward counts them)
        self.touched = set()        # id() of the parameters whose gradient was written since zero()

    def zero(self):
        self.flat.zero_()
        for p, v in zip(self.params, self.views):
            if p.grad is not v:
                p.grad = v
        self.fresh = True
        self.arrived = 0
        self.nodes = 0
        self.touched = set()

    def direct(self):
        """May a backward WRITE the gradients into the views (and hand the bucket to the all-reduce at once)? Only when it is
        the one autograd node of this module in the step: a second node (the encoder called twice before backward) has to
        accumulate, and the reduction must wait for both."""
        return self.fresh and self.nodes <= 1

    def notify(self):
        """All gradients of the bucket have been written in place (no autograd accumulation, hence no grad hooks)."""
        self.fresh = False
        self.touched = {id(p) for p in self.params}
        if self.on_ready is not None:
            self.on_ready()

    def received(self, p):
        """Did p get a gradient since zero()? (Adam skips the others, as torch skips parameters whose .grad is None.)"""
        return id(p) in self.touched

    def release(self):
        for p in self.params:
            p.grad = None
            for a in ('_witw_grad_view', '_witw_bucket'):
                if hasattr(p, a):
                    delattr(p, a)


class OverlappedGradReducer:
    """Gradient all-reduce (SUM) overlapped with the backward: one GradBucket per module. The backward of an encoder
    is ONE autograd node (cvig_fov._EncoderFn), so all of its weight gradients appear together -- written by the wgrad
    kernels directly into the bucket -- and that encoder's all-reduce is launched asynchronously the moment the node has
    run: RCCL moves the 29 MB bucket over xGMI while the OTHER encoder's backward kernels run. Modules whose gradients come
    through autograd's accumulation (in place, into the same views) launch from a post-accumulate-grad hook on their last
    parameter instead. wait() (before optimizer.step()) joins the collectives and launches buckets whose gradients never all
    arrived. With one rank nothing is sent. all_reduce_grads() is the same reduction without the overlap or the bucket."""

    def __init__(self, modules):
        self.modules = list(modules)
        self.buckets = []
        self.inflight = {}
        self._hooks = []
        for bi, m in enumerate(self.modules):
            b = GradBucket([p for p in m.parameters() if p.requires_grad], on_ready=self._ready(bi))
            self.buckets.append(b)
            m._grad_bucket = b
            for p in b.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._hook(bi)))

    def _ready(self, bi):
        def fn():
            self.buckets[bi].arrived = len(self.buckets[bi].params)
            self._launch(bi)
        return fn

    def _hook(self, bi):
        def fn(p):
            b = self.buckets[bi]
            b.arrived += 1
            b.touched.add(id(p))
            b.fresh = False
            # autograd sums a leaf's gradient over the nodes of ONE backward() before it accumulates (one round of hooks), but a
            # module used by two nodes may also be driven by two backward() calls: the reduction of such a bucket waits for wait()
            if b.nodes <= 1 and b.arrived == len(b.params):
                self._launch(bi)
        return fn

    def _launch(self, bi):
        if world() == 1 or bi in self.inflight or not self.buckets[bi].params:
            return
        e0 = PHASES.begin() if PHASES is not None and self.buckets[bi].flat.is_cuda else None
        self.inflight[bi] = (dist.all_reduce(self.buckets[bi].flat, op=dist.ReduceOp.SUM, async_op=True), e0)

    def wait(self):
        """-> number of floats reduced."""
        for bi in range(len(self.buckets)):
            if self.buckets[bi].arrived > 0:    # a bucket some of whose gradients never came (unused parameters) goes now
                self._launch(bi)
        n = 0
        with phase('reducer_wait_stall'):
            for bi, (work, e0) in sorted(self.inflight.items()):
                work.wait()
                if e0 is not None and PHASES is not None:
                    PHASES.end('grad_bucket%d_all_reduce_issue_to_joined' % bi, e0)
                n += self.buckets[bi].flat.numel()
        self.inflight = {}
        for b in self.buckets:
            b.arrived = 0
        return n

    def close(self):
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for m, b in zip(self.modules, self.buckets):
            b.release()
            if getattr(m, '_grad_bucket', None) is b:
                del m._grad_bucket


class CapturedStep:
    """A launch-bound step as ONE hipGraph launch. fn(*inputs) -> tensor or tuple of tensors must be a pure sequence
    of device work on the current stream (the C-ABI launchers and torch allocations qualify; no host synchronisation,
    no collectives, weights already packed). The step is run `warmup` times eagerly (caches, LUTs, allocator), then
    captured with torch.cuda.CUDAGraph (hipStreamBeginCapture / hipGraphInstantiate on ROCm) on static copies of the
    inputs; __call__ copies new inputs into those buffers and replays. At 8 pairs the bf16 inference step issues ~60
    kernels of 5-30 us each and is bound by their launches; the graph removes that bound."""

    def __init__(self, fn, example_inputs, warmup=2, capture_error_mode='global'):
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):               # capture needs a non-default stream; warm up on it too
            for _ in range(warmup):
                fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # capture_error_mode 'thread_local': HIP calls of OTHER host threads (loader / pinning / event-reaper threads of a driver)
        # do not invalidate the capture; 'global' (torch's default) makes any such call an error
        with torch.cuda.graph(self.graph, capture_error_mode=capture_error_mode):
            out = fn(*self.static_in)
        self.static_out = out

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src)
        self.graph.replay()
        return self.static_out


def all_gather_ragged(t):
    """cat over ranks of tensors whose first dimension differs from rank to rank (shard_range splits)."""
    if world() == 1:
        return t
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = _all_gather_cat(n).tolist()
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    allp = _all_gather_cat(pad)
    return torch.cat([allp[i * m:i * m + k] for i, k in enumerate(sizes)], dim=0)


def reduce_scatter_rows(t, rows):
    """SUM t [world*rows, ...] over the ranks and keep this rank's block of `rows` rows."""
    if world() == 1:
        return t
    t = t.contiguous()
    r = rank()
    if dist.get_backend() == 'gloo':
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return t[r * rows:(r + 1) * rows].contiguous()
    out = torch.empty((rows,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.reduce_scatter_tensor(out, t, op=dist.ReduceOp.SUM)
    return out


def broadcast_parameters(modules, src=0):
    """Make every rank start from rank `src`'s weights."""
    if world() == 1:
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src)


def shard_range(n, r=None, n_ranks=None):
    """Contiguous [begin, end) share of n items for rank r (first n % ranks shares are one longer)."""
    r = rank() if r is None else r
    n_ranks = world() if n_ranks is None else n_ranks
    q, rem = divmod(n, n_ranks)
    begin = r * q + min(r, rem)
    return begin, begin + q + (1 if r < rem else 0)


def all_reduce_sum_(t):
    if world() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t
