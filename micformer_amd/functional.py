"""torch.autograd.Function wrappers: one per fused unit of the hot path; forward and backward are sequences of C-ABI
launches (micformer_amd.ops).  Activations are channels-last (B, D, H, W, C) fp32; parameters keep the reference's
state_dict layout.  Parameter gradients are accumulated by the kernels into zero-filled buffers and handed to autograd.
"""
import torch

from . import ops
from . import _lib
from ._lib import block_region

_zl = torch.zeros_like


def _in_block(fn):
    """Decorator: the launches of this forward / backward are transformer-block kernels (profiler region tag)."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with block_region():
            return fn(*a, **k)
    return wrapped


def effective_window(dims, window):
    """get_window_size (MS.py:135-145): a window dim clamps to the volume dim when dim <= window."""
    return tuple(d if d <= w else w for d, w in zip(dims, window))


def _padded(dims, ws):
    return tuple(d + (-d) % w for d, w in zip(dims, ws))


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


# ----------------------------------------------------------------------------- the launch plan's state: ONE object per step owner
class StepContext:
    """Everything the launch plan of a step keeps between its launches -- the switches an engine sets for its own forward +
    backward, the queues of parameter-gradient work that left the data-gradient chain, and the mailboxes modules use to hand a
    tensor to the NEXT launch (keyed on data_ptr(): valid inside one step of one owner only).  Round 5: these were ~20
    module-level globals; two engines in a process, or a validation forward between an engine's forward and backward, shared them.
    Now every TrainEngine owns one StepContext and installs it for the duration of its forward + backward (`use_context`, which
    refuses re-entry); engine-less forwards / backwards run on the module's default context.

      switches    defer_wgrad / defer_calls  queue linear weight gradients / other parameter-gradient closures instead of launching
                  flush_points, flush_budget launch what is queued at stage boundaries (how many of them may: data-parallel step)
                  lazy_ln_ok                 cross pairs may park their LayerNorm-1 backward for the self pair's launch
                  segmenter                  the StepSegmenter capturing this step as a sequence of graphs, or None
      queues      deferred, deferred_ln, deferred_calls, queued_dw, pending_flush (flush batches set aside until the main chain
                  has launched its next kernel: LAZY_FLUSH), wside_used (devices whose side stream must be joined)
      hooks       entry_hooks / entry_seen (forward: work parked until the n-th stage entry), backward_hooks (per stage module)
      mailboxes   loss_mail (target in, fused loss out), cross_after_self (the self pair's outputs BasicLayer is about to hand to
                  the cross pair), next_ln / next_ln_out (the cross pair's LayerNorm 1 as the self pair launch's epilogue), lazy_ln (parked LayerNorm-1 backward records), skip_tokens (ConvDownFn inputs of THIS forward),
                  carry (step_many: the batches carried to the next step's head)
    """

    def __init__(self):
        self.defer_wgrad = False
        self.defer_calls = False
        self.flush_points = False
        self.flush_budget = 1 << 30
        self.lazy_ln_ok = False
        self.segmenter = None
        self.deferred = []                 # (dy, a, dw, db, dp_scale, rows_per_sample) of nn.Linear weight gradients
        self.deferred_ln = []              # (partials, blocks, C, dgamma, dbeta) of LayerNorm backward calls
        self.deferred_calls = []           # (callable, tensors it reads, queued from inside a transformer block?)
        self.queued_dw = set()             # destinations already in the queue: the grouped kernel owns each dW element exclusively
        self.pending_flush = []            # (event, device, linear items, LayerNorm items, closures)
        self.wside_used = set()
        self.entry_hooks = []              # [callable, the stage entry it waits for]
        self.entry_seen = 0                # stage entries seen since clear_entry_hooks()
        self.backward_hooks = {}           # id(stage module) -> callable run when the backward has left that stage
        self.loss_mail = {"target": None, "result": None}
        self.cross_after_self = None       # (data_ptr, data_ptr)
        self.next_ln = None                # [(gamma, beta)] * 2: the cross pair's norm1, for the self pair launch about to be issued
        self.next_ln_out = None            # ((data_ptr, data_ptr) of the self pair's outputs, [(xn, mean, rstd)] * 2, hid | None)
        self.lazy_ln = {}                  # data_ptr of the partial-sum buffer handed to autograd -> the parked record
        self.skip_tokens = {}              # data_ptr of a ConvDownFn input of THIS forward -> token
        self.carry = {"on": False, "open": False, "stash": []}


CTX = StepContext()                        # the context in force (module default: engine-less forwards and backwards)
_DEFAULT_CTX = CTX
_CTX_OWNER = [None]


class use_context:
    """`with use_context(ctx):` -- ctx is the launch plan's state until the block exits.  Not re-entrant: a second owner inside
    the block (another engine's step, a nested step of the same engine) would cross the mailboxes."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        global CTX
        if _CTX_OWNER[0] is not None:
            raise RuntimeError("a StepContext is already in force: steps of two owners must not nest")
        _CTX_OWNER[0], CTX = self.ctx, self.ctx
        return self.ctx

    def __exit__(self, *exc):
        global CTX
        _CTX_OWNER[0], CTX = None, _DEFAULT_CTX
        return False


# ----------------------------------------------------------------------------- deferred weight gradients (engine mode)
# Nothing downstream of a backward pass consumes the parameter gradients until the optimiser step.  In engine mode (the
# kernels accumulate straight into the flat gradient buffer) the weight gradients of the linear layers are therefore not
# launched where autograd reaches them: the (output-gradient, input) pair is queued and TrainEngine flushes the queue once
# after backward through micf_linear_bwd_weight_grouped -- a few chip-filling launches instead of two small ones per
# layer, and the data-gradient chain (the critical path) gets shorter.  The queue holds references, so the caching
# allocator cannot hand the queued buffers to anyone else before the flush.
DEFER_MAX_TOKENS = 1 << 17          # larger layers launch immediately (their operands are still hot in L2 / MALL)



# Any other parameter-gradient work (conv weight gradients, the head tail's decomposition, bias column sums ...) can leave the
# data-gradient chain the same way: single-GPU engine mode queues it as a closure that the next flush launches (on the side
# stream at a flush point).  Only work that accumulates into the engine's flat gradient buffer qualifies: autograd must not be
# waiting for a returned tensor.


def _defer(ok, fn, *tensors):
    """Run `fn` now, or -- engine mode, every destination a view of the flat gradient buffer (`ok`) -- at the next flush."""
    if ok and CTX.defer_calls and CTX.defer_wgrad:
        launch_pending_flush()
        CTX.deferred_calls.append((fn, tuple(t for t in tensors if t is not None), _lib.BLOCK_DEPTH > 0))
    else:
        fn()


# Module-level drop-in (no engine: autograd waits for the returned gradient tensors, nothing can be deferred past the Function):
# the nn.Linear weight gradients and LayerNorm partial finishes of ONE pair backward are still independent of each other, so they are
# collected while the Function runs and launched as one grouped call each before it returns (10 + 4-6 C-ABI calls -> 2 per pair:
# the eager loop is host-bound, bench.py `stock_loop`).
_LOCAL = None


class _local_batch:
    def __enter__(self):
        global _LOCAL
        self.prev, _LOCAL = _LOCAL, ([], [])
        return self

    def __exit__(self, et, ev, tb):
        global _LOCAL
        items, ln = _LOCAL
        _LOCAL = self.prev
        if et is None:
            if ln:
                ops.layernorm_bwd_finish(ln)
            if items:
                ops.linear_bwd_weight_grouped(items)
        return False


def _batched(fn):
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **k):
        with _local_batch():
            return fn(*a, **k)
    return wrapped


def _lin_wgrad(defer, dy, a, dw, db, dp_scale=None, rows_per_sample=0):
    if _LOCAL is not None and not (defer and CTX.defer_wgrad) and dy.dtype == a.dtype and ops.wgrad_groupable(dy, a, dp_scale, rows_per_sample) \
            and all(it[2].data_ptr() != dw.data_ptr() for it in _LOCAL[0]):
        _LOCAL[0].append((dy, a, dw, db, dp_scale, rows_per_sample))
        return
    if dy.dtype == torch.bfloat16 or a.dtype == torch.bfloat16:
        # operands the fused block kernels stored as bf16: only the grouped entry point reads them (shapes it does not take -- fewer
        # than 32 tokens -- are widened and go the fp32 way)
        if dy.dtype != a.dtype or not ops.wgrad_groupable(dy, a, dp_scale, rows_per_sample):
            dy, a = dy.float(), a.float()
        elif not (defer and CTX.defer_wgrad and dw.data_ptr() not in CTX.queued_dw):
            ops.linear_bwd_weight_grouped([(dy, a, dw, db, dp_scale, rows_per_sample)])
            return
    if defer and CTX.defer_wgrad and dy.shape[0] <= DEFER_MAX_TOKENS and ops.wgrad_groupable(dy, a, dp_scale, rows_per_sample) \
            and dw.data_ptr() not in CTX.queued_dw:          # (a layer applied twice in one step launches its second use at once)
        launch_pending_flush()
        CTX.queued_dw.add(dw.data_ptr())
        CTX.deferred.append((dy, a, dw, db, dp_scale, rows_per_sample))
    elif dy.dtype == torch.bfloat16:
        # bf16 operands of a layer too long to queue (> DEFER_MAX_TOKENS rows: batch >= 5 at 128^3): the per-layer entry point is
        # fp32 only, the grouped one reads bf16 pairs -- launch it now for this one layer
        ops.linear_bwd_weight_grouped([(dy, a, dw, db, dp_scale, rows_per_sample)])
    else:
        ops.linear_bwd_weight(dy, a, dw, db, dp_scale=dp_scale, rows_per_sample=rows_per_sample)


def take_deferred():
    """Hand the queued (not yet launched) weight gradients and LayerNorm partials to the caller (TrainEngine's data-parallel step)."""
    launch_pending_flush()
    if CTX.deferred_calls:
        raise RuntimeError("deferred gradient closures are a single-GPU engine feature (CTX.defer_calls) and cannot be planned")
    items, ln = list(CTX.deferred), list(CTX.deferred_ln)
    CTX.deferred.clear()
    CTX.deferred_ln.clear()
    CTX.queued_dw.clear()
    return items, ln


def drop_deferred():
    """Forget anything still queued (TrainEngine calls this at the start of a step: entries left behind by a backward that
    raised must not be flushed into the next step's gradients)."""
    CTX.deferred.clear()
    CTX.deferred_ln.clear()
    CTX.deferred_calls.clear()
    CTX.queued_dw.clear()
    CTX.pending_flush.clear()
    CTX.lazy_ln.clear()
    CTX.carry["on"], CTX.carry["open"] = False, False
    CTX.carry["stash"].clear()


def _ln_defer(on):
    """The list LayerNorm backward should queue its parameter-gradient partials in, or None (accumulate immediately)."""
    return CTX.deferred_ln if (on and CTX.defer_wgrad) else None


# Flush points (engine mode, single GPU): an identity node at every stage boundary whose BACKWARD launches the weight
# gradients queued so far on a side stream, so they run under the rest of the backward chain (which is latency-bound, not
# throughput-bound, since the blocks were fused) instead of in a tail after it.  join_wgrad_stream() re-joins before Adam.
                                    # allows the first few only: the rest stays queued for the launches that overlap the all-reduce)
FLUSH_MAX_TOKENS = 1 << 30   # (measured: flushing at every point wins, 19.6 vs 20.2 ms small stages only)
_WSIDE = {}


def _wgrad_stream(device):
    st = _WSIDE.get(device)
    if st is None:
        st = _WSIDE[device] = torch.cuda.Stream(device=device)
    return st


# A flush point forks the side stream off the main chain.  Captured into a HIP graph with the side branch created first, the
# main chain's next kernel moved to another hardware queue and started 130-190 us late (rocprofv3 kernel trace of the 8^3
# stage's backward).  So the flush is launched lazily: the point only records an event on the main stream and sets the batch
# aside; the batch goes to the side stream (waiting for that event) once the main chain has launched its next kernel -- i.e.
# when the next entry is queued.  Measured 13.9 -> 13.4 ms per step (LAZY_FLUSH = False restores the eager order).
LAZY_FLUSH = True


# Segmented capture (TrainEngine(segmented=True)): instead of ONE HIP graph whose executor decides on which hardware queue the
# side branch runs (round 2: 12.5 ms when it happens to overlap, 15.8 ms when it serialises, and the deciding factor was node
# creation order), the step is captured as a SEQUENCE of graphs -- the main chain is cut at every flush point, every side batch
# is its own graph -- and replayed on two streams with explicit events: main segments back to back on the launch stream, side
# batch k on the weight-gradient stream after main segment k.  Each graph is a plain chain (plus short fork / joins), so nothing
# is left for the executor to place.  Memory: main segments share one private pool, side segments another (graphs of one pool
# replay in capture order on one stream); tensors a side batch reads were allocated by main segments and are kept referenced
# until the capture is complete, so no later main segment can be handed their blocks.
SEG_SERIAL = __import__("os").environ.get("MICF_SEG_SERIAL", "0") == "1"      # debug: side segments on the main stream too
SEG_SKIP_SIDE = __import__("os").environ.get("MICF_SEG_SKIP_SIDE", "0") == "1"  # measurement: the main chain alone (WRONG results)


class StepSegmenter:
    def __init__(self, side_stream):
        self.side = side_stream
        self.pool_m, self.pool_s = torch.cuda.graph_pool_handle(), torch.cuda.graph_pool_handle()
        self.segments = []              # ("main" | "side", CUDAGraph) | ("join", None)
        self.keep = []
        self.cur = None

    def begin(self):
        """Start the next main segment on the CURRENT stream (relaxed mode: autograd ends / begins captures from its own thread)."""
        self.cur = torch.cuda.CUDAGraph()
        self.cur.capture_begin(pool=self.pool_m, capture_error_mode="relaxed")
        self._mark = _lib.LAUNCHES[0]

    def _end_main(self):
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")         # ("The CUDA Graph is empty": two cuts with nothing between them)
            self.cur.capture_end()
        if _lib.LAUNCHES[0] != self._mark:          # (an empty segment is dropped, not replayed)
            self.segments.append(("main", self.cur))
        self.cur = None

    def run_side(self, work, keep=None):
        """Cut the main chain here; capture `work` (callables) as one graph on the side stream; resume the main chain."""
        self._end_main()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.stream(self.side):
            g.capture_begin(pool=self.pool_s, capture_error_mode="relaxed")
            for w in work:
                w()
            g.capture_end()
        self.segments.append(("side", g))
        self.keep.append(keep)
        self.begin()

    def run_side_groups(self, works, keep=None):
        """Cut the main chain here; capture every entry of `works` (a list of callables each) as its OWN graph on the side stream,
        replayed back to back there with an event behind each; -> the events (TrainEngine.step_many: the stage groups of the
        carried parameter work, each awaited by the main chain only where the forward first reads that group's parameters)."""
        self._end_main()
        evs = []
        for k, work in enumerate(works):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.stream(self.side):
                g.capture_begin(pool=self.pool_s, capture_error_mode="relaxed")
                for w in work:
                    w()
                g.capture_end()
            ev = torch.cuda.Event()
            self.segments.append(("side_ev", (g, ev, k == 0)))
            evs.append(ev)
        self.keep.append(keep)
        self.begin()
        return evs

    def wait(self, ev):
        """Cut the main chain here; it continues once `ev` (of run_side_groups) has been recorded in this replay."""
        self._end_main()
        self.segments.append(("wait", ev))
        self.begin()

    def join(self):
        self._end_main()
        self.segments.append(("join", None))
        self.begin()

    def finish(self):
        self._end_main()
        self.keep.clear()               # (the pools keep every block a captured graph uses; the Python references can go)

    def replay(self):
        main = torch.cuda.current_stream()
        for kind, g in self.segments:
            if kind in ("side", "side_ev") and SEG_SKIP_SIDE:
                continue
            if kind == "side_ev":
                graph, ev, first = g
                if SEG_SERIAL:
                    graph.replay()
                    continue
                if first:
                    self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    graph.replay()
                    ev.record(self.side)
            elif kind == "wait":
                if not (SEG_SERIAL or SEG_SKIP_SIDE):
                    main.wait_event(g)
            elif kind == "main" or (kind == "side" and SEG_SERIAL):
                g.replay()
            elif kind == "side":
                self.side.wait_stream(main)
                with torch.cuda.stream(self.side):
                    g.replay()
            else:
                main.wait_stream(self.side)


def launch_pending_flush():
    while CTX.pending_flush:
        ev, dev, items, ln, calls = CTX.pending_flush.pop(0)
        side = _wgrad_stream(dev)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            _launch_batch(items, ln, calls)
        CTX.wside_used.add(dev)


def flush_wgrad_side(calls_only=False, lazy=False, extra=None):
    """Launch everything queued so far on the weight-gradient side stream (ordered after the current stream's work).
    calls_only: just the deferred closures (data-parallel mode: the grouped linear weight gradients stay queued for the
    engine, which interleaves them with the gradient all-reduce after the replay).  lazy: see LAZY_FLUSH.
    extra: callables launched behind the batch on the same stream (segmented capture: the early optimiser step)."""
    if CTX.segmenter is not None:
        items, ln = ([], []) if calls_only else (list(CTX.deferred), list(CTX.deferred_ln))
        calls = list(CTX.deferred_calls)
        if not calls_only:
            CTX.deferred.clear()
            CTX.deferred_ln.clear()
            CTX.queued_dw.clear()
        CTX.deferred_calls.clear()
        work = [lambda: _launch_batch(items, ln, calls)] if (items or ln or calls) else []
        work += list(extra or [])
        if work:
            CTX.segmenter.run_side(work, keep=(items, ln, calls))
        return
    launch_pending_flush()
    if not ((not calls_only and (CTX.deferred or CTX.deferred_ln)) or CTX.deferred_calls):
        return
    first = CTX.deferred_calls[0][1][0] if CTX.deferred_calls else (CTX.deferred[0][0] if CTX.deferred else CTX.deferred_ln[0][0])
    dev = first.device
    main, side = torch.cuda.current_stream(dev), _wgrad_stream(dev)
    if lazy and LAZY_FLUSH:
        ev = torch.cuda.Event()
        ev.record(main)
        items, ln = ([], []) if calls_only else (list(CTX.deferred), list(CTX.deferred_ln))
        calls = list(CTX.deferred_calls)
        for it in items:
            for t in (it[0], it[1], it[4]):
                if t is not None:
                    t.record_stream(side)
        for it in ln:
            it[0].record_stream(side)
        for _, tensors, _blk in calls:
            for t in tensors:
                t.record_stream(side)
        if not calls_only:
            CTX.deferred.clear()
            CTX.deferred_ln.clear()
            CTX.queued_dw.clear()
        CTX.deferred_calls.clear()
        CTX.pending_flush.append((ev, dev, items, ln, calls))
        return
    side.wait_stream(main)
    if not calls_only:
        for it in CTX.deferred:                # the queue's references die at the flush: tell the allocator who still reads them
            for t in (it[0], it[1], it[4]):
                if t is not None:
                    t.record_stream(side)
        for it in CTX.deferred_ln:
            it[0].record_stream(side)
    for _, tensors, _blk in CTX.deferred_calls:
        for t in tensors:
            t.record_stream(side)
    with torch.cuda.stream(side):
        flush_wgrad(calls_only)
    CTX.wside_used.add(dev)


def join_wgrad_stream():
    if CTX.segmenter is not None:
        CTX.segmenter.join()
        return
    launch_pending_flush()
    for dev in list(CTX.wside_used):
        torch.cuda.current_stream(dev).wait_stream(_wgrad_stream(dev))
    CTX.wside_used.clear()


# forward side of a stage entry: the engine may park a callable here that runs when the forward reaches its n-th stage (work
# only the backward needs -- zeroing the gradient buffer, transposed shadow weights -- is launched on a side stream under the
# latency-bound small stages instead of in front of the step)


def clear_entry_hooks():
    CTX.entry_hooks.clear()
    CTX.entry_seen = 0


def park_entry_hook(fn, at, front=False):
    """front: runs before the hooks already parked for the same entry."""
    if front:
        CTX.entry_hooks.insert(0, [fn, at])
    else:
        CTX.entry_hooks.append([fn, at])


def run_entry_hook(force=False):
    """Called at every stage entry: runs the parked callables whose entry this is (or all of them now, `force`)."""
    CTX.entry_seen += 1
    due = [h for h in CTX.entry_hooks if force or CTX.entry_seen >= h[1]]
    for h in due:
        CTX.entry_hooks.remove(h)
        h[0]()


# backward side: callables the engine parks per stage (key = id of the stage module); run when the backward has left that stage

# Carry (TrainEngine.step_many: several steps in ONE captured graph).  The parameter-gradient batches of the flush points the backward
# reaches FIRST -- decoder, head, last encoder stage: exactly the gradients of the flat buffers' tail [cut, total) -- are not
# launched under the backward (where they only queue up in front of the encoder's batches and push those into a tail behind the
# step) but set aside; the engine launches them, and the Adam update of that tail, on the side stream at the START of the next
# step of the same graph, under its encoder forward -- which does not read those parameters and leaves half the chip idle in its
# 8^3 stage.  "open": the backward is still inside that first region (the engine's hook at the last encoder stage closes it).


def _carry_stash(key):
    """Set aside everything queued so far (linear items, LayerNorm partials, closures) for the next step's head.  key: id of the
    stage module whose backward this flush point closes (None: a flush point inside a stage)."""
    launch_pending_flush()
    batch = (list(CTX.deferred), list(CTX.deferred_ln), list(CTX.deferred_calls))
    CTX.deferred.clear()
    CTX.deferred_ln.clear()
    CTX.deferred_calls.clear()
    CTX.queued_dw.clear()
    CTX.carry["stash"].append((key, batch))


def launch_carried(batches):
    """The engine, at the head of the next step (current stream = the weight-gradient side stream)."""
    side = torch.cuda.current_stream()
    for items, ln, calls in batches:
        # (the batch's references die here: tell the allocator the side stream still reads the tensors)
        for it in items:
            for t in (it[0], it[1], it[4]):
                if t is not None:
                    t.record_stream(side)
        for it in ln:
            it[0].record_stream(side)
        for _, tensors, _blk in calls:
            for t in tensors:
                t.record_stream(side)
        _launch_batch(items, ln, calls)


class FlushPointFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, xa, key=None):
        ctx.key = key
        return x.view_as(x), xa.view_as(xa)

    @staticmethod
    def backward(ctx, dx, dxa):
        # (a token cap exists for experiments: restricting the flushes to the latency-bound small stages was measured slower --
        # the weight gradients of the big stages then pile up in the tail)
        full = CTX.flush_points and CTX.defer_wgrad and dx.numel() // dx.shape[-1] <= FLUSH_MAX_TOKENS and CTX.flush_budget > 0
        hook = CTX.backward_hooks.pop(ctx.key, None) if ctx.key is not None else None
        if CTX.carry["on"] and CTX.carry["open"] and full:
            _carry_stash(ctx.key)                       # (launched at the head of the next step: see CTX.carry)
            if hook is not None:
                hook()
            return dx, dxa, None
        if CTX.segmenter is not None:                       # the hook (early Adam) rides in the same side segment as the batch
            if full or (CTX.defer_calls and CTX.defer_wgrad) or hook is not None:
                if full:
                    CTX.flush_budget -= 1
                flush_wgrad_side(calls_only=not full, extra=[hook] if hook is not None else None)
            return dx, dxa, None
        if full:
            CTX.flush_budget -= 1
            flush_wgrad_side(lazy=True)
        elif CTX.defer_calls and CTX.defer_wgrad:
            flush_wgrad_side(calls_only=True, lazy=True)
        if hook is not None:
            hook()
        return dx, dxa, None


GROUP_CONV_WGRAD = True


def _defer_conv_wgrad(ok, dhid, xn, xa, dw, db, dims):
    """Offset-conv weight gradient: now, or queued like _defer -- the flush then launches all queued layers of one shape (both
    modalities of every cross pair since the last flush) as ONE micf_conv3_bwd_weight_grouped call."""
    def fn():
        ops.conv3_bwd_weight(dhid, xn, dw, db, dims, x2=xa)
    fn.conv = (dhid, xn, xa, dw, db, tuple(dims))
    _defer(ok, fn, dhid, xn, xa)


def _launch_batch(items, ln, calls):
    # (parameter-gradient side work: it may be launched lazily from inside a block Function, but belongs to no 8(d) unit of the chain)
    unit, _lib.UNIT = _lib.UNIT, None
    try:
        _launch_batch_body(items, ln, calls)
    finally:
        _lib.UNIT = unit


def _launch_batch_body(items, ln, calls):
    with block_region():
        if ln:
            ops.layernorm_bwd_finish(ln)
        if items:
            ops.linear_bwd_weight_grouped(items)
    convs, finishes = {}, []
    for fn, _, blk in calls:
        fz = getattr(fn, "finish", None) if GROUP_CONV_WGRAD else None
        if fz is not None:
            finishes.append(fz)
            continue
        cv = getattr(fn, "conv", None) if GROUP_CONV_WGRAD else None
        if cv is not None:
            key = (cv[5], cv[1].shape, None if cv[2] is None else cv[2].shape, cv[3].shape, cv[4] is None)
            grp = convs.setdefault(key, [])
            if all(g[3].data_ptr() != cv[3].data_ptr() for g in grp):       # (a layer applied twice: its second use runs alone)
                grp.append(cv)
                continue
        if blk:
            with block_region():
                fn()
        else:
            fn()
    if finishes:
        with block_region():
            ops.offset_head_bwd_finish_grouped(finishes)
    for grp in convs.values():
        with block_region():
            ops.conv3_bwd_weight_grouped([(dy, x1, x2, dw, db) for dy, x1, x2, dw, db, _ in grp], grp[0][5])


def flush_wgrad(calls_only=False):
    """Launch every queued weight gradient on the current stream (which must be ordered after their producers)."""
    if CTX.segmenter is not None:                           # segmented capture: one more side segment (join_wgrad_stream follows)
        flush_wgrad_side(calls_only)
        return
    launch_pending_flush()
    items, ln = [], []
    if not calls_only:
        items, ln = list(CTX.deferred), list(CTX.deferred_ln)
        CTX.deferred.clear()
        CTX.deferred_ln.clear()
        CTX.queued_dw.clear()
    calls = list(CTX.deferred_calls)
    CTX.deferred_calls.clear()
    _launch_batch(items, ln, calls)


def _targets(params):
    """Engine mode: a parameter may carry `_micf_grad`, a view of the flat gradient buffer.  The backward kernels then
    accumulate straight into it (they atomically add anyway) and autograd gets None -- no zero-fill, no `grad += g` pass."""
    return tuple(getattr(p, "_micf_grad", None) for p in params)


def _grad_buf(target, like):
    return target if target is not None else torch.zeros_like(like)


_GRAD_LAYOUTS = {}


def _grad_bufs(P, tg):
    """{name: gradient buffer} for a block's parameters: the engine's views of the flat gradient buffer where present; otherwise ONE
    zero-filled allocation carved into 16-byte-aligned views (engine-less loop: one fill per block instead of one per parameter)."""
    if all(t is None for t in tg):
        # the stock loop's case, 96 times per backward: the layout of a block's gradients is cached by its parameter shapes and
        # every view is ONE as_strided call (a slice + a view per parameter were 1.6k tensor-method calls per step)
        vals = list(P.values())
        key = tuple(v.shape for v in vals)
        lay = _GRAD_LAYOUTS.get(key)
        if lay is None:
            views, total = [], 0
            for v in vals:
                st, acc = [], 1
                for n in reversed(v.shape):
                    st.append(acc)
                    acc *= n
                views.append((tuple(v.shape), tuple(reversed(st)), total))
                total += (v.numel() + 3) // 4 * 4
            lay = _GRAD_LAYOUTS[key] = (total, views)
        flat = torch.zeros(lay[0], dtype=vals[0].dtype, device=vals[0].device)
        return {k: flat.as_strided(sh, st, o) for k, (sh, st, o) in zip(P, lay[1])}
    missing = [(k, v) for (k, v), t in zip(P.items(), tg) if t is None]
    out = {k: t for (k, _), t in zip(P.items(), tg) if t is not None}
    if missing:
        offs, total = [], 0
        for _, v in missing:
            offs.append(total)
            total += (v.numel() + 3) // 4 * 4
        flat = torch.zeros(total, dtype=missing[0][1].dtype, device=missing[0][1].device)
        for (k, v), o in zip(missing, offs):
            out[k] = flat[o:o + v.numel()].view(v.shape)
    return out


def _ret(target, buf):
    return None if target is not None else buf


# ============================================================================= both modalities as ONE tensor
# The modules between the stages (patch merging / expand, their LayerNorms, the skip linears, the final norm) are shared by the two
# modalities and per token: run once on [2B, ...] they are half the launches and need no second stream.  The pair kernels write
# the two modalities' outputs (and input gradients) as the halves of one buffer, so joining and splitting are views, not copies.
def _adjacent(a, b):
    return a.is_contiguous() and b.is_contiguous() and a.shape == b.shape and a.dtype == b.dtype and \
        a.data_ptr() + a.numel() * a.element_size() == b.data_ptr() and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()


def _joined(a, b):
    """[2B, ...] over the memory of two adjacent halves (no copy), else their concatenation."""
    if _adjacent(a, b):
        return a.new_empty(0).set_(a.untyped_storage(), a.storage_offset(), (2 * a.shape[0],) + tuple(a.shape[1:]))
    return torch.cat([a, b], 0)


class JoinFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.B = a.shape[0]
        ctx.set_materialize_grads(False)        # (SkipMailFn hands its gradient over out of band: no zero tensor, no sum)
        return _joined(_c(a.detach()), _c(b.detach()))

    @staticmethod
    def backward(ctx, g):
        if g is None:
            return None, None
        g = _c(g)
        return g[:ctx.B], g[ctx.B:]


class SplitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ab):
        ab = _c(ab.detach())
        B = ab.shape[0] // 2
        return ab[:B], ab[B:]

    @staticmethod
    def backward(ctx, ga, gb):
        return _joined(_c(ga), _c(gb))


# ============================================================================= LayerNorm
class LayerNormFn(torch.autograd.Function):
    """nn.LayerNorm over the last dim of [..., C] (optionally over cat[x, x2], MS.py:1033-1034)."""

    @staticmethod
    def forward(ctx, x, x2, gamma, beta, eps):
        x = _c(x)
        C1 = x.shape[-1]
        x2f = _c(x2).reshape(-1, x2.shape[-1]) if x2 is not None else None
        y, mean, rstd = ops.layernorm_fwd(x.reshape(-1, C1), gamma, beta, eps, x2f)
        ctx.save_for_backward(x, x2 if x2 is None else _c(x2), gamma, mean, rstd)
        ctx.tg = _targets((gamma, beta))
        return y.reshape(x.shape[:-1] + (gamma.numel(),))

    @staticmethod
    def backward(ctx, dy):
        x, x2, gamma, mean, rstd = ctx.saved_tensors
        dy = _c(dy).reshape(-1, gamma.numel())
        dg, db = _grad_buf(ctx.tg[0], gamma), _grad_buf(ctx.tg[1], gamma)
        x2f = x2.reshape(-1, x2.shape[-1]) if x2 is not None else None
        r = ops.layernorm_bwd(dy, x.reshape(-1, x.shape[-1]), mean, rstd, gamma, dg, db, x2f,
                              defer=_ln_defer(ctx.tg[0] is not None and ctx.tg[1] is not None))
        dg, db = _ret(ctx.tg[0], dg), _ret(ctx.tg[1], db)
        if x2 is None:
            return r.reshape(x.shape), None, dg, db, None
        return r[0].reshape(x.shape), r[1].reshape(x2.shape), dg, db, None


# ============================================================================= Linear (plain / on a concatenation)
class LinearFn(torch.autograd.Function):
    """F.linear on [..., K] (optionally on cat[a, a2] without materialising it: concat_back_dim, MS.py:1027-1030)."""

    @staticmethod
    def forward(ctx, a, a2, w, b):
        a = _c(a)
        a2 = _c(a2) if a2 is not None else None
        y = ops.linear_fwd(a.reshape(-1, a.shape[-1]), w, b, a2.reshape(-1, a2.shape[-1]) if a2 is not None else None)
        ctx.save_for_backward(a, a2, w)
        ctx.has_bias = b is not None
        ctx.tg = _targets((w, b))
        return y.reshape(a.shape[:-1] + (w.shape[0],))

    @staticmethod
    def backward(ctx, dy):
        a, a2, w = ctx.saved_tensors
        dy = _c(dy).reshape(-1, w.shape[0])
        k1 = a.shape[-1]
        af = a.reshape(-1, k1)
        a2f = a2.reshape(-1, a2.shape[-1]) if a2 is not None else None
        dw = _grad_buf(ctx.tg[0], w)
        db = (ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)) if ctx.has_bias else None
        engine = ctx.tg[0] is not None and (ctx.tg[1] is not None or not ctx.has_bias)
        if a2f is not None and engine and CTX.defer_wgrad and k1 % 4 == 0 and a2f.shape[1] % 4 == 0 and (k1 * 4) % 16 == 0 \
                and dy.shape[0] <= DEFER_MAX_TOKENS and ops.wgrad_groupable(dy, af) and ops.wgrad_groupable(dy, a2f) \
                and dw.data_ptr() not in CTX.queued_dw:
            # a layer on a concatenation [a | a2]: two items of the grouped launch, each a column block of dw (row stride K)
            _lin_wgrad(True, dy, af, dw[:, :k1], db)
            _lin_wgrad(True, dy, a2f, dw[:, k1:], None)
        else:
            _defer(engine, lambda: ops.linear_bwd_weight(dy, af, dw, db, a2f), dy, af, a2f)
        r = ops.linear_bwd_data(dy, w, k1=k1)
        rdw, rdb = _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db)     # (new names: the deferred closure above still reads dw / db)
        if a2 is None:
            return r.reshape(a.shape), None, rdw, rdb
        return r[0].reshape(a.shape), r[1].reshape(a2.shape), rdw, rdb


# ============================================================================= shared pieces of the two block types
def _mlp_fwd(x1f, dims, P, s2, eps):
    """part2 (MS.py:403-404/501-502 + :424/:522): x1 + s2 * fc2(GELU(fc1(LN2(x1)))).  Returns y, saved."""
    B, D, H, W = dims
    rps = D * H * W
    xn2, m2, r2 = ops.layernorm_fwd(x1f, P["norm2.weight"], P["norm2.bias"], eps)
    g, h = ops.linear_fwd(xn2, P["mlp.fc1.weight"], P["mlp.fc1.bias"], act=1, want_pre=True)
    y = ops.linear_fwd(g, P["mlp.fc2.weight"], P["mlp.fc2.bias"], resid=x1f, dp_scale=s2, rows_per_sample=rps)
    return y, (xn2, m2, r2, h, g)


def _mlp_bwd(dy, x1f, saved, dims, P, G, s2, side=False):
    """Returns dx1 = dy + LN2'(...) and accumulates the part2 parameter gradients into G."""
    B, D, H, W = dims
    rps = D * H * W
    xn2, m2, r2, h, g = saved      # g = GELU(h) is kept (HBM is plentiful) so the fc2 weight gradient is a plain GEMM
    _lin_wgrad(side, dy, g, G["mlp.fc2.weight"], G["mlp.fc2.bias"], s2, rps)
    dh = ops.linear_bwd_data(dy, P["mlp.fc2.weight"], dp_scale=s2, rows_per_sample=rps, pre_act=h)
    _lin_wgrad(side, dh, xn2, G["mlp.fc1.weight"], G["mlp.fc1.bias"])
    dxn2 = ops.linear_bwd_data(dh, P["mlp.fc1.weight"])
    return ops.layernorm_bwd(dxn2, x1f, m2, r2, P["norm2.weight"], G["norm2.weight"], G["norm2.bias"], add=dy, defer=_ln_defer(side))


SELF_KEYS = ("norm1.weight", "norm1.bias", "self_attn.q.weight", "self_attn.q.bias", "self_attn.kv.weight",
             "self_attn.kv.bias", "self_attn.proj.weight", "self_attn.proj.bias", "norm2.weight", "norm2.bias",
             "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")
CROSS_KEYS = ("norm1.weight", "norm1.bias", "cross_attn.q.weight", "cross_attn.q.bias", "cross_attn.kv.weight",
              "cross_attn.kv.bias", "cross_attn.proj.weight", "cross_attn.proj.bias", "conv_offset.0.weight",
              "conv_offset.0.bias", "conv_offset.1.norm.weight", "conv_offset.1.norm.bias", "conv_offset.3.weight",
              "norm2.weight", "norm2.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")


def _packed_qkv(P, G):
    """If q.weight | kv.weight and q.bias | kv.bias (and their gradient targets) sit back to back in memory -- TrainEngine lays
    the flat buffers out that way -- return ([3C, C] weight view, [3C] bias view, gradient views); else None."""
    wq, wkv, bq, bkv = (P[k] for k in ("self_attn.q.weight", "self_attn.kv.weight", "self_attn.q.bias", "self_attn.kv.bias"))
    gs = [G.get(k) for k in ("self_attn.q.weight", "self_attn.kv.weight", "self_attn.q.bias", "self_attn.kv.bias")]
    if any(g is None for g in gs):
        return None
    C = wq.shape[0]
    def adj(a, b):
        return a.data_ptr() + 4 * a.numel() == b.data_ptr()
    if not (adj(wq, wkv) and adj(bq, bkv) and adj(gs[0], gs[1]) and adj(gs[2], gs[3])):
        return None
    return (wq.as_strided((3 * C, C), (C, 1)), bq.as_strided((3 * C,), (1,)),
            gs[0].as_strided((3 * C, C), (C, 1)), gs[2].as_strided((3 * C,), (1,)))


# ============================================================================= TransformerBlock3D (MS.py:430-524)
class SelfBlockFn(torch.autograd.Function):
    @staticmethod
    @_in_block
    def forward(ctx, x, s1, s2, heads, window, eps, *params):
        P = dict(zip(SELF_KEYS, params))
        x = _c(x)
        B, D, H, W, C = x.shape
        dims = (B, D, H, W)
        ws = effective_window((D, H, W), window)
        pd = _padded((D, H, W), ws)
        padded = pd != (D, H, W)
        pdims = (B,) + pd
        rps = D * H * W
        scale = (C // heads) ** -0.5
        xf = x.reshape(-1, C)
        ctx.tg = _targets(params)
        if not padded and _fusable(dims, C, heads, ws, P):
            sv = _self_fwd_fused([xf], [P], [(s1, s2)], dims, heads, eps)[0]
            ctx.save_for_backward(xf, *[sv[k] for k in _SV_KEYS], s1, s2, *params)
            ctx.meta = (dims, heads)
            ctx.fused = True
            return sv["y"].reshape(x.shape)
        ctx.fused = False
        xn, m1, r1 = ops.layernorm_fwd(xf, P["norm1.weight"], P["norm1.bias"], eps)
        xnp = ops.pad3d(xn, dims, pd) if padded else xn             # F.pad AFTER the norm (MS.py:477-483)
        fused = _packed_qkv(P, dict(zip(SELF_KEYS, ctx.tg)))
        if fused is not None:
            # engine mode: q.weight | kv.weight (and the biases) are adjacent in the flat buffer -> ONE [3C, C] projection
            q = ops.linear_fwd(xnp, fused[0], fused[1])             # packed [T, 3C] = [q | k | v]
            kv = q
            o = ops.window_attn_fwd_qkv(q, pdims, heads, ws, scale)
        else:
            q = ops.linear_fwd(xnp, P["self_attn.q.weight"], P["self_attn.q.bias"])
            kv = ops.linear_fwd(xnp, P["self_attn.kv.weight"], P["self_attn.kv.bias"])
            o = ops.window_attn_fwd(q, kv, pdims, heads, ws, scale)
        if padded:
            o = ops.crop3d(o, dims, pd)                              # proj is per token: crop before it (MS.py:497-498)
        x1 = ops.linear_fwd(o, P["self_attn.proj.weight"], P["self_attn.proj.bias"], resid=xf, dp_scale=s1,
                            rows_per_sample=rps)
        y, mlp_saved = _mlp_fwd(x1, dims, P, s2, eps)
        ctx.save_for_backward(xf, m1, r1, xnp, q, kv, o, x1, s1, s2, *mlp_saved, *params)
        ctx.meta = (dims, ws, pd, padded, heads, scale, fused is not None)
        return y.reshape(x.shape)

    @staticmethod
    @_in_block
    def backward(ctx, dy):
        sv = ctx.saved_tensors
        if ctx.fused:
            m = len(_SV_KEYS)
            xf, svd, (s1, s2), params = sv[0], dict(zip(_SV_KEYS, sv[1:1 + m])), sv[1 + m:3 + m], sv[3 + m:]
            P = dict(zip(SELF_KEYS, params))
            G = {k: _grad_buf(t, v) for (k, v), t in zip(P.items(), ctx.tg)}
            dims, heads = ctx.meta
            C = xf.shape[1]
            side = all(t is not None for t in ctx.tg)
            dx = _self_bwd_fused([_c(dy).reshape(-1, C)], [xf], [svd], [P], [G], [(s1, s2)], dims, heads, [side])[0]
            return (dx.reshape(dims + (C,)), None, None, None, None, None) + tuple(_ret(t, G[k]) for k, t in zip(SELF_KEYS, ctx.tg))
        xf, m1, r1, xnp, q, kv, o, x1, s1, s2 = sv[:10]
        mlp_saved = sv[10:15]
        params = sv[15:]
        P = dict(zip(SELF_KEYS, params))
        G = {k: _grad_buf(t, v) for (k, v), t in zip(P.items(), ctx.tg)}
        dims, ws, pd, padded, heads, scale, packed = ctx.meta
        B, D, H, W = dims
        pdims = (B,) + pd
        rps = D * H * W
        C = xf.shape[1]
        dy = _c(dy).reshape(-1, C)
        side = all(t is not None for t in ctx.tg)      # engine mode: gradients land in the flat buffer, nobody waits for them
        dx1 = _mlp_bwd(dy, x1, mlp_saved, dims, P, G, s2, side)
        _lin_wgrad(side, dx1, o, G["self_attn.proj.weight"], G["self_attn.proj.bias"], s1, rps)
        do = ops.linear_bwd_data(dx1, P["self_attn.proj.weight"], dp_scale=s1, rows_per_sample=rps)
        if padded:
            do = ops.pad3d(do, dims, pd)
        if packed:
            wqkv, _, gw, gb = _packed_qkv(P, G)
            dqkv = ops.window_attn_bwd_qkv(q, do, pdims, heads, ws, scale)
            _lin_wgrad(side, dqkv, xnp, gw, gb)
            dxn = ops.linear_bwd_data(dqkv, wqkv)
        else:
            dq, dkv = ops.window_attn_bwd(q, kv, do, pdims, heads, ws, scale)
            _lin_wgrad(side, dq, xnp, G["self_attn.q.weight"], G["self_attn.q.bias"])
            _lin_wgrad(side, dkv, xnp, G["self_attn.kv.weight"], G["self_attn.kv.bias"])
            dxn = ops.linear_bwd_data(dq, P["self_attn.q.weight"])
            ops.linear_bwd_data(dkv, P["self_attn.kv.weight"], out=dxn, accumulate=True)
        if padded:
            dxn = ops.crop3d(dxn, dims, pd)
        dx = ops.layernorm_bwd(dxn, xf, m1, r1, P["norm1.weight"], G["norm1.weight"], G["norm1.bias"], add=dx1, defer=_ln_defer(side))
        return (dx.reshape(B, D, H, W, C), None, None, None, None, None) + \
            tuple(_ret(t, G[k]) for k, t in zip(SELF_KEYS, ctx.tg))


# ============================================================================= CrossTransformerBlock3D (MS.py:277-426)
class CrossBlockFn(torch.autograd.Function):
    @staticmethod
    @_in_block
    def forward(ctx, x, xa, s1, s2, heads, window, eps, *params):
        P = dict(zip(CROSS_KEYS, params))
        x, xa = _c(x), _c(xa)
        B, D, H, W, C = x.shape
        dims = (B, D, H, W)
        ws = effective_window((D, H, W), window)
        pd = _padded((D, H, W), ws)
        padded = pd != (D, H, W)
        pdims = (B,) + pd
        rps = D * H * W
        scale = (C // heads) ** -0.5
        xf, xaf = x.reshape(-1, C), xa.reshape(-1, C)
        if not padded and _fusable(dims, C, heads, ws, P):
            hd = _cross_head_fwd(xf, xaf, P, dims, eps)
            sv = ops.block_fwd([{"x": xf, "kvsrc": hd[5], "P": P, "attn": "cross_attn", "s1": s1, "s2": s2, "want_xn": False}], dims, C,
                               heads, eps, scale)[0]
            ctx.save_for_backward(xf, xaf, *hd, *[sv[k] for k in _CSV_KEYS], s1, s2, *params)
            ctx.meta = (dims, heads, eps)
            ctx.tg = _targets(params)
            ctx.fused = True
            return sv["y"].reshape(x.shape)
        ctx.fused = False
        xn, m1, r1 = ops.layernorm_fwd(xf, P["norm1.weight"], P["norm1.bias"], eps)   # only x is normed (MS.py:343)
        xnp = ops.pad3d(xn, dims, pd) if padded else xn
        xap = ops.pad3d(xaf, dims, pd) if padded else xaf
        hid = ops.conv3_fwd(xnp, P["conv_offset.0.weight"], P["conv_offset.0.bias"], pdims, x2=xap)
        w1 = P["conv_offset.3.weight"]
        flow, xs = ops.offset_sample_fwd(hid, P["conv_offset.1.norm.weight"], P["conv_offset.1.norm.bias"], w1, xap,
                                         pdims, eps)
        q = ops.linear_fwd(xnp, P["cross_attn.q.weight"], P["cross_attn.q.bias"])
        kv = ops.linear_fwd(xs, P["cross_attn.kv.weight"], P["cross_attn.kv.bias"])
        o = ops.window_attn_fwd(q, kv, pdims, heads, ws, scale)
        if padded:
            o = ops.crop3d(o, dims, pd)
        x1 = ops.linear_fwd(o, P["cross_attn.proj.weight"], P["cross_attn.proj.bias"], resid=xf, dp_scale=s1,
                            rows_per_sample=rps)
        y, mlp_saved = _mlp_fwd(x1, dims, P, s2, eps)
        ctx.save_for_backward(xf, m1, r1, xnp, xap, hid, flow, xs, q, kv, o, x1, s1, s2, *mlp_saved, *params)
        ctx.meta = (dims, ws, pd, padded, heads, scale, eps)
        ctx.tg = _targets(params)
        return y.reshape(x.shape)

    @staticmethod
    @_in_block
    def backward(ctx, dy):
        sv = ctx.saved_tensors
        if ctx.fused:
            m = len(_CSV_KEYS)
            xf, xaf, hd = sv[0], sv[1], sv[2:8]
            svd, (s1, s2), params = dict(zip(_CSV_KEYS, sv[8:8 + m])), sv[8 + m:10 + m], sv[10 + m:]
            P = dict(zip(CROSS_KEYS, params))
            G = {k: _grad_buf(t, v) for (k, v), t in zip(P.items(), ctx.tg)}
            dims, heads, eps = ctx.meta
            C = xf.shape[1]
            rps = dims[1] * dims[2] * dims[3]
            side = all(t is not None for t in ctx.tg)
            dyf = _c(dy).reshape(-1, C)
            bo = ops.block_bwd([{"dy": dyf, "x": None, "x1": svd["x1"], "stats": svd["stats"], "q": svd["q"], "kv": svd["kv"],
                                 "h": svd["h"], "xn2": svd["xn2"], "P": P, "attn": "cross_attn", "s1": s1, "s2": s2, "cross": True,
                                 "want_copy": True}], dims, C, heads, (C // heads) ** -0.5)[0]
            xn, m1, r1, hid, flow, xsamp = hd
            _queue_block_wgrads(side, P, G, "cross_attn", svd, bo, dyf, xn, xsamp, s1, s2, rps)
            _ln_partials(side, bo["ln2_part"], bo["tiles"], C, G["norm2.weight"], G["norm2.bias"])
            dxa = ops.zero_(torch.empty_like(xaf))                  # scatter target of the sampler (one memset node)
            dx = _cross_head_bwd(side, P, G, dims, eps, xf, xaf, xn, m1, r1, hid, flow, bo["dx"], bo["dxs"], dxa, bo["dx1_copy"])
            shape = dims + (C,)
            return (dx.reshape(shape), dxa.reshape(shape), None, None, None, None, None) + \
                tuple(_ret(t, G[k]) for k, t in zip(CROSS_KEYS, ctx.tg))
        xf, m1, r1, xnp, xap, hid, flow, xs, q, kv, o, x1, s1, s2 = sv[:14]
        mlp_saved = sv[14:19]
        params = sv[19:]
        P = dict(zip(CROSS_KEYS, params))
        G = {k: _grad_buf(t, v) for (k, v), t in zip(P.items(), ctx.tg)}
        dims, ws, pd, padded, heads, scale, eps = ctx.meta
        B, D, H, W = dims
        pdims = (B,) + pd
        rps = D * H * W
        C = xf.shape[1]
        dy = _c(dy).reshape(-1, C)
        side = all(t is not None for t in ctx.tg)
        dx1 = _mlp_bwd(dy, x1, mlp_saved, dims, P, G, s2, side)
        _lin_wgrad(side, dx1, o, G["cross_attn.proj.weight"], G["cross_attn.proj.bias"], s1, rps)
        do = ops.linear_bwd_data(dx1, P["cross_attn.proj.weight"], dp_scale=s1, rows_per_sample=rps)
        if padded:
            do = ops.pad3d(do, dims, pd)
        dq, dkv = ops.window_attn_bwd(q, kv, do, pdims, heads, ws, scale)
        _lin_wgrad(side, dq, xnp, G["cross_attn.q.weight"], G["cross_attn.q.bias"])
        _lin_wgrad(side, dkv, xs, G["cross_attn.kv.weight"], G["cross_attn.kv.bias"])
        dxnp = ops.linear_bwd_data(dq, P["cross_attn.q.weight"])
        dxs = ops.linear_bwd_data(dkv, P["cross_attn.kv.weight"])
        dxap = ops.zero_(torch.empty_like(xap))                    # atomic scatter target of the sampler (one memset node)
        dhid = ops.offset_sample_bwd(dxs, hid, P["conv_offset.1.norm.weight"], P["conv_offset.1.norm.bias"],
                                     P["conv_offset.3.weight"], xap, flow, dxap, G["conv_offset.1.norm.weight"],
                                     G["conv_offset.1.norm.bias"], G["conv_offset.3.weight"], pdims, eps)
        _defer_conv_wgrad(side, dhid, xnp, xap, G["conv_offset.0.weight"], G["conv_offset.0.bias"], pdims)
        ops.conv3_bwd_data(dhid, P["conv_offset.0.weight"], pdims, C, C, dx1=dxnp, dx2=dxap, acc1=True, acc2=True)
        if padded:
            dxn = ops.crop3d(dxnp, dims, pd)
            dxa = ops.crop3d(dxap, dims, pd)
        else:
            dxn, dxa = dxnp, dxap
        dx = ops.layernorm_bwd(dxn, xf, m1, r1, P["norm1.weight"], G["norm1.weight"], G["norm1.bias"], add=dx1, defer=_ln_defer(side))
        return (dx.reshape(B, D, H, W, C), dxa.reshape(B, D, H, W, C), None, None, None, None, None) + \
            tuple(_ret(t, G[k]) for k, t in zip(CROSS_KEYS, ctx.tg))


# ============================================================================= fused window-local blocks (block_fwd / block_bwd)
def _fusable(dims, C, heads, window, P):
    """Tokens per tile if the fused block kernels take this shape (even grid, 2x2x2 windows, head_dim 16 / 32, C <= 384), else 0."""
    if tuple(window) != (2, 2, 2) or not FUSE_BLOCKS:
        return 0
    return ops.block_tile_tokens(dims, C, heads, P["mlp.fc1.weight"].shape[0])


import os as _os
FUSE_BLOCKS = True   # off = the round-1 per-op launch sequence everywhere


def _queue_block_wgrads(side, P, G, attn, sv, bo, dy, xn, kv_in, s1, s2, rps):
    """The five nn.Linear weight gradients of a block from the operands the fused kernels left in HBM (deferred in engine mode)."""
    if bo.get("dy16") is not None:          # bf16 storage: every operand pair was left as bf16 (block_fused.h block_saves_bf16)
        dy, xn = bo["dy16"], sv["xn"]
        kv_in = sv["kvs16"] if sv.get("kvs16") is not None else sv["xn"]
    _lin_wgrad(side, dy, sv["g"], G["mlp.fc2.weight"], G["mlp.fc2.bias"], s2, rps)
    _lin_wgrad(side, bo["dh"], sv["xn2"], G["mlp.fc1.weight"], G["mlp.fc1.bias"])
    _lin_wgrad(side, bo["dx1"], sv["o"], G[f"{attn}.proj.weight"], G[f"{attn}.proj.bias"], s1, rps)
    _lin_wgrad(side, bo["dq"], xn, G[f"{attn}.q.weight"], G[f"{attn}.q.bias"])
    _lin_wgrad(side, bo["dkv"], kv_in, G[f"{attn}.kv.weight"], G[f"{attn}.kv.bias"])


def _ln_partials(side, part, tiles, C, dg, db):
    """Per-tile LayerNorm gain / bias partial sums of a fused backward: queued for the grouped finish, or finished now."""
    item = (part, tiles, C, dg, db)
    lst = _ln_defer(side)
    if lst is not None:
        lst.append(item)
    elif _LOCAL is not None:
        _LOCAL[1].append(item)
    else:
        ops.layernorm_bwd_finish([item])


# The cross pair of a depth slot starts with LayerNorm 1 of the self pair's outputs (MS.py:343; its result feeds conv_offset[0]): the
# self pair's launch holds those rows when it writes them, so it writes their LayerNorm too (micf_block_fwd_group.nln_g) -- one
# launch and one re-read of y per slot off the forward chain.  BasicLayer hands the cross blocks' norm1 over in CTX.next_ln.
FUSE_NEXT_LN = True


def _self_fwd_fused(xs, Ps, scales, dims, heads, eps, save=True, next_ln=None):
    """xs: 1 or 2 [T, C] inputs (the two modalities); one launch.  Returns the per-group saved dicts (save=False: 'y' only).
    next_ln: [(gamma, beta)] per group -> every dict also has "nln" = (LayerNorm(y), mean, rstd); the launch clears `hid` where
    the offset convolution accumulates into it (returned as the second value then)."""
    C = xs[0].shape[1]
    groups = [{"x": x, "kvsrc": None, "P": P, "attn": "self_attn", "s1": s[0], "s2": s[1]} for x, P, s in zip(xs, Ps, scales)]
    if next_ln is None:
        return ops.block_fwd(groups, dims, C, heads, eps, (C // heads) ** -0.5, save=save)
    hid = None
    if ops.offset_head_needs_zero(dims, C):
        hid = torch.empty((len(xs), xs[0].shape[0], 16), dtype=torch.float32, device=xs[0].device)
    for i, gd in enumerate(groups):
        gd["next_ln"] = (next_ln[i][0], next_ln[i][1], hid[i] if hid is not None else None)
    return ops.block_fwd(groups, dims, C, heads, eps, (C // heads) ** -0.5, save=save), hid


# The LayerNorm-1 backward of a cross pair (MS.py:343 through autograd) produces exactly the output gradients of the self pair of the
# same depth slot, whose backward launch is next on the chain: in engine mode with bf16 storage the cross pair therefore does NOT
# launch it -- it hands the self pair its partial sums (the returned "gradients": residual path + the other branches) and parks
# what the LayerNorm backward needs here; micf_block_bwd runs it as its prologue (micf_block_bwd_group.pre_d).  24 launches and
# a [T, C] round trip per slot off the chain.  Only BasicLayer's self -> cross sequence sets it up (`lazy_ln` of CrossPairFn), only
# while an engine step scopes CTX.lazy_ln_ok, and the engine checks after backward that nothing parked was left unconsumed.
LAZY_LN_DEFAULT = True


def lazy_ln_pending():
    return len(CTX.lazy_ln)


def _self_bwd_fused(dys, xs, svs, Ps, Gs, scales, dims, heads, sides):
    C = xs[0].shape[1]
    rps = dims[1] * dims[2] * dims[3]
    groups = [{"dy": dy, "x": x, "x1": sv["x1"], "stats": sv["stats"], "q": sv["q"], "kv": sv["kv"], "h": sv["h"], "xn2": sv["xn2"], "P": P,
               "attn": "self_attn", "s1": s[0], "s2": s[1], "cross": False} for dy, x, sv, P, s in zip(dys, xs, svs, Ps, scales)]
    pres = [CTX.lazy_ln.pop(dy.data_ptr(), None) for dy in dys]
    for gd, pre in zip(groups, pres):
        if pre is not None:
            if tuple(pre["d"].shape) != tuple(gd["dy"].shape):
                raise RuntimeError("parked LayerNorm backward does not match the gradient it was parked for")
            gd["pre"] = pre
    bos = ops.block_bwd(groups, dims, C, heads, (C // heads) ** -0.5)
    for bo, pre in zip(bos, pres):
        if pre is not None:
            _ln_partials(pre["side"], bo["pre_part"], bo["tiles"], C, pre["dgamma"], pre["dbeta"])
    for dy, sv, bo, P, G, s, side in zip(dys, svs, bos, Ps, Gs, scales, sides):
        _queue_block_wgrads(side, P, G, "self_attn", sv, bo, dy, sv["xn"], sv["xn"], s[0], s[1], rps)
        _ln_partials(side, bo["ln2_part"], bo["tiles"], C, G["norm2.weight"], G["norm2.bias"])
        _ln_partials(side, bo["ln1_part"], bo["tiles"], C, G["norm1.weight"], G["norm1.bias"])
    return [bo["dx"] for bo in bos]


_SV_KEYS = ("xn", "q", "kv", "o", "x1", "xn2", "h", "g", "stats")


class SelfPairFn(torch.autograd.Function):
    """self_blocks1[i](x), self_blocks2[i](xa) (MS.py:700): two independent TransformerBlock3D of the same shape, ONE fused launch
    forward and ONE backward."""

    @staticmethod
    @_in_block
    def forward(ctx, x, xa, sa1, sa2, sb1, sb2, heads, eps, grad_mode, *params):
        n = len(SELF_KEYS)
        Ps = [dict(zip(SELF_KEYS, params[:n])), dict(zip(SELF_KEYS, params[n:]))]
        x, xa = _c(x), _c(xa)
        B, D, H, W, C = x.shape
        dims = (B, D, H, W)
        xs = [x.reshape(-1, C), xa.reshape(-1, C)]
        _lib.set_unit("self_fwd", 2, xs[0].shape[0], C)
        scales = [(sa1, sa2), (sb1, sb2)]
        # grad_mode = torch.is_grad_enabled() read by the CALLER: inside Function.forward grad mode is always off, and under
        # torch.no_grad() needs_input_grad still reports the trainable parameters.  No gradient will be asked for (validation, the
        # sliding-window inference): nothing is saved, the launch writes y only.
        save = bool(grad_mode) and any(ctx.needs_input_grad)
        nl, CTX.next_ln, CTX.next_ln_out = CTX.next_ln, None, None
        # (a record left behind by a forward that raised belongs to another stage: only one that fits this launch is used)
        if nl is not None and FUSE_NEXT_LN and GROUP_CROSS_HEADS and ops.block_fuses_sampler(C, heads) \
                and all(t.numel() == C and t.device == x.device for pr in nl for t in pr):
            svs, hid = _self_fwd_fused(xs, Ps, scales, dims, heads, eps, save=save, next_ln=nl)
            CTX.next_ln_out = ((svs[0]["y"].data_ptr(), svs[1]["y"].data_ptr()), [svs[0]["nln"], svs[1]["nln"]], hid)
        else:
            svs = _self_fwd_fused(xs, Ps, scales, dims, heads, eps, save=save)
        if save:
            ctx.save_for_backward(*xs, *[sv[k] for sv in svs for k in _SV_KEYS], sa1, sa2, sb1, sb2, *params)
            ctx.meta = (dims, heads)
            ctx.tg = _targets(params)
        return svs[0]["y"].reshape(x.shape), svs[1]["y"].reshape(x.shape)

    @staticmethod
    @_in_block
    def backward(ctx, dy, dya):
        sv = ctx.saved_tensors
        dims, heads = ctx.meta
        n, m = len(SELF_KEYS), len(_SV_KEYS)
        xs = list(sv[:2])
        svs = [dict(zip(_SV_KEYS, sv[2:2 + m])), dict(zip(_SV_KEYS, sv[2 + m:2 + 2 * m]))]
        sa1, sa2, sb1, sb2 = sv[2 + 2 * m:6 + 2 * m]
        params = sv[6 + 2 * m:]
        Ps = [dict(zip(SELF_KEYS, params[:n])), dict(zip(SELF_KEYS, params[n:]))]
        tgs = [ctx.tg[:n], ctx.tg[n:]]
        Gs = [_grad_bufs(P, tg) for P, tg in zip(Ps, tgs)]
        sides = [all(t is not None for t in tg) for tg in tgs]
        C = xs[0].shape[1]
        _lib.set_unit("self_bwd", 2, xs[0].shape[0], C)
        dys = [_c(dy).reshape(-1, C), _c(dya).reshape(-1, C)]
        with _local_batch():
            dxs = _self_bwd_fused(dys, xs, svs, Ps, Gs, [(sa1, sa2), (sb1, sb2)], dims, heads, sides)
        shape = dims + (C,)
        grads = tuple(_ret(t, G[k]) for G, tg in zip(Gs, tgs) for k, t in zip(SELF_KEYS, tg))
        return (dxs[0].reshape(shape), dxs[1].reshape(shape), None, None, None, None, None, None, None) + grads


def _cross_head_fwd(xf, xaf, P, dims, eps):
    """LN1(x), offset conv on cat[LN1(x), raw xa], offset head + deformable sampling of raw xa (MS.py:343-384)."""
    xn, m1, r1 = ops.layernorm_fwd(xf, P["norm1.weight"], P["norm1.bias"], eps)
    hid = ops.conv3_fwd(xn, P["conv_offset.0.weight"], P["conv_offset.0.bias"], dims, x2=xaf)
    flow, xs = ops.offset_sample_fwd(hid, P["conv_offset.1.norm.weight"], P["conv_offset.1.norm.bias"], P["conv_offset.3.weight"], xaf,
                                     dims, eps)
    return xn, m1, r1, hid, flow, xs


def _cross_head_bwd(side, P, G, dims, eps, xf, xaf, xn, m1, r1, hid, flow, dxq, dxs, dxa_acc, add, out=None):
    """Adjoint of _cross_head_fwd.  dxq: the q path's pre-LayerNorm gradient (accumulated into in place); dxs: gradient of the
    sampled K/V source; dxa_acc: buffer the raw-xa gradient is ACCUMULATED into (zero, or already holding other terms);
    add / out: LayerNorm-1 backward computes out = add + LN1'(dxq) (out may alias add)."""
    C = xf.shape[1]
    dhid = ops.offset_sample_bwd(dxs, hid, P["conv_offset.1.norm.weight"], P["conv_offset.1.norm.bias"], P["conv_offset.3.weight"],
                                 xaf, flow, dxa_acc, G["conv_offset.1.norm.weight"], G["conv_offset.1.norm.bias"],
                                 G["conv_offset.3.weight"], dims, eps)
    _defer_conv_wgrad(side, dhid, xn, xaf, G["conv_offset.0.weight"], G["conv_offset.0.bias"], dims)
    ops.conv3_bwd_data(dhid, P["conv_offset.0.weight"], dims, C, C, dx1=dxq, dx2=dxa_acc, acc1=True, acc2=True)
    return ops.layernorm_bwd(dxq, xf, m1, r1, P["norm1.weight"], G["norm1.weight"], G["norm1.bias"], add=add, defer=_ln_defer(side),
                             out=out)


_CSV_KEYS = ("q", "kv", "o", "x1", "xn2", "h", "g", "stats", "xn", "kvs16")      # (xn / kvs16: bf16 storage only, else None)

# The two offset heads of a cross pair (LN1 -> 3^3 conv -> sampling, and their adjoints) are independent per-op launch chains:
# the second one runs on a side stream (a fork / join in the captured graph), as round 1 did for whole blocks.
OVERLAP_CROSS_HEADS = True
# ... superseded by the grouped entry points: both heads of a pair in every launch (micf_offset_head_fwd / _bwd)
GROUP_CROSS_HEADS = True
# ... and the sampling half of the head (LayerNorm-16 / GELU / 1^3 conv / reference points / trilinear gather) inside the cross
# pair's block_fwd launch: one launch and one [T, C] round trip less per cross pair
FUSE_SAMPLER = True


def _conv_offset_wgrad(side, dhid, xn, xa, G, dims):
    _defer_conv_wgrad(side, dhid, xn, xa, G["conv_offset.0.weight"], G["conv_offset.0.bias"], dims)
_SIDE = {}


def _side_stream(device):
    st = _SIDE.get(device)
    if st is None:
        st = _SIDE[device] = torch.cuda.Stream(device=device)
    return st


def _cross_head_adjoint(i, hds, bos, Ps, Gs, xs, acc, dims, eps, C, sides):
    """Adjoint of block i's offset head: sampler backward (raw-xa gradient -> the OTHER input's accumulator), conv weight
    gradient, conv data gradient (-> block i's pre-LayerNorm gradient and the other input's accumulator)."""
    xn, m1, r1, hid, flow, xsamp = hds[i]
    dhid = ops.offset_sample_bwd(bos[i]["dxs"], hid, Ps[i]["conv_offset.1.norm.weight"], Ps[i]["conv_offset.1.norm.bias"],
                                 Ps[i]["conv_offset.3.weight"], xs[1 - i], flow, acc[1 - i], Gs[i]["conv_offset.1.norm.weight"],
                                 Gs[i]["conv_offset.1.norm.bias"], Gs[i]["conv_offset.3.weight"], dims, eps)
    _defer_conv_wgrad(sides[i], dhid, xn, xs[1 - i], Gs[i]["conv_offset.0.weight"], Gs[i]["conv_offset.0.bias"], dims)
    ops.conv3_bwd_data(dhid, Ps[i]["conv_offset.0.weight"], dims, C, C, dx1=bos[i]["dx"], dx2=acc[1 - i], acc1=True, acc2=True)


class CrossPairFn(torch.autograd.Function):
    """blocks1[i](x, xa), blocks2[i](xa, x) (MS.py:701; both read the PRE-update pair): per modality LN1 -> offset conv ->
    deformable sampling (per-op kernels: they need token neighbourhoods), then ONE fused launch for both blocks' window-local
    part.  Backward: ONE fused launch, then the two offset heads' adjoints; each input's gradient (own block's LN1 backward +
    skip + the OTHER block's raw-K/V-source gradient) is accumulated in one buffer that starts as the fused kernel's second copy of
    dx1 -- no zero fill, no autograd add."""

    @staticmethod
    @_in_block
    def forward(ctx, x, xa, sa1, sa2, sb1, sb2, heads, eps, grad_mode, *params):
        n = len(CROSS_KEYS)
        Ps = [dict(zip(CROSS_KEYS, params[:n])), dict(zip(CROSS_KEYS, params[n:]))]
        x, xa = _c(x), _c(xa)
        # (the inputs are the outputs of the self pair of this depth slot and nobody else reads them: see CTX.lazy_ln)
        ctx.lazy_ln = CTX.cross_after_self == (x.data_ptr(), xa.data_ptr())
        CTX.cross_after_self = None
        B, D, H, W, C = x.shape
        dims = (B, D, H, W)
        xs = [x.reshape(-1, C), xa.reshape(-1, C)]
        _lib.set_unit("cross_fwd", 2, xs[0].shape[0], C)
        fuse_sampler = GROUP_CROSS_HEADS and FUSE_SAMPLER and ops.block_fuses_sampler(C, heads)
        if GROUP_CROSS_HEADS:
            # both offset heads per launch (micf_offset_head_fwd): no fork / join inside the captured graph
            pre, CTX.next_ln_out = CTX.next_ln_out, None
            if pre is not None and pre[0] == (xs[0].data_ptr(), xs[1].data_ptr()):
                lns, hid = pre[1], pre[2]               # (written, and `hid` cleared, by the self pair's launch: FUSE_NEXT_LN)
            else:
                hid = None
                if ops.offset_head_needs_zero(dims, C):     # (atomically accumulated conv output: cleared by the LayerNorm launch)
                    hid = torch.empty((2, xs[0].shape[0], 16), dtype=torch.float32, device=x.device)
                lns = ops.layernorm_fwd_pair(xs, [P["norm1.weight"] for P in Ps], [P["norm1.bias"] for P in Ps], eps, zero=hid)
            # (fuse_sampler: the 3^3 conv only -- LayerNorm(16) / GELU / 1^3 conv / sampling run inside the block launch below)
            outs = ops.offset_head_fwd([{"xn": lns[i][0], "xa": xs[1 - i], "P": Ps[i]} for i in (0, 1)], dims, eps, hid,
                                       sample=not fuse_sampler)
            heads_ = [lns[i] + outs[i] for i in (0, 1)]
        elif OVERLAP_CROSS_HEADS:
            main, side = torch.cuda.current_stream(), _side_stream(x.device)
            side.wait_stream(main)
            h0 = _cross_head_fwd(xs[0], xs[1], Ps[0], dims, eps)
            with torch.cuda.stream(side):
                h1 = _cross_head_fwd(xs[1], xs[0], Ps[1], dims, eps)
            main.wait_stream(side)
            for t in h1:
                t.record_stream(main)
            heads_ = [h0, h1]
        else:
            heads_ = [_cross_head_fwd(xs[i], xs[1 - i], Ps[i], dims, eps) for i in (0, 1)]
        scales = [(sa1, sa2), (sb1, sb2)]
        groups = [{"x": xs[i], "kvsrc": heads_[i][5], "P": Ps[i], "attn": "cross_attn", "s1": scales[i][0], "s2": scales[i][1],
                   "want_xn": False} for i in (0, 1)]
        if fuse_sampler:
            for i in (0, 1):
                groups[i].update(kvsrc=None, hid=heads_[i][3], samp_src=xs[1 - i])
        save = bool(grad_mode) and any(ctx.needs_input_grad)   # (grad_mode: the caller's torch.is_grad_enabled(), see SelfPairFn)
        svs = ops.block_fwd(groups, dims, C, heads, eps, (C // heads) ** -0.5, save=save)
        if not save:
            return svs[0]["y"].reshape(x.shape), svs[1]["y"].reshape(x.shape)
        if fuse_sampler:                                 # (flow: saved for the sampler's adjoint; xs32: fp32 storage's kv-gradient operand)
            heads_ = [heads_[i][:4] + (svs[i]["flow"], svs[i]["xs32"]) for i in (0, 1)]
        ctx.save_for_backward(*xs, *[t for hd in heads_ for t in hd], *[sv[k] for sv in svs for k in _CSV_KEYS], sa1, sa2, sb1, sb2,
                              *params)
        ctx.meta = (dims, heads, eps)
        ctx.tg = _targets(params)
        return svs[0]["y"].reshape(x.shape), svs[1]["y"].reshape(x.shape)

    @staticmethod
    @_in_block
    @_batched
    def backward(ctx, dy, dya):
        sv = ctx.saved_tensors
        dims, heads, eps = ctx.meta
        n, m = len(CROSS_KEYS), len(_CSV_KEYS)
        xs = list(sv[:2])
        hds = [sv[2:8], sv[8:14]]
        svs = [dict(zip(_CSV_KEYS, sv[14:14 + m])), dict(zip(_CSV_KEYS, sv[14 + m:14 + 2 * m]))]
        sa1, sa2, sb1, sb2 = sv[14 + 2 * m:18 + 2 * m]
        scales = [(sa1, sa2), (sb1, sb2)]
        params = sv[18 + 2 * m:]
        Ps = [dict(zip(CROSS_KEYS, params[:n])), dict(zip(CROSS_KEYS, params[n:]))]
        tgs = [ctx.tg[:n], ctx.tg[n:]]
        Gs = [_grad_bufs(P, tg) for P, tg in zip(Ps, tgs)]
        sides = [all(t is not None for t in tg) for tg in tgs]
        C = xs[0].shape[1]
        _lib.set_unit("cross_bwd", 2, xs[0].shape[0], C)
        rps = dims[1] * dims[2] * dims[3]
        dys = [_c(dy).reshape(-1, C), _c(dya).reshape(-1, C)]
        groups = [{"dy": dys[i], "x": None, "x1": svs[i]["x1"], "stats": svs[i]["stats"], "q": svs[i]["q"], "kv": svs[i]["kv"],
                   "h": svs[i]["h"], "xn2": svs[i]["xn2"], "P": Ps[i], "attn": "cross_attn", "s1": scales[i][0], "s2": scales[i][1], "cross": True,
                   "want_copy": True} for i in (0, 1)]
        bos = ops.block_bwd(groups, dims, C, heads, (C // heads) ** -0.5)
        acc = [bos[0]["dx1_copy"], bos[1]["dx1_copy"]]       # acc[i] becomes d(input i): starts as dx1 of block i
        for i in (0, 1):
            xn, m1, r1, hid, flow, xsamp = hds[i]
            _queue_block_wgrads(sides[i], Ps[i], Gs[i], "cross_attn", svs[i], bos[i], dys[i], xn, xsamp, scales[i][0], scales[i][1], rps)
            _ln_partials(sides[i], bos[i]["ln2_part"], bos[i]["tiles"], C, Gs[i]["norm2.weight"], Gs[i]["norm2.bias"])
        if GROUP_CROSS_HEADS:
            # block i's raw-xa gradient goes to the OTHER input's buffer; its own LN1 backward lands in acc[i] (in place)
            hgroups = [{"dxs": bos[i]["dxs"], "hid": hds[i][3], "flow": hds[i][4], "xa": xs[1 - i], "P": Ps[i],
                        "G": Gs[i], "dxa": acc[1 - i], "dxn": bos[i]["dx"]} for i in (0, 1)]
            # small grids: the sampler's finishing launch only sums head-parameter partials -- off the data-gradient chain with
            # the other parameter gradients (its partial table then lives in a workspace of its own until the flush)
            defer_ws = None
            if all(sides) and CTX.defer_calls and CTX.defer_wgrad and ops.offset_head_finish_deferrable(dims):
                defer_ws = torch.empty(ops.offset_head_bwd_workspace(2, dims), dtype=torch.float32, device=xs[0].device)
            dhids = ops.offset_head_bwd(hgroups, dims, eps, defer_ws=defer_ws)
            if defer_ws is not None:
                def fin():
                    ops.offset_head_bwd_finish(hgroups, dhids, dims, defer_ws)
                fin.finish = (hgroups, dhids, tuple(dims), defer_ws)        # (the flush groups these into one launch)
                _defer(True, fin, defer_ws, *dhids)
            for i in (0, 1):
                _conv_offset_wgrad(sides[i], dhids[i], hds[i][0], xs[1 - i], Gs[i], dims)
        else:
            main = torch.cuda.current_stream()
            side = _side_stream(dys[0].device) if OVERLAP_CROSS_HEADS else None
            if side is not None:
                side.wait_stream(main)
            for i in (0, 1):
                with torch.cuda.stream(side if (i == 1 and side is not None) else main):
                    _cross_head_adjoint(i, hds, bos, Ps, Gs, xs, acc, dims, eps, C, sides)
            if side is not None:
                main.wait_stream(side)
        lazy = (ctx.lazy_ln and CTX.lazy_ln_ok and GROUP_CROSS_HEADS and all(sides) and CTX.defer_wgrad and bos[0].get("dy16") is not None
                and ops.block_fuses_sampler(C, heads))
        if lazy:
            # parked for the self pair's backward launch (see CTX.lazy_ln): acc[i] leaves as the PARTIAL gradient of input i
            for i in (0, 1):
                CTX.lazy_ln[acc[i].data_ptr()] = {"d": bos[i]["dx"], "x": xs[i], "mean": hds[i][1], "rstd": hds[i][2],
                                               "gamma": Ps[i]["norm1.weight"], "dgamma": Gs[i]["norm1.weight"],
                                               "dbeta": Gs[i]["norm1.bias"], "side": sides[i]}
        elif GROUP_CROSS_HEADS:
            ops.layernorm_bwd_pair([{"dy": bos[i]["dx"], "x": xs[i], "mean": hds[i][1], "rstd": hds[i][2], "gamma": Ps[i]["norm1.weight"],
                                     "dgamma": Gs[i]["norm1.weight"], "dbeta": Gs[i]["norm1.bias"], "add": acc[i], "out": acc[i]}
                                    for i in (0, 1)], [_ln_defer(sides[i]) for i in (0, 1)])
        else:
            for i in (0, 1):
                xn, m1, r1 = hds[i][:3]
                ops.layernorm_bwd(bos[i]["dx"], xs[i], m1, r1, Ps[i]["norm1.weight"], Gs[i]["norm1.weight"], Gs[i]["norm1.bias"],
                                  add=acc[i], defer=_ln_defer(sides[i]), out=acc[i])
        shape = dims + (C,)
        grads = tuple(_ret(t, G[k]) for G, tg in zip(Gs, tgs) for k, t in zip(CROSS_KEYS, tg))
        return (acc[0].reshape(shape), acc[1].reshape(shape), None, None, None, None, None, None, None) + grads


# ============================================================================= patch embed / merging / expand / head
PATCH_GEMM = True
class PatchEmbedFn(torch.autograd.Function):
    """PatchEmbed3D (MS.py:860-878) on modality `mod` of vol [B, nmod, D, H, W] -> (B, D', H', W', E) channels-last."""

    # The k = s convolutions run as space-to-depth + the linear GEMMs (ops.space_to_depth; weights and gradients in place)
    # when PATCH_GEMM is on; off = the element-gather GEMM entry points (micf_patch_embed_* / micf_conv_down_* / micf_conv_up_*).
    @staticmethod
    def forward(ctx, vol, mod, w, b, p):
        vol = _c(vol)
        B, nmod, D, H, W = vol.shape
        E = w.shape[0]
        if PATCH_GEMM and w.shape[1] == 1 and p in (2, 4):
            a = ops.space_to_depth(vol, (B, D, H, W), 1, p, batch_stride=nmod * D * H * W, offset=mod * D * H * W)
            y = ops.linear_fwd(a, w.reshape(E, p ** 3), b).reshape(B, -(-D // p), -(-H // p), -(-W // p), E)
            ctx.save_for_backward(a, w)
        else:
            y = ops.patch_embed_fwd(vol, mod, w, b, p)
            ctx.save_for_backward(vol, w)
        ctx.meta = (mod, p, PATCH_GEMM and w.shape[1] == 1 and p in (2, 4))
        ctx.tg = _targets((w, b))
        return y

    @staticmethod
    def backward(ctx, dy):
        vol, w = ctx.saved_tensors
        mod, p, gemm = ctx.meta
        dw = _grad_buf(ctx.tg[0], w)
        db = ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        dy = _c(dy)
        if gemm:
            _lin_wgrad(ctx.tg[0] is not None and ctx.tg[1] is not None, dy.reshape(-1, w.shape[0]), vol, dw.view(w.shape[0], -1), db)
        else:
            _defer(ctx.tg[0] is not None and ctx.tg[1] is not None, lambda: ops.patch_embed_bwd_weight(dy, vol, mod, dw, db, p), dy, vol)
        return None, None, _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db), None   # the input volume is data: no gradient (train.py:177-185)


class PatchRowsEmbedFn(torch.autograd.Function):
    """PatchEmbed3D on an already gathered [tokens, k^3] patch-row matrix (ops.patch_rows_prepared: the input tail fused into the
    gather) -> tokens [rows, E].  The rows are data: no gradient."""

    @staticmethod
    def forward(ctx, a, w, b):
        E = w.shape[0]
        ctx.save_for_backward(a, w)
        ctx.tg = _targets((w, b))
        return ops.linear_fwd(a, w.reshape(E, -1), b)

    @staticmethod
    def backward(ctx, dy):
        a, w = ctx.saved_tensors
        dw = _grad_buf(ctx.tg[0], w)
        db = ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        _lin_wgrad(ctx.tg[0] is not None and ctx.tg[1] is not None, _c(dy).reshape(-1, w.shape[0]), a, dw.view(w.shape[0], -1), db)
        return None, _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db)


class PatchEmbedPairFn(torch.autograd.Function):
    """PatchEmbed3D (shared weights, MS.py:1003-1004) on BOTH modalities of vol [B, 2, D, H, W] in one GEMM:
    -> [2B, D', H', W', E] (modality-major: the first B samples are modality 0)."""

    @staticmethod
    def forward(ctx, vol, w, b, p):
        vol = _c(vol)
        B, nmod, D, H, W = vol.shape
        E = w.shape[0]
        rows = B * (-(-D // p)) * (-(-H // p)) * (-(-W // p))
        a = torch.empty((2 * rows, p ** 3), dtype=torch.float32, device=vol.device)
        for m in (0, 1):
            ops.space_to_depth(vol, (B, D, H, W), 1, p, batch_stride=nmod * D * H * W, offset=m * D * H * W, out=a[m * rows:(m + 1) * rows])
        ctx.save_for_backward(a, w)
        ctx.tg = _targets((w, b))
        return ops.linear_fwd(a, w.reshape(E, p ** 3), b).reshape(2 * B, -(-D // p), -(-H // p), -(-W // p), E)

    @staticmethod
    def backward(ctx, dy):
        a, w = ctx.saved_tensors
        dw = _grad_buf(ctx.tg[0], w)
        db = ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        _lin_wgrad(ctx.tg[0] is not None and ctx.tg[1] is not None, _c(dy).reshape(-1, w.shape[0]), a, dw.view(w.shape[0], -1), db)
        return None, _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db), None


# The output of an encoder stage has two consumers: PatchMerging (next stage) and the decoder's skip concat.  autograd sums their
# gradients with an ATen elementwise launch per tensor.  Instead the skip gradient is handed over Python-side: ConvDownFn.forward
# registers a token under its input's address; the decoder passes the skip through SkipMailFn (created late in the forward, so
# its backward runs as soon as the concat linear's backward has produced the skip gradient -- long before this stage's
# PatchMerging backward, which depends on it through the whole deeper network), whose backward parks the gradient in the token
# and returns None (no second gradient path for autograd to sum); ConvDownFn.backward adds it inside its depth-to-space scatter.
SKIP_MAIL = True


class _SkipToken:
    # mailed / received: SkipMailFn forwards that took this token / backwards that delivered; consumed: ConvDownFn's backward ran
    __slots__ = ("grad", "mailed", "received", "consumed")

    def __init__(self):
        self.grad = None
        self.mailed = self.received = 0
        self.consumed = False


def clear_skip_tokens():
    CTX.skip_tokens.clear()


def skip_token(t):
    """The token of the PatchMerging launch that consumed exactly this tensor in the current forward, or None."""
    return CTX.skip_tokens.get((t.data_ptr(), tuple(t.shape))) if SKIP_MAIL else None


class SkipMailFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, token):
        ctx.token = token
        token.mailed += 1
        return t.view_as(t)

    @staticmethod
    def backward(ctx, g):
        tok = ctx.token
        if tok.consumed:
            # the hand-over relies on autograd reaching the skip consumer (decoder) before PatchMerging's backward (encoder); any
            # other order (autograd.grad on a sub-graph, a second pass over a retained graph) would drop this gradient silently
            raise RuntimeError("SkipMailFn.backward ran after the PatchMerging backward that should have added its gradient "
                               "(set functional.SKIP_MAIL = False for backward passes in a non-standard order)")
        tok.grad = g if tok.grad is None else tok.grad + g
        tok.received += 1
        return None, None


class ConvDownFn(torch.autograd.Function):
    """Conv3d(C->N, k=s=2) of PatchMerging on channels-last x (MS.py:548-557)."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _c(x)
        ctx.tg = _targets((w, b))
        ctx.xshape = tuple(x.shape)
        ctx.gemm = PATCH_GEMM and tuple(w.shape[2:]) == (2, 2, 2)
        ctx.token = None
        if ctx.gemm and SKIP_MAIL and x.requires_grad:
            ctx.token = CTX.skip_tokens[(x.data_ptr(), tuple(x.shape))] = _SkipToken()
        if ctx.gemm:
            B, D, H, W, C = x.shape
            N = w.shape[0]
            a = ops.space_to_depth(x, (B, D, H, W), C, 2)
            ctx.save_for_backward(a, w)
            return ops.linear_fwd(a, w.reshape(N, 8 * C), b).reshape(B, -(-D // 2), -(-H // 2), -(-W // 2), N)
        ctx.save_for_backward(x, w)
        return ops.conv_down_fwd(x, w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dw = _grad_buf(ctx.tg[0], w)
        db = ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        if ctx.gemm:
            B, D, H, W, C = ctx.xshape
            N = w.shape[0]
            dy2 = dy.reshape(-1, N)
            _lin_wgrad(ctx.tg[0] is not None and ctx.tg[1] is not None, dy2, x, dw.view(N, 8 * C), db)
            da = ops.linear_bwd_data(dy2, w.reshape(N, 8 * C))
            skip = ctx.token.grad if ctx.token is not None else None      # the skip connection's gradient of x (SkipMailFn)
            if ctx.token is not None:
                if ctx.token.received != ctx.token.mailed and not ctx.token.consumed:
                    raise RuntimeError(f"PatchMerging backward reached before its skip connection's gradient arrived "
                                       f"({ctx.token.received} of {ctx.token.mailed} mailed): functional.SKIP_MAIL = False for this pass order")
                ctx.token.consumed = True
                ctx.token.grad = None
            if skip is not None:
                skip = _c(skip)
            return ops.depth_to_space(da, (B, D, H, W), C, 2, add=skip), _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db)
        _defer(ctx.tg[0] is not None and ctx.tg[1] is not None, lambda: ops.conv_down_bwd_weight(dy, x, dw, db), dy, x)
        return ops.conv_down_bwd_data(dy, w, ctx.xshape), _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db)


class ConvUpFn(torch.autograd.Function):
    """ConvTranspose3d(C->N, k=s) on channels-last x (PatchExpand MS.py:575-577; reverse_patch_embedding MS.py:1037)."""

    @staticmethod
    def forward(ctx, x, w, b, k):
        x = _c(x)
        ctx.save_for_backward(x, w)
        ctx.k = k
        ctx.tg = _targets((w, b))
        ctx.gemm = PATCH_GEMM and k in (2, 4) and (w.shape[1] * k ** 3) % 4 == 0
        if ctx.gemm:
            B, D, H, W, C = x.shape
            N = w.shape[1]
            ya = ops.linear_bwd_data(x.reshape(-1, C), w.reshape(C, N * k ** 3))      # [coarse voxels, (n, tap)]
            return ops.depth_to_space(ya, (B, D * k, H * k, W * k), N, k, bias=b)
        return ops.conv_up_fwd(x, w, b, k)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        dw = _grad_buf(ctx.tg[0], w)
        db = ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[1], dtype=w.dtype, device=w.device)
        k = ctx.k
        if ctx.gemm:
            B, D, H, W, C = x.shape
            N = w.shape[1]
            dya = ops.space_to_depth(dy, (B, D * k, H * k, W * k), N, k)             # [coarse voxels, (n, tap)]
            ok = ctx.tg[0] is not None and ctx.tg[1] is not None
            _lin_wgrad(ok, x.reshape(-1, C), dya, dw.view(C, N * k ** 3), None)
            _defer(ok, lambda: ops.colsum_(dy.reshape(-1, N), db), dy)
            dx = ops.linear_fwd(dya, w.reshape(C, N * k ** 3), None).reshape(x.shape)
            return dx, _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db), None
        _defer(ctx.tg[0] is not None and ctx.tg[1] is not None, lambda: ops.conv_up_bwd_weight(dy, x, dw, db, k), dy, x)
        return ops.conv_up_bwd_data(dy, w, tuple(x.shape), ctx.k), _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db), None


class OutConvFn(torch.autograd.Function):
    """Head.out_conv: Conv3d(E/2 -> classes, 3, padding=1) from the channels-last feature to NCDHW logits (MS.py:1053)."""

    @staticmethod
    def forward(ctx, feat, w, b):
        feat = _c(feat)
        B, D, H, W, C = feat.shape
        ctx.save_for_backward(feat, w)
        ctx.tg = _targets((w, b))
        return ops.conv3_fwd(feat.reshape(-1, C), w, b, (B, D, H, W), ncdhw_out=True)

    @staticmethod
    def backward(ctx, dy):
        feat, w = ctx.saved_tensors
        B, D, H, W, C = feat.shape
        dy = _c(dy)
        dw = _grad_buf(ctx.tg[0], w)
        db = ctx.tg[1] if ctx.tg[1] is not None else torch.zeros(w.shape[0], dtype=w.dtype, device=w.device)
        f2 = feat.reshape(-1, C)
        ops.conv3_bwd_weight(dy, f2, dw, db, (B, D, H, W), ncdhw=True)
        dx, _ = ops.conv3_bwd_data(dy, w, (B, D, H, W), C, 0, ncdhw=True)
        return dx.reshape(feat.shape), _ret(ctx.tg[0], dw), _ret(ctx.tg[1], db)


FUSE_TAIL_PATCHES = True


# MDiceLoss's forward inside the head's logits store (SURVEY 8 row A19 as worded).  The reference's call shape stays
# `loss = criterion(model(x), target)` (train.py:185-187): a caller that knows the target before the forward (TrainEngine) leaves it
# here; HeadTailFn's fused forward then folds the Dice / BCE sums of every logit it stores and parks (logits, target, loss, sums);
# DiceBCEFn picks the parked result up when it is handed exactly those two tensors, and computes it itself otherwise.
FUSE_LOSS = True


class HeadTailFn(torch.autograd.Function):
    """reverse_patch_embedding (ConvTranspose3d 2E -> E/2, k = s = P; MS.py:1037) + Head.out_conv (Conv3d E/2 -> classes, 3,
    padding=1; MS.py:1053) composed into one linear map on the coarse grid (csrc/head_tail.hip): channels-last coarse
    feature x (B, Dc, Hc, Wc, 2E) -> NCDHW logits (B, classes, P*Dc, P*Hc, P*Wc).  The E/2-channel fine feature is never built."""

    @staticmethod
    def forward(ctx, x, w_up, b_up, w_out, b_out, wb=None, bf=None, w_up_t=None, packs=None):
        x = _c(x)
        B, Dc, Hc, Wc, Ci = x.shape
        P = w_up.shape[2]
        if wb is None:                                  # (else: composed earlier, off the critical path)
            w_up_t = ops.head_tail_transposed_up(w_up)
            wb, bf = ops.head_tail_compose(w_up, b_up, w_out, w_up_t)
        xf = x.reshape(-1, Ci)
        fused = FUSE_TAIL_PATCHES and ops.head_tail_fused_supported((B, Dc, Hc, Wc), Ci, b_out.shape[0], P)
        if fused:                                       # bf16 mode: no T / U patch matrices (head_tail_fused.hip)
            if packs is None:
                packs = ops.head_tail_pack(wb, bf, b_out, P)
            Co = b_out.shape[0]
            tgt = CTX.loss_mail["target"] if FUSE_LOSS else None
            CTX.loss_mail["result"] = None
            if tgt is not None:
                from .loss.dice import as_target
                tgt = as_target((B, Co, P * Dc, P * Hc, P * Wc), tgt)
                want = (B, Co, P * Dc, P * Hc, P * Wc) if tgt.dtype != torch.uint8 else (B, P * Dc, P * Hc, P * Wc)
                if tuple(tgt.shape) != want or not tgt.is_contiguous() or not tgt.is_cuda:
                    tgt = None
            if tgt is not None:
                y, loss, sums = ops.head_tail_fwd_loss_fused(xf, packs[0], (B, Dc, Hc, Wc), Co, P, tgt)
                CTX.loss_mail["result"] = (y, tgt, loss, sums)
            else:
                y = ops.head_tail_fwd_fused(xf, packs[0], (B, Dc, Hc, Wc), Co, P)
        else:
            packs = (None, None)
            t = ops.linear_fwd(xf, wb, bf)
            y = ops.head_tail_col2im(t, b_out, (B, Dc, Hc, Wc), P)
        ctx.save_for_backward(xf, wb, w_up, b_up, w_out, w_up_t, packs[1])
        ctx.dims = (B, Dc, Hc, Wc)
        ctx.tg = _targets((w_up, b_up, w_out, b_out))
        return y

    @staticmethod
    def backward(ctx, dy):
        xf, wb, w_up, b_up, w_out, w_up_t, pack_bwd = ctx.saved_tensors
        B, Dc, Hc, Wc = ctx.dims
        P = w_up.shape[2]
        dy = _c(dy)
        grads = [_grad_buf(t, p) for t, p in zip(ctx.tg, (w_up, b_up, w_out, w_out.new_empty(w_out.shape[0])))]
        if pack_bwd is not None:
            dx = ops.head_tail_bwd_data_fused(dy, pack_bwd, ctx.dims, xf.shape[1], P)
            u = None
        else:
            u = ops.head_tail_im2col(dy, ctx.dims, P)
            dx = ops.linear_bwd_data(u, wb)

        def weight_grads():       # composed-map gradient, then its decomposition into the two layers' parameters
            got = ops.head_tail_bwd_weight_fused(dy, xf, ctx.dims, P) if u is None else None
            if got is not None:
                dwb, dbf = got
            else:
                uu = u if u is not None else ops.head_tail_im2col(dy, ctx.dims, P)
                dwb, dbf = ops.zero_(torch.empty_like(wb)), ops.zero_(torch.empty(wb.shape[0], dtype=wb.dtype, device=wb.device))
                ops.linear_bwd_weight(uu, xf, dwb, dbf)
            ops.head_tail_decompose(dwb, dbf, w_up, b_up, w_out, *grads, w_up_t=w_up_t)

        _defer(all(t is not None for t in ctx.tg), weight_grads, u if u is not None else dy, xf, wb)
        return (dx.reshape(B, Dc, Hc, Wc, -1),) + tuple(_ret(t, g) for t, g in zip(ctx.tg, grads)) + (None, None, None, None)


class ResizeTrilinearFn(torch.autograd.Function):
    """F.interpolate(mode='trilinear', align_corners=True) on channels-last volumes (MS.py:1018-1025)."""

    @staticmethod
    def forward(ctx, x, size):
        x = _c(x)
        ctx.xshape = tuple(x.shape)
        return ops.resize_trilinear_fwd(x, size)

    @staticmethod
    def backward(ctx, dy):
        return ops.resize_trilinear_bwd(_c(dy), ctx.xshape), None


class DiceBCEFn(torch.autograd.Function):
    """MDiceLoss.forward (dice.py:158-166)."""

    @staticmethod
    def forward(ctx, logits, target):
        logits, target = _c(logits), _c(target)
        parked, CTX.loss_mail["result"] = CTX.loss_mail["result"], None
        if parked is not None and parked[0].data_ptr() == logits.data_ptr() and parked[0].shape == logits.shape \
                and parked[1].data_ptr() == target.data_ptr() and parked[1].shape == target.shape and parked[1].dtype == target.dtype:
            loss, sums = parked[2], parked[3]           # folded into the head's logits store (HeadTailFn)
        else:
            loss, sums = ops.dice_bce_fwd(logits, target)
        ctx.save_for_backward(logits, target, sums)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        logits, target, sums = ctx.saved_tensors
        return ops.dice_bce_bwd(logits, target, sums, _c(g).reshape(1)), None
