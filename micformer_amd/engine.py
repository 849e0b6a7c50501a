"""Training-step engine for the MicFormer hot path on MI355X.

One step == the reference's loop body (train_mmwhs_noPad.py:183-207):
    optimizer.zero_grad(); segs = model(x); loss = MDiceLoss(segs, labels); loss.backward();
    Adam(lr=1e-4, wd=0).step(); CosineAnnealingLR.step()
with MI355X-first plumbing around the HIP kernels:
  * all parameters live in ONE flat fp32 HBM buffer (and their gradients / Adam moments in three more), so zero_grad
    is one memset, Adam is one fused streaming kernel and the data-parallel gradient exchange is one RCCL all-reduce
    over xGMI; `state_dict()` keys/shapes are untouched (the nn.Parameters become views);
  * the LR schedule and Adam's step counter live on the device (micf_adam_tick), so the whole step -- forward,
    loss, backward, Adam -- can be captured once into a HIP graph and replayed without host work;
  * data parallelism: one process per GPU, whole CT+MR pairs sharded across ranks, per-rank loss (the Dice sums are
    per-rank batches -- the accepted DDP semantics), gradient all-reduce(sum)/world, rank-identical weights.
"""
import torch
import torch.distributed as dist

from . import _lib, ops
from .dist import FlatGradSync, OverlappedGradReduce, flatten_views, last_writer_per_bucket, module_buckets
from .loss.dice import MDiceLoss


class TrainEngine:
    def __init__(self, model, base_lr=1e-4, t_max=150, eta_min=0.0, betas=(0.9, 0.999), eps=1e-8, criterion=None,
                 use_graph=False, process_group=None, grad_bucket_bytes=64 << 20, parallel_modalities=True,
                 defer_wgrad=True, split_step=None, always_collective=False, flush_points=True, early_adam=True,
                 dp_graph_flushes=6, grad_bf16=None, segmented=None, uniform_batches=False):
        self.model = model
        self.criterion = criterion if criterion is not None else MDiceLoss()
        self.base_lr, self.t_max, self.eta_min = base_lr, t_max, eta_min
        self.betas, self.eps = betas, eps
        self.sync = FlatGradSync(process_group, grad_bucket_bytes, always=always_collective)
        self.world = self.sync.world
        self._flatten()
        self.sync.broadcast_params(self.flat_p)                    # rank-identical initial weights
        # the CT and MR branches of every depth slot are independent: issue them on two streams (see BasicLayer.forward)
        # Both switches are process-global module flags of functional / models: the engine sets them only for the duration of
        # its own forward + backward (see _scoped_flags) so a manual loss.backward(), a second model or a gradient check
        # outside step() still gets its weight gradients computed in place.
        self.parallel_modalities = bool(parallel_modalities)
        self.defer_wgrad = bool(defer_wgrad)     # linear weight gradients: queued in backward, one grouped flush
        self.flush_points = bool(flush_points)   # ... launched stage by stage on a side stream while backward continues
        self.early_adam = bool(early_adam)       # Adam over the decoder + last-stage parameters starts under the encoder's backward
        # Data parallel (or split_step=True, a single-GPU test hook): the graph holds forward + backward only; the queued weight
        # gradients are then launched group by group and every gradient slice is all-reduced as soon as its last writer is
        # done, overlapping RCCL with the remaining weight-gradient launches (see _flush_and_reduce).
        self.split_step = (self.world > 1) if split_step is None else bool(split_step)
        # Data-parallel step: the first `dp_graph_flushes` flush points the backward reaches (decoder and 4^3 stages = most of
        # the parameter bytes) launch their weight gradients INSIDE the graph, under the backward chain, exactly as on one GPU;
        # those slices of the flat gradient are complete when the replay ends and are all-reduced at once, under the remaining
        # (encoder) weight-gradient groups that are launched after the replay.  Measured on one GPU (data-parallel layout):
        # 0 -> 13.3 ms, 4 -> 12.9, 6 -> 12.8, all -> 12.7 (but then nothing is left to hide RCCL behind).
        self.dp_graph_flushes = int(dp_graph_flushes)
        # bf16 wire format of the gradient exchange (123 MB instead of 247 MB per step at base): each slice is rounded to bf16,
        # sum-reduced, widened back; Adam still reads fp32.  Default (None): ON for world > 1 in the bf16 arithmetic mode -- the
        # budget of tools/dp_budget.py: at 8 ranks the fp32 exchange is ~1.65 ms of ring time against ~1 ms of
        # post-replay launches to hide it behind (7.4 x of 8), the bf16 one ~0.93 ms (7.6-7.7 x); the gradients of this mode
        # already carry bf16 operand rounding, and the 2-rank test holds the result within 2^-7 of the fp32 exchange.  The fp32
        # parity mode keeps the exact fp32 exchange.
        # Auto (None) is resolved PER STEP from the arithmetic mode in force (an engine built in bf16 and switched to the fp32
        # parity mode exchanges fp32 again); an explicit bool is the caller's decision in every mode.
        self._grad_bf16_auto = grad_bf16 is None
        self._grad_bf16_set = bool(grad_bf16)
        self._eager_exchange = {}
        # Graph layout of a captured step: ONE HIP graph (round 2; where its side branch runs is the graph executor's choice) or
        # a sequence of graphs replayed on two streams with explicit events (functional.StepSegmenter; opt-in, see _lib.py).
        # (needs the runtime's graph packet capture off, see _lib.GRAPH_SEGMENTS_OK: one graph otherwise)
        self.segmented = (__import__("os").environ.get("MICF_SEGMENTED", "0") == "1") if segmented is None else bool(segmented)
        # ... and single-process only: with an initialised NCCL (= RCCL) process group hipGraphLaunch of a segment crashed in the
        # host runtime (tests/test_gpu_model.py::test_split_step_with_rccl_on_one_rank, ROCm 7.2), so data-parallel jobs keep the
        # one-graph layout that the round-2 RCCL tests ran on.
        self.segmented = self.segmented and _lib.GRAPH_SEGMENTS_OK and not (dist.is_available() and dist.is_initialized())
        # Data parallel: every step asks all ranks whether THEIR batch fits the captured graph (one tiny all-reduce + a host read of
        # the result, i.e. a host-device synchronisation per step: the host cannot run ahead of the device).  A caller whose batches
        # have the same shape on every rank at every step (bench.py's synthetic batch; a DistributedSampler that pads or drops the
        # last batch) promises so with uniform_batches=True and the decision is local.
        self.uniform_batches = bool(uniform_batches)
        self._wplan = None
        self.use_graph = use_graph
        self._graph = None
        self._static = None
        self._static_mode = None
        self._many, self._carried, self._carry_groups = None, None, None
        self._param_at = {o: p for p, o in zip(self.params, self.offsets)}
        self.steps_done = 0
        from . import functional as _fn
        self.ctx = _fn.StepContext()             # this engine's launch-plan state (queues, mailboxes, switches)

    @property
    def grad_bf16(self):
        """Is the gradient exchange of a step taken NOW on the bf16 wire (else the exact fp32 all-reduce)?"""
        return self.grad_wire is not None

    @property
    def grad_wire(self):
        """Wire format of the gradient exchange of a step taken NOW: None (exact fp32 all-reduce), "bf16" (bf16 on the links, fp32
        in the sums: dist.WireExchange) or "bf16-ring" (MICF_GRAD_WIRE=bf16-ring: the backend's all-reduce IN bf16, round 5's form,
        kept as the comparison)."""
        env = __import__("os").environ.get("MICF_GRAD_WIRE", "bf16")
        if not self._grad_bf16_auto:
            on = self._grad_bf16_set
        else:
            on = self.world > 1 and ops.compute_dtype() == "bf16" and env != "fp32"
        if not on:
            return None
        return "bf16-ring" if env == "bf16-ring" else "bf16"

    def _exchange_for(self, buckets, bucket_last):
        return OverlappedGradReduce(self.sync, self.flat_g, buckets, bucket_last, wire=self.grad_wire, wire_ops=ops.HipWireOps())

    # ------------------------------------------------------------------ flat parameter / gradient storage
    def _flatten(self):
        params = [p for p in self.model.parameters()]
        dev = params[0].device
        if dev.type != "cuda":
            raise RuntimeError("TrainEngine needs the model on a CUDA (ROCm) device")
        sizes = [p.numel() for p in params]
        # 4-element (16 B) alignment of every tensor so kernels can use 16-byte accesses on parameter views
        # Flat order = model order, except that every self-attention's q.weight | kv.weight and q.bias | kv.bias are made
        # neighbours, so the two projections (same input) run as ONE [3C, C] GEMM (functional._packed_qkv).
        names = [n for n, _ in self.model.named_parameters()]
        pos = {n: i for i, n in enumerate(names)}
        order, seen = [], set()
        for i, n in enumerate(names):
            if i in seen:
                continue
            if n.endswith("self_attn.q.weight"):
                base = n[:-len("q.weight")]
                group = [pos.get(base + k) for k in ("q.weight", "kv.weight", "q.bias", "kv.bias")]
                if all(g is not None for g in group):
                    order += group
                    seen.update(group)
                    continue
            order.append(i)
            seen.add(i)
        offs_perm, total = flatten_views([params[i] for i in order])
        offs = [0] * len(params)
        for o, i in zip(offs_perm, order):
            offs[i] = o
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o, n in zip(params, offs, sizes):
                self.flat_p[o:o + n].copy_(p.reshape(-1))
                p.data = self.flat_p[o:o + n].view(p.shape)
                p.grad = self.flat_g[o:o + n].view(p.shape)
                p._micf_grad = p.grad       # backward kernels accumulate straight into the flat gradient buffer
        self.params, self.offsets, self.sizes = params, offs, sizes
        self.adam_state = ops.adam_state(dev)
        # Early optimiser step: the backward visits the stages in reverse, so once it has left the LAST encoder stage every
        # gradient of that stage, the whole decoder and the head is final.  If those parameters form the tail [cut, total) of
        # the flat buffers (model order: patch embed, encoder stages, decoder, head), Adam runs over that tail on the side stream
        # while the backward of the earlier encoder stages continues, and only [0, cut) is left for the end of the step.
        self._early_cut, self._early_layer, self._adam_tail_done = None, None, False
        stages = getattr(getattr(self.model, "swin", self.model), "layers", None)
        self._stages = stages if stages is not None else []
        self._anchor_layer = stages[len(stages) - 1] if stages is not None and len(stages) >= 2 else None
        self._anchor_buf = None
        if stages is not None and len(stages) >= 2:
            pre = [n.split(".layers.")[0] for n in names if ".layers." in n]
            root = pre[0] if pre else None
            early = lambda n: root is not None and (n.startswith(root + ".patch_embed.") or any(
                n.startswith(f"{root}.layers.{i}.") for i in range(len(stages) - 1)))
            ends = [o + s for n, o, s in zip(names, offs, sizes) if early(n)]
            if ends:
                cut = (max(ends) + 3) // 4 * 4
                if all(o >= cut for n, o in zip(names, offs) if not early(n)) and cut < total:
                    self._early_cut, self._early_layer = cut, stages[len(stages) - 1]
        # What the fused kernels stream instead of the parameters themselves, all refreshed once per step (the weights only
        # change in Adam):
        #   * forward: copies of the block linears' weights (bf16 in bf16 mode, fp32 otherwise) in the K16-blocked order the kernels' matrix-core
        #     fragments read (one grouped launch on a side stream in front of the forward, joined at the first stage);
        #   * backward: transposed copies of the same weights (W^T fp32 row-major, or bf16 K16-blocked) -- one grouped launch
        #     on a side stream, issued when the forward reaches the small stages, next to the zero fill of the gradient buffer;
        #   * the re-laid-out offset-conv weights of the direct conv kernels: one small grouped launch in front of the forward.
        pick = lambda suffixes: [(n, p) for n, p in zip(names, params) if n.endswith(suffixes)]
        self._shadow_w = [(n, p) for n, p in pick(("attn.q.weight", "attn.kv.weight", "attn.proj.weight", "mlp.fc1.weight",
                                                   "mlp.fc2.weight"))
                          if p.dim() == 2 and min(p.shape) <= 384 and not ((p.shape[0] | p.shape[1]) & 15)]   # (fusable widths)
        self._conv_w = [(n, p) for n, p in pick(("conv_offset.0.weight",)) if p.dim() == 5 and p.shape[0] <= 16]
        self._offset_of = {id(p): o for p, o in zip(params, offs)}
        self._prep_plans, self._shadow_bufs, self._prep_split = {}, {}, {}
        self._prep_stream = torch.cuda.Stream(device=dev)

    def _weight_prep(self):
        """(forward | None, backward, conv) launch plans of the current arithmetic mode; creates the shadow buffers on first use.
        The copies are rewritten by every step, so nothing else has to track who wrote the parameters."""
        mode = ops.compute_dtype()
        plans = self._prep_plans.get(mode)
        if plans is None:
            ws = [p for _, p in self._shadow_w]
            offs, total = flatten_views(ws, align=8)
            # Forward copies: the fused kernels only stream K16-blocked operands, so BOTH modes have one (bf16 mode: bf16, half the
            # parameter bytes; fp32 mode: a second full fp32 copy of the block weights, 4 B / parameter, rewritten every step).
            fwd, spec = None, ops.shadow_spec(False)
            if spec is not None:
                buf = self._shadow_bufs[spec[0]] = torch.empty(max(total, 8), dtype=spec[2], device=self.flat_p.device)
                ftrip = []
                for p, o in zip(ws, offs):
                    setattr(p, spec[0], buf[o:o + p.numel()].view(p.shape))
                    ftrip.append((p.data, getattr(p, spec[0]), None))
                fwd = ops.WeightPrepPlan(ftrip, blocked=True)
            attr, transposed, dtype = ops.shadow_spec(True)
            buf = self._shadow_bufs[attr] = torch.empty(max(total, 8), dtype=dtype, device=self.flat_p.device)
            trip = []
            for p, o in zip(ws, offs):
                setattr(p, attr, buf[o:o + p.numel()].view(p.shape[1], p.shape[0]))
                trip.append((p.data, None, getattr(p, attr)))
            if "conv" not in self._shadow_bufs:
                ctrip = []
                for _, p in self._conv_w:
                    p._micf_c3f, p._micf_c3b = ops.conv3_prepared_like(p)
                    ctrip.append((p.data, p._micf_c3f, p._micf_c3b))
                self._shadow_bufs["conv"] = ops.Conv3PrepPlan(ctrip)
            plans = self._prep_plans[mode] = (fwd, ops.WeightPrepPlan(trip, blocked=True), self._shadow_bufs["conv"])
            # the same plans cut at the early-Adam boundary (step_many: the copies of the tail's weights are refreshed only after
            # the tail's carried Adam update of the previous step): (forward early, forward late, conv early, conv late)
            cut = self._early_cut if self._early_cut is not None else self.flat_p.numel()
            late = lambda p: self._offset_of[id(p)] >= cut
            if fwd is not None:
                fe = ops.WeightPrepPlan([t for t, p in zip(ftrip, ws) if not late(p)], blocked=True)
                fl = ops.WeightPrepPlan([t for t, p in zip(ftrip, ws) if late(p)], blocked=True)
            else:
                fe = fl = None
            ctr = [(p.data, p._micf_c3f, p._micf_c3b) for _, p in self._conv_w]
            ce = ops.Conv3PrepPlan([t for t, (_, p) in zip(ctr, self._conv_w) if not late(p)])
            cl = ops.Conv3PrepPlan([t for t, (_, p) in zip(ctr, self._conv_w) if late(p)])
            self._prep_split[mode] = (fe, fl, ce, cl)
        return plans

    # ------------------------------------------------------------------ one optimisation step
    def _scoped_flags(self):
        """Context manager: this engine's launch-layout switches are in force inside, the previous values outside."""
        import contextlib
        from . import functional as _fn
        from .models import MICFormer_self as _ms

        @contextlib.contextmanager
        def scope():
            prev = (_ms.PARALLEL_MODALITIES, self.ctx.defer_wgrad, self.ctx.flush_points, self.ctx.defer_calls, ops.ENGINE_SHADOWS)
            prev_fork, prev_lazy = _ms.FORK_AUTOGRAD_STREAMS, self.ctx.lazy_ln_ok
            self.ctx.lazy_ln_ok = _fn.LAZY_LN_DEFAULT              # cross pairs park their LayerNorm-1 backward for the self pair's launch
            _ms.FORK_AUTOGRAD_STREAMS = not (self.segmented and self.use_graph)
            prev_budget = self.ctx.flush_budget
            _ms.PARALLEL_MODALITIES, self.ctx.defer_wgrad = self.parallel_modalities, self.defer_wgrad
            self.ctx.flush_points = self.defer_wgrad and self.flush_points and (not self.split_step or self.dp_graph_flushes > 0)
            self.ctx.flush_budget = self.dp_graph_flushes if self.split_step else 1 << 30
            self.ctx.defer_calls = self.defer_wgrad
            ops.ENGINE_SHADOWS = True                               # the parameters' shadow copies are current in this scope only
            try:
                yield
            finally:
                _ms.PARALLEL_MODALITIES, self.ctx.defer_wgrad, self.ctx.flush_points, self.ctx.defer_calls, ops.ENGINE_SHADOWS = prev
                self.ctx.flush_budget = prev_budget
                _ms.FORK_AUTOGRAD_STREAMS = prev_fork
                self.ctx.lazy_ln_ok = prev_lazy
        return scope()

    def _fwd_bwd(self, x, target, flush=True, carry_in=None, carry_out=False):
        """Forward + loss + backward of one batch with THIS engine's launch-plan state in force (functional.StepContext: switches,
        queues, mailboxes -- nothing is shared with another engine or with engine-less forwards in the same process)."""
        from . import functional as _fn
        with _fn.use_context(self.ctx):
            return self._fwd_bwd_in_ctx(x, target, flush, carry_in, carry_out)

    def _fwd_bwd_in_ctx(self, x, target, flush, carry_in, carry_out):
        from . import functional as _fn
        _fn.drop_deferred()                                         # nothing left over from a backward that raised
        with self._scoped_flags():
            fwd, bwd, conv = self._weight_prep()
            main, side = torch.cuda.current_stream(), self._prep_stream
            _fn.clear_entry_hooks()
            carried = [None]
            if carry_in is not None:
                # step_many: the previous step of this graph left the parameter-gradient batches of its first flush points and the
                # Adam update of the flat tail [cut, total) to us.  They run NOW on the weight-gradient side stream, stage group by
                # stage group in the order this forward first reads their parameters (last encoder stage, then the decoder stages
                # upwards, the head last): each group = its batches, Adam over its slices of the flat buffers, the forward shadow
                # copies / conv layouts of its weights, then an event; the main chain waits for a group's event at the entry of its
                # stage.  The encoder forward in front of it reads [0, cut) only.
                fe, _, ce, _ = self._prep_split[ops.compute_dtype()]
                ce.launch()
                if fe is not None:
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        fe.launch()
                    _fn.park_entry_hook(lambda: main.wait_stream(side), at=1)
                carried = [None]                                   # (the last group's event, once launched)

                def launch_carried_groups_segmented():
                    # the step as a sequence of graphs: every stage group is its own graph on the side stream with an event behind
                    # it, the main chain is cut where it must wait for one; the backward preparation rides as the last side graph
                    def group_work(grp):
                        def w():
                            _fn.launch_carried(carry_in.get(grp["key"], []))
                            for lo, hi in grp["ranges"]:
                                self._adam_range(lo, hi, 1.0)
                            fl, cl = self._carry_prep(grp)
                            cl.launch()
                            if fl is not None:
                                fl.launch()
                        return [w]

                    def prep():
                        ops.zero_(self.flat_g)
                        bwd.launch()
                    evs = self.ctx.segmenter.run_side_groups([group_work(g) for g in self._carry_groups] + [[prep]], keep=carry_in)
                    for grp, ev in zip(self._carry_groups, evs):
                        _fn.park_entry_hook((lambda e: (lambda: self.ctx.segmenter.wait(e)))(ev), at=grp["entry"], front=True)
                    carried[0] = evs[-1]

                def launch_carried_groups():
                    if self.ctx.segmenter is not None:
                        return launch_carried_groups_segmented()
                    # Created AFTER the main chain's first kernels of this step (stage entry 1: patch embedding launched): a side
                    # branch created first makes the captured graph's executor run it in front of the main chain (measured: the
                    # carried work then sat in a 2 ms tail between the steps instead of under the encoder forward).
                    wside = _fn._wgrad_stream(self.flat_p.device)
                    wside.wait_stream(main)
                    with torch.cuda.stream(wside):
                        for grp in self._carry_groups:
                            _fn.launch_carried(carry_in.get(grp["key"], []))
                            for lo, hi in grp["ranges"]:
                                self._adam_range(lo, hi, 1.0)       # (same tick as the previous step's [0, cut))
                            fl, cl = self._carry_prep(grp)
                            cl.launch()
                            if fl is not None:
                                fl.launch()
                            ev = torch.cuda.Event()
                            ev.record(wside)
                            _fn.park_entry_hook((lambda e: (lambda: main.wait_event(e)))(ev), at=grp["entry"])
                            carried[0] = ev
                    self.ctx.wside_used.add(self.flat_p.device)
                # (entry 3 = the 8^3 encoder stage, where the main chain leaves half the chip idle; it must not be later than the
                #  backward preparation parked there, which waits for the last group's event)
                _fn.park_entry_hook(launch_carried_groups, at=3)
            else:
                conv.launch()                                       # offset-conv weight layouts (one small launch)
                if fwd is not None:                                 # K16-blocked block weights, next to the patch embedding
                    side.wait_stream(main)
                    with torch.cuda.stream(side):
                        fwd.launch()
                    _fn.park_entry_hook(lambda: main.wait_stream(side), at=1)

            def backward_prep():                                    # under the latency-bound small stages of the forward
                if carry_in is not None and self.ctx.segmenter is not None:
                    return                                          # (segmented step_many: part of the carried side graphs)
                side.wait_stream(main)
                if carried[0] is not None:
                    side.wait_event(carried[0])                     # (the carried Adam reads the gradients this zero fill clears)
                with torch.cuda.stream(side):
                    ops.zero_(self.flat_g)                          # optimizer.zero_grad()        train.py:183
                    bwd.launch()                                    # W^T shadows of the fused backward
            _fn.park_entry_hook(backward_prep, at=3)
            self.ctx.backward_hooks.clear()
            self._adam_tail_done = False
            # Early Adam reads the tail of the flat gradient mid-backward: only valid when the flush point that fires it has
            # launched EVERY weight gradient queued so far (flush points on, unbounded budget, no token cap) -- otherwise the
            # linear / LayerNorm gradients of the tail would still sit in the queue and be applied one step late, never.
            full_flush = self.ctx.flush_points and self.ctx.flush_budget >= (1 << 30) and _fn.FLUSH_MAX_TOKENS >= (1 << 30)
            if carry_out:
                # the first flush points' batches are set aside for the next step's head (functional.CARRY); the region ends
                # where early Adam would start: that hook closes it (and leaves the side-stream anchor node there)
                assert flush and self.world == 1 and self._early_cut is not None and full_flush
                self.ctx.carry["on"], self.ctx.carry["open"], self.ctx.carry["stash"] = True, True, []

                def close_carry():
                    self.ctx.carry["open"] = False
                    if self.ctx.segmenter is None:
                        self._side_anchor()
                self.ctx.backward_hooks[id(self._early_layer)] = close_carry
            elif flush and self.world == 1 and self._early_cut is not None and self.early_adam and full_flush:
                self.ctx.backward_hooks[id(self._early_layer)] = self._early_adam_segment if self.ctx.segmenter is not None else self._early_adam
            elif self._anchor_layer is not None and self.ctx.segmenter is None:
                self.ctx.backward_hooks[id(self._anchor_layer)] = self._side_anchor
            # (the plain MDiceLoss: its forward sums are folded into the head's logits store -- functional.LOSS_MAIL)
            self.ctx.loss_mail["target"] = target if type(self.criterion) is MDiceLoss else None
            from .models import MICFormer_self as _msh
            # (Head.forward composes its weights behind the carried update: an event, or -- a sequence of graphs -- just later)
            _msh.HEAD_WEIGHTS_AFTER = (carried if self.ctx.segmenter is None else [None]) if carry_in is not None else None
            try:
                logits = self.model(x)                              #                              train.py:185
            finally:
                self.ctx.loss_mail["target"] = None
                _msh.HEAD_WEIGHTS_AFTER = None
            loss = self.criterion(logits, target)                   #                              train.py:187
            self.ctx.loss_mail["result"] = None
            _fn.run_entry_hook(force=True)                          # (fewer than 3 stages: launched here)
            main.wait_stream(side)
            if carry_in is not None and self.ctx.segmenter is not None:
                self.ctx.segmenter.wait(carried[0])                      # (zero fill + W^T shadows: the last carried side graph)
            if self.ctx.segmenter is not None:
                # segmented capture: the flush points end / begin stream captures from inside backward; keep autograd on the
                # calling thread so every hipStreamBeginCapture / EndCapture of this step is issued by ONE host thread
                with torch.autograd.set_multithreading_enabled(False):
                    loss.backward()
            else:
                loss.backward()                                     #                              train.py:200
            if _fn.lazy_ln_pending():                               # (a parked LayerNorm backward nobody ran: never silently)
                n = _fn.lazy_ln_pending()
                _fn.drop_deferred()
                raise RuntimeError(f"{n} parked LayerNorm backward(s) were not consumed by a self-pair launch (functional._LAZY_LN)")
            _fn.flush_wgrad(calls_only=not flush)                   # what is still queued: grouped linear weight gradients (left
            _fn.join_wgrad_stream()                                 # to the data-parallel tail when flush=False), closures
            if carry_out:
                if self.ctx.carry["open"]:
                    raise RuntimeError("the carry region was never closed (no flush point at the last encoder stage)")
                groups, cur = {}, []
                for key, batch in self.ctx.carry["stash"]:               # (flush points inside a stage belong to the stage being left)
                    cur.append(batch)
                    if key is not None:
                        groups.setdefault(key, []).extend(cur)
                        cur = []
                if cur or set(groups) - {g["key"] for g in self._carry_groups}:
                    raise RuntimeError("carried parameter-gradient batches do not map onto the stage groups")
                self._carried = groups
                self.ctx.carry["on"], self.ctx.carry["stash"] = False, []
        return loss.detach()

    def _adam(self, grad_scale):
        lo, hi = 0, self.flat_p.numel()
        if self._adam_tail_done:                                    # [cut, total) was updated mid-backward (same tick)
            hi, self._adam_tail_done = self._early_cut, False
        else:
            ops.adam_tick(self.adam_state, self.base_lr, self.eta_min, self.t_max)   # scheduler (per iteration) train.py:206-207
        self._adam_range(lo, hi, grad_scale)

    def _adam_range(self, lo, hi, grad_scale):
        ops.adam_step(self.flat_p[lo:hi], self.flat_g[lo:hi], self.flat_m[lo:hi], self.flat_v[lo:hi], self.adam_state,
                      self.betas[0], self.betas[1], self.eps, grad_scale=grad_scale)          # optimizer.step()   train.py:201

    def _early_adam(self):
        """Backward hook (the last encoder stage is done): Adam over the flat tail on the weight-gradient side stream."""
        from . import functional as _fn
        def tail():
            ops.adam_tick(self.adam_state, self.base_lr, self.eta_min, self.t_max)
            self._adam_range(self._early_cut, self.flat_p.numel(), 1.0)
        # Behind the batch the flush point just set aside (same stream), ordered after main's in-place gradient accumulations.
        # NOT launched lazily like that batch: creating the Adam nodes after the main chain's next kernels was measured at
        # 15.3 ms per step against 13.3 (the graph executor's placement is sensitive to the creation order either way).
        _fn.launch_pending_flush()
        dev = self.flat_p.device
        main, side = torch.cuda.current_stream(dev), _fn._wgrad_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            tail()
        self.ctx.wside_used.add(dev)
        self._adam_tail_done = True

    def _early_adam_segment(self):
        """The same inside a segmented capture: called while the side segment of this flush point is being captured (the current
        stream IS the side stream, ordered after the main chain by the replay's events)."""
        ops.adam_tick(self.adam_state, self.base_lr, self.eta_min, self.t_max)
        self._adam_range(self._early_cut, self.flat_p.numel(), 1.0)
        self._adam_tail_done = True

    def _side_anchor(self):
        """Backward hook at the same point when there is no early Adam (data-parallel step, early_adam=False): ONE tiny node on the
        weight-gradient side stream, forked from the main chain right here.  Purely a placement aid for the captured graph:
        without a side node created eagerly at this point the replay runs the side batches almost serially with the main
        chain (single GPU 14.1 vs 12.9 ms; data-parallel layout 14.8 vs 13.5 ms), with it -- early Adam's first kernel, or this
        4 KB fill -- they overlap.  More than one such node, or one at another stage, measured worse (LABNOTES.md, rounds 2-3)."""
        from . import functional as _fn
        _fn.launch_pending_flush()
        dev = self.flat_p.device
        main, side = torch.cuda.current_stream(dev), _fn._wgrad_stream(dev)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            if self._anchor_buf is None:
                self._anchor_buf = torch.empty(1024, dtype=torch.float32, device=dev)
            ops.zero_(self._anchor_buf)
        self.ctx.wside_used.add(dev)

    def _update(self):
        """Un-overlapped form (eager steps): all-reduce(sum) the whole flat gradient, then Adam reads it as g / world."""
        if self.world > 1:
            self._allreduce_grads()
        self._adam(1.0 / self.world)

    # ------------------------------------------------------------------ data-parallel tail of a replayed step
    def _plan_split(self, items, ln_items=()):
        """After capturing forward + backward: the queued weight gradients (tensors in graph memory: same addresses at every
        replay) become a reusable launch plan; the flat gradient is cut into per-stage slices and each slice learns which
        grouped launch writes into it last."""
        self._wplan = ops.GroupedWgradPlan(items)
        self._lnplan = ops.LnFinishPlan(ln_items)               # LayerNorm gain / bias partials: one grouped finish, first
        names = [n for n, _ in self.model.named_parameters()]
        self._buckets = module_buckets(names, self.offsets, self.sizes, self.flat_g.numel())
        base = self.flat_g.data_ptr()
        writes = []
        for k, (dy, a, dw, db, sc, rps) in enumerate(items):
            for t in (dw, db):
                if t is not None:
                    writes.append((k // ops.GROUP_ITEMS, (t.data_ptr() - base) // 4, t.numel()))
        self._bucket_last = last_writer_per_bucket(self._buckets, writes)
        self._overlap = self._exchange_for(self._buckets, self._bucket_last)

    def _flush_and_reduce(self, then_update=False):
        """Launch the queued weight gradients group by group; all-reduce every slice of the flat gradient right after the
        launch that completes it (slices nothing queued writes to go first), so RCCL runs under the remaining launches
        (dist.OverlappedGradReduce -- the same object the 2-rank gloo test drives on CPU)."""
        from ._lib import block_region
        plan = self._wplan
        ngroups = (plan.n + ops.GROUP_ITEMS - 1) // ops.GROUP_ITEMS

        def launch_group(gi):
            with block_region():
                plan.launch(gi * ops.GROUP_ITEMS, min(ops.GROUP_ITEMS, plan.n - gi * ops.GROUP_ITEMS))

        def pre():
            with block_region():
                self._lnplan.launch()

        if then_update:
            self._overlap.step_tail(ngroups, launch_group, self._adam, pre)
        else:
            self._overlap.run(ngroups, launch_group, pre)

    def _step_impl(self, x, target):
        loss = self._fwd_bwd(x, target)
        self._update()
        return loss

    def _allreduce_grads(self):
        """Un-overlapped gradient exchange of an eager step: the same per-stage slices and the same arithmetic as the replayed
        step's (dist.OverlappedGradReduce.reduce_all), every slice at once; the 1/world goes into Adam."""
        mode = self.grad_wire
        ex = self._eager_exchange.get(mode)
        if ex is None:
            names = [n for n, _ in self.model.named_parameters()]
            buckets = module_buckets(names, self.offsets, self.sizes, self.flat_g.numel())
            ex = self._eager_exchange[mode] = self._exchange_for(buckets, [-1] * len(buckets))
        ex.reduce_all()

    def step(self, x, target):
        """Run one training step; returns the (device) loss of this rank's batch."""
        if not self.use_graph:
            loss = self._step_impl(x, target)
        elif self._graph is not None and not self._matches_static(x, target):
            # e.g. the smaller last batch of an epoch (the reference's loader has drop_last=False): the captured graph bakes in
            # the shapes, so this batch runs eagerly (same kernels, same update) instead of being broadcast into the static buffers
            loss = self._step_impl(x, target)
        else:
            if self._graph is None:
                self._capture(x, target)
            if x is not self._static[0]:                            # (a loader that fills input_buffers() itself skips the copies)
                self._static[0].copy_(x, non_blocking=True)
            if target is not self._static[1]:
                self._static[1].copy_(target, non_blocking=True)
            self._graph.replay()
            if self.split_step:                                     # weight-gradient groups, RCCL and Adam stay outside the graph
                self._flush_and_reduce(then_update=True)
            loss = self._static[2]
        self.steps_done += 1
        ops.PARAM_EPOCH[0] += 1                                     # (copies cached for engine-less forwards are stale now)
        return loss

    # ------------------------------------------------------------------ several steps per captured graph
    def _build_carry_groups(self):
        """Stage groups of the flat tail [cut, total) in the order a forward first reads them: for each, the id of the stage module
        whose flush point closes its gradients in the backward, the stage-entry count at which the forward reaches it, and its
        slices of the flat buffers.  None when the model does not have the MicFormer layout (step_many then runs step by step)."""
        sw = getattr(self.model, "swin", None)
        layers, ups = getattr(sw, "layers", None), getattr(sw, "up_layers", None)
        if sw is None or layers is None or ups is None or self._early_cut is None or len(ups) != len(layers):
            return None
        n = len(layers)
        if n < 4:
            # The carried batches + the tail's Adam are launched from the stage-entry-3 hook and the first group's wait is parked
            # for entry n: with fewer than 4 stages the last encoder stage would read its parameters before (n == 3: while) the
            # carried update runs.  step_many then keeps its promise by running step by step.
            return None
        names = [nm for nm, _ in self.model.named_parameters()]
        root = "swin"
        spec = [(layers[n - 1], n, [f"{root}.layers.{n - 1}.", f"{root}.norm."])]
        for j in range(n):
            pre = [f"{root}.up_layers.{j}."]
            if j + 1 < n:
                pre.append(f"{root}.concat_back_dim.{j + 1}.")
            else:
                pre += [f"{root}.norm2.", f"{root}.reverse_patch_embedding.", "out_conv.", f"{root}.concat_back_dim.0."]
            spec.append((ups[j], n + 1 + j, pre))
        groups, seen = [], set()
        for mod, entry, pre in spec:
            idx = [i for i, nm in enumerate(names) if any(nm.startswith(p) for p in pre)]
            seen.update(idx)
            ivs = sorted((self.offsets[i], self.offsets[i] + (self.sizes[i] + 3) // 4 * 4) for i in idx)
            ranges = []
            for lo, hi in ivs:
                if ranges and lo <= ranges[-1][1]:
                    ranges[-1][1] = max(ranges[-1][1], hi)
                else:
                    ranges.append([lo, hi])
            ranges = [(lo, min(hi, self.flat_p.numel())) for lo, hi in ranges]
            groups.append({"key": id(mod), "entry": entry, "ranges": ranges, "params": {id(self.params[i]) for i in idx}})
        tail = {i for i, o in enumerate(self.offsets) if o >= self._early_cut}
        if seen != tail or self._early_layer is not layers[n - 1]:
            return None
        return groups

    def _carry_prep(self, grp):
        """(forward shadow-weight plan | None, conv layout plan) of one stage group in the current arithmetic mode."""
        mode = ops.compute_dtype()
        cache = grp.setdefault("prep", {})
        if mode not in cache:
            _, fl_all, _, cl_all = self._prep_split[mode]
            mine = grp["params"]
            fl = None
            if fl_all is not None:
                fl = ops.WeightPrepPlan([t for t in fl_all.triples if id(self._owner_of(t[0])) in mine], blocked=True)
            cl = ops.Conv3PrepPlan([t for t in cl_all.triples if id(self._owner_of(t[0])) in mine])
            cache[mode] = (fl, cl)
        return cache[mode]

    def _owner_of(self, data):
        """The parameter whose storage a plan triple's source view is (plans hold p.data views of the flat buffer)."""
        off = (data.data_ptr() - self.flat_p.data_ptr()) // 4
        return self._param_at[off]

    def step_many(self, xs, targets):
        """k = len(xs) consecutive training steps (batch i = xs[i], targets[i]; one optimiser update each, in order) replayed from
        ONE HIP graph in which step i + 1's encoder forward runs beside the parameter-gradient work and the Adam update of step i's
        decoder / head / last encoder stage (functional.CARRY): that work otherwise queues up in front of the encoder's batches
        and drains in a tail behind every step.  Returns the k (device) losses.  Single GPU, graph mode; the result is the same
        sequence of updates as k calls of step()."""
        k = len(xs)
        if self._carry_groups is None:
            self._carry_groups = self._build_carry_groups() or False
        from . import functional as _fn
        # (the carry region needs what early Adam needs: every flush point launches everything queued so far)
        full_flush = (self.defer_wgrad and self.flush_points and not self.split_step and _fn.FLUSH_MAX_TOKENS >= (1 << 30))
        if k < 2 or not self.use_graph or self.world != 1 or not self._carry_groups or not full_flush:
            return self._step_by_step(xs, targets)
        ent = self._many
        if ent is not None and not (ent["k"] == k and ent["mode"] == ops.arith_mode() and all(
                x.shape == sx.shape and x.dtype == sx.dtype and type(x) is type(sx) and t.shape == st.shape and t.dtype == st.dtype
                for x, t, sx, st in zip(xs, targets, ent["x"], ent["t"]))):
            return self._step_by_step(xs, targets)
        if ent is None:
            ent = self._many = self._capture_many(xs, targets)
        for sx, x in zip(ent["x"], xs):
            sx.copy_(x, non_blocking=True)
        for st, t in zip(ent["t"], targets):
            st.copy_(t, non_blocking=True)
        ent["graph"].replay()
        self.steps_done += k
        ops.PARAM_EPOCH[0] += 1
        return list(ent["loss"])

    def _step_by_step(self, xs, targets):
        """step_many's fallback: k plain steps.  (A replayed step() hands out its ONE static loss tensor: copy each.)"""
        return [self.step(x, t).clone() for x, t in zip(xs, targets)]

    def _many_body(self, sxs, sts):
        """The k steps as one launch sequence: step i hands the first flush points' batches + the tail's Adam to step i + 1."""
        losses, carry = [], None
        for i, (x, t) in enumerate(zip(sxs, sts)):
            last = i == len(sxs) - 1
            losses.append(self._fwd_bwd(x, t, carry_in=carry, carry_out=not last))
            if last:
                self._update()                                      # (its own early Adam ran under its backward)
                carry = None
            else:
                ops.adam_tick(self.adam_state, self.base_lr, self.eta_min, self.t_max)
                self._adam_range(0, self._early_cut, 1.0)           # [cut, total) follows at the head of step i + 1
                carry = self._carried
        return losses

    def _capture_many(self, xs, targets):
        sxs, sts = [x.clone() for x in xs], [t.clone() for t in targets]
        keep = [t.clone() for t in (self.flat_p, self.flat_m, self.flat_v, self.adam_state)]
        dp_rng = self._drop_path_rng_tensors(sxs[0])
        dp_keep = [t.clone() for t in dp_rng]
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(sxs[0].device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(side):
                self._step_impl(sxs[0], sts[0])                     # (sizes scratch buffers / allocator pools: undone below)
                self._many_body(sxs, sts)
        finally:
            # the warm-up ran k + 1 real updates: undo them whether or not it finished (a raise must not leave the parameters,
            # the Adam state, the RNG streams or the carry switch behind)
            from . import functional as _fn0
            self.ctx.carry["on"], self.ctx.carry["open"], self.ctx.carry["stash"] = False, False, []
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            with torch.no_grad():
                for dst, src in zip((self.flat_p, self.flat_m, self.flat_v, self.adam_state), keep):
                    dst.copy_(src)
                for dst, src in zip(dp_rng, dp_keep):
                    dst.copy_(src)
            torch.set_rng_state(rng_cpu)
            torch.cuda.set_rng_state(rng_dev, sxs[0].device)
            del keep
        if self.segmented:
            from . import functional as _fn
            torch.cuda.empty_cache()
            g = _fn.StepSegmenter(_fn._wgrad_stream(sxs[0].device))
            cs = torch.cuda.Stream(device=sxs[0].device)
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                self.ctx.segmenter = g
                try:
                    g.begin()
                    with torch.autograd.set_multithreading_enabled(False):
                        losses = self._many_body(sxs, sts)
                    g.finish()
                finally:
                    self.ctx.segmenter = None
            torch.cuda.current_stream().wait_stream(cs)
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                losses = self._many_body(sxs, sts)
        self._carried = None
        return {"k": len(sxs), "graph": g, "x": sxs, "t": sts, "loss": losses, "mode": ops.arith_mode()}

    def input_buffers(self):
        """(x, target) device buffers the captured step reads, or None before the first graph step.  A data pipeline that writes its
        batches straight into them (H2D copies, or the device-side input tail) and passes THESE objects to step() saves the
        device-to-device staging copies of every step (201 MB at base / 128^3 / batch 2: ~0.1 ms)."""
        return None if self._static is None else (self._static[0], self._static[1])

    def _matches_static(self, x, target):
        sx, st = self._static[0], self._static[1]
        ok = x.shape == sx.shape and target.shape == st.shape and x.dtype == sx.dtype and target.dtype == st.dtype \
            and self._static_mode == ops.arith_mode()            # (the arithmetic mode is baked into the captured launches)
        # a plain tensor and a data.RawBatch (and a RawBatch with / without augmentation draws) take different launches in the
        # patch embedding: the captured graph holds exactly one of them, anything else runs eagerly
        ok = ok and type(x) is type(sx) and (getattr(x, "params", None) is None) == (getattr(sx, "params", None) is None)
        if self.world > 1 and not self.uniform_batches:
            # Data parallel: the replayed step and the eager step cut the gradient exchange differently (per-stage slices vs
            # whole-buffer buckets), so every rank must take the same path: ONE small all-reduce(MIN) of the flag per step.
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=x.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.sync.pg)
            ok = bool(flag.item())
        elif self.world > 1 and not ok:
            # uniform_batches promised one batch shape on every rank and skipped the agreement above: a rank that fell back to the
            # eager step alone would cut the gradient exchange differently from its peers (a hang, or silently wrong sums)
            raise RuntimeError("TrainEngine(uniform_batches=True): this rank's batch does not match the captured step "
                               f"(x {tuple(x.shape)} vs {tuple(sx.shape)}, target {tuple(target.shape)} vs {tuple(st.shape)}, mode "
                               f"{ops.arith_mode()} vs {self._static_mode}); ragged batches need uniform_batches=False")
        return ok

    def _capture(self, x, target):
        """Warm up eagerly on a side stream, then capture ONE step into a HIP graph.  The warm-up runs real kernels (it sizes the
        scratch buffers and the allocator pools) but must not train: parameters, Adam moments, the device step counter / LR and
        the RNG streams are snapshotted before and restored after, so the first step() applies exactly one update to its batch
        (the reference does one optimizer.step() + scheduler.step() per batch, train.py:200-207)."""
        sx, st = x.clone(), target.clone()
        keep = [t.clone() for t in (self.flat_p, self.flat_m, self.flat_v, self.adam_state)]
        # (the DropPath stream lives on the device -- seed + counter, created on first use from torch's CPU generator: create it
        # BEFORE the snapshot so the warm-up neither draws its seed nor advances its counter for good)
        dp_rng = self._drop_path_rng_tensors(sx)
        dp_keep = [t.clone() for t in dp_rng]
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(sx.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                self._step_impl(sx, st)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for dst, src in zip((self.flat_p, self.flat_m, self.flat_v, self.adam_state), keep):
                dst.copy_(src)
            for dst, src in zip(dp_rng, dp_keep):
                dst.copy_(src)
        torch.set_rng_state(rng_cpu)
        torch.cuda.set_rng_state(rng_dev, sx.device)
        del keep
        # single GPU: the whole step; data parallel: forward + backward only (weight-gradient groups, collective and Adam
        # are issued eagerly after the replay)
        body = (lambda: self._fwd_bwd(sx, st, flush=False)) if self.split_step else (lambda: self._step_impl(sx, st))
        if self.segmented:
            from . import functional as _fn
            torch.cuda.empty_cache()
            g = _fn.StepSegmenter(_fn._wgrad_stream(sx.device))
            cs = torch.cuda.Stream(device=sx.device)
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                self.ctx.segmenter = g
                try:
                    g.begin()
                    sl = body()
                    g.finish()
                finally:
                    self.ctx.segmenter = None
            torch.cuda.current_stream().wait_stream(cs)
        else:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                sl = body()
        if self.split_step:
            from . import functional as _fn
            with _fn.use_context(self.ctx):
                self._plan_split(*_fn.take_deferred())              # (capture records, it does not run: step() replays next)
        self._graph, self._static, self._static_mode = g, (sx, st, sl), ops.arith_mode()

    def _drop_path_rng_tensors(self, x):
        """The device-side DropPath RNG states ({seed, counter}) of the model, created now if the model is in train mode and has
        not drawn yet."""
        out = []
        for m in self.model.modules():
            pre = getattr(m, "_predraw_drop_path", None)
            if pre is None:
                continue
            if m.training and not m.__dict__.get("_dp_keep_cache", {}).get(x.device):
                pre(x.shape[0], x.device)                           # creates the entry (one draw; restored below)
                ent = m.__dict__.get("_dp_keep_cache", {}).get(x.device)
                if ent is not None:
                    ent[1][1] = 0                                   # counter back to the fresh state
            ent = m.__dict__.get("_dp_keep_cache", {}).get(x.device)
            if ent is not None:
                out.append(ent[1])
        return out

    # ------------------------------------------------------------------ checkpoints (utils.py:57-65, 108-138)
    def optimizer_state_dict(self):
        """State of the fused Adam in `torch.optim.Adam.state_dict()` layout (what train.py:233-241 saves under 'optimizer'),
        so a checkpoint written here resumes under the reference's optimizer and vice versa."""
        step = int(self.adam_state[0].item())
        state = {}
        if step > 0:
            for i, (o, n, p) in enumerate(zip(self.offsets, self.sizes, self.params)):
                state[i] = {"step": torch.tensor(float(step)), "exp_avg": self.flat_m[o:o + n].view(p.shape).clone(),
                            "exp_avg_sq": self.flat_v[o:o + n].view(p.shape).clone()}
        group = {"lr": self.scheduled_lr(step), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "initial_lr": self.base_lr, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        steps = {int(float(v["step"])) for v in sd["state"].values()}
        if len(steps) > 1:
            raise ValueError("per-parameter Adam step counts differ; the fused optimizer keeps one")
        with torch.no_grad():
            self.flat_m.zero_()
            self.flat_v.zero_()
            for i, st in sd["state"].items():
                o, n = self.offsets[int(i)], self.sizes[int(i)]
                self.flat_m[o:o + n].copy_(st["exp_avg"].reshape(-1))
                self.flat_v[o:o + n].copy_(st["exp_avg_sq"].reshape(-1))
            self.adam_state[0] = steps.pop() if steps else 0
        g = sd["param_groups"][0]
        self.betas, self.eps = tuple(g["betas"]), g["eps"]
        self.base_lr = g.get("initial_lr", self.base_lr)
        self._graph = self._many = None                         # betas / eps / lr constants are baked into a captured step

    def scheduler_state_dict(self):
        """CosineAnnealingLR.state_dict() layout (stepped once per iteration, train.py:148, 206-207)."""
        step = int(self.adam_state[0].item())
        return {"T_max": self.t_max, "eta_min": self.eta_min, "base_lrs": [self.base_lr], "last_epoch": step,
                "_step_count": step + 1, "_last_lr": [self.scheduled_lr(step)]}

    def load_scheduler_state_dict(self, sd):
        self.t_max, self.eta_min = sd["T_max"], sd["eta_min"]
        self.base_lr = sd["base_lrs"][0]
        self._graph = self._many = None

    def checkpoint(self, epoch):
        """The dict the reference writes with torch.save (train.py:233-241): epoch, state_dict, optimizer, scheduler."""
        return {"epoch": epoch, "state_dict": {k: v.detach().clone() for k, v in self.model.state_dict().items()},
                "optimizer": self.optimizer_state_dict(), "scheduler": self.scheduler_state_dict()}

    def load_checkpoint(self, ckpt):
        """Counterpart of utils.reload_ckpt (utils.py:108-122).  Parameters stay views of the flat buffer."""
        with torch.no_grad():
            own = self.model.state_dict()
            missing = set(own) ^ set(ckpt["state_dict"])
            if missing:
                raise KeyError(f"state_dict keys differ: {sorted(missing)[:5]} ...")
            for k, v in ckpt["state_dict"].items():
                own[k].copy_(v)
        ops.PARAM_EPOCH[0] += 1                                     # weight copies cached for engine-less forwards are stale
        if "optimizer" in ckpt:
            self.load_optimizer_state_dict(ckpt["optimizer"])
        if "scheduler" in ckpt:
            self.load_scheduler_state_dict(ckpt["scheduler"])
        return ckpt.get("epoch", 0)

    # ------------------------------------------------------------------ helpers
    def lr(self):
        """The learning rate the LAST optimiser step used (device-side schedule: cosine at scheduler epoch step - 1)."""
        st = self.adam_state.cpu()
        return float(st[1:2].view(torch.float64)[0])

    def scheduled_lr(self, epoch):
        """CosineAnnealingLR closed form at scheduler epoch `epoch`: after N iterations torch has called scheduler.step() N
        times, so param_groups[0]['lr'] and _last_lr hold cosine(N) -- the rate the NEXT step will use (train.py:206-207)."""
        import math
        return self.eta_min + (self.base_lr - self.eta_min) * (1.0 + math.cos(math.pi * epoch / self.t_max)) / 2.0
