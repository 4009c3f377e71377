"""One GS-LoRA forgetting step on the HIP path (shared by engine_cl.py and engine.py).

Reference loop body: engine_cl.py:59-125 / engine.py:237-330 —
    total = beta * relu(BND - CE_forget) + CE_remain + alpha * L_structure
            + w_f * relu(BND_pro - KL_forget) + w_r * KL_remain
Data parallel (one process per GPU, torch.distributed over RCCL): the two hinge arguments are
batch MEANS, so their global values are all-reduced (one packed 8-float message) before the
hinges are evaluated — that keeps the single-GPU / nn.DataParallel semantics of the reference —
and the flat LoRA gradient bucket (0.94 MiB for ViT-P8S8 r=8) is sum-all-reduced after backward.
"""
import os

import torch
import torch.distributed as dist
import torch.nn as nn

from . import losses, ops

# launch-bound batches, one process: the loss section of the step as one launch (gsl_loss_tail); False = the separate kernels
# (a decided choice, no environment read: tests/test_hip_graph.py patches the attribute to pin the two forms against each other)
LOSS_TAIL = True
# rows (remain + forget images) up to which it is used: the one workgroup handles 16 rows at a time (92 us at 96 rows of a 768-wide
# embedding — no better than the ~19 short launches it replaces; 4+4 images: one pass)
LOSS_TAIL_ROWS = 32


def _world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _dp_active():
    """The data-parallel form of the step (packed scalar all-reduce, gradient all-reduce, graph segments) runs when the process group has
    more than one rank. (tests/test_hip_graph.py patches this to drive the same form through a ONE-rank RCCL group: every collective is
    then an identity, so the results must equal the single-process step.)"""
    return _world() > 1


def _cat_labels(y_r, y_f):
    """torch.cat((y_r, y_f)) — or, when the two are adjacent views of one buffer (GraphedStep lays its static label buffers out that way),
    a view over both: one launch less per replayed step."""
    if (y_r.dim() == 1 and y_f.dim() == 1 and y_r.dtype == y_f.dtype and y_r.device == y_f.device and y_r.is_contiguous() and y_f.is_contiguous()
            and y_r.untyped_storage().data_ptr() == y_f.untyped_storage().data_ptr()
            and y_f.storage_offset() == y_r.storage_offset() + y_r.numel()):
        return y_r.as_strided((y_r.numel() + y_f.numel(),), (1,))
    return torch.cat((y_r, y_f), 0)


def _plain_ce(criterion):
    return (type(criterion) is nn.CrossEntropyLoss and criterion.weight is None and criterion.reduction == "mean"
            and getattr(criterion, "label_smoothing", 0.0) == 0.0 and criterion.ignore_index == -100)


def _globalize(local_sum, global_sum_detached):
    """value = global sum, gradient = d(local sum): the straight-through form used for DP means."""
    return local_sum + (global_sum_detached - local_sum.detach())


class HipBackend:
    """The loss / gradient-bucket operations of the step on the HIP kernels (the only product backend).
    tests/test_ddp_gloo.py injects a CPU stand-in with the same methods to exercise the data-parallel
    arithmetic of gs_lora_step under gloo; nothing in the package ever selects another backend."""
    ce_sum_top1 = staticmethod(losses.ce_sum_top1)
    proto_kl_sum = staticmethod(losses.proto_kl_sum)
    ce_sum_top1_split = staticmethod(losses.ce_sum_top1_split)
    proto_kl_sum_split = staticmethod(losses.proto_kl_sum_split)
    structure_loss = staticmethod(losses.structure_loss)
    combine = staticmethod(losses.combine)
    combine_pack = staticmethod(losses.combine_pack)
    loss_tail = staticmethod(ops.loss_tail)
    loss_tail_max_rows = staticmethod(ops.loss_tail_max_rows)

    @staticmethod
    def grad_bucket(net):
        bucket = net.lora_bucket()
        bucket.attach_grads()
        return bucket.grad

    @staticmethod
    def early_grad_slice(net):
        """(flat gradient bucket, first element of block 1): the LoRA gradients of blocks 1 .. L-1 are final once the backward chain
        reaches block 0, so their slice of the bucket can be all-reduced while block 0's backward still runs."""
        b = net.lora_bucket()
        if b is None or b.ngroups_block < 2:
            return None
        return b.grad, b.offsets[b.per_layer]


def _bucket_messages(net, backend):
    """The gradient exchange of one data-parallel step as an ordered list of slices of the flat LoRA-gradient bucket. EVERY mode of the
    step (eager with the overlapped first message, eager without overlap, HIP-graph segments) issues exactly these all-reduces in this
    order, so ranks that momentarily run different modes (one replays a captured graph, another runs its first eager step of a new
    configuration, a third fell back after a failed capture) still post matching collectives."""
    sl = backend.early_grad_slice(net) if hasattr(backend, "early_grad_slice") else None
    if sl is None:
        return [backend.grad_bucket(net)]
    flat, split = sl
    return [flat[split:], flat[:split]]      # blocks 1 .. L-1 (final once the backward reaches block 0), then block 0


def _guard_message(net):
    """fp16 operands under data parallelism: the overflow guard of the backward that just ran (ViTRunner.overflow_guard(), one device float)
    as the int32 word every rank MAX-reduces after its gradient messages — non-negative floats, +inf and NaN order as their bit patterns, so
    an integer MAX agrees on "some rank saturated" whatever the backend does with a floating NaN. All ranks then skip (or take) the update
    together and settle the same next loss-scale exponent from the same value. None when the backward did not run on loss-scaled gradients
    (every rank of a job computes in the same format, so either all ranks post this message or none does)."""
    runner = getattr(net, "_runner", None)
    g = runner.overflow_guard() if runner is not None and hasattr(runner, "overflow_guard") else None
    return None if g is None else g[:1].view(torch.int32)


def _post_guard(net):
    gm = _guard_message(net)
    if gm is not None:
        dist.all_reduce(gm, op=dist.ReduceOp.MAX)


class _OverlappedBucketReduce:
    """Gradient all-reduce in two messages: blocks 1..L-1 on a side stream as soon as block 1's gradients are written (it runs
    under block 0's FFN backward), block 0 afterwards. The structure-loss gradient is parameter-only and pre-scaled by 1/world; its
    autograd node is created after the network's, so it has been added to the bucket before the network backward starts.
    `overlap` is False when the step runs two network backwards (fuse_batches=False: the hook would fire during the first one while
    the second still accumulates into the slice): the same two messages are then posted after the backward."""

    def __init__(self, net, backend, overlap=True):
        self.net, self.backend, self.work, self.side, self.runner = net, backend, None, None, None
        # the two-slice form is known up front (the bucket exists); a backend without slices hands over its bucket after the backward
        self.msgs = _bucket_messages(net, backend) if hasattr(backend, "early_grad_slice") else None
        runner = net.runner() if hasattr(net, "runner") else None
        if overlap and self.msgs is not None and len(self.msgs) == 2 and runner is not None:
            self.runner = runner
            runner.grad_hook = self._hook

    def _hook(self, layer):
        if layer != 1 or self.work is not None:
            return
        first = self.msgs[0]
        if first.is_cuda:
            if self.side is None:
                self.side = _side_stream(first.device)
            self.side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.side):
                self.work = dist.all_reduce(first, async_op=True)
        else:
            self.work = dist.all_reduce(first, async_op=True)

    def cancel(self):
        """The backward raised: take the hook off the runner (an in-flight first message is waited for, nothing else is posted)."""
        if self.runner is not None:
            self.runner.grad_hook = None
        if self.work is not None:
            self.work.wait()

    def finish(self):
        if self.runner is not None:
            self.runner.grad_hook = None
        if self.work is None:        # no overlap (single block, two backwards, or a hook that never fired): the same messages, in order
            for m in (self.msgs if self.msgs is not None else _bucket_messages(self.net, self.backend)):
                dist.all_reduce(m)
            _post_guard(self.net)
            return
        dist.all_reduce(self.msgs[1])
        _post_guard(self.net)
        self.work.wait()
        if self.msgs[0].is_cuda:
            torch.cuda.current_stream().wait_stream(self.side)


_side_streams = {}


def _side_stream(device):
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = torch.cuda.Stream(device=device)
    return s


def _check_not_replicated(model):
    """nn.DataParallel over several devices replicates the module per forward; the replicas' hand-written backward would write into
    throw-away gradient buckets and the real LoRA parameters would only ever see the structure-loss gradient. The multi-GPU mode of
    this build is one process per GPU under torch.distributed."""
    if isinstance(model, nn.DataParallel) and len(model.device_ids or []) > 1:
        raise RuntimeError("gs-lora_amd: nn.DataParallel over several devices is not supported by the HIP path (replicas cannot "
                           "back-propagate into the flat LoRA gradient bucket). Launch one process per GPU with torch.distributed "
                           "(python -m torch.distributed.run --nproc-per-node N ...): gs_lora_step is data-parallel there.")


class _EagerComm:
    """The two exchanges of a data-parallel step, issued eagerly (the gradient one overlapped with the end of the backward)."""

    @staticmethod
    def all_reduce_scalars(pack):
        dist.all_reduce(pack)

    @staticmethod
    def bucket_reducer(net, backend, overlap=True):
        return _OverlappedBucketReduce(net, backend, overlap)


def _arm_overflow_guard(net, optimizer):
    """fp16 operands: hand the optimizer the overflow guard of the backward that just ran (ViTRunner.overflow_guard(): the device float the
    LayerNorm backwards raised to the largest scaled gradient they saw) — FusedAdamW then skips the update of a step in which a 16-bit gradient
    store saturated (torch.cuda.amp.GradScaler.step semantics, no host sync); the next backward lowers its loss scale on the device. Under
    data parallelism the ranks have agreed on the guard by then (one 4-byte MAX all-reduce behind the gradient messages, _guard_message):
    they skip together, and they lower their scales together."""
    if not hasattr(optimizer, "overflow_guard"):
        return
    runner = getattr(net, "_runner", None)
    g = runner.overflow_guard() if runner is not None and hasattr(runner, "overflow_guard") else None
    optimizer.overflow_guard = g


def gs_lora_step(model, optimizer, criterion, x_r, y_r, x_f, y_f, *, beta, alpha, BND, use_structure=True,
                 group_type="block", use_prototype=False, proto_table=None, w_f=0.0, w_r=0.0, BND_pro=0.0,
                 backend=HipBackend, fuse_batches=True, _comm=_EagerComm):
    """Runs forward x2, the three-term loss, backward, gradient all-reduce and optimizer.step().
    Returns a packed DEVICE tensor of the 8 meter values (no host sync here):
      [beta*loss_forget, loss_remain, total, alpha*structure, top1_forget%, top1_remain%,
       w_f*relu(BND_pro-KL_f), w_r*KL_r]"""
    _check_not_replicated(model)
    net = model.module if isinstance(model, nn.DataParallel) else model
    world = _world()
    dev = x_r.device
    split = None
    if fuse_batches:
        # every operation of the network is per-sample (no BatchNorm), so one forward over the concatenated batch is
        # arithmetically identical to the reference's two forwards and halves the number of kernel launches / tile tails
        nr = x_r.size(0)
        y_all = _cat_labels(y_r, y_f)
        # (the HIP model takes the two image batches as a tuple and patchifies each into its row range: no 2 x 77 MB concatenated
        #  copy at batch 512 + 512; any other module gets the concatenated tensor)
        both = (x_r, x_f) if getattr(net, "accepts_batch_tuple", False) and x_r.shape[1:] == x_f.shape[1:] else torch.cat((x_r.float(), x_f.float()), 0)
        out, emb = model(both, y_all)
        out_r, out_f, emb_r, emb_f = out[:nr], out[nr:], emb[:nr], emb[nr:]
        if _plain_ce(criterion) and hasattr(backend, "ce_sum_top1_split"):
            split = (out, emb, y_all, nr)       # losses on the two row ranges of the un-sliced tensors (one gradient buffer each)
    else:
        out_r, emb_r = model(x_r.float(), y_r)
        out_f, emb_f = model(x_f.float(), y_f)
    n_r, n_f = float(x_r.size(0)), float(x_f.size(0))
    if (split is not None and LOSS_TAIL and not _dp_active() and hasattr(backend, "loss_tail") and out.dtype == torch.float32
            and out.is_contiguous() and out.dim() == 2 and 0 < nr < out.shape[0] <= min(LOSS_TAIL_ROWS, backend.loss_tail_max_rows()) and out.shape[1] <= 1024
            and (not use_prototype or emb.shape[1] <= 1024)
            and y_all.dtype == torch.int64 and y_all.device == out.device
            and (not use_prototype or (emb.dtype == torch.float32 and emb.is_contiguous() and proto_table is not None))):
        # launch-bound batches, one process: the whole loss section — CE / KL rows and sums of both row ranges, hinges, meters, and the
        # backward down to dlogits / demb — is ONE launch (gsl_loss_tail) instead of ~20; the gradients enter the network's autograd
        # node directly, the group-lasso node gets its coefficient alpha. Same values as the multi-launch path below, bit for bit.
        structure = backend.structure_loss(net, group_type, grad_scale=1.0) if use_structure else None
        total, meters, coefs, dlogits, demb = backend.loss_tail(out.detach(), y_all, nr, emb.detach() if use_prototype else None,
                                                                proto_table if use_prototype else None,
                                                                None if structure is None else structure.detach(), beta, BND, alpha,
                                                                w_f, w_r, BND_pro)
        optimizer.zero_grad()
        roots, grads = [out], [dlogits]
        if use_prototype:
            roots.append(emb)
            grads.append(demb)
        if structure is not None:
            roots.append(structure)
            grads.append(coefs[4])
        torch.autograd.backward(roots, grads)
        _arm_overflow_guard(net, optimizer)
        optimizer.step()
        return meters
    if split is not None:
        ce_r_sum, hit_r, ce_f_sum, hit_f = backend.ce_sum_top1_split(split[0], split[2], split[3])
    elif _plain_ce(criterion):
        ce_r_sum, hit_r = backend.ce_sum_top1(out_r, y_r)
        ce_f_sum, hit_f = backend.ce_sum_top1(out_f, y_f)
    else:   # exotic criterion: keep its semantics (mean over the local batch), top-1 from the HIP kernel
        ce_r_sum = criterion(out_r, y_r) * n_r
        ce_f_sum = criterion(out_f, y_f) * n_f
        hit_r = backend.ce_sum_top1(out_r.detach(), y_r)[1]
        hit_f = backend.ce_sum_top1(out_f.detach(), y_f)[1]
    if use_prototype and split is not None:
        kl_f_sum, kl_r_sum = backend.proto_kl_sum_split(split[1], split[2], proto_table, split[3])
    elif use_prototype:
        kl_f_sum = backend.proto_kl_sum(emb_f, y_f, proto_table)
        kl_r_sum = backend.proto_kl_sum(emb_r, y_r, proto_table)
    else:
        kl_f_sum = kl_r_sum = None
    # the scalar tail (hinges, weighted sum, meters, the five upstream gradients) is ONE kernel forward and one 5-element multiply
    # backward instead of ~35 one-element torch kernels (losses.combine / gsl_loss_combine[_pack])
    if not _dp_active():
        structure = backend.structure_loss(net, group_type, grad_scale=1.0) if use_structure else None
        total, meters = backend.combine(ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, hit_r, hit_f, n_r, n_f, beta, BND, alpha,
                                        w_f, w_r, BND_pro)
        optimizer.zero_grad()
        total.backward()
        _arm_overflow_guard(net, optimizer)
        optimizer.step()
        return meters
    # data parallel: the eight batch sums travel in ONE packed all-reduce, so the hinges see the GLOBAL batch means. Every entry is
    # detached: the collective must not become a node of the autograd graph. The two batch sizes ride along as device scalars.
    f = lambda t: t.detach().float()
    kz = torch.zeros((), device=dev) if not use_prototype else None
    pack = torch.stack([f(ce_r_sum), f(ce_f_sum), f(hit_r), f(hit_f), torch.full((), n_r, device=dev), torch.full((), n_f, device=dev),
                        f(kl_f_sum) if use_prototype else kz, f(kl_r_sum) if use_prototype else kz])
    _comm.all_reduce_scalars(pack)
    structure = backend.structure_loss(net, group_type, grad_scale=1.0 / world) if use_structure else None
    total, meters = backend.combine_pack(pack, ce_r_sum, ce_f_sum, kl_f_sum, kl_r_sum, structure, beta, BND, alpha, w_f, w_r, BND_pro)
    optimizer.zero_grad()
    reducer = _comm.bucket_reducer(net, backend, overlap=fuse_batches)      # two forwards = two network backwards: no early message
    try:
        total.backward()
    except BaseException:
        reducer.cancel()
        raise
    reducer.finish()
    _arm_overflow_guard(net, optimizer)
    optimizer.step()
    return meters


class _SegmentedCapture:
    """Captures a data-parallel step as HIP-graph SEGMENTS around its two collectives: [forward, loss sums, pack] | all-reduce(pack) |
    [scalar tail, backward] | all-reduce(gradient bucket) | [AdamW, meters]. The collectives stay eager (any backend; nothing of RCCL is
    captured), the ~190-370 kernel launches between them are replayed. The capture pass itself sends nothing; replays post the eager
    step's messages in the eager step's order (_bucket_messages), so ranks in different modes stay matched. All segments share one
    memory pool."""

    def __init__(self):
        self.pool = torch.cuda.graph_pool_handle()
        self.graphs, self.colls, self._ctx = [], [], None

    def begin(self):
        g = torch.cuda.CUDAGraph()
        self._ctx = torch.cuda.graph(g, pool=self.pool, capture_error_mode="thread_local")
        self._ctx.__enter__()
        self.graphs.append(g)

    def end(self, exc=(None, None, None)):
        ctx, self._ctx = self._ctx, None
        if ctx is not None:
            ctx.__exit__(*exc)

    def _cut(self, tensors, maxed=()):
        """Close the current segment; `tensors` are all-reduced (sum), then `maxed` (max), in order, between it and the next one at every replay. Nothing is sent
        during the capture pass itself (no kernel runs while capturing, the buffers hold no values yet): a rank that captures posts
        exactly one step's worth of collectives — those of the replay that follows — like a rank that runs the same step eagerly."""
        self.end()
        self.colls.append([(t, dist.ReduceOp.SUM) for t in tensors] + [(t, dist.ReduceOp.MAX) for t in maxed])
        self.begin()

    def all_reduce_scalars(self, pack):
        self._cut([pack])

    def bucket_reducer(self, net, backend, overlap=True):
        cap = self

        class _R:
            def finish(self_inner):
                gm = _guard_message(net)
                cap._cut(_bucket_messages(net, backend), [] if gm is None else [gm])      # the eager step's messages, in its order

            def cancel(self_inner):
                pass
        return _R()

    def replay(self):
        for i, g in enumerate(self.graphs):
            g.replay()
            if i < len(self.colls):
                for t, op in self.colls[i]:
                    dist.all_reduce(t, op=op)


class GraphedStep:
    """gs_lora_step captured ONCE per configuration as a HIP graph and replayed: below ~64 images per forward the eager step is
    bound by the ~370 kernel launches it makes from Python (4.7 ms at batch 4+4 and 5.9 ms at the reference's batch 48+48 on
    MI355X), not by the GPU. Everything that varies from step to step is read from device memory by the captured kernels: the batch
    (static input buffers), the dropout seed (GSL_SEED_ON_DEVICE), AdamW's step count and learning rate (gsl_adamw_flat_dev).
    Everything else is part of the key — shapes, the loss hyper-parameters, the prototype table, train/eval state, the versions of
    the frozen weights (eval()/train() merge round trips, load_state_dict) — and a key change falls back to one eager step (which
    also refreshes the operand caches) followed by a fresh capture. Replays are bit-identical to eager steps (same kernels, same
    seeds; tests/test_hip_graph.py). Under torch.distributed (one process per GPU) the step is captured as three graph segments
    around its two eager collectives (_SegmentedCapture): the few-shot / batch-48 regimes are launch-bound on every rank."""

    MAX_GRAPHS = 3      # captured configurations kept (e.g. the regular batch and the ragged last batches of the two loaders)

    def __init__(self, model, optimizer, criterion):
        self.model, self.optimizer, self.criterion = model, optimizer, criterion
        self.net = model.module if isinstance(model, nn.DataParallel) else model
        self._frozen = [p for n, p in self.net.named_parameters() if "lora_" not in n]
        self.graphs = {}            # key -> dict(graph, static, nfwd), insertion-ordered (oldest evicted)
        self.pending = None         # key seen once (ran eagerly); captured at its second sighting
        self.failed = set()         # keys whose capture failed: they run eagerly
        self.seed_dev, self._seed_val = None, None
        self.replays = self.captures = self.eager_steps = 0

    def _key(self, x_r, y_r, x_f, y_f, kw):
        pt = kw.get("proto_table")
        # everything a captured kernel launch bakes in by value: shapes, loss hyper-parameters, train/eval state, frozen weights, and the
        # optimizer's betas / eps / weight decay and the dropout rates (lr and the step count are read from device memory)
        hyper = tuple((g["betas"], g["eps"], g["weight_decay"]) for g in self.optimizer.param_groups)
        drop = (getattr(self.net, "dropout_p", None), getattr(self.net, "emb_dropout_p", None))
        return (tuple(x_r.shape), tuple(x_f.shape), x_r.dtype, y_r.dtype, self.net.training, self.net.compute_dtype,
                sum(p._version for p in self._frozen), None if pt is None else (pt.data_ptr(), tuple(pt.shape)),
                tuple(sorted((k, v) for k, v in kw.items() if k != "proto_table")), hyper, drop)

    def _usable(self):
        return (hasattr(self.optimizer, "graph_capturable") and _plain_ce(self.criterion) and not isinstance(self.model, nn.DataParallel))

    def _eager(self, x_r, y_r, x_f, y_f, kw):
        self.eager_steps += 1
        return gs_lora_step(self.model, self.optimizer, self.criterion, x_r, y_r, x_f, y_f, **kw)

    def __call__(self, x_r, y_r, x_f, y_f, **kw):
        if not self._usable():
            return self._eager(x_r, y_r, x_f, y_f, kw)
        key = self._key(x_r, y_r, x_f, y_f, kw)
        if key in self.failed:
            return self._eager(x_r, y_r, x_f, y_f, kw)
        ent = self.graphs.get(key)
        if ent is None:
            if key != self.pending or not self.optimizer.graph_capturable():
                # first sighting of this configuration: one eager step (warms operand caches, optimizer state, gradient bucket)
                self.pending = key
                return self._eager(x_r, y_r, x_f, y_f, kw)
            try:
                ent = self._capture(x_r, y_r, x_f, y_f, kw)
            except Exception as exc:      # a capture that cannot complete (e.g. another thread touched the allocator): stay eager for this key
                import warnings
                warnings.warn(f"gs-lora_amd: HIP-graph capture of the step failed ({type(exc).__name__}: {exc}); this configuration runs eagerly")
                self.failed.add(key)
                self.pending = None
                return self._eager(x_r, y_r, x_f, y_f, kw)
            frozen_now = key[6]
            for k in [k for k in self.graphs if k[6] != frozen_now]:      # graphs captured against older frozen weights are stale
                del self.graphs[k]
            while len(self.graphs) >= self.MAX_GRAPHS:
                del self.graphs[next(iter(self.graphs))]
            self.graphs[key] = ent
            self.pending = None
        # the captured kernels read the batch from static buffers; a caller that fills those buffers itself (static_inputs(): the H2D copy
        # of a prefetcher, or device-resident synthetic data) hands them back and no staging copy is launched
        for dst, src in zip(ent["static"][:4], (x_r, y_r, x_f, y_f)):
            if dst is not src:
                dst.copy_(src, non_blocking=True)
        # device-resident step state: AdamW (step, lr) and the dropout seed follow the host values
        self.optimizer.graph_sync()
        r = self.net.runner()
        want = (r.drop_seed << 20) + r.drop_calls
        if self._seed_val != want:
            self.seed_dev.fill_(want)
            self._seed_val = want
        ent["graph"].replay()
        r.drop_calls += ent["nfwd"]
        self._seed_val += ent["nfwd"]
        self.optimizer.graph_replayed()
        self.replays += 1
        return ent["static"][4].clone()

    def static_inputs(self, x_r, y_r, x_f, y_f, **kw):
        """The static batch buffers (x_r, y_r, x_f, y_f) of the graph captured for this configuration, or None before its capture.
        Filling them in place and passing them to the call skips the four staging copies of a replay."""
        ent = self.graphs.get(self._key(x_r, y_r, x_f, y_f, kw)) if self._usable() else None
        return None if ent is None else tuple(ent["static"][:4])

    def _capture(self, x_r, y_r, x_f, y_f, kw):
        r = self.net.runner()
        if y_r.dim() == 1 and y_f.dim() == 1 and y_r.dtype == y_f.dtype:      # one label buffer, two adjacent views (see _cat_labels)
            y_both = torch.cat((y_r, y_f), 0)
            y_r_s, y_f_s = y_both[:y_r.numel()], y_both[y_r.numel():]
        else:
            y_r_s, y_f_s = y_r.clone(), y_f.clone()
        static = [x_r.clone(), y_r_s, x_f.clone(), y_f_s, None]
        if self.seed_dev is None:
            self.seed_dev = torch.zeros(1, device=x_r.device, dtype=torch.int64)
        self._seed_val = None
        self.optimizer.graph_sync()
        calls0 = r.drop_calls
        torch.cuda.synchronize()
        self.optimizer.graph_mode, r.seed_dev = True, self.seed_dev
        # No cyclic garbage collection while the stream is capturing: an older model's GraphedStep that the collector happens to free inside the
        # capture destroys a hipGraph / events on a capturing thread and the runtime aborts the process (seen with 20 models built in sequence,
        # tests/test_hip_engines.py acc-stat cells). Collect first, keep the collector off until the capture has ended.
        import gc
        gc_was_on = gc.isenabled()
        gc.collect()
        gc.disable()
        try:
            if _dp_active():
                graph = _SegmentedCapture()
                graph.begin()
                try:
                    static[4] = gs_lora_step(self.model, self.optimizer, self.criterion, *static[:4], _comm=graph, **kw)
                except BaseException:
                    import sys
                    graph.end(sys.exc_info())
                    raise
                graph.end()
            else:
                graph = torch.cuda.CUDAGraph()
                # thread_local: allocator / stream activity of OTHER host threads (a DataLoader's pin_memory thread) does not invalidate the capture
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    static[4] = gs_lora_step(self.model, self.optimizer, self.criterion, *static[:4], **kw)
        finally:
            if gc_was_on:
                gc.enable()
            self.optimizer.graph_mode, r.seed_dev = False, None     # eager forwards keep passing the seed by value
            nfwd = r.drop_calls - calls0
            r.drop_calls = calls0                                     # nothing ran during capture
        self.captures += 1
        return dict(graph=graph, static=static, nfwd=nfwd)


def graphed_step_for(model, optimizer, criterion):
    """The engines call train_one_epoch once per epoch: the captured graph lives on the optimizer object across calls."""
    cache = optimizer.__dict__.setdefault("_gsl_graphed", {})
    k = (id(model), id(criterion))
    if k not in cache or cache[k].model is not model:
        cache.clear()
        cache[k] = GraphedStep(model, optimizer, criterion)
    return cache[k]


GRAPH_AUTO_MAX_IMAGES = 256     # "auto": capture when a step carries at most this many images (launch-bound regime)


def pick_stepper(model, optimizer, criterion, cfg, n_images):
    """cfg["HIP_GRAPH"]: True / False / "auto" (default). Returns a callable with gs_lora_step's data/keyword signature."""
    mode = (cfg or {}).get("HIP_GRAPH", "auto")
    if mode is True or (mode == "auto" and n_images <= GRAPH_AUTO_MAX_IMAGES):
        return graphed_step_for(model, optimizer, criterion)
    return lambda x_r, y_r, x_f, y_f, **kw: gs_lora_step(model, optimizer, criterion, x_r, y_r, x_f, y_f, **kw)


class MeterQueue:
    """Defers the D2H read of the per-step meter packs: the reference calls .item() >= 8 times per
    step (engine_cl.py:68-115); here the packs stay on the device until someone needs the numbers
    (display / evaluation / return) and are then fetched with ONE sync, and replayed into the
    AverageMeters in order — the meters end up with exactly the values the reference computes."""

    ORDER = ("losses_forget", "losses_remain", "losses_total", "losses_structure", "top1_forget", "top1_remain",
             "losses_prototype_forget", "losses_prototype_remain")
    # which batch size each meter is weighted with in the reference (engine_cl.py:68-120)
    WEIGHT = ("f", "r", "r", "r", "f", "r", "r", "r")

    def __init__(self):
        self.pending = []

    def push(self, pack, n_r, n_f):
        self.pending.append((pack, n_r, n_f))

    def flush(self, meters):
        if not self.pending:
            return
        rows = torch.stack([p for p, _, _ in self.pending]).tolist()     # the single host sync
        for k, row in enumerate(rows):
            # the reference stops with a KeyError on a label without a prototype (engine_cl.py:587-589) and never trains on a NaN loss; here
            # such a batch poisons the device-resident total, so the first read of the meters is where the run must stop
            if not all(v == v and abs(v) != float("inf") for v in row):
                self.pending = []
                raise FloatingPointError(f"gs-lora_amd: non-finite training meters {dict(zip(self.ORDER, row))} at deferred step {k} of "
                                         f"{len(rows)} (a label without a prototype, or diverged weights); the LoRA weights are no longer usable")
        for row, (_, n_r, n_f) in zip(rows, self.pending):
            for name, w, v in zip(self.ORDER, self.WEIGHT, row):
                m = meters.get(name)
                if m is not None:
                    m.update(v, n_f if w == "f" else n_r)
        self.pending = []
