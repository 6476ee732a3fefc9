"""Train step of the reference (train.py:41-107, :286-299, :535-548) for one process per GPU.

The reference wraps the model in single-process ``nn.DataParallel`` (train.py:517): per step it scatters
the batch, re-broadcasts every parameter, gathers logits and reduce-adds all gradients to GPU 0.  Here
every rank owns a replica and a contiguous batch shard; the only data-path collective is ONE
sum-all-reduce (RCCL over xGMI; ``gloo`` in the CPU tests) of a flat fp32 gradient buffer per step:

  * parameter ``.grad`` tensors are views into the flat buffer, so autograd accumulates in place and no
    gather/scatter copy is needed around the collective;
  * SUM, not mean: the reference loss is a sum over the global batch (KLDivLoss(size_average=False),
    train.py:541) and DataParallel adds replica gradients, so clip_grad_norm_(0.25) (train.py:82) sees
    the same global-batch gradient on every rank;
  * step order as train.py:63-86: forward, loss, scheduler.step(), zero_grad, backward, [all-reduce],
    clip, optimizer.step().  lr follows ExponentialLR(gamma = 0.5 ** (1/50000)) stepped every iteration.
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F


def kld_sum_loss(logits, target):
    """train.py:536-544: KLDivLoss(size_average=False)(F.log_softmax(logits), target).  GPU tensors go through the
    fused HIP kernel (ops.KldSumLoss); CPU tensors (the gloo tests of the host logic) through the torch ops."""
    if logits.is_cuda:
        from . import ops
        return ops.kld_sum_loss(logits, target)
    return F.kl_div(F.log_softmax(logits, dim=1), target, reduction="sum")


_HIP_NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 6: "wait_event",
                   7: "event_record"}


def graph_node_types(graph):
    """{node type: count} of a captured (keep_graph=True, not yet re-captured) torch.cuda.CUDAGraph, read through
    hipGraphGetNodes / hipGraphNodeGetType.  Returns {} when the runtime library cannot be queried."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        handle = ctypes.c_void_p(graph.raw_cuda_graph())
        count = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(handle, None, ctypes.byref(count)) != 0:
            return {}
        nodes = (ctypes.c_void_p * count.value)()
        if hip.hipGraphGetNodes(handle, nodes, ctypes.byref(count)) != 0:
            return {}
        out = {}
        for node in nodes:
            kind = ctypes.c_int(-1)
            if hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(kind)) != 0:
                return {}
            name = _HIP_NODE_TYPES.get(kind.value, "type%d" % kind.value)
            out[name] = out.get(name, 0) + 1
        return out
    except (OSError, AttributeError, RuntimeError):
        return {}


class FlatGradients:
    """One contiguous fp32 buffer holding every parameter's gradient (views installed as p.grad)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.buffer = torch.zeros(total, device=dev, dtype=torch.float32)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.buffer[off:off + n].view_as(p)
            off += n

    def zero(self):
        self.buffer.zero_()

    def all_reduce_sum(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.buffer, op=dist.ReduceOp.SUM, group=group)

    def clip_(self, max_norm):
        """torch.nn.utils.clip_grad_norm_ semantics (L2, eps 1e-6) on the flat buffer; returns the norm."""
        # accumulate in fp64: one long fp32 sum over ~12M elements is visibly (1e-3) off on some backends
        norm = torch.linalg.vector_norm(self.buffer, dtype=torch.float64).to(torch.float32)
        coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
        self.buffer.mul_(coef)
        return norm


def collect_stack_groups(model):
    """Every module's ``stack_groups()`` (lists of same-shaped parameters that are stacked per step)."""
    groups = []
    for m in model.modules():
        fn = getattr(m, "stack_groups", None)
        if callable(fn):
            groups.extend(fn())
    return groups


class FlatState:
    """GPU path: parameters, gradients and both Adam moments each live in ONE flat fp32 buffer (segments aligned
    to 16 bytes, the members of a stack group packed back to back).  ``p.data`` become views of the parameter buffer (state_dict / load_state_dict keep working);
    after backward the per-parameter gradients autograd produced are gathered into the gradient buffer by one
    multi-tensor copy (cheaper than zero + ~70 accumulate-adds), which is then the all-reduce payload and the
    input of the fused clip + Adam kernels (csrc/optimizer.hip)."""

    def __init__(self, params, stack_groups=()):
        # members of a stack group (same-shaped parameters that layers.my_linears stacks for one batched GEMM) are laid
        # out next to each other, in group order, so that the stack is a strided view of this buffer (ops.StackParams)
        params = [p for p in params if p.requires_grad]
        member = {}
        for group in stack_groups:
            group = [p for p in group if p.requires_grad]
            if len(group) > 1 and not any(id(p) in member for p in group):
                for p in group:
                    member[id(p)] = group
        ordered, placed = [], set()
        for p in params:
            for q in member.get(id(p), [p]):
                if id(q) not in placed:
                    placed.add(id(q))
                    ordered.append(q)
        self.params = ordered
        dev = self.params[0].device
        # segments start 16-byte aligned; the members of a stack group follow each other without padding (8-byte
        # aligned: even sizes), so the stack [G, ...] is a CONTIGUOUS view (ops.StackParams) -- e.g. the G biases of 510
        offs, off = [], 0
        for i, p in enumerate(self.params):
            group = member.get(id(p))
            tight = group is not None and p.numel() % 2 == 0
            inside = tight and i > 0 and member.get(id(self.params[i - 1])) is group
            if not inside:
                off = (off + 3) // 4 * 4
            offs.append(off)
            off += p.numel()
        off = (off + 3) // 4 * 4
        self.total = off
        self.offsets = offs
        self.p = torch.zeros(off, device=dev, dtype=torch.float32)
        self.g = torch.zeros(off, device=dev, dtype=torch.float32)
        self.m = torch.zeros(off, device=dev, dtype=torch.float32)
        self.v = torch.zeros(off, device=dev, dtype=torch.float32)
        self.g_views = []
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                view = self.p[o:o + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                self.g_views.append(self.g[o:o + p.numel()].view_as(p))
        from . import _lib
        self.norm_and_coef = torch.zeros(2, device=dev, dtype=torch.float32)
        self.workspace = torch.empty(_lib.lib().vqa_grad_norm_workspace_bytes() // 8, device=dev, dtype=torch.float64)

    def drop_grads(self):
        for p in self.params:
            p.grad = None

    def store_grads(self, params, grads):
        """Gradients computed by torch.autograd.grad for a subset of the parameters -> their slots of the flat buffer."""
        if not hasattr(self, "_view_of"):
            self._view_of = {id(p): v for p, v in zip(self.params, self.g_views)}
        src, dst = [], []
        for p, g in zip(params, grads):
            view = self._view_of[id(p)]
            if g is None:
                view.zero_()
            elif g.data_ptr() != view.data_ptr() or g.stride() != view.stride():
                src.append(g)
                dst.append(view)
            p.grad = view
        if src:
            torch._foreach_copy_(dst, src)

    def offset_of(self, p):
        return self.offsets[next(i for i, q in enumerate(self.params) if q is p)]

    def gather_grads(self):
        """Whatever backward left in p.grad -> the flat gradient buffer.  Gradients the kernels wrote there themselves
        (ops._grad_like: autograd adopted the slot as .grad) are in place already; only the rest is copied."""
        src, dst = [], []
        for p, view in zip(self.params, self.g_views):
            if p.grad is None:
                view.zero_()
            elif p.grad.data_ptr() != view.data_ptr() or p.grad.stride() != view.stride():
                src.append(p.grad)
                dst.append(view)
        self.last_gathered = len(src)       # (tests: 0 when every gradient was written in place)
        if src:
            torch._foreach_copy_(dst, src)
        for p, view in zip(self.params, self.g_views):
            p.grad = view

    def begin_backward(self):
        """Called right before a backward pass: parameter gradients may be written straight into the flat buffer."""
        from . import ops
        ops.set_grad_slots(self.p, self.g)
        self.drop_grads()

    def end_backward(self):
        from . import ops
        ops.set_grad_slots(self.p, None)


class DataParallelTrainer:
    """Replicated model + flat-gradient sum-all-reduce + the reference's clip/Adam/ExponentialLR."""

    # graph=True: steps launched kernel by kernel before the step is captured (the capture itself runs one more forward +
    # backward on a side stream first, torch's recipe).  Two suffice for the allocator, the kernels' LDS attributes and the
    # recorded GEMM solutions; every further step of a short warm-up is then already a replay at its steady clock.
    EAGER_STEPS_BEFORE_CAPTURE = 2

    def __init__(self, model, lr=1e-4, clip=0.25, gamma=0.5 ** (1 / 50000), broadcast=True, group=None,
                 fused_adam=None, graph=False, adopt_inputs=False, overlap=None, input_slots=1):
        self.model = model
        self.group = group
        self.clip = clip
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # the gradient all-reduce runs whenever there is more than one rank; VQA_FORCE_ALLREDUCE=1 also issues it in a
        # process group of ONE rank (an identity), which is how a single-GPU box exercises the real RCCL path -- init, the
        # collective between / beside the replayed hipGraphs, the async work handle (tests/test_gpu_nccl1.py)
        import os as _os
        self.reduce = self.world > 1 or (_os.environ.get("VQA_FORCE_ALLREDUCE") == "1" and dist.is_available()
                                         and dist.is_initialized())
        if broadcast and self.world > 1:
            # identical initial weights on every rank, once (replaces DataParallel's per-step broadcast)
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=group)
        self.gamma = gamma
        self.iteration = 0          # scheduler.last_epoch: restarts with every (re-)created scheduler, also on resume
        self.adam_steps = 0         # Adam's per-parameter `step`: travels with the optimizer state
        self.base_lr = lr
        self._lr = lr
        self.betas, self.eps = (0.9, 0.999), 1e-8           # torch.optim.Adam defaults (train.py:290)
        first = next(p for p in model.parameters() if p.requires_grad)
        self.hip = first.is_cuda                              # GPU: fused HIP tail; CPU (gloo tests): torch ops
        if self.hip:
            # the [B,*]-sized layers stay on library GEMMs: look their shapes up in the shipped solution table (tuned_gemms.py)
            from . import tuned_gemms
            tuned_gemms.enable()
        # graph=True: after a few eager steps the step is captured into two hipGraphs (forward+loss+backward+gather,
        # and clip+Adam) with the all-reduce launched eagerly between them; replays cost ~3 host launches instead of
        # ~600.  Falls back to eager if a forward draws a host-side dropout seed (fused K2/K5 masks).
        self.want_graph = bool(graph)
        # The replayed graphs read their batch from fixed buffers.  By default those are private copies and every batch is
        # copied into them (one device-to-device pass over the batch per step).  adopt_inputs=True makes the tensors of
        # the batch the step was captured on the graph's input buffers: no copy when the caller hands over the same
        # tensors every step (resident synthetic data), but later batches OVERWRITE those tensors -- never combine it
        # with a feeder that recycles its own buffers (feed.DevicePrefetcher).
        self.adopt_inputs = bool(adopt_inputs)
        # input_slots > 1 (with adopt_inputs): the feeder hands over batches in a RING of that many resident buffers (a DMA
        # target per slot).  The forward + backward graph is then captured once PER SLOT, reading that slot's tensors in
        # place -- the first step that sees a new slot runs kernel by kernel and captures it -- so no step pays a
        # device-to-device copy of its batch into a private input buffer; clip + Adam (which read no inputs) stay one graph.
        # The graphs share one memory pool: they never run concurrently.  A batch in none of the slots is copied into slot 0.
        self.input_slots = max(1, int(input_slots)) if self.adopt_inputs else 1
        self._slots = []
        self._graph = None
        self._eager_steps = 0
        # overlap (GPU path, models that offer late_parameters() / forward_with_cut(): CoR2, ODA): backward runs in two halves;
        # the gradients of the second reasoning step -- complete after the first half -- are all-reduced while the second
        # half runs.  OPT-IN (overlap=True or VQA_DP_OVERLAP=1): what it costs a single GPU is the split of the backward
        # graph (0.03 ms at B = 512), what it could hide is ~60 % of the all-reduce payload behind ~40 % of the backward --
        # but its gain over xGMI has never been measured (no multi-GPU node in this build's reach) and the register-tile
        # kernels run one or two workgroups per CU, so a concurrent RCCL kernel can also slow the backward it hides behind.
        # Until a scaling run says otherwise the default is the single all-reduce between backward and clip.
        # tests/test_gpu_dp2.py runs both on two ranks sharing one GPU.  "force" also splits at world size 1.
        requested = overlap
        if overlap is None:
            import os
            requested = overlap = os.environ.get("VQA_DP_OVERLAP", "0") == "1"
        self.overlap = False
        if self.hip:
            params = list(model.parameters())
            can_split = bool(overlap) and hasattr(model, "late_parameters") and hasattr(model, "forward_with_cut") and \
                (self.world > 1 or overlap == "force")
            if can_split:
                late_ids = {id(p) for p in model.late_parameters()}
                params = [p for p in params if id(p) in late_ids] + [p for p in params if id(p) not in late_ids]
            self.flat = FlatState(params, collect_stack_groups(model))
            if can_split:
                late = [p for p in self.flat.params if id(p) in late_ids]
                early = [p for p in self.flat.params if id(p) not in late_ids]
                if late and early:
                    split = min(self.flat.offset_of(p) for p in early)
                    if all(self.flat.offset_of(p) + p.numel() <= split for p in late):
                        self.overlap, self._late, self._early, self._split = True, late, early, split
            # say once which reduction schedule is in effect -- also when the split was asked for and could not be had
            # (the model offers no cut, or a stack group mixes late and early parameters so the late bucket is not a prefix)
            if self.overlap:
                self.overlap_note = "two-half backward, late bucket %d of %d floats reduced under the second half" % (
                    self._split, self.flat.total)
            elif requested:
                if not (hasattr(model, "late_parameters") and hasattr(model, "forward_with_cut")):
                    why = "model has no late_parameters()/forward_with_cut()"
                elif not can_split:
                    why = "world size 1"
                else:
                    why = "late parameters are not a prefix of the flat buffer"
                self.overlap_note = "single all-reduce after backward (overlap requested but unavailable: %s)" % why
            else:
                self.overlap_note = "single all-reduce after backward"
            if (self.world > 1 or requested) and (not dist.is_initialized() or dist.get_rank(group) == 0):
                import sys
                print("[vqa trainer] gradient reduction: %s (world %d)" % (self.overlap_note, self.world), file=sys.stderr)
            # the per-step device words the replayed graphs read -- Adam's two step scalars (fp32) and the dropout seed of the
            # fused kernels (int64; every replay reads the current value) -- are ONE 16-byte block, refreshed by ONE
            # host-to-device copy per step (a pinned-memory copy is a blit kernel of ~8 us that the step waits for)
            self._step_block = torch.zeros(16, device=first.device, dtype=torch.uint8)
            self.step_scalars = self._step_block[:8].view(torch.float32)
            self.seed_word = self._step_block[8:].view(torch.int64)
            # pinned staging: a ring, so a slot is not rewritten while an earlier (asynchronous) copy from it may still be
            # pending on the stream
            self._ring = 8
            self._ring_events = [None] * self._ring       # recorded behind a slot's copy; waited for before it is rewritten
            self._step_block_host = torch.zeros(self._ring, 16, dtype=torch.uint8).pin_memory()
            self._step_scalars_host = self._step_block_host[:, :8].view(torch.float32)      # [ring, 2]
            self._seed_host = self._step_block_host[:, 8:].view(torch.int64)                # [ring, 1]
            self.grads = None
            self.optimizer = None
        else:
            self.grads = FlatGradients(model.parameters())
            self.optimizer = torch.optim.Adam(self.grads.params, lr=lr)

    @staticmethod
    def _model_keys(sample):
        """The entries of a sample dict the models read (config/CoR2.py:203-205): feeder batches also carry the target
        'a' and 'q_id', which must not be cloned / re-copied into the graph's input buffers every step."""
        return {"v", "q_idxes"} if "q_idxes" in sample else {"v", "q"}

    def shard(self, tensor):
        """This rank's contiguous slice of a global-batch tensor (rank r gets [r*B/P, (r+1)*B/P))."""
        if self.world == 1:
            return tensor
        rank = dist.get_rank(self.group)
        per = tensor.size(0) // self.world
        return tensor[rank * per:(rank + 1) * per]

    def step(self, sample, target):
        """One training step on this rank's shard; returns (local loss tensor, global grad norm tensor).
        Replayed steps return the graph's OWN output tensors: the same two tensors every step, overwritten by the next
        replay -- read them (``.item()`` / ``.clone()``) before the next step if you keep a history; eager steps return
        fresh tensors."""
        if self.hip and self.want_graph:
            return self._graph_step(sample, target)
        return self.step_eager(sample, target)

    def step_eager(self, sample, target):
        """The same step launched kernel by kernel (no graph replay)."""
        if self.overlap:
            return self._step_split_eager(sample, target)
        logits = self.model(sample)
        loss = kld_sum_loss(logits, target)
        # scheduler.step() precedes optimizer.step() in the reference (train.py:75-86): step t uses lr0*gamma^t
        self.iteration += 1
        self.adam_steps += 1
        self._lr = lr = self.base_lr * self.gamma ** self.iteration
        if self.hip:
            from . import ops
            f = self.flat
            f.begin_backward()
            try:
                loss.backward()
            finally:
                f.end_backward()
            f.gather_grads()
            if self.reduce:
                dist.all_reduce(f.g, op=dist.ReduceOp.SUM, group=self.group)
            ops.grad_norm_clip_coef(f.g, self.clip if self.clip else 0.0, f.norm_and_coef, f.workspace)
            ops.adam_step(f.p, f.g, f.m, f.v, f.norm_and_coef, lr, self.betas[0], self.betas[1], self.eps, self.adam_steps)
            return loss.detach(), f.norm_and_coef[0]
        for gp in self.optimizer.param_groups:
            gp["lr"] = lr
        self.grads.zero()
        loss.backward()
        self.grads.all_reduce_sum(self.group)
        norm = self.grads.clip_(self.clip) if self.clip else None
        self.optimizer.step()
        return loss.detach(), norm

    # ---- hipGraph replay of the step ---------------------------------------------------------------------------
    def _front(self, sample, target):
        """forward + loss + backward + gradient gather (graph 1)."""
        from . import ops
        f = self.flat
        ops.set_device_seed(self.seed_word)
        ops.begin_step_salts()
        try:
            logits = self.model(sample)
            loss, d_logits = ops.kld_sum_loss_and_grad(logits, target)     # (loss and its gradient from one kernel)
        finally:
            ops.set_device_seed(None)
        f.begin_backward()
        try:
            torch.autograd.backward(logits, d_logits)
        finally:
            f.end_backward()
        f.gather_grads()
        return loss

    def _tail(self):
        """clip + Adam with the per-step scalars read from device memory (graph 2)."""
        from . import ops
        f = self.flat
        ops.grad_norm_clip_coef(f.g, self.clip if self.clip else 0.0, f.norm_and_coef, f.workspace)
        ops.adam_step_dyn(f.p, f.g, f.m, f.v, f.norm_and_coef, self.step_scalars, self.betas[0], self.betas[1], self.eps)

    def _set_step_scalars(self):
        self.iteration += 1
        self.adam_steps += 1
        self._lr = lr = self.base_lr * self.gamma ** self.iteration
        slot = self.adam_steps % self._ring
        # A replayed step costs the host a few launches per ~3 ms of GPU work, so a loop that never synchronises runs far
        # ahead: without this wait the slot of step t could be rewritten with step t + ring's values before the GPU has
        # executed the copy of step t (wrong lr / bias correction, the same dropout seed twice).
        pending = self._ring_events[slot]
        if pending is not None:
            pending.synchronize()
        self._step_scalars_host[slot, 0] = lr / (1.0 - self.betas[0] ** self.adam_steps)
        self._step_scalars_host[slot, 1] = 1.0 / (1.0 - self.betas[1] ** self.adam_steps) ** 0.5
        self._seed_host[slot, 0] = int(torch.randint(0, 2 ** 62, (1,), device="cpu").item())   # torch.manual_seed governs it
        self._step_block.copy_(self._step_block_host[slot], non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self._ring_events[slot] = done

    def _graph_step(self, sample, target):
        from . import ops
        f = self.flat
        if self.overlap:
            return self._graph_step_split(sample, target)
        if self._graph is None:
            seeds_before = ops.host_seed_draws
            self._set_step_scalars()
            loss = self._front(sample, target)
            if self.reduce:
                dist.all_reduce(f.g, op=dist.ReduceOp.SUM, group=self.group)
            self._tail()
            self._eager_steps += 1
            if ops.host_seed_draws != seeds_before:
                self.want_graph = False       # host-seeded dropout mask in the forward: replay would freeze it
            elif self._eager_steps >= self.EAGER_STEPS_BEFORE_CAPTURE:   # warmed up (allocator, LDS attributes, GEMM table): capture
                try:
                    self._capture(sample, target)
                except Exception as e:        # noqa: BLE001 -- any capture failure: keep training, kernel by kernel
                    import sys
                    print("[vqa trainer] hipGraph capture failed (%s: %s); continuing with eager launches"
                          % (type(e).__name__, str(e).splitlines()[0] if str(e) else ""), file=sys.stderr)
                    self._graph = None
                    self.want_graph = False
                    torch.cuda.synchronize()
            return loss, f.norm_and_coef[0]
        g = self._graph
        if self.model.training != g["training"] or target.shape != g["target"].shape or any(k not in sample or sample[k].shape != t.shape or sample[k].dtype != t.dtype
                                                    for k, t in g["sample"].items()):
            # a batch of another shape (the last one of an epoch), or the model switched between train() and eval() since
            # the capture (dropout is baked into the graph): this step is launched kernel by kernel; the captured graphs
            # stay valid for the regular steps that follow
            return self.step_eager(sample, target)
        slot = self._slot_of(sample, target)
        if slot is None and len(self._slots) < self.input_slots:
            # a slot of the feeder's ring the step has not been captured on yet: this step runs kernel by kernel (the same
            # launches, device-side seed and step scalars as a replay), then the front graph is captured on the slot's tensors
            self._set_step_scalars()
            loss = self._front(sample, target)
            if self.reduce:
                dist.all_reduce(f.g, op=dist.ReduceOp.SUM, group=self.group)
            self._tail()
            try:
                self._slots.append(self._capture_front(sample, target, g["pool"], ("front",)))
            except Exception as e:        # noqa: BLE001 -- keep training: further batches of this slot are copied into slot 0
                import sys
                print("[vqa trainer] capture of input slot %d failed (%s: %s); its batches will be copied into slot 0"
                      % (len(self._slots), type(e).__name__, str(e).splitlines()[0] if str(e) else ""), file=sys.stderr)
                self.input_slots = len(self._slots)
                torch.cuda.synchronize()
            return loss, f.norm_and_coef[0]
        if slot is None:
            slot = g
            for k, t in g["sample"].items():
                if sample[k].data_ptr() != t.data_ptr():
                    t.copy_(sample[k], non_blocking=True)
            if target.data_ptr() != g["target"].data_ptr():
                g["target"].copy_(target, non_blocking=True)
        self._set_step_scalars()
        slot["front"].replay()
        if self.reduce:
            dist.all_reduce(f.g, op=dist.ReduceOp.SUM, group=self.group)
        g["tail"].replay()
        return slot["loss"], f.norm_and_coef[0]

    def _slot_of(self, sample, target):
        """The captured input slot whose tensors ARE this batch's (same addresses), or None."""
        for slot in self._slots:
            if target.data_ptr() == slot["target"].data_ptr() and all(
                    sample[k].data_ptr() == t.data_ptr() for k, t in slot["sample"].items()):
                return slot
        return None

    def _capture_front(self, sample, target, pool, names):
        """Capture forward + loss + backward + gather (names = ("front",)) or its two halves (("front_a", "front_b")) reading
        `sample` / `target` -- the caller's tensors with adopt_inputs, private copies otherwise.  -> the slot record."""
        static_sample = {k: (v if self.adopt_inputs else v.clone()) for k, v in sample.items()
                         if isinstance(v, torch.Tensor) and k in self._model_keys(sample)}
        if not self.adopt_inputs:
            target = target.clone()
        split = len(names) == 2
        # torch's capture recipe: one forward+backward on a side stream first, so the parameters' AccumulateGrad
        # nodes belong to a capturable stream (nodes created on the default stream would run there and abort the
        # capture).  It only refills the gradient buffer; no parameter is updated.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            if split:
                self._front_a(static_sample, target)
                self._front_b()
            else:
                self._front(static_sample, target)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        mode = "global"
        if self.reduce:
            # the collective library's watchdog thread polls events of outstanding work; let every rank drain its
            # work list first and capture in thread-local mode so another thread's query cannot invalidate the capture
            import time
            dist.barrier(group=self.group)
            torch.cuda.synchronize()
            time.sleep(1.0)
            mode = "thread_local"
        graphs = {k: torch.cuda.CUDAGraph(keep_graph=True) for k in names}
        if split:
            with torch.cuda.graph(graphs["front_a"], pool=pool, capture_error_mode=mode):
                loss = self._front_a(static_sample, target)
            with torch.cuda.graph(graphs["front_b"], pool=pool, capture_error_mode=mode):
                self._front_b()
        else:
            with torch.cuda.graph(graphs["front"], pool=pool, capture_error_mode=mode):
                loss = self._front(static_sample, target)
        self._audit_and_instantiate(graphs)
        return dict(graphs, loss=loss, sample=static_sample, target=target, mode=mode)

    def _audit_and_instantiate(self, graphs):
        # Audit before instantiating: a memset node (hipMemsetAsync under capture) is replayed correctly once and then
        # writes garbage on ROCm 7.2 (tools/graph_memset_check.py).  The library and the model issue none -- zero fills
        # are kernels, the loss and the bias gradients avoid torch's semaphore-based reductions -- and a step that
        # contains one anyway (a user-supplied seq2vec, a new torch op) must not be replayed.
        nodes = {k: graph_node_types(v) for k, v in graphs.items()}
        self.graph_nodes = dict(getattr(self, "graph_nodes", None) or {}, **nodes)
        if any("kernel" not in c for c in nodes.values()):
            # the audit could not read the graphs (runtime library not queryable): unknown is not "clean" -- stay eager
            raise RuntimeError("cannot audit the captured graphs for memset nodes (hipGraphGetNodes unavailable)")
        memsets = sum(c.get("memset", 0) for c in nodes.values())
        if memsets:
            raise RuntimeError("captured step holds %d memset node(s), which do not replay reliably" % memsets)
        for v in graphs.values():
            v.instantiate()
        torch.cuda.synchronize()

    def _capture(self, sample, target):
        pool = torch.cuda.graph_pool_handle()
        self.graph_nodes = {}
        slot = self._capture_front(sample, target, pool, ("front",))
        tail = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(tail, pool=pool, capture_error_mode=slot["mode"]):
            self._tail()
        self._audit_and_instantiate({"tail": tail})
        self._slots = [slot]
        self._graph = dict(slot, tail=tail, pool=pool, training=self.model.training)

    # ---- backward in two halves, the first all-reduce under the second (overlap=True) ---------------------------------
    def _front_a(self, sample, target, device_seed=True):
        """forward + loss + backward from the loss down to the cut + gather of the late parameters' gradients."""
        from . import ops
        if device_seed:
            ops.set_device_seed(self.seed_word)
            ops.begin_step_salts()
        try:
            logits, outs, ins = self.model.forward_with_cut(sample)
            loss, d_logits = ops.kld_sum_loss_and_grad(logits, target)
        finally:
            if device_seed:
                ops.set_device_seed(None)
        live = [(o, i) for o, i in zip(outs, ins) if i.requires_grad]
        ops.set_grad_slots(self.flat.p, self.flat.g)
        try:
            grads = torch.autograd.grad(logits, self._late + [i for _, i in live], grad_outputs=d_logits, allow_unused=True)
        finally:
            ops.set_grad_slots(self.flat.p, None)
        self.flat.store_grads(self._late, grads[:len(self._late)])
        pairs = [(o, g) for (o, _), g in zip(live, grads[len(self._late):]) if g is not None]
        self._cut = ([o for o, _ in pairs], [g for _, g in pairs])
        return loss

    def _front_b(self):
        """backward from the cut to the inputs + gather of the early parameters' gradients."""
        tensors, grads_in = self._cut
        self._cut = None
        from . import ops
        ops.set_grad_slots(self.flat.p, self.flat.g)
        try:
            grads = torch.autograd.grad(tensors, self._early, grad_outputs=grads_in, allow_unused=True)
        finally:
            ops.set_grad_slots(self.flat.p, None)
        self.flat.store_grads(self._early, grads)

    def _reduce_late_async(self):
        if self.reduce:
            return dist.all_reduce(self.flat.g[:self._split], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return None

    def _reduce_early(self, pending):
        if self.reduce:
            dist.all_reduce(self.flat.g[self._split:], op=dist.ReduceOp.SUM, group=self.group)
        if pending is not None:
            pending.wait()

    def _step_split_eager(self, sample, target):
        from . import ops
        f = self.flat
        loss = self._front_a(sample, target, device_seed=False)
        pending = self._reduce_late_async()
        self._front_b()
        self._reduce_early(pending)
        self.iteration += 1
        self.adam_steps += 1
        self._lr = lr = self.base_lr * self.gamma ** self.iteration
        ops.grad_norm_clip_coef(f.g, self.clip if self.clip else 0.0, f.norm_and_coef, f.workspace)
        ops.adam_step(f.p, f.g, f.m, f.v, f.norm_and_coef, lr, self.betas[0], self.betas[1], self.eps, self.adam_steps)
        return loss, f.norm_and_coef[0]

    def _graph_step_split(self, sample, target):
        from . import ops
        f = self.flat
        if self._graph is None:
            seeds_before = ops.host_seed_draws
            self._set_step_scalars()
            loss = self._front_a(sample, target)
            pending = self._reduce_late_async()
            self._front_b()
            self._reduce_early(pending)
            self._tail()
            self._eager_steps += 1
            if ops.host_seed_draws != seeds_before:
                self.want_graph = False
            elif self._eager_steps >= self.EAGER_STEPS_BEFORE_CAPTURE:
                try:
                    self._capture_split(sample, target)
                except Exception as e:        # noqa: BLE001 -- any capture failure: keep training, kernel by kernel
                    import sys
                    print("[vqa trainer] hipGraph capture failed (%s: %s); continuing with eager launches"
                          % (type(e).__name__, str(e).splitlines()[0] if str(e) else ""), file=sys.stderr)
                    self._graph = None
                    self.want_graph = False
                    torch.cuda.synchronize()
            return loss, f.norm_and_coef[0]
        g = self._graph
        if self.model.training != g["training"] or target.shape != g["target"].shape or any(
                k not in sample or sample[k].shape != t.shape or sample[k].dtype != t.dtype for k, t in g["sample"].items()):
            return self.step_eager(sample, target)
        slot = self._slot_of(sample, target)
        if slot is None and len(self._slots) < self.input_slots:      # a new slot of the feeder's ring: see _graph_step
            self._set_step_scalars()
            loss = self._front_a(sample, target)
            pending = self._reduce_late_async()
            self._front_b()
            self._reduce_early(pending)
            self._tail()
            try:
                self._slots.append(self._capture_front(sample, target, g["pool"], ("front_a", "front_b")))
            except Exception as e:        # noqa: BLE001
                import sys
                print("[vqa trainer] capture of input slot %d failed (%s: %s); its batches will be copied into slot 0"
                      % (len(self._slots), type(e).__name__, str(e).splitlines()[0] if str(e) else ""), file=sys.stderr)
                self.input_slots = len(self._slots)
                torch.cuda.synchronize()
            return loss, f.norm_and_coef[0]
        if slot is None:
            slot = g
            for k, t in g["sample"].items():
                if sample[k].data_ptr() != t.data_ptr():
                    t.copy_(sample[k], non_blocking=True)
            if target.data_ptr() != g["target"].data_ptr():
                g["target"].copy_(target, non_blocking=True)
        self._set_step_scalars()
        slot["front_a"].replay()
        pending = self._reduce_late_async()
        slot["front_b"].replay()
        self._reduce_early(pending)
        g["tail"].replay()
        return slot["loss"], f.norm_and_coef[0]

    def _capture_split(self, sample, target):
        pool = torch.cuda.graph_pool_handle()
        self.graph_nodes = {}
        slot = self._capture_front(sample, target, pool, ("front_a", "front_b"))
        tail = torch.cuda.CUDAGraph(keep_graph=True)
        with torch.cuda.graph(tail, pool=pool, capture_error_mode=slot["mode"]):
            self._tail()
        self._audit_and_instantiate({"tail": tail})
        self._slots = [slot]
        self._graph = dict(slot, tail=tail, pool=pool, training=self.model.training)

    @property
    def lr(self):
        return self._lr

    # ---- optimizer (re-)creation and the reference's checkpoint files -----------------------------------------------
    def _optimizer_params(self):
        # train.py:288-292: filter(lambda p: p.requires_grad, model.parameters()) -- the order of the saved state
        return [p for p in self.model.parameters() if p.requires_grad]

    def reset_optimizer(self, lr=None):
        """``optimizer, scheduler = learning_scheduler(cf)`` (train.py:286-299): fresh Adam moments and step count, the
        learning rate back at ``cf.lr`` and a scheduler that starts over."""
        if lr is not None:
            self.base_lr = lr
        self.iteration = 0
        self.adam_steps = 0
        self._lr = self.base_lr
        if self.hip:
            # fills are kernels and the moment buffers keep their addresses: a captured step stays valid
            self.flat.m.zero_()
            self.flat.v.zero_()
        else:
            self.optimizer = torch.optim.Adam(self.grads.params, lr=self.base_lr)

    def begin_epoch(self, epoch, cf):
        """The optimizer re-creation rule at the top of the reference's epoch loop (train.py:718-721): at
        ``cf.restart_epoch``, and before every epoch below ``cf.keeping_epoch``.  Returns True if it was re-created."""
        lr = getattr(cf, "lr", None)
        if hasattr(cf, "restart_epoch") and epoch == cf.restart_epoch:
            self.reset_optimizer(lr)
            return True
        if getattr(cf, "keeping_epoch", None) is not None and epoch < cf.keeping_epoch:
            self.reset_optimizer(lr)
            return True
        return False

    def optimizer_state_dict(self):
        """``optimizer.state_dict()`` of the reference's torch.optim.Adam (train.py:290, saved at :729): per-parameter
        ``step`` / ``exp_avg`` / ``exp_avg_sq`` keyed by the index of the parameter in the filtered
        ``model.parameters()`` order, one param group.  Loadable by ``torch.optim.Adam.load_state_dict``."""
        if not self.hip:
            return self.optimizer.state_dict()
        f = self.flat
        where = {id(p): (o, p) for p, o in zip(f.params, f.offsets)}
        params = self._optimizer_params()
        state = {}
        if self.adam_steps > 0:
            for i, p in enumerate(params):
                o, _ = where[id(p)]
                n = p.numel()
                state[i] = {"step": self.adam_steps,
                            "exp_avg": f.m[o:o + n].view_as(p).clone(),
                            "exp_avg_sq": f.v[o:o + n].view_as(p).clone()}
        group = {"lr": self._lr, "betas": self.betas, "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "initial_lr": self.base_lr, "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd):
        """``optimizer.load_state_dict`` (train.py:283).  As in the reference, the scheduler is not part of the
        checkpoint: it was created just before the load (train.py:519,:575) and starts over from ``cf.lr``, while
        Adam's moments and step count continue."""
        if not self.hip:
            self.optimizer.load_state_dict(sd)
            return
        f = self.flat
        where = {id(p): o for p, o in zip(f.params, f.offsets)}
        params = self._optimizer_params()
        group = sd["param_groups"][0]
        if len(sd["param_groups"]) != 1 or len(group["params"]) != len(params):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        self.betas = tuple(group.get("betas", self.betas))
        self.eps = group.get("eps", self.eps)
        steps = set()
        with torch.no_grad():
            f.m.zero_()
            f.v.zero_()
            for key, p in zip(group["params"], params):
                st = sd["state"].get(key)
                if st is None:
                    continue
                if tuple(st["exp_avg"].shape) != tuple(p.shape):
                    raise ValueError("optimizer state of parameter %d has shape %s, expected %s"
                                     % (key, tuple(st["exp_avg"].shape), tuple(p.shape)))
                o, n = where[id(p)], p.numel()
                f.m[o:o + n].view_as(p).copy_(st["exp_avg"])
                f.v[o:o + n].view_as(p).copy_(st["exp_avg_sq"])
                steps.add(int(st["step"]))
        if len(steps) > 1:
            raise ValueError("per-parameter Adam step counts differ (%s): one fused step count is kept" % sorted(steps))
        self.adam_steps = steps.pop() if steps else 0
        self.iteration = 0
        self._lr = self.base_lr

    def save_checkpoint(self, info, log_dir):
        """train.py:250-267: ``log_dir/epoch_<n>/ckpt_{info,model,optim}.pth.tar`` holding ``info``, the model's
        ``state_dict()`` (the reference's ``model.module.state_dict()`` names) and the optimizer state.
        Replicas are identical, so only rank 0 writes."""
        import os
        if self.world > 1 and dist.get_rank(self.group) != 0:
            return None
        path = os.path.join(log_dir, "epoch_%d" % info["epoch"])
        os.makedirs(path, exist_ok=True)
        logger = info.get("exp_logger")
        if logger is not None and hasattr(logger, "to_json"):
            logger.to_json(os.path.join(path, "logger.json"))
        torch.save(info, os.path.join(path, "ckpt_info.pth.tar"))
        torch.save({k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                   os.path.join(path, "ckpt_model.pth.tar"))
        optim = self.optimizer_state_dict()
        for st in optim["state"].values():
            for k, v in st.items():
                if torch.is_tensor(v):
                    st[k] = v.cpu()
        torch.save(optim, os.path.join(path, "ckpt_optim.pth.tar"))
        return path

    def load_checkpoint(self, path_ckpt):
        """train.py:270-284; returns ``info['exp_logger']`` like the reference (None if the file holds none)."""
        import os
        info = torch.load(os.path.join(path_ckpt, "ckpt_info.pth.tar"), weights_only=False)
        model_state = torch.load(os.path.join(path_ckpt, "ckpt_model.pth.tar"), map_location="cpu")
        self.model.load_state_dict(model_state)      # copies into the flat parameter buffer's views
        self.load_optimizer_state_dict(torch.load(os.path.join(path_ckpt, "ckpt_optim.pth.tar"), map_location="cpu"))
        return info.get("exp_logger") if isinstance(info, dict) else None
