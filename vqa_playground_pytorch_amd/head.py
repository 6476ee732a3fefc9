"""The [B, .]-sized layers of the heads as grouped launches (K6; csrc/grouped_gemm.hip).

Everything of CoR2 / ODA that is not region-sized -- the question projections, the sigmoid gates, Mutan's question-side
rank factors, the per-glimpse projections, fusion_final's two sides with its rank product, the classifier (MyLinear /
putils.Linear / MutanFusion: config/CoR2.py:94-122,133-134,170,180-189, config/ODA.py:183-198, putils/__init__.py:16-33,
205-241) -- runs here in PHASES.  A phase is every GEMM whose inputs are ready at the same point of the step: one
``vqa_grouped_gemm`` launch (all of them, forward / data-gradient / weight-gradient forms mixed, split over their
contraction so the union fills the chip) and one ``vqa_grouped_epilogue`` launch (fixed-order reduction of the partial
products + bias, activation, the consumer's input dropout, rank products, activation gradients, layouts).  Per layer the
reference issues dropout -> linear -> activation and autograd four more kernels backward; per step that was 28 library
GEMMs surrounded by ~40 launch-floor kernels.

Each phase is one ``torch.autograd.Function``.  Contract between consecutive phases (they are each other's only
consumers, which the models guarantee by construction): a phase returns the gradient of its input ALREADY multiplied by
the producing layer's activation / dropout gradient -- it holds the producer's stored output, whose sign carries both --
so the producer's backward starts at its GEMMs.  Activations that feed a dropout layer are stored dropped out (0 or
x/(1-p)), like K3 stores the pooled glimpses: "stored value > 0" is then "kept and active".
"""
import ctypes
import heapq
import math
import os

import torch

from . import _lib, ops

_c = ctypes
MAX_GROUP = 24          # epilogue jobs per launch (VQA_GROUPED_MAX)
MAX_GEMMS = 16          # GEMM problems per launch (VQA_GROUPED_GEMM_MAX)


class GemmProblem(_c.Structure):          # VqaGemmProblem (include/vqa_mi355x.h)
    _fields_ = [("A", _c.c_void_p), ("B", _c.c_void_p), ("slab", _c.c_void_p), ("colsum", _c.c_void_p),
                ("slab_stride", _c.c_longlong), ("lda", _c.c_int), ("ldb", _c.c_int), ("M", _c.c_int), ("N", _c.c_int),
                ("K", _c.c_int), ("form", _c.c_int), ("ksplit", _c.c_int), ("slab_base", _c.c_int), ("Ka", _c.c_int),
                ("Kb", _c.c_int), ("Ma", _c.c_int), ("Nb", _c.c_int),
                # direct output (the tile is finished in the GEMM kernel; no slab, no epilogue job)
                ("out", _c.c_void_p), ("colsum_out", _c.c_void_p), ("bias", _c.c_void_p), ("gate_y", _c.c_void_p),
                ("seed_ptr", _c.c_void_p), ("seed", _c.c_uint64), ("ldo", _c.c_int), ("ld_gate", _c.c_int), ("act", _c.c_int),
                ("gate", _c.c_int), ("drop_base", _c.c_uint32), ("drop_ld", _c.c_uint32), ("p_drop", _c.c_float),
                ("gate_scale", _c.c_float)]


class EpilogueJob(_c.Structure):          # VqaEpilogueJob
    _fields_ = [("slab", _c.c_void_p), ("bias", _c.c_void_p), ("aux", _c.c_void_p), ("aux2", _c.c_void_p),
                ("out", _c.c_void_p), ("out2", _c.c_void_p), ("seed_ptr", _c.c_void_p), ("seed", _c.c_uint64),
                ("slab_stride", _c.c_longlong), ("S", _c.c_int), ("M", _c.c_int), ("N", _c.c_int), ("kind", _c.c_int),
                ("ldo", _c.c_int), ("ld_aux", _c.c_int), ("act", _c.c_int), ("gate", _c.c_int), ("R", _c.c_int),
                ("seg", _c.c_int), ("seg_ld", _c.c_int), ("drop_base", _c.c_uint32), ("drop_ld", _c.c_uint32),
                ("p_drop", _c.c_float), ("gate_scale", _c.c_float)]


NT, NN, TN, NN_A4, TN_A4 = 0, 1, 2, 3, 4
EPI_SUM, EPI_LINEAR, EPI_RANK_PRODUCT, EPI_GRAD, EPI_RANK_PRODUCT_BWD = 0, 1, 2, 3, 4
ACT = {None: 0, "": 0, "relu": 1, "sigmoid": 2}

# VQA_HEAD: which of a model's [B,.]-sized layers run as grouped phases.
#   auto (default)  what measured fastest per model on one MI355X at B = 512 (profiles/): CoR2 runs its
#                   question / gate / rank-factor / fusion / classifier phases grouped (4-10 products per launch) and its
#                   glimpse projections as the one batched library GEMM (155-wide blocks: odd widths force 4-byte operand
#                   loads and three 64-column tiles for 155 columns); ODA, whose phases hold one or two products each (no
#                   grouping to gain, a split-K epilogue to pay), keeps library GEMMs + the per-layer epilogue kernels
#   grouped         every phase of both models grouped (glimpses included)
#   legacy          library GEMMs + per-layer epilogue kernels everywhere
MODE = os.environ.get("VQA_HEAD", "auto")
if MODE not in ("auto", "grouped", "legacy"):
    raise ValueError("VQA_HEAD must be auto, grouped or legacy (got %r)" % MODE)
_mask_spy = None      # tests: callable(site, rows, cols, p, seed) told about every dropout mask an epilogue applies


def _ptr(t, offset=0):
    return t.data_ptr() + 4 * int(offset)


class _Target:
    """One accumulation target of a phase: sum of the partial products of every problem added to it."""

    def __init__(self, M, N):
        self.M, self.N = int(M), int(N)
        self.problems = []
        self.slab = None
        self.colsum = None
        self.S = 0
        self.direct = None


class Phase:
    """Collects the GEMM problems and epilogue jobs of one phase, sizes the contraction splits so that the phase's tiles
    fill the chip, allocates the slabs and launches the two kernels."""

    # How a phase is cut into work items (64x64 tile x contraction part).  No item is longer than MAX_PART contraction
    # steps (a phase mixes K = 155 .. 2400: the longest items would otherwise set the launch time), and when the phase has
    # fewer items than the target (see _size) the parts are shortened until it has.
    # A product that ends up in ONE part and is the only contribution to its result is finished inside the GEMM kernel
    # (direct output: bias / activation / gate / dropout on the accumulators) and needs neither slab nor epilogue job.
    # VQA_GROUPED_ITEMS / VQA_GROUPED_PART / VQA_GROUPED_BM (tile rows 64 | 128, read by the library as well): measurement knobs.
    MIN_ITEMS = int(os.environ.get("VQA_GROUPED_ITEMS", "0"))     # 0: by the phase's rows (below)
    MAX_PART = int(os.environ.get("VQA_GROUPED_PART", "640"))
    TILE_M = 128 if os.environ.get("VQA_GROUPED_BM") == "128" else 64
    DIRECT = os.environ.get("VQA_GROUPED_DIRECT", "1") == "1"
    STEP_K = 16        # the kernel's K step: a contraction part is a whole number of steps
    # Which matrix engine runs a phase's GEMM launch.  "mfma" = csrc/grouped_gemm.hip on the fp32 MFMA; "split" =
    # csrc/grouped_gemm_split.hip (128 x 128|160 tiles, 32-deep steps); "mixed" = per phase, whichever measured faster at
    # B = 512 (tools/gg_bench.py: SPLIT_PHASES run on the split engine); "auto" (default) = mixed when the step's other fp32
    # products run on the split engine (ops.f32_products()), else mfma.
    ENGINE = os.environ.get("VQA_GROUPED_ENGINE", "auto")
    SPLIT_PHASES = tuple(x for x in os.environ.get(
        "VQA_GROUPED_SPLIT_PHASES", "vector_fusion_bwd,classifier_bwd").split(",") if x)   # (a measurement knob)
    # (round 6 re-measured the table after the fp32 grouped kernel lost its scratch accesses -- the weight-gradient phases are the
    #  ones that gained: none 1.965-1.967 ms, q_proj_fwd 1.965-1.968, gates_h2_bwd 1.973-1.978, vector_fusion_bwd 1.962-1.963,
    #  classifier_bwd 1.963-1.966, round 5's four 1.970)
    SPLIT_MIN_ROWS = 384     # mixed: only at the training batch -- at 128 rows (one rank's share of BASELINE configs[4]) a phase is
                             # a chip's worth of 2-3-step items and the split kernel's fixed cost per item loses: 0.997 vs 0.955 ms
    SPLIT_UNITS = int(os.environ.get("VQA_GROUPED_SPLIT_UNITS", "256"))   # workgroups the chip runs at once (one per CU)
    SPLIT_OVERHEAD = float(os.environ.get("VQA_GROUPED_SPLIT_OVERHEAD", "3"))   # an item's fixed cost, in contraction steps
    _split_plans = {}

    def __init__(self, device, name):
        self.device, self.name = device, name
        self.targets, self.pre_jobs, self.jobs = [], [], []

    def target(self, M, N):
        t = _Target(M, N)
        self.targets.append(t)
        return t

    def gemm(self, target, form, A, lda, B, ldb, K, a_off=0, b_off=0, colsum=False, Ka=0, Kb=0, Ma=0, Nb=0):
        """target (+)= A (.) B over K.  A, B: tensors; a_off / b_off: element offsets of the operand's first element."""
        target.problems.append(dict(form=form, A=A, a_off=a_off, lda=int(lda), B=B, b_off=b_off, ldb=int(ldb), K=int(K),
                                    colsum=bool(colsum), Ka=Ka, Kb=Kb, Ma=Ma, Nb=Nb))

    def job(self, kind, source, out, ldo, out_off=0, pre=False, **kw):
        """source: a _Target (its reduced slabs) or a tensor [M,N] (S = 1: an elementwise job on existing data)."""
        (self.pre_jobs if pre else self.jobs).append(dict(kind=kind, source=source, out=out, ldo=int(ldo), out_off=out_off, **kw))

    @classmethod
    def engine(cls, name=None, rows=None):
        """The engine of the phase called `name` whose products have `rows` batch rows (None: a phase outside the table)."""
        mode = cls.ENGINE
        if mode not in ("auto", "mixed", "split", "mfma"):
            raise ValueError("VQA_GROUPED_ENGINE must be auto, mixed, split or mfma (got %r)" % mode)
        if mode == "auto":
            mode = "mixed" if ops.f32_products() == "split" else "mfma"
        if mode == "mixed":
            return "split" if (name in cls.SPLIT_PHASES and rows is not None and rows >= cls.SPLIT_MIN_ROWS) else "mfma"
        return mode

    def _engine(self):
        if getattr(self, "force_engine", None):     # (ops._grouped_products: the caller knows the engine)
            return self.force_engine
        rows = [p["K"] if p["form"] in (TN, TN_A4) else t.M for t in self.targets for p in t.problems]
        return self.engine(self.name, min(rows) if rows else None)

    @classmethod
    def _plan_split(cls, shapes):
        """Contraction steps per part for a launch on the split engine.  shapes: (tiles, steps) per problem -- 128 x BN tiles,
        32-deep steps.  One workgroup per CU, items dispatched in order: the part length p (every problem is cut into
        ceil(steps / p) equal parts) is the one whose in-order schedule over SPLIT_UNITS units finishes first, an item
        costing its steps + SPLIT_OVERHEAD (prologue, the store of a 128 x BN slab)."""
        key = tuple(shapes)
        plan = cls._split_plans.get(key)
        if plan is not None:
            return plan
        best = None
        for part in range(2, max(st for _, st in shapes) + 1):
            lens = []
            for tiles, st in shapes:
                n = math.ceil(st / part)
                lens += [math.ceil(st / n) + cls.SPLIT_OVERHEAD] * (tiles * n)
            if len(lens) > (1 << 14):
                continue
            units = [0.0] * cls.SPLIT_UNITS
            # (in-order dispatch: the next item goes to the unit that frees first)
            heapq.heapify(units)
            for ln in lens:
                heapq.heappush(units, heapq.heappop(units) + ln)
            span = max(units)
            if best is None or span < best[0] - 1e-9:
                best = (span, part)
        plan = best[1] if best is not None else 2
        cls._split_plans[key] = plan
        return plan

    @staticmethod
    def split_tile_cols(N):
        """Tile width of csrc/grouped_gemm_split.hip for an N-wide output (vqa_grouped_gemm_split_tile_cols): 128 or 160,
        whichever pads N less; ties go to the wider tile."""
        w128, w160 = math.ceil(N / 128) * 128, math.ceil(N / 160) * 160
        return 160 if w160 <= w128 else 128

    def _size_split(self):
        probs = [(t, p) for t in self.targets for p in t.problems]
        shapes = [(math.ceil(t.M / 128) * math.ceil(t.N / self.split_tile_cols(t.N)), math.ceil(p["K"] / 32)) for t, p in probs]
        part = self._plan_split(shapes)
        for (t, p), (_, st) in zip(probs, shapes):
            n = math.ceil(st / part)
            p["ksplit"] = math.ceil(st / n) * 32
            p["splits"] = math.ceil(p["K"] / p["ksplit"])
        return probs

    def _size(self):
        if self._engine() == "split":
            return self._size_split()
        probs = [(t, p) for t in self.targets for p in t.problems]
        tiles = lambda t: math.ceil(t.M / self.TILE_M) * math.ceil(t.N / 64)  # noqa: E731
        part = self.MAX_PART
        # work items a launch aims for: 512 (two per CU) at the training batch; a phase whose products have <= 128 rows (one
        # rank's share of BASELINE configs[4]: 128 samples) holds one or two row tiles per product and needs deeper splits to
        # fill the chip -- measured at B = 128: 768 / 1024 / 1536 / 2048 items -> 0.158 / 0.151 / 0.151 / 0.151 ms of grouped
        # GEMM per step (512: 0.172), at B = 512: 768 / 1024 are 0.5 / 1 % slower than 512
        min_items = self.MIN_ITEMS or (1024 if min(t.M for t in self.targets) <= 128 else 512)
        while True:
            items = sum(tiles(t) * math.ceil(p["K"] / part) for t, p in probs)
            if items >= min_items or part <= 128:
                break
            part = max(128, part - 64)
        for t, p in probs:
            splits = math.ceil(p["K"] / part)
            p["ksplit"] = math.ceil(p["K"] / splits / self.STEP_K) * self.STEP_K
            p["splits"] = math.ceil(p["K"] / p["ksplit"])
        return probs

    def run(self):
        L_ = _lib.lib()
        if self.pre_jobs:
            self._epilogue(L_, self.pre_jobs, "pre")
        jobs = list(self.jobs)
        if self.targets:
            engine = self._engine()
            probs = self._size()
            flops = 0
            for t in self.targets:
                t.S = sum(p["splits"] for p in t.problems)
                mine = [j for j in jobs if j["source"] is t]
                main = [j for j in mine if not j.get("colsum")]
                t.direct = None
                if (self.DIRECT and t.S == 1 and len(main) == 1 and main[0]["kind"] in (EPI_SUM, EPI_LINEAR, EPI_GRAD)
                        and all(j["kind"] == EPI_SUM for j in mine if j.get("colsum")) and len(mine) <= 2):
                    t.direct = (main[0], next((j for j in mine if j.get("colsum")), None))
                    jobs = [j for j in jobs if j["source"] is not t]
                else:
                    t.slab = torch.empty(t.S, t.M, t.N, device=self.device, dtype=torch.float32)
                    if any(p["colsum"] for p in t.problems):
                        # the colsum job sums all S slabs of the target, and only problems with colsum=True write theirs:
                        # a target that mixed the two kinds would add uninitialised rows into a bias gradient
                        if not all(p["colsum"] for p in t.problems):
                            raise _lib.VqaLibraryError("phase %s: the products of one target must agree on colsum" % self.name)
                        t.colsum = torch.empty(t.S, t.M, device=self.device, dtype=torch.float32)
                base = 0
                for p in t.problems:
                    p["slab_base"] = base
                    base += p["splits"]
                    flops += 2 * t.M * t.N * p["K"]
            for lo in range(0, len(probs), MAX_GEMMS):
                chunk = probs[lo:lo + MAX_GEMMS]
                arr = (GemmProblem * len(chunk))()
                for i, (t, p) in enumerate(chunk):
                    gp = GemmProblem(_ptr(p["A"], p["a_off"]), _ptr(p["B"], p["b_off"]),
                                     t.slab.data_ptr() if t.slab is not None else None,
                                     t.colsum.data_ptr() if (p["colsum"] and t.colsum is not None) else None,
                                     t.M * t.N, p["lda"], p["ldb"], t.M, t.N, p["K"], p["form"], p["ksplit"], p["slab_base"],
                                     p["Ka"], p["Kb"], p["Ma"], p["Nb"])
                    if t.direct is not None:
                        j, jc = t.direct
                        gp.out = _ptr(j["out"], j["out_off"])
                        gp.ldo = j["ldo"]
                        if jc is not None:
                            gp.colsum_out = _ptr(jc["out"], jc["out_off"])
                        bias, aux = j.get("bias"), j.get("aux")
                        gp.bias = bias.data_ptr() if bias is not None else None
                        gp.act = int(j.get("act", 0))
                        gp.gate = int(j.get("gate", 0))
                        gp.gate_y = _ptr(aux, j.get("aux_off", 0)) if aux is not None else None
                        gp.ld_gate = int(j.get("ld_aux", 0))
                        gp.gate_scale = float(j.get("gate_scale", 1.0))
                        gp.p_drop = float(j.get("p_drop", 0.0))
                        if gp.p_drop:
                            sv, sp = ops._seed_args(j.get("seed", 0))
                            gp.seed, gp.seed_ptr = sv, (sp.value if sp is not None else None)
                            gp.drop_base, gp.drop_ld = int(j.get("drop_base", 0)), int(j.get("drop_ld", 0))
                    arr[i] = gp
                if engine == "split":
                    ops._launch("grouped_gemm_split", (self.name, len(probs), flops if lo == 0 else 0), L_.vqa_grouped_gemm_split,
                                arr, len(chunk))
                else:
                    ops._launch("grouped_gemm", (self.name, len(probs), flops if lo == 0 else 0), L_.vqa_grouped_gemm, arr, len(chunk))
        if jobs:
            for lo in range(0, len(jobs), MAX_GROUP):
                self._epilogue(L_, jobs[lo:lo + MAX_GROUP], "post")
        # break the target <-> job reference cycle: the jobs hold saved tensors of the autograd graph, and a cycle would keep
        # that graph (and its AccumulateGrad nodes) alive until the garbage collector runs -- into the next step, which is
        # what makes a later hipGraph capture of the step fail ("AccumulateGrad node's stream does not match")
        for t in self.targets:
            t.direct = None
            t.problems = []
        self.targets, self.pre_jobs, self.jobs = [], [], []

    def _epilogue(self, L_, jobs, tag):
        arr = (EpilogueJob * len(jobs))()
        elems = 0
        for i, j in enumerate(jobs):
            src = j["source"]
            if isinstance(src, _Target):
                colsum = j.get("colsum", False)
                slab = src.colsum if colsum else src.slab
                S, M, N = src.S, (1 if colsum else src.M), (src.M if colsum else src.N)
                stride = src.M if colsum else src.M * src.N
            else:
                slab, S, (M, N) = src, 1, (src.shape[0], src.numel() // src.shape[0])
                stride = M * N
            seed = j.get("seed", 0)
            sv, sp = ops._seed_args(seed) if j.get("p_drop", 0.0) else (0, None)
            aux, aux2, out2, bias = j.get("aux"), j.get("aux2"), j.get("out2"), j.get("bias")
            arr[i] = EpilogueJob(slab.data_ptr(), bias.data_ptr() if bias is not None else None,
                                 _ptr(aux, j.get("aux_off", 0)) if aux is not None else None,
                                 _ptr(aux2, j.get("aux_off", 0)) if aux2 is not None else None,
                                 _ptr(j["out"], j["out_off"]), _ptr(out2, j.get("out2_off", 0)) if out2 is not None else None,
                                 sp.value if sp is not None else None, sv, stride, S, M, N, j["kind"], j["ldo"],
                                 int(j.get("ld_aux", 0)), int(j.get("act", 0)), int(j.get("gate", 0)), int(j.get("R", 1)),
                                 int(j.get("seg", 0)), int(j.get("seg_ld", 0)), int(j.get("drop_base", 0)),
                                 int(j.get("drop_ld", 0)), float(j.get("p_drop", 0.0)), float(j.get("gate_scale", 1.0)))
            elems += M * N * (S + 1)
        ops._launch("grouped_epilogue", (self.name + "/" + tag, len(jobs), elems), L_.vqa_grouped_epilogue, arr, len(jobs))


def _note_mask(site, rows, cols, p, seed):
    if _mask_spy is not None and p:
        _mask_spy(site, rows, cols, p, seed)


def _adjacent(params):
    """The same-shaped 2-D (or 1-D) parameters lie back to back in memory (trainer.FlatState lays stack groups out so):
    they can be addressed as ONE matrix of the concatenated rows."""
    first = params[0]
    n = first.numel()
    return all(p.is_contiguous() and p.shape == first.shape and p.data_ptr() == first.data_ptr() + 4 * n * i
               for i, p in enumerate(params))


def _f32c(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise _lib.VqaLibraryError("grouped head: tensors must be contiguous fp32 GPU tensors (no CPU fallback)")


def supported(model, *dims):
    """Does `model` ("cor2" | "oda") run its [B,.]-sized layers as grouped phases under VQA_HEAD?  Every feature
    dimension that ends up as the contiguous axis of an 8-byte-loaded operand must be even."""
    on = MODE == "grouped" or (MODE == "auto" and model == "cor2")
    return on and all(int(d) % 2 == 0 for d in dims)


def glimpses_grouped():
    """The per-glimpse projections as a grouped phase (VQA_HEAD=grouped) or as one batched library GEMM (auto)."""
    return MODE == "grouped"


# ------------------------------------------------------------------------------------------------ phase 1
class QuestionProjections(torch.autograd.Function):
    """low_g = drop_g(relu(drop(q) W_g^T + b_g)) for the G MyLinear(2400 -> 310, p, relu) that read the question vector
    (config/CoR2.py:170,180,183,186 applied at :205,:193-194,:230; config/ODA.py:185,193 at :207,:233).  Each layer draws its
    own input mask over q (the reference calls them one after the other); groups listed in `dropped` are stored with the
    input dropout of THEIR consumer applied (expand_q_{1,2}, config/CoR2.py:184,187).  -> G tensors [B,A].
    The gradient of an output must arrive gated (see the module docstring) unless its group is listed in `ungated` (its
    consumer is not a phase of this module: ODA's object-difference kernel reads low_0) -- those are gated here, by a
    pre-job of the backward launch.  q gets no gradient here."""

    @staticmethod
    def forward(ctx, q, p_in, seed_in, dropped, p_out, seed_out, ungated, *params):
        G = len(params) // 2
        ws, bs = params[:G], params[G:]
        _f32c(q, *ws, *bs)
        B, K = q.shape
        A = ws[0].shape[0]
        dev = q.device
        if p_in:
            qd = ops.DropoutGroups.apply(q.detach(), p_in, seed_in, G)          # [G,B,K]: G independent draws
            _note_mask("question_in", G * B, K, p_in, seed_in)
        else:
            qd = q.detach()
        low = torch.empty(G, B, A, device=dev, dtype=torch.float32)
        ph = Phase(dev, "q_proj_fwd")
        for g in range(G):
            t = ph.target(B, A)
            ph.gemm(t, NT, qd, K, ws[g], K, K, a_off=g * B * K if p_in else 0)
            slot = dropped.index(g) if g in dropped else -1
            ph.job(EPI_LINEAR, t, low, A, out_off=g * B * A, bias=bs[g], act=1,
                   p_drop=p_out if slot >= 0 else 0.0, seed=seed_out, drop_base=max(slot, 0) * B * A, drop_ld=A)
        if p_out and dropped:
            _note_mask("question_out", len(dropped) * B, A, p_out, seed_out)
        ph.run()
        ctx.save_for_backward(qd, low, *ws, *bs)
        ctx.cfg = (G, B, K, A, bool(p_in), tuple(ungated), tuple(dropped), float(p_out))
        return tuple(low[g] for g in range(G))

    @staticmethod
    def backward(ctx, *d_lows):
        G, B, K, A, dropped_in, ungated, dropped, p_out = ctx.cfg
        qd, low = ctx.saved_tensors[:2]
        ws, bs = ctx.saved_tensors[2:2 + G], ctx.saved_tensors[2 + G:]
        dev = low.device
        ph = Phase(dev, "q_proj_bwd")
        grads_w, grads_b = [], []
        for g in range(G):
            d = d_lows[g]
            if d is None:
                grads_w.append(None)
                grads_b.append(None)
                continue
            d = d.contiguous()
            if g in ungated:
                scale = 1.0 / (1.0 - p_out) if (g in dropped and p_out) else 1.0
                gated = torch.empty(B, A, device=dev, dtype=torch.float32)
                ph.job(EPI_GRAD, d, gated, A, pre=True, gate=1, aux=low, aux_off=g * B * A, ld_aux=A, gate_scale=scale)
                d = gated
            t = ph.target(A, K)
            ph.gemm(t, TN, d, A, qd, K, B, b_off=g * B * K if dropped_in else 0, colsum=True)
            gw, gb = ops._grad_like(ws[g]), ops._grad_like(bs[g])
            ph.job(EPI_SUM, t, gw, K)
            ph.job(EPI_SUM, t, gb, A, colsum=True)
            grads_w.append(gw)
            grads_b.append(gb)
        ph.run()
        return (None, None, None, None, None, None, None, *grads_w, *grads_b)


# ------------------------------------------------------------------------------------------------ phase 2
class GatesAndRankFactors(torch.autograd.Function):
    """Everything that reads the projected question: the sigmoid gates expand_q_{1,2}(low_2), (low_3) (config/CoR2.py:
    184,187: MyLinear(310 -> 2048, p, sigmoid); their input dropout is already in the stored low) and the question-side rank
    factors h2 = Linear2_r(q) of the Mutan fusions (putils/__init__.py:232-238), each reading one of the n_low tensors.
    gate_groups: which low tensor each gate reads; fusions: (low index, R) per fusion; gate_scale[g]: 1, or 1/(1-p) when
    low_g is stored dropped out.  -> (gate outputs [B,D] ..., h2 [B,R,H] per fusion).
    Returns the gradient of every low tensor it read already gated by that tensor's relu / dropout (None for the others)."""

    @staticmethod
    def forward(ctx, n_low, gate_groups, fusions, gate_scale, *rest):
        # rest: the n_low tensors [B,A]; then per gate (W [D,A], b [D]); then per fusion R weights [H,A] and R biases [H]
        lows, params = rest[:n_low], rest[n_low:]
        B, A = lows[0].shape
        dev = lows[0].device
        _f32c(*lows, *params)
        ph = Phase(dev, "gates_h2_fwd")
        outs, idx = [], 0
        for g in gate_groups:
            W, b = params[idx], params[idx + 1]
            idx += 2
            D = W.shape[0]
            y = torch.empty(B, D, device=dev, dtype=torch.float32)
            t = ph.target(B, D)
            ph.gemm(t, NT, lows[g], A, W, A, A)
            ph.job(EPI_LINEAR, t, y, D, bias=b, act=2)
            outs.append(y)
        for g, R in fusions:
            ws, bs = params[idx:idx + R], params[idx + R:idx + 2 * R]
            idx += 2 * R
            H = ws[0].shape[0]
            h2 = torch.empty(B, R, H, device=dev, dtype=torch.float32)
            if _adjacent(ws) and _adjacent(bs):
                t = ph.target(B, R * H)
                ph.gemm(t, NT, lows[g], A, ws[0], A, A)
                ph.job(EPI_LINEAR, t, h2, R * H, bias=bs[0].as_strided((R * H,), (1,)))
            else:
                for r in range(R):
                    t = ph.target(B, H)
                    ph.gemm(t, NT, lows[g], A, ws[r], A, A)
                    ph.job(EPI_LINEAR, t, h2, R * H, out_off=r * H, bias=bs[r])
            outs.append(h2)
        ph.run()
        ng = len(gate_groups)
        ctx.save_for_backward(*lows, *outs[:ng], *params)
        ctx.cfg = (n_low, B, A, tuple(gate_groups), tuple(fusions), tuple(gate_scale))
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        G, B, A, gate_groups, fusions, gate_scale = ctx.cfg
        lows = ctx.saved_tensors[:G]
        ng = len(gate_groups)
        ys = ctx.saved_tensors[G:G + ng]
        params = ctx.saved_tensors[G + ng:]
        dev = lows[0].device
        ph = Phase(dev, "gates_h2_bwd")
        d_low_t = [ph.target(B, A) for _ in range(G)]
        out_grads, idx = [], 0
        # gates: dg = d_y * y (1 - y) first (an elementwise pre-job), then dg W (data) and dg^T low_g (weight)
        for i, g in enumerate(gate_groups):
            W, b = params[idx], params[idx + 1]
            idx += 2
            D = W.shape[0]
            gy = grads[i]
            if gy is None:
                out_grads += [None, None]
                continue
            gy = gy.contiguous()
            dg = torch.empty(B, D, device=dev, dtype=torch.float32)
            ph.job(EPI_GRAD, gy, dg, D, pre=True, gate=2, aux=ys[i], ld_aux=D)
            ph.gemm(d_low_t[g], NN, dg, D, W, A, D)
            t = ph.target(D, A)
            ph.gemm(t, TN, dg, D, lows[g], A, B, colsum=True)
            gw, gb = ops._grad_like(W), ops._grad_like(b)
            ph.job(EPI_SUM, t, gw, A)
            ph.job(EPI_SUM, t, gb, D, colsum=True)
            out_grads += [gw, gb]
        for k, (g, R) in enumerate(fusions):
            ws, bs = params[idx:idx + R], params[idx + R:idx + 2 * R]
            idx += 2 * R
            H = ws[0].shape[0]
            gh = grads[ng + k]
            if gh is None:
                out_grads += [None] * (2 * R)
                continue
            gh = gh.contiguous()           # [B,R,H]
            gws = [ops._grad_like(w) for w in ws]
            gbs = [ops._grad_like(b) for b in bs]
            if _adjacent(ws):
                ph.gemm(d_low_t[g], NN, gh, R * H, ws[0], A, R * H)
            else:
                for r in range(R):
                    ph.gemm(d_low_t[g], NN, gh, R * H, ws[r], A, H, a_off=r * H)
            if _adjacent(ws) and _adjacent(gws) and _adjacent(bs) and _adjacent(gbs):
                t = ph.target(R * H, A)
                ph.gemm(t, TN, gh, R * H, lows[g], A, B, colsum=True)
                ph.job(EPI_SUM, t, gws[0], A)
                ph.job(EPI_SUM, t, gbs[0], R * H, colsum=True)
            else:
                for r in range(R):
                    t = ph.target(H, A)
                    ph.gemm(t, TN, gh, R * H, lows[g], A, B, a_off=r * H, colsum=True)
                    ph.job(EPI_SUM, t, gws[r], A)
                    ph.job(EPI_SUM, t, gbs[r], H, colsum=True)
            out_grads += gws + gbs
        d_lows = []
        for g in range(G):
            if d_low_t[g].problems:    # gate: relu of the producer; a dropped-out tensor's stored value carries mask and factor
                d = torch.empty(B, A, device=dev, dtype=torch.float32)
                ph.job(EPI_GRAD, d_low_t[g], d, A, gate=1, aux=lows[g], ld_aux=A, gate_scale=gate_scale[g])
                d_lows.append(d)
            else:
                d_lows.append(None)
        ph.targets = [t for t in ph.targets if t.problems]
        ph.run()
        return (None, None, None, None, *d_lows, *out_grads)


# ------------------------------------------------------------------------------------------------ phase 3 / 4
class GlimpseProjections(torch.autograd.Function):
    """x_v = cat_g relu(W_g pooled[:, g, :] + b_g): MyATT's per-glimpse MyLinear list (config/CoR2.py:133-134,143-147) on the
    pooled features, whose input dropout the producer has applied already (K3 / the relation map).  -> [B, G*A].
    Expects its output's gradient already gated by the relu (the fusion phase that consumes x_v does that)."""

    @staticmethod
    def forward(ctx, pooled, *params):
        G = len(params) // 2
        ws, bs = params[:G], params[G:]
        _f32c(pooled, *ws, *bs)
        B, G2, D = pooled.shape
        if G2 != G:
            raise ValueError("glimpse projections: pooled has %d glimpses, %d layers given" % (G2, G))
        A = ws[0].shape[0]
        out = torch.empty(B, G * A, device=pooled.device, dtype=torch.float32)
        ph = Phase(pooled.device, "glimpse_fwd")
        for g in range(G):
            t = ph.target(B, A)
            ph.gemm(t, NT, pooled, G * D, ws[g], D, D, a_off=g * D)
            ph.job(EPI_LINEAR, t, out, G * A, out_off=g * A, bias=bs[g], act=1)
        ph.run()
        ctx.save_for_backward(pooled, *ws, *bs)
        ctx.cfg = (B, G, D, A)
        return out

    @staticmethod
    def backward(ctx, d_pre):
        B, G, D, A = ctx.cfg
        pooled = ctx.saved_tensors[0]
        ws, bs = ctx.saved_tensors[1:1 + G], ctx.saved_tensors[1 + G:]
        d_pre = d_pre.contiguous()                    # [B, G*A], gated
        dev = d_pre.device
        d_pooled = torch.empty(B, G, D, device=dev, dtype=torch.float32) if ctx.needs_input_grad[0] else None
        ph = Phase(dev, "glimpse_bwd")
        gws, gbs = [], []
        odd = A % 2 != 0      # a 155-wide block of the [B,620] gradient: 4-byte aligned operand -> the scalar-load forms
        for g in range(G):
            if d_pooled is not None:
                t = ph.target(B, D)
                ph.gemm(t, NN_A4 if odd else NN, d_pre, G * A, ws[g], D, A, a_off=g * A)
                ph.job(EPI_GRAD, t, d_pooled, G * D, out_off=g * D)
            t = ph.target(A, D)
            ph.gemm(t, TN_A4 if odd else TN, d_pre, G * A, pooled, G * D, B, a_off=g * A, b_off=g * D, colsum=True)
            gw, gb = ops._grad_like(ws[g]), ops._grad_like(bs[g])
            ph.job(EPI_SUM, t, gw, D)
            ph.job(EPI_SUM, t, gb, A, colsum=True)
            gws.append(gw)
            gbs.append(gb)
        ph.run()
        return (d_pooled, *gws, *gbs)


# ------------------------------------------------------------------------------------------------ phase 5
class VectorFusion(torch.autograd.Function):
    """x = drop(sum_r (W1_r [xs...] + b1_r) * h2[:, r, :]): the vector-vector Mutan fusion (fusion_final, putils/__init__.py:
    232-238 with 2-D inputs; config/CoR2.py:182,236, config/ODA.py:197,239) whose first input is the concatenation of
    `xs` (never materialised: each part contracts against its column block of W1), with the classifier's input dropout
    (config/CoR2.py:106-109 at :189) applied to the stored result.  xs are relu outputs of glimpse phases: their gradients
    are returned gated.  Expects d x already multiplied by the dropout mask (the classifier phase does that)."""

    @staticmethod
    def forward(ctx, h2, p_out, seed_out, n_x, *rest):
        xs = rest[:n_x]
        params = rest[n_x:]
        R = len(params) // 2
        ws, bs = params[:R], params[R:]
        _f32c(h2, *xs, *ws, *bs)
        B = h2.shape[0]
        H, Ktot = ws[0].shape
        dev = h2.device
        h1 = torch.empty(B, R, H, device=dev, dtype=torch.float32)
        x = torch.empty(B, H, device=dev, dtype=torch.float32)
        ph = Phase(dev, "vector_fusion_fwd")
        adj = _adjacent(ws) and _adjacent(bs)
        if adj:
            t = ph.target(B, R * H)
            off = 0
            for xpart in xs:
                k = xpart.shape[1]
                ph.gemm(t, NT, xpart, k, ws[0], Ktot, k, b_off=off)
                off += k
            ph.job(EPI_RANK_PRODUCT, t, h1, H, bias=bs[0].as_strided((R * H,), (1,)), R=R, aux=h2, ld_aux=R * H, out2=x,
                   p_drop=p_out, seed=seed_out, drop_ld=H)
        else:
            # ranks not contiguous in memory (no flat parameter buffer): one linear job per rank, then the rank product as
            # an elementwise job of its own
            for r in range(R):
                t = ph.target(B, H)
                off = 0
                for xpart in xs:
                    k = xpart.shape[1]
                    ph.gemm(t, NT, xpart, k, ws[r], Ktot, k, b_off=off)
                    off += k
                ph.job(EPI_LINEAR, t, h1, R * H, out_off=r * H, bias=bs[r])
        if p_out:
            _note_mask("fusion_out", B, H, p_out, seed_out)
        ph.run()
        if not adj:
            ph2 = Phase(dev, "vector_fusion_fwd")
            zero = torch.zeros(R * H, device=dev, dtype=torch.float32)
            ph2.job(EPI_RANK_PRODUCT, h1.view(B, R * H), torch.empty_like(h1), H, bias=zero, R=R, aux=h2, ld_aux=R * H, out2=x,
                    p_drop=p_out, seed=seed_out, drop_ld=H)
            ph2.run()
        ctx.save_for_backward(h1, h2, *xs, *ws, *bs)
        ctx.cfg = (B, R, H, Ktot, n_x)
        return x

    @staticmethod
    def backward(ctx, d_x):
        B, R, H, Ktot, n_x = ctx.cfg
        h1, h2 = ctx.saved_tensors[:2]
        xs = ctx.saved_tensors[2:2 + n_x]
        ws = ctx.saved_tensors[2 + n_x:2 + n_x + R]
        bs = ctx.saved_tensors[2 + n_x + R:]
        dev = d_x.device
        d_x = d_x.contiguous()
        d_h1 = torch.empty(B, R, H, device=dev, dtype=torch.float32)
        d_h2 = torch.empty(B, R, H, device=dev, dtype=torch.float32)
        ph = Phase(dev, "vector_fusion_bwd")
        ph.job(EPI_RANK_PRODUCT_BWD, d_x, d_h1, R * H, pre=True, R=R, aux=h2, aux2=h1, ld_aux=R * H, out2=d_h2)
        gws = [ops._grad_like(w) for w in ws]
        gbs = [ops._grad_like(b) for b in bs]
        adj = _adjacent(ws) and _adjacent(gws) and _adjacent(bs) and _adjacent(gbs)
        d_xs = []
        off = 0
        for i, xpart in enumerate(xs):
            k = xpart.shape[1]
            t = ph.target(B, k)
            if adj:
                ph.gemm(t, NN, d_h1, R * H, ws[0], Ktot, R * H, b_off=off)
            else:
                for r in range(R):
                    ph.gemm(t, NN, d_h1, R * H, ws[r], Ktot, H, a_off=r * H, b_off=off)
            dx = torch.empty(B, k, device=dev, dtype=torch.float32)
            ph.job(EPI_GRAD, t, dx, k, gate=1, aux=xpart, ld_aux=k)             # relu gate of the glimpse phase's output
            d_xs.append(dx)
            if adj:
                tw = ph.target(R * H, k)
                ph.gemm(tw, TN, d_h1, R * H, xpart, k, B, colsum=(i == 0))
                ph.job(EPI_SUM, tw, gws[0], Ktot, out_off=off)
                if i == 0:
                    ph.job(EPI_SUM, tw, gbs[0], R * H, colsum=True)
            else:
                for r in range(R):
                    tw = ph.target(H, k)
                    ph.gemm(tw, TN, d_h1, R * H, xpart, k, B, a_off=r * H, colsum=(i == 0))
                    ph.job(EPI_SUM, tw, gws[r], Ktot, out_off=off)
                    if i == 0:
                        ph.job(EPI_SUM, tw, gbs[r], H, colsum=True)
            off += k
        ph.run()
        return (d_h2, None, None, None, *d_xs, *gws, *gbs)


# ------------------------------------------------------------------------------------------------ phase 6
class Classifier(torch.autograd.Function):
    """logits = W x + b for x stored already dropped out (linear_classif, config/CoR2.py:189,237: MyLinear(510 -> C, p)).
    Returns d x multiplied by x's dropout mask (regenerated from p / seed): the gradient at the fusion's undropped
    output."""

    @staticmethod
    def forward(ctx, x, w, b, p_x, seed_x):
        _f32c(x, w, b)
        B, K = x.shape
        C = w.shape[0]
        logits = torch.empty(B, C, device=x.device, dtype=torch.float32)
        ph = Phase(x.device, "classifier_fwd")
        t = ph.target(B, C)
        ph.gemm(t, NT, x, K, w, K, K)
        ph.job(EPI_LINEAR, t, logits, C, bias=b)
        ph.run()
        ctx.save_for_backward(x, w, b)
        ctx.cfg = (B, K, C, float(p_x), seed_x)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        B, K, C, p_x, seed_x = ctx.cfg
        x, w, b = ctx.saved_tensors
        d_logits = d_logits.contiguous()
        dev = x.device
        ph = Phase(dev, "classifier_bwd")
        d_x = None
        if ctx.needs_input_grad[0]:
            t = ph.target(B, K)
            ph.gemm(t, NN, d_logits, C, w, K, C)
            d_x = torch.empty(B, K, device=dev, dtype=torch.float32)
            ph.job(EPI_GRAD, t, d_x, K, p_drop=p_x, seed=seed_x, drop_ld=K)
        tw = ph.target(C, K)
        ph.gemm(tw, TN, d_logits, C, x, K, B, colsum=True)
        gw, gb = ops._grad_like(w), ops._grad_like(b)
        ph.job(EPI_SUM, tw, gw, K)
        ph.job(EPI_SUM, tw, gb, C, colsum=True)
        ph.run()
        return d_x, gw, gb, None, None
