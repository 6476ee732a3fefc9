"""torch.autograd.Function wrappers around the C ABI (include/vqa_mi355x.h).

PyTorch is plumbing here: it owns device memory and the stream; every forward/backward
below is one call into libvqa_mi355x.so on ``torch.cuda.current_stream()``.  Tensors must be
fp32 CUDA(=HIP) tensors -- or, for the region-side tensors of the mixed-precision path (BASELINE
configs[4]), bf16, which selects the ``*_bf16`` entry points; anything else raises -- there is no
eager/CPU fallback.
"""
import ctypes
import os
import threading

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class KernelTimer:
    """Optional per-launch timing with HIP events recorded on the stream the kernels are launched on
    (torch's current stream).  bench.py installs one around its timed region; None = no overhead."""

    def __init__(self, names=None):
        self.names = set(names) if names else None
        self.events = {}
        self.grids = {}       # (name, shape) -> ((grid in work-items, kernel expression), ...) of the kernels the call launched

    def wants(self, name):
        return self.names is None or name in self.names

    def summary(self):
        """{(name, shape): (launches, mean_ms)} -- call after torch.cuda.synchronize()."""
        out = {}
        for key, evs in self.events.items():
            ms = [a.elapsed_time(b) for a, b in evs]
            out[key] = (len(ms), sum(ms) / max(len(ms), 1))
        return out


_timer = None


def set_kernel_timer(timer):
    global _timer
    _timer = timer


def _launch(name, shape, fn, *args):
    """Call one C-ABI launcher on the current stream, check its return code, optionally time it."""
    t = _timer
    if t is not None and t.wants(name):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        handle = _lib.lib()
        handle.vqa_launch_log_reset()
        a.record()
        rc = fn(*args, _stream())
        b.record()
        t.events.setdefault((name, shape), []).append((a, b))
        if (name, shape) not in t.grids:      # the device kernels behind this call: grid sizes in work-items, launch order
            buf = (ctypes.c_ulonglong * 16)()
            n = handle.vqa_launch_log(buf, 16)
            t.grids[(name, shape)] = tuple((int(buf[i]), (handle.vqa_launch_log_kernel(i) or b"").decode())
                                           for i in range(min(n, 16)))
    else:
        rc = fn(*args, _stream())
    _lib.check(rc, name)


class _Timed:
    """Event pair around library (torch / hipBLASLt) ops on the current stream, so that bench.py's per-kernel table covers
    the whole step and not only the C-ABI launches.  A no-op unless a KernelTimer is installed."""

    __slots__ = ("key", "a")

    def __init__(self, name, shape):
        self.key = (name, tuple(int(x) for x in shape))
        self.a = None

    def __enter__(self):
        t = _timer
        if t is not None and t.wants(self.key[0]):
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if self.a is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            t = _timer
            if t is not None:
                t.events.setdefault(self.key, []).append((self.a, b))
        return False


def timed(name, shape):
    return _Timed(name, shape)


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _prep(name, t, dtypes=(torch.float32,)):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _lib.VqaLibraryError(
            "%s must be a GPU tensor: the MI355X HIP path has no CPU fallback (got %s)"
            % (name, t.device if isinstance(t, torch.Tensor) else type(t)))
    if t.dtype not in dtypes:
        raise _lib.VqaLibraryError("%s must be %s, got %s" % (name, " or ".join(str(d) for d in dtypes), t.dtype))
    return t.contiguous()


_REGION_DTYPES = (torch.float32, torch.bfloat16)   # storage types of the region-side tensors
BF16_PAD = 64                                       # feature dims of the bf16 path are zero-padded to this multiple


def _sfx(dtype):
    return "_bf16" if dtype == torch.bfloat16 else ""


def pad_to(n, multiple=BF16_PAD):
    return (n + multiple - 1) // multiple * multiple


# ---- gradient slots ---------------------------------------------------------------------------------------------
# A trainer that keeps all parameters in ONE flat buffer and all gradients in another (trainer.FlatState) registers the
# pair here.  The backward of every op that produces a parameter gradient then asks `_grad_like(w)` for its output
# tensor: when `w` lives in the flat parameter buffer the answer is the view of the flat GRADIENT buffer at the same
# offset, so the kernel writes the gradient where the all-reduce / clip / Adam kernels read it.  autograd's
# AccumulateGrad adopts a freshly produced gradient as `.grad` without a copy, so the per-step gather of ~70 gradients
# (one 48 MB multi-tensor copy, 31 us at B = 512) has nothing left to move.  A slot is handed out once per backward
# pass (`begin_backward`): a parameter used twice gets a fresh tensor the second time and autograd adds it in.
# Registrations are keyed by the flat parameter buffer (several trainers -- one per replica thread, say -- may be inside
# their backward windows at once; autograd runs the nodes on its own per-device threads, so the calling thread says nothing):
# `_grad_like` looks for the registration whose parameter buffer holds `w`, and ending one trainer's window neither drops
# another's registration nor resets its hand-out list.
_slot_lock = threading.Lock()
_slot_regs = {}       # (device index, p_flat.data_ptr()) -> [p_flat, g_flat, [(lo, hi) handed out this backward pass]]


def _slot_key(p_flat):
    return (p_flat.device.index, p_flat.data_ptr())


def set_grad_slots(p_flat, g_flat):
    """Register the flat parameter / gradient buffer pair whose offsets correspond (g_flat None: drop p_flat's
    registration).  A fresh registration starts with nothing handed out."""
    with _slot_lock:
        if g_flat is None:
            if p_flat is not None:
                _slot_regs.pop(_slot_key(p_flat), None)
        else:
            _slot_regs[_slot_key(p_flat)] = [p_flat, g_flat, []]


def begin_backward(p_flat=None):
    """Forget which slots were handed out (of p_flat's registration; of all when None)."""
    with _slot_lock:
        for key, reg in _slot_regs.items():
            if p_flat is None or key == _slot_key(p_flat):
                reg[2] = []


def _grad_like(w, rows_strided=False):
    """rows_strided: a 2-D `w` whose rows are contiguous but spaced (a stack of odd-sized biases, padded apart in the flat
    buffer) is matched too -- for callers whose kernel takes a row stride."""
    if _slot_regs and w.dtype == torch.float32 and w.numel() > 0:
        dense = w.is_contiguous()
        spaced = (not dense) and rows_strided and w.dim() == 2 and w.stride(1) == 1 and w.stride(0) >= w.shape[1]
        if dense or spaced:
            n = w.numel() if dense else (w.shape[0] - 1) * w.stride(0) + w.shape[1]
            ptr, dev = w.data_ptr(), w.device.index
            with _slot_lock:
                for (rdev, base), (p_flat, g_flat, taken) in _slot_regs.items():
                    off = ptr - base
                    if rdev != dev or off < 0 or off % 4 or off // 4 + n > p_flat.numel():
                        continue            # `w` does not live in this registration's parameter buffer
                    lo = off // 4
                    if all(lo + n <= a or lo >= b for a, b in taken):
                        taken.append((lo, lo + n))
                        if dense:
                            return g_flat[lo:lo + n].view(w.shape)
                        return g_flat.as_strided(tuple(w.shape), tuple(w.stride()), lo)
                    break                   # a second use in this pass: a fresh tensor, autograd adds it in
    return torch.empty_like(w) if w.is_contiguous() else torch.empty(w.shape, device=w.device, dtype=w.dtype)


def _seed_args(seed):
    """seed: an int (host seed), or (device int64 tensor [1], int salt) -> (c_uint64 value, device pointer or None).
    With a device tensor the kernels read the step's seed at run time (*tensor + salt), which is what lets a captured
    hipGraph draw a fresh mask on every replay."""
    if isinstance(seed, tuple):
        tensor, salt = seed
        if not (isinstance(tensor, torch.Tensor) and tensor.is_cuda and tensor.dtype == torch.int64 and tensor.numel() >= 1):
            raise ValueError("device seed must be a CUDA int64 tensor")
        return int(salt) & 0x7FFFFFFFFFFFFFFF, ctypes.c_void_p(tensor.data_ptr())
    return int(seed) & 0xFFFFFFFFFFFFFFFF, None


def _ptr_array(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


class PairwiseRelationReduce(torch.autograd.Function):
    """K1.  v2[b,j,:] = sum_i alpha[b,i,glimpse] * (v[b,i,:]*q1[b,:] + v[b,j,:]*q2[b,:]).
    Replaces config/CoR2.py:191-199 + :216."""

    @staticmethod
    def forward(ctx, v, q1, q2, alpha, glimpse, mode, dual):
        v, q1, q2, alpha = _prep("v", v, _REGION_DTYPES), _prep("q1", q1), _prep("q2", q2), _prep("alpha", alpha)
        B, N, D = v.shape
        if alpha.dim() != 3 or alpha.shape[0] != B or alpha.shape[1] != N or not 0 <= glimpse < alpha.shape[2]:
            raise ValueError("alpha must be [B,N,G] with glimpse < G, got %s" % (tuple(alpha.shape),))
        if q1.shape != (B, D) or q2.shape != (B, D):
            raise ValueError("q1/q2 must be [B,D]")
        G = alpha.shape[2]
        v2 = torch.empty_like(v)
        a_ptr = ctypes.c_void_p(alpha.data_ptr() + 4 * glimpse)
        name = "pairwise_relation_reduce_fwd" + _sfx(v.dtype)
        _launch(name, (B, N, D, int(mode)), getattr(_lib.lib(), "vqa_" + name),
                _p(v), _p(q1), _p(q2), a_ptr, G, _p(v2), B, N, D, int(mode))
        ctx.save_for_backward(v, q1, q2, alpha)
        ctx.glimpse = glimpse
        ctx.set_materialize_grads(False)
        if dual:   # two aliases of the one result: autograd hands their gradients back separately (summed in the kernel)
            return v2, v2.view_as(v2)
        return v2

    @staticmethod
    def backward(ctx, g, g_b=None):
        v, q1, q2, alpha = ctx.saved_tensors
        if g is None:
            g, g_b = g_b, None
        if g is None:
            return None, None, None, None, None, None, None
        g = _prep("grad_v2", g.to(v.dtype), _REGION_DTYPES)
        g_b = _prep("grad_v2_b", g_b.to(v.dtype), _REGION_DTYPES) if g_b is not None else None
        B, N, D = v.shape
        G = alpha.shape[2]
        d_alpha = torch.empty(B, N, device=v.device, dtype=torch.float32)
        d_q1 = torch.empty_like(q1)
        d_q2 = torch.empty_like(q2)
        d_v = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        a_ptr = ctypes.c_void_p(alpha.data_ptr() + 4 * ctx.glimpse)
        name = "pairwise_relation_reduce_bwd" + _sfx(v.dtype)
        _launch(name, (B, N, D, d_v is not None, g_b is not None), getattr(_lib.lib(), "vqa_" + name),
                _p(v), _p(q1), _p(q2), a_ptr, G, _p(g), _p(g_b), _p(d_alpha), _p(d_q1), _p(d_q2), _p(d_v), B, N, D)
        d_alpha_full = torch.zeros_like(alpha)
        d_alpha_full[:, :, ctx.glimpse] = d_alpha
        return d_v, d_q1, d_q2, d_alpha_full, None, None, None


class RelationApply(torch.autograd.Function):
    """K1, closed form with the pooled feature given: out[b,n,:] = keep * (t[b,:] + c2[b,:] * v[b,n,:]) -- the relation
    step of config/CoR2.py:191-199,216 when s = sum_i alpha_i v_i comes from the first attention's pooled output
    (t = q1 * s, c2 = q2 for a softmax alpha), with the dropout of the second compress layer (config/CoR2.py:72-75 at
    :218) applied in the same pass.  v [B,N,D] fp32 or bf16; t, c2 [B,D] fp32."""

    @staticmethod
    def forward(ctx, v, t, c2, p_drop, seed):
        v, t, c2 = _prep("v", v, _REGION_DTYPES), _prep("t", t), _prep("c2", c2)
        B, N, D = v.shape
        if t.shape != (B, D) or c2.shape != (B, D):
            raise ValueError("relation_apply: t and c2 must be [B,D] = %s" % ((B, D),))
        out = torch.empty_like(v)
        sv, sp = _seed_args(seed)
        name = "relation_apply_fwd" + _sfx(v.dtype)
        _launch(name, (B, N, D, float(p_drop) > 0), getattr(_lib.lib(), "vqa_" + name), _p(v), _p(t), _p(c2), _p(out),
                float(p_drop), sv, sp, B, N, D)
        ctx.save_for_backward(v, c2)
        ctx.cfg = (float(p_drop), seed)
        return out

    @staticmethod
    def backward(ctx, g):
        v, c2 = ctx.saved_tensors
        p_drop, seed = ctx.cfg
        B, N, D = v.shape
        g = _prep("grad_out", g.to(v.dtype), _REGION_DTYPES)
        d_t = torch.empty(B, D, device=v.device, dtype=torch.float32)
        d_c2 = torch.empty(B, D, device=v.device, dtype=torch.float32)
        d_v = torch.empty_like(v) if ctx.needs_input_grad[0] else None
        sv, sp = _seed_args(seed)
        name = "relation_apply_bwd" + _sfx(v.dtype)
        _launch(name, (B, N, D, p_drop > 0, d_v is not None), getattr(_lib.lib(), "vqa_" + name), _p(v), _p(c2), _p(g),
                _p(d_t), _p(d_c2), _p(d_v), p_drop, sv, sp, B, N, D)
        return d_v, d_t, d_c2, None, None


def relation_apply(v, t, c2, p_drop=0.0, seed=0):
    return RelationApply.apply(v, t, c2, p_drop, seed)


class SoftmaxAttentionPool(torch.autograd.Function):
    """K3.  alpha = softmax over regions of logits [B,N,G]; pooled[b,g,:] = sum_n alpha[b,n,g] v[b,n,:].
    Replaces F.softmax(dim=1) (config/CoR2.py:83-87,:132) + putils.bmatmul (config/CoR2.py:142)."""

    @staticmethod
    def forward(ctx, logits, v):
        logits, v = _prep("logits", logits), _prep("v", v, _REGION_DTYPES)
        B, N, G = logits.shape
        if v.dim() != 3 or v.shape[0] != B or v.shape[1] != N:
            raise ValueError("v must be [B,N,D] matching logits [B,N,G]")
        D = v.shape[2]
        alpha = torch.empty_like(logits)
        pooled = torch.empty(B, G, D, device=v.device, dtype=torch.float32)
        name = "softmax_attention_pool_fwd" + _sfx(v.dtype)
        _launch(name, (B, N, D, G), getattr(_lib.lib(), "vqa_" + name), _p(logits), _p(v), _p(alpha), _p(pooled), B, N, D, G)
        ctx.save_for_backward(alpha, v)
        ctx.set_materialize_grads(False)
        return alpha, pooled

    @staticmethod
    def backward(ctx, d_alpha, d_pooled):
        alpha, v = ctx.saved_tensors
        B, N, G = alpha.shape
        D = v.shape[2]
        if d_pooled is None:
            d_pooled = torch.zeros(B, G, D, device=v.device, dtype=torch.float32)
        d_pooled = _prep("grad_pooled", d_pooled)
        d_alpha = _prep("grad_alpha", d_alpha) if d_alpha is not None else None
        d_logits = torch.empty_like(alpha)
        d_v = torch.empty_like(v) if ctx.needs_input_grad[1] else None
        name = "softmax_attention_pool_bwd" + _sfx(v.dtype)
        _launch(name, (B, N, D, G, d_v is not None), getattr(_lib.lib(), "vqa_" + name),
                _p(alpha), _p(v), _p(d_pooled), _p(d_alpha), _p(d_logits), _p(d_v), B, N, D, G)
        return d_logits, d_v


class SoftmaxAttentionPoolDrop(torch.autograd.Function):
    """K3 + the input dropout of MyATT's glimpse projections (config/CoR2.py:143-147) + the undropped glimpse 0 for CoR2's
    relation step: (alpha, pooled_dropped [B,G,D], first [B,D] or None).  Replaces softmax_attention_pool followed by a
    dropout pass over [B,G,D] and, backward, the mask multiply, a clone of the full gradient and the slice add."""

    @staticmethod
    def forward(ctx, logits, v, p_drop, seed, want_first):
        logits, v = _prep("logits", logits), _prep("v", v, _REGION_DTYPES)
        B, N, G = logits.shape
        if v.dim() != 3 or v.shape[0] != B or v.shape[1] != N:
            raise ValueError("v must be [B,N,D] matching logits [B,N,G]")
        D = v.shape[2]
        alpha = torch.empty_like(logits)
        pooled = torch.empty(B, G, D, device=v.device, dtype=torch.float32)
        first = torch.empty(B, D, device=v.device, dtype=torch.float32) if want_first else None
        sv, sp = _seed_args(seed)
        name = "softmax_attention_pool_drop_fwd" + _sfx(v.dtype)
        _launch(name, (B, N, D, G), getattr(_lib.lib(), "vqa_" + name), _p(logits), _p(v), _p(alpha), _p(pooled), _p(first),
                float(p_drop), sv, sp, B, N, D, G)
        ctx.save_for_backward(alpha, v)
        ctx.cfg = (float(p_drop), seed)
        ctx.set_materialize_grads(False)
        if first is None:
            return alpha, pooled
        return alpha, pooled, first

    @staticmethod
    def backward(ctx, d_alpha, d_pooled, d_first=None):
        alpha, v = ctx.saved_tensors
        p_drop, seed = ctx.cfg
        B, N, G = alpha.shape
        D = v.shape[2]
        if d_pooled is None:
            d_pooled = torch.zeros(B, G, D, device=v.device, dtype=torch.float32)
        d_pooled = _prep("grad_pooled", d_pooled)
        d_alpha = _prep("grad_alpha", d_alpha) if d_alpha is not None else None
        d_first = _prep("grad_first", d_first) if d_first is not None else None
        d_logits = torch.empty_like(alpha)
        d_v = torch.empty_like(v) if ctx.needs_input_grad[1] else None
        sv, sp = _seed_args(seed)
        name = "softmax_attention_pool_drop_bwd" + _sfx(v.dtype)
        _launch(name, (B, N, D, G, d_v is not None), getattr(_lib.lib(), "vqa_" + name), _p(alpha), _p(v), _p(d_pooled),
                _p(d_first), _p(d_alpha), _p(d_logits), _p(d_v), p_drop, sv, sp, B, N, D, G)
        return d_logits, d_v, None, None, None


def softmax_attention_pool_drop(logits, v, p_drop=0.0, seed=0, want_first=False):
    return SoftmaxAttentionPoolDrop.apply(logits, v, p_drop, seed, want_first)


class AttentionLogits(torch.autograd.Function):
    """K3a.  logits[..., g] = bias[g] + sum_k w[g,k] * keep * x[..., k] -- the dropout + 1x1 conv in front of MyATT's
    softmax (config/CoR2.py:72-82 as configured at :132) in one pass over x; x fp32 [.., K] or bf16 [.., Kp >= K]."""

    @staticmethod
    def forward(ctx, x, w, bias, p_drop, seed):
        x, w, bias = _prep("x", x, _REGION_DTYPES), _prep("w", w), _prep("bias", bias)
        G, K = w.shape
        ld = x.shape[-1]
        M = x.numel() // ld
        if bias.shape != (G,) or ld < K:
            raise ValueError("attention_logits: w must be [G,K], bias [G], x's last dim >= K (got %s, %s, %s)"
                             % (tuple(w.shape), tuple(bias.shape), tuple(x.shape)))
        logits = torch.empty(*x.shape[:-1], G, device=x.device, dtype=torch.float32)
        sv, sp = _seed_args(seed)
        name = "attention_logits_fwd" + _sfx(x.dtype)
        _launch(name, (M, K, G, float(p_drop) > 0), getattr(_lib.lib(), "vqa_" + name), _p(x), ld, _p(w), _p(bias),
                _p(logits), float(p_drop), sv, sp, M, K, G)
        ctx.save_for_backward(x, w)
        ctx.bias = bias
        ctx.cfg = (float(p_drop), seed, M, K, G, ld)
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        x, w = ctx.saved_tensors
        p_drop, seed, M, K, G, ld = ctx.cfg
        d_logits = _prep("grad_logits", d_logits)
        d_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        d_w = _grad_like(w)
        d_b = _grad_like(ctx.bias)
        L_ = _lib.lib()
        ws_bytes = L_.vqa_attention_logits_bwd_workspace_bytes(M, K, G)
        ws = torch.empty((ws_bytes + 3) // 4, device=x.device, dtype=torch.float32)
        sv, sp = _seed_args(seed)
        name = "attention_logits_bwd" + _sfx(x.dtype)
        _launch(name, (M, K, G, p_drop > 0, d_x is not None), getattr(L_, "vqa_" + name), _p(x), ld, _p(w), _p(d_logits),
                _p(d_x), _p(d_w), _p(d_b), _p(ws), ws_bytes, p_drop, sv, sp, M, K, G)
        return d_x, d_w, d_b, None, None


def attention_logits(x, w, bias, p_drop=0.0, seed=0):
    return AttentionLogits.apply(x, w, bias, p_drop, seed)


def attention_logits_supported(x, in_features, out_features):
    return x.is_cuda and out_features <= 8 and in_features <= 512 and in_features % 2 == 0 and \
        x.shape[-1] % (4 if x.dtype == torch.bfloat16 else 2) == 0


# "auto": rank-folded K4 where it is supported and the batch gives its per-sample workgroups enough to do (a handful of
# samples runs faster on the R-GEMM tile-engine kernels); "folded": wherever supported; "engine": never
_K4_FORM = os.environ.get("VQA_K4_FORM", "auto")
_K4_FOLD_MIN_BATCH = 32


class LowRankBilinearFusion(torch.autograd.Function):
    """K4.  out[b,n,:] = sum_r (x[b,n,:] W1_r^T + b1_r) * h2[b,r,:]   (x may also be [B,L]).
    Replaces the region side of putils.MutanFusion.forward (putils/__init__.py:232-238)."""

    @staticmethod
    def forward(ctx, x, h2, gate_dx, *params):
        # gate_dx: x is the relu output of the layer in front (compress_v / compress_v2); d_x comes back already multiplied
        # by (x > 0), that layer's own relu gradient (in the store of the data-gradient kernel where the folded form runs)
        ctx.gate_dx = bool(gate_dx)
        R = len(params) // 2
        w1 = [_prep("w1[%d]" % r, params[r]) for r in range(R)]
        b1 = [_prep("b1[%d]" % r, params[R + r]) for r in range(R)]
        x, h2 = _prep("x", x), _prep("h2", h2)
        lead = x.shape[:-1]
        B = x.shape[0]
        L = x.shape[-1]
        N = 1
        for s in lead[1:]:
            N *= s
        H = w1[0].shape[0]
        if h2.shape != (B, R, H):
            raise ValueError("h2 must be [B,R,H] = %s, got %s" % ((B, R, H), tuple(h2.shape)))
        for r in range(R):
            if w1[r].shape != (H, L) or b1[r].shape != (H,):
                raise ValueError("rank %d: weight %s / bias %s do not match (H=%d, L=%d)"
                                 % (r, tuple(w1[r].shape), tuple(b1[r].shape), H, L))
        need_bwd = any(ctx.needs_input_grad)
        out = torch.empty(*lead, H, device=x.device, dtype=torch.float32)
        L_ = _lib.lib()
        # rank-folded form (csrc/bilinear_folded.hip): one contraction per sample against sum_r h2_r (.) W1_r, nothing
        # saved for backward but the inputs; the R-GEMM tile-engine form serves the shapes it does not cover
        folded = (N > 1 and _K4_FORM != "engine" and (_K4_FORM == "folded" or B >= _K4_FOLD_MIN_BATCH)
                  and bool(L_.vqa_lowrank_bilinear_fusion_folded_supported(B, N, L, H, R)))
        ctx.folded = folded
        if folded:
            _launch("lowrank_bilinear_fusion_fwd", (B, N, L, H, R, need_bwd), L_.vqa_lowrank_bilinear_fusion_folded_fwd,
                    _p(x), L, _ptr_array(w1), _ptr_array(b1), _p(h2), _p(out), B, N, L, H, R)
            if need_bwd:
                ctx.save_for_backward(x, h2, *w1, *b1)
        else:
            h1 = torch.empty(B * N, R, H, device=x.device, dtype=torch.float32) if need_bwd else None
            _launch("lowrank_bilinear_fusion_fwd", (B, N, L, H, R, need_bwd), L_.vqa_lowrank_bilinear_fusion_fwd,
                    _p(x), L, _ptr_array(w1), _ptr_array(b1), _p(h2), _p(out), _p(h1), B, N, L, H, R)
            if need_bwd:
                ctx.save_for_backward(x, h2, h1, *w1)
                ctx.b1 = b1
        ctx.dims = (B, N, L, H, R)
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, L, H, R = ctx.dims
        g = _prep("grad_out", g)
        L_ = _lib.lib()
        if ctx.folded:
            x, h2 = ctx.saved_tensors[:2]
            w1, b1 = list(ctx.saved_tensors[2:2 + R]), list(ctx.saved_tensors[2 + R:])
        else:
            x, h2, h1 = ctx.saved_tensors[:3]
            w1 = list(ctx.saved_tensors[3:])
        dev = x.device
        d_x = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        d_h2 = torch.empty_like(h2)
        d_w1 = [_grad_like(w) for w in w1]
        d_b1 = [_grad_like(b) for b in (b1 if ctx.folded else ctx.b1)]
        if ctx.folded:
            ws_bytes = L_.vqa_lowrank_bilinear_fusion_folded_bwd_workspace_bytes(B, N, L, H, R)
            ws = torch.empty((ws_bytes + 3) // 4, device=dev, dtype=torch.float32)
            _launch("lowrank_bilinear_fusion_bwd", (B, N, L, H, R, d_x is not None), L_.vqa_lowrank_bilinear_fusion_folded_bwd_gated,
                    _p(x), L, _ptr_array(w1), _ptr_array(b1), _p(h2), _p(g), _p(d_x), _ptr_array(d_w1), _ptr_array(d_b1),
                    _p(d_h2), _p(ws), ws_bytes, B, N, L, H, R, int(ctx.gate_dx and d_x is not None))
            return (d_x, d_h2, None, *d_w1, *d_b1)
        ws_bytes = L_.vqa_lowrank_bilinear_fusion_bwd_workspace_bytes(B, N, L, H, R)
        ws = torch.empty((ws_bytes + 3) // 4, device=dev, dtype=torch.float32)
        _launch("lowrank_bilinear_fusion_bwd", (B, N, L, H, R, d_x is not None), L_.vqa_lowrank_bilinear_fusion_bwd,
                _p(x), L, _ptr_array(w1), _p(h2), _p(h1), _p(g), _p(d_x), _ptr_array(d_w1), _ptr_array(d_b1), _p(d_h2),
                _p(ws), ws_bytes, B, N, L, H, R)
        if ctx.gate_dx and d_x is not None:      # (the R-GEMM form serves small batches: the gate is a torch op there)
            d_x = torch.ops.aten.threshold_backward(d_x, x, 0)
        return (d_x, d_h2, None, *d_w1, *d_b1)


def widen_bf16(x):
    """bf16 GPU tensor -> fp32 (exact), one HIP pass: the feed's bf16 transport of the region features into the fp32 path."""
    x = _prep("x", x, (torch.bfloat16,))
    out = torch.empty(x.shape, device=x.device, dtype=torch.float32)
    _launch("widen_bf16", (x.numel(),), _lib.lib().vqa_widen_bf16, _p(x), _p(out), x.numel())
    return out


def pack_bf16(src, dst, batch_stride, row_stride, col_stride, zero_fill=True, offset=0):
    """fp32 [batch, rows, cols] (or [rows, cols]) -> bf16 scattered into ``dst`` (from element ``offset`` on) with the
    given element strides; with zero_fill that part of ``dst`` is zeroed first (the pads of the padded / transposed
    weight shadows)."""
    src = _prep("src", src)
    if src.dim() == 2:
        src = src.unsqueeze(0)
    if src.dim() != 3 or dst.dtype != torch.bfloat16 or not dst.is_cuda or not dst.is_contiguous():
        raise ValueError("pack_bf16: src must be [batch,rows,cols] fp32, dst a contiguous CUDA bf16 tensor")
    b, r, c = src.shape
    _launch("pack_bf16", (b, r, c), _lib.lib().vqa_pack_bf16, _p(src), b, r, c,
            ctypes.c_void_p(dst.data_ptr() + 2 * int(offset)), int(batch_stride), int(row_stride), int(col_stride),
            dst.numel() - int(offset), int(bool(zero_fill)))
    return dst


_shadows = {}
_shadow_lock = threading.Lock()


def _shadow(master, shape, tag, dtype=torch.bfloat16):
    """Persistent zero-initialised buffer for the padded (or transposed) bf16 shadow -- or padded fp32 copy -- of an fp32
    master parameter.  The pads never change, so they are zeroed ONCE; every step only repacks the interior -- instead of
    an allocation, a zero-fill kernel and a pack kernel per weight and step.  Keyed by the master's address: parameters
    keep theirs (views of the trainer's flat buffer), and a stale entry only ever has its interior overwritten."""
    key = (master.data_ptr(), tuple(shape), tag, master.device.index, dtype)
    with _shadow_lock:      # (nn.DataParallel-style callers drive one replica per thread)
        buf = _shadows.get(key)
        if buf is None:
            # never evicted: a captured hipGraph holds the ADDRESS of a shadow and no Python reference to it, so freeing
            # one would let replays write packed weights into memory the allocator has handed to someone else.  The set
            # is bounded by (weights x layouts) of the models alive in the process.
            buf = _shadows[key] = torch.zeros(shape, device=master.device, dtype=dtype)
    return buf


PACK_BLOCK = 1024      # VQA_PACK_BLOCK (include/vqa_mi355x.h)


class ShadowPlan:
    """Every bf16 / padded-fp32 shadow a model's mixed-precision forward + backward reads, packed from the fp32 masters by
    ONE kernel per step (vqa_pack_many) instead of one pack kernel -- plus a bias copy -- per weight and layout (11 + 11
    launches per CoR2 step).  The model builds the plan once, calls ``pack()`` at the top of every forward (the masters
    change only in the optimizer step) and hands each layer its shadows explicitly."""

    def __init__(self):
        self.jobs = []          # (src parameter, dst buffer, element offset into dst, row stride, col stride)
        self._table = None
        self._key = None
        self._total = 0

    def add(self, src, dst, row_stride, col_stride=1, offset=0):
        if src.dim() == 1:
            rows, cols = 1, src.numel()
        else:
            rows, cols = src.shape[0], src.numel() // src.shape[0]
        last = offset + (rows - 1) * row_stride + (cols - 1) * col_stride
        if dst.dtype not in (torch.bfloat16, torch.float32) or not dst.is_contiguous() or last >= dst.numel():
            raise ValueError("ShadowPlan.add: destination too small or of the wrong kind")
        self.jobs.append((src, dst, int(offset), int(row_stride), int(col_stride)))
        self._key = None
        return dst

    def pack(self):
        if not self.jobs:
            return
        key = tuple((s.data_ptr(), d.data_ptr()) for s, d, *_ in self.jobs)
        if key != self._key:        # first use, or the parameters moved (trainer.FlatState re-homes them once)
            rows, first = [], 0
            for src, dst, off, rs, cs in self.jobs:
                if not (src.is_cuda and src.dtype == torch.float32 and src.is_contiguous()):
                    raise _lib.VqaLibraryError("ShadowPlan: masters must be contiguous fp32 GPU tensors")
                r, c = (1, src.numel()) if src.dim() == 1 else (src.shape[0], src.numel() // src.shape[0])
                rows.append([src.data_ptr(), dst.data_ptr() + off * dst.element_size(), r, c, rs, cs,
                             0 if dst.dtype == torch.bfloat16 else 1, first])
                # numbers a job takes: its elements -- a transposed one: its 32 x 32 tiles x 1024 (include/vqa_mi355x.h)
                total = first + (-(-r // 32) * -(-c // 32) * PACK_BLOCK if (rs == 1 and cs != 1) else r * c)
                first = -(-total // PACK_BLOCK) * PACK_BLOCK      # every job starts on a multiple of VQA_PACK_BLOCK
            dev = self.jobs[0][0].device
            self._table = torch.tensor(rows, dtype=torch.int64).to(dev)
            self._total = total
            self._key = key
        _launch("pack_many", (len(self.jobs), self._total), _lib.lib().vqa_pack_many, _p(self._table), len(self.jobs),
                self._total)


class PackedWeightBf16(torch.autograd.Function):
    """w fp32 [rows, cols] -> bf16 [rows_p, cols_p] zero-padded shadow (the operand layout of the bf16 GEMMs);
    backward crops the gradient back to the master shape in fp32."""

    @staticmethod
    def forward(ctx, w, rows_p, cols_p):
        w = _prep("w", w)
        rows, cols = w.shape
        if rows_p < rows or cols_p < cols:
            raise ValueError("PackedWeightBf16: padded shape (%d,%d) smaller than %s" % (rows_p, cols_p, tuple(w.shape)))
        ctx.shape = (rows, cols)
        return pack_bf16(w, _shadow(w, (rows_p, cols_p), "p"), 0, cols_p, 1, zero_fill=False)

    @staticmethod
    def backward(ctx, g):
        rows, cols = ctx.shape
        return g[:rows, :cols].float(), None, None


def gemm_bf16_nt(a, b, bias=None, act=None, out=None, gate=None, p_drop=0.0, seed=0):
    """act(drop_p(a)[M,K] @ b[N,K]^T + bias) -> bf16 [M,N] on the bf16 MFMA tile engine (K % 64 == 0).  p_drop = 0.5: the
    input dropout is applied to `a` inside the kernel (the counter-hash mask linear_dropout_mask exports).  gate [M,>=N]
    bf16: the result is zeroed where gate <= 0 (a relu gradient applied in the store)."""
    a, b = _prep("a", a, (torch.bfloat16,)), _prep("b", b, (torch.bfloat16,))
    bias = _prep("bias", bias) if bias is not None else None
    M, K = a.shape
    N = b.shape[0]
    if b.shape != (N, K) or (bias is not None and bias.shape != (N,)):
        raise ValueError("gemm_bf16_nt: b must be [N,K] = [%d,%d], bias [N]" % (N, K))
    code = {None: 0, "": 0, "relu": 1}.get(act)
    if code is None:
        raise ValueError("gemm_bf16_nt: act must be None or 'relu', got %r" % (act,))
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.bfloat16)
    if gate is None and not p_drop:
        _launch("gemm_bf16_nt", (M, N, K, code), _lib.lib().vqa_gemm_bf16_nt, _p(a), K, _p(b), K, _p(bias), _p(out), N,
                M, N, K, code)
        return out
    if gate is not None:
        gate = _prep("gate", gate, (torch.bfloat16,))
        if gate.dim() != 2 or gate.shape[0] != M or gate.shape[1] < N:
            raise ValueError("gemm_bf16_nt: gate must be [M, >= N] = [%d, >= %d]" % (M, N))
    sv, sp = _seed_args(seed)
    _launch("gemm_bf16_nt", (M, N, K, code), _lib.lib().vqa_gemm_bf16_nt_ex, _p(a), K, _p(b), K, _p(bias), _p(out), N,
            M, N, K, code, _p(gate), gate.shape[1] if gate is not None else 0, float(p_drop), sv, sp)
    return out


def gemm_bf16_tn(a, b, outs=None, out_rows=None, out_cols=None, p_drop=0.0, seed=0):
    """a[K,N1]^T @ drop_p(b)[K,N2] -> fp32 (weight-gradient contraction over the batch rows; fixed-order split-K).
    outs None: one dense [N1,N2] tensor is returned.  outs = G fp32 tensors [out_rows, out_cols] (dense): rows
    [g*N1/G, g*N1/G + out_rows) x columns [0, out_cols) of the product go to outs[g] -- the master-shaped gradients of G
    padded layers, written in place (no slice / copy kernels).  p_drop = 0.5: `b` is dropped out inside the kernel."""
    a, b = _prep("a", a, (torch.bfloat16,)), _prep("b", b, (torch.bfloat16,))
    K, N1 = a.shape
    N2 = b.shape[1]
    if b.shape[0] != K:
        raise ValueError("gemm_bf16_tn: a [K,N1] and b [K,N2] must share K")
    L_ = _lib.lib()
    ws_bytes = L_.vqa_gemm_bf16_tn_workspace_bytes(K, N1, N2)
    ws = torch.empty((ws_bytes + 3) // 4, device=a.device, dtype=torch.float32)
    if outs is None and not p_drop:
        c = torch.empty(N1, N2, device=a.device, dtype=torch.float32)
        _launch("gemm_bf16_tn", (K, N1, N2), L_.vqa_gemm_bf16_tn, _p(a), N1, _p(b), N2, _p(c), _p(ws), ws_bytes, K, N1, N2)
        return c
    single = outs is None
    if single:
        outs, out_rows, out_cols = [torch.empty(N1, N2, device=a.device, dtype=torch.float32)], N1, N2
    G = len(outs)
    if N1 % G or any(o.dtype != torch.float32 or not o.is_contiguous() or tuple(o.shape) != (out_rows, out_cols) for o in outs):
        raise ValueError("gemm_bf16_tn: outs must be %d dense fp32 [%s,%s] tensors" % (G, out_rows, out_cols))
    sv, sp = _seed_args(seed)
    _launch("gemm_bf16_tn", (K, N1, N2), L_.vqa_gemm_bf16_tn_ex, _p(a), N1, _p(b), N2, _ptr_array(outs), G, N1 // G,
            int(out_rows), int(out_cols), int(out_cols), _p(ws), ws_bytes, K, N1, N2, float(p_drop), sv, sp)
    return outs[0] if single else outs


class LinearBf16(torch.autograd.Function):
    """y = act(drop_p(x) W^T + b) on the bf16 MFMA engine: x bf16 [..., Kp] (Kp >= in_features, zero-padded), master W fp32
    [out, in] / b fp32 [out]; returns bf16 [..., Np] with Np = out padded to 64 (pad columns exactly zero).
    The nn.Linear / 1x1 nn.Conv1d contraction of MyLinear / MyConv1d (config/CoR2.py:56-122) in mixed precision, with the
    layer's input dropout (config/CoR2.py:72-75) inside the kernels at p = 0.5: the forward masks x while it stages it,
    the weight gradient regenerates the mask the same way -- no dropped copy of x, no mask tensor.
    packed = (wp [Np,Kp] bf16, bp [Np] fp32 or None, wpt [Kp,Np] bf16 or None): the shadows of W / b / W^T when a
    ShadowPlan of the caller has packed them already this step; None: packed here.
    pregated: whoever consumes y returns its gradient already multiplied by relu'(y) (K4's data gradient with gate_dx)."""

    @staticmethod
    def forward(ctx, x, w, bias, act, p_drop, seed, pregated, packed):
        x, w = _prep("x", x, (torch.bfloat16,)), _prep("w", w)
        Kp = x.shape[-1]
        out_f, in_f = w.shape
        if Kp % BF16_PAD or Kp < in_f:
            raise ValueError("linear_bf16: x's last dim (%d) must be in_features=%d zero-padded to a multiple of %d"
                             % (Kp, in_f, BF16_PAD))
        Np = pad_to(out_f)
        if packed is not None:
            wp, bp, wpt = packed
        else:
            wp = pack_bf16(w, _shadow(w, (Np, Kp), "nk"), 0, Kp, 1, zero_fill=False)
            bp, wpt = None, None
            if bias is not None:
                bp = _shadow(bias, (Np,), "b", torch.float32)
                bp[:out_f].copy_(_prep("bias", bias))
        x2 = x.reshape(-1, Kp)
        y = gemm_bf16_nt(x2, wp, bp, act, p_drop=p_drop, seed=seed)
        ctx.save_for_backward(x2, w, y)
        ctx.bias = bias
        ctx.wpt = wpt
        ctx.cfg = (act, Np, float(p_drop), seed, bool(pregated) and act == "relu")
        return y.view(*x.shape[:-1], Np)

    @staticmethod
    def backward(ctx, gy):
        x2, w, y = ctx.saved_tensors
        act, Np, p_drop, seed, pregated = ctx.cfg
        out_f, in_f = w.shape
        Kp = x2.shape[1]
        gz = gy.reshape(-1, Np).to(torch.bfloat16)
        if act == "relu" and not pregated:
            gz = gz * (y > 0)
        gz = gz.contiguous()
        d_x = None
        if ctx.needs_input_grad[0]:
            if p_drop:
                raise _lib.VqaLibraryError("linear_bf16: the data gradient of a layer with in-kernel input dropout is not "
                                           "implemented (no model on the path needs it: compress_v reads the model input)")
            wpt = ctx.wpt
            if wpt is None:
                wpt = pack_bf16(w, _shadow(w, (Kp, Np), "kn"), 0, 1, Np, zero_fill=False)   # [Kp, Np] = W^T
            d_x = gemm_bf16_nt(gz, wpt).view(*gy.shape[:-1], Kp)
        d_w = _grad_like(w)
        d_b = _grad_like(ctx.bias) if ctx.bias is not None else None
        gemm_bf16_tn(gz, x2, outs=[d_w], out_rows=out_f, out_cols=in_f, p_drop=p_drop, seed=seed)
        if d_b is not None:
            column_sum(gz, out=d_b, cols=out_f)
        return d_x, d_w, d_b, None, None, None, None, None


def linear_bf16(x, w, bias=None, act=None, p_drop=0.0, seed=0, pregated=False, packed=None):
    if act not in (None, "", "relu"):
        raise ValueError("linear_bf16: act must be None or 'relu', got %r" % (act,))
    if p_drop and float(p_drop) != 0.5:
        raise ValueError("linear_bf16: in-kernel dropout exists at p = 0.5 only; drop the input beforehand for p=%r" % (p_drop,))
    return LinearBf16.apply(x, w, bias, act or None, float(p_drop or 0.0), seed, pregated, packed)


# "rgemm" (default: R products on the LDS-tile engine, csrc/bf16_path.hip) | "fold" (csrc/bilinear_fold_bf16.hip: one product per
# sample with the rank-folded weight -- half the MFMA work, but a per-sample weight has no reuse across samples and the kernels are
# L2-bound at the config's size: 139 us per fusion against 108, docs/measured_negatives_r05.md; kept as a tested alternative)
K4_BF16_FORM = os.environ.get("VQA_K4_BF16_FORM", "rgemm")


class LowRankBilinearFusionBf16(torch.autograd.Function):
    """K4 on the bf16 MFMA engine.  x bf16 [B,(N,)Lp] with Lp >= L zero-padded to a multiple of 64; h2 fp32 [B,R,H];
    master weights / biases fp32 ([H,L] / [H] per rank).  Returns bf16 [B,(N,)Hp], Hp = H padded to 256 (pad = 0).
    packed = (w1p [R,Hp,Lp] bf16, b1p [R,Hp] fp32, w1t [Lp,R*Hp] bf16): shadows a ShadowPlan packed this step, or None.
    gate_dx: x is the relu output of the layer in front and d_x comes back multiplied by (x > 0)."""

    @staticmethod
    def forward(ctx, x, h2, gate_dx, packed, *params):
        R = len(params) // 2
        w1 = [_prep("w1[%d]" % r, params[r]) for r in range(R)]
        b1 = [_prep("b1[%d]" % r, params[R + r]) for r in range(R)]
        x, h2 = _prep("x", x, (torch.bfloat16,)), _prep("h2", h2)
        lead = x.shape[:-1]
        B, Lp = x.shape[0], x.shape[-1]
        N = 1
        for s in lead[1:]:
            N *= s
        H, L = w1[0].shape
        Hp = pad_to(H, 256)
        if Lp % BF16_PAD or Lp < L:
            raise ValueError("bf16 fusion: x's last dim (%d) must be L=%d zero-padded to a multiple of %d" % (Lp, L, BF16_PAD))
        if h2.shape != (B, R, H):
            raise ValueError("h2 must be [B,R,H] = %s, got %s" % ((B, R, H), tuple(h2.shape)))
        for r in range(R):
            if w1[r].shape != (H, L) or b1[r].shape != (H,):
                raise ValueError("rank %d: weight %s / bias %s do not match (H=%d, L=%d)"
                                 % (r, tuple(w1[r].shape), tuple(b1[r].shape), H, L))
        dev = x.device
        if packed is not None:
            w1p, b1p, w1t = packed
        else:
            w1p = _shadow(w1[0], (R, Hp, Lp), "k4")
            b1p = _shadow(b1[0], (R, Hp), "k4b", torch.float32)
            for r in range(R):
                pack_bf16(w1[r], w1p, 0, Lp, 1, zero_fill=False, offset=r * Hp * Lp)
                b1p[r, :H].copy_(b1[r])
            w1t = None
        need_bwd = any(ctx.needs_input_grad)
        out = torch.empty(*lead, Hp, device=dev, dtype=torch.bfloat16)
        # VQA_K4_BF16_FORM=fold: rank-folded (csrc/bilinear_fold_bf16.hip: one product per sample with the weight
        # sum_r diag(h2_r[b]) W1_r built in registers; no h1 saved) where the library offers it -- R = 2, N <= 128
        fold = K4_BF16_FORM == "fold" and _lib.lib().vqa_bilinear_fold_bf16_supported(B, N, Lp, Hp, R) == 1
        h1 = None
        if fold:
            _launch("lowrank_bilinear_fusion_fwd_bf16", (B, N, Lp, Hp, R, need_bwd), _lib.lib().vqa_bilinear_fold_fwd_bf16,
                    _p(x), _p(w1p), _p(b1p), _p(h2), _p(out), B, N, Lp, Hp, R, H)
        else:
            h1 = torch.empty(B * N, R, Hp, device=dev, dtype=torch.bfloat16) if need_bwd else None
            _launch("lowrank_bilinear_fusion_fwd_bf16", (B, N, Lp, Hp, R, need_bwd),
                    _lib.lib().vqa_lowrank_bilinear_fusion_fwd_bf16, _p(x), _p(w1p), _p(b1p), _p(h2), _p(out), _p(h1),
                    B, N, Lp, Hp, R, H)
        if need_bwd:
            ctx.save_for_backward(x, h2, h1, *w1)
            ctx.b1 = b1
            ctx.w1t = w1t
            ctx.fold = (w1p, b1p) if fold else None
        ctx.dims = (B, N, L, Lp, H, Hp, R, bool(gate_dx))
        return out

    @staticmethod
    def backward(ctx, g):
        B, N, L, Lp, H, Hp, R, gate_dx = ctx.dims
        x, h2, h1 = ctx.saved_tensors[:3]
        w1 = ctx.saved_tensors[3:]
        g = _prep("grad_out", g.to(torch.bfloat16), (torch.bfloat16,))
        dev = x.device
        w1t = None
        d_x = None
        if ctx.needs_input_grad[0]:
            w1t = ctx.w1t
            if w1t is None:
                w1t = _shadow(w1[0], (Lp, R * Hp), "k4t")   # w1t[l, r*Hp+h] = w1[r][h,l]; pads zeroed once
                for r in range(R):   # rank r fills columns r*Hp .. r*Hp+H of every row
                    pack_bf16(w1[r], w1t, 0, 1, R * Hp, zero_fill=False, offset=r * Hp)
            d_x = torch.empty_like(x)
        d_w1 = [_grad_like(w) for w in w1]
        d_b1 = [_grad_like(b) for b in ctx.b1]
        d_h2 = torch.empty(B, R, H, device=dev, dtype=torch.float32)
        L_ = _lib.lib()
        if ctx.fold is not None:
            w1p, b1p = ctx.fold
            ws_bytes = L_.vqa_bilinear_fold_bwd_bf16_workspace_bytes(B, N, Lp, Hp, R)
            ws = torch.empty((ws_bytes + 3) // 4, device=dev, dtype=torch.float32)
            _launch("lowrank_bilinear_fusion_bwd_bf16", (B, N, Lp, Hp, R, d_x is not None), L_.vqa_bilinear_fold_bwd_bf16,
                    _p(x), _p(w1p), _p(w1t), _p(b1p), _p(h2), _p(g), _p(d_x), _ptr_array(d_w1), _ptr_array(d_b1), _p(d_h2),
                    _p(ws), ws_bytes, B, N, Lp, Hp, R, H, L, int(gate_dx and d_x is not None))
            return (d_x, d_h2, None, None, *d_w1, *d_b1)
        ws_bytes = L_.vqa_lowrank_bilinear_fusion_bwd_bf16_workspace_bytes(B, N, Lp, Hp, R)
        ws = torch.empty((ws_bytes + 3) // 4, device=dev, dtype=torch.float32)
        args = (_p(x), _p(w1t), _p(h2), _p(h1), _p(g), _p(d_x), _ptr_array(d_w1), _ptr_array(d_b1), _p(d_h2), _p(ws), ws_bytes,
                B, N, Lp, Hp, R, H, L, int(gate_dx and d_x is not None))
        _launch("lowrank_bilinear_fusion_bwd_bf16", (B, N, Lp, Hp, R, d_x is not None),
                L_.vqa_lowrank_bilinear_fusion_bwd_bf16, *args, 3)
        return (d_x, d_h2, None, None, *d_w1, *d_b1)


class ObjectDifferenceAttention(torch.autograd.Function):
    """K2.  logits[b,i,g] = bias[g] + sum_{j,d} w[g,j*L+d] * keep * (vl[b,i,d]-vl[b,j,d]) * ql[b,d].
    Replaces config/ODA.py:216-222 + the dropout and 1x1 conv of config/ODA.py:149.
    gate_dvl: vl is the relu output of the layer in front and nothing else reads it; d_vl comes back multiplied by (vl > 0)."""

    @staticmethod
    def forward(ctx, vl, ql, w, bias, p_drop, seed, gate_dvl=False):
        vl, ql, w, bias = _prep("vl", vl), _prep("ql", ql), _prep("w", w), _prep("bias", bias)
        B, N, L = vl.shape
        G = bias.shape[0]
        if ql.shape != (B, L) or w.numel() != G * N * L:
            raise ValueError("object_difference_attention: ql must be [B,L], w must hold G*N*L = %d values (got %d)"
                             % (G * N * L, w.numel()))
        logits = torch.empty(B, N, G, device=vl.device, dtype=torch.float32)
        sv, sp = _seed_args(seed)
        _launch("object_difference_attention_fwd", (B, N, L, G, float(p_drop) > 0),
                _lib.lib().vqa_object_difference_attention_fwd, _p(vl), _p(ql), _p(w), _p(bias), _p(logits),
                float(p_drop), sv, sp, B, N, L, G)
        ctx.save_for_backward(vl, ql, w)
        ctx.bias = bias
        ctx.cfg = (float(p_drop), seed, G, bool(gate_dvl))
        return logits

    @staticmethod
    def backward(ctx, d_logits):
        vl, ql, w = ctx.saved_tensors
        p_drop, seed, G, gate_dvl = ctx.cfg
        B, N, L = vl.shape
        d_logits = _prep("grad_logits", d_logits)
        d_vl, d_ql, d_w = torch.empty_like(vl), torch.empty_like(ql), _grad_like(w)
        d_bias = _grad_like(ctx.bias)
        L_ = _lib.lib()
        ws_bytes = L_.vqa_object_difference_attention_bwd_workspace_bytes(B, N, L, G)
        ws = torch.empty((ws_bytes + 3) // 4, device=vl.device, dtype=torch.float32)
        sv, sp = _seed_args(seed)
        _launch("object_difference_attention_bwd", (B, N, L, G, p_drop > 0), L_.vqa_object_difference_attention_bwd,
                _p(vl), _p(ql), _p(w), _p(d_logits), _p(d_vl), _p(d_ql), _p(d_w), _p(d_bias), _p(ws), ws_bytes,
                p_drop, sv, sp, B, N, L, G, int(gate_dvl))
        return d_vl, d_ql, d_w, d_bias, None, None, None


def f32_products():
    """How the tall fp32 projections form their products: "split" (the default since round 5: exact three-way bf16 splits of both
    operands, six partial products on v_mfma_f32_16x16x32_bf16, fp32 accumulation -- csrc/gemm_f32_split.hpp; an fp32 GEMM in
    error and in domain, tests/test_gpu_split.py) or "mfma" (v_mfma_f32_16x16x4_f32).  Host-side selection, read per call from the
    environment variable VQA_F32_PRODUCTS -- not a library option (vqa_set_option has no say in it)."""
    mode = os.environ.get("VQA_F32_PRODUCTS", "split")
    if mode not in ("split", "mfma"):
        raise ValueError("VQA_F32_PRODUCTS must be 'split' or 'mfma', got %r" % (mode,))
    return mode


def split_products(M, K, N, ldx, p_drop, weight_gradient=False, tensors=()):
    """True when this tall projection runs on the split engine (csrc/gemm_f32_split.hpp: fp32 products from exact three-way
    bf16 splits on the bf16 matrix pipe, fp32 accumulation): f32_products() says "split" (the default) and the library takes the
    shape (vqa_linear_split_supported); everything else runs on the fp32 MFMA engine.
    `tensors`: the operands the engine reads with 16-byte loads (a view at an odd offset stays on the fp32 MFMA engine)."""
    if f32_products() != "split":
        return False
    if weight_gradient and K % 128 != 0:
        return False
    if any(t is not None and t.data_ptr() % 16 != 0 for t in tensors):
        return False
    return _lib.lib().vqa_linear_split_supported(M, K, N, ldx, float(p_drop)) == 1


def _linear_fwd(x, w, bias, y, M, K, N, act, p_drop, seed):
    """y = act(dropout(x) W^T + b), on the engine split_products() selects."""
    L_ = _lib.lib()
    sv, sp = _seed_args(seed)
    if split_products(M, K, N, K, p_drop, tensors=(x, w)):
        ws_bytes = L_.vqa_linear_act_fwd_split_workspace_bytes(K, N)
        ws = torch.empty((ws_bytes + 3) // 4, device=x.device, dtype=torch.float32)
        _launch("linear_act_fwd_split", (M, K, N, float(p_drop) > 0), L_.vqa_linear_act_fwd_split, _p(x), K, _p(w), _p(bias),
                _p(y), _p(ws), ws_bytes, M, K, N, int(act), float(p_drop), sv, sp)
    else:
        _launch("linear_act_fwd", (M, K, N, float(p_drop) > 0), L_.vqa_linear_act_fwd,
                _p(x), K, _p(w), _p(bias), _p(y), M, K, N, int(act), float(p_drop), sv, sp)


def _linear_dw(x, w, y, gy, d_w, d_b, M, K, N, act, p_drop, seed, gz_out=None):
    """d_w, d_b of y = act(dropout(x) W^T + b) (no data gradient), on the engine split_products() selects.  gz_out (split engine,
    act = relu only): receives gy * (y > 0), which the pass that packs the gradient for the GEMM has in hand."""
    L_ = _lib.lib()
    sv, sp = _seed_args(seed)
    if split_products(M, K, N, K, p_drop, weight_gradient=True, tensors=(x, d_w, gy, y if act == 1 else None)):
        ws_bytes = L_.vqa_linear_act_dw_split_workspace_bytes(M, K, N)
        ws = torch.empty((ws_bytes + 3) // 4, device=x.device, dtype=torch.float32)
        _launch("linear_act_dw_split", (M, K, N, p_drop > 0, act), L_.vqa_linear_act_dw_split, _p(x), K,
                _p(y) if act == 1 else None, _p(gy), _p(d_w), _p(d_b), _p(gz_out) if gz_out is not None else None, _p(ws), ws_bytes,
                M, K, N, act, p_drop, sv, sp)
    else:
        assert gz_out is None
        ws_bytes = L_.vqa_linear_act_bwd_workspace_bytes(M, K, N)
        ws = torch.empty((ws_bytes + 3) // 4, device=x.device, dtype=torch.float32)
        _launch("linear_act_bwd", (M, K, N, p_drop > 0, False), L_.vqa_linear_act_bwd,
                _p(x), K, _p(w), _p(y), _p(gy), None, _p(d_w), _p(d_b), _p(ws), ws_bytes, M, K, N, act, p_drop, sv, sp)


class LinearAct(torch.autograd.Function):
    """K5.  y = act(dropout_p(x) W^T + b) on the fp32 MFMA tile engine; act in {None, 'relu'}.
    Replaces MyConv1d(k=1).forward (config/CoR2.py:72-88) / MyLinear.forward (config/CoR2.py:106-121)."""

    @staticmethod
    def forward(ctx, x, w, bias, act, p_drop, seed, pregated=False):
        # pregated: whoever consumes y returns its gradient already multiplied by relu'(y) (ops.lowrank_bilinear_fusion with
        # gate_dx) -- backward then runs the ungated kernels (no y operand in the weight gradient, no masking pass)
        ctx.pregated = bool(pregated) and int(act) == 1
        x, w = _prep("x", x), _prep("w", w)
        bias = _prep("bias", bias) if bias is not None else None
        K = x.shape[-1]
        M = x.numel() // K
        N = w.shape[0]
        if w.shape != (N, K) or (bias is not None and bias.shape != (N,)):
            raise ValueError("linear_act: weight must be [N,K] = [%d,%d], bias [N]" % (N, K))
        y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
        _linear_fwd(x, w, bias, y, M, K, N, int(act), float(p_drop), seed)
        ctx.save_for_backward(x, w, y)
        ctx.bias = bias
        ctx.cfg = (M, K, N, int(act), float(p_drop), seed, bias is not None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        M, K, N, act, p_drop, seed, has_bias = ctx.cfg
        if ctx.pregated:
            act = 0
        gy = _prep("grad_y", gy)
        d_x = None
        if ctx.needs_input_grad[0] and p_drop == 0 and M >= 4096 and LinearAct.library_dgrad:
            # tall data gradient without a dropout mask (compress_v2 behind the relation kernel): the gated gradient is
            # materialised once, the data gradient is a plain library GEMM on it (at [18432,310] x [310,2048] it and the
            # register-tile engine tie: tools/rt_probe.hip), and the weight gradient below runs ungated on the same tensor
            if act == 1:
                gy = torch.ops.aten.threshold_backward(gy, y, 0)
                act = 0
            with timed("library_gemm", (M, K, N)):
                d_x = (gy.reshape(M, N) @ w).view(x.shape)
        elif ctx.needs_input_grad[0]:
            d_x = torch.empty_like(x)
        in_kernel_dx = d_x is not None and not (p_drop == 0 and M >= 4096 and LinearAct.library_dgrad)
        d_w = _grad_like(w)
        d_b = _grad_like(ctx.bias) if has_bias else None
        if not in_kernel_dx:
            _linear_dw(x, w, y, gy, d_w, d_b, M, K, N, act, p_drop, seed)
            return d_x, d_w, d_b, None, None, None, None
        L_ = _lib.lib()
        ws_bytes = L_.vqa_linear_act_bwd_workspace_bytes(M, K, N)
        ws = torch.empty((ws_bytes + 3) // 4, device=x.device, dtype=torch.float32)
        sv, sp = _seed_args(seed)
        _launch("linear_act_bwd", (M, K, N, p_drop > 0, in_kernel_dx), L_.vqa_linear_act_bwd,
                _p(x), K, _p(w), _p(y), _p(gy), _p(d_x), _p(d_w), _p(d_b), _p(ws), ws_bytes,
                M, K, N, act, p_drop, sv, sp)
        return d_x, d_w, d_b, None, None, None, None

    library_dgrad = os.environ.get("VQA_LIBRARY_DGRAD", "1") == "1"


def linear_act(x, w, bias=None, act=None, p_drop=0.0, seed=0, pregated=False):
    code = {None: 0, "": 0, "relu": 1}.get(act)
    if code is None:
        raise ValueError("linear_act: act must be None or 'relu', got %r" % (act,))
    return LinearAct.apply(x, w, bias, code, p_drop, seed, pregated)


class RelationProjection(torch.autograd.Function):
    """K1 -> K5: y = relu(keep * (t + c2 * v) W^T + b) -- the closed-form relation step (RelationApply) feeding the second
    region projection (LinearAct) as ONE autograd node, so that backward can skip the projection's data gradient as a
    tensor: d_t and d_c2 come out of the GEMM tile directly (csrc/relation_dgrad.hip; [B,N,D] is neither written nor
    re-read).  v must not need a gradient (it is the model's input); shapes vqa_relation_projection_dgrad_supported()."""

    @staticmethod
    def forward(ctx, v, t, c2, w, bias, p_drop, seed, pregated, pairwise=None):
        # pairwise = (q1, q2, alpha [B,N,G], glimpse): the relation tensor is built by the PAIRWISE kernel -- every (i, j) term
        # alpha_i (v_i q1 + v_j q2) summed, the reference's structure (config/CoR2.py:191-199,216) -- instead of the closed
        # form t + c2 v that equals it for t = q1 sum_i alpha_i v_i, c2 = (sum_i alpha_i) q2; the gradient still flows through
        # (t, c2), whose producers carry it on to q1, q2 and alpha
        v, t, c2, w = _prep("v", v), _prep("t", t), _prep("c2", c2), _prep("w", w)
        bias = _prep("bias", bias) if bias is not None else None
        B, N, D = v.shape
        L = w.shape[0]
        if t.shape != (B, D) or c2.shape != (B, D) or w.shape != (L, D):
            raise ValueError("relation_projection: t, c2 must be [B,D] and w [L,D] for v [B,N,D]")
        L_ = _lib.lib()
        sv, sp = _seed_args(seed)
        M = B * N
        y = torch.empty(B, N, L, device=v.device, dtype=torch.float32)
        # Round 6 (VERDICT r05 missing #2, SURVEY 8f row 1 second half): the closed-form relation step is applied to the
        # projection's A fragments in registers (vqa_relation_linear_fwd_split) -- v2 is neither written nor read back; backward
        # recomputes it the same way inside the weight-gradient kernel.  The materialising path stays for the pairwise forward
        # (relation_mode 0), the fp32 MFMA engine and shapes outside the fused form.
        fused = pairwise is None and relation_linear_fused(v, t, c2, w, p_drop)
        if fused:
            ws_bytes = L_.vqa_linear_act_fwd_split_workspace_bytes(D, L)
            ws = torch.empty((ws_bytes + 3) // 4, device=v.device, dtype=torch.float32)
            _launch("relation_linear_fwd_split", (B, N, D, L, float(p_drop) > 0), L_.vqa_relation_linear_fwd_split, _p(v), _p(t), _p(c2),
                    _p(w), _p(bias), _p(y), _p(ws), ws_bytes, B, N, D, L, 1, float(p_drop), sv, sp)
            ctx.save_for_backward(v, t, c2, w, y)
        else:
            x = torch.empty_like(v)
            if pairwise is not None:
                q1, q2, alpha, glimpse = pairwise
                q1, q2, alpha = _prep("q1", q1.detach()), _prep("q2", q2.detach()), _prep("alpha", alpha.detach())
                a_ptr = ctypes.c_void_p(alpha.data_ptr() + 4 * glimpse)
                _launch("pairwise_relation_reduce_fwd", (B, N, D, 0), L_.vqa_pairwise_relation_reduce_drop_fwd, _p(v), _p(q1), _p(q2),
                        a_ptr, alpha.shape[2], _p(x), float(p_drop), sv, sp, B, N, D)
            else:
                _launch("relation_apply_fwd", (B, N, D, float(p_drop) > 0), L_.vqa_relation_apply_fwd, _p(v), _p(t), _p(c2), _p(x),
                        float(p_drop), sv, sp, B, N, D)
            _linear_fwd(x, w, bias, y, M, D, L, 1, 0.0, 0)
            ctx.save_for_backward(v, x, w, y)
        ctx.fused = fused
        ctx.bias = bias
        ctx.cfg = (float(p_drop), seed, bool(pregated))
        return y

    @staticmethod
    def backward(ctx, gy):
        if ctx.fused:
            v, t, c2, w, y = ctx.saved_tensors
            x = None
        else:
            v, x, w, y = ctx.saved_tensors
        p_drop, seed, pregated = ctx.cfg
        B, N, D = v.shape
        L = w.shape[0]
        M = B * N
        gy = _prep("grad_y", gy)
        L_ = _lib.lib()
        d_w = _grad_like(w)
        d_b = _grad_like(ctx.bias) if ctx.bias is not None else None
        if ctx.fused:
            # the layer input t + c2 v (times its mask) is recomputed from v inside the weight-gradient GEMM; pregated: gy arrives
            # already multiplied by relu'(y) -- the packing pass then neither gates nor writes a copy
            sv, sp = _seed_args(seed)
            gz = gy if pregated else torch.empty_like(gy)
            ws_bytes = L_.vqa_linear_act_dw_split_workspace_bytes(M, D, L)
            ws = torch.empty((ws_bytes + 3) // 4, device=v.device, dtype=torch.float32)
            _launch("relation_linear_dw_split", (M, D, L, p_drop > 0, 0 if pregated else 1), L_.vqa_relation_linear_dw_split, _p(v), _p(t),
                    _p(c2), None if pregated else _p(y), _p(gy), _p(d_w), _p(d_b), None if pregated else _p(gz), _p(ws), ws_bytes,
                    B, N, D, L, 0 if pregated else 1, p_drop, sv, sp)
        elif not pregated and split_products(M, D, L, D, 0.0, weight_gradient=True, tensors=(x, d_w)):
            gz = torch.empty_like(gy)                 # the split engine's packing pass gates the gradient and writes it out as well
            _linear_dw(x, w, y, gy, d_w, d_b, M, D, L, 1, 0.0, 0, gz_out=gz)
        else:
            gz = gy if pregated else torch.ops.aten.threshold_backward(gy, y, 0)     # one gated tensor for both gradients
            _linear_dw(x, w, y, gz, d_w, d_b, M, D, L, 0, 0.0, 0)
        d_t = torch.empty(B, D, device=v.device, dtype=torch.float32)
        d_c2 = torch.empty(B, D, device=v.device, dtype=torch.float32)
        sv, sp = _seed_args(seed)
        if (f32_products() == "split" and L_.vqa_relation_projection_dgrad_split_supported(B, N, D, L) == 1
                and all(t.data_ptr() % 16 == 0 for t in (v, gz, w))):
            ws_bytes = L_.vqa_relation_projection_dgrad_split_workspace_bytes(D, L)
            ws = torch.empty((ws_bytes + 3) // 4, device=v.device, dtype=torch.float32)
            _launch("relation_projection_dgrad_split", (B, N, D, L, p_drop > 0), L_.vqa_relation_projection_dgrad_split, _p(gz), _p(w),
                    _p(v), _p(d_t), _p(d_c2), _p(ws), ws_bytes, p_drop, sv, sp, B, N, D, L)
        else:
            _launch("relation_projection_dgrad", (B, N, D, L, p_drop > 0), L_.vqa_relation_projection_dgrad, _p(gz), _p(w), _p(v),
                    _p(d_t), _p(d_c2), p_drop, sv, sp, B, N, D, L)
        return None, d_t, d_c2, d_w, d_b, None, None, None, None


def relation_linear_fused(v, t, c2, w, p_drop):
    """Does RelationProjection run K1 -> K5 as one kernel forward and one weight-gradient kernel backward (v2 never in HBM)?  The split
    engine is selected (f32_products()), the library takes the shape, the operands are 16-byte aligned; VQA_RELATION_FUSED=0 keeps the
    materialising path (measurement knob)."""
    if f32_products() != "split" or os.environ.get("VQA_RELATION_FUSED", "1") == "0":
        return False
    B, N, D = v.shape
    if any(x.data_ptr() % 16 != 0 for x in (v, t, c2, w)):
        return False
    return _lib.lib().vqa_relation_linear_split_supported(B, N, D, w.shape[0], float(p_drop)) == 1


def relation_projection_supported(v, w):
    """Can relation_projection serve v [B,N,D] (fp32, no gradient wanted) and the projection weight w [L,D]?"""
    return (v.is_cuda and v.dtype == torch.float32 and v.dim() == 3 and not v.requires_grad and w.dtype == torch.float32 and
            bool(_lib.lib().vqa_relation_projection_dgrad_supported(v.shape[0], v.shape[1], v.shape[2], w.shape[0])))


def relation_projection(v, t, c2, w, bias, p_drop=0.0, seed=0, pregated=False, pairwise=None):
    return RelationProjection.apply(v, t, c2, w, bias, p_drop, seed, pregated, pairwise)


def pairwise_projection_supported(v):
    """Can relation_projection's pairwise forward (the in-register N x N kernel with dropout in its store) take v?"""
    return bool(_lib.lib().vqa_pairwise_relation_reduce_drop_supported(v.shape[0], v.shape[1], v.shape[2]))


def column_sum(x, out=None, cols=None):
    """out[n] = sum_m x[m,n] for a 2-D fp32 / bf16 matrix -> fp32 [N]; fixed-order reduction, safe under graph replay.
    cols: only the first `cols` columns are summed (a padded activation's real width)."""
    x = _prep("x", x, _REGION_DTYPES)
    if x.dim() != 2:
        raise ValueError("column_sum: x must be 2-D, got %s" % (tuple(x.shape),))
    M, ld = x.shape
    N = ld if cols is None else int(cols)
    if not 0 < N <= ld:
        raise ValueError("column_sum: cols=%r outside (0, %d]" % (cols, ld))
    L_ = _lib.lib()
    ws_bytes = L_.vqa_column_sum_workspace_bytes(M, N)
    ws = torch.empty((ws_bytes + 3) // 4, device=x.device, dtype=torch.float32) if ws_bytes else None
    if out is None or out.dtype != torch.float32 or out.shape != (N,) or not out.is_contiguous():
        out = torch.empty(N, device=x.device, dtype=torch.float32)
    name = "column_sum" + _sfx(x.dtype)
    _launch(name, (M, N), getattr(L_, "vqa_" + name), _p(x), ld, _p(out), _p(ws), ws_bytes, M, N)
    return out


class LinearFn(torch.autograd.Function):
    """y = x W^T + b with the library GEMMs (rocBLAS / hipBLASLt through torch) and the bias gradient from
    ``column_sum``: torch.autograd's own linear backward takes grad_output.sum(0) with a multi-workgroup reduction whose
    semaphores are zeroed by a memset node, which replays wrongly inside a hipGraph on ROCm 7.2 (csrc/api.hip)."""

    @staticmethod
    def forward(ctx, x, w, b, act=None):
        # act "relu": the activation rides in the library GEMM's epilogue (one kernel instead of GEMM + clamp), and its
        # gradient mask is applied by whoever consumes grad_output first (see backward)
        ctx.has_bias = b is not None
        ctx.relu = act == "relu"
        if act not in (None, "relu"):
            raise ValueError("linear: act must be None or 'relu', got %r" % (act,))
        gemm = (x.numel() // x.shape[-1], w.shape[0], x.shape[-1])     # (M, N, K) of the library GEMM, for the timer
        if ctx.relu:
            with timed("library_gemm", gemm):
                if b is not None and x.dtype == torch.float32 and hasattr(torch, "_addmm_activation"):
                    y = torch._addmm_activation(b, x.reshape(-1, x.shape[-1]), w.t(), use_gelu=False).view(*x.shape[:-1], w.shape[0])
                else:
                    y = torch.relu(torch.nn.functional.linear(x, w, b))
            ctx.save_for_backward(x, w, y)
            ctx.bias = b
            return y
        ctx.save_for_backward(x, w)
        ctx.bias = b
        with timed("library_gemm", gemm):
            return torch.nn.functional.linear(x, w, b)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors[:2]
        y = ctx.saved_tensors[2] if ctx.relu else None
        x2 = x.reshape(-1, x.shape[-1])
        M, K = x2.shape
        N = w.shape[0]
        engine = (LinearFn.engine_dw and ctx.needs_input_grad[1] and M >= 4096 and x.dtype == torch.float32 and K % 2 == 0
                  and N % 2 == 0)
        mask_in_kernel = ctx.relu and engine and not ctx.needs_input_grad[0]
        if ctx.relu and not mask_in_kernel:
            gy = torch.ops.aten.threshold_backward(gy, y, 0)
        gy2 = gy.reshape(-1, gy.shape[-1])
        d_x = None
        if ctx.needs_input_grad[0]:
            with timed("library_gemm", (M, K, N)):
                d_x = (gy2 @ w).view(x.shape)
        if engine:
            # tall weight gradient (the region projections, M = B*N rows): the fp32 tile engine's split-row form, which
            # also returns the bias gradient as the column sums of its A fragments (no separate reduction) and, when no
            # data gradient needs the masked grad_output as a tensor, applies the relu mask while it stages the operand
            gy2, x2 = gy2.contiguous(), x2.contiguous()
            y2 = y.reshape(-1, N).contiguous() if mask_in_kernel else gy2
            d_w = _grad_like(w)
            d_b = _grad_like(ctx.bias) if ctx.has_bias else None
            L_ = _lib.lib()
            ws_bytes = L_.vqa_linear_act_bwd_workspace_bytes(M, K, N)
            ws = torch.empty((ws_bytes + 3) // 4, device=x.device, dtype=torch.float32)
            _launch("linear_act_bwd", (M, K, N, False, False), L_.vqa_linear_act_bwd, _p(x2), K, _p(w), _p(y2), _p(gy2),
                    None, _p(d_w), _p(d_b), _p(ws), ws_bytes, M, K, N, 1 if mask_in_kernel else 0, 0.0, 0, None)
            return d_x, d_w, d_b, None
        d_w = None
        if ctx.needs_input_grad[1]:
            with timed("library_gemm", (N, K, M)):
                if gy2.dtype == torch.float32 and x2.dtype == torch.float32 and w.dtype == torch.float32:
                    d_w = torch.mm(gy2.t(), x2, out=_grad_like(w))      # (straight into the flat gradient buffer, if any)
                else:
                    d_w = gy2.t() @ x2
        d_b = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            d_b = column_sum(gy2, out=_grad_like(ctx.bias) if ctx.bias.dtype == torch.float32 else None).to(gy.dtype)
        return d_x, d_w, d_b, None

    engine_dw = __import__("os").environ.get("VQA_ENGINE_DW", "1") == "1"


class StackParams(torch.autograd.Function):
    """torch.stack(params) without the copy when the G same-shaped parameters sit at equal spacing in one buffer (the
    trainer's FlatState lays the members of a stack group out that way): the result is then a strided view of that
    buffer.  Backward hands each parameter its slice of the stacked gradient (views, no kernel either)."""

    @staticmethod
    def forward(ctx, *ps):
        first = ps[0]
        G = len(ps)
        same = all(p.shape == first.shape and p.stride() == first.stride() and p.dtype == first.dtype for p in ps)
        if same and G > 1 and first.is_contiguous():
            esz = first.element_size()
            step = (ps[1].data_ptr() - first.data_ptr()) // esz
            base = first.untyped_storage().data_ptr()
            ok = step >= first.numel() and all(ps[i].data_ptr() - first.data_ptr() == i * step * esz and
                                               ps[i].untyped_storage().data_ptr() == base for i in range(G))
            if ok:
                return first.detach().as_strided((G,) + tuple(first.shape), (step,) + tuple(first.stride()),
                                                 first.storage_offset())
        return torch.stack([p.detach() for p in ps])

    @staticmethod
    def backward(ctx, g):
        return tuple(g.unbind(0))


def stack_params(params):
    return StackParams.apply(*params)


_ACT_CODES = {None: 0, "": 0, "relu": 1, "sigmoid": 2}


class BatchedLinearFn(torch.autograd.Function):
    """out[b,g,:] = act(x[b,g,:] W_g^T + bias_g) for G same-shaped layers at once (MyATT's per-glimpse MyLinear list,
    config/CoR2.py:133-134,143-147; the question projections; Mutan's question-side ranks): one batched library GEMM
    forward and two backward, with everything around them in one HIP kernel each way (bias + activation + layout;
    activation gradient + the G bias gradients) -- instead of G times {GEMM, bias, activation, slice} plus the zero-fill /
    add chain autograd builds for G slices."""

    @staticmethod
    def forward(ctx, x, w, b, group_first, act):
        # x [B,G,K] (any batch / group strides, K contiguous), w [G,A,K], b [G,A] (rows at any stride) or None
        #   -> [B,G,A] contiguous, or [G,B,A] contiguous when group_first
        B, G, K = x.shape
        A = w.shape[1]
        code = _ACT_CODES[act]
        # tall operands (the question encoder's input projections: 13 312 rows) on the split engine's batched NT kernel
        # (csrc/gru_gemm.hip), bias in its store; the [B,.]-sized uses (glimpse projections) stay on the batched library GEMM
        ctx.split = bool(B >= 1152 and x.is_cuda and f32_products() == "split" and x.stride(2) == 1 and x.stride(0) % 4 == 0
                         and x.stride(1) % 4 == 0 and w.is_contiguous() and K % 2 == 0 and A % 4 == 0
                         and gemm_nt_split_batched_ok(B, A, K, x.stride(0), x, w))
        if ctx.split:
            img = split_weights(w)
            direct = code == 0 and group_first and (b is None or b.stride(-1) == 1)
            y = torch.empty(G, B, A, device=x.device, dtype=torch.float32)
            gemm_nt_split_batched(x, 0, x.stride(1), x.stride(0), img, y, b if direct else None, w, False, G, B, A, K)
            if direct:
                ctx.bias = b
                ctx.save_for_backward(x, w, y)
                ctx.cfg = (True, code, b is not None)
                return y
        else:
            with timed("library_gemm", (x.shape[1] * x.shape[0], w.shape[1], w.shape[2])):
                y = torch.bmm(x.transpose(0, 1), w.transpose(1, 2))      # [G,B,A]
        out = torch.empty((G, B, A) if group_first else (B, G, A), device=y.device, dtype=torch.float32)
        ctx.bias = b
        if b is not None and b.stride(-1) != 1:
            b = b.contiguous()
        _launch("bias_act", (G, B, A, code), _lib.lib().vqa_bias_act, _p(y), _p(b), b.stride(0) if b is not None else 0,
                _p(out), G, B, A, code, int(bool(group_first)))
        ctx.save_for_backward(x, w, out)
        ctx.cfg = (bool(group_first), code, b is not None)
        return out

    @staticmethod
    def backward(ctx, gy):
        x, w, out = ctx.saved_tensors
        group_first, code, has_bias = ctx.cfg
        gy = _prep("grad_out", gy)
        G, B, A = out.shape if group_first else (out.shape[1], out.shape[0], out.shape[2])
        d_b = None
        if has_bias and ctx.needs_input_grad[2]:
            d_b = _grad_like(ctx.bias, rows_strided=True) if tuple(ctx.bias.shape) == (G, A) else \
                torch.empty(G, A, device=gy.device, dtype=torch.float32)
        if code == 0 and group_first and B >= 4096:
            # no activation, the gradient already lies [G,B,A]: it IS gz (no copy of a tall tensor -- 383 MB for the question
            # encoder's input projections); the bias gradients are G column sums
            gz = gy
            if d_b is not None:
                for g in range(G):
                    column_sum(gy[g], out=d_b[g])
        else:
            gz = torch.empty(G, B, A, device=gy.device, dtype=torch.float32)
            _launch("act_bwd_colsum", (G, B, A, code), _lib.lib().vqa_act_bwd_colsum, _p(gy), _p(out), _p(gz), _p(d_b),
                    d_b.stride(0) if d_b is not None else A, G, B, A, code, int(group_first))
        d_x = None
        K = w.shape[2]
        if ctx.split:
            from . import head
            if ctx.needs_input_grad[0]:      # d_x[b,g,:] = gz[g,b,:] W_g: the batched NT kernel against the transposed image
                d_x = torch.empty(B, G, K, device=gy.device, dtype=torch.float32)
                gemm_nt_split_batched(gz, 0, B * A, A, split_weights(w, transposed=True), d_x, None, w, True, G, B, K, A, c_gs=K, ldc=G * K)
            d_w = None
            if ctx.needs_input_grad[1]:      # d_w[g] = gz[g]^T x[:, g, :] over the B rows: one grouped launch (TN form)
                d_w = _grad_like(w)
                rest = []
                wide = A >= 2 * K        # the kernel PACKS its first operand and splits the second in its loop: give it the narrow one
                tmp = torch.empty(K, A, device=gy.device, dtype=torch.float32) if wide else None     # (to pack: 13312 x 620, not x 2400)
                for g in range(G):
                    if wide and gemm_tn_split(x, g * x.stride(1), x.stride(0), gz, g * B * A, A, tmp, B, K, A):
                        d_w[g].copy_(tmp.t())                                   # d_w^T came out; 6 MB transposed back
                    elif not gemm_tn_split(gz, g * B * A, A, x, g * x.stride(1), x.stride(0), d_w[g], B, A, K):
                        rest.append(g)
                if rest:
                    _grouped_products("batched_linear_dw", [(head.TN, gz, g * B * A, A, x, g * x.stride(1), x.stride(0), A, K, B, d_w[g])
                                                            for g in rest], engine="split")
            return d_x, d_w, d_b, None, None
        if ctx.needs_input_grad[0]:
            # written in the consumer's [B,G,K] layout (row stride G*K, batch stride K: a layout the strided-batched GEMM
            # takes as it is), so the kernels behind it get a contiguous gradient without a copy
            d_x = torch.empty(B, G, w.shape[2], device=gy.device, dtype=gz.dtype)
            with timed("library_gemm", (G * B, w.shape[2], A)):
                torch.bmm(gz, w, out=d_x.transpose(0, 1))
        d_w = None
        if ctx.needs_input_grad[1]:
            with timed("library_gemm", (G * A, w.shape[2], B)):
                if w.dtype == torch.float32 and x.dtype == torch.float32:
                    d_w = torch.bmm(gz.transpose(1, 2), x.transpose(0, 1), out=_grad_like(w))   # (flat gradient buffer, if any)
                else:
                    d_w = torch.bmm(gz.transpose(1, 2), x.transpose(0, 1))
        return d_x, d_w, d_b, None, None


def batched_linear(x, w, b=None, group_first=False, act=None):
    if act not in _ACT_CODES:
        raise ValueError("batched_linear: act must be None, 'relu' or 'sigmoid', got %r" % (act,))
    return BatchedLinearFn.apply(x, w, b, group_first, act)


class DropoutGroups(torch.autograd.Function):
    """F.dropout(x, p) with the counter-hash mask of the fused kernels, G independent draws over one input at once:
    x [..., K] -> [G, ..., K] (G = 0: a single draw, same shape as x).  csrc/epilogue.hip."""

    @staticmethod
    def forward(ctx, x, p_drop, seed, groups):
        x = _prep("x", x)
        K = x.shape[-1]
        M = x.numel() // K
        G = max(int(groups), 1)
        out = torch.empty(((G,) if groups else ()) + tuple(x.shape), device=x.device, dtype=torch.float32)
        sv, sp = _seed_args(seed)
        _launch("dropout_groups_fwd", (G, M, K), _lib.lib().vqa_dropout_groups_fwd, _p(x), K, _p(out), float(p_drop), sv, sp, G, M, K)
        ctx.cfg = (float(p_drop), seed, G, M, K, tuple(x.shape))
        return out

    @staticmethod
    def backward(ctx, gy):
        p_drop, seed, G, M, K, shape = ctx.cfg
        gy = _prep("grad_out", gy)
        d_x = torch.empty(shape, device=gy.device, dtype=torch.float32)
        sv, sp = _seed_args(seed)
        _launch("dropout_groups_bwd", (G, M, K), _lib.lib().vqa_dropout_groups_bwd, _p(gy), _p(d_x), p_drop, sv, sp, G, M, K)
        return d_x, None, None, None


def dropout(x, p_drop, groups=0):
    """Training-mode dropout of an fp32 GPU tensor (a fresh seed per call from the step's seed stream)."""
    if not p_drop:
        return x if not groups else x.unsqueeze(0).expand((groups,) + tuple(x.shape))
    return DropoutGroups.apply(x.contiguous(), p_drop, next_dropout_seed(), groups)


class RankProduct(torch.autograd.Function):
    """out[b,:] = sum_r h1[b,r,:] * h2[b,r,:] -- the rank sum of the vector-vector Mutan fusion (fusion_final), one kernel
    each way (csrc/epilogue.hip) instead of multiply + reduce forward and two broadcast multiplies backward."""

    @staticmethod
    def forward(ctx, h1, h2):
        h1, h2 = _prep("h1", h1), _prep("h2", h2)
        if h1.dim() != 3 or h1.shape != h2.shape:
            raise ValueError("rank_product: h1 and h2 must both be [B,R,H], got %s / %s" % (tuple(h1.shape), tuple(h2.shape)))
        B, R, H = h1.shape
        out = torch.empty(B, H, device=h1.device, dtype=torch.float32)
        _launch("rank_product_fwd", (B, R, H), _lib.lib().vqa_rank_product_fwd, _p(h1), _p(h2), _p(out), B, R, H)
        ctx.save_for_backward(h1, h2)
        return out

    @staticmethod
    def backward(ctx, g):
        h1, h2 = ctx.saved_tensors
        g = _prep("grad_out", g)
        B, R, H = h1.shape
        d_h1, d_h2 = torch.empty_like(h1), torch.empty_like(h2)
        _launch("rank_product_bwd", (B, R, H), _lib.lib().vqa_rank_product_bwd, _p(g), _p(h1), _p(h2), _p(d_h1), _p(d_h2), B, R, H)
        return d_h1, d_h2


def rank_product(h1, h2):
    if h1.dtype != torch.float32 or h1.shape[-1] % 2:
        return (h1 * h2).sum(dim=1)
    return RankProduct.apply(h1, h2)


class RelationGates(torch.autograd.Function):
    """(q1, q2, pooled) -> (t, c2, t', c2') with t = q1 * pooled, c2 = q2: the two operands of the closed-form relation step
    (config/CoR2.py:191-199, :216), handed out TWICE -- once for the relation / projection node, once for the second
    attention's pooled map -- so that both consumers' gradients reach THIS node's backward together: one kernel
    (vqa_gate_product_bwd) forms d q1 = (g + g') * pooled, d pooled = (g + g') * q1, d q2 = h + h' instead of autograd's two
    accumulations and two products.  All [B,D] fp32."""

    @staticmethod
    def forward(ctx, q1, q2, pooled):
        q1, q2 = _prep("q1", q1), _prep("q2", q2)
        # pooled: [B,D] dense, or glimpse 0 of the [B,G,D] pooled tensor read in place (rows D floats wide, G * D apart)
        if pooled.dim() != 2 or pooled.stride(1) != 1 or pooled.dtype != torch.float32 or not pooled.is_cuda or \
                pooled.stride(0) % 4 or pooled.data_ptr() % 16:
            pooled = _prep("pooled", pooled)
        if q1.shape != pooled.shape or q2.shape != pooled.shape or pooled.shape[1] % 4:
            raise ValueError("relation_gates: q1, q2, pooled must share one [B,D] shape with D a multiple of 4")
        rows, cols = pooled.shape
        t = torch.empty_like(q1)
        _launch("gate_product_fwd", (q1.numel(),), _lib.lib().vqa_gate_product_fwd, _p(q1), _p(pooled), pooled.stride(0), _p(t),
                rows, cols)
        ctx.save_for_backward(q1, pooled)
        return t, q2.view_as(q2), t.view_as(t), q2.view_as(q2)

    @staticmethod
    def backward(ctx, g1, h1, g2, h2):
        q1, pooled = ctx.saved_tensors
        if g1 is None:
            g1, g2 = g2, None
        if h1 is None:
            h1, h2 = h2, None
        if g1 is None:
            g1 = torch.zeros_like(q1)
        if h1 is None:
            h1 = torch.zeros_like(q1)
        g1, h1 = _prep("d_t", g1), _prep("d_c2", h1)
        g2 = _prep("d_t'", g2) if g2 is not None else None
        h2 = _prep("d_c2'", h2) if h2 is not None else None
        d_q1, d_pooled, d_q2 = torch.empty_like(q1), torch.empty_like(q1), torch.empty_like(q1)
        rows, cols = pooled.shape
        _launch("gate_product_bwd", (q1.numel(),), _lib.lib().vqa_gate_product_bwd, _p(g1), _p(g2), _p(h1), _p(h2), _p(q1),
                _p(pooled), pooled.stride(0), _p(d_q1), _p(d_pooled), _p(d_q2), rows, cols)
        return d_q1, d_q2, d_pooled


def relation_gates(q1, q2, pooled):
    return RelationGates.apply(q1, q2, pooled)


class WithFirstGroup(torch.autograd.Function):
    """pooled [B,G,D] -> (pooled, pooled[:,0]) for the two consumers of the first attention's pooled features (its own
    glimpse projections; the relation step, which reads glimpse 0).  Backward adds the slice gradient into a copy of
    the full one -- instead of autograd's zero-fill of a full-size tensor, slice copy and full-size add."""

    @staticmethod
    def forward(ctx, pooled):
        ctx.shape = tuple(pooled.shape)
        return pooled.view_as(pooled), pooled[:, 0]

    @staticmethod
    def backward(ctx, g_full, g_first):
        if g_full is None and g_first is None:
            return None
        ref = g_full if g_full is not None else g_first
        # (the full gradient is the fresh output of the glimpse projections' backward and has no other reader: the slice is
        #  added IN PLACE -- a clone first was a 16.8 MB device copy per step)
        if g_full is None:
            g = torch.zeros(ctx.shape, device=ref.device, dtype=ref.dtype)
        elif g_first is not None and not g_full.is_contiguous():      # (an expanded gradient -- sum().backward() -- is not writable)
            g = g_full.clone()
        else:
            g = g_full
        if g_first is not None:
            g[:, 0].add_(g_first)
        return g


def with_first_group(pooled):
    return WithFirstGroup.apply(pooled)


class SplitGroups(torch.autograd.Function):
    """t [G, ...] -> its chunks along dim 0 (``sizes``; a chunk of one group loses the dim) as views.  Forward is free;
    backward is ONE concatenation of the chunks' gradients -- the slices ``t[0], t[1], t[2:4]`` would instead make autograd
    zero-fill a full-size tensor per slice, copy the slice gradient in and add the tensors up (7 kernels for the four
    question projections of CoR2, 5 for its two gates)."""

    @staticmethod
    def forward(ctx, t, sizes):
        if sum(sizes) != t.shape[0]:
            raise ValueError("split_groups: sizes %s do not add up to %d groups" % (sizes, t.shape[0]))
        ctx.sizes, ctx.tail = tuple(sizes), tuple(t.shape[1:])
        outs, o = [], 0
        for n in sizes:
            outs.append(t[o] if n == 1 else t[o:o + n])
            o += n
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        parts = []
        for n, g in zip(ctx.sizes, grads):
            if g is None:
                g = torch.zeros((n,) + ctx.tail, device=next(x for x in grads if x is not None).device, dtype=torch.float32)
            parts.append(g.reshape((n,) + ctx.tail))
        return torch.cat(parts, 0), None


def split_groups(t, sizes):
    return SplitGroups.apply(t, tuple(sizes))


def linear(x, w, b=None, act=None):
    """act(F.linear) for GPU tensors with a replay-safe bias gradient (see LinearFn); CPU tensors take the torch ops."""
    if x.is_cuda:
        return LinearFn.apply(x, w, b, act)
    y = torch.nn.functional.linear(x, w, b)
    return torch.relu(y) if act == "relu" else y


def split_weights(w, transposed=False):
    """The packed three-plane bf16 image of G weight matrices for vqa_gemm_nt_split_batched (csrc/gru_gemm.hip): w [G,N,K], or
    [G,K,N] read transposed (the image then multiplies by W instead of W^T: a data gradient).  Split once, reused by every launch
    that multiplies against these weights (all 26 time steps of the question encoder)."""
    w = _prep("w", w)
    G, d1, d2 = w.shape
    N, K = (d2, d1) if transposed else (d1, d2)
    L_ = _lib.lib()
    nbytes = L_.vqa_split_weights_bytes(G, N, K)
    img = torch.empty((nbytes + 3) // 4, device=w.device, dtype=torch.float32)
    _launch("split_weights_pack", (G, N, K, bool(transposed)), L_.vqa_split_weights_pack, _p(w), w.stride(0), w.stride(1), int(transposed),
            _p(img), nbytes, G, N, K)
    return img


def gemm_nt_split_batched(a, a_off, a_gs, lda, img, c, bias, w, transposed, G, M, N, K, c_gs=None, ldc=None):
    """c[g] [M,N] = a_g [M,K] W_g^T (+ bias[g]) on the split engine, G problems in one launch.  a: a tensor holding the operands,
    a_g starting `a_off + g * a_gs` elements in, rows of stride lda; img = split_weights(w, transposed); c [G,M,N] contiguous, or
    any layout with problem stride c_gs and row stride ldc (elements)."""
    L_ = _lib.lib()
    ptr = ctypes.c_void_p(a.data_ptr() + 4 * int(a_off))
    w_sn, w_sk = (1, w.stride(1)) if transposed else (w.stride(1), 1)
    _launch("gemm_nt_split_batched", (G, M, N, K), L_.vqa_gemm_nt_split_batched, ptr, int(a_gs), int(lda), _p(img), _p(c),
            M * N if c_gs is None else int(c_gs), N if ldc is None else int(ldc),
            _p(bias), bias.stride(0) if bias is not None else 0, _p(w), w.stride(0), int(w_sn), int(w_sk), G, M, N, K)
    return c


def gemm_nt_split_batched_ok(M, N, K, lda, *tensors):
    return (_lib.lib().vqa_gemm_nt_split_batched_supported(M, N, K, lda, N) == 1
            and all(t is None or (t.dtype == torch.float32 and t.data_ptr() % 16 == 0) for t in tensors))


def gemm_tn_split(g, g_off, ldg, x, x_off, ldx, d_w, M, N1, N2):
    """d_w [N1,N2] = g^T x over M rows on the split engine's weight-gradient kernels (csrc/gru_gemm.hip: vqa_gemm_tn_split); g, x:
    tensors holding the operands from element offsets g_off / x_off, rows of stride ldg / ldx.  False when the shape or the
    alignment is outside of the engine (the caller then takes the grouped launch)."""
    L_ = _lib.lib()
    gp, xp = g.data_ptr() + 4 * int(g_off), x.data_ptr() + 4 * int(x_off)
    if os.environ.get("VQA_TN_SPLIT", "1") == "0":      # (measurement knob: the grouped launch instead)
        return False
    if (L_.vqa_gemm_tn_split_supported(M, N1, N2, int(ldg), int(ldx)) != 1 or gp % 8 or xp % 16 or d_w.data_ptr() % 16
            or not d_w.is_contiguous()):
        return False
    nbytes = L_.vqa_gemm_tn_split_workspace_bytes(M, N1, N2)
    ws = torch.empty((nbytes + 3) // 4, device=d_w.device, dtype=torch.float32)
    _launch("gemm_tn_split", (M, N1, N2), L_.vqa_gemm_tn_split, ctypes.c_void_p(gp), int(ldg), ctypes.c_void_p(xp), int(ldx), _p(d_w), _p(ws),
            nbytes, M, N1, N2)
    return True


def _grouped_products(name, problems, engine=None):
    """A list of (form, A, a_off, lda, Bm, b_off, ldb, M, N, K, out) products as ONE grouped launch of the K6 engine (head.Phase:
    csrc/grouped_gemm{,_split}.hip) -- the repo's generic hand-written GEMM, for shapes the register-tile engines do not take."""
    from . import head
    ph = head.Phase(problems[0][1].device, name)
    ph.force_engine = engine
    for form, A, a_off, lda, Bm, b_off, ldb, M, N, K, out in problems:
        t = ph.target(M, N)
        ph.gemm(t, form, A, lda, Bm, ldb, K, a_off=a_off, b_off=b_off)
        ph.job(head.EPI_SUM, t, out, N)
    ph.run()


class GruSequence(torch.autograd.Function):
    """The recurrent part of putils.BayesianGRU.forward (putils/__init__.py:704-731): T steps of
    r,i = sigmoid(gi_{r,i}[t] + W_h{r,i}(h*m)), n = af(gi_n[t] + r * W_hn(h*m_n)), h = (1-i) n + i h.
    gi [3,B,T,H] (the input-side projections of all steps), w [3,H,H] (W_hr, W_hi, W_hn stacked), masks [3,B,H] or
    None (sequence-shared dropout) -> all hidden states [T,B,H].
    Round 6: every product is the repo's own kernel (rounds 1-5: torch.bmm).  Per step ONE launch of the split engine's batched
    NT GEMM (csrc/gru_gemm.hip: the three gates' [B,H] x [H,H]^T against weight images split once per pass -- W for the forward,
    W^T for the data gradients) + ONE gate kernel (csrc/gru.hip) each way; the recurrent weight gradient is one grouped launch on
    the split engine over all T*B rows at the end (csrc/grouped_gemm_split.hip, TN form).  Shapes the batched kernel does not take
    (B < 64, H % 4) run the same products as grouped launches of the K6 engine."""

    @staticmethod
    def forward(ctx, gi, w, masks, af):
        gi, w = _prep("gi", gi), _prep("w", w)
        masks = _prep("masks", masks) if masks is not None else None
        G, B, T, H = gi.shape
        if G != 3 or w.shape != (3, H, H) or (masks is not None and masks.shape != (3, B, H)):
            raise ValueError("gru_sequence: gi [3,B,T,H], w [3,H,H], masks [3,B,H] expected")
        code = {"relu": 1, "tanh": 3}[af]
        dev = gi.device
        L_ = _lib.lib()
        out = torch.empty(T, B, H, device=dev, dtype=torch.float32)
        hist = torch.zeros(3, T, B, H, device=dev, dtype=torch.float32)        # hm_t = h_{t-1} * m_g; hm_0 = 0
        saved = torch.empty(4, T, B, H, device=dev, dtype=torch.float32)       # r, i, n, a_n
        h0 = torch.zeros(B, H, device=dev, dtype=torch.float32)
        a = torch.empty(3, B, H, device=dev, dtype=torch.float32)
        gs = T * B * H
        fast = gemm_nt_split_batched_ok(B, H, H, H, hist, w)
        img = split_weights(w) if fast else None
        for t in range(T):
            if t == 0:
                a.zero_()                                                        # hm_0 = 0: no product to form
            elif fast:
                gemm_nt_split_batched(hist, t * B * H, gs, H, img, a, None, w, False, 3, B, H, H)          # a[g] = hm_t[g] W_g^T
            else:
                from . import head
                _grouped_products("gru_step_fwd", [(head.NT, hist, g * gs + t * B * H, H, w, g * H * H, H, B, H, H, a[g]) for g in range(3)])
            nxt = ctypes.c_void_p(hist.data_ptr() + 4 * (t + 1) * B * H) if t + 1 < T else None
            _launch("gru_gates_fwd", (B, T, H), L_.vqa_gru_gates_fwd, _p(gi), _p(a), _p(out[t - 1] if t else h0), _p(masks),
                    _p(out[t]), nxt, gs, _p(saved[0, t]), _p(saved[1, t]), _p(saved[2, t]), _p(saved[3, t]), B, T, H, t, code)
        ctx.save_for_backward(w, masks, out, hist, saved, h0)
        ctx.cfg = (B, T, H, code, fast)
        return out

    @staticmethod
    def backward(ctx, d_out):
        from . import head
        w, masks, out, hist, saved, h0 = ctx.saved_tensors
        B, T, H, code, fast = ctx.cfg
        d_out = _prep("grad_out", d_out)
        dev = d_out.device
        L_ = _lib.lib()
        gz = torch.empty(3, T, B, H, device=dev, dtype=torch.float32)
        d_gi = torch.empty(3, B, T, H, device=dev, dtype=torch.float32)
        carry = [torch.empty(B, H, device=dev, dtype=torch.float32) for _ in range(2)]
        gs = T * B * H
        buf = torch.empty(3, B, H, device=dev, dtype=torch.float32)
        img_t = split_weights(w, transposed=True) if fast else None
        dhm = None
        for t in range(T - 1, -1, -1):
            _launch("gru_gates_bwd", (B, T, H), L_.vqa_gru_gates_bwd, _p(d_out[t]), _p(carry[(t + 1) & 1]) if t + 1 < T else None,
                    _p(dhm), _p(masks), _p(saved[0, t]), _p(saved[1, t]), _p(saved[2, t]), _p(saved[3, t]),
                    _p(out[t - 1] if t else h0), ctypes.c_void_p(gz.data_ptr() + 4 * t * B * H), gs, _p(d_gi), _p(carry[t & 1]),
                    B, T, H, t, code)
            if t > 0:                                                           # gradient at hm_t: dhm[g] = gz_t[g] W_g   [3,B,H]
                if fast:
                    dhm = gemm_nt_split_batched(gz, t * B * H, gs, H, img_t, buf, None, w, True, 3, B, H, H)
                else:
                    _grouped_products("gru_step_bwd", [(head.NN, gz, g * gs + t * B * H, H, w, g * H * H, H, B, H, H, buf[g]) for g in range(3)])
                    dhm = buf
        d_w = None
        if ctx.needs_input_grad[1]:
            # d_w[g] = gz[g]^T hm[g] over all T*B rows: one grouped launch, contraction split into slabs and summed in fixed order
            d_w = _grad_like(w)
            rest = [g for g in range(3) if not (fast and gemm_tn_split(gz, g * gs, H, hist, g * gs, H, d_w[g], T * B, H, H))]
            if rest:      # (short sequences / odd widths: the grouped launch, contraction split into slabs and summed in fixed order)
                _grouped_products("gru_dw", [(head.TN, gz, g * gs, H, hist, g * gs, H, H, H, T * B, d_w[g]) for g in rest],
                                  engine="split" if (f32_products() == "split" and T * B >= 384) else None)
        return d_gi, d_w, None, None


def gru_sequence(gi, w, masks, af):
    if af not in ("relu", "tanh"):
        raise ValueError("gru_sequence: af must be 'relu' or 'tanh', got %r" % (af,))
    return GruSequence.apply(gi, w, masks, af)


class EmbeddingFn(torch.autograd.Function):
    """nn.Embedding lookup whose backward is a zero-filled table + index_add_ (atomic adds): torch's own
    embedding_dense_backward issues a memset, which must not sit inside a replayed hipGraph (csrc/api.hip)."""

    @staticmethod
    def forward(ctx, weight, idx, padding_idx):
        ctx.save_for_backward(idx)
        ctx.cfg = (weight.shape, padding_idx)
        return weight.index_select(0, idx.reshape(-1)).view(*idx.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        shape, padding_idx = ctx.cfg
        flat = idx.reshape(-1)
        g2 = g.reshape(flat.numel(), shape[1])
        if padding_idx is not None:
            g2 = g2 * (flat != padding_idx).unsqueeze(1)
        d_w = torch.zeros(shape, device=g.device, dtype=g.dtype).index_add_(0, flat, g2)
        return d_w, None, None


def embedding(weight, idx, padding_idx=None):
    return EmbeddingFn.apply(weight, idx, padding_idx)


class KldSumLoss(torch.autograd.Function):
    """KLDivLoss(size_average=False)(log_softmax(logits), target) (train.py:536-544) with its gradient from the same
    pass over the logits; the B row losses are added in a fixed order (no atomics, no memset: replays cleanly)."""

    @staticmethod
    def forward(ctx, logits, target):
        logits, target = _prep("logits", logits), _prep("target", target)
        if logits.dim() != 2 or target.shape != logits.shape:
            raise ValueError("kld_sum_loss: logits and target must both be [B,C], got %s and %s"
                             % (tuple(logits.shape), tuple(target.shape)))
        B, C = logits.shape
        loss = torch.empty((), device=logits.device, dtype=torch.float32)
        need = ctx.needs_input_grad[0]
        d_logits = torch.empty_like(logits) if need else None
        ws = torch.empty(B, device=logits.device, dtype=torch.float32)
        _launch("kld_sum_loss", (B, C, need), _lib.lib().vqa_kld_sum_loss, _p(logits), _p(target), _p(loss), _p(d_logits),
                _p(ws), 4 * B, B, C)
        if need:
            ctx.save_for_backward(d_logits)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d_logits,) = ctx.saved_tensors
        return d_logits * g, None


def kld_sum_loss_and_grad(logits, target):
    """(loss, dL/dlogits) of the KLD-sum loss from ONE pass, outside autograd: a train step that backpropagates from the
    logits with this gradient (``torch.autograd.backward(logits, d_logits)``) skips the loss node's multiply by the
    incoming scalar 1."""
    lg, target = _prep("logits", logits.detach()), _prep("target", target)
    if lg.dim() != 2 or target.shape != lg.shape:
        raise ValueError("kld_sum_loss: logits and target must both be [B,C], got %s and %s" % (tuple(lg.shape), tuple(target.shape)))
    B, C = lg.shape
    loss = torch.empty((), device=lg.device, dtype=torch.float32)
    d_logits = torch.empty_like(lg)
    ws = torch.empty(B, device=lg.device, dtype=torch.float32)
    _launch("kld_sum_loss", (B, C, True), _lib.lib().vqa_kld_sum_loss, _p(lg), _p(target), _p(loss), _p(d_logits), _p(ws), 4 * B, B, C)
    return loss, d_logits


def kld_sum_loss(logits, target):
    return KldSumLoss.apply(logits, target)


def grad_norm_clip_coef(g_flat, max_norm, out, workspace):
    """out[0] = ||g_flat||_2, out[1] = min(1, max_norm/(norm+1e-6)) -- clip_grad_norm_ semantics (train.py:82)."""
    g_flat = _prep("g_flat", g_flat)
    _launch("grad_norm_clip_coef", (g_flat.numel(),), _lib.lib().vqa_grad_norm_clip_coef, _p(g_flat), g_flat.numel(),
            float(max_norm), _p(out), _p(workspace), workspace.numel() * workspace.element_size())


def adam_step(p_flat, g_flat, m_flat, v_flat, norm_and_coef, lr, beta1, beta2, eps, step):
    """One fused Adam update of the flat parameter buffer on gradients scaled by norm_and_coef[1] (train.py:86)."""
    for name, t in (("p", p_flat), ("g", g_flat), ("m", m_flat), ("v", v_flat)):
        if _prep(name, t) is not t:
            raise ValueError("adam_step: %s must be contiguous" % name)
    _launch("adam_step", (p_flat.numel(),), _lib.lib().vqa_adam_step, _p(p_flat), _p(g_flat), _p(m_flat), _p(v_flat),
            p_flat.numel(), _p(norm_and_coef), float(lr), float(beta1), float(beta2), float(eps), int(step))


def adam_step_dyn(p_flat, g_flat, m_flat, v_flat, norm_and_coef, step_scalars, beta1, beta2, eps):
    """adam_step with {lr/(1-b1^t), 1/sqrt(1-b2^t)} read from the device tensor ``step_scalars`` (graph replays)."""
    _launch("adam_step_dyn", (p_flat.numel(),), _lib.lib().vqa_adam_step_dyn, _p(p_flat), _p(g_flat), _p(m_flat),
            _p(v_flat), p_flat.numel(), _p(norm_and_coef), _p(step_scalars), float(beta1), float(beta2), float(eps))


host_seed_draws = 0  # bumped whenever a kernel's dropout seed is drawn on the host (such a step cannot be graph-replayed)
# [int64 CUDA tensor [1], next salt] installed by the trainer for graph-replayed steps.  Per thread: nn.DataParallel
# drives one replica per thread, each on its own device, and a seed word belongs to one device
_seed_state = threading.local()
_SALT_STRIDE = 0x9E3779B97F4A7C15 & 0x3FFFFFFFFFFFFFFF


def draw_host_seed():
    """A fresh 62-bit dropout seed from torch's CPU generator (so torch.manual_seed governs the fused masks)."""
    global host_seed_draws
    host_seed_draws += 1
    return int(torch.randint(0, 2 ** 62, (1,), device="cpu").item())


def set_device_seed(tensor):
    """Install (or with None remove) the device word the fused dropout kernels read their per-step seed from."""
    _seed_state.word = [tensor, 0] if tensor is not None else None


def next_dropout_seed():
    """Seed for one fused-dropout call: (device tensor, salt) when the trainer installed a device seed word -- every
    call site of a step gets its own salt, the word itself moves once per step -- else a fresh host seed."""
    word = getattr(_seed_state, "word", None)
    if word is None:
        return draw_host_seed()
    word[1] += 1
    return (word[0], (word[1] * _SALT_STRIDE) & 0x3FFFFFFFFFFFFFFF)


def begin_step_salts():
    """Restart the per-call salt sequence (call at the start of every forward so eager and replayed steps agree)."""
    word = getattr(_seed_state, "word", None)
    if word is not None:
        word[1] = 0


def linear_dropout_mask(M, K, p_drop, seed, device):
    """The keep/(1-p) mask [M,K] exactly as K5 draws it (tests hand it to the oracle)."""
    mask = torch.empty(M, K, device=device, dtype=torch.float32)
    sv, sp = _seed_args(seed)
    _launch("linear_dropout_mask", (M, K), _lib.lib().vqa_linear_dropout_mask, _p(mask), float(p_drop), sv, sp, M, K)
    return mask


def object_difference_dropout_mask(B, N, L, p_drop, seed, device):
    """The keep/(1-p) mask [B,N,N*L] exactly as K2 draws it (tests hand it to the oracle)."""
    mask = torch.empty(B, N, N * L, device=device, dtype=torch.float32)
    sv, sp = _seed_args(seed)
    _launch("object_difference_dropout_mask", (B, N, L), _lib.lib().vqa_object_difference_dropout_mask,
            _p(mask), float(p_drop), sv, sp, B, N, L)
    return mask


def pairwise_relation_reduce(v, q1, q2, alpha, glimpse=0, mode=1, dual=False):
    """dual=True returns (v2, alias of v2): give one to each of two consumers and their gradients are added inside the
    backward kernel instead of by an extra full-size add."""
    return PairwiseRelationReduce.apply(v, q1, q2, alpha, glimpse, mode, dual)


def softmax_attention_pool(logits, v):
    return SoftmaxAttentionPool.apply(logits, v)


def lowrank_bilinear_fusion(x, h2, weights, biases, gate_dx=False, packed=None):
    """gate_dx: the gradient returned for x is already multiplied by (x > 0) -- see LowRankBilinearFusion.
    packed (bf16 path): the layer's shadows from the caller's ShadowPlan (LowRankBilinearFusionBf16)."""
    if x.dtype == torch.bfloat16:
        return LowRankBilinearFusionBf16.apply(x, h2, gate_dx, packed, *weights, *biases)
    return LowRankBilinearFusion.apply(x, h2, gate_dx, *weights, *biases)


def object_difference_attention(vl, ql, w, bias, p_drop=0.0, seed=0, gate_dvl=False):
    return ObjectDifferenceAttention.apply(vl, ql, w, bias, p_drop, seed, gate_dvl)
