"""float64 numpy closed forms -- forward AND hand-derived backward -- of the four kernels
the HIP library implements.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

These are independent of torch autograd: tests check them against autograd of
``reference_faithful`` (and against the golden vectors), and then check the HIP
kernels against them.  Layouts are the C-ABI's (include/vqa_mi355x.h): row-major,
B = samples, N = regions, D = region feature width, G = glimpses, L = low dim,
H = hidden dim, R = rank.
"""
import numpy as np

F64 = np.float64


# --------------------------------------------------------------------------------------
# K1  pairwise relation build + alpha-weighted reduce   (config/CoR2.py:191-199 and :216)
# --------------------------------------------------------------------------------------
def pairwise_relation_reduce_fwd(v, q1, q2, alpha):
    """v [B,N,D], q1,q2 [B,D], alpha [B,N]  ->  v2 [B,N,D]
    v2[b,j,:] = sum_i alpha[b,i] * (v[b,i,:]*q1[b,:] + v[b,j,:]*q2[b,:])   (pairwise order kept)."""
    v, q1, q2, alpha = (np.asarray(x, F64) for x in (v, q1, q2, alpha))
    B, N, D = v.shape
    out = np.zeros((B, N, D), F64)
    for b in range(B):
        left = v[b] * q1[b]                      # [N,D]  v_i * q1
        right = v[b] * q2[b]                     # [N,D]  v_j * q2
        for i in range(N):
            out[b] += alpha[b, i] * (left[i][None, :] + right)
    return out


def pairwise_relation_reduce_bwd(v, q1, q2, alpha, g):
    """g = dL/dv2 [B,N,D]  ->  (dalpha [B,N], dq1 [B,D], dq2 [B,D], dv [B,N,D])   (SURVEY App. B)."""
    v, q1, q2, alpha, g = (np.asarray(x, F64) for x in (v, q1, q2, alpha, g))
    gsum = g.sum(axis=1)                                           # [B,D]   sum_j g[j,:]
    asum = alpha.sum(axis=1)                                       # [B]
    pooled = np.einsum("bn,bnd->bd", alpha, v)                     # sum_i alpha_i v_i
    # dalpha[i] = sum_{j,d} g[j,d]*(v[i,d]q1[d] + v[j,d]q2[d])
    dalpha = np.einsum("bnd,bd->bn", v, q1 * gsum) + np.einsum("bjd,bjd,bd->b", g, v, q2)[:, None]
    dq1 = pooled * gsum
    dq2 = asum[:, None] * np.einsum("bjd,bjd->bd", g, v)
    dv = alpha[:, :, None] * (q1 * gsum)[:, None, :] + asum[:, None, None] * q2[:, None, :] * g
    return dalpha, dq1, dq2, dv


# --------------------------------------------------------------------------------------
# K3  softmax over regions + attention-weighted region pooling  (config/CoR2.py:132,142; putils:89-95)
# --------------------------------------------------------------------------------------
def softmax_attention_pool_fwd(logits, v):
    """logits [B,N,G], v [B,N,D] -> alpha [B,N,G] (softmax over N), pooled [B,G,D] = alpha^T v."""
    logits, v = np.asarray(logits, F64), np.asarray(v, F64)
    z = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(z)
    alpha = e / e.sum(axis=1, keepdims=True)
    pooled = np.einsum("bng,bnd->bgd", alpha, v)
    return alpha, pooled


def softmax_attention_pool_bwd(alpha, v, dpooled, dalpha_ext=None):
    """-> (dlogits [B,N,G], dv [B,N,D]).  dalpha_ext is a gradient that reaches alpha directly
    (CoR2 uses alpha1[...,0] as the relation weights)."""
    alpha, v, dpooled = (np.asarray(x, F64) for x in (alpha, v, dpooled))
    dalpha = np.einsum("bgd,bnd->bng", dpooled, v)
    if dalpha_ext is not None:
        dalpha = dalpha + np.asarray(dalpha_ext, F64)
    inner = (alpha * dalpha).sum(axis=1, keepdims=True)
    dlogits = alpha * (dalpha - inner)
    dv = np.einsum("bng,bgd->bnd", alpha, dpooled)
    return dlogits, dv


# --------------------------------------------------------------------------------------
# K4  low-rank bilinear (Mutan) fusion   (putils/__init__.py:205-241)
# --------------------------------------------------------------------------------------
def lowrank_bilinear_fusion_fwd(x, w1, b1, h2):
    """x [B,N,L]; w1 [R,H,L]; b1 [R,H]; h2 [B,R,H] (= Linear2_r(x2), the question-side factor)
    -> out [B,N,H] = sum_r (x W1_r^T + b1_r) * h2[:,r,None,:];  also returns h1 [B,N,R,H]."""
    x, w1, b1, h2 = (np.asarray(a, F64) for a in (x, w1, b1, h2))
    h1 = np.einsum("bnl,rhl->bnrh", x, w1) + b1[None, None]
    out = (h1 * h2[:, None]).sum(axis=2)
    return out, h1


def lowrank_bilinear_fusion_bwd(x, w1, b1, h2, g):
    """g = dL/dout [B,N,H] -> (dx [B,N,L], dw1 [R,H,L], db1 [R,H], dh2 [B,R,H])."""
    x, w1, b1, h2, g = (np.asarray(a, F64) for a in (x, w1, b1, h2, g))
    h1 = np.einsum("bnl,rhl->bnrh", x, w1) + b1[None, None]
    gh = g[:, :, None, :] * h2[:, None]                            # [B,N,R,H]  dL/dh1
    dx = np.einsum("bnrh,rhl->bnl", gh, w1)
    dw1 = np.einsum("bnrh,bnl->rhl", gh, x)
    db1 = gh.sum(axis=(0, 1))
    dh2 = (g[:, :, None, :] * h1).sum(axis=1)
    return dx, dw1, db1, dh2


# --------------------------------------------------------------------------------------
# K2  object-difference attention logits  (config/ODA.py:216-222 + the 1x1 conv of :149)
# --------------------------------------------------------------------------------------
def object_difference_logits_fwd(vl, ql, w, bias, mask=None):
    """vl [B,N,L], ql [B,L], w [G,N*L], bias [G], mask [B,N,N*L] (dropout keep/(1-p), or None)
    -> logits [B,N,G];  logits[b,i,g] = bias[g] + sum_{j,d} w[g,j*L+d]*mask[b,i,j*L+d]*(vl[b,i,d]-vl[b,j,d])*ql[b,d]."""
    vl, ql, w, bias = (np.asarray(a, F64) for a in (vl, ql, w, bias))
    B, N, L = vl.shape
    diff = (vl[:, :, None, :] - vl[:, None, :, :]) * ql[:, None, None, :]      # [B,N(i),N(j),L]
    vq = diff.reshape(B, N, N * L)
    if mask is not None:
        vq = vq * np.asarray(mask, F64)
    return vq @ w.T + bias[None, None]


def object_difference_logits_bwd(vl, ql, w, dlogits, mask=None):
    """-> (dvl [B,N,L], dql [B,L], dw [G,N*L], dbias [G])."""
    vl, ql, w, dlogits = (np.asarray(a, F64) for a in (vl, ql, w, dlogits))
    B, N, L = vl.shape
    diff = (vl[:, :, None, :] - vl[:, None, :, :])                             # [B,i,j,L]
    m = np.ones((B, N, N, L), F64) if mask is None else np.asarray(mask, F64).reshape(B, N, N, L)
    vq = (diff * ql[:, None, None, :] * m).reshape(B, N, N * L)
    dw = np.einsum("big,bik->gk", dlogits, vq)
    dbias = dlogits.sum(axis=(0, 1))
    dvq = (dlogits @ w).reshape(B, N, N, L) * m                                # dL/d(diff*ql) masked
    dql = (dvq * diff).sum(axis=(1, 2))
    dd = dvq * ql[:, None, None, :]                                            # dL/ddiff [B,i,j,L]
    dvl = dd.sum(axis=2) - dd.sum(axis=1)
    return dvl, dql, dw, dbias


# --------------------------------------------------------------------------------------
# K5  dropout + linear + bias + activation   (config/CoR2.py:72-88 MyConv1d k=1; :106-121 MyLinear)
# --------------------------------------------------------------------------------------
def linear_act_fwd(x, w, bias, act=None, mask=None):
    """x [M,K], w [N,K], bias [N] or None, mask [M,K] (keep/(1-p)) or None -> y [M,N] = act((x*mask) w^T + b)."""
    x, w = np.asarray(x, F64), np.asarray(w, F64)
    xd = x if mask is None else x * np.asarray(mask, F64)
    z = xd @ w.T
    if bias is not None:
        z = z + np.asarray(bias, F64)
    return np.maximum(z, 0.0) if act == "relu" else z


def linear_act_bwd(x, w, y, gy, act=None, mask=None):
    """-> (dx [M,K], dw [N,K], db [N]); y is the forward output (its sign is the relu mask)."""
    x, w, y, gy = (np.asarray(a, F64) for a in (x, w, y, gy))
    m = 1.0 if mask is None else np.asarray(mask, F64)
    gz = gy * (y > 0) if act == "relu" else gy
    dx = (gz @ w) * m
    dw = gz.T @ (x * m)
    db = gz.sum(axis=0)
    return dx, dw, db
