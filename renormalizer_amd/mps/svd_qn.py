"""Quantum-number blocked decompositions on the device (counterpart of renormalizer/mps/svd_qn.py).

The integer bookkeeping (which rows / columns form a symmetry block, the qn lists of the new
bond) is done here on the host exactly like the reference (:99-240, 305-317); the floating
point work per block runs in the engine (Householder QR/RQ, one-sided Jacobi SVD)."""
import ctypes as C

import numpy as np

from ..engine import DeviceTensor, get_engine


def add_outer(a, b):
    """svd_qn.py:305-313"""
    a = np.asarray(a)
    b = np.asarray(b)
    assert a.shape[-1] == b.shape[-1]
    return a.reshape(a.shape[:-1] + (1,) * (b.ndim - 1) + a.shape[-1:]) + b


def get_qn_mask(qnmat, qntot):
    """svd_qn.py:316-317"""
    return np.all(np.asarray(qnmat) == np.asarray(qntot), axis=-1)


def qn_blocks(qnbigl, qnbigr, qntot):
    """[(nl, nr, row indices, column indices)] for every left qn ``nl`` that has partner columns
    with ``qntot - nl`` (svd_qn.py:177-182); blocks in lexicographic order of nl."""
    qntot = np.asarray(qntot)
    q = len(qntot)
    lq = np.ascontiguousarray(np.asarray(qnbigl).reshape(-1, q))
    rq = np.ascontiguousarray(np.asarray(qnbigr).reshape(-1, q))
    out = []
    uniq, inv = np.unique(lq, axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    for k, nl in enumerate(uniq):
        nr = qntot - nl
        rset = np.nonzero(np.all(rq == nr, axis=1))[0]
        if len(rset) == 0:
            continue
        lset = np.nonzero(inv == k)[0]
        out.append((nl.astype(int), nr.astype(int), lset.astype(np.int64), rset.astype(np.int64)))
    return out


_PLAN_CACHE = {}
_PLAN_CACHE_MAX = 4096


def block_plan(qnbigl, qnbigr, qntot):
    """Everything ``svd_qn`` derives from the quantum numbers alone: concatenated row / column index lists of the
    blocks, their offsets, the number of singular values per block and the qn labels of the new bond.  A fixed-bond
    sweep meets the same (qnbigl, qnbigr, qntot) at a site again every half-sweep, so the plan is memoised on the
    bytes of the three integer arrays (the host is otherwise busy with np.unique / argsort while the GPU idles)."""
    qntot = np.ascontiguousarray(np.asarray(qntot, dtype=np.int64))
    q = len(qntot)
    lq = np.ascontiguousarray(np.asarray(qnbigl, dtype=np.int64).reshape(-1, q))
    rq = np.ascontiguousarray(np.asarray(qnbigr, dtype=np.int64).reshape(-1, q))
    key = (lq.tobytes(), rq.tobytes(), qntot.tobytes())
    plan = _PLAN_CACHE.get(key)
    if plan is not None:
        return plan
    blocks = qn_blocks(lq, rq, qntot)
    if len(blocks) == 0:
        raise ValueError("Invalid quantum number")
    dims = [min(len(b[2]), len(b[3])) for b in blocks]
    new_qnl, new_qnr = [], []
    for b, k in zip(blocks, dims):
        new_qnl += [b[0].tolist()] * k
        new_qnr += [b[1].tolist()] * k
    plan = dict(blocks=blocks, dims=dims, K=int(sum(dims)), new_qnl=new_qnl, new_qnr=new_qnr,
                rows=np.concatenate([b[2] for b in blocks]), cols=np.concatenate([b[3] for b in blocks]),
                roff=np.cumsum([0] + [len(b[2]) for b in blocks]).astype(np.int64),
                coff=np.cumsum([0] + [len(b[3]) for b in blocks]).astype(np.int64))
    if len(_PLAN_CACHE) >= _PLAN_CACHE_MAX:
        _PLAN_CACHE.clear()
    _PLAN_CACHE[key] = plan
    return plan


class TransposedView:
    """What the reference returns as ``v`` (ncol x K); the engine produces v.T directly."""

    def __init__(self, vt: DeviceTensor):
        self.T = vt
        self.shape = (vt.shape[1], vt.shape[0])

    def to_host(self):
        return self.T.to_host().T


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def svd_qn(coef_array, qnbigl, qnbigr, qntot, QR=False, system=None, full_matrices=True, opt_full_matrices=True,
           plan=None, householder=False):
    """Block decomposition of the centre tensor.  Returns ``(u, new_qnl, v, new_qnr)`` for QR and
    ``(u, su, new_qnl, v, sv, new_qnr)`` for SVD with ``coef == u @ diag(s) @ v.T`` (``v.T`` is
    available as ``v.T``, a device tensor).  ``plan``: the result of ``block_plan`` for these quantum numbers when
    the caller has it already (the sweeps prepare it while the GPU is still busy with the preceding solve).
    ``householder`` (QR only): this decomposition by the Householder kernels (bit 1 of ``system_is_R``)."""
    eng = get_engine()
    coef = eng.asdevice(coef_array)
    qntot = np.asarray(qntot)
    nrow = int(np.prod(np.asarray(qnbigl).shape[:-1]))
    ncol = int(np.prod(np.asarray(qnbigr).shape[:-1]))
    if coef.size != nrow * ncol:
        raise ValueError(f"coefficient array {coef.shape} does not match quantum numbers ({nrow}x{ncol})")
    if plan is None:
        plan = block_plan(qnbigl, qnbigr, qntot)
    blocks, dims, K = plan["blocks"], plan["dims"], plan["K"]
    rows, cols, roff, coff = plan["rows"], plan["cols"], plan["roff"], plan["coff"]
    new_qnl, new_qnr = list(plan["new_qnl"]), list(plan["new_qnr"])
    if QR:
        if system not in ("L", "R"):
            raise ValueError("system must be 'L' or 'R' for QR")
        u = eng.empty((nrow, K), coef.dtype)
        vt = eng.empty((K, ncol), coef.dtype)
        eng._check(eng.lib.mpse_block_qr(eng.ctx, coef.code, coef.ptr, nrow, ncol, len(blocks), _p64(rows), _p64(roff),
                                         _p64(cols), _p64(coff), int(system == "R") | (2 if householder else 0), u.ptr,
                                         vt.ptr, K))
        if not full_matrices:
            return u, new_qnl, TransposedView(vt), new_qnr
        # full_matrices (svd_qn.py:194-197, ``scipy.linalg.qr / rq(mode="full")`` per block; no caller in the reference's
        # sweeps): the isometry of every block is completed to a square unitary - the extra vectors span the orthogonal
        # complement of the block's range, which is also what the full SVD of the block appends, so they are taken from
        # ``mpse_block_svd_full`` (Householder completion on the device) - and the triangular factor gets zero rows /
        # columns to match, listed after the economic part with the block's quantum number like the reference's
        # ``blockappend``.  (For RQ of a block with fewer rows than columns LAPACK's full mode puts the triangle at the
        # right edge of R and the reference then files its leading columns under "non-zero": here the economic factors
        # always come first.)
        iso_rows = system == "L"
        extra = np.zeros(len(blocks), dtype=np.int64)
        qn_iso0, qn_tri0 = [], []
        for ib, (b, k) in enumerate(zip(blocks, dims)):
            m, n = len(b[2]), len(b[3])
            ex = (m - k) if iso_rows else (n - k)
            if ex > 0:
                extra[ib] = ex
                qn_iso0 += [(b[0] if iso_rows else b[1]).tolist()] * ex
                qn_tri0 += [(b[1] if iso_rows else b[0]).tolist()] * ex
        nex = int(extra.sum())
        if nex == 0:
            return u, new_qnl, TransposedView(vt), new_qnr
        # (an isometry side that needs completing IS the taller side of its block: mpse_block_svd_full appends exactly
        # ``extra[ib]`` completion vectors to that side and none to the other)
        s = np.zeros(K)
        su = eng.empty((nrow, K + (nex if iso_rows else 0)), coef.dtype)
        svt = eng.empty((K + (0 if iso_rows else nex), ncol), coef.dtype)
        eng._check(eng.lib.mpse_block_svd_full(eng.ctx, coef.code, coef.ptr, nrow, ncol, len(blocks), _p64(rows),
                                               _p64(roff), _p64(cols), _p64(coff), _p64(extra), su.ptr, su.shape[1],
                                               svt.ptr, svt.shape[0], s.ctypes.data_as(C.POINTER(C.c_double)), K))
        if iso_rows:
            uf = eng.empty((nrow, K + nex), coef.dtype)
            eng.copy_block(uf, 0, 0, u)
            eng.copy_sub(uf, 0, K, su, 0, K, nrow, nex)
            vtf = eng.zeros((K + nex, ncol), coef.dtype)
            eng.copy_block(vtf, 0, 0, vt)
            return uf, new_qnl + qn_iso0, TransposedView(vtf), new_qnr + qn_tri0
        vtf = eng.empty((K + nex, ncol), coef.dtype)
        eng.copy_block(vtf, 0, 0, vt)
        eng.copy_block(vtf, K, 0, svt.row_block(K, K + nex))
        uf = eng.zeros((nrow, K + nex), coef.dtype)
        eng.copy_block(uf, 0, 0, u)
        return uf, new_qnl + qn_tri0, TransposedView(vtf), new_qnr + qn_iso0
    s = np.zeros(K)
    sp = s.ctypes.data_as(C.POINTER(C.c_double))
    if full_matrices:
        # null-space completion of the taller side of every block (svd_qn.py:12-49, 65-86): all of it while the
        # aspect ratio is below 3, otherwise as many vectors as the block has singular values
        extra = np.zeros(len(blocks), dtype=np.int64)
        qnl0, qnr0 = [], []
        for ib, (b, k) in enumerate(zip(blocks, dims)):
            m, n = len(b[2]), len(b[3])
            tall = max(m, n)
            ex = tall - k
            if opt_full_matrices and not (1 / 3 < m / n < 3):
                ex = min(ex, k)
            extra[ib] = ex
            if m >= n:
                qnl0 += [b[0].tolist()] * ex
            else:
                qnr0 += [b[1].tolist()] * ex
        KU, KV = K + len(qnl0), K + len(qnr0)
        u = eng.empty((nrow, KU), coef.dtype)
        vt = eng.empty((KV, ncol), coef.dtype)
        eng._check(eng.lib.mpse_block_svd_full(eng.ctx, coef.code, coef.ptr, nrow, ncol, len(blocks), _p64(rows),
                                               _p64(roff), _p64(cols), _p64(coff), _p64(extra), u.ptr, KU, vt.ptr, KV,
                                               sp, K))
        su = np.concatenate([s, np.zeros(KU - K)])
        sv = np.concatenate([s, np.zeros(KV - K)])
        return u, su, new_qnl + qnl0, TransposedView(vt), sv, new_qnr + qnr0
    u = eng.empty((nrow, K), coef.dtype)
    vt = eng.empty((K, ncol), coef.dtype)
    eng._check(eng.lib.mpse_block_svd(eng.ctx, coef.code, coef.ptr, nrow, ncol, len(blocks), _p64(rows), _p64(roff),
                                      _p64(cols), _p64(coff), u.ptr, vt.ptr, sp, K))
    # economic SVD: globally sort by singular value, descending (svd_qn.py:231-239)
    order = np.argsort(s)[::-1].astype(np.int64)
    if not np.array_equal(order, np.arange(K)):
        u2 = eng.empty((nrow, K), coef.dtype)
        vt2 = eng.empty((K, ncol), coef.dtype)
        eng._check(eng.lib.mpse_gather_cols(eng.ctx, u.code, u2.ptr, u.ptr, nrow, K, _p64(order), None, K))
        eng._check(eng.lib.mpse_gather_rows(eng.ctx, vt.code, vt2.ptr, vt.ptr, ncol, _p64(order), None, K))
        u, vt = u2, vt2
        s = s[order]
        new_qnl = np.array(new_qnl)[order].tolist()
        new_qnr = np.array(new_qnr)[order].tolist()
    return u, s, new_qnl, TransposedView(vt), s, new_qnr


def eigh_qn(dm, qnbigl, qnbigr, qntot, system):
    """Diagonalisation of a reduced density matrix by quantum-number block (mps/svd_qn.py:243-302, used by the
    multi-state update mps/mp.py:800): returns ``(u, s, new_qn)`` with s = sqrt(eigenvalue) and the eigenvectors as the
    columns of ``u``, blocks in lexicographic order of their quantum number.  A density matrix is Hermitian positive
    semi-definite, so its eigen-decomposition is its singular value decomposition: the blocks go through the engine's
    batched one-sided Jacobi SVD (``mpse_block_svd``; eigenvalues come out descending inside a block where LAPACK's
    eigh returns them ascending - an ordering the callers do not rely on, they select by ``s``)."""
    assert system in ["L", "R"]
    eng = get_engine()
    qnbig, comp = (qnbigl, qnbigr) if system == "L" else (qnbigr, qnbigl)
    qntot = np.asarray(qntot)
    q = len(qntot)
    local = np.ascontiguousarray(np.asarray(qnbig).reshape(-1, q))
    comp = np.asarray(comp).reshape(-1, q)
    n = len(local)
    d = eng.asdevice(dm)
    if d.size != n * n:
        raise ValueError(f"density matrix {d.shape} does not match the quantum numbers ({n} states)")
    uniq, inv = np.unique(local, axis=0, return_inverse=True)
    inv = np.asarray(inv).reshape(-1)
    sets, new_qn = [], []
    for k, nl in enumerate(uniq):
        if not np.any(np.all(comp == qntot - nl, axis=1)):
            continue
        lset = np.nonzero(inv == k)[0].astype(np.int64)
        sets.append(lset)
        new_qn += [nl.astype(int).tolist()] * len(lset)
    if not sets:
        raise ValueError("Invalid quantum number")
    idx = np.ascontiguousarray(np.concatenate(sets))
    off = np.cumsum([0] + [len(x) for x in sets]).astype(np.int64)
    K = int(off[-1])
    u = eng.empty((n, K), d.dtype)
    vt = eng.empty((K, n), d.dtype)
    s2 = np.zeros(K)
    eng._check(eng.lib.mpse_block_svd(eng.ctx, d.code, d.ptr, n, n, len(sets), _p64(idx), _p64(off), _p64(idx), _p64(off),
                                      u.ptr, vt.ptr, s2.ctypes.data_as(C.POINTER(C.c_double)), K))
    return u, np.sqrt(np.maximum(s2, 0.0)), new_qn
