"""Module-level tensor helpers of the drop-in boundary (the names consumers import from
``renormalizer.mps.matrix`` - renormalizer/mps/matrix.py:13-322): ``Matrix``, ``asnumpy``, ``asxp``, ``tensordot``,
``multi_tensor_contract`` and the small constructors, here over ``DeviceTensor`` handles of the HIP engine.

``tensordot`` never materialises a transposed copy (the reference's ``xp.tensordot`` does, before every ?gemm): the
axes of both operands are folded into the two-level strided indices of ``mpse_gemm``; what does not fit two levels is
peeled into a host loop over launches (offsets into the same buffers)."""
import itertools
from typing import List

import numpy as np

from ..engine import DeviceTensor, get_engine, idx1, idx2


def Matrix(array, dtype=None) -> DeviceTensor:
    """Device-resident tensor from anything array-like (the reference's ``Matrix(array, dtype)``, matrix.py:15-29;
    the shape helpers ``pdim`` / ``bond_dim`` / ``l_combine`` ... live on ``DeviceTensor``)."""
    return get_engine().asdevice(array, dtype)


def asnumpy(array):
    """Host copy (None passes through; lists become arrays) - matrix.py:298-311."""
    if array is None:
        return None
    if isinstance(array, DeviceTensor):
        return array.to_host()
    if isinstance(array, (list, tuple)):
        return np.array([asnumpy(a) if isinstance(a, DeviceTensor) else a for a in array])
    return np.asarray(array)


def asxp(array):
    """Device handle (None passes through) - matrix.py:314-322."""
    if array is None:
        return None
    return get_engine().asdevice(array)


def zeros(shape, dtype=None):
    return get_engine().zeros(shape, dtype or np.float64)


def ones(shape, dtype=None):
    return get_engine().ones(shape, dtype or np.float64)


def eye(N, M=None, dtype=None):
    return get_engine().asdevice(np.eye(N, M, dtype=dtype or np.float64))


def allclose(a, b, rtol=1.0e-5, atol=1.0e-8):
    return bool(np.allclose(asnumpy(a), asnumpy(b), rtol=rtol, atol=atol))


def _levels(shape, strides, axes):
    """(extent, stride) levels, outer -> inner, of the index that runs over ``axes`` in that order; neighbouring axes
    whose strides are contiguous are merged, extent-1 axes dropped."""
    lv = []
    for ax in axes:
        e, st = shape[ax], strides[ax]
        if e == 1:
            continue
        if lv and lv[-1][1] == e * st:
            lv[-1] = (lv[-1][0] * e, st)
        else:
            lv.append((e, st))
    return lv


def _index(levels):
    if not levels:
        return idx1(1, 1)
    if len(levels) == 1:
        return idx1(levels[0][0], levels[0][1])
    (h, sh), (lo, sl) = levels
    return idx2(h, lo, sh, sl)


def tensordot(a, b, axes):
    """``numpy.tensordot`` semantics on device tensors (matrix.py:210-211): the result's axes are the free axes of
    ``a`` followed by those of ``b``.  One strided FP64-MFMA launch when every index folds into two stride levels;
    outer levels beyond that become a host loop over launches."""
    eng = get_engine()
    a, b = eng.asdevice(a), eng.asdevice(b)
    if isinstance(axes, int):
        axes = (list(range(a.ndim - axes, a.ndim)), list(range(axes)))
    ax_a, ax_b = ([int(x) % t.ndim for x in (ax if isinstance(ax, (list, tuple)) else [ax])]
                  for ax, t in zip(axes, (a, b)))
    if len(ax_a) != len(ax_b) or any(a.shape[i] != b.shape[j] for i, j in zip(ax_a, ax_b)):
        raise ValueError(f"tensordot: shape mismatch {a.shape} {b.shape} over {axes}")
    if a.is_complex != b.is_complex:
        a, b = a.to_complex(), b.to_complex()
    free_a = [i for i in range(a.ndim) if i not in ax_a]
    free_b = [i for i in range(b.ndim) if i not in ax_b]
    oshape = tuple(a.shape[i] for i in free_a) + tuple(b.shape[i] for i in free_b)
    out = eng.empty(oshape if oshape else (1,), a.dtype)

    def cstrides(shape):
        st, acc = [], 1
        for e in reversed(shape):
            st.append(acc)
            acc *= e
        return st[::-1]

    sa, sb, so = cstrides(a.shape), cstrides(b.shape), cstrides(oshape)
    M = int(np.prod([a.shape[i] for i in free_a])) if free_a else 1
    N = int(np.prod([b.shape[i] for i in free_b])) if free_b else 1
    if M == 0 or N == 0:
        return out.reshape(oshape)
    # contracted axes in the order that leaves the fewest stride levels on both operands
    best = None
    for perm in itertools.permutations(range(len(ax_a))) if len(ax_a) <= 4 else [tuple(range(len(ax_a)))]:
        ka = _levels(a.shape, sa, [ax_a[p] for p in perm])
        kb = _levels(b.shape, sb, [ax_b[p] for p in perm])
        cost = max(len(ka), len(kb))
        if best is None or cost < best[0]:
            best = (cost, perm)
    perm = best[1]
    k_axes_a, k_axes_b = [ax_a[p] for p in perm], [ax_b[p] for p in perm]

    def split(shape, strides, axes_):
        """axes -> (loop axes (outermost), index axes) such that the index has at most two levels"""
        loops = list(axes_)
        keep = []
        while loops and len(_levels(shape, strides, [loops[-1]] + keep)) <= 2:
            keep.insert(0, loops.pop())
        return loops, keep

    # K: the same leading contracted axes are looped on both operands
    kl_a, kk_a = split(a.shape, sa, k_axes_a)
    kl_b, kk_b = split(b.shape, sb, k_axes_b)
    nloop_k = max(len(kl_a), len(kl_b))
    kl_a, kk_a = k_axes_a[:nloop_k], k_axes_a[nloop_k:]
    kl_b, kk_b = k_axes_b[:nloop_k], k_axes_b[nloop_k:]
    ml, mk = split(a.shape, sa, free_a)
    nl, nk = split(b.shape, sb, free_b)
    m_a, k_a = _index(_levels(a.shape, sa, mk)), _index(_levels(a.shape, sa, kk_a))
    k_b, n_b = _index(_levels(b.shape, sb, kk_b)), _index(_levels(b.shape, sb, nk))
    # the result is contiguous: its kept M axes and kept N axes are single levels
    o_m = _index(_levels(oshape, so, list(range(len(ml), len(free_a)))))
    o_n = _index(_levels(oshape, so, list(range(len(free_a) + len(nl), len(oshape)))))
    for mi in itertools.product(*[range(a.shape[i]) for i in ml]):
        off_am = sum(v * sa[i] for v, i in zip(mi, ml))
        off_om = sum(v * so[j] for v, j in zip(mi, range(len(ml))))
        for ni in itertools.product(*[range(b.shape[i]) for i in nl]):
            off_bn = sum(v * sb[i] for v, i in zip(ni, nl))
            off_on = sum(v * so[len(free_a) + j] for v, j in zip(ni, range(len(nl))))
            first = True
            for ki in itertools.product(*[range(a.shape[i]) for i in kl_a]):
                off_ak = sum(v * sa[i] for v, i in zip(ki, kl_a))
                off_bk = sum(v * sb[i] for v, i in zip(ki, kl_b))
                eng.gemm(a.ravel().shifted(off_am + off_ak), b.ravel().shifted(off_bn + off_bk),
                         out.ravel().shifted(off_om + off_on), m_a, k_a, k_b, n_b, o_m, o_n,
                         beta=0.0 if first else 1.0)
                first = False
    return out.reshape(oshape)


def _pair_contract(left, in_left, right, in_right, removed):
    lp = [in_left.find(s) for s in removed]
    rp = [in_right.find(s) for s in removed]
    return tensordot(left, right, axes=(lp, rp))


def multi_tensor_contract(path, *operands: List):
    """Chain of pairwise contractions (matrix.py:243-280): every step names two operand positions and an einsum-like
    string ``"fdla, abc -> fdlbc"`` whose result order is (free indices of the first, free indices of the second); the
    two operands leave the list, the result joins its end."""
    ops = list(operands)
    for positions, spec in path:
        ins, res = spec.split("->")
        lhs, rhs = (x.replace(" ", "") for x in ins.split(","))
        removed = sorted(set(lhs + rhs) - set(res.replace(" ", "")))
        tmp = _pair_contract(ops[positions[0]], lhs, ops[positions[1]], rhs, removed)
        for x in sorted(positions, reverse=True):
            del ops[x]
        ops.append(tmp)
    return ops[0]
