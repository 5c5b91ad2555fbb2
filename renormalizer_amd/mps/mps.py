"""Matrix product states resident in HBM and their sweep algorithms.

Kept API surface of renormalizer/mps/mp.py (MatrixProduct) and renormalizer/mps/mps.py (Mps):
constructors, qn bookkeeping, ``expectation(s)``, ``evolve`` (TDVP-PS), ``canonicalise`` /
``compress``.  The Python here is sweep *control* only (order of operations copied from the
reference drivers, cited per method); every tensor is a ``DeviceTensor`` and every flop runs in
libmpsengine.so.  Integer quantum-number bookkeeping stays on the host in NumPy int arrays."""
import ctypes
import logging
import os
import threading
from typing import Dict, List

import numpy as np

from ..engine import DeviceTensor, EngineError, get_engine
from ..lib.krylov import expm_krylov
from ..model import Model, Op, OpSum
from ..utils import CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod, OptimizeConfig
from . import svd_qn
from .hop_expr import centre_tile_mask, hop_expr
from . import lib as _lib
from .lib import Environ, contract_one_site
from .mpo import Mpo
from .svd_qn import add_outer, get_qn_mask

logger = logging.getLogger("renormalizer_amd")


def _vn_entropy(p):
    """-sum p ln p of normalised non-negative weights (utils/utils.py:41-48)"""
    p = np.asarray(p, dtype=float)
    assert np.allclose(p[p < 0], 0)
    p = p / p.sum()
    p = p[p > 0]
    return float(-(p * np.log(p)).sum())


def _is_identity_site(w):
    """MPO site tensor (w_l, d, d, w_r) that is exactly the identity pass-through."""
    w = np.asarray(w)
    return w.shape[0] == 1 and w.shape[3] == 1 and np.array_equal(w[0, :, :, 0], np.eye(w.shape[1]))


class Mps:
    def __init__(self):
        self._mp: List[DeviceTensor] = []
        self.dtype = np.dtype(np.float64)
        self.model: Model = None
        self.compress_config: CompressConfig = CompressConfig()
        self.evolve_config: EvolveConfig = EvolveConfig()
        self.optimize_config: OptimizeConfig = OptimizeConfig()
        self.qn: List[np.ndarray] = []
        self.qnidx: int = None
        self.qntot: np.ndarray = None
        self.to_right: bool = None
        self.coeff = 1.0

    # ------------------------------------------------------------------ constructors
    @classmethod
    def from_arrays(cls, model, arrays, qn, qnidx, qntot, to_right, coeff=1.0):
        """Upload host site tensors (D_l, d, D_r) and their quantum numbers."""
        eng = get_engine()
        mps = cls()
        mps.model = model
        cplx = any(np.iscomplexobj(a) for a in arrays)
        mps.dtype = np.dtype(np.complex128 if cplx else np.float64)
        mps._mp = [eng.asdevice(np.asarray(a), mps.dtype) for a in arrays]
        q = len(np.atleast_1d(qntot))
        mps.qn = [np.asarray(x, dtype=int).reshape(-1, q) for x in qn]
        mps.qnidx = int(qnidx)
        mps.qntot = np.asarray(qntot, dtype=int).reshape(q)
        mps.to_right = bool(to_right)
        mps.coeff = coeff
        return mps

    @classmethod
    def random(cls, model, qntot, m_max, percent=1.0, rng=None) -> "Mps":
        """Random quantum-number-conserving MPS, construction of mps/mps.py:119-185: per site a
        random orthogonal basis inside every symmetry block of the (left bond x physical) space,
        ``m_max`` of them kept (``percent`` of the slots shared equally between blocks), the
        last site filled with masked random numbers and normalised.  Host-side state
        preparation (runs once, outside the sweep); the result is uploaded to the device."""
        from .basis_select import select_basis_indices
        if rng is None:
            rng = np.random.default_rng()
        if isinstance(qntot, (int, np.integer)):
            qntot = np.array([qntot])
        qntot = np.asarray(qntot, dtype=int)
        q = len(qntot)
        assert q == model.qn_size
        qn = [np.zeros((1, q), dtype=int)]
        dims = [1]
        arrays = []
        for imps in range(model.nsite - 1):
            sig = np.array(model.basis[imps].sigmaqn)
            qnbig = add_outer(qn[imps], sig).reshape(-1, q)
            u_set, s_set, qnset = [], [], []
            for blk in sorted(set(map(tuple, qnbig.tolist()))):
                if np.all(qntot < np.array(blk)):
                    continue
                idx = np.nonzero(np.all(qnbig == np.array(blk), axis=1))[0]
                a = rng.random((len(idx), len(idx))) - 0.5
                s, u = np.linalg.eigh(a + a.T)
                full = np.zeros((len(qnbig), len(idx)))
                full[idx, :] = u
                u_set.append(full)
                s_set.append(s)
                qnset += [blk] * len(idx)
            u_set = np.concatenate(u_set, axis=1)
            s_set = np.concatenate(s_set)
            mmax = m_max[imps + 1] if isinstance(m_max, (list, tuple, np.ndarray)) else m_max
            keep = select_basis_indices(s_set, qnset, mmax, percent)
            dims.append(len(keep))
            arrays.append(u_set[:, keep].reshape(dims[imps], -1, dims[imps + 1]))
            qn.append(np.array([qnset[i] for i in keep], dtype=int).reshape(len(keep), q))
        qn.append(np.zeros((1, q), dtype=int))
        last = rng.random((dims[-1], model.pbond_list[-1], 1)) - 0.5
        mask = get_qn_mask(add_outer(qn[-2], np.array(model.basis[-1].sigmaqn)), qntot)
        last[~mask.reshape(last.shape[:2])] = 0
        last /= np.linalg.norm(last)
        arrays.append(last)
        return cls.from_arrays(model, arrays, qn, model.nsite - 1, qntot, False)

    @classmethod
    def from_dense(cls, model, wfn: np.ndarray) -> "Mps":
        """Exact MPS of a dense wavefunction by successive QR (mps/mps.py:389-406: a debugging helper; host-side
        state preparation like ``random``).  The quantum numbers are left empty (all zero), as in the reference."""
        wfn = np.asarray(wfn)
        dims = [b.nbas for b in model.basis]
        res = wfn.reshape([1] + dims + [1])
        arrays = []
        for _ in range(len(dims) - 1):
            q, r = np.linalg.qr(res.reshape(res.shape[0] * res.shape[1], -1))
            arrays.append(q.reshape(res.shape[0], res.shape[1], q.shape[1]))
            res = r.reshape([r.shape[0]] + list(res.shape[2:]))
        arrays.append(res)
        qsz = model.qn_size
        qn = [np.zeros((a.shape[0], qsz), dtype=int) for a in arrays] + [np.zeros((1, qsz), dtype=int)]
        return cls.from_arrays(model, arrays, qn, len(arrays) - 1, np.zeros(qsz, dtype=int), False)

    def todense(self) -> np.ndarray:
        """The full wavefunction (mp.py:996-1007), site by site on the device; exponential in the number of sites."""
        eng = get_engine()
        if int(np.prod([float(d) for d in self.pbond_dims])) > 2 ** 26:
            raise ValueError("todense: the dense wavefunction would not fit")
        t = self[0].reshape(-1, self[0].shape[-1])
        for ms in self._mp[1:]:
            t = eng.matmul(t, ms.reshape(ms.shape[0], -1)).reshape(-1, ms.shape[-1])
        return t.to_host().reshape(list(self.pbond_dims)) * self.coeff

    @classmethod
    def hartree_product_state(cls, model, condition: Dict = None, qn_idx: int = None):
        """mps/mps.py:187-262"""
        condition = dict(condition or {})
        idx = [model.dof_to_siteidx[k] for k in condition]
        assert len(idx) == len(set(idx))
        condition = {model.dof_to_siteidx[k]: v for k, v in condition.items()}
        q = model.qn_size
        arrays, qn = [], [np.zeros((1, q), dtype=int)]
        for isite, b in enumerate(model.basis):
            ms = np.zeros((1, b.nbas, 1))
            st = condition.pop(isite, 0)
            if isinstance(st, (int, np.integer)):
                ms[0, st, 0] = 1.0
                sq = b.sigmaqn[st]
            else:
                ms[0, :, 0] = st
                allq = np.array(b.sigmaqn)[np.nonzero(st)]
                if not np.allclose(allq.std(axis=0), 0):
                    raise ValueError("Quantum numbers are mixed in the condition.")
                sq = allq[0]
            arrays.append(ms)
            qn.append(qn[-1] + np.asarray(sq).reshape(1, q))
        if condition:
            raise ValueError(f"Condition not completely used: {condition}")
        mps = cls.from_arrays(model, arrays, qn, model.nsite, qn[-1][0], False)
        mps.move_qnidx(model.nsite - 1 if qn_idx is None else qn_idx)
        return mps

    @classmethod
    def ground_state(cls, model, max_entangled: bool, normalize: bool = True, condition: Dict = None):
        """mps/mps.py:264-350 (phonons in |0> or maximally entangled, electrons empty)."""
        q = model.qn_size
        arrays = []
        for b in model.basis:
            ms = np.zeros((1, b.nbas, 1))
            if (b.is_phonon or b.is_spin) and max_entangled:
                ms[0, :, 0] = 1.0 / np.sqrt(b.nbas) if normalize else 1.0
            else:
                ms[0, 0, 0] = 1.0
            arrays.append(ms)
        qn = [np.zeros((1, q), dtype=int)] * (model.nsite + 1)
        return cls.from_arrays(model, arrays, qn, model.nsite - 1, np.zeros(q, dtype=int), False)

    # ------------------------------------------------------------------ container protocol
    def __len__(self):
        return len(self._mp)

    def __getitem__(self, i):
        return self._mp[i]

    def __setitem__(self, i, t):
        if not isinstance(t, DeviceTensor):
            t = get_engine().asdevice(np.ascontiguousarray(t))      # host arrays are uploaded (mps/mp.py:1220-1236)
        self._mp[i] = t

    def __iter__(self):
        return iter(self._mp)

    @property
    def site_num(self):
        return len(self._mp)

    @property
    def is_complex(self):
        return self.dtype == np.complex128

    @property
    def bond_dims(self):
        return [t.shape[0] for t in self._mp] + [self._mp[-1].shape[-1]]

    bond_list = vbond_dims = bond_dims

    @property
    def bond_dims_mean(self):
        return int(round(np.mean(self.bond_dims)))

    @property
    def pbond_dims(self):
        return self.model.pbond_list

    pbond_list = pbond_dims

    @property
    def bond_dims_exact(self):
        p = np.array(self.pbond_dims, dtype=float)
        if self.is_mpdm:
            p = p ** 2                      # mp.py:131-142: both physical legs count
        with np.errstate(over="ignore"):
            d1 = [1] + list(np.cumprod(p))
            d2 = ([1] + list(np.cumprod(p[::-1])))[::-1]
        return np.minimum(d1, d2)

    @property
    def nexciton(self):
        return self.qntot

    is_mps, is_mpo, is_mpdm = True, False, False

    def to_arrays(self):
        """Download all site tensors (debugging / checkpointing)."""
        return [t.to_host() for t in self._mp]

    # ------------------------------------------------------------------ checkpoint wire format
    def dump(self, fname):
        """Write the reference's npz protocol "0.4" (mps/mp.py:1085-1113, mps/mps.py:1795-1796): ``mt_i`` site
        arrays, ``qnidx``, ``qntot``, ``qn`` (object array) + ``subqn_i``, ``to_right``, ``coeff`` - files are
        interchangeable with ``renormalizer.Mps.load`` / ``dump``."""
        data = {"version": "0.4", "nsites": self.site_num}
        for i, t in enumerate(self._mp):
            data[f"mt_{i}"] = t.to_host()
        qn = np.empty(len(self.qn), object)
        qn[:] = [np.asarray(q) for q in self.qn]
        data.update(qnidx=self.qnidx, qntot=self.qntot, qn=qn, to_right=self.to_right, coeff=self.coeff)
        for i, q in enumerate(self.qn):
            data[f"subqn_{i}"] = np.asarray(q)
        np.savez(fname, **data)

    @classmethod
    def load(cls, model, fname: str):
        """Read dump files of every protocol version the reference reads (mps/mps.py:352-386): 0.3 / 0.4 carry
        ``to_right`` and ``coeff``; 0.2 carries ``to_right`` and the coefficient as the last entry of the (otherwise
        obsolete) time-dependent-Hartree part; 0.1 calls the direction ``left`` and has no coefficient."""
        z = np.load(fname, allow_pickle=True)
        version = str(z["version"])
        n = int(z["nsites"])
        arrays = [z[f"mt_{i}"] for i in range(n)]
        if f"subqn_{n}" in z.files:
            qn = [np.asarray(z[f"subqn_{i}"]).astype(int) for i in range(n + 1)]
        else:
            qn = [np.asarray(q).astype(int) for q in z["qn"]]
        if version == "0.1":
            logger.warning("Using old dump/load protocol. TD Hartree part will be lost")
            to_right, coeff = bool(z["left"]), 1
        elif version == "0.2":
            logger.warning("Using old dump/load protocol. TD Hartree part will be lost")
            to_right, coeff = bool(z["to_right"]), np.asarray(z["tdh_wfns"], dtype=object)[-1]
            coeff = complex(np.asarray(coeff).item(0)) if np.iscomplexobj(coeff) else float(np.asarray(coeff).item(0))
        elif version in ("0.3", "0.4"):
            to_right, coeff = bool(z["to_right"]), z["coeff"].item(0)
        else:
            raise ValueError(f"Unknown dump version: {version}")
        qn = [q.reshape(len(q), -1) for q in qn]            # (protocols before 0.4 stored one number per state)
        return cls.from_arrays(model, arrays, qn, int(z["qnidx"]), np.asarray(z["qntot"]).astype(int).reshape(-1),
                               to_right, coeff)

    def _get_sigmaqn(self, idx):
        return np.array(self.model.basis[idx].sigmaqn)

    # ------------------------------------------------------------------ copies
    def metacopy(self):
        new = self.__class__.__new__(self.__class__)
        new.__dict__ = self.__dict__.copy()
        new._mp = [None] * len(self._mp)
        new.qn = [q.copy() for q in self.qn]
        new.qntot = self.qntot.copy()
        new.compress_config = self.compress_config.copy()
        new.evolve_config = self.evolve_config.copy()
        new.optimize_config = self.optimize_config.copy()
        notes = self.__dict__.get("_qr_notes")
        if notes is not None:
            new._qr_notes = notes.copy()          # part of the state: a copy evolves with its own history of noted sites
        return new

    def copy(self):
        """Site buffers are immutable by convention, so a copy shares them (copy-on-write)."""
        new = self.metacopy()
        new._mp = list(self._mp)
        return new

    def conj(self):
        new = self.metacopy()
        new._mp = [t.conj() if t.is_complex else t for t in self._mp]
        new.coeff = np.conjugate(self.coeff)          # mps/mps.py:415-418
        return new

    def to_complex(self, inplace=False):
        """mps/mp.py:996-1007.  Site buffers are immutable by convention (every update installs a
        new handle), so an already complex MPS shares its tensors with the result."""
        new = self if inplace else self.metacopy()
        new.dtype = np.dtype(np.complex128)
        new._mp = [t.to_complex() for t in self._mp]
        return new

    # ------------------------------------------------------------------ qn bookkeeping (integer, host)
    def move_qnidx(self, dstidx: int):
        """mps/mp.py:159-172"""
        for idx in range(self.qnidx + 1, self.site_num + 1):
            self.qn[idx] = self.qntot - self.qn[idx]
        for idx in range(self.site_num, dstidx, -1):
            self.qn[idx] = self.qntot - self.qn[idx]
        self.qnidx = dstidx

    def iter_idx_list(self, full: bool, stop_idx: int = None):
        """mps/mp.py:230-243"""
        if self.to_right:
            last = stop_idx if stop_idx is not None else (self.site_num if full else self.site_num - 1)
            return range(self.qnidx, last)
        last = stop_idx if stop_idx is not None else (-1 if full else 0)
        return range(self.qnidx, last, -1)

    def _switch_direction(self):
        """mps/mp.py:297-306"""
        assert self.to_right is not None
        if self.to_right:
            self.qnidx, self.to_right = self.site_num - 1, False
        else:
            self.qnidx, self.to_right = 0, True

    def _get_big_qn(self, cidx: List[int], swap=False, need_mat=True):
        """mps/mp.py:308-352.  ``need_mat=False`` skips the full outer sum (the DMRG mask); QR / SVD only need
        the row and column quantum numbers."""
        cidx = sorted(cidx)
        assert len(cidx) in (1, 2) and self.qnidx in cidx
        sigmaqn = [self._get_sigmaqn(i) for i in cidx]
        if swap:
            sigmaqn = sigmaqn[::-1]
        qnl = np.array(self.qn[cidx[0]])
        qnr = np.array(self.qn[cidx[-1] + 1])
        if len(cidx) == 1:
            if self.to_right:
                qnbigl, qnbigr = add_outer(qnl, sigmaqn[0]), qnr
            else:
                qnbigl, qnbigr = qnl, add_outer(sigmaqn[0], qnr)
        else:
            qnbigl, qnbigr = add_outer(qnl, sigmaqn[0]), add_outer(sigmaqn[1], qnr)
        return qnbigl, qnbigr, (add_outer(qnbigl, qnbigr) if need_mat else None)

    # ------------------------------------------------------------------ scalar products and norms
    def dot(self, other: "Mps", self_is_conj=True) -> complex:
        """<self|other> with ``self`` holding the already-conjugated bra (mps/mp.py:933-956);
        ``self_is_conj=False`` conjugates on the fly inside the contraction instead."""
        eng = get_engine()
        assert len(self) == len(other)
        e = eng.ones((1, 1), np.float64)
        for b, k in zip(self._mp, other._mp):
            t = eng.matmul(e, k.reshape(k.shape[0], -1))                      # (Db, d*Dr_k)
            t = t.reshape(-1, k.shape[-1])                                     # (Db*d, Dr_k)
            e = eng.matmul(b.reshape(-1, b.shape[-1]), t, trans_a=True, conj_a=not self_is_conj)
        return complex(e.to_host()[0, 0])

    @property
    def mp_norm(self) -> float:
        """mps/mp.py:354-372"""
        res = self.dot(self, self_is_conj=False).real
        if res < 0:
            assert abs(res) < 1e-8
            res = 0.0
        return float(np.sqrt(res))

    @property
    def norm(self):
        return float(np.linalg.norm(self.coeff) * self.mp_norm)

    def scale(self, val, inplace=False):
        """mps/mp.py:984-994: multiplies the qn-centre site."""
        new = self if inplace else self.copy()
        val = complex(val)
        if val.imag != 0:
            new.to_complex(inplace=True)
        else:
            val = val.real
        new._mp[new.qnidx] = new._mp[new.qnidx].copy().scale_(val)
        return new

    def normalize(self, kind):
        """mps/mps.py:2025-2059"""
        nrm = self.mp_norm
        if kind == "mps_only":
            new_coeff = self.coeff
        elif kind == "mps_and_coeff":
            new_coeff = self.coeff / np.linalg.norm(self.coeff)
        elif kind == "mps_norm_to_coeff":
            new_coeff = self.coeff * nrm
        else:
            raise ValueError(f"kind={kind} is not valid.")
        self.scale(1.0 / nrm, inplace=True)
        self.coeff = new_coeff
        return self

    # ------------------------------------------------------------------ observables
    def expectation(self, mpo, self_conj: "Mps" = None):
        """<psi|O|psi> (mps/mps.py:471-525): R-environment sweep and a closing contraction at site 0."""
        if isinstance(mpo, (Op, OpSum)):
            mpo = Mpo(self.model, mpo)
        conj_sites = None if self_conj is None else self_conj._mp
        environ = Environ(self, mpo, "R", mps_conj=conj_sites)
        r = environ.read("R", 1) if len(self) > 1 else environ.sentinel
        val = contract_one_site(r, self[0], environ._mo(mpo, 0), "R",
                                ms_conj=None if conj_sites is None else conj_sites[0]).to_host().reshape(-1)[0]
        val = complex(val)
        return float(val.real) if np.isclose(val.imag, 0) else val

    def expectations(self, mpos, self_conj: "Mps" = None, opt=True) -> np.ndarray:
        """Expectation values of several operators with shared environments, mps/mps.py:527-575 and
        ``_construct_freq_environ`` :2103-2146.  The reference caches L/R prefixes keyed by the hash of the MPO
        site matrices; here every MPO is split into [identity sites | non-trivial window | identity sites], the
        identity-MPO environments (plain overlap matrices, w = 1) are built once for all operators, and only the
        window of each operator is contracted."""
        mpos = [Mpo(self.model, m) if isinstance(m, (Op, OpSum)) else m for m in mpos]
        if not opt or len(mpos) < 2:
            return np.array([self.expectation(m, self_conj) for m in mpos])
        eng = get_engine()
        n = len(self)
        conj_sites = None if self_conj is None else self_conj._mp
        windows = []
        for m in mpos:
            nontrivial = [i for i in range(n) if not _is_identity_site(m[i])]
            windows.append((nontrivial[0], nontrivial[-1]) if nontrivial else (0, -1))
        lmax = max([w[0] for w in windows if w[1] >= 0], default=0)       # L environments needed up to lmax-1
        rmin = min([w[1] for w in windows if w[1] >= 0], default=n - 1)   # R environments needed from rmin+1
        ident = {}

        def ident_w(i):
            d = self[i].shape[1]
            if d not in ident:
                ident[d] = eng.asdevice(np.eye(d).reshape(1, d, d, 1))
            return ident[d]

        def cj(i):
            return None if conj_sites is None else conj_sites[i]

        sentinel = eng.ones((1, 1, 1), np.float64)
        sentinel.unit = 1
        lenv = {-1: sentinel}
        for i in range(0, lmax):
            lenv[i] = contract_one_site(lenv[i - 1], self[i], ident_w(i), "L", ms_conj=cj(i))
        renv = {n: sentinel}
        for i in range(n - 1, rmin, -1):
            renv[i] = contract_one_site(renv[i + 1], self[i], ident_w(i), "R", ms_conj=cj(i))
        out = []
        for m, (first, last) in zip(mpos, windows):
            if last < 0:                       # pure identity (possibly scaled): fall back to the plain path
                out.append(self.expectation(m, self_conj))
                continue
            t = lenv[first - 1]
            for i in range(first, last + 1):
                t = contract_one_site(t, self[i], m.device(i, eng), "L", ms_conj=cj(i))
            r = renv[last + 1]
            # close: sum_{a,c} t[a,0,c] r[a,0,c]   (both (D_bra, 1, D_ket))
            a = t.to_complex() if (t.is_complex or r.is_complex) else t
            b = r.to_complex() if (t.is_complex or r.is_complex) else r
            prod = eng.matmul(a.reshape(1, -1), b.reshape(-1, 1)).to_host().reshape(-1)[0]
            prod = complex(prod)
            out.append(float(prod.real) if np.isclose(prod.imag, 0) else prod)
        return np.array(out)

    @property
    def e_occupations(self):
        """mps/mps.py:600-609"""
        key = "e_occupations"
        if key not in self.model.mpos:
            self.model.mpos[key] = [Mpo(self.model, Op(r"a^\dagger a", dof)) for dof in self.model.e_dofs]
        return self.expectations(self.model.mpos[key])

    @property
    def ph_occupations(self):
        """mps/mps.py:578-594: occupation numbers of the vibrational degrees of freedom, order of model.v_dofs"""
        key = "ph_occupations"
        if key not in self.model.mpos:
            self.model.mpos[key] = [Mpo(self.model, Op("n", dof)) for dof in self.model.v_dofs]
        return self.expectations(self.model.mpos[key])

    def calc_1site_rdm(self, idx=None) -> Dict[int, np.ndarray]:
        """One-site reduced density matrices (mps/mps.py:1547-1598): rdm[i][p, p'] = sum conj(A[a,p,b]) L[a,a']
        R[b,b'] A[a',p',b'] with the identity-operator environments of the other sites.  The environments are one
        right-to-left pass plus one left-to-right pass on the device; only the d x d results come to the host."""
        from ..engine import idx1, idx2
        eng = get_engine()
        n = self.site_num
        if idx is None:
            idx = list(range(n))
        elif isinstance(idx, (int, np.integer)):
            idx = [int(idx)]
        else:
            idx = list(idx)
        ident = [eng.asdevice(np.eye(self[i].shape[1]).reshape(1, self[i].shape[1], self[i].shape[1], 1))
                 for i in range(n)]
        sentinel = eng.ones((1, 1, 1), np.float64)
        sentinel.unit = 1
        renv = {n: sentinel}
        for i in range(n - 1, 0, -1):
            renv[i] = contract_one_site(renv[i + 1], self[i], ident[i], "R")
        rdm = {}
        lenv = sentinel
        for i in range(n):
            ms = self[i]
            if i in idx:
                # density-operator sites (Dl, d, d_anc, Dr): the ancilla leg is traced with the bonds
                # (mps/mps.py:1590-1593) - for the products below it is part of the right bond
                Dl, d, Dr = ms.shape[0], ms.shape[1], int(np.prod(ms.shape[2:]))
                Dr1 = ms.shape[-1]
                L = lenv.reshape(lenv.shape[0], lenv.shape[2])
                R = renv[i + 1].reshape(renv[i + 1].shape[0], renv[i + 1].shape[2])
                # X[a', (p, g, b)] = sum_a L[a, a'] conj(A)[a, (p, g, b)] ; Y[(a', p, g), b'] = sum_b X[(a', p, g), b] R[b, b']
                x = eng.matmul(L, ms.reshape(Dl, d * Dr), trans_a=True, conj_b=True)
                y = eng.matmul(x.reshape(Dl * d * (Dr // Dr1), Dr1), R)
                a = ms.to_complex() if y.is_complex and not ms.is_complex else ms
                # rdm[p, p'] = sum_{a', g, b'} Y[a', p, g, b'] A[a', p', g, b']
                out = eng.empty((d, d), np.complex128 if (y.is_complex or a.is_complex) else np.float64)
                eng.gemm(y, a, out, idx1(d, Dr), idx2(Dl, Dr, d * Dr, 1), idx2(Dl, Dr, d * Dr, 1), idx1(d, Dr),
                         idx1(d, d), idx1(d, 1))
                rdm[i] = out.to_host()
                assert np.allclose(rdm[i], rdm[i].T.conj())
            if i < n - 1:
                lenv = contract_one_site(lenv, ms, ident[i], "L")
        return rdm

    def calc_2site_rdm(self) -> Dict:
        """Two-site reduced density matrices of all pairs i < j (mps/mps.py:1600-1655):
        rdm[(i, j)][(p, q), (p', q')] = sum conj(A_i)[.,p,.] conj(A_j)[.,q,.] A_i[.,p',.] A_j[.,q',.] with the identity
        environments outside and transfer matrices in between.  All contractions are strided GEMMs on the device
        (the running object T[(p,p'),(b,b')] is pushed one site at a time); only the (d_i d_j)^2 results come back."""
        from ..engine import idx1, idx2
        eng = get_engine()
        n = self.site_num
        # density-operator sites (Dl, d, d_anc, Dr): the ancilla leg g is traced wherever a site meets its conjugate
        # (mps/mps.py:1624-1627, 1634-1637, 1648-1651); ``da`` below is 1 for a pure state
        ident = [eng.asdevice(np.eye(self[i].shape[1]).reshape(1, self[i].shape[1], self[i].shape[1], 1))
                 for i in range(n)]
        sentinel = eng.ones((1, 1, 1), np.float64)
        sentinel.unit = 1
        lenv = {-1: sentinel}
        for i in range(n - 1):
            lenv[i] = contract_one_site(lenv[i - 1], self[i], ident[i], "L")
        renv = {n: sentinel}
        for i in range(n - 1, 0, -1):
            renv[i] = contract_one_site(renv[i + 1], self[i], ident[i], "R")
        cplx = self.is_complex
        dt = np.complex128 if cplx else np.float64

        def left_component(i):
            """T[(p,p'),(b,b')] = sum_{a,a'} L[a,a'] conj(A)[a,p,b] A[a',p',b']"""
            a = self[i]
            Dl, d, Dr = a.shape[0], a.shape[1], a.shape[-1]
            da = a.size // (Dl * d * Dr)
            L = lenv[i - 1].reshape(Dl, Dl)
            x = eng.matmul(L, a.reshape(Dl, d * da * Dr), trans_a=True, conj_b=True)     # [a', (p, g, b)]
            t = eng.empty((d * d, Dr * Dr), np.complex128 if (x.is_complex or a.is_complex) else np.float64)
            a2 = a.to_complex() if t.is_complex else a
            x2 = x.to_complex() if t.is_complex else x
            # M = (p, b), K = (a', g), N = (p', b')
            eng.gemm(x2, a2, t, idx2(d, Dr, da * Dr, 1), idx2(Dl, da, d * da * Dr, Dr), idx2(Dl, da, d * da * Dr, Dr),
                     idx2(d, Dr, da * Dr, 1), idx2(d, Dr, d * Dr * Dr, Dr), idx2(d, Dr, Dr * Dr, 1))
            return t

        def right_component(j):
            """Rc[(a,a'),(q,q')] = sum_{c,c'} conj(A)[a,q,c] R[c,c'] A[a',q',c']"""
            a = self[j]
            Dl, d, Dr = a.shape[0], a.shape[1], a.shape[-1]
            da = a.size // (Dl * d * Dr)
            R = renv[j + 1].reshape(Dr, Dr)
            y = eng.matmul(a.reshape(Dl * d * da, Dr), R, conj_a=True)                    # [(a, q, g), c']
            rc = eng.empty((Dl * Dl, d * d), np.complex128 if (y.is_complex or a.is_complex) else np.float64)
            a2 = a.to_complex() if rc.is_complex else a
            y2 = y.to_complex() if rc.is_complex else y
            # M = (a, q), K = (g, c') (contiguous), N = (a', q')
            eng.gemm(y2, a2, rc, idx1(Dl * d, da * Dr), idx1(da * Dr, 1), idx1(da * Dr, 1), idx1(Dl * d, da * Dr),
                     idx2(Dl, d, Dl * d * d, d), idx2(Dl, d, d * d, 1))
            return rc

        def transfer(t, k, dd):
            """T'[(pp'),(c,c')] = sum_{b,b',s} T[(pp'),(b,b')] conj(A_k)[b,s,c] A_k[b',s,c'] (s: physical and ancilla leg)"""
            a = self[k]
            D, Dc = a.shape[0], a.shape[-1]
            dk = a.size // (D * Dc)
            cdt = np.complex128 if (t.is_complex or a.is_complex) else np.float64
            a2 = a.to_complex() if cdt == np.complex128 else a
            t2 = t.to_complex() if cdt == np.complex128 else t
            u = eng.empty((dd * D, dk * Dc), cdt)                                            # [(pp', b'), (s, c)]
            eng.gemm(t2, a2, u, idx2(dd, D, D * D, 1), idx1(D, D), idx1(D, dk * Dc), idx1(dk * Dc, 1),
                     idx1(dd * D, dk * Dc), idx1(dk * Dc, 1), conj_b=True)
            out = eng.empty((dd, Dc * Dc), cdt)
            eng.gemm(u, a2, out, idx2(dd, Dc, D * dk * Dc, 1), idx1(D * dk, Dc), idx1(D * dk, Dc), idx1(Dc, 1),
                     idx2(dd, Dc, Dc * Dc, Dc), idx1(Dc, 1))
            return out

        rcs = {j: right_component(j) for j in range(1, n)}
        rdm = {}
        for i in range(n - 1):
            di = self[i].shape[1]
            t = left_component(i)
            for j in range(i + 1, n):
                if j != i + 1:
                    t = transfer(t, j - 1, di * di)
                dj = self[j].shape[1]
                res = eng.matmul(t, rcs[j]).to_host().reshape(di, di, dj, dj)                # [p, p', q, q']
                rdm[(i, j)] = res.transpose(0, 2, 1, 3).reshape(di * dj, di * dj)
        return rdm

    def calc_2site_mutual_entropy(self) -> np.ndarray:
        """m_ij = (s_i + s_j - s_ij) / 2 (mps/mps.py:1734-1757)"""
        s1 = self.calc_entropy("1site")
        s2 = self.calc_entropy("2site")
        n = self.site_num
        m = np.zeros((n, n))
        for (i, j), v in s2.items():
            m[i, j] = (s1[i] + s1[j] - v) / 2
        return m + m.T

    def calc_edof_rdm(self) -> np.ndarray:
        """rho_ij = <a_i^dagger a_j> over the electronic degrees of freedom (mps/mps.py:1657-1687)"""
        key = "edof_reduced_density_matrix"
        e_dofs = self.model.e_dofs
        n_e = len(e_dofs)
        if key not in self.model.mpos:
            self.model.mpos[key] = [Mpo(self.model, Op(r"a^\dagger a", [d1, d2]))
                                    for i, d1 in enumerate(e_dofs) for d2 in e_dofs[i:]]
        vals = list(np.atleast_1d(self.expectations(self.model.mpos[key])))
        rho = np.zeros((n_e, n_e), dtype=np.complex128)
        k = 0
        for i in range(n_e):
            for j in range(i, n_e):
                rho[i, j] = vals[k]
                rho[j, i] = np.conj(vals[k])
                k += 1
        return rho

    def calc_bond_singular_values(self) -> np.ndarray:
        """mps/mps.py:1759-1773: singular values at every bond (rows padded with zeros), on a copy"""
        mps = self.copy()
        mps.ensure_right_canonical()
        _, s_array = mps.compress(temp_m_trunc=np.inf, ret_s=True)
        return s_array

    def calc_bond_entropy(self, s_array=None) -> np.ndarray:
        """von Neumann entropy of every bipartition (mps/mps.py:1775-1795)"""
        if s_array is None:
            s_array = self.calc_bond_singular_values()
        return np.array([_vn_entropy(np.asarray(sigma) ** 2) for sigma in s_array])

    def calc_entropy(self, entropy_type):
        """mps/mps.py:1689-1732"""
        if entropy_type == "1site":
            out = {}
            for k, dm in self.calc_1site_rdm().items():
                w = np.linalg.eigvalsh(dm)
                out[k] = _vn_entropy(w)
            return out
        if entropy_type == "2site":
            return {k: _vn_entropy(np.linalg.eigvalsh((dm + dm.conj().T) / 2)) for k, dm in self.calc_2site_rdm().items()}
        if entropy_type == "mutual":
            return self.calc_2site_mutual_entropy()
        if entropy_type == "bond":
            return self.calc_bond_entropy()
        raise ValueError(f"unsupported entropy type {entropy_type}")

    # ------------------------------------------------------------------ canonical form / compression
    def _update_ms(self, idx, u, vt, sigma=None, qnlset=None, qnrset=None, m_trunc=None):
        """mps/mp.py:245-295 for an MPS: keep the first m_trunc columns, push sigma/R to the neighbour."""
        eng = get_engine()
        K = u.shape[1]
        if m_trunc is None:
            m_trunc = K
        m_trunc = int(m_trunc)
        q = len(self.qntot)
        if m_trunc < K or sigma is not None:
            keep = np.arange(m_trunc, dtype=np.int64)
            p = svd_qn._p64(keep)
            su = sv = None
            if sigma is not None:
                sg = np.ascontiguousarray(np.asarray(sigma, dtype=float)[:m_trunc])
                sp = sg.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
                su, sv = (None, sp) if self.to_right else (sp, None)
            u2 = eng.empty((u.shape[0], m_trunc), u.dtype)
            vt2 = eng.empty((m_trunc, vt.shape[1]), vt.dtype)
            eng._check(eng.lib.mpse_gather_cols(eng.ctx, u.code, u2.ptr, u.ptr, u.shape[0], K, p, su, m_trunc))
            eng._check(eng.lib.mpse_gather_rows(eng.ctx, vt.code, vt2.ptr, vt.ptr, vt.shape[1], p, sv, m_trunc))
            u, vt = u2, vt2
        pdim = self[idx].shape[1:-1]
        if self.to_right:
            nxt = self[idx + 1]
            self[idx + 1] = eng.matmul(vt, nxt.reshape(nxt.shape[0], -1)).reshape((m_trunc,) + nxt.shape[1:])
            self[idx] = u.reshape((-1,) + tuple(pdim) + (m_trunc,))
            if qnlset is not None:
                self.qn[idx + 1] = np.array(qnlset[:m_trunc], dtype=int).reshape(m_trunc, q)
                self.qnidx = idx + 1
        else:
            prv = self[idx - 1]
            self[idx - 1] = eng.matmul(prv.reshape(-1, prv.shape[-1]), u).reshape(prv.shape[:-1] + (m_trunc,))
            self[idx] = vt.reshape((m_trunc,) + tuple(pdim) + (-1,))
            if qnrset is not None:
                self.qn[idx] = np.array(qnrset[:m_trunc], dtype=int).reshape(m_trunc, q)
                self.qnidx = idx - 1

    def _push_cano(self, idx):
        """mps/mp.py:890-908"""
        qnbigl, qnbigr, _ = self._get_big_qn([idx], need_mat=False)
        system = "L" if self.to_right else "R"
        u, qnlset, v, qnrset = svd_qn.svd_qn(self[idx], qnbigl, qnbigr, self.qntot, QR=True, system=system,
                                             full_matrices=False)
        self._update_ms(idx, u, v.T, sigma=None, qnlset=qnlset, qnrset=qnrset)

    def canonicalise(self, stop_idx: int = None):
        """mps/mp.py:910-922"""
        if self.to_right:
            assert self.qnidx == 0
        else:
            assert self.qnidx == self.site_num - 1
        idx = None
        for idx in self.iter_idx_list(full=False, stop_idx=stop_idx):
            self._push_cano(idx)
        if idx is not None and ((not self.to_right and idx == 1) or (self.to_right and idx == self.site_num - 2)):
            self._switch_direction()
        return self

    def compress(self, temp_m_trunc=None, ret_s=False):
        """SVD sweep, mps/mp.py:437-511 (the caller canonicalises first, as in the reference)."""
        if self.to_right:
            assert self.qnidx == 0
        else:
            assert self.qnidx == self.site_num - 1
        if self.compress_config.bonddim_should_set:
            self.compress_config.set_bonddim(len(self) + 1)
        system = "L" if self.to_right else "R"
        s_list = []
        for idx in self.iter_idx_list(full=False):
            qnbigl, qnbigr, _ = self._get_big_qn([idx], need_mat=False)
            u, sigma, qnlset, v, sigma, qnrset = svd_qn.svd_qn(self[idx], qnbigl, qnbigr, self.qntot, system=system,
                                                               full_matrices=False)
            s_list.append(sigma)
            if temp_m_trunc is None:
                m_trunc = self.compress_config.compute_m_trunc(sigma, idx, self.to_right)
            else:
                if isinstance(temp_m_trunc, (list, tuple, np.ndarray)):
                    m_trunc = temp_m_trunc[idx + 1 if self.to_right else idx]
                else:
                    m_trunc = temp_m_trunc
                m_trunc = len(sigma) if np.isinf(m_trunc) else min(int(m_trunc), len(sigma))
            self._update_ms(idx, u, v.T, sigma, qnlset, qnrset, m_trunc)
        self._switch_direction()
        if not ret_s:
            return self
        width = max(len(s) for s in s_list)
        return self, np.array([np.pad(s, (0, width - len(s))) for s in s_list])

    # ------------------------------------------------------------------ DMRG support
    @property
    def threshold(self):
        return self.compress_config.threshold

    @threshold.setter
    def threshold(self, v):
        self.compress_config.threshold = v

    def evolve_exact(self, h_mpo, evolve_dt, space):
        """exp(-i H dt) for the local (purely vibrational) Hamiltonian of a Holstein model in the electron-free ("GS")
        or the single-site excited ("EX") space: one application of the bond-dimension-1 propagator
        (mps/mps.py:1519-1523, mps/mpdm.py:76-83); the energy offset of ``h_mpo`` goes into ``coeff`` as a phase."""
        from .mpo import Mpo
        offset = getattr(h_mpo, "offset", 0.0)
        offset = offset.as_au() if hasattr(offset, "as_au") else float(offset)
        prop = Mpo.exact_propagator(self.model, -1.0j * evolve_dt, space=space, shift=-offset)
        new = prop.apply(self, canonicalise=True)
        new.coeff = new.coeff * np.exp(-1.0j * offset * evolve_dt)
        return new

    @property
    def digest(self):
        """variance / mean / peak-to-peak of the dense state: a quick fingerprint for comparing two small pure states
        (mps/mps.py:1525-1534); None for more than ten sites or density operators"""
        if 10 < self.site_num or self.is_mpdm:
            return None
        dense = self.todense() / self.coeff
        return {"var": dense.var(), "mean": dense.mean(), "ptp": np.ptp(dense)}

    def variational_compress(self, mpo=None, guess=None):
        """Approximation of ``mpo @ self`` at the bond dimensions of ``compress_config.vprocedure`` by sweeps that
        maximise the overlap with the exact product (mps/mp.py:513-650): per step the effective operator
        <guess| mpo |self> around the centre is applied to the centre of ``self`` - an effective-Hamiltonian product
        whose bra bonds are those of the guess - and the result replaces the centre of the guess through the
        basis-selection update (1-site or 2-site, ``vmethod``); converged when a sweep without forced basis
        mixing moves the state by less than ``vrtol``.  ``self`` is not modified, ``guess`` is.

        Without a guess the reference applies an SVD-compressed copy of the MPO (bond ``vguess_m[0]``) to an
        SVD-compressed copy of the state (bond ``vguess_m[1]``); here the full MPO is applied to the compressed state
        and the product is compressed to ``vguess_m[0] * vguess_m[1]``, the bond dimension that product would have."""
        from ..utils import CompressConfig, CompressCriteria
        if mpo is None:
            raise NotImplementedError("Recommend to use svd to compress a single mps/mpo/mpdm.")
        eng = get_engine()
        if guess is None:
            small = self.copy().canonicalise()
            small.compress(temp_m_trunc=self.compress_config.vguess_m[1])
            guess = mpo.apply(small).canonicalise()
            guess.compress(temp_m_trunc=self.compress_config.vguess_m[0] * self.compress_config.vguess_m[1])
            guess.compress_config = self.compress_config.copy()
        mps = guess
        mps.ensure_left_canonical()
        procedure, method = mps.compress_config.vprocedure, mps.compress_config.vmethod
        assert method in ("1site", "2site")
        n = mps.site_num
        environ = Environ(self, mpo, "L", mps_conj=mps.conj())
        mps_old, converged = None, False
        for isweep, (cc, percent) in enumerate(procedure):
            if isinstance(cc, CompressConfig):
                mps.compress_config = cc
            else:
                mps.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=int(cc))
            for imps in mps.iter_idx_list(full=True):
                if method == "2site" and ((mps.to_right and imps == n - 1) or (not mps.to_right and imps == 0)):
                    break
                lmethod, rmethod = ("System", "Enviro") if mps.to_right else ("Enviro", "System")
                if method == "1site":
                    lidx, cidx, ridx = imps - 1, [imps], imps + 1
                elif mps.to_right:
                    lidx, cidx, ridx = imps - 1, [imps, imps + 1], imps + 2
                else:
                    lidx, cidx, ridx = imps - 2, [imps - 1, imps], imps + 1
                conj = mps.conj()
                ltensor = environ.GetLR("L", lidx, self, mpo, itensor=None, method=lmethod, mps_conj=conj)
                rtensor = environ.GetLR("R", ridx, self, mpo, itensor=None, method=rmethod, mps_conj=conj)
                qnbigl, qnbigr, qnmat = mps._get_big_qn(cidx)
                qn_mask = svd_qn.get_qn_mask(qnmat, mps.qntot)
                if method == "1site":
                    cms = self[cidx[0]]
                else:
                    a, b = self[cidx[0]], self[cidx[1]]
                    cms = eng.matmul(a.reshape(-1, a.shape[-1]), b.reshape(b.shape[0], -1)).reshape(a.shape[:-1] + b.shape[1:])
                hop = hop_expr(ltensor, rtensor, [mpo.device(i, eng) for i in cidx], cms.shape)
                cout = hop(cms)
                assert tuple(cout.shape) == qn_mask.shape, (cout.shape, qn_mask.shape)
                mask = eng.asdevice(qn_mask.astype(np.float64))           # symmetry-forbidden entries are rounding noise
                eng._check(eng.lib.mpse_mul_real(eng.ctx, cout.code, cout.ptr, mask.ptr, cout.size))
                mps._update_mps(cout, cidx, qnbigl, qnbigr, percent)
                if mps.compress_config.ofs is not None:
                    raise NotImplementedError("OFS for variational compress not implemented")
            mps._switch_direction()
            if isweep > 0 and percent == 0:
                error = mps.distance(mps_old) / np.sqrt(abs(mps.dot(mps, self_is_conj=False).real))
                if error < mps.compress_config.vrtol:
                    converged = True
                    break
            mps_old = mps.copy()
        if not converged:
            logger.warning("Variational compress is not converged! Please increase the procedure!")
        return mps.canonicalise()

    @property
    def is_left_canonical(self):
        """mps/mp.py:192-197 (position of the qn centre only)"""
        return self.qnidx == self.site_num - 1

    @property
    def is_right_canonical(self):
        return self.qnidx == 0

    def _check_ortho(self, i, left: bool, rtol=None, atol=None) -> bool:
        """A_i^+ A_i = 1 (left) or A_i A_i^+ = 1 (right) within the backend tolerances (mps/matrix.py:121-150); the
        Gram matrix is formed on the device, only D x D numbers come back."""
        from .backend import backend
        rtol = backend.canonical_rtol if rtol is None else rtol
        atol = backend.canonical_atol if atol is None else atol
        eng = get_engine()
        a = self[i]
        m = a.reshape(-1, a.shape[-1]) if left else a.reshape(a.shape[0], -1)
        g = (eng.matmul(m, m, trans_a=True, conj_a=True) if left else eng.matmul(m, m, trans_b=True, conj_b=True)).to_host()
        return bool(np.allclose(g, np.eye(g.shape[0]), rtol=rtol, atol=atol))

    def check_left_canonical(self, rtol: float = None, atol: float = None) -> bool:
        """mps/mp.py:174-181"""
        return all(self._check_ortho(i, True, rtol, atol) for i in range(len(self) - 1))

    def check_right_canonical(self, rtol: float = None, atol: float = None) -> bool:
        """mps/mp.py:183-190"""
        return all(self._check_ortho(i, False, rtol, atol) for i in range(1, len(self)))

    def angle(self, other) -> float:
        """|<self|other>| (mps/mp.py:981-982)"""
        return abs(self.dot(other, self_is_conj=False))

    @property
    def total_bytes(self) -> int:
        """device memory held by the site tensors (mps/mp.py:1116-1117)"""
        return int(sum(t.nbytes for t in self._mp))

    def ensure_left_canonical(self, rtol: float = None, atol: float = None):
        """mps/mp.py:206-216: canonicalise only when the direction flags or the site Gram matrices say it is needed -
        a QR pass over an already canonical state would still rotate its weightless (padded) bond directions, which
        the regularised inverses of TDVP-VMF / CMF are sensitive to"""
        if self.to_right or self.qnidx != self.site_num - 1 or not self.check_left_canonical(rtol, atol):
            self.move_qnidx(0)
            self.to_right = True
            return self.canonicalise()
        return self

    def ensure_right_canonical(self, rtol: float = None, atol: float = None):
        """mps/mp.py:218-228"""
        if (not self.to_right) or self.qnidx != 0 or not self.check_right_canonical(rtol, atol):
            self.move_qnidx(self.site_num - 1)
            self.to_right = False
            return self.canonicalise()
        return self

    def _ofs_try_swap(self, cstruct, cidx, s_keep, system):
        """On-the-fly swapping of the two centre sites (mps/mp.py:696-757, J. Chem. Theory Comput. 18, 6437): the
        two-site tensor is decomposed in the current and in the exchanged site order; the order with the smaller
        discarded weight at the fixed bond dimension (OFS-D), the smaller entanglement entropy (OFS-S), or the
        hybrid of the two (OFS-D/S) is kept.  On a swap the model's site order is exchanged; the caller swaps the
        MPO sites (``Mpo.try_swap_site``).  Returns the decomposition in the new order, or None to keep the old."""
        from ..model import HolsteinModel, Model
        from ..utils import OFS, CompressCriteria
        cfg = self.compress_config
        if isinstance(self.model, HolsteinModel):
            raise NotImplementedError("Can't perform OFS on Holstein model")      # its site order is part of the class
        assert cfg.criteria is CompressCriteria.fixed
        eng = get_engine()
        c = eng.asdevice(cstruct)
        dl, dr = c.shape[0], c.shape[-1]
        half = (c.ndim - 2) // 2
        p1, p2 = int(np.prod(c.shape[1:1 + half])), int(np.prod(c.shape[1 + half:-1]))
        # (dl, p1, p2, dr) -> (dl, p2, p1, dr) with two inner transposes on the device
        t1 = eng.empty((dl, p2 * dr, p1), c.dtype)
        eng._check(eng.lib.mpse_transpose_inner(eng.ctx, c.code, t1.ptr, c.ptr, dl, p1, p2 * dr, 0))
        c2 = eng.empty((dl * p2, p1, dr), c.dtype)
        eng._check(eng.lib.mpse_transpose_inner(eng.ctx, c.code, c2.ptr, t1.ptr, dl * p2, dr, p1, 0))
        c2 = c2.reshape((dl,) + tuple(c.shape[1 + half:-1]) + tuple(c.shape[1:1 + half]) + (dr,))
        if cfg.ofs_swap_jw:
            # fermionic modes of a Jordan-Wigner chain: the exchange flips the sign of the amplitudes with both modes
            # occupied (mps/mp.py:711-714)
            if c2.ndim != 4 or c2.shape[1] != 2 or c2.shape[2] != 2:
                raise ValueError("ofs_swap_jw: the two centre sites must be two-level Jordan-Wigner sites")
            sign = np.ones((dl, 2, 2, dr))
            sign[:, 1, 1, :] = -1.0
            eng._check(eng.lib.mpse_mul_real(eng.ctx, c2.code, c2.ptr, eng.asdevice(sign).ptr, c2.size))
        qnbigl2, qnbigr2, _ = self._get_big_qn(cidx, swap=True, need_mat=False)
        U2, SU2, qnl2, V2, SV2, qnr2 = svd_qn.svd_qn(c2, qnbigl2, qnbigr2, self.qntot, system=system)
        s1, s2 = np.asarray(s_keep, dtype=float), np.asarray(SU2, dtype=float)
        ent1, ent2 = _vn_entropy(s1 ** 2), _vn_entropy(s2 ** 2)
        mmax = cfg.bond_dim_max_value
        loss1 = float((np.sort(s1)[::-1][mmax:] ** 2).sum())
        loss2 = float((np.sort(s2)[::-1][mmax:] ** 2).sum())
        if cfg.ofs is OFS.ofs_d:
            retain = loss1 <= loss2
        elif cfg.ofs is OFS.ofs_ds:
            retain = (ent1 <= ent2) if (loss1 < 1e-10 and loss2 < 1e-10) else (loss1 <= loss2)
        elif cfg.ofs is OFS.ofs_s:
            retain = ent1 <= ent2
        else:
            assert cfg.ofs is OFS.ofs_debug
            retain = True
        logger.debug(f"OFS: sites {cidx}, swap: {not retain}, S: {ent1}, {ent2}, loss: {loss1}, {loss2}")
        if retain:
            return None
        basis = list(self.model.basis)
        basis[cidx[0]:cidx[1] + 1] = basis[cidx[0]:cidx[1] + 1][::-1]
        self.model = Model(basis, self.model.ham_terms, self.model.dipole, self.model.output_ordering)
        return U2, SU2, qnl2, V2, SV2, qnr2, qnbigl2, qnbigr2

    def _update_mps(self, cstruct, cidx, qnbigl, qnbigr, percent=0):
        """Basis selection update after a DMRG / two-site step, mps/mp.py:651-888 for a single state:
        full block SVD of the centre, m_trunc from the compress config, ``select_basis`` (equal quota per qn
        block for ``percent`` of the slots, then by singular value), new site = selected vectors, the
        complementary part (sigma-weighted) goes to the neighbour."""
        from .basis_select import select_basis_indices
        eng = get_engine()
        system = "L" if self.to_right else "R"
        if self.compress_config.bonddim_should_set:
            self.compress_config.set_bonddim(len(self) + 1)
        q = len(self.qntot)
        roots = None
        if isinstance(cstruct, (list, tuple)):
            # state-averaged DMRG (mp.py:700-756): the eigenvectors of the averaged reduced density matrix
            # sum_r c_r c_r^+ / nroots are the singular vectors of the root-stacked centre matrix, so the same
            # block SVD (with null-space completion) serves; "singular values" handed on are the eigenvalues
            roots = [eng.asdevice(c) for c in cstruct]
            nroot = len(roots)
            nrow = int(np.prod(np.asarray(qnbigl).shape[:-1]))
            ncol = int(np.prod(np.asarray(qnbigr).shape[:-1]))
            dt = np.complex128 if any(c.is_complex for c in roots) else np.float64
            roots = [(c.to_complex() if dt == np.complex128 else c).reshape(nrow, ncol) for c in roots]
            if self.to_right:
                stack = eng.empty((nrow, nroot * ncol), dt)
                for r, c in enumerate(roots):
                    eng.copy_block(stack, 0, r * ncol, c)
                qr_flat = np.asarray(qnbigr).reshape(ncol, q)
                qnl_s, qnr_s = qnbigl, np.concatenate([qr_flat] * nroot, axis=0)
            else:
                stack = eng.empty((nroot * nrow, ncol), dt)
                for r, c in enumerate(roots):
                    eng.copy_block(stack, r * nrow, 0, c)
                ql_flat = np.asarray(qnbigl).reshape(nrow, q)
                qnl_s, qnr_s = np.concatenate([ql_flat] * nroot, axis=0), qnbigr
            stack.scale_(1.0 / np.sqrt(nroot))
            U, SU, qnlnew, V, SV, qnrnew = svd_qn.svd_qn(stack, qnl_s, qnr_s, self.qntot, system=system)
            SU, SV = np.asarray(SU) ** 2, np.asarray(SV) ** 2
        else:
            U, SU, qnlnew, V, SV, qnrnew = svd_qn.svd_qn(cstruct, qnbigl, qnbigr, self.qntot, system=system)
            if self.compress_config.ofs is not None and len(cidx) == 2:
                swapped = self._ofs_try_swap(cstruct, cidx, SU, system)
                if swapped is not None:
                    U, SU, qnlnew, V, SV, qnrnew, qnbigl, qnbigr = swapped
        Vt = V.T
        if self.to_right:
            m_trunc = self.compress_config.compute_m_trunc(SU, cidx[0], self.to_right)
            sidx = select_basis_indices(SU, qnlnew, m_trunc, percent)
            sset, qnsel, nsel = SU, qnlnew, len(sidx)
        else:
            m_trunc = self.compress_config.compute_m_trunc(SV, cidx[-1], self.to_right)
            sidx = select_basis_indices(SV, qnrnew, m_trunc, percent)
            sset, qnsel, nsel = SV, qnrnew, len(sidx)
        msqn = np.array([qnsel[i] for i in sidx], dtype=int).reshape(nsel, q)
        idx = np.array(sidx, dtype=np.int64)
        sig = np.ascontiguousarray(np.asarray(sset, dtype=float)[idx])
        nrow, ncol = U.shape[0], Vt.shape[1]
        lib = eng.lib

        def cols_of(mat, index, scale):
            """columns `index` of a row-major matrix; columns past its width are zero (lib.py:311-314)"""
            width = mat.shape[1]
            ok = index < width
            out = eng.empty((mat.shape[0], len(index)), mat.dtype)
            sc = None if scale is None else np.where(ok, scale, 0.0)
            if scale is None and not ok.all():
                sc = ok.astype(float)
            safe = np.ascontiguousarray(np.where(ok, index, 0))
            eng._check(lib.mpse_gather_cols(eng.ctx, mat.code, out.ptr, mat.ptr, mat.shape[0], width, svd_qn._p64(safe),
                                            None if sc is None else np.ascontiguousarray(sc).ctypes.data_as(
                                                ctypes.POINTER(ctypes.c_double)), len(index)))
            return out

        def rows_of(mat, index, scale):
            height = mat.shape[0]
            ok = index < height
            out = eng.empty((len(index), mat.shape[1]), mat.dtype)
            sc = None if scale is None else np.where(ok, scale, 0.0)
            if scale is None and not ok.all():
                sc = ok.astype(float)
            safe = np.ascontiguousarray(np.where(ok, index, 0))
            eng._check(lib.mpse_gather_rows(eng.ctx, mat.code, out.ptr, mat.ptr, mat.shape[1], svd_qn._p64(safe),
                                            None if sc is None else np.ascontiguousarray(sc).ctypes.data_as(
                                                ctypes.POINTER(ctypes.c_double)), len(index)))
            return out

        lshape = list(np.asarray(qnbigl).shape[:-1])
        rshape = list(np.asarray(qnbigr).shape[:-1])
        rotated, averaged_ms = None, None
        if roots is not None:
            # new basis = selected eigenvectors; every root is rotated into it (mp.py:729-756)
            averaged_ms = []
            if self.to_right:
                ms2 = cols_of(U, idx, None)                                     # (nrow, M)
                ms = ms2.reshape(lshape + [nsel])
                rotated = [eng.matmul(ms2, c, trans_a=True, conj_a=True).reshape([nsel] + rshape) for c in roots]
            else:
                ms2 = rows_of(Vt, idx, None)                                    # (M, ncol)
                ms = ms2.reshape([nsel] + rshape)
                rotated = [eng.matmul(c, ms2, trans_b=True, conj_b=True).reshape(lshape + [nsel]) for c in roots]
            compms = rotated[0]
        elif self.to_right:
            ms = cols_of(U, idx, None).reshape(lshape + [nsel])                 # (D_l, d, M)
            compms = rows_of(Vt, idx, sig).reshape([nsel] + rshape)             # (M, [d,] D_r) = moveaxis(V sigma)
        else:
            ms = rows_of(Vt, idx, None).reshape([nsel] + rshape)                # (M, d, D_r)
            compms = cols_of(U, idx, sig).reshape(lshape + [nsel])              # (D_l, [d,] M)
        if len(cidx) == 1:
            c = cidx[0]
            self[c] = ms
            if rotated is not None:
                # better initial guesses for the next centre: every root absorbed into the neighbour (mp.py:838-868)
                if self.to_right:
                    if c != self.site_num - 1:
                        nxt = self[c + 1]
                        averaged_ms = [eng.matmul(rc.reshape(nsel, -1), nxt.reshape(nxt.shape[0], -1))
                                       .reshape((nsel,) + nxt.shape[1:]) for rc in rotated]
                    else:
                        averaged_ms = [eng.matmul(ms.reshape(-1, nsel), rc.reshape(nsel, -1)).reshape(lshape + [-1])
                                       for rc in rotated]
                else:
                    if c != 0:
                        prv = self[c - 1]
                        averaged_ms = [eng.matmul(prv.reshape(-1, prv.shape[-1]), rc.reshape(-1, nsel))
                                       .reshape(prv.shape[:-1] + (nsel,)) for rc in rotated]
                    else:
                        averaged_ms = [eng.matmul(rc.reshape(-1, nsel), ms.reshape(nsel, -1)).reshape([-1] + rshape)
                                       for rc in rotated]
            if self.to_right:
                if c != self.site_num - 1:
                    nxt = self[c + 1]
                    self[c + 1] = eng.matmul(compms.reshape(nsel, -1), nxt.reshape(nxt.shape[0], -1)) \
                        .reshape((nsel,) + nxt.shape[1:])
                    self.qn[c + 1] = msqn
                    self.qnidx = c + 1
                else:
                    self[c] = eng.matmul(ms.reshape(-1, nsel), compms.reshape(nsel, -1)).reshape(lshape + [-1])
                    self.qnidx = self.site_num - 1
            else:
                if c != 0:
                    prv = self[c - 1]
                    self[c - 1] = eng.matmul(prv.reshape(-1, prv.shape[-1]), compms.reshape(-1, nsel)) \
                        .reshape(prv.shape[:-1] + (nsel,))
                    self.qn[c] = msqn
                    self.qnidx = c - 1
                else:
                    self[c] = eng.matmul(compms.reshape(-1, nsel), ms.reshape(nsel, -1)).reshape([-1] + rshape)
                    self.qnidx = 0
        else:
            if self.to_right:
                self[cidx[0]] = ms
                self[cidx[1]] = compms
                self.qnidx = cidx[1]
            else:
                self[cidx[1]] = ms
                self[cidx[0]] = compms
                self.qnidx = cidx[0]
            self.qn[cidx[1]] = msqn
            if rotated is not None:
                averaged_ms = rotated
        return averaged_ms

    # ------------------------------------------------------------------ sums of states
    def add(self, other: "Mps") -> "Mps":
        """Direct sum of the bond spaces (block-diagonal site tensors), mps/mp.py:374-435."""
        eng = get_engine()
        assert np.all(self.qntot == other.qntot)
        assert self.site_num == other.site_num
        if not np.allclose(self.coeff, other.coeff):
            # mps/mps.py:1802-1808: different prefactors are folded into the states before the sum
            a, b = self.scale(self.coeff), other.scale(other.coeff)
            a.coeff = b.coeff = 1
            return a.add(b)
        new = self.metacopy()
        cplx = self.is_complex or other.is_complex
        dt = np.dtype(np.complex128 if cplx else np.float64)
        new.dtype = dt
        n = self.site_num
        sites = []
        for i in range(n):
            a = self[i].to_complex() if cplx else self[i]
            b = other[i].to_complex() if cplx else other[i]
            assert a.shape[1:-1] == b.shape[1:-1]
            phys = tuple(a.shape[1:-1])
            d = int(np.prod(phys))
            la, ra, lb, rb = a.shape[0], a.shape[-1], b.shape[0], b.shape[-1]
            if n == 1:
                raise NotImplementedError("add of single-site states")
            if i == 0:
                L, R, l0, r0 = 1, ra + rb, 0, ra
            elif i == n - 1:
                L, R, l0, r0 = la + lb, 1, la, 0
            else:
                L, R, l0, r0 = la + lb, ra + rb, la, ra
            out = eng.zeros((L,) + phys + (R,), dt)
            o2 = out.reshape(L * d, R)
            eng.copy_block(o2, 0, 0, a.reshape(la * d, ra))
            eng.copy_block(o2, l0 * d, r0, b.reshape(lb * d, rb))
            sites.append(out)
        new._mp = sites
        new.move_qnidx(other.qnidx)
        new.to_right = other.to_right
        new.qn = [np.concatenate([q1, q2]) for q1, q2 in zip(new.qn, other.qn)]
        q = len(self.qntot)
        new.qn[0] = np.zeros((1, q), dtype=int)
        new.qn[-1] = np.zeros((1, q), dtype=int)
        return new

    __add__ = add

    def distance(self, other) -> float:
        """mps/mp.py:1009-1023"""
        l1 = self.dot(self, self_is_conj=False)
        l2 = other.dot(other, self_is_conj=False)
        l12 = self.dot(other, self_is_conj=False)
        d2 = (l1 + l2 - l12 - l12.conjugate()).real
        if d2 < 0:
            assert d2 / l1.real < 1e-8
            return 0.0
        return float(np.sqrt(d2))

    def expand_bond_dimension(self, hint_mpo=None, coef=1e-10, include_ex=True):
        """Grow the bonds to ``compress_config.max_dims`` with states reachable through ``hint_mpo``
        (repeated H|psi>, compressed sums) added with a tiny weight; mps/mps.py:1934-2023."""
        mps = self
        ex_mps = None
        if hint_mpo is not None and include_ex:
            if mps.is_mpdm:
                assert int(mps.qntot[0]) == 1
                ex_state = mps.__class__.max_entangled_ex(mps.model)
            else:
                ex_state = Mps.ground_state(mps.model, False)
                assert mps.model.qn_size == 1
                for _ in range(int(mps.qntot[0])):
                    ex_state = Mpo.onsite(mps.model, r"a^\dagger").apply(ex_state)
            ex_state.compress_config = mps.compress_config
            ex_state.move_qnidx(mps.qnidx)
            ex_state.to_right = mps.to_right
            ex_mps = ex_state
        mps.compress_config.set_bonddim(len(mps.bond_dims))
        m_target = np.minimum(np.array(mps.compress_config.max_dims) - np.array(mps.bond_dims), mps.bond_dims_exact)
        m_target = np.array(m_target, dtype=int)
        if hint_mpo is None:
            expander = mps.__class__.random(mps.model, mps.qntot, m_target)
            expander.compress_config = mps.compress_config.copy()
        else:
            lastone = mps if ex_mps is None else mps + ex_mps
            expander_list = []
            expander_dims = np.zeros_like(m_target)
            while True:
                lastone = hint_mpo.apply(lastone).normalize("mps_and_coeff")
                lastone = lastone.canonicalise().compress(int(np.max(m_target)))
                expander_list.append(lastone)
                expander = compressed_sum(expander_list, temp_m_trunc=m_target)
                if np.all(np.array(expander.bond_dims) >= m_target):
                    break
                if np.all(np.array(expander.bond_dims) == expander_dims):
                    logger.warning("Expander does not increase anymore. The expand target is too high")
                    m_target2 = int(np.max(m_target - np.array(expander_dims)))
                    expander2 = hint_mpo.apply(lastone).canonicalise().compress(max(m_target2, 1))
                    expander = expander + expander2
                    break
                expander_dims = np.array(expander.bond_dims)
                temp = int(np.max(m_target) / np.max(hint_mpo.bond_dims)) + 1
                lastone = lastone.canonicalise().compress(temp)
        summed = mps + expander.scale(coef * mps.norm, inplace=True)
        return summed.canonicalise().compress(mps.compress_config.max_dims).normalize("mps_norm_to_coeff")

    # ------------------------------------------------------------------ time evolution
    def evolve(self, mpo, evolve_dt, normalize=True) -> "Mps":
        """mps/mps.py:644-662"""
        method = self.evolve_config.method
        if method in (EvolveMethod.tdvp_ps, EvolveMethod.tdvp_ps2):
            step = Mps._evolve_tdvp_ps if method is EvolveMethod.tdvp_ps else Mps._evolve_tdvp_ps2
            if self.evolve_config.adaptive:
                new_mps = _adaptive_tdvp(step, self, mpo, evolve_dt)
            else:
                new_mps = step(self, mpo, evolve_dt)
        elif method is EvolveMethod.prop_and_compress:
            new_mps = self._evolve_prop_and_compress(mpo, evolve_dt)
        elif method is EvolveMethod.prop_and_compress_tdrk4:
            new_mps = self._evolve_prop_and_compress_tdrk4(mpo, evolve_dt)
        elif method is EvolveMethod.prop_and_compress_tdrk:
            new_mps = self._evolve_prop_and_compress_tdrk(mpo, evolve_dt)
        elif method in (EvolveMethod.tdvp_vmf, EvolveMethod.tdvp_mu_vmf):
            from .tdvp_vmf import evolve_tdvp_mu_vmf
            new_mps = evolve_tdvp_mu_vmf(self, mpo, evolve_dt)
        elif method is EvolveMethod.tdvp_mu_cmf:
            from .tdvp_vmf import evolve_tdvp_mu_cmf
            if self.evolve_config.adaptive:
                new_mps = _adaptive_tdvp(evolve_tdvp_mu_cmf, self, mpo, evolve_dt)
            else:
                new_mps = evolve_tdvp_mu_cmf(self, mpo, evolve_dt)
        else:
            raise NotImplementedError(f"{method} is not implemented in the MI355X engine yet (TDVP-PS is)")
        if normalize:
            if np.iscomplex(evolve_dt):
                new_mps.normalize("mps_and_coeff")
            else:
                new_mps.normalize("mps_only")
        return new_mps

    def _evolve_tdvp_ps(self, mpo, evolve_dt) -> "Mps":
        """One-site TDVP with projector splitting (see ``_evolve_tdvp_ps_sweeps``).  The block QR of tall centres runs
        through the engine's Cholesky-QR kernels, which cannot decide rank-deficient / extremely ill-conditioned blocks
        (the 1e-10 padding of ``expand_bond_dimension`` in the first steps of a run); reading their breakdown flag
        back after every decomposition would stall the host exactly where it should be enqueueing the next local solve
        (measured: ~80 us of idle GPU per decomposition).  The step is therefore run OPTIMISTICALLY: the flag is a sticky
        device word read once at the end, and a step in which any decomposition broke down is discarded and repeated
        with every decomposition verified (Householder where needed) - ``self`` is untouched until the step returns.
        ``MPSE_QR_OPTIMISTIC=0``: verify every decomposition as it happens."""
        eng = get_engine()
        if os.environ.get("MPSE_QR_OPTIMISTIC", "1") == "0" or os.environ.get("MPSE_DEFER", "1") == "0":
            return self._evolve_tdvp_ps_sweeps(mpo, evolve_dt, learn_qr=True)
        eng.block_qr_optimistic(True)
        try:
            new = self._evolve_tdvp_ps_sweeps(mpo, evolve_dt)
            failed = eng.block_qr_check()
        except Exception:
            # a decomposition that broke down leaves something that is not an isometry (bad pivots are replaced by 1,
            # tiny ones give factors ~1e150, later values may be inf / NaN): what follows may fail in many ways - engine
            # status codes, LinAlgError, but also the plain assertions of the Krylov solver (`nrmv > 0`) or of the
            # environment bookkeeping.  Only if the flag is up is the failure the optimistic mode's own: then the step
            # is repeated verified; anything else is raised as it is
            if not eng.block_qr_check():
                raise
            failed = True
        finally:
            eng.block_qr_optimistic(False)
        if failed:
            clear_evolve_cache()
            _OPTIMISTIC_REDONE[0] += 1
            # the verified run notes WHICH sites sent the Cholesky-QR path back to the Householder kernels: the next
            # steps decompose those sites by Householder from the start (the rank-deficient blocks next to the chain
            # ends stay rank deficient for several steps - without the note every one of those steps ran twice)
            new = self._evolve_tdvp_ps_sweeps(mpo, evolve_dt, learn_qr=True)
        return new

    def _evolve_tdvp_ps_sweeps(self, mpo, evolve_dt, learn_qr=False) -> "Mps":
        """One-site TDVP with projector splitting, PhysRevB 94, 165116; order of operations of
        mps/mps.py:1267-1404: two half sweeps; per site a forward step -i dt/2 of the centre
        tensor (Lanczos), QR/RQ by quantum-number block, one environment update, a backward
        step +i dt/2 of the bond factor (0-site Lanczos), absorbed into the next site; the last
        site of each half sweep is not split."""
        cfg = self.evolve_config
        eng = get_engine()
        if np.iscomplex(evolve_dt):
            mps = self.copy()
        else:
            mps = self.to_complex()
        evolve_dt = complex(evolve_dt)
        n = len(mps)
        # Only the environments ahead of the sweep are needed: the reference builds both
        # directions and discards half (mps.py:1281-1283).  The second half sweep of the previous step left exactly
        # those environments behind, computed from the very site tensors this step starts with: they are taken over
        # when every tensor they depend on is still the same object (site buffers are immutable by convention).
        ahead = "R" if mps.to_right else "L"
        environ = _carried_environ(mps, mpo, ahead)
        if environ is None:
            environ = Environ(mps, mpo, ahead)
        local_steps = []
        q = len(mps.qntot)
        hh_sites = _householder_sites(mps)         # (site, direction) whose block QR broke the Cholesky-QR path before
        hh_sites.tick()                            # one evolve less on the Householder kernels for every noted site
        qr_redone = [eng.block_qr_stats()[2]]

        def note_qr(imps):
            if learn_qr:
                now = eng.block_qr_stats()[2]
                if now != qr_redone[0]:
                    hh_sites.note((imps, mps.to_right))
                qr_redone[0] = now

        use_cmask = os.environ.get("MPSE_CENTRE_MASK", "1") != "0"

        def prepare(imps, shape):
            """Everything the update of site ``imps`` needs that does not depend on the preceding solve: the
            effective-Hamiltonian descriptor and the quantum-number block plan of its QR.  Called BEFORE the solve whose
            result the site waits for, so that the host has nothing left to do but enqueue when that solve returns."""
            hop = hop_expr(environ.read("L", imps - 1), environ.read("R", imps + 1), [mpo.device(imps, eng)], shape)
            split = (not mps.to_right and imps != 0) or (mps.to_right and imps != n - 1)
            qnbigl = qnbigr = plan = None
            if split:
                qnbigl, qnbigr, _ = mps._get_big_qn([imps], need_mat=False)
                plan = svd_qn.block_plan(qnbigl, qnbigr, mps.qntot)
                # large centres: the engine skips the empty tiles of the Krylov vectors by the pattern of their
                # quantum numbers instead of scanning every vector (row / column grouping as the QR sees the site)
                if use_cmask and len(shape) == 3 and shape[0] * shape[0] * shape[1] * shape[2] >= (1 << 22):
                    if mps.to_right:
                        ql, qr = qnbigl, qnbigr
                    else:               # sweeping left the QR groups the site as (a | sigma, b): regroup (a, sigma | b)
                        ql = add_outer(np.array(mps.qn[imps]), mps._get_sigmaqn(imps))
                        qr = np.array(mps.qn[imps + 1])
                    hop.cmask = centre_tile_mask(eng, ql, qr, mps.qntot, shape)
            return hop, split, qnbigl, qnbigr, plan

        def split_site(imps, centre, ready, shape):
            """What follows the forward step of a split site, on ``centre`` (the result of that step, or the buffer
            that will receive it): QR / RQ by quantum-number block, installation of the isometry, one environment
            update.  Returns the effective Hamiltonian of the bond factor, the bond factor and the site it goes to."""
            hop, _, qnbigl, qnbigr, plan = ready
            l_array, r_array = hop.l, hop.r
            u, qnlset, v, qnrset = svd_qn.svd_qn(centre, qnbigl, qnbigr, mps.qntot, QR=True,
                                                 system="L" if mps.to_right else "R", full_matrices=False, plan=plan,
                                                 householder=(imps, mps.to_right) in hh_sites)
            vt = v.T
            if not mps.to_right:
                mps[imps] = vt.reshape([-1] + shape[1:])
                mps.qn[imps] = np.array(qnrset, dtype=int).reshape(-1, q)
                mps.qnidx = imps - 1
                r_array = environ.GetLR("R", imps, mps, mpo, itensor=r_array, method="System", canonical=True)
                hop_b = hop_expr(l_array, r_array, [], u.shape)
                if use_cmask and u.shape[0] * u.shape[1] >= (1 << 14):
                    # structural tile mask of the bond factor (rows: the left bond, columns: the new bond's states)
                    hop_b.cmask = centre_tile_mask(eng, qnbigl, np.array(qnrset, dtype=int).reshape(-1, q), mps.qntot,
                                                   u.shape)
                return hop_b, u, imps - 1
            mps[imps] = u.reshape(shape[:-1] + [-1])
            mps.qn[imps + 1] = np.array(qnlset, dtype=int).reshape(-1, q)
            mps.qnidx = imps + 1
            l_array = environ.GetLR("L", imps, mps, mpo, itensor=l_array, method="System", canonical=True)
            hop_b = hop_expr(l_array, r_array, [], vt.shape)
            if use_cmask and vt.shape[0] * vt.shape[1] >= (1 << 14):
                hop_b.cmask = centre_tile_mask(eng, np.array(qnlset, dtype=int).reshape(-1, q), qnbigr, mps.qntot,
                                               vt.shape)
            return hop_b, vt, imps + 1

        def absorb(bond, nbr):
            """the evolved bond factor times the neighbouring site: the next centre"""
            t = mps[nbr]
            if mps.to_right:
                return eng.matmul(bond, t.reshape(t.shape[0], -1)).reshape((bond.shape[0],) + t.shape[1:])
            return eng.matmul(t.reshape(-1, t.shape[-1]), bond).reshape(t.shape[:-1] + (bond.shape[1],))

        # The calls that follow a local solve do not depend on WHEN it converges, only on the buffer its result lands
        # in: they are recorded ahead (Engine.recording) and issued by the engine the moment the solve has been enqueued
        # to its end, so the GPU does not wait for the host language between a solve and its QR / absorption (56 us of
        # idle time after every site solve otherwise).  Same tensors bit for bit; +1.1 % on the headline run since the
        # contraction launches were rebalanced (neutral before: DESIGN.md section 5).  MPSE_DEFER=0: the plain loop.
        same_dtype = mps.is_complex or (evolve_dt.real == 0 and not mpo.is_complex)
        pipelined = (cfg.ivp_solver == "krylov" and same_dtype and os.environ.get("MPSE_DEFER", "1") != "0"
                     and not _lib.VERIFY_UNIT)
        try:
            for _ in range(2):
                order = list(mps.iter_idx_list(full=True))
                centre = mps[order[0]]
                ready = prepare(order[0], list(centre.shape))
                after_site = None
                if pipelined and ready[1]:
                    out = eng.empty(centre.shape, centre.dtype)
                    with eng.recording(0):
                        after_site = (out,) + split_site(order[0], out, ready, list(centre.shape))
                for imps in order:
                    shape = list(centre.shape)
                    hop, split = ready[0], ready[1]
                    if not split:
                        mps_t, j = _local_propagate(cfg, hop, -1j * evolve_dt / 2, centre)
                        local_steps.append(j)
                        mps[imps] = mps_t.reshape(shape)
                        continue
                    if not pipelined:
                        mps_t, j = _local_propagate(cfg, hop, -1j * evolve_dt / 2, centre)
                        local_steps.append(j)
                        hop_b, bond, nbr = split_site(imps, mps_t, ready, shape)
                        note_qr(imps)
                        ready = prepare(nbr, list(mps[nbr].shape[:-1]) + [bond.shape[1]] if not mps.to_right
                                        else [bond.shape[0]] + list(mps[nbr].shape[1:]))
                        b_t, j = _local_propagate(cfg, hop_b, 1j * evolve_dt / 2, bond)
                        local_steps.append(j)
                        centre = mps[nbr] = absorb(b_t.reshape(bond.shape), nbr)
                        continue
                    out, hop_b, bond, nbr = after_site
                    eng.arm(0)
                    _, j = expm_krylov(hop, -1j * evolve_dt / 2, centre, out=out)       # + QR, environment update
                    local_steps.append(j)
                    note_qr(imps)
                    b_out = eng.empty(bond.shape, bond.dtype)
                    with eng.recording(1):
                        new_centre = absorb(b_out, nbr)
                    mps[nbr] = new_centre
                    ready = prepare(nbr, list(new_centre.shape))
                    after_site = None
                    if ready[1]:
                        out = eng.empty(new_centre.shape, new_centre.dtype)
                        with eng.recording(0):
                            after_site = (out,) + split_site(nbr, out, ready, list(new_centre.shape))
                    eng.arm(1)
                    _, j = expm_krylov(hop_b, 1j * evolve_dt / 2, bond, out=b_out)        # + absorption
                    local_steps.append(j)
                    centre = new_centre
                mps._switch_direction()
        except BaseException:
            eng.defer_discard()
            raise
        mps.evolve_config.stat = dict(nobs=len(local_steps), min=int(np.min(local_steps)),
                                      max=int(np.max(local_steps)), mean=float(np.mean(local_steps)),
                                      steps=list(local_steps))
        _carry_environ(mps, mpo, environ)
        return mps


# steps the optimistic block QR had to repeat (diagnostics; bench.py reports it)
_OPTIMISTIC_REDONE = [0]

# Per state (carried from an Mps to the states evolved from it, like evolve_config; copied by metacopy): the (site, sweep
# direction) pairs whose block QR sent the Cholesky-QR path back to the Householder kernels in a verified run.  Which
# kernels decompose a site therefore depends on the state object alone, not on what else the calling thread evolved
# before (rounds 4-5 kept the notes in a thread-local keyed by the chain's shape).  A stale entry costs a slower
# decomposition, never correctness; the notes are not written to checkpoints.


class _QrNotes:
    """(site, direction) -> [evolves left on the Householder kernels, patience].  A noted site goes back to the Cholesky-QR
    path after ``patience`` evolves (the rank-deficient blocks of a freshly expanded state fill up within a few steps:
    profiles/r05_qr_trips.md); if it breaks down again its patience doubles."""
    FIRST = 8

    def __init__(self):
        self.d = {}

    def copy(self):
        new = _QrNotes()
        new.d = {k: list(v) for k, v in self.d.items()}
        return new

    def __contains__(self, key):
        e = self.d.get(key)
        return e is not None and e[0] > 0

    def note(self, key):
        e = self.d.get(key)
        patience = min(2 * e[1], 1 << 20) if e else self.FIRST
        self.d[key] = [patience, patience]

    def tick(self):
        for e in self.d.values():
            if e[0] > 0:
                e[0] -= 1


def _householder_sites(mps):
    notes = mps.__dict__.get("_qr_notes")
    if notes is None:
        notes = mps._qr_notes = _QrNotes()
    return notes

# One slot per host thread (= per trajectory): the environments ahead of the next half sweep, as the last TDVP-PS step
# left them, with the objects they were computed from.  Bounded: a new step replaces the slot.
_CARRY = threading.local()


def _carry_environ(mps, mpo, environ):
    ahead = "R" if mps.to_right else "L"
    environ.drop("L" if ahead == "R" else "R")
    # the take-over test compares the MPO sites as OBJECTS and by their content versions (Mpo.site_version: a host
    # snapshot per site): a site replaced (as try_swap_site does) or edited in place (a time-dependent Hamiltonian) drops
    # the carried environments AND the cached device copy of that site - the caller's arrays are left as they are
    sites = list(mpo._mp) if hasattr(mpo, "_mp") else None
    _CARRY.slot = (mpo, sites, ahead, environ, list(mps._mp), _mpo_fingerprint(mpo, sites))


def _mpo_fingerprint(mpo, sites):
    if sites is None:
        return None
    return mpo.versions() if hasattr(mpo, "versions") else tuple(id(w) for w in sites)


def clear_evolve_cache():
    """Drop the environments the calling thread's last TDVP-PS step left for the next one (they pin one set of
    environments and the site list in HBM until that thread evolves again).  ``MPSE_ENV_CARRY=0`` disables the
    take-over altogether."""
    _CARRY.slot = None


Mps.clear_evolve_cache = staticmethod(clear_evolve_cache)


def _carried_environ(mps, mpo, ahead):
    slot = getattr(_CARRY, "slot", None)
    _CARRY.slot = None
    if slot is None or os.environ.get("MPSE_ENV_CARRY", "1") == "0":
        return None
    cmpo, cmpo_sites, cahead, environ, csites, cprint = slot
    n = len(mps)
    if cmpo is not mpo or cmpo_sites is None or cahead != ahead or len(csites) != n or len(cmpo_sites) != n:
        return None
    if any(a is not b for a, b in zip(mpo._mp, cmpo_sites)) or _mpo_fingerprint(mpo, list(mpo._mp)) != cprint:
        return None
    # R(i) depends on the sites i .. n-1 (needed for i >= 1), L(i) on 0 .. i (needed for i <= n-2); the centre site
    # (rescaled by normalize) is in neither
    sites = range(1, n) if ahead == "R" else range(0, n - 1)
    if any(mps._mp[i] is not csites[i] for i in sites):
        return None
    return environ


def _local_propagate(config, hop, factor, y):
    """exp(factor * H_eff) y for one centre tensor: the engine's Lanczos exponential (``ivp_solver="krylov"``, the
    default), or an explicit scheme named by ``ivp_solver`` with ``ivp_rtol / ivp_atol`` as in mps/mps.py:1299-1315:
    dy/dt = (factor / |factor|) H y over (0, |factor|).  "RK45" runs on device-resident vectors (lib/rk45.py, scipy's
    step-size rules on the host); "RK23" / "DOP853" go through ``scipy.integrate.solve_ivp`` with every H y on the
    device.  Returns (tensor, number of H y)."""
    if config.ivp_solver == "krylov":
        return expm_krylov(hop, factor, y)
    eng = get_engine()
    span = abs(factor)
    phase = complex(factor) / span
    phase = phase.real if phase.imag == 0 else phase
    cplx = np.iscomplexobj(phase) or hop.operator_is_complex
    if config.ivp_solver == "RK45":
        # Dormand-Prince on the device (lib/rk45.py): state, stages and error estimate stay in HBM, the step control
        # follows scipy's rules
        from ..lib.rk45 import solve_rk45
        y0 = y.to_complex() if cplx else y

        def rhs_dev(t, v):
            out = hop(v)
            return out.scale_(phase) if phase != 1.0 else out

        res, nfev, _ = solve_rk45(rhs_dev, span, y0, rtol=config.ivp_rtol, atol=config.ivp_atol)
        return res, nfev
    from scipy.integrate import solve_ivp
    y0 = y.to_host()
    shape = y0.shape
    if cplx:
        y0 = y0.astype(complex)

    def rhs(t, v):
        return hop(eng.asdevice(v.reshape(shape))).to_host().ravel() * phase

    sol = solve_ivp(rhs, (0, span), y0.ravel(), method=config.ivp_solver, rtol=config.ivp_rtol, atol=config.ivp_atol)
    if not sol.success:
        raise RuntimeError(f"solve_ivp({config.ivp_solver}) failed on a local TDVP problem: {sol.message}")
    return eng.asdevice(np.ascontiguousarray(sol.y[:, -1].reshape(shape))), sol.nfev


def _evolve_tdvp_ps2(self, mpo, evolve_dt) -> "Mps":
    """Two-site TDVP with projector splitting, mps/mps.py:1406-1517: forward step -i dt/2 of the two-site
    centre, ``_update_mps`` (block SVD + truncation), environment update, backward step +i dt/2 of the
    next one-site centre, ``_push_cano``; the last pair of each half sweep is not stepped back."""
    cfg = self.evolve_config
    eng = get_engine()
    mps = self.copy() if np.iscomplex(evolve_dt) else self.to_complex()
    evolve_dt = complex(evolve_dt)
    n = len(mps)
    environ = Environ(mps, mpo, "R" if mps.to_right else "L")
    local_steps = []
    for _ in range(2):
        for imps in mps.iter_idx_list(full=False):
            if mps.to_right:
                lidx, c0, c1, ridx = imps - 1, imps, imps + 1, imps + 2
                c2, last_idx = c1, n - 2
            else:
                lidx, c0, c1, ridx = imps - 2, imps - 1, imps, imps + 1
                c2, last_idx = c0, 1
            l_array = environ.read("L", lidx)
            r_array = environ.read("R", ridx)
            a, b = mps[c0], mps[c1]
            ms2 = eng.matmul(a.reshape(-1, a.shape[-1]), b.reshape(b.shape[0], -1)).reshape(a.shape[:-1] + b.shape[1:])
            hop = hop_expr(l_array, r_array, [mpo.device(c0, eng), mpo.device(c1, eng)], ms2.shape)
            mps_t, j = _local_propagate(cfg, hop, -1j * evolve_dt / 2, ms2)
            local_steps.append(j)
            qnbigl, qnbigr, _ = mps._get_big_qn([c0, c1], need_mat=False)
            mps._update_mps(mps_t.reshape(ms2.shape), [c0, c1], qnbigl, qnbigr)
            if mps.compress_config.ofs is not None:
                mpo.try_swap_site(mps.model, mps.compress_config.ofs_swap_jw)
            if imps == last_idx:
                continue
            if mps.to_right:
                l_array = environ.GetLR("L", lidx + 1, mps, mpo, itensor=l_array, method="System")
            else:
                r_array = environ.GetLR("R", ridx - 1, mps, mpo, itensor=r_array, method="System")
            ms1 = mps[c2]
            hop1 = hop_expr(l_array, r_array, [mpo.device(c2, eng)], ms1.shape)
            mps_b, j = _local_propagate(cfg, hop1, 1j * evolve_dt / 2, ms1)
            local_steps.append(j)
            mps[c2] = mps_b.reshape(ms1.shape)
            mps._push_cano(c2)
        mps._switch_direction()
    mps.evolve_config.stat = dict(nobs=len(local_steps), min=int(np.min(local_steps)), max=int(np.max(local_steps)),
                                  mean=float(np.mean(local_steps)), steps=list(local_steps))
    return mps


Mps._evolve_tdvp_ps2 = _evolve_tdvp_ps2


# Taylor coefficients of exp(x) up to 4th order (utils/rk.py "C_RK4" tableau used by P&C, configs.py:364-369)
_TAYLOR4 = [1.0, 1.0, 1.0 / 2, 1.0 / 6, 1.0 / 24]


def _evolve_prop_and_compress(self, mpo, evolve_dt) -> "Mps":
    """Global propagation & compression with a Taylor propagator (mps/mps.py:794-885): terms H^k|psi> by contract
    (apply, canonicalise, compress), scaled by (-i dt)^k / k! and summed with compression.  Adaptive mode: the
    distance between the sums with and without the last term estimates the local error and sets the next step
    through p = (rtol / relative distance)^(1/order), clipped to [0.1, 2]; p < 0.5 repeats the sub-step."""
    import math
    from ..utils import CompressCriteria
    config = self.evolve_config
    coeff = [1.0 / math.factorial(k) for k in range(config.taylor_order + 1)]
    order = len(coeff) - 1
    termlist = [self]
    orig = self.compress_config
    tmp = self.compress_config.copy()
    if tmp.criteria is CompressCriteria.threshold:
        tmp.criteria = CompressCriteria.both        # the bonds must not grow while contracting
    self.compress_config = tmp
    while len(termlist) < len(coeff):
        termlist.append(mpo.contract(termlist[-1]))
    self.compress_config = orig
    for t in termlist:
        t.compress_config = orig
    if not config.adaptive:
        return compressed_sum([t.scale((-1.0j * evolve_dt) ** k * coeff[k]) for k, t in enumerate(termlist)])
    config.check_valid_dt(evolve_dt)
    p_restart, p_min, p_max = 0.5, 0.1, 2.0
    while True:
        dt = _min_abs(config.guess_dt, evolve_dt)
        scaled = [t.scale((-1.0j * dt) ** k * coeff[k]) for k, t in enumerate(termlist)]
        new_mps1 = compressed_sum(scaled[:-1])
        new_mps2 = compressed_sum([new_mps1, scaled[-1]])
        dis = new_mps1.distance(new_mps2)
        p = (config.adaptive_rtol / (dis / new_mps2.mp_norm + 1e-30)) ** (1.0 / order)
        if np.allclose(dt, evolve_dt):
            if p < p_restart:                               # the last sub-step is not accurate enough: repeat it
                config.guess_dt = dt * max(p_min, p)
            else:
                new_mps2.evolve_config.guess_dt = _min_abs(dt * p, config.guess_dt)
                return new_mps2
        elif p < p_restart:
            config.guess_dt *= max(p_min, p)
        else:
            config.guess_dt *= min(p, p_max)
            new_mps2.evolve_config.guess_dt = config.guess_dt
            return new_mps2._evolve_prop_and_compress(mpo, evolve_dt - dt)


Mps._evolve_prop_and_compress = _evolve_prop_and_compress


def _as_mpo_of_t(mpo):
    """A fixed MPO or a callable t -> MPO (t measured from the start of the step; the stage state is offered as the
    keyword ``mps``), mps/mps.py:669-676"""
    if callable(mpo):
        return mpo
    if not hasattr(mpo, "contract"):
        raise TypeError(f"unsupported mpo type: {mpo}")
    return lambda t, *args, **kwargs: mpo


def _evolve_prop_and_compress_tdrk4(self, mpo, evolve_dt) -> "Mps":
    """Classical fourth-order Runge-Kutta step of i dpsi/dt = H(t) psi with compression after every stage
    (mps/mps.py:664-699): k_i = -i H(t_i) y_i by contract, stage states y + a k dt canonicalised and compressed,
    and the final combination y + dt (k1 + 2 k2 + 2 k3 + k4) / 6 as one compressed sum."""
    mpo_t = _as_mpo_of_t(mpo)

    def stage(k, weight):
        y = self.add(k.scale(weight * evolve_dt))
        y.canonicalise().compress()
        return y

    k1 = mpo_t(0).contract(self).scale(-1j)
    k2 = mpo_t(0.5 * evolve_dt).contract(stage(k1, 0.5)).scale(-1j)
    k3 = mpo_t(0.5 * evolve_dt).contract(stage(k2, 0.5)).scale(-1j)
    k4 = mpo_t(evolve_dt).contract(stage(k3, 1.0)).scale(-1j)
    return compressed_sum([self, k1.scale(1 / 6 * evolve_dt), k2.scale(2 / 6 * evolve_dt), k3.scale(2 / 6 * evolve_dt),
                           k4.scale(1 / 6 * evolve_dt)])


def _evolve_prop_and_compress_tdrk(self, mpo, evolve_dt) -> "Mps":
    """Explicit Runge-Kutta step with the tableau of ``evolve_config.rk_config`` for fixed or time-dependent H
    (mps/mps.py:701-792).  Adaptive mode needs an embedded pair: the difference of the two weight rows applied to
    the stage derivatives is the error estimate, p = (rtol / relative error)^(1/order) clipped to [0.1, 2] scales the
    next sub-step and p < 0.5 repeats the current one."""
    mpo_t = _as_mpo_of_t(mpo)
    config = self.evolve_config
    rk = config.rk_config
    a, b, c = rk.tableau

    def sub_step(y, tau, t0):
        ks = []
        for s in range(rk.stage):
            ys = compressed_sum([y] + [ks[i].scale(a[s, i] * tau) for i in range(s) if a[s, i] != 0], batchsize=6)
            ks.append(mpo_t(c[s] * tau + t0, mps=ys).contract(ys).scale(-1j))
        new = compressed_sum([y] + [ks[s].scale(b[0, s] * tau) for s in range(rk.stage) if b[0, s] != 0], batchsize=6)
        if not config.adaptive:
            assert len(rk.order) == 1
            return new, 0.0
        assert len(rk.order) == 2 and rk.order[0] - rk.order[1] == 1
        err = None
        for s in range(rk.stage):
            if np.allclose(b[0, s], b[1, s]):
                continue
            term = ks[s].scale((b[0, s] - b[1, s]) * tau)
            err = term if err is None else err.add(term)
        return new, err.mp_norm / new.mp_norm

    config.check_valid_dt(evolve_dt)
    if not config.adaptive:
        return sub_step(self, evolve_dt, 0)[0]
    p_restart, p_min, p_max = 0.5, 0.1, 2.0
    evolved, new = 0, self
    while True:
        cfg = new.evolve_config
        dt = _min_abs(cfg.guess_dt, evolve_dt - evolved)
        cand, error = sub_step(new, dt, evolved)
        p = (cand.evolve_config.adaptive_rtol / (error + 1e-30)) ** (1 / rk.order[0])
        if p < p_restart:
            # mps/mps.py:765-770: the rejected candidate replaces the state all the same; the next trial starts from it
            new = cand
            new.evolve_config.guess_dt = dt * max(p_min, p)
            continue
        new = cand
        if np.allclose(dt + evolved, evolve_dt):
            new.evolve_config.guess_dt = _min_abs(dt * p, new.evolve_config.guess_dt)
            return new
        new.evolve_config.guess_dt *= min(p, p_max)
        evolved += dt


Mps._evolve_prop_and_compress_tdrk4 = _evolve_prop_and_compress_tdrk4
Mps._evolve_prop_and_compress_tdrk = _evolve_prop_and_compress_tdrk


def _min_abs(t1, t2):
    return t1 if abs(t1) < abs(t2) else t2


def _adaptive_tdvp(fun, cur_mps, mpo, evolve_target_t):
    """Step-size control of the projector-splitting integrators (mps/mps.py:46-115, J. Chem. Phys. 146, 174107):
    a step dt is compared with two steps dt/2; the O(dt^3) splitting error sets the next step through
    p = (0.75 rtol / relative distance)^(1/3), clipped to [0.1, 2]; p < 0.5 rejects the step."""
    config = cur_mps.evolve_config.copy()
    config.check_valid_dt(evolve_target_t)
    p_restart, p_min, p_max = 0.5, 0.1, 2.0
    evolved_t = 0
    while True:
        dt = _min_abs(config.guess_dt, evolve_target_t - evolved_t)
        half1 = fun(cur_mps, mpo, dt / 2)
        half2 = fun(half1, mpo, dt / 2)
        full = fun(cur_mps, mpo, dt)
        dis = full.distance(half2)
        del half1, full
        p = (0.75 * config.adaptive_rtol / (dis / half2.mp_norm + 1e-30)) ** (1.0 / 3)
        p = min(max(p, p_min), p_max)
        if p < p_restart:
            config.guess_dt = dt * p
            continue
        evolved_t += dt
        if np.allclose(evolved_t, evolve_target_t):
            half2.evolve_config.guess_dt = config.guess_dt
            return half2
        config.guess_dt *= p
        cur_mps = half2


def compressed_sum(mps_list, batchsize=5, temp_m_trunc=None):
    """mps/lib.py:417-439"""
    from collections import deque
    assert len(mps_list) != 0
    queue = deque(mps_list)
    if len(queue) == 1:
        new = mps_list[0].canonicalise()
        new.compress(temp_m_trunc=temp_m_trunc)
        return new
    while len(queue) != 1:
        terms = [queue.popleft() for _ in range(min(batchsize, len(queue)))]
        s = terms[0]
        for t in terms[1:]:
            s = s.add(t)
        s.canonicalise()
        s.compress(temp_m_trunc=temp_m_trunc)
        queue.append(s)
    return queue[0]
