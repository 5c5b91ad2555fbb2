#!/usr/bin/env python3
"""DEV CONTAINER ONLY (imports /root/reference): wall time of one TDVP-PS `evolve` of the reference next to the same
evolve of the oracle (oracle/mps_oracle.py), same state, same 4 BLAS threads - the cross-check BASELINE.md section 3
item 4 asks for (is the CPU stand-in timed on the GPU box representative of the real reference?).

    cd /tmp && python /root/repo/oracle/time_vs_reference.py [nmol=10] [D=64]
    cd /tmp && python /root/repo/oracle/time_vs_reference.py site [Dl=256] [d=16] [Dr=256] [w=5]

``site``: ONE local update at the headline shape - the effective Hamiltonian of a (Dl, d, Dr) centre between random
environments with MPO bond w, propagated by the Krylov exponential - through the reference's hop_expr + expm_krylov
and through the oracle's hop_apply + expm_krylov on the same arrays: the shape the cpu_baseline of bench.py is timed
at, where the whole-evolve comparison above would take the reference several minutes.
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "RENO_NUM_THREADS"):
    os.environ[v] = "4"
os.environ.setdefault("RENO_LOG_LEVEL", "40")
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(HERE, "shims"))
sys.path.insert(0, os.path.dirname(HERE))

from renormalizer.model import Phonon, Mol, HolsteinModel  # noqa: E402
from renormalizer.mps import Mps, Mpo  # noqa: E402
from renormalizer.utils import Quantity, CompressConfig, CompressCriteria, EvolveConfig, EvolveMethod  # noqa: E402
import numpy as np  # noqa: E402
from oracle import mps_oracle as orc  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "site":
    from renormalizer.mps.hop_expr import hop_expr  # noqa: E402
    from renormalizer.lib import expm_krylov  # noqa: E402
    from renormalizer.mps.backend import xp  # noqa: E402
    Dl, d, Dr, wb = [int(a) for a in (sys.argv[2:6] + ["256", "16", "256", "5"][len(sys.argv) - 2:])]
    rng = np.random.default_rng(7)

    def herm_env(D):
        """(D, w, D) environment of a Hermitian operator: channel-wise A + A^H"""
        a = rng.standard_normal((D, wb, D)) + 1j * rng.standard_normal((D, wb, D))
        return (a + a.conj().transpose(2, 1, 0)) / np.sqrt(D)

    ltensor, rtensor = herm_env(Dl), herm_env(Dr)
    mo = rng.standard_normal((wb, d, d, wb))
    mo = (mo + mo.transpose(0, 2, 1, 3)) / np.sqrt(d * wb)
    c = rng.standard_normal((Dl, d, Dr)) + 1j * rng.standard_normal((Dl, d, Dr))
    c /= np.linalg.norm(c)
    dt = -0.05j
    hop = hop_expr(ltensor.copy(), rtensor.copy(), [mo.copy()], c.shape)
    hop(c)                                              # first call: contraction path search, BLAS thread start-up
    orc.hop_apply(ltensor, rtensor, [mo], c)
    t0 = time.perf_counter()
    ref, j_ref = expm_krylov(lambda x: hop(x.reshape(c.shape)).ravel(), dt, xp.asarray(c.ravel()))
    t_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    out, j_orc = orc.expm_krylov(lambda x: orc.hop_apply(ltensor, rtensor, [mo], x.reshape(c.shape)).ravel(), dt,
                                 c.ravel(), return_stat=True)
    t_orc = time.perf_counter() - t0
    err = np.abs(np.asarray(ref) - out).max()
    print(f"site update ({Dl}, {d}, {Dr}), MPO bond {wb}, complex128, 4 BLAS threads, {os.cpu_count()} vCPUs")
    print(f"reference {t_ref:.2f} s (Krylov dim {j_ref})   oracle {t_orc:.2f} s (Krylov dim {j_orc})   "
          f"ratio oracle/reference {t_orc / t_ref:.2f}   max |difference| {err:.1e}")
    sys.exit(0)

nmol = int(sys.argv[1]) if len(sys.argv) > 1 else 10
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
ph = Phonon.simple_phonon(Quantity(6.128e-3), Quantity(16.274571056529368), 16)
model = HolsteinModel([Mol(Quantity(0), [ph])] * nmol, Quantity(3.0e-2), 3)
init = Mpo.onsite(model, r"a^\dagger", dof_set={nmol // 2}).apply(Mps.ground_state(model, False))
mpo = Mpo(model, offset=Quantity(init.expectation(Mpo(model))))
init.compress_config = CompressConfig(CompressCriteria.fixed, max_bonddim=D)
init.evolve_config = EvolveConfig(EvolveMethod.tdvp_ps)
np.random.seed(1)
mps = init.expand_bond_dimension(mpo)
mps.canonicalise()
mps = mps.evolve(mpo, 10.0)                      # one step so that both codes start from an evolved state
st = orc.MpsState([np.asarray(m.array) for m in mps], [np.asarray(q).reshape(len(q), -1) for q in mps.qn], mps.qnidx,
                  np.asarray(mps.qntot), mps.to_right, [np.array(b.sigmaqn) for b in model.basis])
w = [np.asarray(m.array) for m in mpo]
t0 = time.perf_counter()
ref = mps.evolve(mpo, 10.0)
t_ref = time.perf_counter() - t0
t0 = time.perf_counter()
st2 = orc.tdvp_ps_step(st, w, 10.0)
t_orc = time.perf_counter() - t0
occ_ref = np.asarray(ref.e_occupations)
from renormalizer.model import Op  # noqa: E402
occ_ops = [Mpo(model, Op(r"a^\dagger a", dof)) for dof in model.e_dofs]
occ_orc = np.array([orc.expectation(st2.sites, [np.asarray(m.array) for m in o]) for o in occ_ops]).real
print(f"Holstein {2 * nmol} sites, D = {D}, d = 2/16, TDVP-PS, 4 BLAS threads, {os.cpu_count()} vCPUs")
print(f"reference evolve {t_ref:.2f} s   oracle evolve {t_orc:.2f} s   ratio oracle/reference {t_orc / t_ref:.2f}")
print(f"max |occupation difference| {np.abs(occ_ref - occ_orc).max():.2e}; mean Krylov dim (reference) {ref.evolve_config.stat.mean:.2f}")
