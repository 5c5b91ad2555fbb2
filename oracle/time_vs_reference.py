#!/usr/bin/env python3
"""DEV CONTAINER ONLY (imports /root/reference): wall time of one TDVP-PS `evolve` of the reference next to the same
evolve of the oracle (oracle/mps_oracle.py), same state, same 4 BLAS threads - the cross-check BASELINE.md section 3
item 4 asks for (is the CPU stand-in timed on the GPU box representative of the real reference?).

    cd /tmp && python /root/repo/oracle/time_vs_reference.py [nmol=10] [D=64]
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
