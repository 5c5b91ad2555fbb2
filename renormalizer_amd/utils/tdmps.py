"""Time-stepping job skeleton (counterpart of renormalizer/utils/tdmps.py: ``TdMpsJob``).

A job owns the latest state, a list of evolution times and whatever ``process_mps`` accumulates; subclasses provide
``init_mps``, ``evolve_single_step``, ``process_mps`` and ``get_dump_dict``.  After every step the accumulated
arrays are written to ``<dump_dir>/<job_name>.npz`` (same file layout as the reference, so its analysis scripts
read the output); ``dump_mps`` = "one" / "all" also checkpoints the state with ``Mps.dump``."""
import logging
import os
import time

import numpy as np

from .configs import EvolveConfig

logger = logging.getLogger("renormalizer_amd")


class TdMpsJob:
    def __init__(self, evolve_config: EvolveConfig = None, dump_mps=None, dump_dir=None, job_name=None):
        self.evolve_config = EvolveConfig() if evolve_config is None else evolve_config
        if dump_mps not in (None, "all", "one"):
            raise ValueError(f"dump_mps should be None, 'all', 'one'. Got {dump_mps}")
        self.dump_mps = dump_mps
        self.dump_dir = dump_dir
        self.job_name = job_name
        self.info_interval = 1
        self.evolve_times = [0]
        mps = self.init_mps()
        if mps is None:
            raise ValueError("init_mps should return an mps. Got None")
        self.latest_mps = mps
        self.process_mps(mps)

    # ---- to be provided by the job
    def init_mps(self):
        raise NotImplementedError

    def process_mps(self, mps):
        raise NotImplementedError

    def evolve_single_step(self, evolve_dt):
        raise NotImplementedError

    def get_dump_dict(self):
        raise NotImplementedError

    def stop_evolve_criteria(self):
        return False

    # ---- driver
    def evolve(self, evolve_dt=None, nsteps=None, evolve_time=None):
        """Any two of (evolve_dt, nsteps, evolve_time) fix the run; with evolve_dt alone the job runs until
        ``stop_evolve_criteria`` fires; nsteps wins over evolve_time when all three are given."""
        if evolve_dt is None:
            if nsteps is None or evolve_time is None:
                raise ValueError(f"The input parameters evolve_dt:{evolve_dt}, nsteps:{nsteps}, "
                                 f"evolve_time:{evolve_time} do not meet the requirements!")
            evolve_dt = evolve_time / float(nsteps)
        elif nsteps is None:
            nsteps = int(1e10) if evolve_time is None else int(abs(evolve_time) // abs(evolve_dt)) + 1
        t_begin = time.perf_counter()
        for i in range(nsteps):
            if self.stop_evolve_criteria():
                logger.info("Criteria to stop the evolution has met. Stop the evolution")
                break
            t0 = time.perf_counter()
            new_mps = self.evolve_single_step(evolve_dt)
            self.evolve_times.append(self.latest_evolve_time + evolve_dt)
            self.process_mps(new_mps)
            self.latest_mps = new_mps
            logger.info(f"step {len(self.evolve_times) - 1} complete, time cost {time.perf_counter() - t0:.3f} s")
            if self._defined_output_path:
                checkpoint = self.dump_mps if (self.info_interval is not None and i % self.info_interval == 0) else None
                try:
                    self.dump_dict(checkpoint)
                except IOError:                 # a full disk must not kill a long run
                    logger.exception("dumping dict failed with IOError")
        logger.info(f"evolution complete, {time.perf_counter() - t_begin:.3f} s")
        return self

    def dump_dict(self, checkpoint=None):
        if not self._defined_output_path:
            raise ValueError("Dump dir or job name not set")
        os.makedirs(self.dump_dir, exist_ok=True)
        path = os.path.join(self.dump_dir, self.job_name + ".npz")
        bak = path + ".bak"
        if os.path.exists(path):                # keep the previous file until the new one is complete
            if os.path.exists(bak):
                os.remove(bak)
            os.rename(path, bak)
        np.savez(path, **self.get_dump_dict())
        if os.path.exists(bak):
            os.remove(bak)
        if checkpoint is not None:
            tag = f"_mps_{len(self.evolve_times) - 1}" if checkpoint == "all" else "_mps"
            self.latest_mps.dump(os.path.join(self.dump_dir, self.job_name + tag + ".npz"))

    @property
    def latest_evolve_time(self):
        return self.evolve_times[-1]

    @property
    def evolve_times_array(self):
        return np.array(self.evolve_times)

    @property
    def _defined_output_path(self):
        return self.dump_dir is not None and self.job_name is not None
