"""Dev-container stand-in for `h5py` (absent): the reference's Davidson imports it
at module import but only instantiates File when spilling to disk, which never
happens at the sizes used for golden vectors.  TEST INFRASTRUCTURE ONLY."""


class File:  # pragma: no cover
    def __init__(self, *a, **k):
        raise RuntimeError("h5py shim: out-of-core Davidson is not supported in the oracle tooling")
