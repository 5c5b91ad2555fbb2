"""Multi-GPU execution model: independent trajectories, one process per GPU.

A TDVP/DMRG sweep is strictly sequential in the site index, so nothing is sharded inside a sweep.
Independent units (trajectories, disorder realisations, parameter points) are dealt round-robin to
ranks; there is no collective on the data path.  At the end every rank contributes its observable
rows to one all_gather (RCCL over xGMI on the GPU box, gloo in CPU tests) - kilobytes, latency bound.
torch.distributed is used for this plumbing only and imported lazily."""
import numpy as np


def trajectory_seed(base_seed: int, unit: int) -> int:
    """Deterministic, distinct RNG seed per independent unit."""
    return int(np.random.SeedSequence([int(base_seed), int(unit)]).generate_state(1)[0])


def units_of_rank(n_units: int, rank: int, world: int):
    return [u for u in range(n_units) if u % world == rank]


def max_over_ranks(value: float, device="cpu") -> float:
    import torch
    import torch.distributed as dist
    t = torch.tensor([value], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_observables(local_rows: np.ndarray, local_units, n_units: int, device="cpu") -> np.ndarray:
    """all_gather of per-unit observable rows; returns the (n_units, nobs) table on every rank."""
    import torch
    import torch.distributed as dist
    local_rows = np.atleast_2d(np.asarray(local_rows, dtype=np.float64))
    nobs = local_rows.shape[1] if local_rows.size else 0
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        out = np.zeros((n_units, nobs))
        out[list(local_units)] = local_rows
        return out
    world = dist.get_world_size()
    meta = torch.tensor([nobs], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX)
    nobs = int(meta.item())
    per_rank = (n_units + world - 1) // world
    buf = torch.full((per_rank, nobs + 1), -1.0, dtype=torch.float64, device=device)
    for k, (u, row) in enumerate(zip(local_units, local_rows)):
        buf[k, 0] = float(u)
        buf[k, 1:] = torch.as_tensor(row, dtype=torch.float64)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = np.zeros((n_units, nobs))
    for part in parts:
        p = part.cpu().numpy()
        for row in p:
            if row[0] >= 0:
                out[int(row[0])] = row[1:]
    return out
