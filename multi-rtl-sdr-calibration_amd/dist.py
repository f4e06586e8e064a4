"""Multi-GPU layer: shard independent (dongle, ARFCN) units across ranks and collect the per-unit
result table with ONE all-gather (RCCL over xGMI on the GPU box; gloo in the CPU tests).

The reference has no distributed layer: dongles are matrix columns processed by a serial loop
(gsm_sync_demod.m:112) and the scanner splits the ARFCN list across dongles
(multi_rtl_sdr_gsm_FCCH_scanner.m:60-65).  Units never read each other's data, so the only exchange
step is gathering the table every rank needs for the inter-dongle comparison
(gsm_sync_demod.m:151-158)."""
from __future__ import annotations

import numpy as np


def shard_range(num_units: int, world: int, rank: int):
    """Block-contiguous partition: rank g owns units [g*U//G, (g+1)*U//G)."""
    return (rank * num_units) // world, ((rank + 1) * num_units) // world


def shard_sizes(num_units: int, world: int):
    return [shard_range(num_units, world, r)[1] - shard_range(num_units, world, r)[0] for r in range(world)]


def allgather_table(local_table, num_units: int, group=None):
    """local_table: torch tensor (n_local, C) on this rank's device -> (num_units, C) on every rank,
    rows in global unit order.  One collective; uneven shards are padded to the largest shard."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    sizes = shard_sizes(num_units, world)
    mx = max(sizes)
    c = local_table.shape[1]
    if local_table.shape[0] != sizes[dist.get_rank(group)]:
        raise ValueError("local table does not match this rank's shard")
    send = local_table
    if send.shape[0] != mx:
        pad = torch.full((mx - send.shape[0], c), float("nan"), dtype=send.dtype, device=send.device)
        send = torch.cat([send, pad], dim=0)
    send = send.contiguous()
    out = torch.empty((world * mx, c), dtype=send.dtype, device=send.device)
    dist.all_gather_into_tensor(out, send, group=group)
    if all(s == mx for s in sizes):
        return out
    return torch.cat([out[r * mx: r * mx + sizes[r]] for r in range(world)], dim=0)


def sampling_phase_difference(pos_info_a, pos_info_b):
    """gsm_sync_demod.m:151-158: per-burst start difference between two dongles' pos_info (8x units)."""
    a = np.atleast_2d(np.asarray(pos_info_a, dtype=np.float64))
    b = np.atleast_2d(np.asarray(pos_info_b, dtype=np.float64))
    n = min(len(a), len(b))
    return b[:n, 0] - a[:n, 0]
