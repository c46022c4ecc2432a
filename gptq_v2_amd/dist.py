"""Multi-GPU layout of the calibration path (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  The reference is single-GPU
(parse_args.py:53-55); this is the only place where ranks exchange data:

  * calibration samples are split over ranks, every rank accumulates its own H, then ONE
    all-reduce(sum) of H (and of n_samples) per input-sharing set of Linears; all ranks then hold
    bit-identical H, so the (deterministic) inversion gives every rank the same perm and U;
  * the rows (output features) of each Linear are split over ranks, rows are independent in the
    column loop; the integer levels, scales and zeros are all-gathered afterwards.

The functions only move tensors; the arithmetic between them is the single-GPU code.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def sample_shard(n_samples: int, rank: int, world_size: int) -> Tuple[int, int]:
    """[begin, end) of the calibration samples rank `rank` accumulates; sizes differ by at most 1."""
    base, extra = divmod(n_samples, world_size)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def row_shard(n_rows: int, rank: int, world_size: int, multiple: int = 64) -> Tuple[int, int]:
    """[begin, end) of the output features rank `rank` quantises.  Boundaries are multiples of
    `multiple` (one wave of the column-loop kernel handles 64 rows; must be a multiple of 4)."""
    units = (n_rows + multiple - 1) // multiple
    base, extra = divmod(units, world_size)
    b = rank * base + min(rank, extra)
    e = b + base + (1 if rank < extra else 0)
    return min(b * multiple, n_rows), min(e * multiple, n_rows)


def allreduce_hessian(hessian: torch.Tensor, n_samples: int) -> int:
    """Sum H over ranks in place and return the global sample count."""
    rank, ws = world()
    if ws == 1:
        return n_samples
    if dist.get_backend() == "nccl":
        dist.all_reduce(hessian, op=dist.ReduceOp.SUM)
        n = torch.tensor([n_samples], dtype=torch.int64, device=hessian.device)
    else:   # gloo (CPU tests, single-GPU rehearsals): through host memory, see allgather_rows
        host = hessian.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        hessian.copy_(host)
        n = torch.tensor([n_samples], dtype=torch.int64)
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return int(n.item())


def allgather_rows(local: Dict[str, torch.Tensor], n_rows: int, multiple: int = 64) -> Dict[str, torch.Tensor]:
    """`local[name]` holds this rank's row slice (dim 0) of a per-row result; returns the full
    tensors, rows in rank order, on every rank."""
    rank, ws = world()
    if ws == 1:
        return local
    bounds = [row_shard(n_rows, r, ws, multiple) for r in range(ws)]
    most = max(e - b for b, e in bounds)
    out: Dict[str, torch.Tensor] = {}
    for name, part in local.items():
        # equal-sized buffers (collectives want them): pad this rank's slice to the largest one
        mine = torch.zeros((most,) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
        mine[: part.shape[0]] = part
        gathered = torch.empty((ws, most) + tuple(part.shape[1:]), dtype=part.dtype, device=part.device)
        if dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(gathered.view(ws * most, *part.shape[1:]), mine)
        else:
            # gloo (CPU tests, single-GPU rehearsals): the list form is the one it implements everywhere,
            # and device tensors go through host memory here (its own device path hung on 22 MB pieces)
            host = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(ws)]
            dist.all_gather(host, mine.cpu())
            for r in range(ws):
                gathered[r].copy_(host[r])
        out[name] = torch.cat([gathered[r, : e - b] for r, (b, e) in enumerate(bounds)], dim=0)
    return out
