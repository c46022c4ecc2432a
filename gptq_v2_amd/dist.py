"""Multi-GPU layout of the calibration path (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on the GPU box, "gloo" in the CPU tests).  The reference is single-GPU
(parse_args.py:53-55); this is the only place where ranks exchange data:

  * calibration samples are split over ranks, every rank accumulates its own H, then ONE
    sum of H (and of n_samples) over ranks per input-sharing set of Linears -- the packed lower
    triangle, reduce-scattered and all-gathered by direct sends; all ranks then hold
    bit-identical H, so the (deterministic) inversion gives every rank the same perm and U;
  * the rows (output features) of each Linear are split over ranks, rows are independent in the
    column loop; the integer levels, scales and zeros are all-gathered afterwards.

The functions only move tensors; the arithmetic between them is the single-GPU code.
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.distributed as dist


# Tests set this to run the collectives even in a process group of one rank (the only way to put the
# RCCL code path on a machine with a single GPU).
FORCE_COLLECTIVES = False


def _single() -> bool:
    return world()[1] == 1 and not (FORCE_COLLECTIVES and dist.is_available() and dist.is_initialized())


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


def _slabs(c: int, n_slabs: int = 16) -> List[Tuple[int, int]]:
    """Row blocks [r0, r1) that cut the lower triangle of a C x C matrix into rectangles: block i is sent
    as rows r0..r1, columns 0..r1 (its part on and below the diagonal plus the small corner above it).
    16 blocks carry 17/32 of the matrix."""
    step = max(4, -(-c // n_slabs))
    return [(r0, min(c, r0 + step)) for r0 in range(0, c, step)]


def pack_lower(hessian: torch.Tensor, multiple: int = 1) -> torch.Tensor:
    """The lower triangle of a symmetric matrix as one flat buffer (row blocks of `_slabs`), zero-padded
    to a multiple of `multiple` elements."""
    c = hessian.shape[0]
    slabs = _slabs(c)
    n = sum((r1 - r0) * r1 for r0, r1 in slabs)
    flat = torch.zeros(-(-n // multiple) * multiple, dtype=hessian.dtype, device=hessian.device)
    off = 0
    for r0, r1 in slabs:
        k = (r1 - r0) * r1
        flat[off:off + k].view(r1 - r0, r1).copy_(hessian[r0:r1, :r1])
        off += k
    return flat


def unpack_lower(flat: torch.Tensor, hessian: torch.Tensor) -> None:
    """Inverse of pack_lower: writes the lower triangle back and mirrors it into the upper one, so the
    matrix comes out exactly symmetric."""
    off = 0
    for r0, r1 in _slabs(hessian.shape[0]):
        k = (r1 - r0) * r1
        blk = flat[off:off + k].view(r1 - r0, r1)
        hessian[r0:r1, :r1].copy_(blk)
        if r0 > 0:
            hessian[:r0, r0:r1].copy_(blk[:, :r0].t())
        off += k
    # the diagonal blocks were written whole (their corner above the diagonal comes from the sender's
    # symmetric matrix; sums of symmetric matrices in the same order are symmetric)


def allreduce_hessian(hessian: torch.Tensor, n_samples: int, algorithm: str = None) -> int:
    """Sum H over ranks in place and return the global sample count.

    H is exactly symmetric on every rank, so only its lower triangle travels (17/32 of the matrix).
    algorithm "direct" (default): every rank owns 1/N of the packed triangle; one all-to-all brings
    the N partial copies of that part to its owner (every pair of GPUs exchanges over its own xGMI
    link at the same time, where a ring is bound by one link), the owner adds them in rank order,
    and a second all-to-all hands the sums out -- a reduce-scatter and an all-gather made of direct
    sends.  "ring": the library's all_reduce on the packed triangle.  Either way every rank ends
    with the same bits."""
    rank, ws = world()
    if _single():
        return n_samples
    if algorithm is None:
        import os
        algorithm = os.environ.get("GQ_ALLREDUCE", "direct")
    on_device = dist.get_backend() == "nccl"
    flat = pack_lower(hessian, multiple=ws)
    work = flat if on_device or flat.device.type == "cpu" else flat.cpu()   # gloo: through host memory
    if algorithm == "ring":
        dist.all_reduce(work, op=dist.ReduceOp.SUM)
    else:
        piece = work.numel() // ws
        parts = torch.empty_like(work)
        dist.all_to_all_single(parts, work)                  # parts[r] = rank r's copy of my piece
        mine = parts.view(ws, piece)[0].clone()
        for r in range(1, ws):                               # fixed order: identical bits whoever owns the piece
            mine += parts.view(ws, piece)[r]
        dist.all_to_all_single(work, mine.repeat(ws))        # work[r] = the sum of piece r, from its owner
    if work is not flat:
        flat.copy_(work)
    unpack_lower(flat, hessian)
    n = torch.tensor([n_samples], dtype=torch.int64, device=hessian.device if on_device else "cpu")
    dist.all_reduce(n, op=dist.ReduceOp.SUM)
    return int(n.item())


def allgather_rows(local: Dict[str, torch.Tensor], n_rows: int, multiple: int = 64) -> Dict[str, torch.Tensor]:
    """`local[name]` holds this rank's row slice (dim 0) of a per-row result; returns the full
    tensors, rows in rank order, on every rank."""
    rank, ws = world()
    if _single():
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


def gather_quant_result(local, layer_fields: dict, n_rows: int, n_columns: int, dtype: torch.dtype, device,
                        multiple: int = 64) -> dict:
    """One Linear's ``{'quant_meta', 'metrics'}`` for all rows from every rank's row shard.
    ``local``: this rank's result of ``gptq_quant`` on its ``row_shard`` rows, or None if that shard is
    empty (a Linear with fewer 64-row units than ranks).  Per-row tensors are concatenated in rank
    order.  ``layer_fields``: the entries of ``quant_meta`` that do not depend on the rows
    (``perm_inv``, ``group_sizes``, ``group_bit_widths``, ...), which every rank can state from the
    Hessian it holds; used where ``local`` is None.  The four metrics are means over elements and
    are combined with the shards' row counts as weights."""
    rank, ws = world()
    b, e = row_shard(n_rows, rank, ws, multiple)
    n_groups = len(layer_fields["group_sizes"])
    if local is None:
        if e != b:
            raise ValueError("a rank with rows must pass its result")
        parts = {"qweight": torch.empty(0, n_columns, dtype=torch.uint8, device=device),
                 "scale": torch.empty(0, n_groups, dtype=dtype, device=device),
                 "qzero": torch.empty(0, n_groups, dtype=torch.uint8, device=device)}
        sums = torch.zeros(4, dtype=torch.float64)
        fields = layer_fields
    else:
        meta = local["quant_meta"]
        parts = {k: meta[k].to(device) for k in ("qweight", "scale", "qzero")}
        m = local["metrics"]
        sums = torch.tensor([m["gptq_error"], m["gptq_norm"], m["l2.0_error"], m["l2.0_norm"]], dtype=torch.float64) * (e - b)
        fields = {k: v for k, v in meta.items() if k not in parts}
    full = allgather_rows(parts, n_rows, multiple)
    if not _single():
        if dist.get_backend() == "nccl":
            sums = sums.to(device)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    sums = (sums.cpu() / n_rows).tolist()
    out_meta = dict(fields)
    out_meta.update(full)
    metrics = {"gptq_error": sums[0], "gptq_norm": sums[1], "l2.0_error": sums[2],
               "l2.0_norm": torch.tensor(sums[3], dtype=torch.float64).to(dtype).item()}
    return {"quant_meta": out_meta, "metrics": metrics}
