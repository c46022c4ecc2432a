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

from typing import Dict, List, Optional, Sequence, Tuple

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


def _tri(c: int) -> int:
    return c * (c + 1) // 2


def pack_lower(hessian: torch.Tensor, multiple: int = 1) -> torch.Tensor:
    """The lower triangle of a symmetric matrix as one flat buffer -- element (r, c <= r) at r (r + 1) / 2 + c --
    zero-padded to a multiple of `multiple` elements.  On the GPU one kernel (gq_pack_lower); CPU tensors (the
    gloo tests) go through an index."""
    c = hessian.shape[0]
    n = _tri(c)
    total = -(-n // multiple) * multiple
    flat = torch.empty(total, dtype=hessian.dtype, device=hessian.device)
    flat[n:].zero_()
    if hessian.is_cuda:
        from . import ops
        ops.pack_lower(hessian.contiguous(), flat)
    else:
        r, col = torch.tril_indices(c, c)
        flat[:n] = hessian[r, col]
    return flat


def unpack_lower(flat: torch.Tensor, hessian: torch.Tensor) -> None:
    """Inverse of pack_lower: writes the lower triangle back and mirrors it into the upper one, so the
    matrix comes out exactly symmetric."""
    c = hessian.shape[0]
    if hessian.is_cuda and hessian.is_contiguous():
        from . import ops
        ops.unpack_lower(flat, hessian)
        return
    r, col = torch.tril_indices(c, c)
    hessian[r, col] = flat[:_tri(c)]
    hessian[col, r] = flat[:_tri(c)]


def allreduce_hessian(hessian: torch.Tensor, n_samples: int, algorithm: str = None, n_total: int = None) -> int:
    """Sum H over ranks in place and return the global sample count.

    H is exactly symmetric on every rank, so only its lower triangle travels (half the matrix).
    algorithm "direct" (default): every rank owns 1/N of the packed triangle; one all-to-all brings
    the N partial copies of that part to its owner (every pair of GPUs exchanges over its own xGMI
    link at the same time, where a ring is bound by one link), the owner adds them in rank order,
    and a second all-to-all hands the sums out -- a reduce-scatter and an all-gather made of direct
    sends.  "ring": the library's all_reduce on the packed triangle.  Either way every rank ends
    with the same bits.  ``n_total``: the global sample count where the caller knows it (the driver
    does: it made the shards) -- the count is then not summed over ranks, which would cost a
    collective and a wait of the host for the device per set."""
    rank, ws = world()
    if _single():
        return n_samples if n_total is None else n_total
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
    if n_total is not None:
        return int(n_total)
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
        gathered = _all_gather_flat(mine.reshape(-1)).view((ws, most) + tuple(part.shape[1:]))
        out[name] = torch.cat([gathered[r, : e - b] for r, (b, e) in enumerate(bounds)], dim=0)
    return out


def _all_gather_flat(mine: torch.Tensor) -> torch.Tensor:
    """(ws, len(mine)): every rank's buffer of the same length"""
    ws = world()[1]
    gathered = torch.empty(ws * mine.numel(), dtype=mine.dtype, device=mine.device)
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(gathered, mine)
    else:
        # gloo (CPU tests, single-GPU rehearsals): the list form is the one it implements everywhere,
        # and device tensors go through host memory here (its own device path hung on 22 MB pieces)
        host = [torch.empty(mine.shape, dtype=mine.dtype) for _ in range(ws)]
        dist.all_gather(host, mine.cpu())
        gathered.copy_(torch.cat(host))
    return gathered.view(ws, mine.numel())


def pack_levels(qweight: torch.Tensor, bits: int) -> torch.Tensor:
    """(R, C) uint8 levels -> the bytes that travel: two levels per byte up to 4 bits (C is even), else as they are."""
    if bits > 4 or qweight.shape[1] % 2:
        return qweight.contiguous()
    pairs = qweight.reshape(qweight.shape[0], qweight.shape[1] // 2, 2)
    return (pairs[..., 0] | (pairs[..., 1] << 4)).contiguous()


def unpack_levels(packed: torch.Tensor, bits: int, n_columns: int) -> torch.Tensor:
    if bits > 4 or n_columns % 2:
        return packed
    return torch.stack([packed & 15, packed >> 4], dim=-1).reshape(packed.shape[0], n_columns)


def gather_quant_set(locals_: Sequence[Optional[dict]], local_sums: Optional[torch.Tensor], rows: Sequence[int],
                     n_columns: int, n_groups: int, dtype: torch.dtype, device, bits: int,
                     multiple: int = 64):
    """The Linears of one input-sharing set from every rank's row shards, in ONE all-gather and one all-reduce.

    ``locals_[i]``: this rank's ``quant_meta`` of Linear i on its ``row_shard(rows[i])`` rows (device tensors,
    valid in stream order; None where the shard is empty).  ``local_sums``: (len(rows), 4) float64 on ``device``,
    the four metric SUMS (not means) of this rank's rows, or None.  Returns ``(parts, sums)``: ``parts[i]`` =
    {"qweight", "scale", "qzero"} for all rows[i] rows, rows in rank order, and the sums over all ranks (still on
    the device: nothing here waits for the host).  What travels per row: the levels two to a byte (C / 2 bytes up
    to 4 bits), the scales in the weight's 16-bit type and the zero points, one byte each."""
    rank, ws = world()
    n = len(rows)
    if local_sums is None:
        local_sums = torch.zeros(n, 4, dtype=torch.float64, device=device)
    if _single():
        parts = [{k: m[k] for k in ("qweight", "scale", "qzero")} for m in locals_]
        return parts, local_sums
    lev_cols = n_columns // 2 if bits <= 4 and n_columns % 2 == 0 else n_columns
    per_row = lev_cols + 3 * n_groups                       # bytes of one row: levels, scales (2 bytes), zero points
    bounds = [[row_shard(r_, rk, ws, multiple) for rk in range(ws)] for r_ in rows]
    most = max(sum((e - b) for (b, e) in (bounds[i][rk] for i in range(n))) for rk in range(ws)) * per_row
    mine = torch.zeros(most, dtype=torch.uint8, device=device)
    off = 0
    for i, meta in enumerate(locals_):
        b, e = bounds[i][rank]
        if e == b:
            continue
        if meta is None:
            raise ValueError("a rank with rows must pass its result")
        k = e - b
        mine[off:off + k * lev_cols].view(k, lev_cols).copy_(pack_levels(meta["qweight"].to(device), bits))
        off += k * lev_cols
        mine[off:off + 2 * k * n_groups].copy_(meta["scale"].to(device).contiguous().view(torch.uint8).reshape(-1))
        off += 2 * k * n_groups
        mine[off:off + k * n_groups].copy_(meta["qzero"].to(device).reshape(-1))
        off += k * n_groups
    gathered = _all_gather_flat(mine)
    parts = []
    offs = [0] * ws
    for i in range(n):
        q, s, z = [], [], []
        for rk in range(ws):
            b, e = bounds[i][rk]
            k = e - b
            buf, o = gathered[rk], offs[rk]
            q.append(buf[o:o + k * lev_cols].view(k, lev_cols))
            o += k * lev_cols
            s.append(buf[o:o + 2 * k * n_groups].view(dtype).view(k, n_groups))
            o += 2 * k * n_groups
            z.append(buf[o:o + k * n_groups].view(k, n_groups))
            offs[rk] = o + k * n_groups
        parts.append({"qweight": unpack_levels(torch.cat(q, dim=0), bits, n_columns), "scale": torch.cat(s, dim=0),
                      "qzero": torch.cat(z, dim=0)})
    sums = local_sums if dist.get_backend() == "nccl" else local_sums.cpu()
    dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    return parts, sums.to(device)


def metrics_from_sums(sums_row, n_rows: int, n_columns: int, dtype: torch.dtype, with_hessian: bool = True) -> dict:
    """The four metrics of gptq_quant (gptq_py.py:195-238) from their sums over a whole layer."""
    n = float(n_rows * n_columns)
    gptq_error = sums_row[0] / n
    return {"gptq_error": gptq_error, "gptq_norm": sums_row[1] / n if with_hessian else gptq_error,
            "l2.0_error": sums_row[2] / n,
            # the reference takes this mean in the weight's 16-bit dtype (gptq_py.py:237)
            "l2.0_norm": torch.tensor(sums_row[3] / n, dtype=torch.float64).to(dtype).item()}


def gather_quant_result(local, layer_fields: dict, n_rows: int, n_columns: int, dtype: torch.dtype, device,
                        multiple: int = 64) -> dict:
    """One Linear's ``{'quant_meta', 'metrics'}`` for all rows from every rank's row shard (``gather_quant_set``
    for a set of one, with the metrics as numbers: this form waits for the device).
    ``local``: this rank's result of ``gptq_quant`` on its ``row_shard`` rows, or None if that shard is
    empty (a Linear with fewer 64-row units than ranks).  ``layer_fields``: the entries of ``quant_meta``
    that do not depend on the rows (``perm_inv``, ``group_sizes``, ``group_bit_widths``, ...), which every
    rank can state from the Hessian it holds; used where ``local`` is None.  The four metrics are means over
    elements and are combined with the shards' element counts as weights."""
    rank, ws = world()
    b, e = row_shard(n_rows, rank, ws, multiple)
    n_groups = len(layer_fields["group_sizes"])
    bits = int(max(layer_fields["group_bit_widths"].tolist()))
    if local is None:
        if e != b:
            raise ValueError("a rank with rows must pass its result")
        sums, fields, meta = None, layer_fields, None
    else:
        meta = local["quant_meta"]
        m = local["metrics"]
        sums = torch.tensor([[m["gptq_error"], m["gptq_norm"], m["l2.0_error"], m["l2.0_norm"]]], dtype=torch.float64,
                            device=device) * float((e - b) * n_columns)
        fields = {k: v for k, v in meta.items() if k not in ("qweight", "scale", "qzero")}
    parts, total = gather_quant_set([meta], sums, [n_rows], n_columns, n_groups, dtype, device, bits, multiple)
    out_meta = dict(fields)
    out_meta.update(parts[0])
    return {"quant_meta": out_meta, "metrics": metrics_from_sums(total[0].tolist(), n_rows, n_columns, dtype)}
