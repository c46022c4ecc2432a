"""Mirror of the reference's ``gptq_py.py`` (HessianHook, gptq_quant) on the HIP kernels.

Same call signatures, state fields, result dictionary and metrics as gptq_py.py:9-251.  Every
tensor stays on the GPU; the arithmetic is in libgptq_amd.so.  There is no CPU path: CPU tensors
are rejected.

``GPTQ`` at the bottom is a thin adapter with the classic ``add_batch`` / ``fasterquant`` names
over the same two objects.
"""
from __future__ import annotations

from typing import Optional

import os

import torch

from . import ops
from .quant import _group_list, with_host_list


class HessianHook:
    """Running Hessian H = sum X^T X of one Linear's inputs and its inverse factor
    (gptq_py.py:9-66)."""

    def __init__(self, overlap_batches: bool = False) -> None:
        """``overlap_batches``: consecutive ``add_batch`` calls alternate between two accumulators on two side
        streams, so that the first workgroups of one call fill the CUs that the last, partly empty round of
        the previous call leaves idle (its launch is 408 workgroups of one per CU at C = 4096 and 946 at
        C = 11008, on 256 CUs).  The caller's stream does not wait: an input must stay unmodified until
        ``hessian`` is read (freeing it is fine).  ``hessian`` is the sum of all accumulators in a fixed
        order, so it does not depend on timing.  Off by default: the reference's ``add_batch`` is ordered
        with the caller's stream.  The FIRST call fixes the layout (how many accumulators, how the tokens
        of a call are split) from its hidden size and token count; later calls of other lengths use it as it is."""
        self._hessian: Optional[torch.Tensor] = None
        # With few output tiles (C = 4096) the tokens of a batch are cut into runs that accumulate into
        # partial Hessians (ops.xtx_accumulate_split); `_hessian` is then parts[0] and reading `hessian`
        # first adds the other parts into it.
        self._parts: Optional[torch.Tensor] = None
        self._parts_dirty = False
        # With many tiles that do not fill whole rounds of one workgroup per CU (C = 11008: 3 rounds + 178 tiles) the
        # tiles of the last round are cut into token runs that accumulate into tile-sized buffers
        # (ops.xtx_accumulate_tail); reading `hessian` first folds them in.
        self._tail: Optional[torch.Tensor] = None
        self._tail_dirty = False
        self._overlap = bool(overlap_batches)
        self._lanes = 1          # accumulators that take turns (2 with overlap_batches where it pays)
        self._per_lane = 1       # partial Hessians of one accumulator (the token split)
        self._calls = 0
        self._lane_streams: list = []
        self.n_samples: int = 0
        self.hessian_inv: Optional[torch.Tensor] = None
        self.perm: Optional[torch.Tensor] = None
        self.perm_inv: Optional[torch.Tensor] = None
        self.tries: int = 0
        self._pending = None   # ops.PendingInversion of invert(wait=False) until confirm() has looked at it
        # counts rewrites of hessian_inv after it was handed out: work enqueued against an earlier generation is void
        # (every gptq_quant_set remembers the generation it read U at and compares when it settles)
        self._generation = 0

    @property
    def hessian(self) -> Optional[torch.Tensor]:
        if self._parts_dirty:
            main = torch.cuda.current_stream(self._parts.device)
            for st in self._lane_streams:
                main.wait_stream(st)
            ops.xtx_fold(self._parts)
            self._parts_dirty = False
        if self._tail_dirty:
            ops.xtx_fold_tail(self._hessian, self._tail)
            self._tail_dirty = False
        return self._hessian

    @hessian.setter
    def hessian(self, value: Optional[torch.Tensor]) -> None:
        self._hessian = value
        self._parts = None
        self._parts_dirty = False
        self._tail = None
        self._tail_dirty = False
        self._lanes = self._per_lane = 1

    @torch.no_grad()
    def add_batch(self, inp: torch.Tensor, use_kernel: bool = True) -> None:
        """inp (..., N, D).  ``n_samples`` counts sequences, a 2-D input is one sequence
        (gptq_py.py:25-27).  ``use_kernel=True``: fp16 / bf16 activations on the 16-bit matrix core
        (the reference's native and Triton ops take no other type, package/gptq/__init__.py:12-13).
        ``use_kernel=False``: the input is cast to fp32 first, whatever its type, and accumulated with
        fp32 products on the fp32 matrix core (the reference's torch.addmm branch, gptq_py.py:31-33);
        for 16-bit inputs both give the same H up to the order of the fp32 additions."""
        assert self.hessian_inv is None
        assert not hasattr(inp, "fake_mode")
        if inp.dim() <= 2:
            inp = inp[None]
        self.n_samples += len(inp)
        d = inp.size(-1)
        x = inp.flatten(end_dim=-2)
        if self._hessian is None:
            n_parts = ops.xtx_split_hint(d, x.shape[0]) if use_kernel and inp.is_cuda else 1
            lanes = 1
            if self._overlap and use_kernel and inp.is_cuda and d % 256 == 0 and x.shape[0] >= 8192:
                # worth a second accumulator when a call is at most two rounds of workgroups and the last one is at
                # most 90 % full (C = 4096: 1.6 rounds).  With more rounds the idle share is small and two calls
                # working on different tokens at once cost more in the XCDs' L2s than the tail gives back
                # (C = 11008, 3.7 rounds: 28.1 -> 30.9 ms per block with two accumulators).
                # (For those longer calls the other way to fill the last round was built and measured too: ONE accumulator,
                # every call as two launches on two streams -- workgroups 0..767 and 768..945 at C = 11008 -- so that the
                # full rounds of call k+1, ordered behind those of call k, may run beside the rest of call k, which
                # touches other tiles.  Bit-identical H, 27.3 -> 27.07 ms per block where filling the round would give
                # 25.3: the second queue's launch is not placed while the first still has workgroups waiting.)
                nt = d // 256
                rounds = nt * (nt + 1) // 2 * n_parts / 256.0
                if rounds <= 2.0 and -(-rounds // 1) - rounds >= 0.1:
                    lanes = 2
            self._lanes, self._per_lane = lanes, n_parts
            if lanes * n_parts > 1:
                self._parts = torch.zeros(lanes * n_parts, d, d, dtype=torch.float32, device=inp.device)
                self._hessian = self._parts[0]
            else:
                self._hessian = torch.zeros(d, d, dtype=torch.float32, device=inp.device)
                tail_bytes = ops.xtx_tail_workspace(d) if use_kernel and inp.is_cuda and x.shape[0] >= 8192 else 0
                if tail_bytes:
                    self._tail = torch.zeros(tail_bytes // 4, dtype=torch.float32, device=inp.device)
        if not use_kernel:
            ops.xtx_accumulate_f32(self.hessian, x.to(dtype=torch.float32).contiguous())
            return
        if not x.is_contiguous():
            x = x.contiguous()
        if self._lanes > 1:
            lane = self._calls % self._lanes
            self._calls += 1
            if not self._lane_streams:
                self._lane_streams = _side_streams(inp.device)[:self._lanes]
                for lane_stream in self._lane_streams:
                    # the accumulators were allocated on the caller's stream and are written on these: a hook that is
                    # dropped without `hessian` ever being read must not hand their memory back while a lane still adds
                    self._parts.record_stream(lane_stream)
            st = self._lane_streams[lane]
            st.wait_stream(torch.cuda.current_stream(inp.device))   # x was produced there
            x.record_stream(st)
            mine = self._parts[lane * self._per_lane:(lane + 1) * self._per_lane]
            with torch.cuda.stream(st):
                if self._per_lane > 1:
                    ops.xtx_accumulate_split(mine, x, save_lower_only=False, compute_lower_only=True)
                else:
                    ops.xtx_accumulate(mine[0], x, save_lower_only=False, compute_lower_only=True)
            self._parts_dirty = True
        elif self._parts is not None:
            ops.xtx_accumulate_split(self._parts, x, save_lower_only=False, compute_lower_only=True)
            self._parts_dirty = True
        elif self._tail is not None:
            ops.xtx_accumulate_tail(self._hessian, self._tail, x)
            self._tail_dirty = True
        else:
            ops.xtx_accumulate(self._hessian, x, save_lower_only=False, compute_lower_only=True)

    @torch.no_grad()
    def invert(self, damp_ratio: float = 1e-2, act_order: bool = True, wait: bool = True) -> torch.Tensor:
        """gptq_py.py:38-66.  Leaves ``hessian`` scaled, permuted and damped, sets ``perm`` /
        ``perm_inv`` (int64, as torch.argsort returns them) and returns ``hessian_inv``.

        The reference's retry loop looks at every try's result on the host (gptq_py.py:62).  With
        ``wait=False`` only the first try is enqueued and nothing is waited for: the host goes on
        enqueueing the work that reads ``hessian_inv`` while the device inverts, and ``confirm()`` -- which
        ``gptq_quant_set`` calls when it fetches its metrics -- reads the verdict; if the damped Hessian
        was not positive definite after all, it makes the remaining tries and the caller redoes what it
        had computed from ``hessian_inv``."""
        if wait:
            u, perm, perm_inv, tries = ops.hessian_invert(self.hessian, float(self.n_samples), damp_ratio, act_order,
                                                          max_try=100)
        else:
            self._pending = ops.hessian_invert_begin(self.hessian, float(self.n_samples), damp_ratio, act_order)
            u, perm, perm_inv, tries = self._pending.u, self._pending.perm, self._pending.perm_inv, 1
        self.hessian_inv = u
        self.tries = tries
        if act_order:
            self.perm = perm.to(torch.int64)
            self.perm_inv = perm_inv.to(torch.int64)
        return self.hessian_inv

    @torch.no_grad()
    def confirm(self) -> bool:
        """Settles ``invert(wait=False)``: waits for the device, and if the first try had failed makes the
        remaining ones (``hessian_inv`` is rewritten in place, ``tries`` updated).  Returns False in that case
        -- whatever was computed from ``hessian_inv`` since is void -- and True otherwise (also when nothing
        was pending)."""
        if self._pending is None:
            return True
        pending, self._pending = self._pending, None
        self.tries = ops.hessian_invert_finish(pending, max_try=100)
        if self.tries != 1:
            self._generation += 1   # U has been written again: whoever read the first try's U has to start over
        return self.tries == 1


# The Linears that share a Hessian are independent given U (and so are the rows of one Linear).
# The column loop is a chain of short kernels (one block of <= 128 columns, then its trailing
# update) that leaves most of the chip idle, so the Linears of a set run on separate HIP streams
# and fill each other's gaps (measured on the Llama-2-7B block: 32.6 -> 27.4 ms).  Slicing the
# rows of ONE Linear over streams does not pay: a slice's chain is barely shorter than the whole
# layer's (1.95 vs 2.56 ms for a quarter of a 4096 x 4096 layer), so ROW_SLICES stays 1.  A set of ONE Linear
# (o_proj, down_proj) has nobody to share the chip with; while the column loop was one launch per 128 columns,
# cutting its rows into three slices on the three side streams paid (one slice's update GEMM beside another's
# block kernel: 20.2 -> 19.4 ms per Llama-2-7B block).  With one launch per 512 columns the slices only get in
# each other's way: o_proj 1.43 / 1.60 / 1.75 ms and down_proj 6.16 / 6.02 / 6.14 ms for 1 / 2 / 3 slices, so
# SINGLE_SLICES is 1 (GQ_SINGLE_SLICES overrides it; the levels do not depend on it, rows are independent).
# The opposite -- the rows of the set's Linears stacked into ONE column loop (bit for bit the same levels;
# block launches of three times the workgroups, trailing updates of three times the tiles) -- was measured
# too and is slower: 21.2 against 19.9 ms per Llama-2-7B block.  A stacked gate/up launch is 344 workgroups
# of one per CU, two rounds, and one chain cannot run a block kernel beside another layer's update GEMM.
# HIP gives a process 4 hardware queues by default and streams beyond that share a queue, where a
# wait of one stream holds up the kernels of the other behind it: three side streams plus the
# caller's stream stay within the four (no set of a Llama block has more than three Linears).
# (Raising the limit instead -- GPU_MAX_HW_QUEUES=8 with four or six side streams and as many slices -- was
# measured: the column loops of a Llama-2-7B block 17.4 -> 22-23 ms, and the inversions, which run on the
# caller's stream alone, 19.9 -> 32 ms.  Dependent kernels of one stream start later when the process has
# more hardware queues to poll.)
# The graph is a plain fork / join -- every side stream waits for the caller's stream once, the
# caller's stream waits once for the levels of every job (an event behind its column loop; the metric
# sums behind that event are waited for by the host only), no side stream waits for another, no event is
# recorded twice, no stream has a priority -- so it has no cycle whatever the queue mapping is.
N_QUANT_STREAMS = 3
ROW_SLICES = 1
SINGLE_SLICES = int(os.environ.get("GQ_SINGLE_SLICES", "1"))
MIN_SLICE_ROWS = 256
_streams: dict = {}


def _side_streams(device: torch.device) -> list:
    key = (device.type, device.index if device.index is not None else torch.cuda.current_device())
    if key not in _streams:
        _streams[key] = [torch.cuda.Stream(device=device) for _ in range(N_QUANT_STREAMS)]
    return _streams[key]


def _row_slices(n_rows: int, want: int) -> list:
    want = max(1, min(want, n_rows // MIN_SLICE_ROWS))
    step = -(-n_rows // want)
    step = -(-step // 64) * 64   # the block kernel works on 64-row groups
    return [(b, min(n_rows, b + step)) for b in range(0, n_rows, step)]


_const_cache: dict = {}


def _const_tensor(values, dtype: torch.dtype, device) -> torch.Tensor:
    """A small constant tensor (group sizes, bit widths) on ``device``: uploaded once per distinct list and
    cloned on the device afterwards, instead of one host-to-device copy per Linear."""
    key = (tuple(values), dtype, str(device))
    t = _const_cache.get(key)
    if t is None:
        if len(_const_cache) > 256:
            _const_cache.clear()
        t = _const_cache[key] = torch.as_tensor(list(values), dtype=dtype, device=device)
    return with_host_list(t.clone(), values)


@torch.no_grad()
def gptq_quant_set(
        weights,
        hessian_hook: Optional[HessianHook],
        group_sizes: torch.Tensor,
        group_bit_widths: torch.Tensor,
        scale_bit_width: torch.Tensor = None,
        gptq_use_kernel: bool = True,
        gptq_block_sizes: torch.Tensor = None,
        quant_symmetric: bool = False,
        quant_mse: bool = True,
        quant_max_shrink: float = .8,
        quant_n_grid: int = 100,
        quant_norm: float = 2.4,
        quant_use_kernel: bool = False,
        save_device: torch.device = torch.device('cpu'),
        wait: bool = True,
) -> list:
    """``gptq_quant`` for every weight of ``weights`` against ONE Hessian -- the Linears of an
    input-sharing set, which the reference quantises one after the other (llama.py:150-170).
    Same arguments and, per weight, the same result dictionary as ``gptq_quant``; the work of all
    of them is enqueued on side streams before anything is waited for.

    ``wait=False`` returns as soon as everything is enqueued: a ``QuantSetResult`` whose ``quant_meta``
    tensors are valid in stream order and whose ``metrics`` are None until ``settle()`` has fetched them.
    The caller enqueues what consumes the levels (reconstruction, packing, the replay of the
    calibration batches) and settles afterwards, so the device never waits for the host."""
    if quant_mse and quant_use_kernel:
        raise NotImplementedError("quant_use_kernel=True calls exllamav2_ext, which the reference does not import "
                                  "either (quant.py:4)")
    gs = _group_list(group_sizes)
    gb = _group_list(group_bit_widths)
    scale_bits = 0 if scale_bit_width is None else int(scale_bit_width)
    blocks = None
    if hessian_hook is not None and not gptq_use_kernel:
        if gptq_block_sizes is None:
            raise ValueError("the PyTorch-loop semantics need gptq_block_sizes")
        blocks = _group_list(gptq_block_sizes)
        if blocks == gs:
            blocks = None
    weights = list(weights)
    if not weights:
        return []
    device = weights[0].device
    perm32 = perm_inv32 = None
    u = None
    if hessian_hook is not None:
        u = hessian_hook.hessian_inv
        if hessian_hook.perm is not None:
            perm32 = hessian_hook.perm.to(torch.int32).contiguous()
            if hessian_hook.perm_inv is not None:
                perm_inv32 = hessian_hook.perm_inv.to(torch.int32).contiguous()
        # gptq_py.py:95-96 zeroes columns whose Hessian diagonal is 0; after invert() no diagonal
        # entry is 0 any more (dead ones were set to 1, then damped), so nothing is zeroed there either.
    sem = ops.SEM_KERNEL if gptq_use_kernel else ops.SEM_PYTHON
    shrink = None
    if quant_mse:   # the candidates exactly as the reference builds them (quant.py:151), on the host
        shrink = (1.0 - torch.arange(0.0, quant_max_shrink, 1.0 / quant_n_grid)).to(device)

    main = torch.cuda.current_stream(device)
    side = _side_streams(device)
    per_weight = 1 if scale_bits > 0 else (SINGLE_SLICES if len(weights) == 1 else ROW_SLICES)   # the step of a group's scale grid is a maximum over all rows

    def generation():
        return hessian_hook._generation if hessian_hook is not None else 0

    def enqueue():
        for st in side:
            st.wait_stream(main)   # inputs were produced on the caller's stream; earlier results of ours were read there
        jobs = []   # (weight index, row range, result tensors, metric sums)
        k = 0
        for wi, weight in enumerate(weights):
            w = weight if weight.is_contiguous() else weight.contiguous()
            if w.shape[0] % 4:
                # the trailing-update GEMMs work on multiples of 4 rows: pad with zero rows (rows are
                # independent; a zero row quantises to zero error and is cut off again below)
                w = torch.cat([w, w.new_zeros(4 - w.shape[0] % 4, w.shape[1])], dim=0)
            for b, e in _row_slices(w.shape[0], per_weight):
                st = side[k % len(side)]
                with torch.cuda.stream(st):
                    res = ops.quantize_layer(w[b:e], perm32, u, gs, gb, sym=quant_symmetric, semantics=sem,
                                             want_quant=True, want_error=hessian_hook is not None, block_sizes=blocks,
                                             scale_bits=scale_bits, shrink=shrink, norm=quant_norm)
                    levels_ready = torch.cuda.Event()
                    levels_ready.record(st)
                    # The metric sums are nobody's input on the device: the caller's stream waits for the levels
                    # only, and the two passes over the layer that the sums take run on the side stream beside
                    # whatever the caller enqueues next (reconstruction, the replay).  They read the weight and U:
                    # both are kept alive below until the sums have been fetched, and must not be overwritten
                    # before (wait=True, or settle()).
                    sums = ops.quant_metrics(w[b:e], perm32, u, res["quant_t"], res["error_t"], perm_inv32)
                    sums_ready = torch.cuda.Event()
                    sums_ready.record(st)
                main.wait_event(levels_ready)   # (waiting for the sums here instead: 5.22 against 5.19 s for the 7B job)
                jobs.append((wi, res, sums, sums_ready, w, u))
                k += 1

        # the result dictionaries first -- device work only, enqueued behind the column loops while those
        # still run -- and then the one wait of the call
        perm_inv_set = hessian_hook.perm_inv.to(torch.int16) if perm32 is not None else None
        metas = []
        for wi, weight in enumerate(weights):
            mine = [j for j in jobs if j[0] == wi]
            n_rows, n_columns = weight.shape
            cat = (lambda name: mine[0][1][name][:n_rows]) if len(mine) == 1 else (lambda name: torch.cat([j[1][name] for j in mine], dim=0)[:n_rows])
            if perm32 is not None:
                perm_inv = perm_inv_set
            else:
                perm_inv = torch.arange(n_columns, dtype=torch.int16, device=device)
            qweight, scale, qzero = cat("qweight"), cat("scale"), cat("qzero")
            qscale = sscale = sbw = None
            out_gs, out_gb = list(gs), list(gb)
            if scale_bit_width is not None:
                # gptq_py.py:202-229: with scale quantisation the groups are stored widest first (bit width,
                # then size, descending; earlier groups first among equals) and the column order follows
                order = sorted(range(len(gs)), key=lambda g: (gb[g], gs[g], -g), reverse=True)
                sbw = torch.as_tensor(scale_bit_width).to(dtype=torch.uint8, device=save_device)
                if scale_bits > 0:
                    qscale, sscale = cat("qscale"), mine[0][1]["sscale"]
                if order != list(range(len(gs))):
                    starts = [0]
                    for v in gs:
                        starts.append(starts[-1] + v)
                    idx_g = torch.tensor(order, dtype=torch.int64, device=device)
                    idx_c = torch.cat([torch.arange(starts[g], starts[g + 1], device=device) for g in order])
                    qweight, scale, qzero = qweight[:, idx_c], scale[:, idx_g], qzero[:, idx_g]
                    if qscale is not None:
                        qscale, sscale = qscale[:, idx_g], sscale[idx_g]
                    perm = hessian_hook.perm if perm32 is not None else torch.arange(n_columns, device=device)
                    stored = perm[idx_c]   # stored column j holds original column stored[j]
                    perm_inv = torch.empty(n_columns, dtype=torch.int64, device=device)
                    perm_inv[stored] = torch.arange(n_columns, device=device)
                    perm_inv = perm_inv.to(torch.int16)
                    out_gs, out_gb = [gs[g] for g in order], [gb[g] for g in order]
            metas.append({
                "qweight": qweight.to(device=save_device),
                "scale": scale.to(device=save_device),
                "qzero": qzero.to(device=save_device),
                "qscale": None if qscale is None else qscale.to(device=save_device),
                "sscale": None if sscale is None else sscale.to(device=save_device),
                "perm_inv": perm_inv.to(device=save_device),
                "group_sizes": _const_tensor(out_gs, torch.int16, save_device),
                "group_bit_widths": _const_tensor(out_gb, torch.uint8, save_device),
                "scale_bit_width": sbw,
            })
        return jobs, metas

    def finish(jobs, metas, all_sums):
        out = []
        for wi, weight in enumerate(weights):
            n_rows, n_columns = weight.shape
            sums = [sum(all_sums[k][i] for k, j in enumerate(jobs) if j[0] == wi) for i in range(4)]
            n = float(n_rows * n_columns)
            gptq_error = sums[0] / n
            metrics = {
                "gptq_error": gptq_error,
                "gptq_norm": sums[1] / n if hessian_hook is not None else gptq_error,
                "l2.0_error": sums[2] / n,
                # the reference takes this mean in the weight's 16-bit dtype (gptq_py.py:237)
                "l2.0_norm": torch.tensor(sums[3] / n, dtype=torch.float64).to(weight.dtype).item(),
            }
            out.append({"quant_meta": metas[wi], "metrics": metrics, "_metric_sums": sums})
        return out

    def run_to_the_end():
        # one device-to-host round trip for the metric sums of the whole set; the same wait settles an
        # inversion that was enqueued without one (if it needed another damping, U has just been written
        # again and the set is quantised once more from it)
        while True:
            read_at = generation()
            jobs, metas = enqueue()
            all_sums = fetch_sums(jobs)
            if hessian_hook is not None:
                hessian_hook.confirm()
            if generation() == read_at:
                return finish(jobs, metas, all_sums)

    def fetch_sums(jobs):
        for j in jobs:
            j[3].synchronize()   # the side stream's metric kernels of this job
        return torch.stack([j[2] for j in jobs]).tolist()

    if wait:
        return run_to_the_end()
    read_at = generation()
    jobs, metas = enqueue()

    def settle_now():
        # The verdict of an inversion enqueued without a wait is read once, by whichever set settles first; every set
        # that read that first try's U -- this one or one enqueued beside it -- notices by the generation that U has
        # been written again since and quantises again from the final U.  (The retries write U on the caller's
        # stream while kernels of a set that has not settled yet may still read it: their results are void anyway.)
        all_sums = fetch_sums(jobs)
        if hessian_hook is not None:
            hessian_hook.confirm()
        if generation() == read_at:
            return finish(jobs, metas, all_sums), True
        return run_to_the_end(), False

    def device_sums():
        out = torch.zeros(len(weights), 4, dtype=torch.float64, device=device)
        here = torch.cuda.current_stream(device)
        for j in jobs:
            here.wait_event(j[3])
            out[j[0]] += j[2]
        return out

    return QuantSetResult([{"quant_meta": m, "metrics": None} for m in metas], settle_now, device_sums)


class QuantSetResult(list):
    """What ``gptq_quant_set(wait=False)`` returns: the per-weight result dictionaries, with ``metrics``
    still None and the tensors of ``quant_meta`` valid in stream order."""

    def __init__(self, items, settle_now, device_sums=None) -> None:
        super().__init__(items)
        self._settle_now = settle_now
        self._device_sums = device_sums

    def metric_sums(self) -> torch.Tensor:
        """(len(self), 4) float64 on the device: per weight the four sums behind the metrics (squared scaled errors,
        squared weights over diag(U), squared rounding errors, squared weights -- sums, not means), valid on the
        current stream without a wait of the host.  For callers that combine shards of a layer on the device
        (``dist.gather_quant_set``) before anything is settled."""
        if self._settle_now is not None:
            return self._device_sums()
        dev = self[0]["quant_meta"]["qweight"].device if len(self) else "cpu"
        return torch.tensor([r["_metric_sums"] for r in self], dtype=torch.float64, device=dev).reshape(len(self), 4)

    def settle(self) -> bool:
        """Waits for the device and fills in ``metrics``.  Returns False if the inversion behind the set had
        been enqueued without a wait (``HessianHook.invert(wait=False)``) and its first try turned out to have
        failed: the set has then been quantised again from the final U, every ``quant_meta`` here has been
        replaced, and whatever the caller derived from the old ones has to be derived again."""
        if self._settle_now is None:
            return True
        final, first_try_held = self._settle_now()
        self._settle_now = None
        for mine, res in zip(self, final):
            mine.update(res)
        return first_try_held


@torch.no_grad()
def gptq_quant(
        weight: torch.Tensor,
        hessian_hook: Optional[HessianHook],
        group_sizes: torch.Tensor,
        group_bit_widths: torch.Tensor,
        scale_bit_width: torch.Tensor = None,
        gptq_use_kernel: bool = True,
        gptq_block_sizes: torch.Tensor = None,
        quant_symmetric: bool = False,
        quant_mse: bool = True,
        quant_max_shrink: float = .8,
        quant_n_grid: int = 100,
        quant_norm: float = 2.4,
        quant_use_kernel: bool = False,
        save_device: torch.device = torch.device('cpu'),
) -> dict:
    """gptq_py.py:69-251 with all of its knobs (mixed groups, ``scale_bit_width``, ``quant_mse``,
    ``gptq_block_sizes``).  ``gptq_use_kernel=True`` selects the semantics of the reference's
    native loop (grid from the unmodified weights, fp16 dequantisation), ``False`` those of its
    pure-PyTorch loop (grid from the error-compensated weights, fp32 dequantisation);
    ``hessian_hook=None`` rounds to nearest.  Returns {'quant_meta': ..., 'metrics': ...}."""
    return gptq_quant_set([weight], hessian_hook, group_sizes, group_bit_widths, scale_bit_width, gptq_use_kernel,
                          gptq_block_sizes, quant_symmetric, quant_mse, quant_max_shrink, quant_n_grid, quant_norm,
                          quant_use_kernel, save_device)[0]


class GPTQ:
    """``add_batch`` / ``fasterquant`` names over HessianHook + gptq_quant for one Linear.

    ``blocksize`` is the lazy-batch block of the column loop.  With ``static_groups`` (the
    reference's native loop, gptq_py.py:125-157) every group's grid comes from the unmodified
    weights and each weight receives the same chain of updates however the columns are cut, so the
    result does not depend on it; without (the PyTorch loop, gptq_py.py:158-193) the grids are taken
    from the weights as they are at the start of the block a group begins in, and ``blocksize`` is
    passed on as ``gptq_block_sizes``."""

    def __init__(self, layer: torch.nn.Linear) -> None:
        self.layer = layer
        self.hook = HessianHook()

    def add_batch(self, inp: torch.Tensor, out: torch.Tensor = None) -> None:
        self.hook.add_batch(inp)

    def fasterquant(self, blocksize: int = 128, percdamp: float = .01, groupsize: int = 128, actorder: bool = True,
                    bits: int = 4, sym: bool = False, static_groups: bool = True,
                    save_device: torch.device = None) -> dict:
        w = self.layer.weight.data
        c = w.shape[1]
        gsz = c if groupsize in (-1, None) else groupsize
        if c % gsz:
            raise ValueError("in_features must be a multiple of the group size")
        if blocksize is None or blocksize <= 0 or blocksize % 4:
            raise ValueError("blocksize must be a positive multiple of 4")
        group_sizes = torch.full([c // gsz], gsz, dtype=torch.int32)
        group_bits = torch.full([c // gsz], bits, dtype=torch.int32)
        blocks = [blocksize] * (c // blocksize) + ([c % blocksize] if c % blocksize else [])
        self.hook.invert(damp_ratio=percdamp, act_order=actorder, wait=False)   # settled inside gptq_quant
        return gptq_quant(w, self.hook, group_sizes, group_bits, gptq_use_kernel=static_groups,
                          gptq_block_sizes=torch.tensor(blocks, dtype=torch.int32), quant_symmetric=sym, quant_mse=False,
                          save_device=save_device if save_device is not None else w.device)
