"""Block-wise GPTQ driver for Llama-family decoders, on the HIP hot path.

What the reference's ``llama.py`` does with its Catcher / RecorderWrapper machinery
(llama.py:20-204, model_utils.py:69-109): walk the decoder blocks in order; inside a block
quantise the Linears in the order their inputs become available --
``[q, k, v] -> [o] -> [gate, up] -> [down]`` -- one Hessian per input-sharing set, every set seeing
the outputs of the sets already quantised before it; then run the quantised block to produce the
next block's inputs.  The result dictionary has the reference's shape
``{'data': {name: quant_meta}, 'metrics': {name: metrics}}`` (llama.py:82-85, 167-173).

Differences in construction, not in result: inputs are captured with forward pre-hooks instead of
the reference's fake-tensor dependency discovery, the Linears of a set are quantised together
(``gptq_quant_set``), and everything stays on the GPU.  A block with the Llama layout is walked in
stages -- attention once, then ``x + o_proj(.)``, the second norm, ``act(gate) * up`` and the
final residual from cached activations -- so every sub-module runs once per block; any other
block is re-run up to the hooked Linear once per set.
"""
from __future__ import annotations

import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
from torch import nn

from . import dist as gd
from . import ops
from .gptq_py import HessianHook, gptq_quant_set
from .quant import HipLinear, QuantLinear, reconstruct_weight

# Linears of one Llama decoder block in dependency order; each inner list shares its input.
LLAMA_SETS: List[List[str]] = [
    ["self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"],
    ["self_attn.o_proj"],
    ["mlp.gate_proj", "mlp.up_proj"],
    ["mlp.down_proj"],
]


class _Stop(Exception):
    """Raised from a pre-hook once the wanted input has been seen: the rest of the forward is not needed."""

    def __init__(self, args=(), kwargs=None):
        super().__init__("stop")
        self.call_args, self.call_kwargs = args, kwargs or {}


def _to_device(obj, device):
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, dict):
        return {k: _to_device(v, device) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_device(v, device) for v in obj)
    return obj


class _HostCopier:
    """Copies finished results to the CPU on a thread of its own.  A copy into pageable memory blocks the calling
    thread for its whole length (about 4 GB/s: 48 ms for the 207 MB of a Llama-2-7B block, a sixth of the block's
    time with the GPU idle behind it); here the enqueueing thread goes on and the copy runs beside the next set's
    work.  ``collect`` waits for everything and returns it; ``close`` ends the thread (also after an exception).

    The copies are ordered on a stream of their own, the process's fifth: the caller's stream and the three side
    streams of the column loops already fill its four hardware queues, so this one shares a queue with one of
    them, and the kernels of that stream can be held up behind a copy's barrier packet (no deadlock: every wait
    names work enqueued earlier).  Measured against the alternative of ordering the copies on the last side
    stream itself (GQ_COPIER_STREAM=side; random-init blocks of Llama-2-7B size, 128 x 2048 tokens): 0.212 s per
    block with the stream of its own, 0.226 s on the side stream, where the third Linear of the next q/k/v set
    queues behind the copy of down_proj's results."""

    def __init__(self, device: torch.device) -> None:
        from concurrent.futures import ThreadPoolExecutor
        import os
        from .gptq_py import _side_streams
        self._pool = ThreadPoolExecutor(max_workers=1)
        self._stream = None
        if device.type == "cuda":
            shared = os.environ.get("GQ_COPIER_STREAM") == "side"
            self._stream = _side_streams(device)[-1] if shared else torch.cuda.Stream(device)
        self._futures: Dict[str, object] = {}
        self._chunks = None   # two pinned staging chunks, made by the copying thread on first use

    def submit(self, key: str, obj) -> None:
        if self._stream is None:
            self._futures[key] = _to_device(obj, torch.device("cpu"))
            return
        ready = torch.cuda.Event()
        ready.record()   # on the caller's stream: everything enqueued so far produces `obj`
        self._futures[key] = self._pool.submit(self._run, obj, ready)

    # Device -> host through two pinned chunks: a copy into pageable memory goes through the runtime's own staging at
    # about 4 GB/s and occupies the GPU with a copy kernel per piece (traced: 38 ms of copy kernels per Llama-2-7B block,
    # beside the kernels of the next set); into pinned memory it is one DMA at the link's rate.  The finished chunk is
    # copied on to the pageable result by this thread while the next one is in flight.
    _CHUNK = 32 << 20

    def _pinned(self):
        if self._chunks is None:
            try:
                self._chunks = [torch.empty(self._CHUNK, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
            except RuntimeError:
                self._chunks = []   # no pinned memory to be had: plain copies
        return self._chunks

    def _tensor_to_host(self, t: torch.Tensor) -> torch.Tensor:
        n = t.numel() * t.element_size()
        if t.device.type != "cuda" or n < (1 << 20) or not self._pinned():
            return t.to("cpu")
        src = t.contiguous().view(-1).view(torch.uint8)
        out = torch.empty(t.shape, dtype=t.dtype)
        dst = out.view(-1).view(torch.uint8)
        pieces = [(o, min(n, o + self._CHUNK)) for o in range(0, n, self._CHUNK)]
        done = [torch.cuda.Event(), torch.cuda.Event()]
        for k in range(len(pieces) + 1):
            if k < len(pieces):
                b, e = pieces[k]
                # (chunk k % 2 last held piece k - 2, which turn k - 1 drained below)
                self._chunks[k % 2][:e - b].copy_(src[b:e], non_blocking=True)
                done[k % 2].record()
            if k >= 1:
                pb, pe = pieces[k - 1]
                done[(k - 1) % 2].synchronize()
                dst[pb:pe].copy_(self._chunks[(k - 1) % 2][:pe - pb])
        return out

    def _to_host(self, obj):
        if isinstance(obj, torch.Tensor):
            return self._tensor_to_host(obj)
        if isinstance(obj, dict):
            return {k: self._to_host(v) for k, v in obj.items()}
        if isinstance(obj, (list, tuple)):
            return type(obj)(self._to_host(v) for v in obj)
        return obj

    def _run(self, obj, ready):
        with torch.cuda.stream(self._stream):
            self._stream.wait_event(ready)
            return self._to_host(obj)   # synchronous: `obj` stays referenced until it is done

    def collect(self) -> Dict[str, dict]:
        return {key: (fut.result() if hasattr(fut, "result") else fut) for key, fut in self._futures.items()}

    def close(self) -> None:
        self._pool.shutdown(wait=True)
        self._futures = {}


def _first(out):
    """Decoder blocks return a tensor or a tuple whose first entry is the hidden state."""
    return out[0] if isinstance(out, (tuple, list)) else out


def _submodule(block: nn.Module, dotted: str) -> nn.Module:
    mod = block
    for part in dotted.split("."):
        mod = getattr(mod, part)
    return mod


def _is_llama_layout(block: nn.Module, sets) -> bool:
    """The decoder layout the staged walk assumes (LlamaDecoderLayer): x + attn(norm1(x)), then
    h + down(act(gate(norm2(h))) * up(norm2(h))), Linears without bias, default sets."""
    if [list(s_) for s_ in sets] != LLAMA_SETS:
        return False
    try:
        mods = [_submodule(block, nm) for names in LLAMA_SETS for nm in names]
        ok = (callable(block.mlp.act_fn) and isinstance(block.post_attention_layernorm, nn.Module)
              and isinstance(block.input_layernorm, nn.Module))
    except AttributeError:
        return False
    return ok and type(block).__name__ == "LlamaDecoderLayer" and all(getattr(m, "bias", None) is None for m in mods)


_HALF = (torch.float16, torch.bfloat16)


def _rms_norm(mod: nn.Module, x: torch.Tensor, fused: bool) -> torch.Tensor:
    """``mod(x)``; for the Llama RMS norm on 16-bit device activations in one pass (ops.rmsnorm keeps the module's
    roundings: seven framework operators and 4.6 GB of traffic per 268 MB batch otherwise)."""
    w = getattr(mod, "weight", None)
    if (fused and type(mod).__name__ == "LlamaRMSNorm" and hasattr(mod, "variance_epsilon") and isinstance(w, torch.Tensor)
            and x.is_cuda and x.dtype in _HALF and w.dtype == x.dtype and w.device == x.device and x.is_contiguous()
            and x.shape[-1] % 8 == 0 and w.shape == (x.shape[-1],)):
        return ops.rmsnorm(x, w.data.contiguous(), float(mod.variance_epsilon))
    return mod(x)


def _act_mul(act_fn, g: torch.Tensor, u: torch.Tensor, fused: bool) -> torch.Tensor:
    """``act_fn(g) * u``; for SiLU on 16-bit device activations in one pass, into g's memory."""
    if (fused and type(act_fn).__name__ in ("SiLU", "SiLUActivation") and g.is_cuda and g.dtype in _HALF and u.dtype == g.dtype
            and g.shape == u.shape and g.is_contiguous() and u.is_contiguous() and g.numel() % 8 == 0):
        return ops.silu_mul(g, u, out=g)
    return act_fn(g) * u


def _fast_attention_ok(attn: nn.Module, kwargs: dict, x: torch.Tensor) -> bool:
    """Whether ``_llama_attention`` computes what ``attn`` would: the plain Llama attention (causal, no mask tensor, no
    cache, SDPA), dense q/k/v of this library without bias, 16-bit activations on the GPU."""
    try:
        cfg = attn.config
        pe = kwargs.get("position_embeddings")
        mods = [attn.q_proj, attn.k_proj, attn.v_proj]
        return (type(attn).__name__ == "LlamaAttention" and getattr(cfg, "_attn_implementation", None) == "sdpa"
                and kwargs.get("attention_mask") is None and kwargs.get("past_key_values") is None
                and isinstance(pe, (tuple, list)) and len(pe) == 2 and all(isinstance(t_, torch.Tensor) and t_.dim() == 3 for t_ in pe)
                and pe[0].dtype == x.dtype and pe[0].shape[-1] == attn.head_dim and pe[0].shape[0] in (1, x.shape[0])
                and bool(getattr(attn, "is_causal", False)) and not attn.training and getattr(attn, "sliding_window", None) is None
                and x.is_cuda and x.dtype in _HALF and x.dim() == 3 and attn.head_dim % 16 == 0
                and all(type(m) is HipLinear and m.bias is None and m.weight.device == x.device and m.weight.dtype == x.dtype for m in mods)
                and attn.q_proj.out_features == cfg.num_attention_heads * attn.head_dim
                and attn.k_proj.out_features == cfg.num_key_value_heads * attn.head_dim
                and attn.v_proj.out_features == attn.k_proj.out_features)
    except AttributeError:
        return False


def _llama_attention(attn: nn.Module, merged: nn.Module, x: torch.Tensor, kwargs: dict) -> torch.Tensor:
    """The input of o_proj for hidden states x (batch, seq, hidden), as LlamaAttention.forward computes it, in four
    launches: q/k/v as one product with the concatenated weight (``merged``), the rotary embedding in place on its q
    and k columns (ops.rope_inplace, the operators' roundings), the framework's fused attention on strided views of that
    buffer, one copy of its output into (batch, seq, heads * head_dim).  The module's own path runs ten element-wise
    operators over q and k for the rotary embedding and three products."""
    b, s, _ = x.shape
    d = attn.head_dim
    hq, hkv = attn.config.num_attention_heads, attn.config.num_key_value_heads
    cos, sin = kwargs["position_embeddings"]
    qkv = merged(x)
    ops.rope_inplace(qkv, cos.contiguous(), sin.contiguous(), hq + hkv, d)
    q = qkv[..., :hq * d].view(b, s, hq, d).transpose(1, 2)
    k = qkv[..., hq * d:(hq + hkv) * d].view(b, s, hkv, d).transpose(1, 2)
    v = qkv[..., (hq + hkv) * d:].view(b, s, hkv, d).transpose(1, 2)
    extra = {"enable_gqa": True} if hkv != hq else {}
    out = torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=None, dropout_p=0.0, scale=attn.scaling,
                                                           is_causal=s > 1, **extra)
    return out.transpose(1, 2).reshape(b, s, hq * d).contiguous()


class _SliceOfMerged(nn.Module):
    """Stands in for one of several Linears that read the same input: the first of them to be called runs ONE product
    with the concatenated weight, each returns its columns of it (a view)."""

    def __init__(self, shared: dict, lo: int, hi: int) -> None:
        super().__init__()
        self._shared, self._lo, self._hi = shared, lo, hi

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        sh = self._shared
        # the input itself is kept (not its address: a freed tensor's memory may be handed to the next one) together
        # with its version counter, so a product is reused only for the very tensor it was made from, unmodified
        if sh.get("x") is not x or sh.get("version") != x._version:
            sh["x"], sh["version"], sh["y"] = x, x._version, sh["linear"](x)
        return sh["y"][..., self._lo:self._hi]


class _merged_linears:
    """``with _merged_linears(parent, names):`` -- while inside, the named dense Linears of ``parent`` (same input, no
    bias, this library's HipLinear) are replaced by slices of one product with their concatenated weight: one launch of
    12288 rows instead of three of 4096 for q/k/v (1352 against 1152 TFLOP/s at 32768 tokens).  Every output element is
    the same k-ascending sum either way, so the values are bit for bit those of the separate Linears.  Anything else
    (packed layers, a bias, another module type) is left as it is."""

    def __init__(self, parent: nn.Module, names: Sequence[str]) -> None:
        self._parent, self._names, self._saved = parent, list(names), None

    def __enter__(self):
        mods = [getattr(self._parent, n, None) for n in self._names]
        if not all(type(m) is HipLinear and m.bias is None and m.weight.is_cuda for m in mods):
            return self
        if len({(m.in_features, m.weight.dtype, m.weight.device) for m in mods}) != 1:
            return self
        shared = {"linear": HipLinear.from_weight(torch.cat([m.weight.data for m in mods], dim=0))}
        self._saved = mods
        lo = 0
        for n, m in zip(self._names, mods):
            setattr(self._parent, n, _SliceOfMerged(shared, lo, lo + m.out_features))
            lo += m.out_features
        self._shared = shared
        return self

    def __exit__(self, *exc):
        if self._saved is not None:
            for n, m in zip(self._names, self._saved):
                setattr(self._parent, n, m)
            self._shared.clear()
            self._saved = None
        return False


@torch.no_grad()
def get_initial_inputs(model: nn.Module, encodings: torch.Tensor, device: torch.device,
                       batch_size: int = 1) -> Tuple[torch.Tensor, dict]:
    """Inputs of the first decoder block for every calibration sequence, and the keyword
    arguments the model passes to its blocks (llama.py:20-61).  encodings (B, N) int64."""
    blocks = model.model.layers
    use_cache, model.config.use_cache = model.config.use_cache, False
    model.model.embed_tokens.to(device)
    if getattr(model.model, "rotary_emb", None) is not None:
        model.model.rotary_emb.to(device)

    def catch(_mod, args, kwargs):
        raise _Stop(args, kwargs)

    handle = blocks[0].register_forward_pre_hook(catch, with_kwargs=True)
    inps, inp_kwargs = [], {}
    try:
        for bi in range(0, len(encodings), batch_size):
            try:
                model(encodings[bi:bi + batch_size].to(device))
            except _Stop as stop:
                hidden = stop.call_args[0] if stop.call_args else stop.call_kwargs["hidden_states"]
                inps.append(hidden)
                inp_kwargs = {k: v for k, v in stop.call_kwargs.items() if k != "hidden_states"}
    finally:
        handle.remove()
        model.config.use_cache = use_cache
    return torch.cat(inps, dim=0), inp_kwargs


@torch.no_grad()
def quantize_llama(
        model: nn.Module,
        encodings: torch.Tensor,
        device: torch.device,
        batch_size: int = 1,
        bits: int = 4,
        group_size: int = 128,
        act_order: bool = True,
        damp_ratio: float = 1e-2,
        pack: bool = False,
        sets: Sequence[Sequence[str]] = LLAMA_SETS,
        on_set: Optional[Callable] = None,
        symmetric: bool = False,
        quant_mse: bool = False,
        scale_bit_width: Optional[int] = None,
        staged: Optional[bool] = None,
        offload: bool = False,
        fused_elementwise: bool = True,
) -> Dict[str, Dict[str, dict]]:
    """Quantise every decoder block of ``model`` in place (llama.py:64-204).

    Under an initialised ``torch.distributed`` process group (one process per GPU) the work is
    shared: every rank walks the blocks with ITS shard of the calibration sequences
    (``dist.sample_shard``), the Hessian of each input-sharing set is summed over ranks, every rank
    inverts it (bit-identical inputs, deterministic kernels: identical ``perm`` and ``U``), the rows
    of each Linear are split over ranks for the column loop and the integer levels, scales and zero
    points are gathered, so every rank ends with the same quantised block and the same results.
    The reference is single-GPU (parse_args.py:53-55).

    After the call each Linear holds the dense reconstructed weight (what the reference leaves in
    the model), or is replaced by a packed ``QuantLinear`` when ``pack`` is set.  ``on_set(block
    index, names, hessian_hook, weights)`` is called after each set's Hessian is inverted and
    before its weights change (for inspection and tests).  ``symmetric``, ``quant_mse`` and
    ``scale_bit_width`` are the grid options the reference hard-codes to off (llama.py:146-163).
    ``staged``: walk Llama-layout blocks in stages from cached activations (None: when the block
    has that layout; False: always re-run the block up to each hooked Linear; True is only valid
    with the default ``sets``).  ``offload``: move each finished block back to the CPU, as the
    reference does (llama.py:194-196); without it the whole quantised model stays on ``device``
    (about 2 bytes per weight: 13 GB for Llama-2-7B, 138 GB for 70B, inside one MI355X's 288 GB).
    ``fused_elementwise``: the staged walk runs the block's two RMS norms and ``act(gate) * up`` as one-pass
    kernels with the modules' own roundings (only the order in which a row's squares are summed differs, so an
    activation can differ from the module's in its last bit); False runs the modules themselves."""
    if staged and [list(s_) for s_ in sets] != LLAMA_SETS:
        raise ValueError("staged=True walks the Llama block layout and needs the default `sets`")
    rank, world = gd.world()
    if world > 1:
        if scale_bit_width is not None and int(scale_bit_width) > 0:
            raise NotImplementedError("EXL2 scale quantisation takes a maximum over all rows of a group: not offered "
                                      "together with row sharding over ranks")
        sb, se = gd.sample_shard(len(encodings), rank, world)
        if se == sb:
            raise ValueError(f"{len(encodings)} calibration sequences cannot be split over {world} ranks")
        n_global = len(encodings)
        encodings = encodings[sb:se]
    inps, inp_kwargs = get_initial_inputs(model, encodings, device, batch_size)
    inp_kwargs = _to_device(inp_kwargs, device)
    use_cache, model.config.use_cache = model.config.use_cache, False
    dtype = inps.dtype
    results: Dict[str, Dict[str, dict]] = {"data": {}, "metrics": {}, "seconds_per_block": []}

    sbw = None if scale_bit_width is None else torch.tensor(int(scale_bit_width))
    to_host = _HostCopier(torch.device(device))
    n = len(inps)
    batches = [(bi, min(n, bi + batch_size)) for bi in range(0, n, batch_size)]

    def quantise_set(gi, block, names, hook):
        """invert + quantise + put the results into the block; the Hessian is complete"""
        mods = [_submodule(block, nm) for nm in names]
        if world > 1:
            # the global count is known here (this function made the shards): no collective and no wait for it
            hook.n_samples = gd.allreduce_hessian(hook.hessian, hook.n_samples, n_total=n_global)
        # no host round trip between inversion and column loops (gptq_quant_set settles it), unless a
        # callback wants to look at the finished hook first
        hook.invert(damp_ratio=damp_ratio, act_order=act_order, wait=on_set is not None)
        weights = [m.weight.data for m in mods]
        c = weights[0].shape[1]
        if c % group_size:
            raise ValueError(f"in_features {c} is not a multiple of the group size {group_size}")
        gs = torch.full([c // group_size], group_size, dtype=torch.int32)
        gb = torch.full([c // group_size], bits, dtype=torch.int32)
        if on_set is not None:
            on_set(gi, list(names), hook, weights)
        # the column loops are enqueued and nothing is waited for: what consumes the levels on the device
        # (gather over ranks, reconstruction or packing) is enqueued behind them, then the set is settled
        # (metrics, verdict of the inversion); only a failed first try makes the loop run twice
        if world > 1:
            shards = [gd.row_shard(w.shape[0], rank, world) for w in weights]
            mine = [(i, w[b:e]) for i, (w, (b, e)) in enumerate(zip(weights, shards)) if e > b]   # a short Linear may leave a rank without rows
        else:
            mine = list(enumerate(weights))
        part = gptq_quant_set([w for _, w in mine], hook, gs, gb, scale_bit_width=sbw, gptq_use_kernel=True,
                              gptq_block_sizes=gs, quant_symmetric=symmetric, quant_mse=quant_mse,
                              save_device=device, wait=False)
        sums = None
        while True:
            if world > 1:
                # one all-gather of the set's levels (two to a byte), scales and zero points and one all-reduce of
                # its metric sums, all on the device: like the single-GPU flow, nothing waits for the host here
                c_ = weights[0].shape[1]
                fields = {"qscale": None, "sscale": None, "scale_bit_width": None if sbw is None else sbw.to(torch.uint8).to(device),
                          "perm_inv": (hook.perm_inv.to(torch.int16) if hook.perm_inv is not None
                                       else torch.arange(c_, dtype=torch.int16, device=device)),
                          "group_sizes": gs.to(dtype=torch.int16, device=device),
                          "group_bit_widths": gb.to(dtype=torch.uint8, device=device)}
                local_meta = [None] * len(weights)
                local_sums = torch.zeros(len(weights), 4, dtype=torch.float64, device=device)
                mine_sums = part.metric_sums() if mine else None
                for k, ((i, _), res) in enumerate(zip(mine, part)):
                    local_meta[i] = res["quant_meta"]
                    local_sums[i] = mine_sums[k]
                parts, sums = gd.gather_quant_set(local_meta, local_sums, [w.shape[0] for w in weights], c_, len(gs),
                                                  weights[0].dtype, device, bits)
                outs = [{"quant_meta": dict(fields, **p_), "metrics": None} for p_ in parts]
            else:
                outs = list(part)
            layers = []
            for mod, res in zip(mods, outs):
                meta = res["quant_meta"]
                if pack:
                    layers.append(QuantLinear.from_quant_meta(meta, device, validate=False, has_perm=bool(act_order)))
                elif mod.bias is None:
                    # the dense reconstructed weight the reference leaves in the model, in an nn.Linear whose
                    # forward (the replay of the calibration batches) runs this library's dense MFMA kernel
                    layers.append(HipLinear.from_weight(reconstruct_weight(meta, dtype, device)))
                else:
                    layers.append(reconstruct_weight(meta, dtype, device))
            if part.settle():
                break
        if world > 1:
            rows_of = sums.tolist()   # the host has just waited for this set in settle(): the sums are there
            for res, w, row in zip(outs, weights, rows_of):
                res["metrics"] = gd.metrics_from_sums(row, w.shape[0], w.shape[1], w.dtype)
        for name, mod, res, layer in zip(names, mods, outs, layers):
            canonical = f"model.layers.{gi}.{name}"
            results["metrics"][canonical] = res["metrics"]
            to_host.submit(canonical, res["quant_meta"])
            parent = _submodule(block, name.rsplit(".", 1)[0]) if "." in name else block
            if isinstance(layer, torch.Tensor):
                mod.weight.data = layer
                layer = mod
            setattr(parent, name.rsplit(".", 1)[-1], layer)

    def run_until(block, name, hook, keep=None, entry=None):
        """forward of every batch up to the input of sub-module `name`: into the Hessian, and appended to the list
        `keep` if given (the tensor the module was about to receive, not a copy: nothing writes it afterwards).
        `entry` = (sub-module, its inputs, one tensor per batch): start there instead of at the block's first layer."""

        def feed(_mod, args):
            x = args[0]
            hook.add_batch(x)
            if keep is not None:
                keep.append(x)
            raise _Stop()

        handle = _submodule(block, name).register_forward_pre_hook(feed)
        try:
            for i in range(len(batches)):
                try:
                    if entry is None:
                        block(xs[i], **inp_kwargs)
                    else:
                        entry[0](hidden_states=entry[1][i], **inp_kwargs)
                except _Stop:
                    pass
        finally:
            handle.remove()

    # Activations are kept as one tensor per batch, each the very tensor its producer returned: a preallocated
    # (sequences, tokens, features) buffer filled slice by slice costs a device copy of every activation at every stage
    # (traced on Llama-2-7B dimensions: 47 ms of 250 ms per block in copy kernels).
    xs = [inps[b:e] for b, e in batches]
    del inps
    try:
        for gi, block in enumerate(model.model.layers):
            t0 = time.time()
            block.to(device)
            nxt = [None] * len(batches)
            if _is_llama_layout(block, sets) if staged is None else staged:
                # stage 1: q, k, v see the first norm's output.  (The hooks alternate between two accumulators where a
                # call is a short launch; nothing below writes an input again before its set is quantised.)
                hook = HessianHook(overlap_batches=True)
                normed = []
                if fused_elementwise:
                    for i in range(len(batches)):
                        normed.append(_rms_norm(block.input_layernorm, xs[i], True))
                        hook.add_batch(normed[i])
                else:
                    run_until(block, sets[0][0], hook, keep=normed)
                quantise_set(gi, block, sets[0], hook)
                # stage 2: attention once, on the first norm's output as stage 1 saw it; its output is o_proj's input
                hook = HessianHook(overlap_batches=True)
                attn = []
                if fused_elementwise and _fast_attention_ok(block.self_attn, inp_kwargs, normed[0]):
                    merged = HipLinear.from_weight(torch.cat([_submodule(block, nm).weight.data for nm in sets[0]], dim=0))
                    for i in range(len(batches)):
                        attn.append(_llama_attention(block.self_attn, merged, normed[i], inp_kwargs))
                        normed[i] = None
                        hook.add_batch(attn[i])
                    del merged
                else:
                    with _merged_linears(block.self_attn, [nm.rsplit(".", 1)[-1] for nm in sets[0]] if fused_elementwise else []):
                        run_until(block, sets[1][0], hook, keep=attn, entry=(block.self_attn, normed))
                del normed
                quantise_set(gi, block, sets[1], hook)
                # stage 3: h = x + o_proj(attn), z = norm2(h) feeds gate and up
                hook = HessianHook(overlap_batches=True)
                o_proj = _submodule(block, sets[1][0])
                h, z = [], []
                for i in range(len(batches)):
                    h.append(xs[i] + o_proj(attn[i]))
                    attn[i] = xs[i] = None
                    z.append(_rms_norm(block.post_attention_layernorm, h[i], fused_elementwise))
                    hook.add_batch(z[i])
                del attn
                quantise_set(gi, block, sets[2], hook)
                # stage 4: a = act(gate(z)) * up(z) feeds down
                hook = HessianHook(overlap_batches=True)
                gate, up = _submodule(block, sets[2][0]), _submodule(block, sets[2][1])
                act = []
                for i in range(len(batches)):
                    act.append(_act_mul(block.mlp.act_fn, gate(z[i]), up(z[i]), fused_elementwise))
                    z[i] = None
                    hook.add_batch(act[i])
                del z
                quantise_set(gi, block, sets[3], hook)
                down = _submodule(block, sets[3][0])
                for i in range(len(batches)):
                    nxt[i] = h[i] + down(act[i])
                    h[i] = act[i] = None
                del h, act, hook
            else:
                for names in sets:
                    hook = HessianHook(overlap_batches=True)
                    run_until(block, names[0], hook)
                    quantise_set(gi, block, names, hook)
                    del hook
                for i in range(len(batches)):
                    nxt[i] = _first(block(xs[i], **inp_kwargs))
                    xs[i] = None
            xs = nxt
            if offload:
                block.to("cpu")
            results["seconds_per_block"].append(time.time() - t0)

        results["data"] = to_host.collect()
    finally:
        # also when a block raises: the copier's thread, and the device tensors its queued copies hold, must go
        to_host.close()
        model.config.use_cache = use_cache
    return results


@torch.no_grad()
def evaluate_llama(model: nn.Module, encodings: torch.Tensor, device: torch.device, batch_size: int = 1,
                   loss_in_fp32: bool = False) -> torch.Tensor:
    """Perplexity over ``encodings`` (B, N), block by block as the reference evaluates
    (llama.py:207-247): mean over sequences of the per-batch token cross-entropy, exponentiated.
    The cross-entropy is taken on the logits in the model's own dtype, as the reference does
    (llama.py:241); ``loss_in_fp32`` casts them to fp32 first."""
    inps, inp_kwargs = get_initial_inputs(model, encodings, device, batch_size)
    inp_kwargs = _to_device(inp_kwargs, device)
    use_cache, model.config.use_cache = model.config.use_cache, False
    xs = [inps[bi:bi + batch_size] for bi in range(0, len(inps), batch_size)]   # one tensor per batch, never copied together
    del inps
    for block in model.model.layers:
        block.to(device)
        for i in range(len(xs)):
            xs[i] = _first(block(xs[i], **inp_kwargs))
    if getattr(model.model, "norm", None) is not None:
        model.model.norm.to(device)
        for i in range(len(xs)):
            xs[i] = model.model.norm(xs[i])
    model.lm_head.to(device)
    loss = nn.CrossEntropyLoss()
    nlls = []
    for i, bi in enumerate(range(0, len(encodings), batch_size)):
        logits = model.lm_head(xs[i])
        xs[i] = None
        labels = encodings[bi:bi + batch_size, 1:].to(device)
        flat = logits[:, :-1, :].flatten(end_dim=-2)
        nll = loss(flat.float() if loss_in_fp32 else flat, labels.flatten())
        nlls.extend([nll] * len(logits))
    model.config.use_cache = use_cache
    return torch.stack(nlls).to(torch.float32).mean().exp()


def save_results(results: Dict[str, Dict[str, dict]], path: str) -> None:
    """The reference's on-disk format (main.py:37-38): ``torch.save`` of
    ``{'data': {name: quant_meta}, 'metrics': {name: metrics}}`` -- tensors, numbers and None only,
    so the file loads with ``torch.load(path, weights_only=True)``."""
    torch.save({"data": results["data"], "metrics": results["metrics"]}, path)


@torch.no_grad()
def load_results_into(model: nn.Module, results_or_path, device: torch.device, pack: bool = True,
                      dtype: torch.dtype = torch.float16) -> nn.Module:
    """Puts saved ``quant_meta`` entries (names ``model.layers.<i>.<linear>``) back into a model:
    packed ``QuantLinear`` layers, or dense reconstructed weights as the reference evaluates
    them (quant.py:262-279)."""
    res = torch.load(results_or_path, weights_only=True) if isinstance(results_or_path, str) else results_or_path
    for name, meta in res["data"].items():
        parent_name, leaf = name.rsplit(".", 1)
        parent = model.get_submodule(parent_name)
        if pack:
            setattr(parent, leaf, QuantLinear.from_quant_meta(meta, device))
        else:
            getattr(parent, leaf).weight.data = reconstruct_weight(meta, dtype, device)
    return model


def save_packed(model: nn.Module, path: str) -> None:
    """The packed variant of the on-disk format: ``{'packed': {module name: QuantLinear.packed_state()}}``
    for every ``QuantLinear`` of ``model`` -- 4 (or 3) bits per weight on disk instead of the
    reference's one byte per level (gptq_py.py:217); loads with ``torch.load(weights_only=True)``."""
    packed = {name: m.packed_state() for name, m in model.named_modules() if isinstance(m, QuantLinear)}
    if not packed:
        raise ValueError("save_packed: the model has no QuantLinear layers")
    torch.save({"packed": packed}, path)


@torch.no_grad()
def load_packed_into(model: nn.Module, path: str, device: torch.device) -> nn.Module:
    """Replaces the named Linears of ``model`` by the ``QuantLinear`` layers of a ``save_packed`` file;
    nothing is re-packed, the words go to the device as they are."""
    res = torch.load(path, weights_only=True)
    for name, state in res["packed"].items():
        parent_name, leaf = name.rsplit(".", 1)
        setattr(model.get_submodule(parent_name), leaf, QuantLinear.from_packed_state(state, device))
    return model
