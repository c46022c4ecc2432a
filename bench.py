#!/usr/bin/env python3
"""Benchmark of the GPTQ hot path (default: the Llama-2-7B configuration -- 4-bit, g=128, act-order,
128 calibration sequences of 2048 tokens), synthetic activations and weights of the real shapes.

One step = the hot path of ONE decoder block, in the order the reference's driver runs it
(llama.py:108-179): for each input-sharing set  [q,k,v] -> [o] -> [gate,up] -> [down]
    HessianHook.add_batch over all calibration batches   (H += X^T X, MFMA kernel)
    HessianHook.invert                                   (Cholesky-inverse factor U)
    for each Linear of the set:
        gptq_quant (native-loop semantics)               (column loop + trailing fp32 GEMMs)
        reconstruct_nn_linear                            (dense 16-bit weight, as the reference keeps)
        QuantLinear.pack                                 (the packed serving form)
    replay of all batches through the reconstructed Linears of the set (llama.py:176-177: dense
    x W^T on the reconstructed weight, on this library's own MFMA kernel)
The reported value is the wall-clock of the whole model: ms_per_step * blocks (all blocks have the
same shapes).  Inputs are resident in HBM before the timed region.  The activations' feature scales (1 .. 4)
are drawn in random order, so the activation order is a permutation without structure (GQ_BENCH_MONOTONE=1:
scales ascending with the column index, the draw of earlier lines in profiles/).

`serving` (rank 0, outside the timed region): QuantLinear.forward -- the packed int4 x fp16 GEMM --
of the 7 Linears of a block at seq 2048 x batch 1 and 8.

N > 1: one process per GPU (launched by torch.distributed.run, or by this script itself when it is
started plainly with --gpus N).  Calibration samples are split over ranks, the lower triangle of H
is reduce-scattered and all-gathered over RCCL, every rank inverts, rows of each Linear are split
over ranks and all-gathered, each rank replays its own samples: strong scaling of the same job.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEQ = 2048
# hidden, ffn, rows of k / v (grouped-query attention in 70B), decoder blocks
MODELS = {
    # OPT-125M (configuration 1 of the baseline: 32 calibration samples, the reference's pure-PyTorch CPU path):
    # q/k/v/out 768 x 768, fc1 3072 x 768, fc2 768 x 3072; fc1 has a Hessian of its own (it reads the second norm)
    "opt125m": dict(name="OPT-125M", hidden=768, ffn=3072, kv=768, blocks=12, opt=True, samples=32),
    "7b": dict(name="Llama-2-7B", hidden=4096, ffn=11008, kv=4096, blocks=32),
    "13b": dict(name="Llama-2-13B", hidden=5120, ffn=13824, kv=5120, blocks=40),
    "70b": dict(name="Llama-2-70B", hidden=8192, ffn=28672, kv=1024, blocks=80),
}
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak of MI355X
MERGE_ROWS = 16384              # Linears of a set are replayed in one launch up to this many rows in all


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", choices=sorted(MODELS), default="7b")
    ap.add_argument("--bits", type=int, choices=(3, 4), default=4)
    ap.add_argument("--samples", type=int, default=None, help="calibration sequences (default 128; 32 for opt125m)")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-serving", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--replay", choices=("dense", "packed", "lib"), default="dense",
                    help="dense: x W^T on the reconstructed weight (what the reference runs); packed: the W4A16 GEMM; "
                         "lib: the dense replay through the library's GEMM, for comparison only")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "gemm_nt_traffic.json"),
                    help="per-launch HBM bytes of the dominant kernel from a rocprofv3 --pmc run, if present")
    return ap.parse_args()


def sets_of(model):
    h, f, kv = model["hidden"], model["ffn"], model["kv"]
    # (in_features, [out_features of the Linears sharing that input])
    if model.get("opt"):
        return [(h, [h, h, h]), (h, [h]), (h, [f]), (f, [h])]
    return [(h, [h, kv, kv]), (h, [h]), (h, [f, f]), (f, [h])]


def self_launch(args):
    """Started plainly with --gpus N > 1: become the launcher.  Nothing has touched the GPU yet (torch
    is not even imported), so the ranks are ordinary child processes of this one."""
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    # the ranks' stdout is read here so that exactly one line, rank 0's JSON, reaches this process's
    # stdout (the gloo backend writes connection notices to stdout); everything else goes to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    result = None
    for line in proc.stdout:
        if line.startswith('{"metric"'):
            result = line
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if result is not None:
        sys.stdout.write(result)
        sys.stdout.flush()
    raise SystemExit(rc if rc else (0 if result is not None else 1))


class Phase:
    """Sums GPU time of one phase with events on the current stream."""

    def __init__(self):
        self.pairs = []
        self.units = 0   # launches inside the phase's windows, where one window holds several

    def __enter__(self):
        import torch
        self.s = torch.cuda.Event(enable_timing=True)
        self.e = torch.cuda.Event(enable_timing=True)
        self.s.record()
        return self

    def __exit__(self, *exc):
        self.e.record()
        self.pairs.append((self.s, self.e))

    def total_ms(self):
        return sum(s.elapsed_time(e) for s, e in self.pairs)

    def count(self):
        return len(self.pairs)


def make_inputs(torch, dev, sets, n_local, seed):
    gen = torch.Generator(device=dev).manual_seed(seed)
    xs = {}
    for c in sorted({c for c, _ in sets}):
        x = torch.empty(n_local, SEQ, c, dtype=torch.float16, device=dev)
        colscale = 1.0 + 3.0 * torch.arange(c, device=dev) / c
        # feature magnitudes in no particular order, as a model's are: the activation order is then a permutation
        # without structure (with the scales ascending it is close to a reversal, which makes every gather by it
        # a nearly contiguous read and flatters the kernels that apply it)
        if not os.environ.get("GQ_BENCH_MONOTONE"):
            colscale = colscale[torch.randperm(c, generator=gen, device=dev)]
        for i in range(0, n_local, 4):
            x[i:i + 4] = (torch.randn(x[i:i + 4].shape, generator=gen, device=dev) * colscale).half()
        xs[c] = x
    ws = []
    for c, outs in sets:
        ws.append([(torch.randn(r, c, generator=gen, device=dev) * 0.02).half() for r in outs])
    return xs, ws


VERBOSE = bool(os.environ.get("GQ_BENCH_VERBOSE"))


def note(rank, msg):
    if VERBOSE:
        print("[rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)


def run_block(torch, amd, gd, sets, xs, ws, args, phases, rank, world, flops):
    """The hot path of one decoder block on this rank's share of samples and rows."""
    dev = next(iter(xs.values())).device
    batch = args.batch
    for (c, outs), weights in zip(sets, ws):
        note(rank, "set C=%d" % c)
        x = xs[c]
        hook = amd.HessianHook(overlap_batches=not os.environ.get("GQ_NO_OVERLAP"))   # the bench's inputs stay as they are
        # one window per set: consecutive calls alternate between two accumulators on two streams and overlap at
        # their ends; reading `hessian` joins the streams and sums the accumulators (part of the accumulation)
        with phases["xtx_%d" % c] as ph:
            for bi in range(0, x.shape[0], batch):
                hook.add_batch(x[bi:bi + batch])
                ph.units += 1
            hook.hessian
        if world > 1:
            with phases["allreduce"]:
                hook.n_samples = gd.allreduce_hessian(hook.hessian, hook.n_samples, n_total=args.samples)
        with phases["invert"]:
            # enqueued without a host round trip; gptq_quant_set reads the verdict with its metrics
            hook.invert(damp_ratio=1e-2, act_order=True, wait=False)
        gs = torch.full([c // 128], 128, dtype=torch.int32)
        gb = torch.full([c // 128], args.bits, dtype=torch.int32)
        shards = [gd.row_shard(w.shape[0], rank, world) for w in weights]
        with phases["quant"]:
            # the Linears of the set share U: all of them are enqueued (on side streams) and nothing is waited
            # for; the host goes on to enqueue what consumes the levels and settles the set afterwards
            mine = [(i, w[rb:re]) for i, (w, (rb, re)) in enumerate(zip(weights, shards)) if re > rb]
            part = amd.gptq_quant_set([w for _, w in mine], hook, gs, gb, scale_bit_width=None, gptq_use_kernel=True,
                                      gptq_block_sizes=gs, quant_symmetric=False, quant_mse=False, save_device=dev,
                                      wait=False)
        while True:
            set_flops = consume_set(torch, amd, gd, args, phases, world, dev, c, x, weights, mine, part, hook, gs, gb)
            # metrics and the verdict of the inversion's first try, read while the replay runs; if that try had
            # failed the set has been quantised again from the final U and its consumers run once more
            if part.settle():
                break
        flops[0] += set_flops
        del part, hook


def consume_set(torch, amd, gd, args, phases, world, dev, c, x, weights, mine, part, hook, gs, gb):
    """Reconstruct, pack and replay the Linears of one set; returns the replay's flops."""
    batch = args.batch
    done = 0.0
    dense, packed = [], []
    if world > 1:
        # one all-gather (levels two to a byte, scales, zero points) and one all-reduce (metric sums) per set, on the
        # device: the host does not wait here
        with phases["allgather"]:
            fields = {"qscale": None, "sscale": None, "scale_bit_width": None, "perm_inv": hook.perm_inv.to(torch.int16),
                      "group_sizes": gs.to(torch.int16).to(dev), "group_bit_widths": gb.to(torch.uint8).to(dev)}
            local_meta = [None] * len(weights)
            local_sums = torch.zeros(len(weights), 4, dtype=torch.float64, device=dev)
            mine_sums = part.metric_sums() if mine else None
            for k, ((i, _), res) in enumerate(zip(mine, part)):
                local_meta[i] = res["quant_meta"]
                local_sums[i] = mine_sums[k]
            parts, _sums = gd.gather_quant_set(local_meta, local_sums, [w.shape[0] for w in weights], c, len(gs),
                                               weights[0].dtype, dev, args.bits)
            metas = [dict(fields, **p_) for p_ in parts]
    else:
        metas = [res["quant_meta"] for res in part]
    for meta in metas:
        with phases["reconstruct"]:
            dense.append(amd.quant.reconstruct_weight(meta, torch.float16, dev))
        with phases["pack"]:
            packed.append(amd.QuantLinear.from_quant_meta(meta, dev, validate=False, has_perm=True, dense3=args.bits == 3))
    del metas
    # replay (llama.py:176-177).  Dense: the Linears of a set read the same input, so up to MERGE_ROWS
    # rows of reconstructed weights are stacked and replayed in one launch per batch (each output row
    # is the same dot product either way).  Packed: one gather of the batch into act-order per set,
    # then the W4A16 GEMM per Linear.
    if args.replay in ("dense", "lib"):
        if len(dense) > 1 and sum(w.shape[0] for w in dense) <= MERGE_ROWS:
            with phases["reconstruct"]:
                dense = [torch.cat(dense, dim=0)]
        for bi in range(0, x.shape[0], batch):
            x2 = x[bi:bi + batch].reshape(-1, c)
            for w in dense:
                with phases["replay_gemm"]:
                    # "lib": the library's fp16 GEMM on the same operands, a reference point for the kernel
                    y = amd.ops.gemm_nt(x2, w) if args.replay == "dense" else torch.matmul(x2, w.t())
                done += 2.0 * x2.shape[0] * w.shape[0] * c
                del y
    else:
        for bi in range(0, x.shape[0], batch):
            with phases["permute"]:
                xp = packed[0].permute_input(x[bi:bi + batch])
            for layer in packed:
                with phases["replay_gemm"]:
                    y = layer.forward_permuted(xp)
                done += 2.0 * xp.shape[0] * xp.shape[1] * layer.out_features * c
                del y
            del xp
    return done


def serving(torch, amd, sets, bits, dev):
    """QuantLinear.forward (activation gather + W4A16 GEMM) of the 7 Linears of a block at seq 2048 x
    batch 1 and 8, random levels / scales / act-order of the real shapes."""
    gen = torch.Generator(device=dev).manual_seed(7)
    layers = []
    for c, outs in sets:
        for r in outs:
            meta = {"qweight": torch.randint(0, 1 << bits, (r, c), generator=gen, device=dev, dtype=torch.uint8),
                    "scale": (torch.rand(r, c // 128, generator=gen, device=dev) * 0.01 + 0.002).half(),
                    "qzero": torch.randint(0, 1 << bits, (r, c // 128), generator=gen, device=dev, dtype=torch.uint8),
                    "perm_inv": torch.argsort(torch.randperm(c, generator=gen, device=dev)).to(torch.int16),
                    "group_sizes": torch.full([c // 128], 128, dtype=torch.int16),
                    "group_bit_widths": torch.full([c // 128], bits, dtype=torch.uint8)}
            layers.append((c, r, amd.QuantLinear.from_quant_meta(meta, dev, validate=False, has_perm=True, dense3=bits == 3)))
            del meta
    out = {"weight_bits_in_memory": 3 if bits == 3 else 4,
           "packed_weight_bytes_per_block": int(sum(layer.qweight.numel() * 4 for _, _, layer in layers))}
    fused = None
    if len(sets) == 4 and len(sets[0][1]) == 3 and len(sets[2][1]) == 2:   # the Llama block: q/k/v, o, gate/up, down
        # the parts of a concatenation share one act-order (they are quantised against one Hessian)
        by_set, at = [], 0
        for c, outs in sets:
            by_set.append([layer for _, _, layer in layers[at:at + len(outs)]])
            at += len(outs)
        for group in by_set:
            for layer in group[1:]:
                layer.perm = group[0].perm
        fused = [(sets[i][0], amd.QuantLinear.concat(group) if len(group) > 1 else group[0]) for i, group in enumerate(by_set)]
    for batch in (1, 8):
        m = batch * SEQ
        xin = {c: torch.randn(m, c, generator=gen, device=dev).half() for c in {c for c, _, _ in layers}}
        flops = sum(2.0 * m * r * c for c, r, _ in layers)
        for _ in range(2):
            for c, r, layer in layers:
                layer(xin[c])
        torch.cuda.synchronize()
        reps = 10
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            for c, r, layer in layers:
                layer(xin[c])
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / reps
        # the GEMMs alone (input already in act-order), for the matrix-pipe figure, in both forms of the forward:
        # "fused" = the packed kernels, levels expanded inside the GEMM (w4a16_kernel / w4a16_256_kernel /
        # wq_stream_kernel); "expand_once" = the levels expanded once per call into a transient dense weight
        # (gq_wq_expand, inside the timed call) and the dense kernel on it.  QuantLinear.forward picks by token count.
        xp = {c: xin[c] for c in xin}

        def gemm_only(min_tokens):
            saved = amd.QuantLinear.expand_min_tokens
            amd.QuantLinear.expand_min_tokens = min_tokens
            try:
                for c, r, layer in layers:
                    layer.forward_permuted(xp[c])
                s2, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s2.record()
                for _ in range(reps):
                    for c, r, layer in layers:
                        layer.forward_permuted(xp[c])
                e2.record()
                torch.cuda.synchronize()
                return s2.elapsed_time(e2) / reps
            finally:
                amd.QuantLinear.expand_min_tokens = saved

        ms_gemm = gemm_only(amd.QuantLinear.expand_min_tokens)
        ms_fused = gemm_only(0)
        ms_expand = gemm_only(1)
        tf = flops / (ms_gemm * 1e-3) / 1e12
        out["seq2048_batch%d" % batch] = {"tokens": m, "block_forward_ms": round(ms, 4), "w4a16_gemm_ms": round(ms_gemm, 4),
                                          "w4a16_tflops": round(tf, 1),
                                          "frac_of_f16_mfma_peak": round(tf / MFMA_F16_PEAK_TFLOPS, 4),
                                          "fused_w4a16_tflops": round(flops / (ms_fused * 1e-3) / 1e12, 1),
                                          "expand_once_tflops": round(flops / (ms_expand * 1e-3) / 1e12, 1)}
        # the Linears that read the same input as one layer each (QuantLinear.concat): q/k/v, gate/up
        if fused is not None:
            s3, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for c, layer in fused:
                layer.forward_permuted(xp[c])
            s3.record()
            for _ in range(reps):
                for c, layer in fused:
                    layer.forward_permuted(xp[c])
            e3.record()
            torch.cuda.synchronize()
            ms_f = s3.elapsed_time(e3) / reps
            out["seq2048_batch%d" % batch].update({"fused_qkv_gateup_gemm_ms": round(ms_f, 4),
                                                   "fused_qkv_gateup_tflops": round(flops / (ms_f * 1e-3) / 1e12, 1)})
        del xin, xp
    out["expand_once_from_tokens"] = amd.QuantLinear.expand_min_tokens
    out["forward_picks"] = ("w4a16_tflops is QuantLinear.forward as shipped: the fused packed kernel below %d tokens per call "
                            "(twice that for layers of fewer than 8192 rows), expand-once from there on; fused_w4a16_tflops and "
                            "expand_once_tflops force one form for every layer" % amd.QuantLinear.expand_min_tokens)
    return out


def cpu_baseline(sets, n_blocks, n_seq, bits):
    """The reference's pure-PyTorch CPU path (gptq_py.py: add_batch(use_kernel=False), invert,
    gptq_quant(gptq_use_kernel=False)) as restated in oracle/, timed on this host's cores on a
    bounded sample of the same workload: per Hessian width one 2048-token sequence of add_batch and
    one inversion, per distinct Linear shape one whole gptq_quant (a row slice, stated, when the
    layer is too large for the time budget).  The block's time is the sum over its own counts of
    sequences, inversions and Linears; nothing is extrapolated across sizes."""
    import torch
    from oracle import gptq_oracle as oracle
    cores = torch.get_num_threads()
    gen = torch.Generator().manual_seed(0)
    widths = sorted({c for c, _ in sets})
    t_add, t_inv, t_layer, notes = {}, {}, {}, []
    for c in widths:
        colscale = 1 + 3 * torch.arange(c) / c
        x = (torch.randn(1, SEQ, c, generator=gen) * colscale).half()
        hook = oracle.OracleHessian()
        t0 = time.perf_counter()
        hook.add_batch(x)
        t_add[c] = time.perf_counter() - t0
        for _ in range((c + SEQ - 1) // SEQ + 1):   # a full-rank Hessian for the inversion and the loop
            hook.add_batch((torch.randn(1, SEQ, c, generator=gen) * colscale).half())
        t0 = time.perf_counter()
        hook.invert(1e-2, True)
        t_inv[c] = time.perf_counter() - t0
        gs = torch.full([c // 128], 128, dtype=torch.int32)
        gb = torch.full([c // 128], bits, dtype=torch.int32)
        for r in sorted({r for cc, outs in sets if cc == c for r in outs}):
            rows = r if r * c <= (1 << 26) else max(256, ((1 << 26) // c) // 256 * 256)
            w = (torch.randn(rows, c, generator=gen) * 0.02).half()
            t0 = time.perf_counter()
            oracle.gptq_quant(w, hook, gs, gb, use_kernel_semantics=False, block_sizes=gs)
            dt = time.perf_counter() - t0
            t_layer[(r, c)] = dt * (r / rows)
            notes.append("gptq_quant %dx%d %.2f s%s" % (r, c, dt, "" if rows == r else " for %d of its rows (scaled by rows)" % rows))
            del w
        notes.append("C=%d: add_batch of one 2048-token sequence %.3f s, invert %.2f s" % (c, t_add[c], t_inv[c]))
        del hook, x
    per_block = 0.0
    for c, outs in sets:
        per_block += n_seq * t_add[c] + t_inv[c] + sum(t_layer[(r, c)] for r in outs)
    return {"value": round(per_block * n_blocks, 1), "unit": "s", "cores": cores, "kind": "port",
            "sample": "measured on this host (PyTorch-loop semantics) -- " + "; ".join(notes) + "; one block = its 4 Hessians of %d "
                      "sequences, 4 inversions and 7 Linears at these times, x %d blocks; the replay GEMMs are not part of "
                      "gptq_py.py and are not in this figure" % (n_seq, n_blocks)}


def cpu_baseline_whole(sets, n_blocks, n_seq, batch, bits, budget_s=25.0):
    """Configuration 1 (OPT-125M): the reference's pure-PyTorch CPU path run as a whole, block after block in the
    order llama.py runs a block -- per set add_batch over every calibration batch, invert, gptq_quant
    (PyTorch-loop semantics) of each Linear -- through the oracle's restatement, not composed from separately
    timed pieces.  Whole blocks are run until the time budget is used (at least one); all blocks have the same
    shapes, so the model's figure is the per-block mean times the block count."""
    import torch
    from oracle import gptq_oracle as oracle
    cores = torch.get_num_threads()
    gen = torch.Generator().manual_seed(0)
    xs = {c: (torch.randn(n_seq, SEQ, c, generator=gen) * (1 + 3 * torch.rand(c, generator=gen))).half()
          for c in sorted({c for c, _ in sets})}
    done, t_total = 0, 0.0
    while done < n_blocks and (done == 0 or t_total + t_total / done <= budget_s):
        ws = [[(torch.randn(r, c, generator=gen) * 0.02).half() for r in outs] for c, outs in sets]
        t0 = time.perf_counter()
        for (c, outs), weights in zip(sets, ws):
            hook = oracle.OracleHessian()
            for bi in range(0, n_seq, batch):
                hook.add_batch(xs[c][bi:bi + batch])
            hook.invert(1e-2, True)
            gs = torch.full([c // 128], 128, dtype=torch.int32)
            gb = torch.full([c // 128], bits, dtype=torch.int32)
            for w in weights:
                oracle.gptq_quant(w, hook, gs, gb, use_kernel_semantics=False, block_sizes=gs)
        t_total += time.perf_counter() - t0
        done += 1
    return {"value": round(t_total / done * n_blocks, 1), "unit": "s", "cores": cores, "kind": "port",
            "sample": "measured on this host: %d whole block(s) of the oracle's PyTorch-loop path (4 Hessians over %d sequences "
                      "of 2048 tokens, 4 inversions, %d Linears each) in %.1f s; value = per-block mean x %d blocks"
                      % (done, n_seq, sum(len(o) for _, o in sets), t_total, n_blocks)}


def end_to_end(torch, dev, bits, n_seq, batch):
    """The block walk itself -- gptq_v2_amd.driver.quantize_llama with attention, norms, rotary embedding, SwiGLU, hooks
    and the copies of the results to the host -- on a random-initialised Llama with Llama-2-7B block dimensions
    (no checkpoint is reachable offline), outside the timed region of the bench line.  Walks of 1 and of 5 blocks
    are timed whole (host clock, device drained at both ends) after an untimed warm-up walk; their difference is four
    blocks without the per-walk costs (embedding, capture of the first block's inputs, the last block's results on
    their way to the host)."""
    try:
        from transformers import LlamaConfig, LlamaForCausalLM
    except Exception as exc:   # the walk needs the model classes only
        return {"error": "transformers not importable: %r" % (exc,)}
    from gptq_v2_amd import driver
    enc = torch.randint(0, 32000, (n_seq, SEQ), generator=torch.Generator().manual_seed(11))
    walls = {}
    for n_blocks in (0, 1, 5):   # 0: an untimed one-block walk that takes the first-call costs (library handles, kernel loading)
        cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=max(n_blocks, 1), num_attention_heads=32,
                          num_key_value_heads=32, vocab_size=32000, max_position_embeddings=SEQ)
        torch.manual_seed(0)
        with torch.device(dev):
            model = LlamaForCausalLM(cfg).half().eval()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        driver.quantize_llama(model, enc, dev, batch_size=batch, bits=bits, group_size=128)
        torch.cuda.synchronize()
        walls[n_blocks] = time.perf_counter() - t0
        del model
        torch.cuda.empty_cache()
    per_block = (walls[5] - walls[1]) / 4.0
    return {"what": "driver.quantize_llama on a random-init Llama with Llama-2-7B block dimensions, %d x %d tokens, batch %d: "
                    "the real block walk (attention, norms, hooks, results copied to the host)" % (n_seq, SEQ, batch),
            "wall_s_1_block": round(walls[1], 3), "wall_s_5_blocks": round(walls[5], 3),
            "seconds_per_block": round(per_block, 4),
            "value_32_blocks": round(32 * per_block, 2),
            "value_32_blocks_is": "extrapolated: 32 x the marginal block of the 5-block walk; no 32-block model was run",
            "reference_log": "123.45 s for the 32 blocks on an unnamed GPU (logs/log.txt:260), 3.78-3.97 s per block"}


def device_peak(torch, dev):
    """The dense 16-bit matrix peak from what the device reports: CUs x maximum clock x 4096 flops per clock and CU
    (4 SIMDs x 1024: one v_mfma_f32_16x16x32_f16, 16384 flops, per 16 cycles and SIMD).  `peak` itself stays the
    vendor's dense figure, 2.5 PFLOP/s."""
    try:
        import ctypes
        from gptq_v2_amd import _lib
        cus, khz = ctypes.c_int(0), ctypes.c_int(0)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().gq_device_info(ctypes.byref(cus), ctypes.byref(khz)), "gq_device_info")
        return {"cus": cus.value, "max_clock_mhz": round(khz.value / 1e3, 1), "flops_per_clock_per_cu": 4096,
                "tflops": round(cus.value * khz.value * 1e3 * 4096 / 1e12, 1)}
    except Exception as exc:
        return {"error": repr(exc)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if VERBOSE:   # a stuck rank shows where: Python stacks of all threads every two minutes
        import faulthandler
        faulthandler.dump_traceback_later(120, repeat=True, file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the hot path has no CPU fallback)")
    # GQ_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share
    # devices, collectives go through gloo): it checks the multi-rank code path, not its speed.
    backend = os.environ.get("GQ_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    import gptq_v2_amd as amd
    from gptq_v2_amd import dist as gd

    model = MODELS[args.model]
    if args.samples is None:
        args.samples = model.get("samples", 128)
    sets = sets_of(model)
    n_blocks = model["blocks"]
    sb, se = gd.sample_shard(args.samples, rank, world)
    xs, ws = make_inputs(torch, dev, sets, se - sb, seed=1234)   # same seed on every rank: same weights
    widths = sorted({c for c, _ in sets})
    names = ["xtx_%d" % c for c in widths] + ["allreduce", "invert", "quant", "allgather", "reconstruct", "pack", "permute",
                                               "replay_gemm"]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    phases = {n: Phase() for n in names}
    flops = [0.0]
    for _ in range(args.warmup):
        run_block(torch, amd, gd, sets, xs, ws, args, phases, rank, world, flops)
    sync()
    phases = {n: Phase() for n in names}
    flops = [0.0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_block(torch, amd, gd, sets, xs, ws, args, phases, rank, world, flops)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)

    if rank == 0:
        steps = max(args.steps, 1)
        breakdown = {n: round(p.total_ms() / steps, 3) for n, p in phases.items() if p.count()}

        def pmc_bytes(path, source):
            """Memory-side bytes per launch from the committed rocprofv3 --pmc passes -- only while the kernel's source
            is the one they were measured on (the file carries its SHA-256); a stale measurement reads as null."""
            import hashlib
            try:
                rec = json.load(open(path))
                now = hashlib.sha256(open(os.path.join(ROOT, "gptq_v2_amd", "csrc", source), "rb").read()).hexdigest()
                return rec.get("hbm_bytes_per_launch") if rec.get("source_sha256") == now else None
            except Exception:
                return None

        hidden = model["hidden"]
        px = phases["xtx_%d" % hidden]
        t_call = px.total_ms() / max(px.units, 1) * 1e-3   # per add_batch call, the final sum of the accumulators included
        tokens_call = min(args.batch, se - sb) * SEQ
        xtx_tflops = 2.0 * tokens_call * hidden * hidden / t_call / 1e12 if t_call > 0 else 0.0   # full product, as the algorithmic figure counts it

        def xtx_figures(c):
            """algorithmic (full product 2 N C^2) and executed (the lower 256 x 256 tiles the kernel computes, diagonal
            tiles whole) TFLOP/s of the XtX calls of width c"""
            ph = phases.get("xtx_%d" % c)
            if ph is None or not ph.units or c % 256:
                return None
            t = ph.total_ms() / ph.units * 1e-3
            nt = c // 256
            executed = nt * (nt + 1) // 2 * 2.0 * 256 * 256 * tokens_call
            return {"C": c, "avg_call_ms": round(t * 1e3, 4), "algorithmic_tflops": round(2.0 * tokens_call * c * c / t / 1e12, 1),
                    "executed_tflops": round(executed / t / 1e12, 1),
                    "executed_frac_of_f16_mfma_peak": round(executed / t / 1e12 / MFMA_F16_PEAK_TFLOPS, 4),
                    "lower_tiles": nt * (nt + 1) // 2}

        # dominant kernel: the replay GEMM; algorithmic flops 2 M R C per launch and launch duration,
        # both as means over the launches of a block
        pr = phases["replay_gemm"]
        replay_tflops = flops[0] / (pr.total_ms() * 1e-3) / 1e12 if pr.count() else 0.0
        replay_kernel = {"dense": "gemm_nt_kernel (y = x W^T on the reconstructed fp16 weight",
                         "lib": "the library's fp16 GEMM (torch.matmul; comparison run, not the product path",
                         "packed": "w4a16_256_kernel (y = x W4^T"}[args.replay]
        line = {
            "metric": ("opt125m_w%dg128_quantize_wallclock" % args.bits) if model.get("opt") else
                      "llama2_%s_w%dg128_quantize_wallclock" % (args.model, args.bits),
            "value": round(ms_per_step * n_blocks / 1e3, 3),
            "unit": "s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f16",
            "data": "synthetic",
            "config": {"workload": "%s %d-bit g=128 act-order, %d calib sequences x 2048 tokens, "
                                   "1 step = hot path of one decoder block (4 Hessians, 4 inversions, 7 Linears "
                                   "quantised, reconstructed, packed, and replayed as %s GEMMs); "
                                   "value = %d blocks.  Against the reference's operation list (llama.py:141-179) the step "
                                   "merges the replay of Linears that read one input into one launch per batch and leaves "
                                   "out the copy of each quant_meta to the host (llama.py:173; the driver makes it on a "
                                   "thread of its own, see end_to_end)" % (model["name"], args.bits, args.samples, args.replay, n_blocks),
                       "batch": args.batch, "parallelism": "dp%d samples + row-sharded quantize" % world},
            "breakdown_ms_per_step": breakdown,
            "roofline": {"kernel": "%s, %d tokens per launch, mean of the launches of a block)" % (replay_kernel, tokens_call),
                         "bound": "mfma", "achieved": round(replay_tflops, 1), "peak": MFMA_F16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(replay_tflops / MFMA_F16_PEAK_TFLOPS, 4),
                         "traffic": pmc_bytes(args.traffic_json, "gemm_nt.hip") if args.replay == "dense" else None,
                         "avg_launch_ms": round(pr.total_ms() / max(pr.count(), 1), 4),
                         "avg_launch_flops": round(flops[0] / max(pr.count(), 1), 1), "launches": pr.count(),
                         "peak_from_device": device_peak(torch, dev)},
            "roofline_xtx": {"kernel": "xtx256_kernel (H += X^T X, C=%d, %d tokens per launch)" % (hidden, tokens_call),
                             "bound": "mfma", "achieved": round(xtx_tflops, 1), "peak": MFMA_F16_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(xtx_tflops / MFMA_F16_PEAK_TFLOPS, 4),
                             "traffic": pmc_bytes(os.path.join(ROOT, "profiles", "xtx_traffic.json"), "xtx.hip"),
                             "avg_launch_ms": round(t_call * 1e3, 4),
                             "achieved_is": "algorithmic: the full product's 2 N C^2 flops; the kernel computes the tiles on and "
                                            "below the diagonal only, see by_width[].executed_tflops",
                             "by_width": [f for f in (xtx_figures(c) for c in widths) if f],
                             "timing": "window of a set's add_batch calls and the final sum, divided by the calls "
                                       "(consecutive calls overlap on two streams), not one launch alone"},
        }
        if world > 1:
            h_bytes = sum(4.0 * c * c for c, _ in sets)
            line["collectives"] = {"allreduce_ms_per_step": breakdown.get("allreduce"),
                                   "allgather_ms_per_step": breakdown.get("allgather"),
                                   "hessian_bytes_per_step": h_bytes,
                                   "allreduce_algorithmic_GBps": round(h_bytes / max(breakdown.get("allreduce", 0.0), 1e-9) / 1e6, 1)}
        del xs, ws
        torch.cuda.empty_cache()
        if world == 1 and not args.no_serving:
            line["serving"] = serving(torch, amd, sets, args.bits, dev)
            best = line["serving"]["seq2048_batch8"]
            line["int4_gemm_tflops"] = best["w4a16_tflops"]
            line["int4_gemm_frac_of_f16_mfma_peak"] = best["frac_of_f16_mfma_peak"]
            line["int4_gemm_tflops_is"] = ("QuantLinear.forward at 16384 tokens = expand-once: packed weights at rest, expanded once per "
                                           "call into a transient dense weight inside the timed call; the fused kernel's own figure is "
                                           "serving.seq2048_batch8.fused_w4a16_tflops")
        if world == 1 and not args.no_end_to_end and args.model == "7b":
            line["end_to_end"] = end_to_end(torch, dev, args.bits, args.samples, args.batch)
        if world == 1 and not args.no_cpu_baseline:
            if model.get("opt"):
                line["cpu_baseline"] = cpu_baseline_whole(sets, n_blocks, args.samples, args.batch, args.bits)
            else:
                line["cpu_baseline"] = cpu_baseline(sets, n_blocks, args.samples, args.bits)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
