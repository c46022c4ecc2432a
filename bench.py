#!/usr/bin/env python3
"""Benchmark of the GPTQ hot path at the Llama-2-7B configuration (4-bit, g=128, act-order,
128 calibration sequences of 2048 tokens), synthetic activations and weights of the real shapes.

One step = the hot path of ONE decoder block, in the order the reference's driver runs it
(llama.py:108-179): for each input-sharing set  [q,k,v] -> [o] -> [gate,up] -> [down]
    HessianHook.add_batch over all calibration batches   (H += X^T X, MFMA kernel)
    HessianHook.invert                                   (Cholesky-inverse factor U)
    for each Linear of the set:
        gptq_quant (native-loop semantics)               (column loop + trailing fp32 GEMMs)
        reconstruct_nn_linear                            (dense fp16 weight, as the reference keeps)
        QuantLinear.pack
    replay of all batches through the packed Linears of the set (W4A16 GEMM: llama.py:176-177)
The reported value is the wall-clock of the 32-block model: ms_per_step * 32 (all blocks have the
same shapes).  Inputs are resident in HBM before the timed region.

N > 1 (launched by torch.distributed.run): calibration samples are split over ranks, H is
all-reduced over RCCL, every rank inverts, rows of each Linear are split over ranks and
all-gathered, each rank replays its own samples: strong scaling of the same job.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HIDDEN, FFN, SEQ = 4096, 11008, 2048
N_BLOCKS = 32
# (in_features, [out_features of the Linears sharing that input])
SETS = [(HIDDEN, [HIDDEN, HIDDEN, HIDDEN]), (HIDDEN, [HIDDEN]), (HIDDEN, [FFN, FFN]), (FFN, [HIDDEN])]
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak of MI355X


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--samples", type=int, default=128)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "xtx_traffic.json"),
                    help="per-launch HBM bytes of the XtX kernel from a rocprofv3 --pmc run, if present")
    ap.add_argument("--w4-traffic-json", default=os.path.join(ROOT, "profiles", "w4a16_traffic.json"),
                    help="per-launch HBM bytes of the W4A16 kernel from a rocprofv3 --pmc run, if present")
    return ap.parse_args()


class Phase:
    """Sums GPU time of one phase with events on the current stream."""

    def __init__(self):
        self.pairs = []

    def __enter__(self):
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


def make_inputs(dev, n_local, seed):
    gen = torch.Generator(device=dev).manual_seed(seed)
    xs = {}
    for c in (HIDDEN, FFN):
        x = torch.empty(n_local, SEQ, c, dtype=torch.float16, device=dev)
        colscale = 1.0 + 3.0 * torch.arange(c, device=dev) / c
        for i in range(0, n_local, 4):
            x[i:i + 4] = (torch.randn(x[i:i + 4].shape, generator=gen, device=dev) * colscale).half()
        xs[c] = x
    ws = []
    for c, outs in SETS:
        ws.append([(torch.randn(r, c, generator=gen, device=dev) * 0.02).half() for r in outs])
    return xs, ws


VERBOSE = bool(os.environ.get("GQ_BENCH_VERBOSE"))
if VERBOSE:   # a stuck rank shows where: Python stacks of all threads every two minutes
    import faulthandler
    faulthandler.dump_traceback_later(120, repeat=True, file=sys.stderr)


def note(rank, msg):
    if VERBOSE:
        print("[rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)


def run_block(amd, gd, xs, ws, batch, phases, rank, world):
    """The hot path of one decoder block on this rank's share of samples and rows."""
    dev = xs[HIDDEN].device
    for (c, outs), weights in zip(SETS, ws):
        note(rank, "set C=%d" % c)
        x = xs[c]
        hook = amd.HessianHook()
        for bi in range(0, x.shape[0], batch):
            with phases["xtx_%d" % c]:
                hook.add_batch(x[bi:bi + batch])
        if world > 1:
            note(rank, "allreduce")
            with phases["allreduce"]:
                hook.n_samples = gd.allreduce_hessian(hook.hessian, hook.n_samples)
        note(rank, "invert")
        with phases["invert"]:
            hook.invert(damp_ratio=1e-2, act_order=True)
        gs = torch.full([c // 128], 128, dtype=torch.int32)
        gb = torch.full([c // 128], 4, dtype=torch.int32)
        layers = []
        shards = [gd.row_shard(w.shape[0], rank, world) for w in weights]
        with phases["quant"]:
            # the Linears of the set share U: all of them are enqueued (row slices on side streams)
            # before the first result is waited for
            results = amd.gptq_quant_set([w[rb:re] for w, (rb, re) in zip(weights, shards)], hook, gs, gb,
                                         scale_bit_width=None, gptq_use_kernel=True, gptq_block_sizes=gs,
                                         quant_symmetric=False, quant_mse=False, save_device=dev)
        if VERBOSE:
            torch.cuda.synchronize()
            note(rank, "quantised")
        for w, res in zip(weights, results):
            r = w.shape[0]
            meta = res["quant_meta"]
            if world > 1:
                note(rank, "allgather")
                with phases["allgather"]:
                    full = gd.allgather_rows({k: meta[k] for k in ("qweight", "scale", "qzero")}, r)
                    meta = dict(meta, **full)
            note(rank, "reconstruct + pack")
            with phases["reconstruct"]:
                lin = amd.reconstruct_nn_linear(meta, dtype=torch.float16, device=dev)
                del lin
            with phases["pack"]:
                layers.append(amd.QuantLinear.from_quant_meta(meta, dev))
            del meta
        del results
        # replay (llama.py:176-177): the Linears of a set share the input and its act-order, so each
        # batch is gathered into permuted column order once and fed to every layer of the set
        note(rank, "replay")
        for bi in range(0, x.shape[0], batch):
            with phases["permute"]:
                xp = layers[0].permute_input(x[bi:bi + batch])
            for layer in layers:
                with phases["w4a16"]:
                    y = layer.forward_permuted(xp)
                del y
            del xp
        del layers
        del hook


def cpu_baseline():
    """The reference's pure-PyTorch CPU path (gptq_py.py: add_batch(use_kernel=False), invert,
    gptq_quant(gptq_use_kernel=False)) as restated in oracle/, timed on this host on a bounded
    sample and extrapolated to the 32-block model by flop ratios."""
    from oracle import gptq_oracle as oracle
    cores = torch.get_num_threads()
    gen = torch.Generator().manual_seed(0)
    c, r = HIDDEN, 1024
    x = (torch.randn(1, SEQ, c, generator=gen) * (1 + 3 * torch.arange(c) / c)).half()
    w = (torch.randn(r, c, generator=gen) * 0.02).half()
    hook = oracle.OracleHessian()
    t0 = time.perf_counter()
    hook.add_batch(x)
    t_add = time.perf_counter() - t0          # one 2048-token sequence, C = 4096
    t0 = time.perf_counter()
    hook.invert(1e-2, True)
    t_inv = time.perf_counter() - t0          # C = 4096
    gs = torch.full([c // 128], 128, dtype=torch.int32)
    gb = torch.full([c // 128], 4, dtype=torch.int32)
    t0 = time.perf_counter()
    oracle.gptq_quant(w, hook, gs, gb, use_kernel_semantics=False, block_sizes=gs)
    t_q = time.perf_counter() - t0            # 1024 x 4096 layer, python-loop semantics
    k = FFN / HIDDEN
    n_seq = 128
    per_block = (3 * n_seq * t_add + n_seq * t_add * k * k          # Hessians: 2*T*C^2
                 + 3 * t_inv + t_inv * k ** 3                        # inversions: ~C^3
                 + t_q * (4 * 4 + 2 * 4 * k + 4 * k * k))            # column loops: ~R*C^2 from a 1024 x 4096 sample
    return {"value": round(per_block * N_BLOCKS, 1), "unit": "s", "cores": cores, "kind": "port",
            "sample": ("measured: add_batch of one 2048x4096 sequence %.3f s, invert C=4096 %.2f s, "
                       "gptq_quant (PyTorch-loop semantics) of a 1024x4096 layer %.2f s; extrapolated to "
                       "32 blocks x (4 Hessians of 128 sequences, 4 inversions, 7 Linears) by flop ratios; "
                       "the replay GEMMs are not part of gptq_py.py and are not in this figure"
                       % (t_add, t_inv, t_q))}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
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

    sb, se = gd.sample_shard(args.samples, rank, world)
    xs, ws = make_inputs(dev, se - sb, seed=1234)   # same seed on every rank: same weights
    names = ["xtx_%d" % HIDDEN, "xtx_%d" % FFN, "allreduce", "invert", "quant", "allgather", "reconstruct", "pack", "permute", "w4a16"]

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    phases = {n: Phase() for n in names}
    for _ in range(args.warmup):
        run_block(amd, gd, xs, ws, args.batch, phases, rank, world)
    sync()
    phases = {n: Phase() for n in names}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_block(amd, gd, xs, ws, args.batch, phases, rank, world)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed * 1e3 / max(args.steps, 1)

    if rank == 0:
        breakdown = {n: round(p.total_ms() / max(args.steps, 1), 3) for n, p in phases.items() if p.count()}
        def pmc_bytes(path):
            try:
                return json.load(open(path)).get("hbm_bytes_per_launch")
            except Exception:
                return None

        # second kernel by share of the step: the XtX MFMA kernel at C = 4096 (one launch per add_batch call)
        px = phases["xtx_%d" % HIDDEN]
        t_call = px.total_ms() / max(px.count(), 1) * 1e-3
        tokens_call = min(args.batch, se - sb) * SEQ
        flops_call = 2.0 * tokens_call * HIDDEN * HIDDEN        # full product, as the algorithmic figure counts it
        achieved = flops_call / t_call / 1e12 if t_call > 0 else 0.0
        traffic = pmc_bytes(args.traffic_json)
        # dominant kernel (about half of the step's GPU time): the W4A16 GEMM of the replay, one launch
        # per Linear and batch; flops and duration are means over the 7 Linear shapes of the block
        pw = phases["w4a16"]
        local_tokens = (se - sb) * SEQ
        w4_flops = sum(2.0 * local_tokens * r * c for c, outs in SETS for r in outs) * args.steps
        w4_tflops = w4_flops / (pw.total_ms() * 1e-3) / 1e12 if pw.count() else 0.0
        w4_launch_ms = pw.total_ms() / max(pw.count(), 1)
        line = {
            "metric": "llama2_7b_w4g128_quantize_wallclock",
            "value": round(ms_per_step * N_BLOCKS / 1e3, 3),
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
            "config": {"workload": "Llama-2-7B 4-bit g=128 act-order, %d calib sequences x 2048 tokens, "
                                   "1 step = hot path of one decoder block (4 Hessians, 4 inversions, 7 Linears "
                                   "quantised, reconstructed, packed and replayed through the W4A16 GEMM); "
                                   "value = 32 blocks" % args.samples,
                       "batch": args.batch, "parallelism": "dp%d samples + row-sharded quantize" % world},
            "int4_gemm_tflops": round(w4_tflops, 1),
            "int4_gemm_frac_of_f16_mfma_peak": round(w4_tflops / MFMA_F16_PEAK_TFLOPS, 4),
            "breakdown_ms_per_step": breakdown,
            "roofline": {"kernel": "w4a16_256_kernel (y = x W4^T, %d tokens per launch, mean of the 7 Linears of a block)" % tokens_call,
                         "bound": "mfma", "achieved": round(w4_tflops, 1), "peak": MFMA_F16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(w4_tflops / MFMA_F16_PEAK_TFLOPS, 4),
                         "traffic": pmc_bytes(args.w4_traffic_json), "avg_launch_ms": round(w4_launch_ms, 4),
                         "launches": pw.count()},
            "roofline_xtx": {"kernel": "xtx256_kernel (H += X^T X, C=4096, %d tokens per launch)" % tokens_call,
                             "bound": "mfma", "achieved": round(achieved, 1), "peak": MFMA_F16_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(achieved / MFMA_F16_PEAK_TFLOPS, 4), "traffic": traffic,
                             "avg_launch_ms": round(t_call * 1e3, 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
