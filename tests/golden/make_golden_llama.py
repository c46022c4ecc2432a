#!/usr/bin/env python3
"""Generates tests/golden/f10_tiny_llama.npz by RUNNING the reference's own callers of the hot path:
``quantize_llama`` (llama.py:65-204, with model_utils.py's Catcher / RecorderWrapper /
extract_dependencies) and ``evaluate_llama`` (llama.py:207-247), imported from /root/reference, on a
two-block random-initialised Llama built from a ``LlamaConfig`` (no checkpoint, no dataset, no fetch).

Run once in the build container:  python tests/golden/make_golden_llama.py
The fixture (inputs + the reference's outputs) is committed; the reference itself never travels.

What stands in for what the container lacks (a CUDA device and the CUDA extension):
  * module ``gptq``: ``gptq_quantize_range`` is the C restatement of gptq_cpp.cpp / gptq_cpp_kernel.cu
    (oracle/kernel_path_ref.c), as in make_golden.py;
  * ``gptq_py.accumulate_hessian_t`` (the Triton op RecorderWrapper reaches through
    ``add_batch(use_kernel=True)``, model_utils.py:102, gptq_py.py:36) is the reference's own fp32
    ``addmm`` branch (gptq_py.py:31-33);
  * ``torch.cuda.set_device`` (gptq_py.py:127) is a no-op; everything runs on the CPU in fp16.
The walk itself -- which Linears share a Hessian, in which order the sets are quantised, which
replayed outputs feed which Hessian, how the perplexity is averaged -- is reference code only.

Two conventions of decoder-block outputs are covered.  With the transformers installed here a block
returns a bare tensor and the reference's 1-tuple unpack (llama.py:188, 223) only works on a batch of
one: the main run uses ``batch_size=1`` and nothing is patched.  The ``*_b4`` perplexities come from a
second pass in which ``LlamaDecoderLayer.forward`` returns ``(hidden,)`` -- the convention of the
transformers release the reference was written against -- so that a ragged last batch (14 sequences
in batches of 4) pins the per-batch weighting of llama.py:235-243.
"""
import copy
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from oracle import gptq_oracle as oracle  # noqa: E402

stub = types.ModuleType("gptq")


def _no_native(*_a, **_k):
    raise RuntimeError("native accumulate_hessian is not available in the fixture generator")


stub.accumulate_hessian = _no_native
stub.gptq_quantize_range = lambda *a: oracle.quantize_range_kernel(*a)
sys.modules["gptq"] = stub
sys.path.insert(0, REF)
import gptq_py as ref_gptq  # noqa: E402  (the reference)


def _addmm_branch(h, x, save_lower_only, compute_lower_only=True):
    x = x.to(dtype=torch.float32)                      # gptq_py.py:32-33
    torch.addmm(h, x.t(), x, out=h)


ref_gptq.accumulate_hessian_t = _addmm_branch
torch.cuda.set_device = lambda *_a, **_k: None

import llama as ref_llama  # noqa: E402  (the reference)

from tests._tiny_llama import CONFIG, N_SEQ, state_digest, tiny_llama, tokens  # noqa: E402


class LoggingHook(ref_gptq.HessianHook):
    """The reference's hook, recording what each input-sharing set saw."""
    log = []

    def invert(self, *args, **kwargs):
        raw_diag = self.hessian.diagonal().clone()
        n = self.n_samples
        u = super().invert(*args, **kwargs)
        LoggingHook.log.append(dict(n_samples=n, raw_diag=raw_diag, hessian_after=self.hessian.clone(),
                                    perm=self.perm.clone(), hessian_inv=u.clone()))
        return u


ref_llama.HessianHook = LoggingHook


def main():
    cpu = torch.device("cpu")
    model = tiny_llama()
    digest = state_digest(model)
    enc = tokens()

    out = {"enc": enc.numpy(), "state_sha256": np.frombuffer(digest.encode(), dtype=np.uint8)}
    out["ppl_fp_b1"] = np.float64(float(ref_llama.evaluate_llama(copy.deepcopy(model), enc, cpu, batch_size=1)))

    qmodel = copy.deepcopy(model)
    res = ref_llama.quantize_llama(qmodel, enc, cpu, batch_size=1)
    out["ppl_q_b1"] = np.float64(float(ref_llama.evaluate_llama(qmodel, enc, cpu, batch_size=1)))

    names = list(res["data"].keys())               # the order the reference quantised the Linears in
    out["names"] = np.array(names)
    for i, name in enumerate(names):
        meta, metrics = res["data"][name], res["metrics"][name]
        for k in ("qweight", "scale", "qzero", "perm_inv", "group_sizes", "group_bit_widths"):
            out[f"L{i}_{k}"] = meta[k].numpy()
        out[f"L{i}_metrics"] = np.array([metrics["gptq_error"], metrics["gptq_norm"], metrics["l2.0_error"],
                                         metrics["l2.0_norm"]], dtype=np.float64)
        # the reconstructed weight the reference left in the model (llama.py:172-174, 199)
        w = qmodel.get_submodule(name).weight.data
        assert torch.equal(w, oracle.construct_weight(meta, torch.float16))
    # which Linears shared a Hessian: consecutive names with equal perm_inv and equal in_features whose
    # set was opened by one LoggingHook.invert
    sets, i = [], 0
    for entry in LoggingHook.log:
        c = entry["perm"].numel()
        members = []
        while i < len(names) and res["data"][names[i]]["qweight"].shape[1] == c and \
                torch.equal(res["data"][names[i]]["perm_inv"].long(), torch.argsort(entry["perm"])):
            members.append(i)
            i += 1
        assert members
        sets.append(members)
    assert i == len(names) and len(sets) == len(LoggingHook.log)
    out["set_start"] = np.array([s[0] for s in sets] + [len(names)], dtype=np.int64)
    for si, entry in enumerate(LoggingHook.log):
        out[f"S{si}_n_samples"] = np.int64(entry["n_samples"])
        out[f"S{si}_raw_diag"] = entry["raw_diag"].numpy()
        out[f"S{si}_after_diag"] = entry["hessian_after"].diagonal().numpy().copy()
        out[f"S{si}_perm"] = entry["perm"].numpy()
        out[f"S{si}_u_diag"] = entry["hessian_inv"].diagonal().numpy().copy()
    # yardstick (not reference output): the same diagonals had every set seen the UNQUANTISED block's
    # activations -- what a walk that forgot to replay the quantised layers' outputs would accumulate
    fp_diag = {}

    def watch(name):
        def pre(_mod, args):
            x = args[0].reshape(-1, args[0].shape[-1]).float()
            fp_diag[name] = fp_diag.get(name, 0) + (x * x).sum(0)
        return pre

    fp_model = copy.deepcopy(model)
    handles = [fp_model.get_submodule(names[s[0]]).register_forward_pre_hook(watch(names[s[0]])) for s in sets]
    with torch.no_grad():
        for bi in range(N_SEQ):
            fp_model(enc[bi:bi + 1])
    for h_ in handles:
        h_.remove()
    for si, s in enumerate(sets):
        out[f"S{si}_raw_diag_fpwalk"] = fp_diag[names[s[0]]].numpy()
    # the first set of the first block in full: its inputs are embedding + first norm only
    out["S0_hessian_after"] = LoggingHook.log[0]["hessian_after"].numpy()
    out["S0_hessian_inv"] = LoggingHook.log[0]["hessian_inv"].numpy()

    # second convention: blocks return a 1-tuple, batches of 4 with a ragged tail
    from transformers.models.llama import modeling_llama
    plain_forward = modeling_llama.LlamaDecoderLayer.forward

    def tuple_forward(self, *args, **kwargs):
        y = plain_forward(self, *args, **kwargs)
        return y if isinstance(y, tuple) else (y,)

    modeling_llama.LlamaDecoderLayer.forward = tuple_forward
    try:
        out["ppl_fp_b4"] = np.float64(float(ref_llama.evaluate_llama(copy.deepcopy(model), enc, cpu, batch_size=4)))
        out["ppl_q_b4"] = np.float64(float(ref_llama.evaluate_llama(qmodel, enc, cpu, batch_size=4)))
        LoggingHook.log = []
        qmodel4 = copy.deepcopy(model)
        res4 = ref_llama.quantize_llama(qmodel4, enc, cpu, batch_size=4)
    finally:
        modeling_llama.LlamaDecoderLayer.forward = plain_forward
    # batches of 4 change the order of the fp32 sums in H only: record how far the levels move, as a
    # yardstick for what a different (GPU) summation order may legitimately do
    # (the order inside a set follows object ids in extract_dependencies, model_utils.py:181, and may differ per run)
    assert sorted(res4["data"].keys()) == sorted(names)
    out["b4_level_mismatch"] = np.array([
        float((res4["data"][n]["qweight"] != res["data"][n]["qweight"]).float().mean()) for n in names])
    out["b4_n_samples"] = np.array([e["n_samples"] for e in LoggingHook.log], dtype=np.int64)

    path = os.path.join(HERE, "f10_tiny_llama.npz")
    np.savez_compressed(path, **out)
    print(f"f10_tiny_llama.npz  {os.path.getsize(path) / 1024:.0f} KiB")
    print("order:", names)
    print("sets:", [[names[j] for j in s] for s in sets])
    print({k: float(out[k]) for k in ("ppl_fp_b1", "ppl_q_b1", "ppl_fp_b4", "ppl_q_b4")})
    print("levels that differ between batch_size 1 and 4:", out["b4_level_mismatch"].round(4).tolist())


if __name__ == "__main__":
    main()
