"""Helper of test_gpu_dist.py (not collected): quantize_llama under torch.distributed.run with two
ranks that share GPU 0 (gloo backend, collectives through host memory)."""
import copy
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import transformers
    rank, ws = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    from gptq_v2_amd import driver
    cfg = transformers.LlamaConfig(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2,
                                   num_attention_heads=4, num_key_value_heads=2, max_position_embeddings=256,
                                   tie_word_embeddings=False)
    torch.manual_seed(0)
    model = transformers.LlamaForCausalLM(cfg).half().eval()
    enc = torch.randint(0, 512, (16, 96), generator=torch.Generator().manual_seed(1))
    qmodel = copy.deepcopy(model)
    res = driver.quantize_llama(qmodel, enc, dev, batch_size=4)
    names = sorted(res["data"])
    assert len(names) == 14
    # every rank ends with the same results: compare a digest of every tensor across ranks
    digest = torch.stack([res["data"][n]["qweight"].long().sum() * 31 + res["data"][n]["qzero"].long().sum() for n in names])
    both = [torch.empty_like(digest) for _ in range(ws)]
    dist.all_gather(both, digest)
    assert all(torch.equal(both[0], b) for b in both)
    w0 = qmodel.model.layers[1].mlp.down_proj.weight.data.float().cpu()
    ws_ = [torch.empty_like(w0) for _ in range(ws)]
    dist.all_gather(ws_, w0)
    assert all(torch.equal(ws_[0], w) for w in ws_)
    # against the single-process run: H differs only in the order of its fp32 additions (two partial sums
    # instead of one), so almost every level is the same and none differs by more than one step in block 0
    single = copy.deepcopy(model)
    # a run without the process group: tear it down first so that the driver sees one rank
    dist.barrier()
    dist.destroy_process_group()
    ref = driver.quantize_llama(single, enc, dev, batch_size=4)
    for n in names:
        if ".layers.0." not in n:
            continue   # block 1 sees block 0's (slightly different) outputs
        a, b = res["data"][n]["qweight"].int(), ref["data"][n]["qweight"].int()
        assert a.shape == b.shape
        assert float((a != b).float().mean()) < 0.02, n
    for k in ("gptq_error", "l2.0_error"):
        a, b = res["metrics"][names[0]][k], ref["metrics"][names[0]][k]
        assert abs(a - b) <= 0.05 * abs(b), (k, a, b)
    print("driver rank ok", flush=True)


if __name__ == "__main__":
    main()
