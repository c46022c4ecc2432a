"""Block-wise driver (quantize_llama) on a random-init Llama with Llama-2-7B block dimensions and a
few blocks: seconds per block with the real model flow -- attention, norms, forward pre-hooks,
re-forwarding the block once per input-sharing set -- next to bench.py's synthetic replay of the
same Linears (experiment helper).  Usage: bench_driver.py [n_blocks] [n_sequences] [dense|packed|both]"""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import LlamaConfig, LlamaForCausalLM
import gptq_v2_amd as amd
from gptq_v2_amd import driver

n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_seq = int(sys.argv[2]) if len(sys.argv) > 2 else 128
which = sys.argv[3] if len(sys.argv) > 3 else "both"
dev = torch.device("cuda:0")
cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=n_blocks, num_attention_heads=32,
                  num_key_value_heads=32, vocab_size=32000, max_position_embeddings=2048)
torch.manual_seed(0)
with torch.device(dev):
    model = LlamaForCausalLM(cfg).half()
model.eval()
enc = torch.randint(0, 32000, (n_seq, 2048))
out = {}
for pack in [p_ for p_ in (False, True) if which == "both" or (which == "packed") == p_]:
    if pack and out:   # fresh weights for the second run
        torch.manual_seed(0)
        with torch.device(dev):
            model = LlamaForCausalLM(cfg).half()
        model.eval()
    torch.cuda.synchronize(); t0 = time.time()
    res = driver.quantize_llama(model, enc, dev, batch_size=16, bits=4, group_size=128, pack=pack)
    torch.cuda.synchronize()
    out["packed replay" if pack else "dense replay"] = {
        "total_s": round(time.time() - t0, 3), "seconds_per_block": [round(v, 3) for v in res["seconds_per_block"]]}
print(json.dumps(out))
