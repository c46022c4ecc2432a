"""Where the host waits in quantize_llama: wall time of the calls that copy results to the CPU, against the
block time (experiment helper).  Usage: driver_host_time.py [n_blocks]"""
import os, sys, json, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import LlamaConfig, LlamaForCausalLM
import gptq_v2_amd as amd
from gptq_v2_amd import driver

n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda:0")
cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=n_blocks, num_attention_heads=32,
                  num_key_value_heads=32, vocab_size=32000, max_position_embeddings=2048)
torch.manual_seed(0)
with torch.device(dev):
    model = LlamaForCausalLM(cfg).half()
model.eval()
enc = torch.randint(0, 32000, (128, 2048))
spent = {"to_cpu": 0.0, "to_cpu_calls": 0, "depth": 0, "bytes": 0}
orig = driver._to_device


def timed(obj, device):
    if torch.device(device).type != "cpu" or spent["depth"]:
        if isinstance(obj, torch.Tensor) and obj.is_cuda and torch.device(device).type == "cpu":
            spent["bytes"] += obj.numel() * obj.element_size()
        return orig(obj, device)
    spent["depth"] += 1
    torch.cuda.synchronize()
    t0 = time.time()
    out = orig(obj, device)
    spent["to_cpu"] += time.time() - t0
    spent["to_cpu_calls"] += 1
    spent["depth"] -= 1
    return out


driver._to_device = timed
torch.cuda.synchronize(); t0 = time.time()
res = driver.quantize_llama(model, enc, dev, batch_size=16, bits=4, group_size=128)
torch.cuda.synchronize()
print(json.dumps({"total_s": round(time.time() - t0, 3), "seconds_per_block": [round(v, 3) for v in res["seconds_per_block"]],
                  "to_cpu_s": round(spent["to_cpu"], 3), "to_cpu_calls": spent["to_cpu_calls"], "MB": round(spent["bytes"] / 1e6)}))
