"""Which torch operators of the block walk cost device time: torch.profiler over a 1-block quantize_llama at Llama-2-7B
block dimensions, grouped by operator and by the Python line that called it (experiment helper)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transformers import LlamaConfig, LlamaForCausalLM
from gptq_v2_amd import driver
from torch.profiler import profile, ProfilerActivity

n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
cfg = LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32,
                  num_key_value_heads=32, vocab_size=32000, max_position_embeddings=2048)
torch.manual_seed(0)
with torch.device(dev):
    model = LlamaForCausalLM(cfg).half()
model.eval()
enc = torch.randint(0, 32000, (n_seq, 2048))
driver.quantize_llama(model, enc, dev, batch_size=16)   # warm-up
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    driver.quantize_llama(model, enc, dev, batch_size=16)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=30, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=6).table(sort_by="self_cuda_time_total", row_limit=25, max_name_column_width=50, max_src_column_width=110))
