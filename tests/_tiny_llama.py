"""The two-block random-initialised Llama behind tests/golden/f10_tiny_llama.npz: built from a
``LlamaConfig`` and a seed (no checkpoint, no fetch).  Shared by the fixture generator and the tests;
the fixture carries a digest of the weights so that a different initialisation is noticed."""
import hashlib

import torch

CONFIG = dict(vocab_size=512, hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
              num_key_value_heads=2, max_position_embeddings=256, tie_word_embeddings=False)
MODEL_SEED, TOKEN_SEED, N_SEQ, N_TOK = 0, 1, 14, 96


def tiny_llama():
    import transformers
    torch.manual_seed(MODEL_SEED)
    return transformers.LlamaForCausalLM(transformers.LlamaConfig(**CONFIG)).half().eval()


def tokens() -> torch.Tensor:
    return torch.randint(0, CONFIG["vocab_size"], (N_SEQ, N_TOK), generator=torch.Generator().manual_seed(TOKEN_SEED))


def state_digest(model) -> str:
    h = hashlib.sha256()
    for name, p in sorted(model.state_dict().items()):
        h.update(name.encode())
        h.update(p.detach().cpu().contiguous().view(torch.uint8).numpy().tobytes())
    return h.hexdigest()
