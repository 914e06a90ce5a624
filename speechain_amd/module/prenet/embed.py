"""``EmbedPrenet``, drop-in for ``speechain.module.prenet.embed.EmbedPrenet`` (speechain/module/prenet/embed.py:13-66): a token embedding
table ``embed.weight`` (vocab, dim), optionally scaled by sqrt(dim).  The table gradient comes from the deterministic HIP kernel behind
``functional.embedding`` (one wave per position, no atomics)."""
import math

import torch

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module


class EmbedPrenet(Module):
    def module_init(self, embedding_dim, vocab_size, scale: bool = False, padding_idx: int = 0):
        self.output_size = embedding_dim
        self.embedding_dim, self.vocab_size, self.scale, self.padding_idx = embedding_dim, vocab_size, scale, padding_idx
        self.embed = torch.nn.Embedding(num_embeddings=vocab_size, embedding_dim=embedding_dim, padding_idx=padding_idx)

    def forward(self, text: torch.Tensor):
        e = Fn.embedding(self.embed.weight, text, self.padding_idx)  # (device tensors only, like every module here: no CPU path)
        return e * math.sqrt(self.embedding_dim) if self.scale else e
