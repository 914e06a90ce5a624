"""Vocabulary projection head, drop-in for ``speechain.module.postnet.token.TokenPostnet`` (speechain/module/postnet/token.py:12-49):
parameter names ``linear.{weight,bias}``; used as the decoder head and as the CTC head."""
import torch

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module


class TokenPostnet(Module):
    def module_init(self, vocab_size: int, input_dim: int = None):
        if self.input_size is not None:
            input_dim = self.input_size
        else:
            assert input_dim is not None
        self.output_size = vocab_size
        self.input_dim, self.vocab_size = input_dim, vocab_size
        self.linear = torch.nn.Linear(in_features=input_dim, out_features=vocab_size)

    def forward(self, input: torch.Tensor):
        return Fn.linear(Fn.to_compute(input), [self.linear.weight], [self.linear.bias])
