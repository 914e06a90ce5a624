"""Vocabulary projection head, drop-in for ``speechain.module.postnet.token.TokenPostnet`` (speechain/module/postnet/token.py:12-49):
parameter names ``linear.{weight,bias}``; used as the decoder head and as the CTC head.

The (tokens, d) x (d, V) product runs on the bf16 / exact-f32 GEMM kernels through ``functional.linear`` (bias in the epilogue); with
V = 5000 the forward product goes to the 256 x 256 LDS-DMA tiles and the weight gradient to the split-K kernel, which also returns
the bias gradient from the same launch."""
import torch

from speechain_amd import functional as Fn
from speechain_amd.module.abs import Module


class TokenPostnet(Module):
    def module_init(self, vocab_size: int, input_dim: int = None):
        width = self.input_size if self.input_size is not None else input_dim
        if width is None:
            raise ValueError("TokenPostnet needs the width of its input: pass input_size (as the model builders do) or input_dim")
        self.input_dim, self.vocab_size, self.output_size = width, vocab_size, vocab_size
        self.linear = torch.nn.Linear(in_features=width, out_features=vocab_size)  # holder of the fp32 master weights

    def forward(self, input: torch.Tensor):
        """(batch, length, input_dim) -> (batch, length, vocab_size) logits in the compute dtype."""
        return Fn.linear(Fn.to_compute(input), [self.linear.weight], [self.linear.bias])

    def extra_repr(self) -> str:
        return f"input_dim={self.input_dim}, vocab_size={self.vocab_size}"
