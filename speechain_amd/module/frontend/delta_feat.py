"""Delta / delta-delta features (speechain/module/frontend/delta_feat.py:12-89) -- used by no recipe; kept as a thin
device-side torch op so ``Speech2MelSpec(delta_order=...)`` keeps working."""
import torch
import torch.nn.functional as F

from speechain_amd.module.abs import Module


class DeltaFeature(Module):
    def module_init(self, delta_order: int = 1, delta_N: int = 2):
        assert delta_order in (1, 2), "delta_order must be 1 or 2"
        self.delta_order, self.delta_N = delta_order, delta_N
        kernel = torch.arange(-delta_N, delta_N + 1, dtype=torch.float32) / (2 * sum(i * i for i in range(1, delta_N + 1)))
        self.register_buffer("kernel", kernel.reshape(1, 1, -1, 1), persistent=False)

    def forward(self, feat: torch.Tensor, feat_len: torch.Tensor):
        outs, cur = [feat], feat
        for _ in range(self.delta_order):
            x = F.pad(cur[:, None], (0, 0, self.delta_N, self.delta_N), mode="replicate")
            cur = F.conv2d(x, self.kernel.to(feat.dtype))[:, 0]
            outs.append(cur)
        return torch.cat(outs, dim=-1), feat_len

    def __repr__(self):
        return f"{self.__class__.__name__}(delta_order={self.delta_order}, delta_N={self.delta_N})"
