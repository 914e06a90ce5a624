"""Plugin base class with the reference's contract (speechain/module/abs.py:13-173).

``Cls(input_size=..., distributed=..., **conf)`` stores the three shared members and hands ``conf`` to
``module_init``; subclasses must set ``self.output_size``.
"""
from abc import ABC, abstractmethod
from typing import Dict, Optional

import torch


class Module(torch.nn.Module, ABC):
    def __init__(self, input_size: int = None, distributed: bool = False, **module_conf):
        super().__init__()
        self.input_size = input_size
        self.output_size = None
        self.distributed = distributed
        self.module_init(**module_conf)

    @abstractmethod
    def module_init(self, **module_conf):
        raise NotImplementedError

    @abstractmethod
    def forward(self, **kwargs):
        raise NotImplementedError

    def recover(self, **kwargs):
        raise NotImplementedError

    def reset_parameters(self):
        raise NotImplementedError

    def get_recordable_para(self) -> Optional[Dict]:
        """Leaf modules record nothing; containers collect their Module children (abs.py:140-173)."""
        children = {n: m for n, m in self._modules.items() if isinstance(m, Module)}
        if not children:
            return None
        return {n: m.get_recordable_para() for n, m in children.items()}
