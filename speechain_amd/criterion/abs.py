"""Callable criterion base (speechain/criterion/abs.py:10-70)."""
from abc import ABC, abstractmethod


class Criterion(ABC):
    def __init__(self, **criterion_conf):
        super().__init__()
        self.criterion_init(**criterion_conf)

    def criterion_init(self, **criterion_conf):
        pass

    @abstractmethod
    def __call__(self, **kwargs):
        raise NotImplementedError
