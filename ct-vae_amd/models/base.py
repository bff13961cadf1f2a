"""``BaseVAE`` — the drop-in boundary (reference: models/base.py:5-28).

Same abstract API (encode / decode / sample / generate / forward / loss_function); concrete models
additionally mix in ``FlatParamMixin`` so parameters live in one packed HBM buffer.
"""
from abc import abstractmethod
from typing import Any, List

from torch import nn

from .packing import FlatParamMixin
from .types_ import Tensor


class BaseVAE(FlatParamMixin, nn.Module):

    def __init__(self) -> None:
        super().__init__()

    def encode(self, input: Tensor) -> List[Tensor]:
        raise NotImplementedError

    def decode(self, input: Tensor) -> Any:
        raise NotImplementedError

    def sample(self, batch_size: int, current_device: int, **kwargs) -> Tensor:
        raise NotImplementedError

    def generate(self, x: Tensor, **kwargs) -> Tensor:
        raise NotImplementedError

    @abstractmethod
    def forward(self, *inputs: Tensor) -> Tensor:
        pass

    @abstractmethod
    def loss_function(self, *inputs: Any, **kwargs) -> Tensor:
        pass
