"""JoinTable / AddTable / ConcatTable -- the reference's table modules (SparseConvNet/sparseconvnet/tables.py:13-55: same
class names, `add`, `input_spatial_size`, and the same state_dict keys: children are registered "0", "1", ...)."""
import torch

from .sparseConvNetTensor import SparseConvNetTensor
from .utils import _sum_features


class _Table(torch.nn.Sequential):
    """what the three have in common: they leave the grid alone, so any output size is also the input size, and their
    result lives on the grid of the first operand"""

    def input_spatial_size(self, out_size):
        return out_size

    @staticmethod
    def _on_grid_of(first, features):
        return SparseConvNetTensor(features, first.metadata, first.spatial_size)


class JoinTable(_Table):
    """planes of all operands side by side"""

    def forward(self, input):
        first = input[0]
        if first.features.numel() == 0:            # an empty grid: nothing to concatenate
            return self._on_grid_of(first, first.features)
        return self._on_grid_of(first, torch.cat([t.features for t in input], 1))


class AddTable(_Table):
    """sum of the operands (operands with fewer planes add into the leading planes: utils.add_feature_planes)"""

    def forward(self, input):
        return self._on_grid_of(input[0], _sum_features(input))


class ConcatTable(_Table):
    """every branch applied to the same input; the list of results"""

    def add(self, module):
        self.add_module(str(len(self)), module)
        return self

    def forward(self, input):
        return [branch(input) for branch in self]

    def input_spatial_size(self, out_size):
        return self[0].input_spatial_size(out_size)
