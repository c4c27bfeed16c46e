"""InputLayer (reference: SparseConvNet/sparseconvnet/ioLayers.py:15-65,163-195).

Same constructor, same (coords, features[, batch_size]) input tuple, same modes.  The
reference forces `coords` to the host (`input[0].cpu().long()`, :60) because its hash grid is
a host structure; here the grid is built on the device, so device-resident coordinates are
used as they are and host coordinates are uploaded once."""
import torch
from torch.autograd import Function
from torch.nn import Module

from . import SCN
from .utils import toLongTensor
from .sparseConvNetTensor import SparseConvNetTensor
from .metadata import Metadata


class InputLayer(Module):
    def __init__(self, dimension, spatial_size, mode=3):
        Module.__init__(self)
        self.dimension = dimension
        self.spatial_size = toLongTensor(dimension, spatial_size)
        self.mode = mode
        self.device = None
        # extension (SCN.Metadata_3): "first_seen" = the reference's site numbering (IOLayersRules.h:86-91), "brick" =
        # brick-major rows over brick grids (same sites and features up to a per-sample permutation of the rows)
        self.site_order = "first_seen"

    def to(self, device):
        self.device = device
        return self

    def prepare(self, coords, device=None, stream=None):
        """Optional extension (not in the reference): build the hash grid / site numbering of a
        coordinate list AHEAD of the forward call, e.g. for the next scene on a side stream while
        the current scene trains.  The geometry depends on the coordinates only, so this is the
        device-side analogue of a data-loader prefetch; `forward` picks the prepared Metadata up
        when it is called with the same coordinate tensor."""
        if device is None:
            device = self.device if self.device is not None else coords.device
        md = Metadata(self.dimension, self.site_order)
        side = stream if stream is not None else torch.cuda.current_stream()
        c64 = coords if coords.dtype == torch.int64 else coords.long()   # converted ONCE; forward reuses c64
        with torch.cuda.stream(side):
            # kernels + an asynchronous read-back of the site count; nothing blocks here
            md.inputLayerEnqueue(self.spatial_size, c64, self.mode if self.mode else 3, device)
            md.prepared_on = side
        if not hasattr(self, "_prepared"):
            self._prepared = []
        # matched by identity + version of the caller's tensor (held here, so its address cannot be recycled)
        self._prepared.append((coords, coords._version, c64, md))

    def forward(self, input):
        prepared = getattr(self, "_prepared", None)
        md, c64 = None, None
        if prepared:
            for i, (src, ver, conv, m) in enumerate(prepared):
                if src is input[0] and ver == input[0]._version:
                    md, c64 = m, conv
                    del prepared[i]
                    break
        if md is not None:
            cur = torch.cuda.current_stream()
            if md.prepared_on != cur:
                cur.wait_stream(md.prepared_on)  # tensors were produced on the side stream
                md.hand_over(cur)                # ... and their memory stays theirs until `cur` is done with it
            SCN._reap_handed_over()
        output = SparseConvNetTensor(metadata=md if md is not None else Metadata(self.dimension, self.site_order),
                                     spatial_size=self.spatial_size)
        output.features = InputLayerFunction.apply(
            self.dimension, output.metadata, self.spatial_size, c64 if c64 is not None else input[0].long(),
            input[1].to(self.device) if self.device else input[1], 0 if len(input) == 2 else input[2],
            self.mode)
        return output


class OutputLayer(Module):
    """Used with an InputLayer for 'autoencoder' style networks: SparseConvNetTensor -> float tensor [N, planes],
    N defined by the InputLayer (reference: ioLayers.py:66-87)."""

    def __init__(self, dimension):
        Module.__init__(self)
        self.dimension = dimension

    def forward(self, input):
        return OutputLayerFunction.apply(self.dimension, input.metadata, input.features)


class OutputLayerFunction(Function):
    @staticmethod
    def forward(ctx, dimension, metadata, input_features):
        output_features = input_features.new()
        ctx.metadata_ = metadata
        ctx.dimension = dimension
        SCN.OutputLayer_updateOutput(metadata, input_features.contiguous(), output_features)
        return output_features

    @staticmethod
    def backward(ctx, grad_output):
        grad_input = grad_output.new()
        SCN.OutputLayer_updateGradInput(ctx.metadata_, grad_input, grad_output.contiguous())
        return None, None, grad_input


class InputLayerFunction(Function):
    @staticmethod
    def forward(ctx, dimension, metadata, spatial_size, coords, input_features, batch_size, mode):
        output_features = input_features.new()
        ctx.dimension = dimension
        ctx.metadata_ = metadata
        SCN.InputLayer_updateOutput(metadata, spatial_size, coords, input_features.contiguous(), output_features,
                                    batch_size, mode)
        return output_features

    @staticmethod
    def backward(ctx, grad_output):
        grad_input = grad_output.new()
        SCN.InputLayer_updateGradInput(ctx.metadata_, grad_input, grad_output.contiguous())
        return None, None, None, None, grad_input, None, None
