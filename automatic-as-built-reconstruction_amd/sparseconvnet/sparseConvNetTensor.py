"""SparseConvNetTensor -- same attributes and methods as the reference's
(SparseConvNet/sparseconvnet/sparseConvNetTensor.py:12-55)."""


class SparseConvNetTensor(object):
    def __init__(self, features=None, metadata=None, spatial_size=None):
        self.features = features
        self.metadata = metadata
        self.spatial_size = spatial_size

    def get_spatial_locations(self, spatial_size=None):
        "Coordinates and batch index for the active spatial locations (LongTensor [V,4], CPU)"
        if spatial_size is None:
            spatial_size = self.spatial_size
        return self.metadata.getSpatialLocations(spatial_size)

    def get_spatial_locations_device(self, spatial_size=None):
        "same, left in device memory (the RPN glue can consume it without a host round trip)"
        if spatial_size is None:
            spatial_size = self.spatial_size
        return self.metadata.getSpatialLocationsDevice(spatial_size)

    def to(self, device):
        self.features = self.features.to(device)
        return self

    def type(self, t=None):
        if t:
            self.features = self.features.type(t)
            return self
        return self.features.type()

    def cuda(self):
        self.features = self.features.cuda()
        return self

    def cpu(self):
        self.features = self.features.cpu()
        return self

    @property
    def requires_grad(self):
        return self.features.requires_grad

    def __repr__(self):
        sl = self.get_spatial_locations() if self.metadata else None
        return ("SparseConvNetTensor<<features=" + repr(self.features) + ",features.shape=" +
                repr(self.features.shape) + ",batch_locations=" + repr(sl) + ",batch_locations.shape=" +
                repr(sl.shape if self.metadata else None) + ",spatial size=" + repr(self.spatial_size) + ">>")
