"""SparseConvNetTensor: a feature matrix [V, planes] + the Metadata that owns the grids it lives on + the spatial size of
its grid.  Attributes and methods of the reference's class (SparseConvNet/sparseconvnet/sparseConvNetTensor.py:12-55);
`get_spatial_locations_device` is an addition."""


class SparseConvNetTensor(object):
    def __init__(self, features=None, metadata=None, spatial_size=None):
        self.features, self.metadata, self.spatial_size = features, metadata, spatial_size

    # ---- where the rows are -------------------------------------------------------------------------------------
    def _size(self, spatial_size):
        return self.spatial_size if spatial_size is None else spatial_size

    def get_spatial_locations(self, spatial_size=None):
        """(x, y, z, batch index) of every active site, LongTensor [V, 4] on the host, in the order of the feature rows"""
        return self.metadata.getSpatialLocations(self._size(spatial_size))

    def get_spatial_locations_device(self, spatial_size=None):
        """the same tensor left in device memory (the RPN glue consumes it without a host round trip)"""
        return self.metadata.getSpatialLocationsDevice(self._size(spatial_size))

    # ---- the feature matrix moved or converted in place; the grid stays ---------------------------------------------
    def _replace(self, features):
        self.features = features
        return self

    def to(self, device):
        return self._replace(self.features.to(device))

    def cuda(self):
        return self._replace(self.features.cuda())

    def cpu(self):
        return self._replace(self.features.cpu())

    def type(self, t=None):
        return self._replace(self.features.type(t)) if t else self.features.type()

    @property
    def requires_grad(self):
        return self.features.requires_grad

    def __repr__(self):
        sites = self.get_spatial_locations() if self.metadata else None
        parts = (("features", self.features), ("features.shape", self.features.shape), ("batch_locations", sites),
                 ("batch_locations.shape", None if sites is None else sites.shape), ("spatial size", self.spatial_size))
        return "SparseConvNetTensor<<" + ",".join("%s=%r" % kv for kv in parts) + ">>"
