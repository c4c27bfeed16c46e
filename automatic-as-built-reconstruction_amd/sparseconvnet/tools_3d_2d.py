"""sparse_3d_to_dense_2d (reference: SparseConvNet/sparseconvnet/tools_3d_2d.py:7-48): densify a sparse
map and crop it to the occupied extent."""
import sparseconvnet as scn


def sparse_3d_to_dense_2d(feat_s3d):
    locations_3d0 = feat_s3d.get_spatial_locations()  # [x, y, z, batch_idx]
    max_map_size = locations_3d0.max(0)[0] + 1
    x_size, y_size, z_size, batch_size = max_map_size
    nPlane0 = feat_s3d.features.shape[1]
    to_dense_layer = scn.sparseToDense.SparseToDense(dimension=4, nPlanes=nPlane0)
    features_3d_flat = to_dense_layer(feat_s3d)  # [batch_size, channels, X, Y, Z]
    return features_3d_flat[:, :, 0:x_size, 0:y_size, 0:z_size]
