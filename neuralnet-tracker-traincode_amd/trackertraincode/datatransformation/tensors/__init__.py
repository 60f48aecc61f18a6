from .affinetrafo import (apply_affine2d, position_normalization, position_unnormalization, transform_coord,  # noqa: F401
                          transform_keypoints, transform_points, transform_roi, transform_rot)
from .normalization import unwhiten_image, whiten_image  # noqa: F401
