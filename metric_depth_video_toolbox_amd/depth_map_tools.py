"""Host-side mirror of the camera helpers of the reference's depth_map_tools.py that the
stereo-rerender loop calls (same names, argument meaning and error behaviour).

Scalar host maths only (a 3x3 matrix per frame); the per-pixel work lives in the HIP kernels.
"""
from __future__ import annotations

import numpy as np


def compute_camera_matrix(fov_horizontal_deg, fov_vertical_deg, image_width, image_height):
    """K = [[fx,0,W/2],[0,fy,H/2],[0,0,1]] (f64) from one or both fields of view in degrees; a
    missing axis copies the other (square pixels).  Reference: depth_map_tools.py:902-934."""
    if fov_horizontal_deg is None and fov_vertical_deg is None:
        # the reference dies with UnboundLocalError here; sr:319-320 raises ValueError before that
        raise ValueError("Either fov_horizontal_deg or fov_vertical_deg must be provided.")
    fx = fy = None
    if fov_horizontal_deg is not None:
        fx = image_width / (2 * np.tan(np.deg2rad(fov_horizontal_deg) / 2))
    if fov_vertical_deg is not None:
        fy = image_height / (2 * np.tan(np.deg2rad(fov_vertical_deg) / 2))
    if fy is None:
        fy = fx
    if fx is None:
        fx = fy
    return np.array([[fx, 0, image_width / 2], [0, fy, image_height / 2], [0, 0, 1]], dtype=np.float64)


def fov_from_camera_matrix(mat):
    """(fov_x, fov_y) in degrees.  Reference: depth_map_tools.py:1640-1649."""
    w, h = mat[0][2] * 2, mat[1][2] * 2
    return (np.rad2deg(2 * np.arctan2(w, 2 * mat[0][0])), np.rad2deg(2 * np.arctan2(h, 2 * mat[1][1])))
