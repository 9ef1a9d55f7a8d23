"""SE3 / Intrinsics value types with the attribute surface of the reference's
libs/geometry/camera_modules.py:14-133 (boundary types: `.pose .inv_pose .R .t`, `.mat .inv_mat .fx .fy
.cx .cy`).  Under the overlay the reference's own classes are used by libs/dfvo.py; these are for
standalone use (tests, bench, the frame-batch driver)."""
import numpy as np


class SE3:
    def __init__(self, np_arr=None):
        self._pose = np.eye(4) if np_arr is None else np_arr

    pose = property(lambda self: self._pose, lambda self, v: setattr(self, "_pose", v))

    @property
    def inv_pose(self):
        return np.linalg.inv(self._pose)

    @inv_pose.setter
    def inv_pose(self, v):
        self._pose = np.linalg.inv(v)

    @property
    def R(self):
        return self._pose[:3, :3]

    @R.setter
    def R(self, v):
        self._pose[:3, :3] = v

    @property
    def t(self):
        return self._pose[:3, 3:]

    @t.setter
    def t(self, v):
        self._pose[:3, 3:] = v


class Intrinsics:
    def __init__(self, param=None):
        if param is None:
            self._mat = np.zeros((3, 3))
        else:
            cx, cy, fx, fy = param
            self._mat = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=np.float64)

    mat = property(lambda self: self._mat, lambda self, v: setattr(self, "_mat", v))

    @property
    def inv_mat(self):
        return np.linalg.inv(self._mat)

    @inv_mat.setter
    def inv_mat(self, v):
        self._mat = np.linalg.inv(v)

    @staticmethod
    def _entry(r, c):  # a focal length / principal point component: reads and writes one entry of the matrix in place
        return property(lambda self: self._mat[r, c], lambda self, v: self._mat.__setitem__((r, c), v))


Intrinsics.fx, Intrinsics.fy = Intrinsics._entry(0, 0), Intrinsics._entry(1, 1)
Intrinsics.cx, Intrinsics.cy = Intrinsics._entry(0, 2), Intrinsics._entry(1, 2)
