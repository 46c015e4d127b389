"""Orbit camera helper with the public surface of ``inference/camera.py`` (``Camera``,
``Orientation``): spherical (distance, pitch, yaw) around a look-at point, six axis presets.

Behaviour restated from ``SuperresolutionNetwork/inference/camera.py:4-97``:
* each orientation carries an up vector, a signed axis permutation (1-based) applied to the
  y-up spherical position, and whether yaw is mirrored (``:5-10``);
* ``fromAngles``: ``(cos p cos y, sin p, cos p sin y) * d`` (``:61-66``);
* pitch is clamped to +-80 degrees while dragging (``:91``); zoom is ``base * 1.1**k`` (``:95-97``).
"""
import math
from enum import Enum

_PITCH_LIMIT = math.radians(80)


class Orientation(Enum):
    #      id  up-vector    axis permutation  mirrored yaw
    Xp = (1, (1, 0, 0), (2, -1, -3), True)
    Xm = (2, (-1, 0, 0), (-2, 1, 3), False)
    Yp = (3, (0, 1, 0), (1, 2, 3), False)
    Ym = (4, (0, -1, 0), (-1, -2, -3), True)
    Zp = (5, (0, 0, 1), (-3, -1, 2), False)
    Zm = (6, (0, 0, -1), (3, 1, -2), True)

    def __new__(cls, ident, up, permute, inv_yaw):
        obj = object.__new__(cls)
        obj._value_ = ident
        obj._up = list(up)
        obj._permute = list(permute)
        obj._inv_yaw = inv_yaw
        return obj

    def __str__(self):
        return str(self.value)

    @property
    def up(self):
        return self._up

    @property
    def permute(self):
        return self._permute

    @property
    def invYaw(self):
        return self._inv_yaw


class Camera:
    def __init__(self, resX, resY, origin=(0, 1, -1.7)):
        self.resX = resX
        self.resY = resY
        self.lookAt = [0, 0, 0]
        self.speed = 0.01
        self.zoomspeed = 1.1
        self.orientation = Orientation.Yp
        self.currentDistance, self.currentPitch, self.currentYaw = Camera.toAngles(origin)
        self.baseDistance = self.currentDistance
        self.zoomvalue = 0

    @staticmethod
    def toAngles(pos):
        x, y, z = pos[0], pos[1], pos[2]
        dist = math.sqrt(x * x + y * y + z * z)
        return dist, math.asin(y / dist), math.atan2(z, x)

    @staticmethod
    def fromAngles(length, pitch, yaw):
        cp = math.cos(pitch)
        return [cp * math.cos(yaw) * length, math.sin(pitch) * length, cp * math.sin(yaw) * length]

    def getLookAt(self):
        return self.lookAt

    def getOrigin(self):
        yaw = -self.currentYaw if self.orientation.invYaw else self.currentYaw
        base = Camera.fromAngles(self.currentDistance, self.currentPitch, yaw)
        return [math.copysign(1, p) * base[abs(p) - 1] for p in self.orientation.permute]

    def getUp(self):
        return self.orientation.up

    def startMove(self):
        self.oldDistance = self.currentDistance
        self.oldPitch = self.currentPitch
        self.oldYaw = self.currentYaw

    def stopMove(self):
        pass

    def move(self, deltax, deltay):
        pitch = self.oldPitch + self.speed * deltay
        self.currentPitch = min(_PITCH_LIMIT, max(-_PITCH_LIMIT, pitch))
        self.currentYaw = self.oldYaw + self.speed * deltax

    def zoom(self, delta):
        self.zoomvalue += delta
        self.currentDistance = self.baseDistance * (self.zoomspeed ** self.zoomvalue)
