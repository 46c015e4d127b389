"""Screen-space Phong shading of the G-buffer the network produces.

Public surface of ``SuperresolutionNetwork/utils/shading.py`` (``ScreenSpaceShading``, builder
style setters, ``forward``, static ``normalize``).  Input ``[B, C>=5, H, W]``: mask in [-1,1],
normal xyz, depth, optional ambient occlusion.  Formula (``shading.py:148-191``):

    ao'   = s*clamp(ao or 1-ao, 0, 1) + (1-s)                     (C >= 6, else 1)
    col   = amb*mat + diff*mat*|l.n| + spec*((e+2)/2pi)*clamp(r.eye, 0, 1)^e     r = 2(l.n)n - l
    col  *= ao';  col = bg + clamp(mask/2+1/2, 0, 1)*(col - bg);  clamp(col, 0, 1)

with eye == (0,0,1) for every pixel (``:141``: the perspective eye rays are commented out).
"""
import math

import numpy as np
import torch
import torch.nn as nn


class ScreenSpaceShading(nn.Module):
    def __init__(self, device):
        super().__init__()
        self._device = device
        self.enable_specular = True
        self.inverse_ao = False
        self._host_vectors = {}
        self._background = self._vec(np.array([0.0, 0.0, 0.0]))
        self._eyedirs = dict()

    def _vec(self, a):
        assert isinstance(a, np.ndarray) and a.shape == (3,)
        t = torch.from_numpy(np.asarray(a, dtype=np.float64)).to(dtype=torch.float32)
        dev = t.to(device=self._device).view(1, 3, 1, 1)
        # host copy of the float32 values for packed_parameters(): reading them back from the device tensor
        # would synchronise the stream once per vector and frame
        self._host_vectors[id(dev)] = (dev, [float(v) for v in t.tolist()])
        return dev

    def fov(self, fov):
        assert isinstance(fov, (float, int))
        assert 0 < fov < 90, "fov has to be in (0,90)"
        self._fov = float(fov)
        self._eyedirs = dict()
        return self

    def get_fov(self):
        return self._fov

    def ambient_light_color(self, color):
        self._ambient_light_color = self._vec(color)
        return self

    def diffuse_light_color(self, color):
        self._diffuse_light_color = self._vec(color)
        return self

    def specular_light_color(self, color):
        self._specular_light_color = self._vec(color)
        return self

    def specular_exponent(self, exponent):
        assert isinstance(exponent, (int, float))
        if isinstance(exponent, float):
            assert exponent.is_integer()
            exponent = int(exponent)
        assert exponent > 0
        self._specular_exponent = exponent

    def light_direction(self, dir):
        assert isinstance(dir, np.ndarray) and dir.shape == (3,)
        self._light_direction = self._vec(dir / np.linalg.norm(dir))
        return self

    def material_color(self, color):
        self._material_color = self._vec(color)
        return self

    def ambient_occlusion(self, ao):
        self._ao = float(ao)
        return self

    def background(self, color):
        self._background = self._vec(color)
        return self

    def packed_parameters(self):
        """18 floats for the fused HIP finish kernel: ambient, diffuse, specular, light, material, background."""
        vals = []
        for t in (self._ambient_light_color, self._diffuse_light_color, self._specular_light_color,
                  self._light_direction, self._material_color, self._background):
            hit = self._host_vectors.get(id(t))
            vals += hit[1] if hit is not None and hit[0] is t else [float(v) for v in t.reshape(-1).tolist()]
        if len(self._host_vectors) > 64:      # setters called many times: drop copies of replaced vectors
            live = {id(t) for t in (self._ambient_light_color, self._diffuse_light_color, self._specular_light_color,
                                    self._light_direction, self._material_color, self._background)}
            self._host_vectors = {k: v for k, v in self._host_vectors.items() if k in live}
        return vals

    def _get_eyedir(self, h, w):
        key = (h, w)
        if key not in self._eyedirs:
            eye = torch.zeros(3, h, w, dtype=torch.float32, device=self._device)
            eye[2] = 1.0
            self._eyedirs[key] = eye
        return self._eyedirs[key]

    def forward(self, input):
        B, C, H, W = input.shape
        assert C >= 5
        mask = input[:, 0:1]
        normal = input[:, 1:4]
        if C >= 6:
            a = input[:, 5:6]
            a = 1.0 - a if self.inverse_ao else a
            ao = self._ao * torch.clamp(a, 0, 1) + (1 - self._ao) * torch.ones_like(a)
        else:
            ao = torch.ones_like(input[:, 4:5])
        color = torch.zeros((B, 3, H, W), dtype=torch.float32, device=input.device)
        color = color + self._ambient_light_color * self._material_color
        ndl = torch.sum(self._light_direction * normal, dim=1, keepdim=True)
        color = color + (self._diffuse_light_color * self._material_color) * torch.abs(ndl)
        if self.enable_specular:
            eyedir = self._get_eyedir(H, W).to(input.device)
            reflect = 2 * ndl * normal - self._light_direction
            spec = ((self._specular_exponent + 2) / (2 * np.pi)) * \
                (torch.clamp(torch.sum(reflect * eyedir, dim=1, keepdim=True), 0, 1) ** self._specular_exponent)
            color = color + spec * self._specular_light_color
        color = color * ao
        color = self._background + torch.clamp(mask * 0.5 + 0.5, 0, 1) * (color - self._background)
        return torch.clamp(color, 0, 1)

    @staticmethod
    def normalize(input, dim):
        """``x / max(||x||, 1e-7)`` along ``dim`` (``shading.py:194-207``)."""
        eps = torch.full((1,), 1e-7, dtype=input.dtype, device=input.device)
        return input / torch.max(torch.norm(input, dim=dim, keepdim=True), eps)
