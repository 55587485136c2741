"""The reference's three view modes as a host-side post-process of fluid_read.

visualizationStep() (kernel.cu:710-756) maps a field to an RGBA float image on the device and
draws it with glDrawPixels; keys 'd' / 'v' / 'p' choose the field (kernel.cu:823-831).  Headless,
the same mappings are applied to the raw fields returned by ``FluidSim.read``:

    density   passthrough                                   kernel.cu:389-400
    velocity  ((1 + u) / 2, (1 + v) / 2, 0, 0)              kernel.cu:402-416
    pressure  (1 + p) / 2 in all four channels               kernel.cu:418-431
"""
from __future__ import annotations

import numpy as np

F32 = np.float32


def density_image(dens: np.ndarray) -> np.ndarray:
    return np.ascontiguousarray(dens, dtype=F32)


def velocity_image(vel: np.ndarray) -> np.ndarray:
    out = np.zeros(vel.shape[:2] + (4,), F32)
    out[..., :2] = (F32(1.0) + vel.astype(F32)) / F32(2.0)
    return out


def pressure_image(prs: np.ndarray) -> np.ndarray:
    v = (F32(1.0) + prs.astype(F32)) / F32(2.0)
    return np.repeat(v[..., None], 4, axis=-1)


def image_of(sim, pkg, method: int) -> np.ndarray:
    """visualizationMethod 0 / 1 / 2 (kernel.cu:473, :712-751) of a simulation context."""
    if method == 0:
        return density_image(sim.read(pkg.FIELD_DENSITY))
    if method == 1:
        return velocity_image(sim.read(pkg.FIELD_VELOCITY))
    if method == 2:
        return pressure_image(sim.read(pkg.FIELD_PRESSURE))
    raise ValueError("visualizationMethod must be 0 (density), 1 (velocity) or 2 (pressure)")


def write_ppm(path: str, rgba: np.ndarray, flip_y: bool = True):
    """8-bit binary PPM of the RGB channels, clamped to [0, 1] as GL_FLOAT pixels are.
    glDrawPixels puts row 0 at the bottom of the window, so rows are flipped by default."""
    img = np.clip(rgba[..., :3], 0.0, 1.0)
    if flip_y:
        img = img[::-1]
    data = (img * 255.0 + 0.5).astype(np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (data.shape[1], data.shape[0]))
        f.write(np.ascontiguousarray(data).tobytes())


def write_pfm(path: str, rgba: np.ndarray):
    """Lossless float image (PFM, little endian, bottom-to-top rows like glDrawPixels)."""
    img = np.ascontiguousarray(rgba[..., :3], dtype="<f4")
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (img.shape[1], img.shape[0]))
        f.write(img.tobytes())
