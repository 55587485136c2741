"""Shared helpers for the parity tests."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

FIELDS = ("velocity", "density", "pressure", "divergence")


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def bind_ref_kernels(oracle):
    """argtypes for the kernel-level ref_* functions of oracle/fluid_ref.h."""
    d = oracle.dll
    F, I, P = C.c_float, C.c_int, C.POINTER(C.c_float)
    d.ref_advection.argtypes = [I, F, P, P]
    d.ref_advection_density.argtypes = [I, F, P, P, P]
    d.ref_diffusion.argtypes = [I, F, F, P, P]
    d.ref_vorticity.argtypes = [I, P, P]
    d.ref_add_vorticity.argtypes = [I, F, P, P]
    d.ref_divergence.argtypes = [I, P, P]
    d.ref_pressure_jacobi.argtypes = [I, P, P, P]
    d.ref_projection.argtypes = [I, P, P, P]
    d.ref_add_force.argtypes = [I, F, F, F, F, F, F, P, P, P]
    for n in ("ref_advection", "ref_advection_density", "ref_diffusion", "ref_vorticity",
              "ref_add_vorticity", "ref_divergence", "ref_pressure_jacobi", "ref_projection",
              "ref_add_force"):
        getattr(d, n).restype = None
    return d


def random_state(n, seed, vmax=60.0):
    """A rough random state: velocities large enough that backtraces span several cells and
    leave the domain near the borders (exercises the clamps)."""
    rng = np.random.default_rng(seed)
    vel = rng.uniform(-vmax, vmax, size=(n, n, 2)).astype(np.float32)
    dens = rng.uniform(0.0, 1.0, size=(n, n, 4)).astype(np.float32)
    prs = rng.uniform(-1.0, 1.0, size=(n, n)).astype(np.float32)
    return vel, dens, prs


def smooth_state(n, seed):
    """A smooth state (sum of a few sinusoids), closer to what splats produce."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:n, 0:n].astype(np.float32) / np.float32(n)
    vel = np.zeros((n, n, 2), np.float32)
    dens = np.zeros((n, n, 4), np.float32)
    for c in range(2):
        for _ in range(3):
            kx, ky = rng.integers(1, 6, 2)
            ph = rng.uniform(0, 6.28)
            vel[..., c] += (rng.uniform(5, 25) * np.sin(6.2831853 * (kx * x + ky * y) + ph)).astype(np.float32)
    for c in range(4):
        kx, ky = rng.integers(1, 6, 2)
        dens[..., c] = (0.5 + 0.5 * np.sin(6.2831853 * (kx * x - ky * y))).astype(np.float32)
    prs = np.zeros((n, n), np.float32)
    return vel, dens, prs


def load_state(sim, pkg, vel, dens, prs):
    sim.write(pkg.FIELD_VELOCITY, vel)
    sim.write(pkg.FIELD_DENSITY, dens)
    sim.write(pkg.FIELD_PRESSURE, prs)


def read_all(sim, pkg):
    return {
        "velocity": sim.read(pkg.FIELD_VELOCITY),
        "density": sim.read(pkg.FIELD_DENSITY),
        "pressure": sim.read(pkg.FIELD_PRESSURE),
        "divergence": sim.read(pkg.FIELD_DIVERGENCE),
    }


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    """Bit-for-bit equality of two fp32 arrays (signed zeros and NaN payloads included)."""
    ba, bb = bits(a), bits(b)
    if not np.array_equal(ba, bb):
        bad = np.argwhere(ba != bb)
        i = tuple(bad[0])
        raise AssertionError(
            f"{what}: {len(bad)} of {ba.size} elements differ; first at {i}: "
            f"{a[i]!r} ({ba[i]:#010x}) vs {b[i]!r} ({bb[i]:#010x})")


def rel_err(a, b):
    """Field-normalised error max|a-b| / max(max|b|, 1e-30)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), 1e-30))
