"""Deterministic "seeded splats" input.

The reference's only input is a mouse-driven splat (kernel.cu:870-878 -> addForce :687-699)
with force = pixel delta and one of four colour presets (keys 1-4, kernel.cu:833-850).  This
module pins a scripted equivalent so oracle and GPU runs see identical inputs.
"""
from __future__ import annotations

MASK64 = (1 << 64) - 1
DEFAULT_SEED = 0x5EEDF1D5

# kernel.cu:833-850  (key '1' white, '2' red, '3' green, '4' blue; alpha follows the key's rule)
COLOR_PRESETS = {
    1: (1.0, 1.0, 1.0, 1.0),
    2: (1.0, 0.0, 0.0, 0.0),
    3: (0.0, 1.0, 0.0, 0.0),
    4: (0.0, 0.0, 1.0, 0.0),
}


class SplatStream:
    """splitmix64 stream; uniform float = (next() >> 40) * 2**-24 (exact in fp32)."""

    def __init__(self, seed: int = DEFAULT_SEED):
        self.state = seed & MASK64

    def next_u64(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & MASK64
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def uniform(self) -> float:
        return (self.next_u64() >> 40) * (2.0 ** -24)

    def splat(self):
        """One splat: (sx, sy, fx, fy, rgba).  sx,sy ~ U[0.25,0.75); fx,fy ~ U[-50,50)."""
        import numpy as np
        f32 = np.float32
        sx = f32(0.25) + f32(0.5) * f32(self.uniform())
        sy = f32(0.25) + f32(0.5) * f32(self.uniform())
        fx = f32(-50.0) + f32(100.0) * f32(self.uniform())
        fy = f32(-50.0) + f32(100.0) * f32(self.uniform())
        preset = 1 + (self.next_u64() % 4)
        return float(sx), float(sy), float(fx), float(fy), COLOR_PRESETS[preset]


def run_seeded(sim, nsteps: int, seed: int = DEFAULT_SEED, every: int = 10):
    """Drive ``sim`` (any object with add_source/step) for nsteps with one splat before step 0
    and before every ``every``-th step."""
    ss = SplatStream(seed)
    s = 0
    while s < nsteps:
        sim.add_source(*ss.splat())
        chunk = min(every, nsteps - s)
        sim.step(chunk)
        s += chunk


def alg_bytes_per_cell(diffusion_iters: int, pressure_iters: int) -> int:
    """Algorithmic HBM bytes per cell per full step, B(D,J) = 88 + 16 D + 12 J."""
    return 88 + 16 * diffusion_iters + 12 * pressure_iters
