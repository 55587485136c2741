"""An independent restatement of the three GATHER stages, and the oracle held to it bit for bit.

tests/test_rect.py restates the four stencil stages (diffusion, divergence, Jacobi, projection) in
numpy.  The stages where the reference's texture semantics live — `advection`
(/root/reference/FluidSimulation/kernel.cu:114-140), `advectionDensity` (:142-165) and
`addForceCUDA` (:357-383) — are restated here, written from those lines and from SURVEY.md
Appendix B (the legacy texture unit: unnormalised coordinates, clamp-to-edge, LINEAR fetch at
(x, y) = bilinear around (x - 0.5, y - 0.5)), NOT from oracle/fluid_ref.c: this file models a
texture unit (`tex_linear`, `tex_point`) and then writes each kernel the way the CUDA text reads,
`tex2D(...)` call for `tex2D(...)` call, where the oracle has per-stage clamped index helpers.
If the two agree on rough fields whose backtraces hit every clamp, a shared misreading between
oracle and HIP kernels (which are compared with each other, not with this file) is much less
plausible.  It does not pin parity with the CUDA binary — nothing here can.

CPU only; the HIP path is compared with the oracle elsewhere (tests/test_parity_gpu.py).
"""
import ctypes as C

import numpy as np
import pytest

from test_rect import SHAPES, bind_rect, rect_state
from util import assert_bit_equal, bind_ref_kernels, fptr, random_state

F32 = np.float32
DT = F32(0.1)          # __constant__ float dt, kernel.cu:31


# ---- the texture unit ----------------------------------------------------
def texel(T, i, j):
    """T[i, j] with clamp-to-edge addressing (B.1 / B.2); T is [y][x][channels...]."""
    ny, nx = T.shape[:2]
    return T[np.clip(j, 0, ny - 1), np.clip(i, 0, nx - 1)]


def mix(x, y, a):
    """kernel.cu:60-71, the author's lerp: x + (y - x) * a, one rounding per operation."""
    a = a.reshape(a.shape + (1,) * (x.ndim - a.ndim))
    return x + (y - x) * a


def bilinear(T, i, j, a, b):
    """B.2: mix in x on both rows, then in y (kernel.cu:88-91)."""
    return mix(mix(texel(T, i, j), texel(T, i + 1, j), a), mix(texel(T, i, j + 1), texel(T, i + 1, j + 1), a), b)


def tex_linear(T, x, y):
    """tex2D under cudaFilterModeLinear at unnormalised (x, y): B.2 with xB = x - 0.5."""
    xb, yb = x - F32(0.5), y - F32(0.5)
    i, j = np.floor(xb), np.floor(yb)
    return bilinear(T, i.astype(np.int64), j.astype(np.int64), xb - i, yb - j)


def tex_linear_half(T, px, py):
    """tex2D(tex, px + 0.5f, py + 0.5f) under LINEAR: the fetch of the two advection gathers.
    B.2's note: base and weights are taken from p itself (i = floor(p.x), a = p.x - i), not from
    (p + 0.5) - 0.5, whose fp32 round trip would perturb the weight at large coordinates."""
    i, j = np.floor(px), np.floor(py)
    return bilinear(T, i.astype(np.int64), j.astype(np.int64), px - i, py - j)


def tex_point(T, x, y):
    """tex2D under cudaFilterModePoint: texel [floor x, floor y], clamped (B.1)."""
    return texel(T, np.floor(x).astype(np.int64), np.floor(y).astype(np.int64))


def ids(nx, ny):
    y, x = np.mgrid[0:ny, 0:nx]
    return x, y


def backtrace(x, y, vel, nx, ny):
    """kernel.cu:126 + :130 (and :156 + :158): p = id - dt * v, clamped to [0, gridResolution]."""
    px = x.astype(F32) - DT * vel[..., 0]
    py = y.astype(F32) - DT * vel[..., 1]
    # helper_math clamp = fmaxf(lo, fminf(v, hi)), per component (Appendix C)
    return np.maximum(F32(0), np.minimum(px, F32(nx))), np.maximum(F32(0), np.minimum(py, F32(ny)))


# ---- the three kernels, as the CUDA text reads --------------------------------------------------
def np_advection(V):
    """kernel.cu:114-140; texture_float2 bound LINEAR (:546)."""
    ny, nx = V.shape[:2]
    x, y = ids(nx, ny)
    xf, yf = x.astype(F32), y.astype(F32)
    velocity = tex_linear(V, xf + F32(0.5), yf + F32(0.5))                       # :125 (a texel centre: weights 0)
    px, py = backtrace(x, y, velocity, nx, ny)                                   # :126, :130
    out = tex_linear_half(V, px, py)                                             # :131
    interior = (x > 0) & (x < nx - 1) & (y > 0) & (y < ny - 1)                   # :122-123
    # else-branch: four independent ifs in this order, a later write replaces an earlier one
    for cond, cx, cy in ((x == 0, xf + 1, yf),                                   # :135
                         (x == nx - 1, xf - 1, yf),                              # :136
                         (y == 0, xf + 1, yf + 1),                               # :137  (x + 1: as written)
                         (y == ny - 1, xf + 1, yf - 1)):                         # :138
        m = cond & ~interior
        out[m] = (-tex_linear(V, cx, cy))[m]                                     # LINEAR at integer coordinates (B.3)
    return out


def np_advection_density(V, D):
    """kernel.cu:142-165; texture_float2 POINT (:669), texture_float4 LINEAR (:672)."""
    ny, nx = V.shape[:2]
    x, y = ids(nx, ny)
    velocity = tex_point(V, x.astype(F32) + F32(0.5), y.astype(F32) + F32(0.5))  # :154
    px, py = backtrace(x, y, velocity, nx, ny)                                   # :156, :158
    out = tex_linear_half(D, px, py)                                             # :159
    interior = (x > 0) & (x < nx - 1) & (y > 0) & (y < ny - 1)
    out[~interior] = 0                                                           # :163
    return out


def np_add_force(V, D, sx, sy, force, colour, exp=np.exp):
    """kernel.cu:357-383: every cell, no guard; returns (vel, dens, c)."""
    ny, nx = V.shape[:2]
    x, y = ids(nx, ny)
    dx = x.astype(F32) / F32(nx) - F32(sx)                                       # :365
    dy = y.astype(F32) / F32(ny) - F32(sy)                                       # :366 (gridResolution per axis)
    radius = F32(0.001)                                                          # :368
    c = exp(-(dx * dx + dy * dy) / radius).astype(F32) * DT                      # :370
    vel = V + c[..., None] * np.asarray(force, F32)                              # :378
    dens = D + c[..., None] * np.asarray(colour, F32)                            # :382
    return vel, dens, c


# ---- the oracle held to it ----------------------------------------------------------------------
def rough(nx, ny, seed, vmax):
    """Velocities up to vmax (backtraces of vmax/10 cells: out of the domain on every side near
    the borders, so clamp-to-0, clamp-to-N and the texel clamp at N-1 all occur) with a tenth of
    the cells set to exact multiples of 10 (backtraces that land exactly on texel centres and on
    the clamp bounds) and a few exact zeros."""
    vel, dens, _ = rect_state(nx, ny, seed, vmax=vmax)
    rng = np.random.default_rng(seed + 1)
    m = rng.random((ny, nx)) < 0.1
    vel[m] = (rng.integers(-6, 7, size=(int(m.sum()), 2)) * 10).astype(F32)
    vel[rng.random((ny, nx)) < 0.02] = 0
    return vel, dens


def clamps_hit(vel):
    ny, nx = vel.shape[:2]
    x, y = ids(nx, ny)
    px = x.astype(F32) - DT * vel[..., 0]
    py = y.astype(F32) - DT * vel[..., 1]
    inner = (x > 0) & (x < nx - 1) & (y > 0) & (y < ny - 1)
    return {"x<0": bool((px[inner] < 0).any()), "x>N": bool((px[inner] > nx).any()),
            "y<0": bool((py[inner] < 0).any()), "y>N": bool((py[inner] > ny).any())}


@pytest.mark.parametrize("n", [4, 5, 33, 64])
@pytest.mark.parametrize("vmax", [8.0, 60.0, 900.0])
def test_square_oracle_gathers_equal_the_texture_unit_restatement(oracle, n, vmax):
    d = bind_ref_kernels(oracle)
    vel, dens = rough(n, n, seed=n * 7 + int(vmax), vmax=vmax)
    if vmax >= 60.0 and n >= 33:      # enough interior cells for every case to occur
        assert all(clamps_hit(vel).values()), clamps_hit(vel)
    out = np.empty_like(vel)
    d.ref_advection(n, 0.1, fptr(vel), fptr(out))
    assert_bit_equal(out, np_advection(vel), f"advection n={n} vmax={vmax}")
    dout = np.empty_like(dens)
    d.ref_advection_density(n, 0.1, fptr(vel), fptr(dens), fptr(dout))
    assert_bit_equal(dout, np_advection_density(vel, dens), f"advectionDensity n={n} vmax={vmax}")


@pytest.mark.parametrize("nx,ny", SHAPES)
def test_rect_oracle_gathers_equal_the_texture_unit_restatement(oracle, nx, ny):
    r = bind_rect(oracle)
    for vmax in (12.0, 70.0):
        vel, dens = rough(nx, ny, seed=nx * 31 + ny, vmax=vmax)
        out = np.empty_like(vel)
        r.ref_advection_rect(nx, ny, 0.1, fptr(vel), fptr(out))
        assert_bit_equal(out, np_advection(vel), f"advection {nx}x{ny} vmax={vmax}")
        dout = np.empty_like(dens)
        r.ref_advection_density_rect(nx, ny, 0.1, fptr(vel), fptr(dens), fptr(dout))
        assert_bit_equal(dout, np_advection_density(vel, dens), f"advectionDensity {nx}x{ny} vmax={vmax}")


def test_advection_border_rows_and_corners_follow_the_written_order(oracle):
    """Distinct values everywhere: the y rules (written last, with x + 1) win at the corners, and
    the LINEAR fetch at integer coordinates is the 2x2 mean of B.3 in mix() form."""
    d = bind_ref_kernels(oracle)
    n = 8
    vel = (np.arange(n * n * 2, dtype=F32).reshape(n, n, 2) * F32(0.37) + F32(1.0))
    out = np.empty_like(vel)
    d.ref_advection(n, 0.1, fptr(vel), fptr(out))
    want = np_advection(vel)
    assert_bit_equal(out, want, "advection 8x8")

    def mean4(i, j):      # B.3: texels [i-1..i] x [j-1..j], weights 1/2, x first
        c = lambda a: min(max(a, 0), n - 1)
        t = lambda a, b: vel[c(b), c(a)]
        h = F32(0.5)
        r0 = t(i - 1, j - 1) + (t(i, j - 1) - t(i - 1, j - 1)) * h
        r1 = t(i - 1, j) + (t(i, j) - t(i - 1, j)) * h
        return r0 + (r1 - r0) * h
    assert_bit_equal(out[3, 0], -mean4(1, 3), "x == 0")                # :135  tex2D(x+1, y)
    assert_bit_equal(out[3, n - 1], -mean4(n - 2, 3), "x == N-1")      # :136  tex2D(x-1, y)
    assert_bit_equal(out[0, 3], -mean4(4, 1), "y == 0")                # :137  tex2D(x+1, y+1)
    assert_bit_equal(out[n - 1, 3], -mean4(4, n - 2), "y == N-1")      # :138  tex2D(x+1, y-1)
    assert_bit_equal(out[0, 0], -mean4(1, 1), "corner (0,0): the y rule wins")
    assert_bit_equal(out[n - 1, n - 1], -mean4(n, n - 2), "corner (N-1,N-1): x+1 clamps to N-1")


def test_fetch_at_p_plus_half_round_trip_is_not_what_the_oracle_does():
    """Documents the B.2 decision: weights come from p, not from (p + 0.5) - 0.5.  At small
    coordinates the two agree to an ulp of the coordinate; they are not bit-identical in general."""
    rng = np.random.default_rng(0)
    T = rng.uniform(-1, 1, (64, 64, 2)).astype(F32)
    px = rng.uniform(0, 63, (64, 64)).astype(F32)
    py = rng.uniform(0, 63, (64, 64)).astype(F32)
    direct = tex_linear_half(T, px, py)
    generic = tex_linear(T, px + F32(0.5), py + F32(0.5))
    assert np.max(np.abs(direct - generic)) <= 2.0 * 2.0 ** -18      # |dT| <= 2, coordinate ulp <= 2^-18 below 64
    on_centres = tex_linear(T, np.floor(px) + F32(0.5), np.floor(py) + F32(0.5))
    assert_bit_equal(on_centres, T[np.floor(py).astype(int), np.floor(px).astype(int)], "a texel centre is the texel")


@pytest.mark.parametrize("nx,ny", [(64, 64), (33, 33), (96, 40), (5, 4)])
def test_oracle_add_force_equals_the_restatement(oracle, nx, ny):
    r = bind_rect(oracle)
    vel, dens = rough(nx, ny, seed=5, vmax=3.0)
    sx, sy, force, colour = 0.375, 0.625, (17.0, -42.0), (1.0, 0.5, 0.25, 0.0)
    v, dn = vel.copy(), dens.copy()
    r.ref_add_force_rect(nx, ny, 0.1, 0.001, sx, sy, force[0], force[1], (C.c_float * 4)(*colour), fptr(v), fptr(dn))
    # glibc expf and numpy's float32 exp may differ in the last place; everything around exp is
    # exact, so with exp evaluated in double and rounded once (what a correctly rounded expf
    # returns) the splat must agree to 1 ulp of c, and bit for bit wherever the two exps agree
    wv, wd, c = np_add_force(vel, dens, sx, sy, force, colour, exp=lambda t: np.exp(t.astype(np.float64)))
    tol = 2.0 ** -23 * float(c.max()) * 1.01
    assert np.max(np.abs(v - wv)) <= tol * max(abs(f) for f in force) + 1e-30
    assert np.max(np.abs(dn - wd)) <= tol + 1e-30
    assert float(np.mean(v.view(np.uint32) == wv.view(np.uint32))) > 0.99
    # far from the splat c underflows to 0 and the fields are untouched, border included (no guard)
    far = c == 0
    if far.any():
        assert_bit_equal(v[far], vel[far], "untouched where c == 0")


def test_square_entry_points_are_the_rect_ones(oracle):
    """ref_X(n) == ref_X_rect(n, n) for the gather stages on the rough state used above."""
    d, r = bind_ref_kernels(oracle), bind_rect(oracle)
    n = 33
    vel, dens = rough(n, n, seed=9, vmax=60.0)
    a, b = np.empty_like(vel), np.empty_like(vel)
    d.ref_advection(n, 0.1, fptr(vel), fptr(a))
    r.ref_advection_rect(n, n, 0.1, fptr(vel), fptr(b))
    assert_bit_equal(a, b, "advection")
    da, db = np.empty_like(dens), np.empty_like(dens)
    d.ref_advection_density(n, 0.1, fptr(vel), fptr(dens), fptr(da))
    r.ref_advection_density_rect(n, n, 0.1, fptr(vel), fptr(dens), fptr(db))
    assert_bit_equal(da, db, "density advection")
