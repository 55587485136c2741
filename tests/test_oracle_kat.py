"""Known-answer tests that pin the CPU oracle.

The reference ships no tests, fixtures or golden vectors and cannot be
compiled here, so the oracle's parity is "unpinned" by the reference; these KATs are the
hand-derivable consequences of the reference's update rules, each citing the kernel.cu lines it
follows from.  CPU only.
"""
import ctypes as C

import numpy as np
import pytest

from util import bind_ref_kernels, fptr, assert_bit_equal

F32 = np.float32


@pytest.fixture(scope="module")
def ref(oracle):
    return bind_ref_kernels(oracle)


def test_kat1_zero_state_is_fixed_point(pkg, oracle):
    # kernel.cu:135-138, :319-322, :350-353: borders are negations of zero => numerically zero
    sim = pkg.FluidSim(32, lib=oracle)
    sim.step(3)
    for f in (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE, pkg.FIELD_DIVERGENCE):
        assert np.all(sim.read(f) == 0.0)


def test_kat2_add_force_centre_cell(ref):
    # kernel.cu:365-382: at (0.5,0.5) cell (N/2,N/2) has dx=dy=0 => c = exp(0)*dt = 0.1f exactly
    n = 64
    vel = np.zeros((n, n, 2), F32)
    dens = np.zeros((n, n, 4), F32)
    rgba = (C.c_float * 4)(1.0, 0.5, 0.25, 1.0)
    ref.ref_add_force(n, 0.1, 0.001, 0.5, 0.5, 10.0, -5.0, rgba, fptr(vel), fptr(dens))
    c = F32(0.1)
    assert vel[n // 2, n // 2, 0] == c * F32(10.0)
    assert vel[n // 2, n // 2, 1] == c * F32(-5.0)
    assert np.array_equal(dens[n // 2, n // 2], c * np.array([1.0, 0.5, 0.25, 1.0], F32))
    # a neighbour: c = expf(-(1/64)^2/0.001f)*0.1f, to 2 ulp of a float64 evaluation
    dx = F32(33) / F32(64) - F32(0.5)
    want = np.exp(-np.float64(dx * dx) / np.float64(F32(0.001))) * 0.1 * 10.0
    assert abs(vel[n // 2, n // 2 + 1, 0] - want) <= 4e-7 * abs(want)
    # every cell is touched, border included (no guard, kernel.cu:363-383): corner gets
    # c = expf(-(0.5)/0.001)*dt = 0 (underflow) so test a near-border cell with a wide radius
    vel[:] = 0
    dens[:] = 0
    ref.ref_add_force(n, 0.1, 1.0, 0.5, 0.5, 1.0, 1.0, rgba, fptr(vel), fptr(dens))
    assert np.all(vel != 0) and np.all(dens != 0)


def test_kat3_divergence_of_linear_field(ref):
    # kernel.cu:282, :287: v=(x,0) => 0.5*((x+1)-(x-1)) = 1 interior, 0 on the border
    n = 48
    vel = np.zeros((n, n, 2), F32)
    vel[..., 0] = np.arange(n, dtype=F32)[None, :]
    div = np.full((n, n), 7.0, F32)
    ref.ref_divergence(n, fptr(vel), fptr(div))
    assert np.all(div[1:-1, 1:-1] == 1.0)
    assert np.all(div[0] == 0) and np.all(div[-1] == 0) and np.all(div[:, 0] == 0) and np.all(div[:, -1] == 0)
    # v=(0,y) likewise through the second term
    vel[:] = 0
    vel[..., 1] = np.arange(n, dtype=F32)[:, None]
    ref.ref_divergence(n, fptr(vel), fptr(div))
    assert np.all(div[1:-1, 1:-1] == 1.0)


def test_kat4_one_jacobi_sweep_from_zero(ref):
    # kernel.cu:303-313: interior p = (0+0+0+0 + (-1)*div)*0.25 = -0.25*div; border -0 (:319-322)
    n = 40
    rng = np.random.default_rng(1)
    div = rng.uniform(-1, 1, (n, n)).astype(F32)
    p0 = np.zeros((n, n), F32)
    p1 = np.full((n, n), 9.0, F32)
    ref.ref_pressure_jacobi(n, fptr(p0), fptr(div), fptr(p1))
    assert np.array_equal(p1[1:-1, 1:-1], F32(-0.25) * div[1:-1, 1:-1])
    border = np.ones((n, n), bool)
    border[1:-1, 1:-1] = False
    assert np.all(p1[border] == 0.0)
    assert np.all(np.signbit(p1[border]))  # negated zero (hazard H8)


def test_kat4b_jacobi_border_reads_input_iterate(ref):
    # kernel.cu:319-322: x=0 -> -p[1,y]; x=N-1 -> -p[N-2,y]; y=0 -> -p[x,1]; y=N-1 -> -p[x,N-2];
    # corners take the y rule (last if wins, hazard H4)
    n = 16
    p = np.arange(n * n, dtype=F32).reshape(n, n) + 1
    out = np.zeros((n, n), F32)
    div = np.zeros((n, n), F32)
    ref.ref_pressure_jacobi(n, fptr(p), fptr(div), fptr(out))
    assert np.array_equal(out[1:-1, 0], -p[1:-1, 1])
    assert np.array_equal(out[1:-1, -1], -p[1:-1, -2])
    assert np.array_equal(out[0, :], -p[1, :])        # includes corners: -p[x=0,1], -p[x=N-1,1]
    assert np.array_equal(out[-1, :], -p[-2, :])


def test_kat5_diffusion_uniform_field(ref):
    # kernel.cu:195: (4u + alpha*u)*beta ~= u within 2 ulp; border copied through (:201-203)
    n = 24
    u = F32(3.25)
    vel = np.full((n, n, 2), u, F32)
    vel[0, :, :] = 11.0  # border values must pass through untouched
    out = np.zeros_like(vel)
    ref.ref_diffusion(n, 0.1, 0.01, fptr(vel), fptr(out))
    assert np.all(np.abs(out[2:-2, 2:-2] - u) <= 2 * np.spacing(u))
    assert np.array_equal(out[0], vel[0]) and np.array_equal(out[:, 0], vel[:, 0])
    assert np.array_equal(out[-1], vel[-1]) and np.array_equal(out[:, -1], vel[:, -1])
    # exact value, in the written order, with alpha/beta computed in fp32 (kernel.cu:181-183)
    alpha = F32(1.0) / (F32(0.01) * F32(0.1))
    beta = F32(1.0) / (F32(4.0) + alpha)
    want = ((((u + u) + u) + u) + alpha * u) * beta
    assert out[5, 5, 0] == want


def test_kat6_projection_linear_pressure(ref):
    # kernel.cu:344: p = x => out = vel - ((x+1)-(x-1), 0) = vel - (2, 0)
    n = 32
    rng = np.random.default_rng(2)
    vel = rng.integers(-8, 8, (n, n, 2)).astype(F32)
    p = np.broadcast_to(np.arange(n, dtype=F32)[None, :], (n, n)).copy()
    out = np.zeros_like(vel)
    ref.ref_projection(n, fptr(vel), fptr(p), fptr(out))
    assert np.array_equal(out[1:-1, 1:-1, 0], vel[1:-1, 1:-1, 0] - 2)
    assert np.array_equal(out[1:-1, 1:-1, 1], vel[1:-1, 1:-1, 1])


def test_kat7_projection_border_and_corner_precedence(ref):
    # kernel.cu:350-353: x=0 -> -V[1,y]; x=N-1 -> -V[N-2,y]; y=0 -> -V[x+1,1]; y=N-1 -> -V[x+1,N-2]
    # (x+1 clamped to N-1, hazard H3); corners: y rule overwrites x rule (H4)
    n = 12
    vel = (np.arange(n * n * 2, dtype=F32).reshape(n, n, 2) + 1)
    p = np.zeros((n, n), F32)
    out = np.zeros_like(vel)
    ref.ref_projection(n, fptr(vel), fptr(p), fptr(out))
    assert np.array_equal(out[1:-1, 0], -vel[1:-1, 1])
    assert np.array_equal(out[1:-1, -1], -vel[1:-1, -2])
    xs = np.minimum(np.arange(n) + 1, n - 1)
    assert np.array_equal(out[0, :], -vel[1, xs])
    assert np.array_equal(out[-1, :], -vel[-2, xs])
    assert np.array_equal(out[0, 0], -vel[1, 1])          # corner (0,0) -> -V[1,1]
    assert np.array_equal(out[0, -1], -vel[1, n - 1])     # (N-1,0) -> -V[N-1,1]
    assert np.array_equal(out[-1, 0], -vel[n - 2, 1])     # (0,N-1) -> -V[1,N-2]
    assert np.array_equal(out[-1, -1], -vel[n - 2, n - 1])


def test_kat8_density_advection_identity_and_shift(ref):
    # kernel.cu:154-163: zero velocity => backtrace lands on the texel centre => weights 0 =>
    # interior is the identity; border forced to 0
    n = 20
    rng = np.random.default_rng(3)
    dens = rng.uniform(0, 1, (n, n, 4)).astype(F32)
    vel = np.zeros((n, n, 2), F32)
    out = np.full_like(dens, 5.0)
    ref.ref_advection_density(n, 0.1, fptr(vel), fptr(dens), fptr(out))
    assert_bit_equal(out[1:-1, 1:-1], dens[1:-1, 1:-1], "identity")
    assert np.all(out[0] == 0) and np.all(out[-1] == 0) and np.all(out[:, 0] == 0) and np.all(out[:, -1] == 0)
    # uniform v = (-10, 0): p.x = x - 0.1f*(-10) = x + 1 exactly => shift by one cell
    vel[..., 0] = -10.0
    ref.ref_advection_density(n, 0.1, fptr(vel), fptr(dens), fptr(out))
    px = F32(5) - F32(0.1) * F32(-10.0)
    assert px == F32(6)
    assert_bit_equal(out[1:-1, 1:-1], dens[1:-1, 2:], "shift")


def test_advection_border_is_negated_2x2_mean(ref):
    # kernel.cu:135-138 with LINEAR filtering at integer coordinates
    n = 10
    rng = np.random.default_rng(4)
    v = rng.integers(-16, 16, (n, n, 2)).astype(F32) * 4  # multiples of 4: means are exact
    out = np.zeros_like(v)
    ref.ref_advection(n, 0.0, fptr(v), fptr(out))

    def mean(xs, ys):
        xs = np.clip(xs, 0, n - 1)
        ys = np.clip(ys, 0, n - 1)
        return (v[ys[0], xs[0]] + v[ys[0], xs[1]] + v[ys[1], xs[0]] + v[ys[1], xs[1]]) / 4

    y = 4
    assert np.array_equal(out[y, 0], -mean((0, 1), (y - 1, y)))              # x == 0
    assert np.array_equal(out[y, n - 1], -mean((n - 3, n - 2), (y - 1, y)))  # x == N-1
    x = 5
    assert np.array_equal(out[0, x], -mean((x, x + 1), (0, 1)))              # y == 0
    assert np.array_equal(out[n - 1, x], -mean((x, x + 1), (n - 3, n - 2)))  # y == N-1
    assert np.array_equal(out[0, 0], -mean((0, 1), (0, 1)))                  # corner: y rule wins
    assert np.array_equal(out[n - 1, n - 1], -mean((n - 1, n), (n - 3, n - 2)))
    # dt = 0: interior identity
    assert_bit_equal(out[1:-1, 1:-1], v[1:-1, 1:-1], "interior dt=0")


def test_advection_bilinear_weights(ref):
    # kernel.cu:126-131: uniform v=(5,0), dt=0.1f => p.x = x - 0.5 => mean of left neighbour and self
    n = 16
    vel = np.zeros((n, n, 2), F32)
    vel[..., 0] = 5.0
    vel[..., 1] = np.arange(n, dtype=F32)[None, :] * 2  # marker carried in v.y... but v.y moves p.y
    vel[..., 1] = 0.0
    dens = np.zeros((n, n, 4), F32)
    dens[..., 0] = np.arange(n, dtype=F32)[None, :] * 2
    out = np.zeros_like(dens)
    ref.ref_advection_density(n, 0.1, fptr(vel), fptr(dens), fptr(out))
    px = F32(7) - F32(0.1) * F32(5.0)
    assert px == F32(6.5)
    assert np.all(out[1:-1, 1:-1, 0] == dens[1:-1, 1:-1, 0] - 1)


def test_vorticity_rules(ref):
    # kernel.cu:222: w = (vR.y - vL.y) - (vT.x - vB.x); v=(−y, x) => w = 2 - (−2) ... = 4
    n = 16
    y, x = np.mgrid[0:n, 0:n].astype(F32)
    vel = np.stack([-y, x], -1).astype(F32)
    w = np.zeros((n, n), F32)
    ref.ref_vorticity(n, fptr(vel), fptr(w))
    assert np.all(w[1:-1, 1:-1] == 4.0)
    assert np.all(w[0] == 0) and np.all(w[:, 0] == 0)
    # uniform vorticity => zero gradient => velocity untouched (kernel.cu:252 guard)
    v2 = vel.copy()
    wu = np.full((n, n), 3.0, F32)
    ref.ref_add_vorticity(n, 0.1, fptr(wu), fptr(v2))
    assert_bit_equal(v2, vel, "no force on uniform vorticity")
    # w = x => grad=(2,0) => force = 0.2*(g.y, -g.x) = (0, -0.4); vel += force*dt  (kernel.cu:254-259)
    wx = np.broadcast_to(np.arange(n, dtype=F32)[None, :], (n, n)).copy()
    ref.ref_add_vorticity(n, 0.1, fptr(wx), fptr(v2))
    want_y = vel[1:-1, 1:-1, 1] + (F32(0.2) * F32(-2.0)) * F32(0.1)
    assert np.array_equal(v2[1:-1, 1:-1, 1], want_y)
    assert np.array_equal(v2[1:-1, 1:-1, 0], vel[1:-1, 1:-1, 0] + (F32(0.2) * F32(0.0)) * F32(0.1))
    assert_bit_equal(v2[0], vel[0], "border untouched")


def test_step_is_composition_of_kernels(pkg, oracle, ref):
    # kernel.cu:701-708 and :604-662: the driver's ordering / ping-pong is the composition of the
    # kernel-level functions
    n, D, J = 24, 10, 10
    rng = np.random.default_rng(5)
    vel = rng.uniform(-20, 20, (n, n, 2)).astype(F32)
    dens = rng.uniform(0, 1, (n, n, 4)).astype(F32)
    sim = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, lib=oracle)
    sim.write(pkg.FIELD_VELOCITY, vel)
    sim.write(pkg.FIELD_DENSITY, dens)
    sim.step(1)
    a, b = vel.copy(), np.zeros_like(vel)
    ref.ref_advection(n, 0.1, fptr(a), fptr(b)); a, b = b, a
    for _ in range(D):
        ref.ref_diffusion(n, 0.1, 0.01, fptr(a), fptr(b)); a, b = b, a
    div = np.zeros((n, n), F32)
    ref.ref_divergence(n, fptr(a), fptr(div))
    p, q = np.zeros((n, n), F32), np.zeros((n, n), F32)
    for _ in range(J):
        ref.ref_pressure_jacobi(n, fptr(p), fptr(div), fptr(q)); p, q = q, p
    ref.ref_projection(n, fptr(a), fptr(p), fptr(b)); a, b = b, a
    d2 = np.zeros_like(dens)
    ref.ref_advection_density(n, 0.1, fptr(a), fptr(dens), fptr(d2))
    assert_bit_equal(sim.read(pkg.FIELD_VELOCITY), a, "velocity")
    assert_bit_equal(sim.read(pkg.FIELD_PRESSURE), p, "pressure")
    assert_bit_equal(sim.read(pkg.FIELD_DIVERGENCE), div, "divergence")
    assert_bit_equal(sim.read(pkg.FIELD_DENSITY), d2, "density")


def test_openmp_build_of_the_oracle_gives_the_same_bits(pkg, oracle):
    """oracle/liboracle_fluid_omp.so (the same source with its row loops shared out over threads;
    bench.py's multi-core CPU baseline) against the serial build: whole steps with splats and the
    vorticity stage on, every field bit for bit."""
    import os
    from util import ROOT
    path = os.path.join(ROOT, "oracle", "liboracle_fluid_omp.so")
    if not os.path.exists(path):
        pytest.skip("OpenMP build of the oracle not present")
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    omp = pkg.FluidLib(path)
    for n, flags in ((97, 0), (160, pkg.FLAG_VORTICITY)):
        a = pkg.FluidSim(n, lib=oracle, pressure_iters=12, flags=flags)
        b = pkg.FluidSim(n, lib=omp, pressure_iters=12, flags=flags)
        sa, sb = pkg.SplatStream(), pkg.SplatStream()
        for _ in range(4):
            a.add_source(*sa.splat())
            b.add_source(*sb.splat())
            a.step(3)
            b.step(3)
        for f in (pkg.FIELD_VELOCITY, pkg.FIELD_DENSITY, pkg.FIELD_PRESSURE, pkg.FIELD_DIVERGENCE):
            assert_bit_equal(a.read(f), b.read(f), f"field {f} n={n}")
