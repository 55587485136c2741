#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (not part of pytest: minutes, not seconds).
  A. fused path vs the CPU oracle, small random grids, random D / J, bit for bit
  B. fused vs straight (FLUID_FLAG_UNFUSED) path, random grids up to 3000^2 incl. odd sizes
  C. slab groups (random count, split tail, pressure groups) vs the whole domain
  D. random launch plans (FLUID_TB_* / FLUID_TILE), whole domain and slabs, fused vs straight path
  E. inner slabs (fluid_params.inner_slabs) vs the one-stream context, random plans
Usage: python tools/fuzz_parity.py [seed] [cases_a cases_b cases_c cases_d [cases_e]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fluid_amd_loader  # noqa: E402
from util import random_state, smooth_state, load_state  # noqa: E402

pkg = fluid_amd_loader.load()
import fluid_simulation_amd.slab as slab  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
na, nb, nc = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (120, 60, 60)
nd = int(sys.argv[5]) if len(sys.argv) > 5 else 60
rng = np.random.default_rng(seed)
oracle = pkg.FluidLib(os.path.join(ROOT, "oracle", "liboracle_fluid.so"))
F = (("velocity", pkg.FIELD_VELOCITY), ("density", pkg.FIELD_DENSITY), ("pressure", pkg.FIELD_PRESSURE),
     ("divergence", pkg.FIELD_DIVERGENCE))
bad = 0


def same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def compare(tag, x, y):
    global bad
    for name, fid in F:
        if not same(x.read(fid), y.read(fid)):
            bad += 1
            print("MISMATCH", tag, name, flush=True)
            return False
    return True


for i in range(na):
    n = int(rng.integers(4, 420))
    D, J = int(rng.integers(0, 13)), int(rng.integers(0, 46))
    st = (random_state(n, seed=i, vmax=float(rng.uniform(1, 60))) if rng.random() < 0.5 else smooth_state(n, seed=i))
    g = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J)
    r = pkg.FluidSim(n, lib=oracle, diffusion_iters=D, pressure_iters=J)
    load_state(g, pkg, *st)
    load_state(r, pkg, *st)
    steps = int(rng.integers(1, 4))
    g.step(steps)
    r.step(steps)
    compare(f"A n={n} D={D} J={J} steps={steps}", g, r)
    g.close(); r.close()
print(f"A done ({na} cases), mismatches so far {bad}", flush=True)

for i in range(na // 2):   # A'': rectangular grids (fluid_params.ny), random tile / pipeline choice
    nx, ny = int(rng.integers(4, 700)), int(rng.integers(4, 700))
    if rng.random() < 0.6:
        nx = max(nx // 4 * 4, 4)
    D, J = int(rng.integers(0, 13)), int(rng.integers(0, 46))
    r_ = np.random.default_rng(900 + i)
    vmax = float(rng.uniform(1, 60))
    st = (r_.uniform(-vmax, vmax, (ny, nx, 2)).astype(np.float32), r_.uniform(0, 1, (ny, nx, 4)).astype(np.float32),
          r_.uniform(-1, 1, (ny, nx)).astype(np.float32))
    os.environ["FLUID_TILE"] = str(int(rng.integers(0, 2)))
    os.environ["FLUID_TILE_TY"] = str(int(rng.choice([16, 32])))
    g = pkg.FluidSim(nx, ny=ny, diffusion_iters=D, pressure_iters=J)
    os.environ.pop("FLUID_TILE"); os.environ.pop("FLUID_TILE_TY")
    r = pkg.FluidSim(nx, ny=ny, lib=oracle, diffusion_iters=D, pressure_iters=J)
    load_state(g, pkg, *st)
    load_state(r, pkg, *st)
    steps = int(rng.integers(1, 3))
    g.step(steps)
    r.step(steps)
    compare(f"A'' {nx}x{ny} D={D} J={J} steps={steps}", g, r)
    g.close(); r.close()
print(f"A'' done ({na // 2} cases), mismatches so far {bad}", flush=True)

for i in range(na // 4):   # A': vorticity confinement on (whole-domain contexts only)
    n = int(rng.integers(4, 300))
    D, J = int(rng.integers(0, 13)), int(rng.integers(0, 30))
    st = random_state(n, seed=500 + i, vmax=float(rng.uniform(1, 40)))
    g = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=pkg.FLAG_VORTICITY)
    r = pkg.FluidSim(n, lib=oracle, diffusion_iters=D, pressure_iters=J, flags=pkg.FLAG_VORTICITY)
    load_state(g, pkg, *st)
    load_state(r, pkg, *st)
    g.step(2)
    r.step(2)
    compare(f"A' vorticity n={n} D={D} J={J}", g, r)
    g.close(); r.close()
print(f"A' done ({na // 4} cases), mismatches so far {bad}", flush=True)

for i in range(nb):
    n = int(rng.integers(300, 3000))
    if rng.random() < 0.6:
        n = n // 4 * 4
    D, J = int(rng.integers(0, 13)), int(rng.integers(0, 46))
    st = smooth_state(n, seed=1000 + i)
    a = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J)
    b = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=pkg.FLAG_UNFUSED)
    load_state(a, pkg, *st)
    load_state(b, pkg, *st)
    a.step(2)
    b.step(2)
    compare(f"B n={n} D={D} J={J}", a, b)
    a.close(); b.close()
print(f"B done ({nb} cases), mismatches so far {bad}", flush=True)

for i in range(nc):
    count = int(rng.integers(2, 7))
    D, J, A = int(rng.integers(0, 11)), int(rng.integers(0, 41)), int(rng.integers(1, 9))
    G = int(rng.integers(0, 4))
    need = (J + max(G, 1) - 1) // max(G, 1) + 1 + D + A + 2
    n = int(rng.integers(max(count * (need + 4), 64), 2200 if rng.random() < 0.7 else 4400))
    if rng.random() < 0.7:
        n = n // 2 * 2   # odd widths take the straight kernels
    split = bool(rng.random() < 0.6)
    fl = pkg.FLAG_UNFUSED if rng.random() < 0.2 else 0
    vt = pkg.FLAG_VORTICITY if rng.random() < 0.3 else 0   # the confinement stage, two halo rows deeper
    try:
        grp = slab.LocalSlabGroup(pkg, n, count, split_tail=split and not fl, early_advect=bool(rng.random() < 0.5),
                                  diffusion_iters=D, pressure_iters=J, halo_rows=A, pressure_groups=G, flags=fl | vt)
    except pkg.FluidError as e:
        print("skip", n, count, D, J, A, G, e)
        continue
    whole = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=vt)
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    for s in range(3):
        whole.add_source(*ss_a.splat())
        grp.add_source(*ss_b.splat())
        whole.step(1)
        grp.step(1)
    for name, fid in F[:3]:
        if not same(grp.read(fid), whole.read(fid)):
            bad += 1
            print("MISMATCH C", n, count, D, J, A, G, split, fl, vt, name, flush=True)
            break
    grp.close(); whole.close()
print(f"C done ({nc} cases), mismatches {bad}", flush=True)

PLAN_ENV = ("FLUID_TB_KJ", "FLUID_TB_KD", "FLUID_TB_CJ", "FLUID_TB_CD", "FLUID_TB_WAVES", "FLUID_TB_MINCHUNK", "FLUID_TILE",
            "FLUID_TILE_TY", "FLUID_INTERIOR_SPLIT", "FLUID_INTERIOR_PIECE_MIN", "FLUID_TB_CHAIN", "FLUID_TB_CHAIN_KINDS")
for i in range(nd):
    for k in PLAN_ENV:
        os.environ.pop(k, None)
    env = {"FLUID_TB_KJ": int(rng.choice([1, 2, 4, 5, 8, 10, 13, 14])), "FLUID_TB_KD": int(rng.choice([1, 2, 4, 5, 8, 10])),
           "FLUID_TB_CJ": int(rng.choice([2, 4])), "FLUID_TB_CD": int(rng.choice([2, 4])),
           "FLUID_TB_WAVES": int(rng.choice([1, 7, 64, 500, 2048, 3900, 20000])),
           "FLUID_TB_MINCHUNK": int(rng.integers(1, 5)), "FLUID_TILE": int(rng.integers(0, 2)),
           "FLUID_TILE_TY": int(rng.choice([16, 32])), "FLUID_INTERIOR_SPLIT": int(rng.integers(1, 9)), "FLUID_INTERIOR_PIECE_MIN": int(rng.choice([1, 5000, 1 << 21])),
           "FLUID_TB_CHAIN": int(rng.integers(0, 2)), "FLUID_TB_CHAIN_KINDS": int(rng.integers(1, 8))}   # chained chunks
    D, J = int(rng.integers(0, 13)), int(rng.integers(0, 46))
    count = int(rng.integers(1, 5))
    A = int(rng.integers(1, 9))
    n = int(rng.integers(max(count * (J + 1 + D + A + 4), 16), 1800)) // 2 * 2
    if rng.random() < 0.5:
        n = n // 4 * 4
    ref = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=pkg.FLAG_UNFUSED)   # plan-independent
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        if count == 1:
            tst = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J)
        else:
            tst = slab.LocalSlabGroup(pkg, n, count, split_tail=bool(rng.random() < 0.5),
                                      early_advect=bool(rng.random() < 0.5), diffusion_iters=D,
                                      pressure_iters=J, halo_rows=A)
    except pkg.FluidError as e:
        print("skip", n, count, D, J, A, env, e)
        ref.close()
        continue
    finally:
        for k in PLAN_ENV:
            os.environ.pop(k, None)
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    for s_ in range(2):
        ref.add_source(*ss_a.splat())
        tst.add_source(*ss_b.splat())
        ref.step(1)
        tst.step(1)
    for name, fid in F[:3]:
        if not same(tst.read(fid), ref.read(fid)):
            bad += 1
            print("MISMATCH D", n, count, D, J, A, env, name, flush=True)
            break
    tst.close(); ref.close()
print(f"D done ({nd} cases), mismatches {bad}", flush=True)

# E: inner slabs (fluid_params.inner_slabs: G slab contexts side by side on the GPU, halo rows pushed
# between their streams) against the one-stream context, random plans, several fluid_step calls of
# random length with splats and a field write between them
ne = int(sys.argv[6]) if len(sys.argv) > 6 else nd
for i in range(ne):
    for k in PLAN_ENV:
        os.environ.pop(k, None)
    G = int(rng.integers(2, 6))
    D, J, A = int(rng.integers(0, 13)), int(rng.integers(0, 46)), int(rng.integers(1, 9))
    need = J + 1 + D + A + 2
    vt = pkg.FLAG_VORTICITY if rng.random() < 0.3 else 0
    n = int(rng.integers(max(G * (need + 4), 32), 1500 if rng.random() < 0.8 else 3000)) // 2 * 2
    ny = int(rng.integers(max(G * (need + 4), 32), 1500)) if rng.random() < 0.3 else 0
    env = {"FLUID_TB_KJ": int(rng.choice([2, 5, 8, 10, 14])), "FLUID_TB_KD": int(rng.choice([2, 5, 10])),
           "FLUID_TB_WAVES": int(rng.choice([7, 64, 500, 2048])), "FLUID_TILE": int(rng.integers(0, 2))} if rng.random() < 0.6 else {}
    ref = pkg.FluidSim(n, ny=ny, diffusion_iters=D, pressure_iters=J, flags=vt)
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        tst = pkg.FluidSim(n, ny=ny, diffusion_iters=D, pressure_iters=J, halo_rows=A, inner_slabs=G, flags=vt)
    except pkg.FluidError as e:
        print("skip", n, ny, G, D, J, A, env, e)
        ref.close()
        continue
    finally:
        for k in PLAN_ENV:
            os.environ.pop(k, None)
    ss_a, ss_b = pkg.SplatStream(), pkg.SplatStream()
    ok = True
    for s_ in range(3):
        if rng.random() < 0.7:
            ref.add_source(*ss_a.splat())
            tst.add_source(*ss_b.splat())
        if rng.random() < 0.2:
            d = ref.read(pkg.FIELD_DENSITY)[:, ::-1].copy()
            ref.write(pkg.FIELD_DENSITY, d)
            tst.write(pkg.FIELD_DENSITY, d)
        k = int(rng.integers(1, 4))
        ref.step(k)
        tst.step(k)
        for name, fid in F:
            if not same(tst.read(fid), ref.read(fid)):
                bad += 1
                ok = False
                print("MISMATCH E", n, ny, G, D, J, A, env, vt, name, "call", s_, flush=True)
                break
        if not ok:
            break
    tst.close(); ref.close()
print(f"E done ({ne} cases), mismatches {bad}", flush=True)
sys.exit(1 if bad else 0)
