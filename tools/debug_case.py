import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import fluid_amd_loader
pkg = fluid_amd_loader.load()
n, D, J = 1080, 3, 2
env = {'FLUID_TB_KJ': 10, 'FLUID_TB_KD': 1, 'FLUID_TB_CJ': 4, 'FLUID_TB_CD': 4, 'FLUID_TB_WAVES': 64, 'FLUID_TB_MINCHUNK': 4, 'FLUID_TILE': 0}
def run(envd, flags=0, steps=1):
    for k in list(os.environ):
        if k.startswith("FLUID_T"): os.environ.pop(k)  # (FLUID_LIB_PATH stays)
    os.environ.update({k: str(v) for k, v in envd.items()})
    kw = {"lib": pkg.FluidLib(os.environ["FLUID_LIB_PATH"])} if os.environ.get("FLUID_LIB_PATH") else {}
    s = pkg.FluidSim(n, diffusion_iters=D, pressure_iters=J, flags=flags, **kw)
    ss = pkg.SplatStream()
    for _ in range(steps):
        s.add_source(*ss.splat()); s.step(1)
    out = {f: s.read(i) for f, i in (("velocity", pkg.FIELD_VELOCITY), ("pressure", pkg.FIELD_PRESSURE), ("divergence", pkg.FIELD_DIVERGENCE), ("density", pkg.FIELD_DENSITY))}
    s.close()
    return out
ref = run({}, pkg.FLAG_UNFUSED)
for name, e in (("full", env), ("cj2", dict(env, FLUID_TB_CJ=2)), ("minchunk1", dict(env, FLUID_TB_MINCHUNK=1)), ("waves3900", dict(env, FLUID_TB_WAVES=3900)), ("cd2", dict(env, FLUID_TB_CD=2)), ("kd5cd4", dict(env, FLUID_TB_KD=5))):
    got = run(e)
    for f in ("velocity", "divergence", "pressure"):
        a, b = got[f].view(np.uint32), ref[f].view(np.uint32)
        if a.ndim == 3: d = (a != b).any(axis=2)
        else: d = a != b
        if d.any():
            ys, xs = np.nonzero(d)
            print(name, f, "mismatch cells", d.sum(), "rows", ys.min(), ys.max(), "cols", xs.min(), xs.max(), "uniq rows", len(np.unique(ys)), "uniq cols", np.unique(xs)[:12])
        else:
            print(name, f, "ok")
got = run(env)
a, b = got["pressure"], ref["pressure"]
d = a.view(np.uint32) != b.view(np.uint32)
ys, xs = np.nonzero(d)
print("nonzero-valued mismatches:", int(((a != b) & d).sum()), "of", int(d.sum()))
for y in (0, 1, 2, 500, 1078, 1079):
    cols = np.nonzero(d[y])[0]
    print("row", y, "cols", cols[:20], "got", a[y, cols[:6]], "ref", b[y, cols[:6]])
cnt = np.bincount(xs, minlength=n)
print("cols with most mismatches:", np.argsort(-cnt)[:20], cnt[np.argsort(-cnt)[:20]])
