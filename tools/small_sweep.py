import itertools, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fluid_amd_loader
pkg = fluid_amd_loader.load()
n = int(sys.argv[1]); J = int(sys.argv[2])
res = []
for kj, kd, cj, waves, mc in itertools.product((2, 4, 5, 8, 10), (2, 5), (2, 4), (2048, 8192), (1, 2, 4)):
    os.environ.update(FLUID_TB_KJ=str(kj), FLUID_TB_KD=str(kd), FLUID_TB_CJ=str(cj), FLUID_TB_CD="2",
                      FLUID_TB_WAVES=str(waves), FLUID_TB_MINCHUNK=str(mc))
    sim = pkg.FluidSim(n, pressure_iters=J)
    ss = pkg.SplatStream()
    for _ in range(3):
        sim.add_source(*ss.splat())
    sim.step(5)
    ms = min(sim.time_steps(50) for _ in range(3)) / 50
    res.append((ms, kj, kd, cj, waves, mc))
    sim.close()
res.sort()
for r in res[:12]:
    print(r)
print("unfused:", end=" ")
sim = pkg.FluidSim(n, pressure_iters=J, flags=pkg.FLAG_UNFUSED); sim.step(5)
print(min(sim.time_steps(50) for _ in range(3)) / 50)
