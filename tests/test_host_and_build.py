"""The C host driver over the C-ABI (GPU) and build-time properties of the HIP kernels (CPU)."""
import json
import os
import re
import subprocess

import numpy as np
import pytest

from util import ROOT, rel_err

CSRC = os.path.join(ROOT, "fluid-simulation_amd", "csrc")


def test_tb_kernels_keep_everything_in_registers(tmp_path):
    """The temporally blocked kernels are only fast while every pipeline row lives in VGPRs: an
    innocent source change (e.g. a select between two row references) silently sends the arrays to
    scratch memory (4x slower, seen once).  Cross-compile the file and read the kernel metadata."""
    out = tmp_path / "tb.s"
    flags = re.search(r"^HIPFLAGS = (.*?)\n(\s+(.*?)\n)?", open(os.path.join(CSRC, "Makefile")).read(), re.M)
    flagstr = (flags.group(1).rstrip("\\") + " " + (flags.group(3) or "")).replace("-fPIC", "")
    cmd = ["/opt/rocm/bin/hipcc"] + flagstr.split() + ["-S", "--cuda-device-only",
                                                        os.path.join(CSRC, "fluid_kernels_tb.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    text = out.read_text()
    kernels = re.findall(r"\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.vgpr_count:\s+(\d+)"
                         r".*?\.vgpr_spill_count:\s+(\d+)", text, re.S)
    assert len(kernels) >= 30
    for name, scratch, vgpr, spill in kernels:
        assert int(scratch) == 0 and int(spill) == 0, (name, scratch, spill)
        assert int(vgpr) <= 256, (name, vgpr)          # at least two waves per SIMD
    assert "v_fma" not in text and "v_mac" not in text  # contraction stays off (bit parity)
    assert "v_pk_add_f32" not in text                   # -fno-slp-vectorize
    assert "wave_shr:1" in text and "wave_shl:1" in text


@pytest.mark.gpu
def test_c_host_driver_matches_oracle(pkg, oracle, tmp_path):
    """host/fluid_headless (plain C over the C-ABI, system HIP runtime, no Python in the loop):
    run the seeded script, dump the fields, compare with the oracle driven by the same script."""
    exe = os.path.join(ROOT, "host", "fluid_headless")
    prefix = str(tmp_path / "dump")
    n, J, warm = 256, 10, 20
    res = subprocess.run([exe, "--n", str(n), "--jacobi", str(J), "--warmup", str(warm), "--steps", "0",
                          "--windows", "1", "--dump", prefix], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads(res.stdout.strip().splitlines()[-1])
    assert line["n"] == n and line["pressure_iters"] == J
    ref = pkg.FluidSim(n, pressure_iters=J, lib=oracle)
    pkg.run_seeded(ref, warm)
    for name, fid, comps in (("velocity", pkg.FIELD_VELOCITY, 2), ("density", pkg.FIELD_DENSITY, 4),
                             ("pressure", pkg.FIELD_PRESSURE, 1)):
        got = np.fromfile(f"{prefix}.{name}.f32", dtype=np.float32).reshape(ref.read(fid).shape)
        e = rel_err(got, ref.read(fid))
        assert e <= 1e-5, (name, e)
