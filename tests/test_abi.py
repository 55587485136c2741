"""Boundary checks that need no GPU: the C-ABI library loads, exports every symbol
include/fluid.h declares, and the host-side pieces (params, splat stream, byte model) behave."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from util import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "fluid.h")).read()
    return re.findall(r"FLUID_API\s+[\w\s\*]+?\b(fluid_\w+)\s*\(", text)


def test_header_and_binding_agree(pkg):
    syms = header_symbols()
    assert len(syms) >= 20
    assert sorted(syms) == sorted(pkg.ABI_SYMBOLS)


def test_ctypes_structures_have_the_layout_of_the_header(pkg, tmp_path):
    """Every struct that crosses the C ABI: size and the offset of every field as gcc lays the
    header's declaration out, against the ctypes mirror in capi.py (a field added on one side
    only would shift everything behind it silently)."""
    pairs = {"fluid_params": pkg.capi.FluidParams, "fluid_slab_info": pkg.capi.SlabInfo,
             "fluid_kernel_stat": pkg.capi.KernelStat, "fluid_diag": pkg.capi.FluidDiag,
             "fluid_transport": pkg.capi.FluidTransport, "fluid_comm_stats": pkg.capi.FluidCommStats}
    lines = ['#include <stddef.h>', '#include <stdio.h>', f'#include "{os.path.join(ROOT, "include", "fluid.h")}"',
             'int main(void) {']
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    seen = 0
    for line in out.splitlines():
        cname, fname, val = line.split()
        cls = pairs[cname]
        want = C.sizeof(cls) if fname == "size" else getattr(cls, fname).offset
        assert int(val) == want, f"{cname}.{fname}: header {val}, ctypes {want}"
        seen += 1
    assert seen == sum(len(c._fields_) + 1 for c in pairs.values())


def test_hip_library_exports_every_declared_symbol(pkg):
    path = pkg.hip_lib_path()
    assert os.path.exists(path), "libfluid_hip.so must be built in-tree (see __graft_entry__.build)"
    dll = C.CDLL(path)
    for s in header_symbols():
        assert hasattr(dll, s), f"{s} declared in include/fluid.h but not exported"


def test_hip_library_does_not_reference_the_oracle(pkg):
    out = subprocess.run(["nm", "-D", pkg.hip_lib_path()], capture_output=True, text=True).stdout
    assert "ref_" not in out
    ldd = subprocess.run(["ldd", pkg.hip_lib_path()], capture_output=True, text=True).stdout
    assert "oracle" not in ldd


def test_default_params_are_the_reference_literals(pkg, hip):
    p = hip.default_params()
    assert (p.n, p.diffusion_iters, p.pressure_iters) == (512, 10, 10)  # kernel.cu:438, :584, :620
    assert p.dt == np.float32(0.1) and p.viscosity == np.float32(0.01)  # kernel.cu:31, :181
    assert p.splat_radius == np.float32(0.001)                          # kernel.cu:368
    assert p.flags == 0 and p.slab_count == 0


def test_oracle_and_hip_agree_on_defaults_and_errors(pkg, hip, oracle):
    a, b = hip.default_params(), oracle.default_params()
    assert bytes(a) == bytes(b)
    for code in range(-8, 1):
        assert hip.dll.fluid_strerror(code) == oracle.dll.fluid_strerror(code)


def test_invalid_arguments_return_codes_not_exits(pkg, hip):
    assert hip.dll.fluid_default_params(None) == -1
    assert hip.dll.fluid_init(None, None) == -1
    assert hip.dll.fluid_step(None, 1) == -1
    assert hip.dll.fluid_read(None, 0, None, 0) == -1
    hip.dll.fluid_destroy(None)  # no-op
    p = hip.default_params()
    p.n = 3
    h = C.c_void_p()
    assert hip.dll.fluid_init(C.byref(h), C.byref(p)) == -1 and not h.value


def test_oracle_rejects_bad_sizes(pkg, oracle):
    sim = pkg.FluidSim(8, lib=oracle)
    buf = np.zeros(8, np.float32)
    assert oracle.dll.fluid_read(sim._h, pkg.FIELD_VELOCITY, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == -4
    assert oracle.dll.fluid_read(sim._h, 99, buf.ctypes.data_as(C.c_void_p), buf.nbytes) == -1
    with pytest.raises(pkg.FluidError):
        sim.write(pkg.FIELD_PRESSURE, np.zeros((4, 4), np.float32))


def test_no_gpu_means_loud_failure_not_fallback(pkg, hip):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.FluidError):
        pkg.FluidSim(16)


def test_splat_stream_matches_c_host_driver(pkg):
    exe = os.path.join(ROOT, "host", "fluid_headless")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "host")], check=True)
    out = subprocess.run([exe, "--splats", "--steps", "20"], capture_output=True, text=True, check=True).stdout
    ss = pkg.SplatStream()
    for line in out.strip().splitlines():
        _, sx, sy, fx, fy, preset = line.split()
        psx, psy, pfx, pfy, rgba = ss.splat()
        got = np.array([sx, sy, fx, fy], np.float32)
        assert np.array_equal(got, np.array([psx, psy, pfx, pfy], np.float32))
        assert pkg.COLOR_PRESETS[int(preset)] == rgba
        assert 0.25 <= psx < 0.75 and 0.25 <= psy < 0.75 and -50 <= pfx < 50 and -50 <= pfy < 50


def test_colour_presets_are_the_reference_keys(pkg):
    # kernel.cu:833-850
    assert pkg.COLOR_PRESETS[1] == (1, 1, 1, 1) and pkg.COLOR_PRESETS[2] == (1, 0, 0, 0)
    assert pkg.COLOR_PRESETS[3] == (0, 1, 0, 0) and pkg.COLOR_PRESETS[4] == (0, 0, 1, 0)


def test_algorithmic_bytes_model(pkg):
    # B(D,J) = 88 + 16 D + 12 J
    assert pkg.alg_bytes_per_cell(10, 10) == 368
    assert pkg.alg_bytes_per_cell(10, 40) == 728
    assert pkg.alg_bytes_per_cell(10, 200) == 2648
