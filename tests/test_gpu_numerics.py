"""Exhaustive on-device verification of the instruction-level shortcuts used by the raster kernel
(tests/gpu_numerics/numerics_check.hip): `as u8` via v_trunc + v_cvt_pk_u8, the hoisted-reciprocal IEEE divide for the
wall mapper's `ay` and the flat mapper's `x / vy`, and the float floor-modulus helper.  Every domain is enumerated on the
GPU; the kernel may only use a shortcut with zero mismatches."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_numerics")


def test_numerics_shortcuts_are_exact():
    exe = os.path.join(HERE, "numerics_check")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
                           "-o", exe, os.path.join(HERE, "numerics_check.hip")])
    r = subprocess.run(["timeout", "-k", "10", "600", exe], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0 and "NUMERICS OK" in r.stdout, r.stdout + r.stderr
