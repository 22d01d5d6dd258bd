import hashlib
import importlib
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


def _has_gpu() -> bool:
    try:
        out = subprocess.run(["/opt/rocm/bin/rocminfo"], capture_output=True, text=True, timeout=60).stdout
        return "gfx950" in out
    except Exception:
        return False


@pytest.fixture(scope="session")
def dg():
    """The product binding (ctypes over libdoomgpu.so)."""
    return importlib.import_module("doom-rust-renderer_amd")


@pytest.fixture(scope="session")
def synth():
    return importlib.import_module("doom-rust-renderer_amd.synth_wad")


@pytest.fixture(scope="session")
def campath_mod():
    return importlib.import_module("doom-rust-renderer_amd.camera_path")


@pytest.fixture(scope="session")
def oracle():
    import doomref
    doomref.build()
    return doomref


@pytest.fixture(scope="session")
def wad1993(synth):
    return synth.build_synth_iwad(1993)


@pytest.fixture(scope="session")
def wad1994(synth):
    return synth.build_synth_iwad(1994, heavy=True)


@pytest.fixture(scope="session")
def wad1995(synth):
    """The vanilla-shaped variant: arbitrary integer vertices / wall angles, rounded BSP splits, closed doors, 1-degree thing
    angles, wall textures with negative and past-the-bottom patch origins (synth_wad.build_synth_iwad, vanilla=True)."""
    return synth.build_synth_iwad(1995, vanilla=True)


def load_path(seed: int) -> np.ndarray:
    return np.fromfile(os.path.join(GOLDEN, f"campath_seed{seed}.f32"), dtype="<f4").reshape(1000, 8)


@pytest.fixture(scope="session")
def path1993():
    return load_path(1993)


@pytest.fixture(scope="session")
def path1994():
    return load_path(1994)


@pytest.fixture(scope="session")
def path1995():
    return load_path(1995)


@pytest.fixture(scope="session")
def golden_frames():
    return {seed: json.load(open(os.path.join(GOLDEN, f"frames_seed{seed}.json"))) for seed in (1993, 1994, 1995)}


@pytest.fixture(scope="session")
def oracle_scene1993(oracle, wad1993):
    return oracle.Scene(wad1993, "e1m1")


@pytest.fixture(scope="session")
def oracle_scene1994(oracle, wad1994):
    return oracle.Scene(wad1994, "e1m1")


@pytest.fixture(scope="session")
def oracle_scene1995(oracle, wad1995):
    return oracle.Scene(wad1995, "e1m1")


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()


def sha(b) -> str:
    return hashlib.sha256(bytes(b)).hexdigest()
