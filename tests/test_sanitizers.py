"""AddressSanitizer + UndefinedBehaviorSanitizer over the product's host code (scene loader, list generation, binner):
path frames at several sizes, thousands of random / hostile viewpoints, truncated and bit-flipped WADs
(tests/emul/asan_driver.cpp).  GPU sanitizers are not available on this pool, so the device code is covered by the parity
tests and the exhaustive numerics checks instead."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "doom-rust-renderer_amd", "csrc")


def test_host_code_is_clean_under_asan_ubsan(tmp_path, synth):
    exe = tmp_path / "asan_driver"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-o", str(exe),
                           os.path.join(ROOT, "tests", "emul", "asan_driver.cpp")] +
                          [os.path.join(CSRC, f) for f in ("scene.cpp", "frontend.cpp", "binner.cpp")])
    wad = tmp_path / "quirks.wad"
    wad.write_bytes(synth.build_synth_iwad(1993, quirks=True))
    r = subprocess.run([str(exe), str(wad), os.path.join(ROOT, "tests", "golden", "campath_seed1993.f32"), "E1M1"],
                       capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"), timeout=600)
    assert r.returncode == 0 and "SANITIZER DRIVER OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
