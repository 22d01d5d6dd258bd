"""The C++ mirror of the reference's Pixels/Renderer API (csrc/doomgpu.hpp) compiles and Pixels behaves like
src/renderer/pixels.rs (bounds rules included)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "doom-rust-renderer_amd/csrc/doomgpu.hpp"
#include <cassert>
#include <cstdio>
int main() {
    doom::Pixels p(8, 4);
    assert(p.pixels.size() == 8 * 4 * 3);
    doom::Color c{1, 2, 3, 255};
    p.set(7, 3, c);  assert(p.pixels[3 * (3 * 8 + 7) + 2] == 3);
    p.set(8, 0, c);  // x >= W: dropped (pixels.rs:23)
    p.set((size_t)-1, 0, c);
    for (auto b : std::vector<uint8_t>(p.pixels.begin(), p.pixels.begin() + 3)) assert(b == 0);
    p.draw_vertical_line(0, 0, 3, c);   // x <= 0 is skipped (pixels.rs:34)
    assert(p.pixels[0] == 0);
    p.draw_vertical_line(2, -5, 50, c);
    assert(p.pixels[3 * (0 * 8 + 2)] == 1 && p.pixels[3 * (3 * 8 + 2)] == 1);
    p.clear();
    for (auto b : p.pixels) assert(b == 0);
    // type-check the Renderer call shape without running it (needs a GPU)
    if (false) {
        std::vector<uint8_t> wad;
        doom::World w(wad, "e1m1");
        doom::Device d(8, 4);
        d.upload(w);
        doom::Player pl = w.player_start();
        doom::Renderer(p, w, pl, 0.0f, d).render();
    }
    std::puts("ok");
    return 0;
}
'''


def test_cpp_mirror_compiles_and_pixels_semantics(tmp_path):
    src = tmp_path / "mirror.cpp"
    src.write_text(SRC)
    exe = tmp_path / "mirror"
    subprocess.check_call(["g++", "-std=c++17", "-I", ROOT, str(src), "-o", str(exe),
                           os.path.join(ROOT, "doom-rust-renderer_amd", "libdoomgpu.so"), "-Wl,-rpath," + os.path.join(ROOT, "doom-rust-renderer_amd")])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    assert out.strip() == "ok"
