// doomgpu.hpp — host-side mirror of the reference's draw API for this path, in C++ because the reference is compiled
// code and no Rust toolchain exists in this image (the Rust binding is shipped as source under rust/).
//
//   reference (src/renderer/pixels.rs:5-47, src/renderer/mod.rs:27-58,118-136, src/game.rs:40-45,505-525)      here
//   Pixels::new() / .pixels / clear / set / draw_vertical_line                                          doom::Pixels
//   Player { position, floor_height, angle }                                                            doom::Player
//   Map + MapObjects + Textures + Sprites + sky_texture + Flats + Palette (borrowed by Renderer::new)   doom::World
//   Renderer::new(&mut pixels, .., &player, timestamp).render()                                         doom::Renderer
//
// Same names, argument meaning and ownership: the caller owns Pixels and World, the Renderer borrows them for one
// frame.  Where the reference panics these wrappers throw doom::Error carrying the dg_status code.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/doomgpu.h"

namespace doom {

struct Error : std::runtime_error {
    int code;
    Error(int c, const char *m) : std::runtime_error(m), code(c) {}
};
inline void check(int rc) { if (rc < 0) throw Error(rc, dg_last_error()); }

struct Color { uint8_t r, g, b, a; };            // sdl2::pixels::Color as the reference uses it
struct Vertex { float x, y; };                   // src/map/vertexes.rs:9-13
struct Player { Vertex position; float floor_height; float angle; };   // src/game.rs:40-45
// What the renderer reads of the game state that thinkers mutate between frames:
struct Sector { int16_t light_level; };                                                          // src/map/sectors.rs:8-17 (src/lights.rs:47-259 writes it)
struct State { const char *sprite; uint8_t frame; bool full_bright; bool is_null; };             // src/info.rs:1265-1273; sprite = "{:?}" of SpriteId; is_null: StateId::S_NULL
struct MapObject { State state; };                                                               // src/map_objects.rs:10-17 (MapObjectThinker :63-121 writes it)

// src/renderer/pixels.rs:5-47 with the frame size a run-time value (the reference's SCREEN_WIDTH/HEIGHT constants).
class Pixels {
public:
    std::vector<uint8_t> pixels;                 // width * height * 3, R,G,B
    int width, height;
    Pixels(int w, int h) : pixels((size_t)w * (size_t)h * 3, 0), width(w), height(h) {}
    void clear() { std::fill(pixels.begin(), pixels.end(), 0); }
    void set(size_t x, size_t y, const Color &c) {                       // pixels.rs:22-31 (y > H, not >=, as written)
        if (x >= (size_t)width || y > (size_t)height) return;
        size_t o = 3 * (y * (size_t)width + x);
        pixels.at(o) = c.r; pixels.at(o + 1) = c.g; pixels.at(o + 2) = c.b;
    }
    void draw_vertical_line(int x, int top, int bottom, const Color &c) {  // pixels.rs:33-47 (skips x <= 0)
        if (x <= 0 || x >= width) return;
        for (int y = top; y < bottom + 1; y++) {
            if (y < 0 || y >= height) continue;
            size_t o = 3 * ((size_t)y * (size_t)width + (size_t)x);
            pixels[o] = c.r; pixels[o + 1] = c.g; pixels[o + 2] = c.b;
        }
    }
};

// Everything Game::new loads besides SDL (src/game.rs:142-167).
class World {
public:
    World(const std::vector<uint8_t> &wad, const std::string &map_name) { check(dg_scene_load_wad(wad.data(), wad.size(), map_name.c_str(), &h_)); }
    ~World() { dg_scene_free(h_); }
    World(const World &) = delete;
    World &operator=(const World &) = delete;
    Player player_start() const {                                        // src/game.rs:151-156 + :376-389
        Player p{{0, 0}, 0.0f, 0.0f};
        check(dg_scene_player_start(h_, &p.position.x, &p.position.y, &p.angle));
        dg_scene_floor_height_at(h_, p.position.x, p.position.y, &p.floor_height);
        return p;
    }
    int sector_count() const { return dg_scene_sector_count(h_); }       // = map.sectors.len()
    int mobj_count() const { return dg_scene_mobj_count(h_); }           // = map_objects.objects.len()
    // Game::new, BEFORE Device::upload: decode a (sprite, frame) a later state may show (Sprites::new loads every sprite lump eagerly,
    // src/graphics/sprites.rs:26-97).  false: the WAD has no such sprite.
    bool preload_sprite_frame(const char *sprite, uint8_t frame) { return dg_scene_sprite_frame(h_, sprite, frame) >= 0; }
    bool sector_floor_height(const Vertex &v, float &out) const { return dg_scene_floor_height_at(h_, v.x, v.y, &out) == 0; }  // bsp.rs:9-44
    dg_scene *handle() const { return h_; }
private:
    dg_scene *h_ = nullptr;
};

// Game::render, before Renderer(..).render(): the light levels and map-object states of this tick (rust/src/gpu.rs sync_state).
// Index = position in `map.sectors` / `map_objects.objects`.
inline void sync_state(World &world, const std::vector<Sector> &sectors, const std::vector<MapObject> &objects) {
    for (size_t i = 0; i < sectors.size(); i++) check(dg_scene_set_sector_light(world.handle(), (int)i, sectors[i].light_level));
    for (size_t i = 0; i < objects.size(); i++) {
        const State &st = objects[i].state;
        check(dg_scene_set_mobj_state(world.handle(), (int)i, st.is_null ? nullptr : st.sprite, st.frame, st.full_bright ? 1 : 0));
    }
}

// One GPU.  Not in the reference (it has no device); plays the role of the borrowed `&mut Pixels` target's backing store.
class Device {
public:
    Device(int width, int height, int max_batch = 1, int device = 0) {
        dg_config cfg{device, width, height, max_batch, 2, 0, DG_FE_AUTO};
        check(dg_create(&cfg, &h_));
    }
    ~Device() { dg_destroy(h_); }
    Device(const Device &) = delete;
    Device &operator=(const Device &) = delete;
    void upload(const World &w) { check(dg_upload_scene(h_, w.handle())); }
    dg_ctx *handle() const { return h_; }
private:
    dg_ctx *h_ = nullptr;
};

// src/renderer/mod.rs:27-58,118-136: construct per frame, call render(), read pixels.pixels.
class Renderer {
public:
    Renderer(Pixels &pixels, const World &world, const Player &player, float timestamp, Device &dev)
        : pixels_(pixels), dev_(dev) {
        (void)world;   // the device already holds the uploaded world; kept in the signature to mirror Renderer::new
        // Vertex::rotate (src/map/vertexes.rs:20-25) evaluates f32::cos / f32::sin of +-angle: the same libm calls, made here,
        // so that the library consumes the caller's bits (trig_valid = 1) instead of evaluating its own
        const float a = player.angle;
        view_ = dg_view{player.position.x, player.position.y, a, player.floor_height, std::cos(a), std::sin(a), std::cos(-a), std::sin(-a), timestamp, 1};
    }
    void render() { check(dg_render_views(dev_.handle(), &view_, 1, pixels_.pixels.data())); }
private:
    Pixels &pixels_;
    Device &dev_;
    dg_view view_;
};

}  // namespace doom
