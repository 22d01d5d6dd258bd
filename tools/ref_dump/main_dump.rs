//! main_dump.rs — headless frame dumper for the UNMODIFIED reference renderer (freewilll/doom-rust-renderer).
//!
//! Purpose: pin parity.  The reference ships no tests or golden frames and cannot be built in this repository's image (no
//! rustc / cargo / libSDL2 / WAD), so every "bit-exact" claim here is against oracle/doomref.c, a restatement.  On a machine
//! that has cargo, libSDL2 (the crate links sdl2 even when no window is opened) and an IWAD, this program renders the
//! reference's own `Pixels` for a list of viewpoints and writes the raw RGB24 bytes, which tools/compare_ref.py then compares
//! with libdoomgpu (dg_frame_checksums / byte for byte) and with the oracle.
//!
//! Install: copy to `src/bin/dump.rs` of the reference checkout and apply tools/ref_dump/README.md (four visibility edits,
//! no behavioural change).  Build with the frame size under test in `src/game.rs:28-29`.
//!
//! Usage: cargo run -r --bin dump -- --wad doom1.wad --map e1m1 --views views.txt --out frames.rgb
//!   views.txt: one `x y angle` per line (f32, radians); floor_height is derived like Game::update_current_player_height
//!   (src/game.rs:376-389).  frames.rgb: the frames back to back, SCREEN_WIDTH * SCREEN_HEIGHT * 3 bytes each.
//!   It also prints, per view, the f32 bit patterns of cos/sin(+-angle) this machine's libm returned, so that the same bits
//!   can be handed to libdoomgpu (`dg_view.trig_valid = 1`).
use std::fs;
use std::io::Write;
use std::rc::Rc;

use doom_rust_renderer::game::{get_sky_texture, Player, SCREEN_HEIGHT, SCREEN_WIDTH};   // made `pub` by the README's edits
use doom_rust_renderer::graphics::{Flats, Palette, Pictures, Sprites, Textures};
use doom_rust_renderer::map::Map;
use doom_rust_renderer::map_objects::MapObjects;
use doom_rust_renderer::renderer::{get_sector_from_vertex, Pixels, Renderer};
use doom_rust_renderer::vertexes::Vertex;
use doom_rust_renderer::wad::WadFile;

fn arg(name: &str) -> String {
    let a: Vec<String> = std::env::args().collect();
    let i = a.iter().position(|x| x == name).unwrap_or_else(|| panic!("missing {}", name));
    a[i + 1].clone()
}

fn main() {
    let wad_file = WadFile::new(fs::read(arg("--wad")).expect("wad"));
    let map_name = arg("--map");
    // Game::new minus SDL: src/game.rs:142-167
    let map = Map::new(&wad_file, map_name.as_str());
    let palette = Palette::new(&wad_file);
    let mut pictures = Pictures::new(&wad_file);
    let mut flats = Flats::new(&wad_file);
    let mut textures = Textures::new(&wad_file, &mut pictures);
    let sky_texture: Rc<_> = get_sky_texture(&map_name, &mut textures);          // src/game.rs:199-227
    let map_objects = MapObjects::new(&map);
    let mut sprites = Sprites::new(&wad_file, &mut pictures);

    let mut out = fs::File::create(arg("--out")).expect("out");
    for line in fs::read_to_string(arg("--views")).expect("views").lines() {
        let f: Vec<f32> = line.split_whitespace().map(|t| t.parse().expect("f32")).collect();
        if f.len() < 3 { continue; }
        let position = Vertex::new(f[0], f[1]);
        // update_current_player_height, src/game.rs:376-389
        let floor_height = match get_sector_from_vertex(&map, &position) {
            Some(sector) => sector.borrow().floor_height as f32,
            None => 0.0,
        };
        let player = Player { position, floor_height, angle: f[2] };
        let mut pixels = Pixels::new();                                          // src/renderer/pixels.rs:10-14
        Renderer::new(&mut pixels, &map, &map_objects, &mut textures, &mut sprites, Rc::clone(&sky_texture), &mut flats, &palette, &player, 0.0)
            .render();                                                           // src/renderer/mod.rs:118-136, timestamp 0.0, no thinker ticks
        assert_eq!(pixels.pixels.len(), SCREEN_WIDTH as usize * SCREEN_HEIGHT as usize * 3);
        out.write_all(&pixels.pixels).expect("write");
        let a = f[2];
        println!("{} {} {} floor {} trig {:08x} {:08x} {:08x} {:08x}", f[0], f[1], a, floor_height,
                 a.cos().to_bits(), a.sin().to_bits(), (-a).cos().to_bits(), (-a).sin().to_bits());
    }
}
