//! main_dump.rs — headless frame dumper for the UNMODIFIED reference renderer (freewilll/doom-rust-renderer).
//!
//! Purpose: pin parity.  The reference ships no tests or golden frames and cannot be built in this repository's image (no
//! rustc / cargo / libSDL2 / WAD), so every "bit-exact" claim here is against oracle/doomref.c, a restatement.  On a machine
//! that has cargo, libSDL2 (the crate links sdl2 even when no window is opened) and an IWAD, this program renders the
//! reference's own `Pixels` for a list of viewpoints and writes the raw RGB24 bytes, which tools/compare_ref.py then compares
//! with libdoomgpu (dg_frame_checksums / byte for byte) and with the oracle.
//!
//! Install: copy to `src/bin/dump.rs` of the reference checkout and apply tools/ref_dump/README.md (one new 10-line file,
//! one `pub`, the frame size; no behavioural change).  README.md also lists every item used below against the reference
//! line that declares it — this file cannot be compiled here, so it has to be right by inspection.
//!
//! Usage: cargo run -r --bin dump -- --wad doom1.wad --map e1m1 --views views.txt --out frames.rgb
//!   views.txt: one view per line, either `x y angle` (f32, radians; `tools/compare_ref.py --emit-views` writes the committed
//!   camera paths in this format with 9 significant digits, which round-trips f32 exactly) or the word `start` for the
//!   Player-1 start exactly as Game::new takes it (src/game.rs:151-156).  floor_height is derived like
//!   Game::update_current_player_height (src/game.rs:376-389).
//!   frames.rgb: the frames back to back, SCREEN_WIDTH * SCREEN_HEIGHT * 3 bytes each.
//!   stdout: per view `x y angle floor F trig C S CN SN` — the position / angle actually used and the f32 bit patterns of
//!   cos(angle), sin(angle), cos(-angle), sin(-angle) this machine's libm returned, so that the same bits can be handed to
//!   libdoomgpu (`dg_view.trig_valid = 1`) and the oracle: pass it to compare_ref.py as --trig.
use std::fs;
use std::io::Write;
use std::rc::Rc;

use doom_rust_renderer::game::{Game, Player, SCREEN_HEIGHT, SCREEN_WIDTH};
use doom_rust_renderer::graphics::{Flats, Palette, Pictures, Sprites, Texture, Textures};
use doom_rust_renderer::map::{get_thing_by_type, Map, ThingTypes, Vertex};
use doom_rust_renderer::map_objects::MapObjects;
use doom_rust_renderer::renderer::{get_sector_from_vertex, Pixels, Renderer};
use doom_rust_renderer::wad::WadFile;

fn arg(name: &str) -> String {
    let a: Vec<String> = std::env::args().collect();
    let i = a.iter().position(|x| x == name).unwrap_or_else(|| panic!("missing {}", name));
    a[i + 1].clone()
}

fn main() {
    // src/main.rs:59-60: the WAD bytes go into an Rc<WadFile>; Pictures / Flats / Textures keep a clone of the Rc
    let wad_file: Rc<WadFile> = Rc::new(WadFile::new(fs::read(arg("--wad")).expect("wad")));
    let map_name: String = arg("--map");

    // Game::new minus SDL, same order: src/game.rs:142-167  (&wad_file is &Rc<WadFile>; it derefs to &WadFile where that is asked for)
    let map: Map = Map::new(&wad_file, map_name.as_str());                                   // src/map/mod.rs:48
    let palette: Palette = Palette::new(&wad_file);                                          // src/graphics/palette.rs:11
    let mut pictures: Pictures = Pictures::new(&wad_file);                                   // src/graphics/pictures.rs:30
    let mut flats: Flats = Flats::new(&wad_file);                                            // src/graphics/flats.rs:25
    let mut textures: Textures = Textures::new(&wad_file);                                   // src/graphics/textures.rs:132 (one argument)
    let sky_texture: Rc<Texture> = Game::get_sky_texture(map_name.as_str(), &mut textures);  // src/game.rs:199 (made `pub` by the README's edit)
    let map_objects: MapObjects = MapObjects::new(&map);                                     // src/map_objects.rs:25
    let mut sprites: Sprites = Sprites::new(&wad_file, &mut pictures);                       // src/graphics/sprites.rs:26

    let mut out = fs::File::create(arg("--out")).expect("out");
    for line in fs::read_to_string(arg("--views")).expect("views").lines() {
        let (position, angle): (Vertex, f32) = if line.trim() == "start" {
            let s = get_thing_by_type(&map.things, ThingTypes::Player1Start);                // src/game.rs:151-156
            (Vertex::new(s.x, s.y), s.angle)
        } else {
            let f: Vec<f32> = line.split_whitespace().map(|t| t.parse().expect("f32")).collect();
            if f.len() < 3 {
                continue;
            }
            (Vertex::new(f[0], f[1]), f[2])
        };
        // update_current_player_height, src/game.rs:376-389: 0.0 (the value Game::new starts from) unless a sector is found
        let floor_height: f32 = match get_sector_from_vertex(&map, &position) {              // src/renderer/bsp.rs:9
            Some(sector) => sector.borrow().floor_height as f32,                             // Sector.floor_height: i16, src/map/sectors.rs:11
            None => 0.0,
        };
        let player = Player { position, floor_height, angle };                               // src/game.rs:40-45 (already pub, pub fields)
        let mut pixels: Pixels = Pixels::new();                                              // src/renderer/pixels.rs:10-14
        // src/game.rs:505-519 with timestamp 0.0 (no animated-flat advance) and no thinker ticks; src/renderer/mod.rs:37-58,118-136
        Renderer::new(&mut pixels, &map, &map_objects, &mut textures, &mut sprites, Rc::clone(&sky_texture), &mut flats, &palette, &player, 0.0)
            .render();
        assert_eq!(pixels.pixels.len(), (SCREEN_WIDTH * SCREEN_HEIGHT * 3) as usize);
        out.write_all(&pixels.pixels).expect("write");
        let a = player.angle;
        println!("{:e} {:e} {:e} floor {} trig {:08x} {:08x} {:08x} {:08x}", player.position.x, player.position.y, a, floor_height,
                 a.cos().to_bits(), a.sin().to_bits(), (-a).cos().to_bits(), (-a).sin().to_bits());
    }
}
