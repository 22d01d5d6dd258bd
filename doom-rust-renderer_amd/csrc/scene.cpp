// scene.cpp — IWAD container, map lumps and graphics decode into dg::Scene.
//
// On-disk formats: SURVEY.md Appendix B (reference loaders: src/wad.rs, src/map/*.rs, src/graphics/*.rs).
// Where the reference's HashMap/Vec semantics decide a result (duplicate lump or texture names, first map
// marker, lazy texture composition with later patches overwriting — including with transparent texels) the
// same choice is made here; where the reference would panic the loader throws and the C-ABI returns DG_ERR_WAD.
#include "scene.hpp"

#include <algorithm>
#include <cctype>
#include <cstring>
#include <stdexcept>
#include <unordered_map>

namespace dg {
namespace {

struct LoadError : std::runtime_error { using std::runtime_error::runtime_error; };

std::string upper(const std::string &s) {
    std::string r = s;
    for (auto &c : r) c = (char)std::toupper((unsigned char)c);
    return r;
}

struct Lump { std::string name; uint32_t off, size; };

struct Wad {
    const std::vector<uint8_t> &b;
    std::vector<Lump> dir;
    std::unordered_map<std::string, int> by_name;  // last lump of a name wins (HashMap insert, src/wad.rs:153-154)

    explicit Wad(const std::vector<uint8_t> &bytes) : b(bytes) {
        if (b.size() < 12) throw LoadError("file shorter than a WAD header");
        if (std::memcmp(b.data(), "IWAD", 4) != 0) throw LoadError("Unhandled WAD file type (src/wad.rs:90-92)");
        uint32_t n = u32(4), doff = u32(8);
        need(doff, (size_t)n * 16, "directory");
        dir.reserve(n);
        for (uint32_t i = 0; i < n; i++) {
            size_t e = (size_t)doff + (size_t)i * 16;
            Lump l{upper(name8(e + 8)), u32(e), u32(e + 4)};
            by_name[l.name] = (int)i;
            dir.push_back(std::move(l));
        }
    }
    void need(size_t off, size_t n, const char *what) const {
        if (off > b.size() || n > b.size() - off) throw LoadError(std::string("out-of-file read in ") + what);
    }
    uint8_t u8(size_t o) const { need(o, 1, "byte"); return b[o]; }
    int16_t i16(size_t o) const { need(o, 2, "i16"); return (int16_t)(uint16_t)(b[o] | (b[o + 1] << 8)); }
    uint32_t u32(size_t o) const {
        need(o, 4, "u32");
        return (uint32_t)b[o] | ((uint32_t)b[o + 1] << 8) | ((uint32_t)b[o + 2] << 16) | ((uint32_t)b[o + 3] << 24);
    }
    // 8 bytes, NUL-terminated when shorter; case preserved (src/wad.rs:112-126)
    std::string name8(size_t o) const {
        need(o, 8, "name");
        size_t n = 0;
        while (n < 8 && b[o + n] != 0) n++;
        if (b[o + 7] != 0) n = 8;
        return std::string((const char *)&b[o], n);
    }
    int find(const std::string &name) const {
        auto it = by_name.find(upper(name));
        return it == by_name.end() ? -1 : it->second;
    }
    const Lump &map_lump(const std::string &map, int k) const {  // first marker + k (src/wad.rs:175-183)
        std::string m = upper(map);
        for (size_t i = 0; i < dir.size(); i++)
            if (dir[i].name == m) {
                if (i + (size_t)k >= dir.size()) break;
                const Lump &l = dir[i + (size_t)k];
                need(l.off, l.size, "map lump");
                return l;
            }
        throw LoadError("Could not find map lumps for " + map);
    }
};

struct Patch { int16_t ox, oy, pnum; };
struct TexDef { std::string name; int16_t w, h; std::vector<Patch> patches; };

// Row-major Option<u8> image while decoding; -1 = None.
struct Image {
    int w = 0, h = 0;
    std::vector<int16_t> px;
    Image(int w_, int h_) : w(w_), h(h_), px((size_t)std::max(w_, 0) * (size_t)std::max(h_, 0), (int16_t)-1) {}
    int16_t &at(int x, int y) { return px[(size_t)y * (size_t)w + (size_t)x]; }
    int16_t at(int x, int y) const { return px[(size_t)y * (size_t)w + (size_t)x]; }
};

struct Builder {
    Scene &sc;
    Wad wad;
    std::vector<std::string> pnames;
    std::vector<TexDef> texdefs;
    std::unordered_map<std::string, int> bitmap_by_key;
    std::unordered_map<std::string, int> flat_by_name;
    int first_sprite = -1, last_sprite = -1;

    explicit Builder(Scene &s) : sc(s), wad(s.wad) {}

    int add_bitmap(const std::string &key, const Image &im, int16_t left, int16_t top) {
        BitmapInfo bi;
        bi.w = im.w; bi.h = im.h; bi.left_offset = left; bi.top_offset = top;
        bi.texel_off = (uint32_t)sc.texel_idx.size();
        size_t n = (size_t)im.w * (size_t)im.h;
        sc.texel_idx.resize(sc.texel_idx.size() + n);
        sc.texel_opq.resize(sc.texel_opq.size() + n);
        for (int x = 0; x < im.w; x++)
            for (int y = 0; y < im.h; y++) {
                int16_t v = im.at(x, y);
                size_t o = bi.texel_off + (size_t)x * (size_t)im.h + (size_t)y;
                sc.texel_idx[o] = v < 0 ? 0 : (uint8_t)v;
                sc.texel_opq[o] = v < 0 ? 0 : 1;
                if (v < 0) bi.has_holes = 1;
            }
        sc.bitmaps.push_back(bi);
        sc.bitmap_names.push_back(key);
        int id = (int)sc.bitmaps.size() - 1;
        bitmap_by_key[key] = id;
        return id;
    }

    // Picture::new + read_pixels (src/graphics/pictures.rs:66-126)
    Image decode_picture(const std::string &lump_name, int16_t &left, int16_t &top) {
        int li = wad.find(lump_name);
        if (li < 0) throw LoadError("Could not find lump " + lump_name);
        size_t off = wad.dir[(size_t)li].off;
        int16_t w = wad.i16(off), h = wad.i16(off + 2);
        left = wad.i16(off + 4);
        top = wad.i16(off + 6);
        if (w < 0 || h < 0) throw LoadError("picture " + lump_name + " has a negative size");
        Image im(w, h);
        for (int col = 0; col < w; col++) {
            size_t co = off + wad.u32(off + 8 + (size_t)col * 4);
            for (;;) {
                uint8_t ytop = wad.u8(co);
                if (ytop == 0xff) break;
                uint8_t len = wad.u8(co + 1);
                for (int r = 0; r < len; r++) {
                    int y = r + ytop;
                    if (y >= h) throw LoadError("picture " + lump_name + ": post beyond height (reference index panic)");
                    im.at(col, y) = wad.u8(co + 3 + (size_t)r);
                }
                co += (size_t)len + 4;
            }
        }
        return im;
    }
    int picture_bitmap(const std::string &lump_name, bool mirrored) {
        std::string key = "P:" + upper(lump_name) + (mirrored ? ":M" : "");
        auto it = bitmap_by_key.find(key);
        if (it != bitmap_by_key.end()) return it->second;
        int16_t left, top;
        Image im = decode_picture(lump_name, left, top);
        if (mirrored) {  // Picture::mirror pictures.rs:129-147
            Image m(im.w, im.h);
            for (int y = 0; y < im.h; y++)
                for (int x = 0; x < im.w; x++) m.at(x, y) = im.at(im.w - 1 - x, y);
            im = std::move(m);
        }
        return add_bitmap(key, im, left, top);
    }

    // Textures::new: PNAMES + TEXTURE1 [+ TEXTURE2] (src/graphics/textures.rs:132-151,182-255)
    void load_texture_defs() {
        int pi = wad.find("PNAMES");
        if (pi < 0) throw LoadError("Could not find lump PNAMES");
        size_t po = wad.dir[(size_t)pi].off;
        uint32_t np = wad.u32(po);
        for (uint32_t i = 0; i < np; i++) pnames.push_back(wad.name8(po + 4 + (size_t)i * 8));
        for (const char *ln : {"TEXTURE1", "TEXTURE2"}) {
            int ti = wad.find(ln);
            if (ti < 0) {
                if (std::strcmp(ln, "TEXTURE1") == 0) throw LoadError("Could not find lump TEXTURE1");
                continue;
            }
            size_t base = wad.dir[(size_t)ti].off;
            uint32_t n = wad.u32(base);
            for (uint32_t i = 0; i < n; i++) {
                size_t o = base + wad.u32(base + 4 + 4 * (size_t)i);
                TexDef d;
                d.name = upper(wad.name8(o));
                d.w = wad.i16(o + 12);
                d.h = wad.i16(o + 14);
                int16_t pc = wad.i16(o + 20);
                for (int j = 0; j < pc; j++) {
                    size_t q = o + 22 + (size_t)j * 10;
                    d.patches.push_back(Patch{wad.i16(q), wad.i16(q + 2), wad.i16(q + 4)});
                }
                texdefs.push_back(std::move(d));
            }
        }
    }
    // Textures::get + Texture::load (textures.rs:154-179, 74-103).  Returns TEX_NONE for "-" is the caller's job.
    int texture_bitmap(const std::string &name) {
        std::string up = upper(name);
        std::string key = "T:" + up;
        auto it = bitmap_by_key.find(key);
        if (it != bitmap_by_key.end()) return it->second;
        const TexDef *d = nullptr;
        for (auto r = texdefs.rbegin(); r != texdefs.rend(); ++r)   // later definition replaced the earlier one
            if (r->name == up) { d = &*r; break; }
        if (!d) return TEX_UNKNOWN;
        if (d->w <= 0 || d->h <= 0) throw LoadError("texture " + up + " has a non-positive size");
        Image im(d->w, d->h);
        for (const Patch &p : d->patches) {
            if (p.pnum < 0 || (size_t)p.pnum >= pnames.size()) throw LoadError("texture " + up + ": patch number out of range");
            int16_t l, t;
            Image pic = decode_picture(pnames[(size_t)p.pnum], l, t);
            for (int x = 0; x < pic.w; x++)
                for (int y = 0; y < pic.h; y++) {
                    int16_t tx = (int16_t)(uint16_t)((uint16_t)x + (uint16_t)p.ox);
                    int16_t ty = (int16_t)(uint16_t)((uint16_t)y + (uint16_t)p.oy);
                    if (tx >= 0 && tx < d->w && ty >= 0 && ty < d->h) im.at(tx, ty) = pic.at(x, y);  // None overwrites too
                }
        }
        return add_bitmap(key, im, 0, 0);
    }
    int sidedef_texture(const std::string &name) { return name == "-" ? TEX_NONE : texture_bitmap(name); }

    // Flats::get + Flat::new (flats.rs:92-100,116-136)
    int flat(const std::string &name) {
        auto it = flat_by_name.find(name);
        if (it != flat_by_name.end()) return it->second;
        int li = wad.find(name);
        int id = FLAT_MISSING;
        if (li >= 0) {
            const Lump &l = wad.dir[(size_t)li];
            wad.need(l.off, 4096, "flat");
            id = (int)sc.flat_names.size();
            sc.flat_names.push_back(name);
            sc.flat_sky.push_back(name.find("SKY") != std::string::npos ? 1 : 0);
            sc.flat_pool.insert(sc.flat_pool.end(), sc.wad.begin() + l.off, sc.wad.begin() + l.off + 4096);
        }
        flat_by_name[name] = id;
        return id;
    }
};

// Animated flat lists (src/graphics/flats.rs:30-75; from p_spec.c)
const char *const kAnim[9][5] = {
    {"NUKAGE1", "NUKAGE2", "NUKAGE3", nullptr, nullptr},   {"FWATER1", "FWATER2", "FWATER3", "FWATER4", nullptr},
    {"SWATER1", "SWATER2", "SWATER3", "SWATER4", nullptr}, {"LAVA1", "LAVA2", "LAVA3", "LAVA4", nullptr},
    {"BLOOD1", "BLOOD2", "BLOOD3", nullptr, nullptr},      {"RROCK05", "RROCK06", "RROCK07", "RROCK08", nullptr},
    {"SLIME01", "SLIME02", "SLIME03", "SLIME04", nullptr}, {"SLIME05", "SLIME06", "SLIME07", "SLIME08", nullptr},
    {"SLIME09", "SLIME10", "SLIME11", "SLIME12", nullptr},
};
int anim_list_of(const std::string &name) {
    for (int l = 0; l < 9; l++)
        for (int k = 0; kAnim[l][k]; k++)
            if (name == kAnim[l][k]) return l;
    return -1;
}

struct SpawnRow { int16_t id; const char *sprite; uint8_t frame, full_bright, is_null; };
const SpawnRow kSpawn[] = {
#include "../../data/mobj_spawn_table.inc"
};

// get_sky_texture (src/game.rs:199-227): regex `e(\d+)m(\d+)` searched anywhere, else `(\d\d)`, else SKY1.
const char *sky_name_for_map(const std::string &m) {
    for (size_t i = 0; i < m.size(); i++) {
        if (m[i] != 'e') continue;
        size_t j = i + 1, e = j;
        while (e < m.size() && std::isdigit((unsigned char)m[e])) e++;
        for (size_t k = e; k > j; k--)   // backtrack the greedy \d+ until 'm' + digit follows
            if (k + 1 < m.size() && m[k] == 'm' && std::isdigit((unsigned char)m[k + 1])) {
                long ep = std::strtol(m.substr(j, k - j).c_str(), nullptr, 10);
                return ep == 2 ? "SKY2" : ep == 3 ? "SKY3" : "SKY1";
            }
    }
    for (size_t i = 0; i + 1 < m.size(); i++)
        if (std::isdigit((unsigned char)m[i]) && std::isdigit((unsigned char)m[i + 1])) {
            int n = (m[i] - '0') * 10 + (m[i + 1] - '0');
            return n < 12 ? "SKY1" : n < 21 ? "SKY2" : "SKY3";
        }
    return "SKY1";
}

const float kPi = 3.14159265358979323846f;

}  // namespace

// Vertex::is_left_of_line for a node's partition (src/map/vertexes.rs:27-34, src/renderer/bsp.rs:15-19)
static inline bool left_of_partition(const NodeRec &n, float px, float py) {
    float v2x = n.x + n.dx, v2y = n.y + n.dy;
    float ax = px - n.x, ay = py - n.y;
    float bx = v2x - n.x, by = v2y - n.y;
    return ax * by - ay * bx <= 0.0f;
}

int Scene::sector_from_vertex(float x, float y) const {
    int ni = (int)nodes.size() - 1;
    for (;;) {
        const NodeRec &n = nodes[(size_t)ni];
        int16_t child = left_of_partition(n, x, y) ? n.lchild : n.rchild;
        if (child & (int16_t)0x8000) {
            const SubSectorRec &ss = subsectors[(size_t)(child & 0x7fff)];
            for (int k = 0; k < ss.count; k++) {
                const SegRec &sg = segs[(size_t)(ss.first + k)];
                const LinedefRec &ld = linedefs[(size_t)sg.linedef];
                int sd = sg.direction ? ld.back : ld.front;
                if (sd >= 0) return sidedefs[(size_t)sd].sector;
            }
            return -1;
        }
        ni = child & 0x7fff;
    }
}

int Scene::texture_id(const std::string &name) const {
    std::string key = "T:" + upper(name);
    for (size_t i = 0; i < bitmap_names.size(); i++)
        if (bitmap_names[i] == key) return (int)i;
    return TEX_UNKNOWN;
}
int Scene::flat_id(const std::string &name, float timestamp) const {
    int l = anim_list_of(name);
    std::string want = name;
    if (l >= 0) {
        size_t n = 0;
        while (kAnim[l][n]) n++;
        float t = timestamp * 3.0f;
        size_t cyc = !(t > 0.0f) ? 0 : (t >= 18446744073709551616.0f ? SIZE_MAX : (size_t)t);
        want = kAnim[l][cyc % n];
    }
    for (size_t i = 0; i < flat_names.size(); i++)
        if (flat_names[i] == want) return (int)i;
    return FLAT_MISSING;
}
int Scene::sprite_bitmap_id(const std::string &sprite, uint8_t frame, uint8_t rotation) const {
    std::string key = sprite + (char)('A' + frame);
    for (size_t i = 0; i < sprite_frame_keys.size(); i++)
        if (sprite_frame_keys[i] == key) {
            if (rotation > 7) return TEX_UNKNOWN;
            const SpriteFrameRec &f = sprite_frames[i];
            return f.rotate ? f.bitmap[rotation] : f.bitmap[0];
        }
    return TEX_UNKNOWN;
}

static int add_sprite_frame(Builder &b, const std::string &sprite, uint8_t frame) {
    Scene &sc = b.sc;
    std::string key = sprite + (char)('A' + frame);
    for (size_t i = 0; i < sc.sprite_frame_keys.size(); i++)
        if (sc.sprite_frame_keys[i] == key) return (int)i;
    // Sprites::new restricted to one frame (src/graphics/sprites.rs:26-97)
    int rot_bitmap[256];
    bool have[256] = {false};
    for (int idx = b.first_sprite; idx < b.last_sprite; idx++) {
        const std::string &nm = b.wad.dir[(size_t)idx].name;
        if (sprite.size() != 4 || nm.compare(0, 4, sprite) != 0) continue;
        if (nm.size() < 6) throw LoadError("sprite lump " + nm + " too short");
        uint8_t fr = (uint8_t)(nm[4] - 65), ro = (uint8_t)(nm[5] - 48);
        if (fr == frame) { rot_bitmap[ro] = b.picture_bitmap(nm, false); have[ro] = true; }
        if (nm.size() > 6) {
            if (nm.size() < 8) throw LoadError("sprite lump " + nm + " malformed");
            uint8_t fr2 = (uint8_t)(nm[6] - 65), ro2 = (uint8_t)(nm[7] - 48);
            if (fr2 == frame) { rot_bitmap[ro2] = b.picture_bitmap(nm, true); have[ro2] = true; }
        }
    }
    int nkeys = 0;
    for (bool h : have) nkeys += h;
    if (nkeys == 0) throw LoadError("Unknown frame " + std::to_string(frame) + " for " + sprite + " (sprites.rs:104)");
    SpriteFrameRec f{};
    f.rotate = nkeys != 1;
    if (f.rotate) {
        if (nkeys != 8) throw LoadError("Got something other than 8 rotations for " + sprite);
        for (int r = 1; r < 9; r++) {
            if (!have[r]) throw LoadError("sprite " + sprite + " misses a rotation");
            f.bitmap[r - 1] = rot_bitmap[r];
        }
    } else {
        if (!have[0]) throw LoadError("sprite " + sprite + ": single rotation is not 0");
        for (int r = 0; r < 8; r++) f.bitmap[r] = rot_bitmap[0];
    }
    sc.sprite_frames.push_back(f);
    sc.sprite_frame_keys.push_back(key);
    return (int)sc.sprite_frames.size() - 1;
}

int Scene::find_or_add_sprite_frame(const std::string &sprite, uint8_t frame, std::string &err) {
    try {
        Builder b(*this);
        // rebuild the caches the builder needs from what is already in the scene
        for (size_t i = 0; i < bitmap_names.size(); i++) b.bitmap_by_key[bitmap_names[i]] = (int)i;
        int a = b.wad.find("S_START"), e = b.wad.find("S_END");
        b.first_sprite = a; b.last_sprite = e;
        int id = add_sprite_frame(b, sprite, frame);
        revision++;
        rebuild_fs_tables();
        return id;
    } catch (const std::exception &ex) {
        err = ex.what();
        return -1;
    }
}

void Scene::rebuild_fs_tables() {
    fs_sectors.resize(sectors.size());
    for (size_t i = 0; i < sectors.size(); i++) {
        const SectorRec &s = sectors[i];
        fs_sectors[i] = FsSector{s.floor_h, s.ceil_h, s.floor_flat, s.ceil_flat, s.floor_anim, s.ceil_anim, s.ceil_tex_sky};
    }
    fs_anims.resize(anim.size());
    for (size_t i = 0; i < anim.size(); i++) {
        fs_anims[i].n = anim[i].n;
        for (int k = 0; k < 4; k++) fs_anims[i].flat[k] = anim[i].flat[k];
    }
    fs_bitmaps.resize(bitmaps.size());
    for (size_t i = 0; i < bitmaps.size(); i++)
        fs_bitmaps[i] = FsBitmap{bitmaps[i].texel_off, (int16_t)bitmaps[i].w, (int16_t)bitmaps[i].h, bitmaps[i].top_offset, (uint16_t)bitmaps[i].has_holes};
    fs_segs.resize(segs.size());
    for (size_t i = 0; i < segs.size(); i++) {
        const SegRec &sg = segs[i];
        const LinedefRec &ld = linedefs[(size_t)sg.linedef];
        const int fsd = sg.direction ? ld.back : ld.front, bsd = sg.direction ? ld.front : ld.back;   // segs.rs:358-362
        FsSeg f;
        std::memset(&f, 0, sizeof f);
        f.v1x = vx[(size_t)sg.v1]; f.v1y = vy[(size_t)sg.v1]; f.v2x = vx[(size_t)sg.v2]; f.v2y = vy[(size_t)sg.v2];
        f.front_sector = -1; f.back_sector = -1;
        f.tex_mid = f.tex_low = f.tex_up = TEX_NONE;
        if (fsd >= 0) {
            const SidedefRec &sd = sidedefs[(size_t)fsd];
            f.front_sector = sd.sector;
            f.sd_xoff = sd.xoff; f.sd_yoff = sd.yoff;
            f.tex_mid = sd.middle; f.tex_low = sd.lower; f.tex_up = sd.upper;
            if (bsd >= 0) f.back_sector = sidedefs[(size_t)bsd].sector;
        }
        f.seg_offset = sg.offset;
        f.ld_flags = (uint16_t)ld.flags;
        fs_segs[i] = f;
    }
    // every seg belongs to at most one leaf (0xffff: to none — the walk never reaches it)
    fs_ok = !may_panic && subsectors.size() < 0xffffu && !nodes.empty() && !segs.empty();
    fs_seg_leaf.assign(segs.size(), (uint16_t)0xffffu);
    fs_leaf_first.resize(subsectors.size());
    for (size_t l = 0; l < subsectors.size(); l++) {
        fs_leaf_first[l] = (uint32_t)subsectors[l].first;
        for (int i = 0; i < subsectors[l].count; i++) {
            const size_t si = (size_t)(subsectors[l].first + i);
            if (subsectors[l].first < 0 || si >= segs.size() || fs_seg_leaf[si] != 0xffffu) { fs_ok = false; continue; }
            fs_seg_leaf[si] = (uint16_t)l;
        }
    }
    // the tree: depth-first from the root (the last node, map/mod.rs:57); a leaf reached twice or a path longer than 256 is not a tree
    fs_anc_off.assign(subsectors.size() + 1, 0);
    fs_anc.clear();
    if (fs_ok) {
        std::vector<std::vector<uint32_t>> chains(subsectors.size());
        std::vector<uint8_t> seen_leaf(subsectors.size(), 0), seen_node(nodes.size(), 0);
        std::vector<uint32_t> path;
        struct Item { int16_t child; uint32_t depth; uint32_t entry; };
        std::vector<Item> stack;
        const uint32_t root = (uint32_t)nodes.size() - 1;
        seen_node[root] = 1;
        stack.push_back(Item{nodes[root].rchild, 0, root});
        stack.push_back(Item{nodes[root].lchild, 0, root | 0x80000000u});
        while (!stack.empty() && fs_ok) {
            const Item it = stack.back();
            stack.pop_back();
            path.resize(it.depth);
            path.push_back(it.entry);
            if (path.size() > 256) { fs_ok = false; break; }
            if (it.child & (int16_t)0x8000) {
                const size_t l = (size_t)(it.child & 0x7fff);
                if (l >= subsectors.size() || seen_leaf[l]) { fs_ok = false; break; }
                seen_leaf[l] = 1;
                chains[l] = path;
            } else {
                const size_t n = (size_t)it.child;
                if (n >= nodes.size() || seen_node[n]) { fs_ok = false; break; }
                seen_node[n] = 1;
                stack.push_back(Item{nodes[n].rchild, (uint32_t)path.size(), (uint32_t)n});
                stack.push_back(Item{nodes[n].lchild, (uint32_t)path.size(), (uint32_t)n | 0x80000000u});
            }
        }
        for (size_t l = 0; l < subsectors.size() && fs_ok; l++)        // a leaf no path leads to: its segs are never visited
            if (!seen_leaf[l])
                for (int i = 0; i < subsectors[l].count; i++) fs_seg_leaf[(size_t)(subsectors[l].first + i)] = (uint16_t)0xffffu;
        std::vector<uint32_t> segs_left(nodes.size(), 0), segs_right(nodes.size(), 0);        // segs below each child of a node
        for (size_t l = 0; l < subsectors.size() && fs_ok; l++)
            for (uint32_t e : chains[l]) (e >> 31 ? segs_left : segs_right)[e & 0x7fffffffu] += (uint32_t)subsectors[l].count;
        for (size_t l = 0; l < subsectors.size() && fs_ok; l++) {
            fs_anc_off[l] = (uint32_t)fs_anc.size();
            for (uint32_t e : chains[l]) {
                const size_t n = e & 0x7fffffffu;
                fs_anc.push_back(FsAnc{nodes[n].x, nodes[n].y, nodes[n].dx, nodes[n].dy, (e >> 31 ? segs_right[n] : segs_left[n]) | (e & 0x80000000u)});
            }
        }
        fs_anc_off[subsectors.size()] = (uint32_t)fs_anc.size();
    }
    fs_mobjs.resize(mobjs.size());
    for (size_t i = 0; i < mobjs.size(); i++) fs_mobjs[i] = FsMobj{mobjs[i].x, mobjs[i].y, mobjs[i].angle, mobjs[i].sector};
}

Scene *load_scene_from_wad(const uint8_t *bytes, size_t len, const char *map_name, std::string &err) {
    Scene *sc = new Scene();
    try {
        sc->wad.assign(bytes, bytes + len);
        sc->map_name = map_name;
        Builder b(*sc);
        const Wad &w = b.wad;
        b.first_sprite = w.find("S_START");
        b.last_sprite = w.find("S_END");
        if (b.first_sprite < 0 || b.last_sprite < 0) throw LoadError("Could not find lump S_START / S_END (src/wad.rs:105-106)");

        enum { THINGS = 1, LINEDEFS, SIDEDEFS, VERTEXES, SEGS, SSECTORS, NODES, SECTORS };
        // VERTEXES (4 B)
        {
            const Lump &l = w.map_lump(map_name, VERTEXES);
            size_t n = l.size / 4;
            sc->vx.resize(n); sc->vy.resize(n);
            for (size_t i = 0; i < n; i++) { sc->vx[i] = (float)w.i16(l.off + i * 4); sc->vy[i] = (float)w.i16(l.off + i * 4 + 2); }
        }
        // SECTORS (26 B)
        std::vector<std::string> sec_floor, sec_ceil;
        {
            const Lump &l = w.map_lump(map_name, SECTORS);
            size_t n = l.size / 26;
            sc->sectors.resize(n);
            for (size_t i = 0; i < n; i++) {
                size_t o = l.off + i * 26;
                SectorRec &s = sc->sectors[i];
                std::memset(&s, 0, sizeof s);
                s.floor_h = w.i16(o); s.ceil_h = w.i16(o + 2); s.light = w.i16(o + 20);
                std::string fn = w.name8(o + 4), cn = w.name8(o + 12);
                s.floor_sky = fn.find("SKY") != std::string::npos;
                s.ceil_sky = s.ceil_tex_sky = cn.find("SKY") != std::string::npos;
                for (int side = 0; side < 2; side++) {
                    const std::string &nm = side ? cn : fn;
                    int al = anim_list_of(nm), fid = FLAT_MISSING, aidx = -1;
                    if (al >= 0) {
                        AnimList a{};
                        while (kAnim[al][a.n]) { a.flat[a.n] = b.flat(kAnim[al][a.n]); a.n++; }
                        sc->anim.push_back(a);
                        aidx = (int)sc->anim.size() - 1;
                    } else {
                        fid = b.flat(nm);
                    }
                    (side ? s.ceil_flat : s.floor_flat) = fid;
                    (side ? s.ceil_anim : s.floor_anim) = aidx;
                }
            }
        }
        // SIDEDEFS (30 B)
        b.load_texture_defs();
        {
            const Lump &l = w.map_lump(map_name, SIDEDEFS);
            size_t n = l.size / 30;
            sc->sidedefs.resize(n);
            for (size_t i = 0; i < n; i++) {
                size_t o = l.off + i * 30;
                SidedefRec &s = sc->sidedefs[i];
                s.xoff = (float)w.i16(o); s.yoff = (float)w.i16(o + 2);
                s.upper = b.sidedef_texture(w.name8(o + 4));
                s.lower = b.sidedef_texture(w.name8(o + 12));
                s.middle = b.sidedef_texture(w.name8(o + 20));
                int16_t sec = w.i16(o + 28);
                if (sec < 0 || (size_t)sec >= sc->sectors.size()) throw LoadError("sidedef references a missing sector");
                s.sector = sec;
            }
        }
        // LINEDEFS (14 B)
        {
            const Lump &l = w.map_lump(map_name, LINEDEFS);
            size_t n = l.size / 14;
            sc->linedefs.resize(n);
            for (size_t i = 0; i < n; i++) {
                size_t o = l.off + i * 14;
                LinedefRec &d = sc->linedefs[i];
                int16_t v1 = w.i16(o), v2 = w.i16(o + 2), f = w.i16(o + 10), bk = w.i16(o + 12);
                d.flags = w.i16(o + 4); d.pad = 0;
                if (v1 < 0 || v2 < 0 || (size_t)v1 >= sc->vx.size() || (size_t)v2 >= sc->vx.size()) throw LoadError("linedef references a missing vertex");
                if ((f != -1 && (f < 0 || (size_t)f >= sc->sidedefs.size())) || (bk != -1 && (bk < 0 || (size_t)bk >= sc->sidedefs.size())))
                    throw LoadError("linedef references a missing sidedef");
                d.v1 = v1; d.v2 = v2; d.front = f; d.back = bk;
            }
        }
        // SEGS (12 B)
        {
            const Lump &l = w.map_lump(map_name, SEGS);
            size_t n = l.size / 12;
            sc->segs.resize(n);
            for (size_t i = 0; i < n; i++) {
                size_t o = l.off + i * 12;
                SegRec &s = sc->segs[i];
                int16_t v1 = w.i16(o), v2 = w.i16(o + 2), ld = w.i16(o + 6);
                if (v1 < 0 || v2 < 0 || ld < 0 || (size_t)v1 >= sc->vx.size() || (size_t)v2 >= sc->vx.size() || (size_t)ld >= sc->linedefs.size())
                    throw LoadError("seg references a missing vertex/linedef");
                s.v1 = v1; s.v2 = v2; s.linedef = ld; s.direction = w.i16(o + 8) != 0; s.offset = w.i16(o + 10); s.pad = 0;
            }
        }
        // SSECTORS (4 B)
        {
            const Lump &l = w.map_lump(map_name, SSECTORS);
            size_t n = l.size / 4;
            sc->subsectors.resize(n);
            for (size_t i = 0; i < n; i++) {
                int16_t cnt = w.i16(l.off + i * 4), first = w.i16(l.off + i * 4 + 2);
                if (cnt < 0 || first < 0 || (size_t)(first + cnt) > sc->segs.size()) throw LoadError("subsector references missing segs");
                sc->subsectors[i] = SubSectorRec{first, cnt};
            }
        }
        // NODES (28 B); children precede parents, root = last (src/map/nodes.rs:45-83, src/map/mod.rs:57)
        {
            const Lump &l = w.map_lump(map_name, NODES);
            size_t n = l.size / 28;
            if (n == 0) throw LoadError("map has no BSP nodes");
            sc->nodes.resize(n);
            for (size_t i = 0; i < n; i++) {
                size_t o = l.off + i * 28;
                NodeRec &d = sc->nodes[i];
                d.x = (float)w.i16(o); d.y = (float)w.i16(o + 2); d.dx = (float)w.i16(o + 4); d.dy = (float)w.i16(o + 6);
                d.rchild = w.i16(o + 24); d.lchild = w.i16(o + 26);
                for (int16_t ch : {d.rchild, d.lchild}) {
                    size_t idx = (size_t)(ch & 0x7fff);
                    if (ch & (int16_t)0x8000) { if (idx >= sc->subsectors.size()) throw LoadError("node references a missing subsector"); }
                    else if (idx >= i) throw LoadError("node references a node that is not loaded yet");
                }
            }
        }
        // Bounding boxes of the subtrees (children precede parents, so one ascending pass suffices) and whether any record
        // could make the reference panic when its seg is processed (then no seg may be skipped unseen, frontend.cpp).
        {
            auto grow = [](float *bb, float x, float y) { bb[0] = std::min(bb[0], x); bb[1] = std::min(bb[1], y); bb[2] = std::max(bb[2], x); bb[3] = std::max(bb[3], y); };
            for (size_t i = 0; i < sc->nodes.size(); i++) {
                NodeRec &d = sc->nodes[i];
                const int16_t child[2] = {d.rchild, d.lchild};
                for (int side = 0; side < 2; side++) {
                    float *bb = d.bb[side];
                    bb[0] = bb[1] = 3.0e38f; bb[2] = bb[3] = -3.0e38f;
                    const size_t idx = (size_t)(child[side] & 0x7fff);
                    if (child[side] & (int16_t)0x8000) {
                        const SubSectorRec &ss = sc->subsectors[idx];
                        for (int k = 0; k < ss.count; k++) {
                            const SegRec &sg = sc->segs[(size_t)(ss.first + k)];
                            grow(bb, sc->vx[(size_t)sg.v1], sc->vy[(size_t)sg.v1]);
                            grow(bb, sc->vx[(size_t)sg.v2], sc->vy[(size_t)sg.v2]);
                        }
                    } else {
                        for (int s2 = 0; s2 < 2; s2++) { grow(bb, sc->nodes[idx].bb[s2][0], sc->nodes[idx].bb[s2][1]); grow(bb, sc->nodes[idx].bb[s2][2], sc->nodes[idx].bb[s2][3]); }
                    }
                }
            }
            sc->may_panic = false;
            for (const SidedefRec &sd : sc->sidedefs) sc->may_panic |= sd.upper == TEX_UNKNOWN || sd.lower == TEX_UNKNOWN || sd.middle == TEX_UNKNOWN;
            for (const SectorRec &se : sc->sectors) sc->may_panic |= (se.floor_anim < 0 && se.floor_flat < 0) || (se.ceil_anim < 0 && se.ceil_flat < 0);
            for (const AnimList &a : sc->anim)
                for (int k = 0; k < a.n; k++) sc->may_panic |= a.flat[k] < 0;
        }
        // Palette (first 768 bytes of PLAYPAL)
        {
            int pi = w.find("PLAYPAL");
            if (pi < 0) throw LoadError("Could not find lump PLAYPAL");
            w.need(w.dir[(size_t)pi].off, 768, "PLAYPAL");
            std::memcpy(sc->palette, &sc->wad[w.dir[(size_t)pi].off], 768);
        }
        sc->sky_bitmap = b.texture_bitmap(sky_name_for_map(map_name));
        if (sc->sky_bitmap < 0) throw LoadError("Unknown texture for the sky (src/game.rs:199-227)");
        // THINGS (10 B) -> player start + map objects in spawn state
        {
            const Lump &l = w.map_lump(map_name, THINGS);
            size_t n = l.size / 10;
            for (size_t i = 0; i < n; i++) {
                size_t o = l.off + i * 10;
                float x = (float)w.i16(o), y = (float)w.i16(o + 2);
                float angle = (float)w.i16(o + 4) * (kPi / 180.0f);   // f32::to_radians (src/map/things.rs:36)
                int16_t type = w.i16(o + 6);
                if (type == 1 && !sc->has_start) { sc->start_x = x; sc->start_y = y; sc->start_angle = angle; sc->has_start = true; }
                if ((type >= 1 && type <= 4) || type == 11) continue;                 // map_objects.rs:31-36
                const SpawnRow *row = nullptr;
                for (const SpawnRow &r : kSpawn)
                    if (r.id == type) row = &r;
                if (!row) throw LoadError("thing type " + std::to_string(type) + " has no MapObjectInfo (reference unwrap panic)");
                MapObjectRec m{};
                m.x = x; m.y = y; m.angle = angle;
                m.full_bright = row->full_bright;
                m.sprite_frame = row->is_null ? -1 : add_sprite_frame(b, row->sprite, row->frame);
                m.sector = sc->sector_from_vertex(x, y);
                sc->mobjs.push_back(m);
            }
        }
        sc->rebuild_fs_tables();
        return sc;
    } catch (const std::exception &ex) {
        err = ex.what();
        delete sc;
        return nullptr;
    }
}

}  // namespace dg
