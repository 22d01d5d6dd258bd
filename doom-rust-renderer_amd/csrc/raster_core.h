// raster_core.h — the per-column and per-pixel arithmetic of the reference's three texture mappers, as
// host/device inline functions.  The HIP kernels (kernels.hip) are the only product code that calls them;
// tests/emul compiles the same bodies on the CPU to check list generation without a GPU.
//
//   wall / masked wall / sprite column : render_vertical_bitmap_line  src/renderer/bitmap_render.rs:213-276
//   floor / ceiling                    : draw_visplane                src/renderer/visplanes.rs:94-130
//   sky                                : draw_sky                     src/renderer/visplanes.rs:42-80
//   lighting                           : diminish_color               src/renderer/bitmap_render.rs:190-208
//
// Every expression keeps the reference's operand order; nothing may be contracted into an FMA.
#pragma once
#include "lists_dev.h"
#include "rust_num.h"

namespace dg {

DG_HD uint32_t f32_bits(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    union { float f; uint32_t u; } v; v.f = f; return v.u;
#endif
}
DG_HD float bits_f32(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    union { float f; uint32_t u; } v; v.u = u; return v.f;
#endif
}
DG_HD int32_t lo_i16(uint32_t w) { return (int32_t)(int16_t)(w & 0xffffu); }
// DevRSpan word 0: ctop (bits 0-13) | plain-flag (bit 14) | immediate-flag (bit 15) | cbot (bits 16-29) | kind (bits 30-31); rows are
// < 16384.  plain: the span's pixels take the short form of their mapper — a wall whose bitmap height is a power of two (mask instead
// of modulus) and whose texture row stays inside i16 on its rows (no saturation in the `as i16`: resolve_wall_span), a floor / ceiling whose
// numerators are inside the prepared divide's verified domain (div_guard_ok).
DG_HD uint32_t pack_w0(int32_t ctop, int32_t cbot, uint32_t kind, bool immediate, bool plain) {
    return (uint32_t)(ctop & 0x3fff) | (plain ? 0x4000u : 0u) | (immediate ? 0x8000u : 0u) | ((uint32_t)(cbot & 0x3fff) << 16) | (kind << 30);
}
DG_HD bool w0_plain(uint32_t w0) { return (w0 & 0x4000u) != 0; }
DG_HD int32_t w0_ctop(uint32_t w0) { return (int32_t)(w0 & 0x3fffu); }
DG_HD int32_t w0_cbot(uint32_t w0) { return (int32_t)((w0 >> 16) & 0x3fffu); }
DG_HD uint32_t w0_kind(uint32_t w0) { return w0 >> 30; }
DG_HD bool w0_immediate(uint32_t w0) { return (w0 & 0x8000u) != 0; }
DG_HD int32_t hi_i16(uint32_t w) { return (int32_t)(int16_t)(w >> 16); }

// ---- instruction-level shortcuts (each one is verified exhaustively on the GPU by tests/gpu_numerics) ----------------
//
// `f as u8` in two instructions: v_cvt_pk_u8_f32 saturates to 0..255 and maps NaN to 0 but ROUNDS (measured: 41.9 M
// of the 2^32 patterns differ from truncation), so the value is truncated first (v_trunc_f32 is exact).
DG_HD uint32_t f32_as_u8_pk(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r;
    float t = __builtin_truncf(f);
    asm("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(r) : "v"(t));
    return r;
#else
    return (uint32_t)f32_as_u8(f);
#endif
}

// IEEE-exact n / d with the denominator-only part of the divide hoisted.  hipcc expands `n / d` to
//   v_div_scale x2, v_rcp, fma, fma | mul, fma, fma, fma, v_div_fmas, v_div_fixup
// and the two v_div_scale are identities (and v_div_fmas a plain fma) unless an operand is denormal / huge / tiny.
// prepare_rcp() is the part left of the bar (a reciprocal refined by one Newton step).  div_prepared() is
//   q = n * r;  q += (n - d*q) * r  (one fused residual correction, Markstein's scheme);  v_div_fixup
// i.e. the compiler's sequence without its second correction, which never changes the result on the two domains the
// kernel uses it for: d = integer row differences with n = integer row offsets (all 131 071 x 81 919 pairs), and
// d = CFY - y (every multiple of 0.5 in [-8192, 8192]) with EVERY f32 numerator inside the guard band
// (7.09e13 quotients) — tests/gpu_numerics/numerics_check.hip enumerates both completely and requires zero mismatches
// against `n / d`.  v_div_fixup supplies the IEEE results for d == 0, n == 0, infinities and NaN.  On the host the plain
// quotient is the same value.
DG_HD float prepare_rcp(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r0 = __builtin_amdgcn_rcpf(d);
    float e = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
#else
    (void)d;
    return 0.0f;
#endif
}
DG_HD float div_prepared(float n, float d, float r) {
#if defined(__HIP_DEVICE_COMPILE__)
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    q = __builtin_fmaf(e, r, q);
    return __builtin_amdgcn_div_fixupf(q, d, n);
#else
    (void)r;
    return n / d;
#endif
}
// div_prepared without the final v_div_fixup, for callers that exclude its special cases themselves: v_div_fixup only changes the
// result when d is 0 / Inf / NaN or n is Inf / NaN (and for results in the denormal / overflow range, which the guard band
// excludes).  The wall mapper's d is a non-zero integer or 0 (then uy1 is NaN and the row is NaN whatever ay is); the flat
// mapper excludes the row with vy == 0 and numerators outside div_guard_ok.  tests/gpu_numerics enumerates both domains for
// this form too.  One difference remains and is harmless: n = -0.0 with d > 0 gives +0.0 where IEEE gives -0.0; every consumer
// of these quotients (`as i16`, products that are summed with a non-zero term or converted) maps both zeros to the same result.
DG_HD float div_prepared_nofix(float n, float d, float r) {
#if defined(__HIP_DEVICE_COMPILE__)
    float q = n * r;
    float e = __builtin_fmaf(-d, q, n);
    return __builtin_fmaf(e, r, q);
#else
    (void)r;
    return n / d;
#endif
}
// Numerators for which div_prepared is used: zero, or 2^-64 <= |n| <= 2^64 (NaN / Inf / tiny / huge take the plain divide).
DG_HD bool div_guard_ok(float n) {
    uint32_t e = (f32_bits(n) >> 23) & 0xffu;
    return (f32_bits(n) & 0x7fffffffu) == 0u || (e >= 127u - 64u && e <= 127u + 64u);
}

// diminish_color's factor for a sector light (already divided by 255) and an i16 distance.
DG_HD float light_factor(float lightf, int32_t distance_i16) {
    float factor = lightf - (float)distance_i16 * (1.0f / (16.0f * 256.0f));
    // `if factor < 0.0 { factor = 0.0 }`: factor is never NaN here and -0.0 vs +0.0 cannot change a `as u8` result,
    // so a single max does it.
    return __builtin_fmaxf(factor, 0.0f);
}

// palette entry (r | g<<8 | b<<16) times factor, each channel `as u8`; returns r | g<<8 | b<<16.
DG_HD uint32_t shade(uint32_t rgbx, float factor) {
#if defined(__HIP_DEVICE_COMPILE__)
    // 4 instructions per channel: v_cvt_f32_ubyteN, v_mul_f32, v_trunc_f32, v_cvt_pk_u8_f32 (packs in place)
    float r = __builtin_truncf((float)(rgbx & 255u) * factor);
    float g = __builtin_truncf((float)((rgbx >> 8) & 255u) * factor);
    float b = __builtin_truncf((float)((rgbx >> 16) & 255u) * factor);
    uint32_t o;
    asm("v_cvt_pk_u8_f32 %0, %1, 0, 0" : "=v"(o) : "v"(r));
    asm("v_cvt_pk_u8_f32 %0, %1, 1, %2" : "=v"(o) : "v"(g), "v"(o));
    asm("v_cvt_pk_u8_f32 %0, %1, 2, %2" : "=v"(o) : "v"(b), "v"(o));
    return o;
#else
    int32_t r = f32_as_u8((float)(rgbx & 255u) * factor);
    int32_t g = f32_as_u8((float)((rgbx >> 8) & 255u) * factor);
    int32_t b = f32_as_u8((float)((rgbx >> 16) & 255u) * factor);
    return (uint32_t)r | ((uint32_t)g << 8) | ((uint32_t)b << 16);
#endif
}

// Approximate reciprocal for the modulus helper only (any value within 2^-22 of 1/n works, see floor_mod_fast).
DG_HD float approx_rcp(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_rcpf(v);
#else
    return 1.0f / v;
#endif
}

// Floor modulus of an i16 value by a bitmap dimension 0 < n <= 32767 without an integer divide:
// power-of-two sizes mask; otherwise q = floor(t * rcp) is within 1 of the true quotient
// (|t| <= 32768 and a 2^-22 relative error give an absolute error < 0.01), and one fix-up step repairs it.
DG_HD int32_t floor_mod_fast(int32_t t, int32_t n, int32_t pow2_mask, float rcp_n) {
    if (pow2_mask) return t & pow2_mask;
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t q = (int32_t)__builtin_floorf((float)t * rcp_n);
#else
    float qf = (float)t * rcp_n;
    int32_t q = (int32_t)qf;
    if ((float)q > qf) q -= 1;
#endif
    int32_t r = t - q * n;
    if (r < 0) r += n;
    if (r >= n) r -= n;
    return r;
}

// ---- setup: DevSpan + record -> self-contained DevRSpan (one lane per span), already in the form the tile kernel evaluates
// per pixel (lists_dev.h): nothing is re-derived when a tile stages the span ----------------------------------------------

// Column-invariant part of render_vertical_bitmap_line (bitmap_render.rs:241-251): texture column + light factor.
DG_HD DevRSpan resolve_wall_span(const DevSpan &sp, const DevWallRec &r) {
    float ax = (float)((int32_t)sp.x - r.start_x) / r.dxf;
    float oma = 1.0f - ax;
    float den = oma * r.C + ax * r.D;
    int32_t tx = f32_as_i16((oma * r.A + ax * r.B) / den);
    tx = wrap_i16(tx + r.off_x);
    tx = floor_mod_i16(tx, r.w);
    int32_t z = f32_as_i16((oma + ax) / den);
    const int32_t h = r.h;
    // ay = (y - top_y) / (bottom_y - top_y): the denominator is a span constant.  When it is 0 the reference's
    // `h + (1.0 - ay) * 0.0 + ay * uy1` is NaN for every row (ay is +-Inf or NaN); a NaN uy1 reproduces exactly that,
    // and for a finite ay the middle term is +-0.0 and drops out (h >= 1), so the kernel evaluates h + ay * uy1.
    const float d = (float)((int32_t)sp.bot_y - (int32_t)sp.top_y);
    const bool pot = (h & (h - 1)) == 0;
    // plain: the pixel takes the short form of the mapper — the modulus is a mask AND `h + ay * uy1` cannot leave i16 on the rows the span
    // writes, so that its `as i16` is a bare conversion.  The expression is monotonic in y (a correctly rounded quotient by a constant, a
    // product with a constant and a sum with a constant are), so its values on the first and the last row bound all of them; a NaN
    // (d == 0) fails both comparisons.  IEEE quotients here (`/`), so that host binner and device scatter set the same bit.
    const float uy1 = d == 0.0f ? bits_f32(0x7fc00000u) : r.uy1;
    const float v_top = (float)h + ((float)((int32_t)sp.ctop - (int32_t)sp.top_y) / d) * uy1;
    const float v_bot = (float)h + ((float)((int32_t)sp.cbot - (int32_t)sp.top_y) / d) * uy1;
    const bool in_i16 = __builtin_fabsf(v_top) < 32000.0f && __builtin_fabsf(v_bot) < 32000.0f;
    DevRSpan o;
    o.w[0] = pack_w0(sp.ctop, sp.cbot, SPAN_WALL, r.has_holes != 0, pot && in_i16);
    o.w[1] = f32_bits(d);
    o.w[2] = r.texel_off + (uint32_t)tx * (uint32_t)h;          // start of the texture column (column-major planes)
    o.w[3] = f32_bits(light_factor(r.lightf, z));
    o.w[4] = d == 0.0f ? 0x7fc00000u : f32_bits(r.uy1);
    o.w[5] = (uint32_t)(uint16_t)sp.top_y | ((uint32_t)(uint16_t)r.off_y << 16);
    o.w[6] = f32_bits(pot ? (float)h : -(float)h);              // the sign says which modulus the pixel needs
    o.w[7] = f32_bits(prepare_rcp(d));
    return o;
}
// flats_rel: offset of the flats from the texel index plane (one allocation, so that every kind gathers from one base).
DG_HD DevRSpan resolve_flat_span(const DevSpan &sp, const DevPlaneRec &p, const DevConsts &k, uint32_t flats_rel) {
    const float vx = (k.CFX - (float)sp.x) / k.ARC;                      // visplanes.rs:108
    const float wzvx = p.wz * vx;                                        // numerator of wy = wz * vx / vy (visplanes.rs:114)
    // plain (the short form of the mapper): both numerators inside the prepared divide's verified domain, and a light level below 7 x 255
    // (the tile kernel's short form leaves `wx as i16` unclamped above: beyond 32767 the factor is negative whatever the clamp says —
    // light / 255 - 7.99 — as long as light / 255 stays below that)
    const bool guard_ok = div_guard_ok(wzvx) && div_guard_ok(p.gwz) && p.lightf < 7.0f;
    DevRSpan o;
    o.w[0] = pack_w0(sp.ctop, sp.cbot, SPAN_FLAT, false, guard_ok);
    o.w[1] = f32_bits(wzvx);
    o.w[2] = flats_rel + p.flat_off;
    o.w[3] = 0;
    o.w[4] = f32_bits(p.gwz);                                            // numerator of wx = GCFX * wz / vy (visplanes.rs:113)
    o.w[5] = f32_bits(p.lightf);
    o.w[6] = guard_ok ? 0x100u : 0u;
    o.w[7] = 0;
    return o;
}
// draw_sky's texture column (visplanes.rs:65-66); factor 0 (and column 0) when the reference would index outside the sky bitmap.
DG_HD DevRSpan resolve_sky_span(const DevSpan &sp, const DevScene &sc, const DevConsts &k, const DevFrame &f) {
    int32_t tx = f32_as_i16((float)sp.x * 256.0f / (float)k.W);
    tx = wrap_i16(tx + f.sky_tx_offset) % 256;
    const bool valid = tx >= 0 && tx < sc.sky_w;
    DevRSpan o;
    o.w[0] = pack_w0(sp.ctop, sp.cbot, SPAN_SKY, sc.sky_has_holes != 0, false);   // a sky bitmap with holes is evaluated in draw order
    o.w[1] = 0;
    o.w[2] = valid ? sc.sky_texel_off + (uint32_t)tx * (uint32_t)sc.sky_h : 0u;
    o.w[3] = f32_bits(valid ? 1.0f : 0.0f);
    o.w[4] = o.w[5] = o.w[6] = o.w[7] = 0;
    return o;
}

// ---- per pixel (all span words are wave-uniform on the GPU) ----------------------------------------------------------

// Texture row of one wall pixel (bitmap_render.rs:256-263): d, uy1, top_y | off_y << 16, h and the prepared reciprocal of d.
DG_HD int32_t wall_texel_row(float d, float r_d, float uy1, uint32_t w5, int32_t h, int32_t y) {
    const int32_t top_y = lo_i16(w5), off_y = hi_i16(w5);
    const float ay = div_prepared((float)(y - top_y), d, r_d);
    int32_t ty = f32_as_i16((float)h + ay * uy1);
    ty = wrap_i16(ty + off_y);
    const int32_t mask = (h & (h - 1)) == 0 ? h - 1 : 0;
    return floor_mod_fast(ty, h, mask, mask ? 0.0f : approx_rcp((float)h));
}
// Texel offset of one wall pixel in the column-major planes from the DevRSpan words.
DG_HD uint32_t wall_texel_offset(uint32_t w1, uint32_t w2, uint32_t w4, uint32_t w5, uint32_t w6, uint32_t w7, int32_t y) {
    const int32_t h = (int32_t)__builtin_fabsf(bits_f32(w6));
    return w2 + (uint32_t)wall_texel_row(bits_f32(w1), bits_f32(w7), bits_f32(w4), w5, h, y);
}

// Texture coordinates of one floor / ceiling pixel and its light factor (visplanes.rs:108-126).
// vy = CFY - y and r_vy = prepare_rcp(vy) are per-row constants.
DG_HD uint32_t flat_texel_offset(const DevFrame &f, uint32_t w1, uint32_t w2, uint32_t w4, uint32_t w5, uint32_t w6,
                                 float vy, float r_vy, float &factor) {
    float wx, wy;
    if (w6 & 0x100u) {
        wx = div_prepared(bits_f32(w4), vy, r_vy);
        wy = div_prepared(bits_f32(w1), vy, r_vy);
    } else {
        wx = bits_f32(w4) / vy;
        wy = bits_f32(w1) / vy;
    }
    float rx = wx * f.cos_a - wy * f.sin_a;
    float ry = wy * f.cos_a + wx * f.sin_a;
    int32_t tx = (f32_as_i16(rx) + f.pos_x_i16) & 63;       // wrapping i16 add, then & 63: the low 6 bits are unaffected by the wrap
    int32_t ty = (f32_as_i16(ry) + f.pos_y_i16) & 63;
    factor = light_factor(bits_f32(w5), f32_as_i16(wx));
    return w2 + (uint32_t)(ty * 64 + tx);               // relative to the texel index plane (the flats follow it in one allocation)
}

// Texture row of a sky pixel (visplanes.rs:68-72): depends on the screen row only; -1 when outside the sky bitmap.
DG_HD int32_t sky_row(const DevScene &sc, const DevConsts &k, int32_t y) {
    int32_t ty = f32_as_i16((float)y * 128.0f * 2.0f / (float)k.H);
    if (ty < 0) ty = wrap_i16(ty + 128);
    ty %= 128;
    return (ty < 0 || ty >= sc.sky_h) ? -1 : ty;
}
// Texel offset of one sky pixel; ~0 when the reference would index outside the sky bitmap (w3: the span's factor, 0 when its column is outside).
DG_HD uint32_t sky_texel_offset(uint32_t w2, uint32_t w3, int32_t row) {
    return (bits_f32(w3) == 0.0f || row < 0) ? 0xffffffffu : w2 + (uint32_t)row;
}

}  // namespace dg
