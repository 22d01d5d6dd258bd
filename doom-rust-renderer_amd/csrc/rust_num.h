// rust_num.h — numeric semantics of the reference's release-mode Rust, shared by host C++ and HIP device code.
//
// The reference renderer (freewilll/doom-rust-renderer, src/renderer/*) is all f32 + `as` casts:
//   * float -> int `as` truncates toward zero, saturates, NaN -> 0        (e.g. bitmap_render.rs:243,251,257)
//   * int -> narrower int `as` wraps                                        (e.g. segs.rs:149,195,202)
//   * i16 `+` `*` wrap in release builds (README.md:15-36 runs `cargo run -r`)
//   * f32 ops are IEEE binary32, never fused: build every TU with -ffp-contract=off, never -ffast-math.
// On the GPU v_cvt_i32_f32 already truncates, saturates and maps NaN to 0, so the i32 cast is one
// instruction and the narrower casts add one v_med3_i32.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define DG_HD __host__ __device__ __forceinline__
#else
#define DG_HD inline
#endif

namespace dg {

DG_HD int32_t f32_as_i32(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
#else
    if (f != f) return 0;
    if (f <= -2147483648.0f) return INT32_MIN;
    if (f >= 2147483648.0f) return INT32_MAX;
    return (int32_t)f;
#endif
}

DG_HD int32_t clamp_i32(int32_t v, int32_t lo, int32_t hi) { return v < lo ? lo : (v > hi ? hi : v); }

// `f as i16`, returned widened to i32 (value in [-32768, 32767])
DG_HD int32_t f32_as_i16(float f) { return clamp_i32(f32_as_i32(f), -32768, 32767); }
// `f as u8`, returned widened
DG_HD int32_t f32_as_u8(float f) { return clamp_i32(f32_as_i32(f), 0, 255); }
// `v as i16` for an i32 (wrap)
DG_HD int32_t wrap_i16(int32_t v) { return (int32_t)(int16_t)(uint16_t)(uint32_t)v; }

// The reference's texture-coordinate fix-up (bitmap_render.rs:244-248 / 259-263), for an i16 `t` and a
// bitmap dimension 0 < n <= 32767, all in wrapping i16 arithmetic:
//     if t < 0 { t += n * (1 - t / n) }   t %= n
// `t / n` truncates, so for t < 0 the bracket is 1 + floor(|t| / n) and t + n*(..) = n - (|t| mod n),
// which lies in [1, n]; being in range, the wrapped intermediate products cannot change it.  The
// final `%` maps n to 0.  Net effect: the non-negative (floor) modulus of t by n.
DG_HD int32_t floor_mod_i16(int32_t t, int32_t n) {
    int32_t r = t % n;
    return r < 0 ? r + n : r;
}

}  // namespace dg
