// emit.h -- one G1 point (signed 30-bit XYZZ) -> an output format: Curve::to_affine (one Fq inversion) + serialisation.
// Host and device: k_emit_points (msm.hip) runs it one thread per point on the GPU; for a lone result that goes to host memory
// capi.hip copies the 224-byte point out and runs the same code on the calling thread (a CPU core inverts in a few
// microseconds what one GPU lane needs ~90 us for, on the critical path of every blocking commit).
// Depends only on the arithmetic headers and the public format constants, so the CPU test suite compiles it with g++
// (tests/host_math.cpp) and checks the bytes against the oracle without a GPU.
#pragma once
#include "../../include/kzg_mi355x.h"
#include "curve30.h"

namespace kzg {

KZG_HD void write_be48(uint8_t *dst, const Fq &canon) {
    for (int i = 0; i < 48; i++) dst[47 - i] = (uint8_t)(canon.v[i >> 2] >> (8 * (i & 3)));
}

KZG_HD bool fq_lexicographically_largest(const Fq &canon) {  // y > (q-1)/2
    // (q-1)/2
    constexpr uint32_t H[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};
    for (int i = 11; i >= 0; i--) {
        if (canon.v[i] > H[i]) return true;
        if (canon.v[i] < H[i]) return false;
    }
    return false;
}

// one point -> `fmt` at `o` (one thread; includes the Fq inversion of to_affine)
KZG_HD void emit_one(const G1Xyzz30 &pt, uint8_t *o, int fmt) {
    const G1Xyzz p = g1_xyzz_from30(pt);
    if (fmt == KZG_G1_JACOBIAN_MONT_144) {
        G1Jacobian j = g1_to_jacobian(p);
        *reinterpret_cast<G1Jacobian *>(o) = j;
        return;
    }
    G1Affine a = g1_to_affine(p);
    if (fmt == KZG_G1_AFFINE_MONT_96) {
        *reinterpret_cast<G1Affine *>(o) = a;
        return;
    }
    Fq x = from_mont(a.x), y = from_mont(a.y);
    if (fmt == KZG_G1_ZCASH_UNCOMPRESSED_96) {
        if (a.is_inf()) {
            for (int k = 0; k < 96; k++) o[k] = 0;
            o[0] = 0x40;
        } else {
            write_be48(o, x);
            write_be48(o + 48, y);
        }
    } else {  // compressed
        if (a.is_inf()) {
            for (int k = 0; k < 48; k++) o[k] = 0;
            o[0] = 0xC0;
        } else {
            write_be48(o, x);
            o[0] |= 0x80;
            if (fq_lexicographically_largest(y)) o[0] |= 0x20;
        }
    }
}

}  // namespace kzg
