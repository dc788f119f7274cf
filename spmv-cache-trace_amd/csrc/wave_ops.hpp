// wave_ops.hpp -- cross-lane fp64 sums for wave64 (gfx950).
//
// A double is moved as two 32-bit halves.  Inside a row of 16 lanes the moves
// are DPP modifiers (no LDS traffic, no extra instruction slot beyond the
// v_mov): quad_perm for lane^1 and lane^2, row_half_mirror / row_mirror to pair
// the two quads / the two half-rows.  Across rows of 16 it is ds_swizzle
// (lane^16 inside a 32-lane half) and a final ds_bpermute for lane^32.
// Every lane must be active when these are called (inactive DPP sources are
// undefined), so callers keep whole waves in the reduction and mask the inputs.
#pragma once

#include <hip/hip_runtime.h>

namespace spmv {

constexpr int kWave = 64;

// dpp_ctrl encodings (LLVM AMDGPU DPP): quad_perm = the four 2-bit selectors.
constexpr int kDppQuadXor1 = 0xB1;       // quad_perm:[1,0,3,2]
constexpr int kDppQuadXor2 = 0x4E;       // quad_perm:[2,3,0,1]
constexpr int kDppRowHalfMirror = 0x141; // lane i <- lane 7-i inside each 8
constexpr int kDppRowMirror = 0x140;     // lane i <- lane 15-i inside each 16

template <int CTRL>
__device__ __forceinline__ double dpp_move(double v)
{
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

// ds_swizzle bit-mask mode: new_lane = ((lane & and) | or) ^ xor within 32 lanes.
template <int XOR>
__device__ __forceinline__ double swizzle_xor(double v)
{
    constexpr int pattern = (XOR << 10) | 0x1F;
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_swizzle(lo, pattern);
    hi = __builtin_amdgcn_ds_swizzle(hi, pattern);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double bpermute_xor32(double v)
{
    const int src = ((int) __lane_id() ^ 32) << 2;
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(src, lo);
    hi = __builtin_amdgcn_ds_bpermute(src, hi);
    return __hiloint2double(hi, lo);
}

// Sum over aligned groups of L consecutive lanes (L = 1,2,4,...,64); every lane
// of a group ends up holding the group's total.  The pairing is a butterfly, so
// the summation order is a balanced tree, not the reference's left-to-right.
template <int L>
__device__ __forceinline__ double group_sum(double v)
{
    static_assert(L == 1 || L == 2 || L == 4 || L == 8 || L == 16 || L == 32 || L == 64,
                  "group size must be a power of two up to the wave size");
    if (L >= 2) v += dpp_move<kDppQuadXor1>(v);
    if (L >= 4) v += dpp_move<kDppQuadXor2>(v);
    if (L >= 8) v += dpp_move<kDppRowHalfMirror>(v);
    if (L >= 16) v += dpp_move<kDppRowMirror>(v);
    if (L >= 32) v += swizzle_xor<16>(v);
    if (L >= 64) v += bpermute_xor32(v);
    return v;
}

// Value of lane (lane - d) for d < 64, 0-filled below lane d is NOT implied:
// callers test lane >= d themselves.
__device__ __forceinline__ double lane_up(double v, int d)
{
    const int src = ((int) __lane_id() - d) << 2;
    int lo = __double2loint(v);
    int hi = __double2hiint(v);
    lo = __builtin_amdgcn_ds_bpermute(src, lo);
    hi = __builtin_amdgcn_ds_bpermute(src, hi);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ int lane_up(int v, int d)
{
    const int src = ((int) __lane_id() - d) << 2;
    return __builtin_amdgcn_ds_bpermute(src, v);
}

__device__ __forceinline__ int lane_down1(int v)
{
    const int src = ((int) __lane_id() + 1) << 2;
    return __builtin_amdgcn_ds_bpermute(src, v);
}

} // namespace spmv
