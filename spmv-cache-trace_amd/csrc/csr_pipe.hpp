// csr_pipe.hpp -- EXPERIMENT (libspmv_hip_experiments.so only, SPMV_HIP_PIPE=1): the narrow-tile path of csr_wavetile_kernel as
// a persistent, software-pipelined kernel.  Every wave walks through the tiles w, w + G, w + 2G, ... (G = waves in the
// grid: the whole grid sweeps the matrix as one front) and requests the NEXT tile's streams before it gathers x for and
// sums the current one, so that a wave always has a tile's worth of loads in flight.  Question it answers: is the gap
// between the pure load mix of a queen-like launch (504 us, tools/probes/tile_stream.hip) and the launch itself
// (620-657 us) the waves' idle time between their load phases?  Only plain narrow fast tiles are multiplied (the others
// are skipped: wrong y by design) -- a timing experiment.
#pragma once

#include "csr_segwin.hpp"

namespace spmv {

template <int TILE>
__global__ __launch_bounds__(256) void csr_pipe_kernel(
    int ntiles, const int4 * __restrict__ desc, const int32_t * __restrict__ p, const uint16_t * __restrict__ j16,
    const double * __restrict__ a, const double * __restrict__ x, const double * y_in, double * y, int cols)
{
    constexpr int QUADS = TILE / 256;
    __shared__ __attribute__((aligned(16))) double prod_all[4][TILE + 4];
    const int wave = __builtin_amdgcn_readfirstlane((int) threadIdx.x >> 6);
    const int lane = (int) __lane_id();
    double * prod = prod_all[wave];
    const int stride = (int) gridDim.x * 4;
    int t = (int) blockIdx.x * 4 + wave;
    SwTile<QUADS> cur, nxt;
    int cbase_cur = 0, cbase_nxt = 0, meta_cur = 0, meta_nxt = 0;
    if (t < ntiles) {
        meta_cur = __builtin_amdgcn_readfirstlane(desc[t].z);
        cbase_cur = __builtin_amdgcn_readfirstlane(desc[t].w);
        sw_load_tile<TILE>(cur, t, desc, p, j16, a, y_in, lane);
    }
    while (t < ntiles) {
        const int tn = t + stride;
        if (tn < ntiles) {
            meta_nxt = __builtin_amdgcn_readfirstlane(desc[tn].z);
            cbase_nxt = __builtin_amdgcn_readfirstlane(desc[tn].w);
            sw_load_tile<TILE>(nxt, tn, desc, p, j16, a, y_in, lane);
        }
        if ((meta_cur & kTileMetaFast) && (meta_cur & kTileMetaNarrow) && !(meta_cur & kTileMetaShifted)) {
            const char * xb = reinterpret_cast<const char *>(x + cbase_cur);
            const unsigned limit = (unsigned) (cols - 1 - cbase_cur);
#pragma unroll
            for (int q = 0; q < QUADS; ++q) {
                const int o = 256 * q + 4 * lane;
                if (o <= cur.last) {
                    const unsigned c0 = min(cur.cx[q] & 0xFFFFu, limit), c1 = min(cur.cx[q] >> 16, limit);
                    const unsigned c2 = min(cur.cy[q] & 0xFFFFu, limit), c3 = min(cur.cy[q] >> 16, limit);
                    const double q0 = cur.va[q].x * *reinterpret_cast<const double *>(xb + (c0 << 3));
                    const double q1 = cur.va[q].y * *reinterpret_cast<const double *>(xb + (c1 << 3));
                    const double q2 = cur.vb[q].x * *reinterpret_cast<const double *>(xb + (c2 << 3));
                    const double q3 = cur.vb[q].y * *reinterpret_cast<const double *>(xb + (c3 << 3));
                    v2d * dst = reinterpret_cast<v2d *>(prod + o);
                    dst[0] = v2d{q0, q1};
                    dst[1] = v2d{q2, q3};
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int sub = lane >> cur.lanes_log2;
            const int part = lane & ((1 << cur.lanes_log2) - 1);
            const int s = cur.ps - cur.kb, e_row = cur.pe - cur.kb;
            double z;
            if (cur.lanes_log2 == 0) {
                z = tile_row_sum<1>(prod, s, e_row, 0, cur.maxlen);
            } else {
                const int trips = (cur.maxlen + (1 << cur.lanes_log2) - 1) >> cur.lanes_log2;
                switch (cur.lanes_log2) {
                case 1: z = tile_row_sum<2>(prod, s, e_row, part, trips); break;
                case 2: z = tile_row_sum<4>(prod, s, e_row, part, trips); break;
                case 3: z = tile_row_sum<8>(prod, s, e_row, part, trips); break;
                case 4: z = tile_row_sum<16>(prod, s, e_row, part, trips); break;
                case 5: z = tile_row_sum<32>(prod, s, e_row, part, trips); break;
                default: z = tile_row_sum<64>(prod, s, e_row, part, trips); break;
                }
            }
            if (sub < cur.nrows && part == 0)
                y[cur.r0 + sub] = cur.yv + z;
            if (cur.second) {
                const double zB = tile_row_sum<1>(prod, cur.psB - cur.kb, cur.peB - cur.kb, 0, cur.maxlen);
                if (lane + kWave < cur.nrows)
                    y[cur.r0 + lane + kWave] = cur.yvB + zB;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        cur = nxt;
        meta_cur = meta_nxt;
        cbase_cur = cbase_nxt;
        t = tn;
    }
}

} // namespace spmv
