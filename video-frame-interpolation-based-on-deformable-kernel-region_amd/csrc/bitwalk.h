// bitwalk.h -- nearest set bit along a bitmap line; shared by the hole fillers of projection.hip and mindepth.hip.
#pragma once
#include "vfi_common.h"

namespace vfi {

// first set bit strictly beyond position p0 in direction dir (-1 / +1) of a bitmap line of `len`
// bits (32 per word); -1 if none.  Words are fetched four at a time (independent loads).
__device__ __forceinline__ int proj_bit_walk(const int* __restrict__ line, int p0, int len, int dir) {
    const int nw = (len + 31) >> 5;
    int wi = p0 >> 5;
    unsigned word = (unsigned)line[wi];
    if (dir < 0) {
        word &= (1u << (p0 & 31)) - 1u;
        if (word) return wi * 32 + 31 - __clz(word);
        for (wi -= 1; wi >= 0; wi -= 4) {
            unsigned q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = (wi - k >= 0) ? (unsigned)line[wi - k] : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (q[k]) return (wi - k) * 32 + 31 - __clz(q[k]);
        }
    } else {
        word &= ((p0 & 31) == 31) ? 0u : ~((2u << (p0 & 31)) - 1u);
        if (word) return wi * 32 + __ffs((int)word) - 1;
        for (wi += 1; wi < nw; wi += 4) {
            unsigned q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) q[k] = (wi + k < nw) ? (unsigned)line[wi + k] : 0u;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (q[k]) return (wi + k) * 32 + __ffs((int)q[k]) - 1;
        }
    }
    return -1;
}

}  // namespace vfi
