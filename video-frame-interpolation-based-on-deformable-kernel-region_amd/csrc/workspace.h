// workspace.h -- device scratch owned by the library, one set of named buffers per (device, stream).
//
// Rules (they make the buffers safe for captured HIP graphs and for callers on several streams):
//  * a buffer is never shrunk and never freed behind the caller's back: when a call needs more than the
//    buffer holds, a NEW buffer is allocated and the old one is retired, not freed -- a graph captured
//    earlier keeps replaying on the old one, which stays valid and self-consistent;
//  * every initialisation is a hipMemsetAsync on the stream the kernels are launched on, so it is ordered
//    with them (torch side streams do not synchronise with the null stream);
//  * retired and live buffers are freed only by vfi_release_workspaces() (include/vfi_hip.h), which the
//    caller invokes when no launch or graph uses them any more.
// hipMalloc is not legal inside a stream capture: reserve (vfi_projection_reserve) or make one warm-up
// call before capturing.
#pragma once
#include "vfi_common.h"

namespace vfi {

enum WsSlot {
    WS_PROJ_WORDS = 0,      // projection: header, super-tile and block tables, per-tile hole counts
    WS_PROJ_BITS,           // projection: row- and column-packed bitmaps of "count != 0"
    WS_PROJ_PLANES,         // projection fallback: three dense fp32 planes, zero between calls
    WS_PROJ_UPFLOW,         // *_forward_up4: the upsampled full-resolution flow (pure scratch)
    WS_MINDEPTH,            // MinDepthFlowProjection: 64-bit keys + bitmaps
    WS_GRADACC,             // backward passes: 64-bit fixed-point image-gradient sums (vfi_common.h: gradacc_*)
    WS_SLOTS
};

// Returns a device buffer of at least `bytes` bytes for (current device, stream, slot), or nullptr.
// *fresh is set when the buffer was (re)allocated by this call; with zero_on_alloc the new buffer has
// been zero-filled by a hipMemsetAsync on `stream` before it is returned.
// *capacity, when asked for, is the size of the returned buffer (at least `bytes`).
void* ws_get(hipStream_t stream, WsSlot slot, size_t bytes, bool zero_on_alloc, bool* fresh, size_t* capacity = nullptr);

// Buffers that carry state for each other (the projection's header and records say what its scratch planes hold) are
// (re)allocated TOGETHER: if any of the n slots is too small for its bytes[i], all n get new buffers (zero-filled on
// `stream` where zero[i]) and the old ones are retired as a set -- a graph captured earlier keeps its own consistent set.
// false on allocation failure.
bool ws_get_group(hipStream_t stream, const WsSlot* slots, const size_t* bytes, const bool* zero, int n, void** ptrs);

}  // namespace vfi
