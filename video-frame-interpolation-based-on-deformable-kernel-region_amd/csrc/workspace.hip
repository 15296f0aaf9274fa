// workspace.hip -- see workspace.h.
#include "workspace.h"

#include <deque>
#include <mutex>
#include <vector>

namespace vfi {

struct WsEntry {
    int device;
    hipStream_t stream;
    void* ptr[WS_SLOTS];
    size_t bytes[WS_SLOTS];
};
static std::mutex g_mutex;
static std::deque<WsEntry> g_entries;             // deque: entries stay put when another stream adds one
static std::vector<std::pair<int, void*>> g_retired;   // (device, pointer) outgrown but possibly still replayed

void* ws_get(hipStream_t stream, WsSlot slot, size_t bytes, bool zero_on_alloc, bool* fresh) {
    if (fresh) *fresh = false;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lock(g_mutex);
    WsEntry* e = nullptr;
    for (auto& x : g_entries)
        if (x.device == dev && x.stream == stream) e = &x;
    if (!e) {
        g_entries.push_back(WsEntry{dev, stream, {}, {}});
        e = &g_entries.back();
    }
    if (e->bytes[slot] >= bytes && e->ptr[slot]) return e->ptr[slot];
    // grow generously (x1.25) so that a sequence of slightly larger frames does not retire a buffer each time
    size_t want = bytes + bytes / 4;
    want = (want + 255) & ~(size_t)255;
    void* p = nullptr;
    if (hipMalloc(&p, want) != hipSuccess) return nullptr;
    if (zero_on_alloc && hipMemsetAsync(p, 0, want, stream) != hipSuccess) {
        (void)hipFree(p);
        return nullptr;
    }
    if (e->ptr[slot]) g_retired.emplace_back(dev, e->ptr[slot]);
    e->ptr[slot] = p;
    e->bytes[slot] = want;
    if (fresh) *fresh = true;
    return p;
}

int device_cu_count() {
    static int cus[64];                             // 0 = not asked yet; racing first calls write the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int v = 0;
        cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    return cus[dev];
}

}  // namespace vfi

using namespace vfi;

extern "C" int vfi_release_workspaces(void) {
    std::lock_guard<std::mutex> lock(g_mutex);
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return VFI_ERR_LAUNCH;
    int rc = VFI_OK;
    auto drop = [&](int dev, void* p) {
        if (!p) return;
        if (hipSetDevice(dev) != hipSuccess || hipDeviceSynchronize() != hipSuccess || hipFree(p) != hipSuccess)
            rc = VFI_ERR_LAUNCH;
    };
    for (auto& e : g_entries)
        for (int s = 0; s < WS_SLOTS; ++s) drop(e.device, e.ptr[s]);
    for (auto& r : g_retired) drop(r.first, r.second);
    g_entries.clear();
    g_retired.clear();
    (void)hipSetDevice(cur);
    return rc;
}
