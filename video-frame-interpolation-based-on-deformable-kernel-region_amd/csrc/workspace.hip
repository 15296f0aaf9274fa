// workspace.hip -- see workspace.h.
#include "workspace.h"

#include <algorithm>
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

void* ws_get(hipStream_t stream, WsSlot slot, size_t bytes, bool zero_on_alloc, bool* fresh, size_t* capacity) {
    if (fresh) *fresh = false;
    if (capacity) *capacity = 0;
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
    if (e->bytes[slot] >= bytes && e->ptr[slot]) {
        if (capacity) *capacity = e->bytes[slot];
        return e->ptr[slot];
    }
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
    if (capacity) *capacity = want;
    return p;
}

bool ws_get_group(hipStream_t stream, const WsSlot* slots, const size_t* bytes, const bool* zero, int n, void** ptrs) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(g_mutex);
    WsEntry* e = nullptr;
    for (auto& x : g_entries)
        if (x.device == dev && x.stream == stream) e = &x;
    if (!e) {
        g_entries.push_back(WsEntry{dev, stream, {}, {}});
        e = &g_entries.back();
    }
    bool fits = true;
    for (int i = 0; i < n; ++i) fits = fits && e->ptr[slots[i]] && e->bytes[slots[i]] >= bytes[i];
    if (!fits) {
        void* fresh[WS_SLOTS] = {};
        size_t want[WS_SLOTS] = {};
        for (int i = 0; i < n; ++i) {
            want[i] = std::max(bytes[i] + bytes[i] / 4, e->bytes[slots[i]]);        // grow generously, never shrink
            want[i] = (want[i] + 255) & ~(size_t)255;
            if (hipMalloc(&fresh[i], want[i]) != hipSuccess || (zero[i] && hipMemsetAsync(fresh[i], 0, want[i], stream) != hipSuccess)) {
                for (int k = 0; k <= i; ++k)
                    if (fresh[k]) (void)hipFree(fresh[k]);
                return false;
            }
        }
        for (int i = 0; i < n; ++i) {
            if (e->ptr[slots[i]]) g_retired.emplace_back(dev, e->ptr[slots[i]]);
            e->ptr[slots[i]] = fresh[i];
            e->bytes[slots[i]] = want[i];
        }
    }
    for (int i = 0; i < n; ++i) ptrs[i] = e->ptr[slots[i]];
    return true;
}

// ---- deterministic image gradients (vfi_common.h)
// largest |element| of a [batch, channel, h, w] tensor into hdr[slot] (non-negative floats order like their bits); a NaN or an
// infinity raises hdr[1]
__global__ __launch_bounds__(256) void gradacc_max(const float* __restrict__ g, int channel, int h, int w, vfi_strides sg, int64_t n,
                                                   int* __restrict__ hdr, int slot, int cells_log2) {
    int m = 0;
    bool bad = false;
    // a block per image row at a time: one set of divisions per row, consecutive lanes on consecutive elements
    const int rows = (int)(n / w);
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int y = row % h, bc = row / h;
        const int c = bc % channel, b = bc / channel;
        const float* p = g + (int64_t)b * sg.b + (int64_t)c * sg.c + (int64_t)y * sg.h;
        for (int x = threadIdx.x; x < w; x += 256) {
            const int bits = __float_as_int(fabsf(p[x]));
            bad = bad || bits >= 0x7f800000;                // infinity or NaN
            m = max(m, bits >= 0x7f800000 ? 0 : bits);
        }
    }
    // one atomic per block, and only from a block that raises the maximum: thousands of atomics on one word serialise
    // in L2 (measured: 0.2 ms per pass when every wave issued its own)
    __shared__ int wm[4], wbad[4];
    m = wave_max_i32(m);
    const bool anybad = __builtin_amdgcn_ballot_w64(bad) != 0ull;       // (all lanes active here)
    if ((threadIdx.x & 63) == 0) {
        wm[threadIdx.x >> 6] = m;
        wbad[threadIdx.x >> 6] = anybad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
        if (m > __atomic_load_n(&hdr[slot], __ATOMIC_RELAXED)) atomicMax(&hdr[slot], m);
        if ((wbad[0] | wbad[1] | wbad[2] | wbad[3]) && !__atomic_load_n(&hdr[1], __ATOMIC_RELAXED)) atomicOr(&hdr[1], 1);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && slot == 0) hdr[3] = cells_log2;
}

__global__ __launch_bounds__(256) void gradacc_convert(const unsigned long long* __restrict__ acc, const int* __restrict__ hdr,
                                                       float* __restrict__ g1, int channel, int h, int w, vfi_strides s1, int64_t n) {
    if (hdr[1] != 0) return;                                // the call scattered with fp32 atomics: nothing in the scratch
    const int k = gradacc_exponent(hdr);
    const int rows = (int)(n / w);
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int y = row % h, bc = row / h;
        const int c = bc % channel, b = bc / channel;
        const unsigned long long* a = acc + (int64_t)row * w;
        float* cells = g1 + (int64_t)b * s1.b + (int64_t)c * s1.c + (int64_t)y * s1.h;
        for (int x = threadIdx.x; x < w; x += 256) {
            const long long sum = (long long)a[x];
            if (sum != 0) cells[x] += (float)ldexp((double)sum, -k);    // exact integer sum -> float once
        }
    }
}

int gradacc_begin(hipStream_t st, const float* gout, int batch, int channel, int h, int w, vfi_strides sg,
                  const float* weights, int wchannel, vfi_strides sw, unsigned long long** acc, int** hdr,
                  int nflags, int** flags) {
    const int64_t n = (int64_t)batch * channel * h * w;
    const size_t bytes = (size_t)n * 8 + 256 + (size_t)nflags * 4;
    void* p = ws_get(st, WS_GRADACC, bytes, false, nullptr);
    if (!p) return VFI_ERR_LAUNCH;
    if (hipMemsetAsync(p, 0, bytes, st) != hipSuccess) return VFI_ERR_LAUNCH;
    *hdr = static_cast<int*>(p);
    *acc = reinterpret_cast<unsigned long long*>(static_cast<char*>(p) + 256);
    if (flags) *flags = reinterpret_cast<int*>(static_cast<char*>(p) + 256 + (size_t)n * 8);
    // addends one cell can receive: every pixel of the frame, with every one of its taps -- border clamping can put all fs x fs
    // taps of a pixel (the four corners of a bilinear sample) on one cell (h w alone left (9/4) h w addends of
    // nearly 2^(62 - L) possible on a corner cell at fs = 6)
    const int64_t taps = wchannel > 4 ? wchannel : 4;
    int cells_log2 = 0;
    while (((int64_t)1 << cells_log2) < (int64_t)h * w * taps) ++cells_log2;
    const int64_t rows = n / w;
    const int blocks = (int)(rows < 2048 ? rows : 2048);
    hipLaunchKernelGGL(gradacc_max, dim3(blocks), dim3(256), 0, st, gout, channel, h, w, sg, n, *hdr, 0, cells_log2);
    if (weights) {
        const int64_t nw = (int64_t)batch * wchannel * h * w;
        const int64_t wrows = nw / w;
        const int wblocks = (int)(wrows < 2048 ? wrows : 2048);
        hipLaunchKernelGGL(gradacc_max, dim3(wblocks), dim3(256), 0, st, weights, wchannel, h, w, sw, nw, *hdr, 2, cells_log2);
    }
    return launch_status();
}

int gradacc_finish(hipStream_t st, const unsigned long long* acc, const int* hdr, float* g1, int batch, int channel, int h, int w,
                   vfi_strides s1) {
    const int64_t n = (int64_t)batch * channel * h * w;
    const int64_t rows = n / w;
    const int blocks = (int)(rows < 8192 ? rows : 8192);
    hipLaunchKernelGGL(gradacc_convert, dim3(blocks), dim3(256), 0, st, acc, hdr, g1, channel, h, w, s1, n);
    return launch_status();
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

int fi_channel_groups(int ntiles, int channel, double prologue) {
    const int slots = device_cu_count() * 2;
    int best = 1;
    double best_cost = 0.0;
    for (int g = 1; g <= 8 && g <= channel; ++g) {
        const double r = (double)ntiles * g / slots;
        const double cost = (channel + prologue * g) * ((r + 0.5) / r) * (1.0 + 0.25 / r);
        if (g == 1 || cost < best_cost) { best_cost = cost; best = g; }
    }
    return best;
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
