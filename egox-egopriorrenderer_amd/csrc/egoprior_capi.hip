// Library-level pieces of the C ABI: error text and version.
#include "egoprior_host.h"

namespace ep_host {
char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}
}  // namespace ep_host

extern "C" const char* egoprior_last_error(void) { return ep_host::error_buffer(); }
extern "C" int egoprior_abi_version(void) { return EGOPRIOR_ABI_VERSION; }

// ---- per-kernel timing ------------------------------------------------------------------------------------------
namespace ep_host {
Prof& prof() {
    static thread_local Prof p;
    return p;
}
}  // namespace ep_host

extern "C" int egoprior_profile_begin(void) {
    ep_host::Prof& p = ep_host::prof();
    for (auto& r : p.recs) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    p.recs.clear();
    p.on = true;
    return 0;
}

extern "C" int egoprior_profile_end(char* names, size_t names_bytes, float* total_ms, int32_t* calls,
                                    int32_t max_kernels, int32_t* n_kernels) {
    ep_host::Prof& p = ep_host::prof();
    p.on = false;
    if (names == nullptr || total_ms == nullptr || calls == nullptr || n_kernels == nullptr || max_kernels <= 0)
        return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    std::vector<const char*> uniq;
    int rc = 0;
    for (auto& r : p.recs) {
        float ms = 0.0f;
        if (hipEventSynchronize(r.stop) != hipSuccess || hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess)
            rc = EGOPRIOR_E_HIP;
        int k = -1;
        for (size_t i = 0; i < uniq.size(); ++i)
            if (std::strcmp(uniq[i], r.name) == 0) k = (int)i;
        if (k < 0 && (int)uniq.size() < max_kernels) {
            uniq.push_back(r.name);
            k = (int)uniq.size() - 1;
            total_ms[k] = 0.0f;
            calls[k] = 0;
        }
        if (k >= 0) {
            total_ms[k] += ms;
            calls[k] += 1;
        }
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    p.recs.clear();
    size_t off = 0;
    for (size_t i = 0; i < uniq.size(); ++i) {
        const size_t len = std::strlen(uniq[i]);
        if (off + len + 2 > names_bytes) break;
        std::memcpy(names + off, uniq[i], len);
        off += len;
        names[off++] = '\n';
    }
    if (names_bytes > 0) names[off < names_bytes ? off : names_bytes - 1] = '\0';
    *n_kernels = (int32_t)uniq.size();
    if (rc != 0) return ep_host::fail(rc, "event timing failed");
    return 0;
}

// ---- options ------------------------------------------------------------------------------------------------------
namespace {
struct Opt {
    const char* name;
    int value;
};
Opt g_opts[] = {
    {"splat_variant", 0},  // 0: sweep, 448-record chunks; 1: register insertion only; 2: sweep, 64-record chunks;
                           // 3: sweep, 1024-record chunks; 4: sweep, 384-record chunks at four waves per SIMD;
                           // 5: sweep, 512-record chunks
    {"fill_pieces", 1},    // 1: fill pass without global atomics where possible (k_tile_fill_pieces), 0: always k_tile_fill
    {"bin_lds", 1},        // 1: tile counts aggregated per workgroup in LDS, 0: straight global atomics
    {"slices", 1},         // 1: tiles with very long lists are split over several waves, 0: one wave per tile always
    {"scan_threads", 256},  // workgroup size of the tile scan: 256 or 1024
    {"scan_once", 1},       // 1: grids of up to 8192 tiles are scanned from registers after one round of loads, 0: always the looping scan
    {"tail_slots", 0},      // > 0: the number of splat waves the GPU holds at once (3072 on MI355X: 256 compute units x
                            // 12); when the tiles with real work would leave the last round of a launch mostly empty,
                            // the tiles of that round are cut into three slices each.  Off by default: it shortens a
                            // lone launch for some views and costs the tile scan more than that on average
};
}  // namespace

namespace ep_host {
int option(const char* name) {
    for (auto& o : g_opts)
        if (std::strcmp(o.name, name) == 0) return o.value;
    return -1;
}
}  // namespace ep_host

extern "C" int egoprior_set_option(const char* name, int32_t value) {
    if (name == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "option name is NULL");
    for (auto& o : g_opts)
        if (std::strcmp(o.name, name) == 0) {
            o.value = value;
            return 0;
        }
    return ep_host::fail(EGOPRIOR_E_INVALID, "unknown option");
}

extern "C" int32_t egoprior_get_option(const char* name) { return name == nullptr ? -1 : ep_host::option(name); }
