// Projection, tile binning and the top-K splatter + compositor for gfx950 (MI355X).
//
// Pipeline per ego frame (all on one stream, no host round trip):
//   k_project_count : world point -> view -> NDC (pinhole or fisheye), conservative tile rectangle, wave-aggregated
//                     per-tile counts                                        [HBM-bound: 16 B in, 16 B out per point]
//   k_tile_scan     : exclusive scan of the tile counts (one workgroup)
//   k_tile_fill     : same rectangle, wave-aggregated slot reservation, writes point indices into the tile lists
//   k_splat         : one workgroup per 16x16 tile, one lane per pixel; streams the tile's points through LDS and
//                     keeps each pixel's K smallest (depth-bits, index) keys, then composites front to back and
//                     writes the uint8 frame (and, on request, pytorch3d's idx / zbuf / dists)
//
// The per-pixel key is the 64-bit value (float bits of z) << 32 | point index.  z >= 0 here, so unsigned integer
// order on the key equals the (z, idx) order of pytorch3d's CPU rasterizer; the selected set and its order do not
// depend on the order in which points arrive, which makes the result deterministic although the tile lists are
// filled with atomics.
#include "egoprior_common.h"

namespace ep {

// ------------------------------------------------------------------------------------------------------------------
// camera math (operation order = oracle/oracle.py project_pinhole / project_fisheye)
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void project_point(const egoprior_camera& c, float x, float y, float z, float& ox, float& oy,
                                              float& oz) {
    const float vx = ((x * c.R[0] + y * c.R[3]) + z * c.R[6]) + c.T[0];
    const float vy = ((x * c.R[1] + y * c.R[4]) + z * c.R[7]) + c.T[1];
    const float vz = ((x * c.R[2] + y * c.R[5]) + z * c.R[8]) + c.T[2];
    if (c.model == EGOPRIOR_CAM_PINHOLE) {
        ox = (vx * c.fx + vz * c.px) / vz;
        oy = (vy * c.fy + vz * c.py) / vz;
    } else {
        const float a = vx / vz;
        const float b = vy / vz;
        const float r = sqrtf(a * a + b * b);
        const float th = atanf(r);
        const float t1 = th * th;
        const float t2 = t1 * t1;
        const float t3 = t2 * t1;
        const float t4 = t3 * t1;
        const float t5 = t4 * t1;
        const float t6 = t5 * t1;
        const float th_radial =
            1.0f + (((((c.radial[0] * t1 + c.radial[1] * t2) + c.radial[2] * t3) + c.radial[3] * t4) +
                     c.radial[4] * t5) +
                    c.radial[5] * t6);
        const float th_divr = (fabsf(r) < 1e-10f) ? 1.0f : th / r;
        const float coeff = th_radial * th_divr;
        const float xr = coeff * a;
        const float yr = coeff * b;
        const float n2 = xr * xr + yr * yr;
        const float temp = 2.0f * (xr * c.tangential[0] + yr * c.tangential[1]);
        float ux = (xr + temp * xr) + n2 * c.tangential[0];
        float uy = (yr + temp * yr) + n2 * c.tangential[1];
        const float n4 = n2 * n2;
        ux = ux + (c.thin_prism[0] * n2 + c.thin_prism[1] * n4);
        uy = uy + (c.thin_prism[2] * n2 + c.thin_prism[3] * n4);
        ox = c.fx * ux + c.px;
        oy = c.fx * uy + c.py;
    }
    oz = vz;
}

// ------------------------------------------------------------------------------------------------------------------
// wave-aggregated per-tile counting / slot reservation
// ------------------------------------------------------------------------------------------------------------------
// Every lane walks its own tile rectangle; step k of all lanes is handled together: lanes that name the same tile
// are found with ballots and one lane issues a single atomic for the group.  FILL=false only counts.
template <bool FILL>
__device__ __forceinline__ void bin_point(bool live, int tx0, int tx1, int ty0, int ty1, int tiles_x,
                                          uint32_t* __restrict__ tile_count, const uint32_t* __restrict__ tile_start,
                                          uint32_t* __restrict__ list, uint32_t capacity, uint32_t point_index) {
    const int w = live ? (tx1 - tx0 + 1) : 0;
    const int ntiles = live ? w * (ty1 - ty0 + 1) : 0;
    const int lane = lane_id();
    for (int k = 0;; ++k) {
        const bool active = k < ntiles;
        unsigned long long todo = __ballot(active);
        if (todo == 0ull) break;
        int t = -1;
        if (active) {
            const int dy = k / w;
            const int dx = k - dy * w;
            t = (ty0 + dy) * tiles_x + (tx0 + dx);
        }
        while (todo != 0ull) {
            const int leader = __builtin_ctzll(todo);
            const int tl = __builtin_amdgcn_readlane(t, leader);
            const unsigned long long same = __ballot(active && t == tl);
            uint32_t base = 0;
            if (lane == leader) base = atomicAdd(&tile_count[tl], (uint32_t)__popcll(same));
            if (FILL) {
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                if (active && t == tl) {
                    const uint32_t slot = tile_start[tl] + base + (uint32_t)lanes_below(same);
                    if (slot < capacity) list[slot] = point_index;
                }
            }
            todo &= ~same;
        }
    }
}

__device__ __forceinline__ int64_t live_count(const int32_t* d_n, int64_t cap) {
    if (d_n == nullptr) return cap;
    int64_t v = (int64_t)(*d_n);
    if (v < 0) v = 0;
    return v < cap ? v : cap;
}

// world cloud [a | b] -> NDC records + tile counts.  PROJECT=false: `a` already holds NDC records (b unused) and
// nothing is written to `out`.
template <bool PROJECT>
__global__ void __launch_bounds__(256) k_project_count(egoprior_camera cam, RasterGeom g,
                                                       const egoprior_point* __restrict__ a, int64_t cap_a,
                                                       const int32_t* __restrict__ d_n_a,
                                                       const egoprior_point* __restrict__ b, int64_t cap_b,
                                                       const int32_t* __restrict__ d_n_b,
                                                       egoprior_point* __restrict__ out, int32_t* __restrict__ d_total,
                                                       uint32_t* __restrict__ tile_count, int do_count) {
    const int64_t na = live_count(d_n_a, cap_a);
    const int64_t nb = (b != nullptr) ? live_count(d_n_b, cap_b) : 0;
    const int64_t n = na + nb;
    if (blockIdx.x == 0 && threadIdx.x == 0 && d_total != nullptr) *d_total = (int32_t)n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // whole waves stay in the loop together (the binning uses wave-wide ballots)
    const int64_t n_round = (n + EP_WAVE - 1) / EP_WAVE * EP_WAVE;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool live = i < n;
        float ox = 0.f, oy = 0.f, oz = -1.f;
        if (live) {
            const egoprior_point p = (i < na) ? a[i] : b[i - na];
            if (PROJECT) {
                project_point(cam, p.x, p.y, p.z, ox, oy, oz);
                egoprior_point q;
                q.x = ox;
                q.y = oy;
                q.z = oz;
                q.rgba = p.rgba;
                out[i] = q;
            } else {
                ox = p.x;
                oy = p.y;
                oz = p.z;
            }
        }
        if (do_count) {
            int tx0 = 0, tx1 = 0, ty0 = 0, ty1 = 0;
            const bool hit = live && tile_range(g, ox, oy, oz, tx0, tx1, ty0, ty1);
            bin_point<false>(hit, tx0, tx1, ty0, ty1, g.tiles_x, tile_count, nullptr, nullptr, 0u, 0u);
        }
    }
}

// exclusive scan of tile counts; also zeroes the cursors the fill pass bumps and reports list overflow
__global__ void __launch_bounds__(1024) k_tile_scan(const uint32_t* __restrict__ tile_count,
                                                    uint32_t* __restrict__ tile_start,
                                                    uint32_t* __restrict__ tile_cursor, int n_tiles, uint32_t capacity,
                                                    int32_t* __restrict__ d_status) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_tiles; base += 1024) {
        const int t = base + tid;
        const uint32_t v = (t < n_tiles) ? tile_count[t] : 0u;
        uint32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_sums[wid] = incl;
        __syncthreads();
        uint32_t woff = 0;
        for (int w2 = 0; w2 < wid; ++w2) woff += wave_sums[w2];
        const uint32_t carry = carry_s;
        if (t < n_tiles) {
            tile_start[t] = carry + woff + incl - v;
            tile_cursor[t] = 0u;
        }
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) {
        tile_start[n_tiles] = carry_s;
        if (d_status != nullptr) *d_status = (carry_s > capacity) ? 1 : 0;
    }
}

__global__ void __launch_bounds__(256) k_tile_fill(RasterGeom g, const egoprior_point* __restrict__ ndc, int64_t cap,
                                                   const int32_t* __restrict__ d_n,
                                                   const uint32_t* __restrict__ tile_start,
                                                   uint32_t* __restrict__ tile_cursor, uint32_t* __restrict__ list,
                                                   uint32_t capacity) {
    const int64_t n = live_count(d_n, cap);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t n_round = (n + EP_WAVE - 1) / EP_WAVE * EP_WAVE;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_round; i += stride) {
        const bool live = i < n;
        float ox = 0.f, oy = 0.f, oz = -1.f;
        if (live) {
            const egoprior_point p = ndc[i];
            ox = p.x;
            oy = p.y;
            oz = p.z;
        }
        int tx0 = 0, tx1 = 0, ty0 = 0, ty1 = 0;
        const bool hit = live && tile_range(g, ox, oy, oz, tx0, tx1, ty0, ty1);
        bin_point<true>(hit, tx0, tx1, ty0, ty1, g.tiles_x, tile_cursor, tile_start, list, capacity, (uint32_t)i);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// splat: per-pixel exact top-K + compositor
// ------------------------------------------------------------------------------------------------------------------
struct SplatOut {
    uint8_t* image_u8;
    float* image_f32;
    int32_t* idx;
    float* zbuf;
    float* dists;
};

#define EP_SPLAT_CHUNK 256

// One workgroup = one 16x16 tile, 4 waves, each wave an 8x8 quadrant.  Per lane: K slots kept sorted ascending by
// key in registers (fully unrolled compare-exchange chain).
__global__ void __launch_bounds__(256) k_splat(RasterGeom g, egoprior_raster_params rp,
                                               const egoprior_point* __restrict__ ndc,
                                               const uint32_t* __restrict__ tile_start,
                                               const uint32_t* __restrict__ list, uint32_t capacity,
                                               const int32_t* __restrict__ d_flags_a,
                                               const int32_t* __restrict__ d_flags_b, SplatOut out) {
    __shared__ float4 recs[EP_SPLAT_CHUNK];
    const int tile = blockIdx.x;
    const int tyi = tile / g.tiles_x, txi = tile - tyi * g.tiles_x;
    const int tid = threadIdx.x;
    const int wid = tid >> 6, lane = tid & 63;
    const int xi = txi * EP_TILE + (wid & 1) * 8 + (lane & 7);
    const int yi = tyi * EP_TILE + (wid >> 1) * 8 + (lane >> 3);
    const bool inside = (xi < g.W) && (yi < g.H);
    const float xf = pix_to_ndc(g.W - 1 - xi, g.W, g.H);
    const float yf = pix_to_ndc(g.H - 1 - yi, g.H, g.W);
    const float r2 = g.radius2;

    uint32_t kz[EP_K];  // float bits of z
    uint32_t ki[EP_K];  // point index
    float kd[EP_K];     // squared NDC distance
#pragma unroll
    for (int k = 0; k < EP_K; ++k) {
        kz[k] = 0xFFFFFFFFu;
        ki[k] = 0xFFFFFFFFu;
        kd[k] = -1.0f;
    }

    uint32_t beg = tile_start[tile], end = tile_start[tile + 1];
    if (end > capacity) end = capacity;
    if (beg > end) beg = end;
    for (uint32_t c0 = beg; c0 < end; c0 += EP_SPLAT_CHUNK) {
        const uint32_t nrec = min((uint32_t)EP_SPLAT_CHUNK, end - c0);
        __syncthreads();
        if ((uint32_t)tid < nrec) {
            const uint32_t pidx = list[c0 + tid];
            const egoprior_point p = ndc[pidx];
            float z = p.z;
            if (z == 0.0f) z = 0.0f;  // -0.0 -> +0.0 so that the key order is the float order
            recs[tid] = make_float4(p.x, p.y, z, __uint_as_float(pidx));
        }
        __syncthreads();
        for (uint32_t j = 0; j < nrec; ++j) {
            const float4 r = recs[j];
            const float dx = r.x - xf;
            const float dy = r.y - yf;
            const float d2 = dx * dx + dy * dy;
            uint32_t nz = __float_as_uint(r.z);
            uint32_t ni = __float_as_uint(r.w);
            const unsigned long long nkey = ((unsigned long long)nz << 32) | ni;
            const unsigned long long last = ((unsigned long long)kz[EP_K - 1] << 32) | ki[EP_K - 1];
            const bool take = inside && (d2 < r2) && (nkey < last);
            if (__ballot(take) == 0ull) continue;
            if (take) {
                float nd = d2;
#pragma unroll
                for (int k = 0; k < EP_K; ++k) {
                    const unsigned long long cur = ((unsigned long long)kz[k] << 32) | ki[k];
                    const unsigned long long nk = ((unsigned long long)nz << 32) | ni;
                    const bool sw = nk < cur;
                    const uint32_t tz = kz[k], ti = ki[k];
                    const float td = kd[k];
                    kz[k] = sw ? nz : tz;
                    ki[k] = sw ? ni : ti;
                    kd[k] = sw ? nd : td;
                    nz = sw ? tz : nz;
                    ni = sw ? ti : ni;
                    nd = sw ? td : nd;
                }
            }
        }
    }

    if (!inside) return;

    // ---- compositor (pytorch3d PointsRenderer weights + alpha_composite), then the reference's post-processing
    int scale255 = rp.colors_gt1;
    if (scale255 < 0) {
        scale255 = 0;
        if (d_flags_a != nullptr && *d_flags_a != 0) scale255 = 1;
        if (d_flags_b != nullptr && *d_flags_b != 0) scale255 = 1;
    }
    const int K = rp.K;
    float cum = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
#pragma unroll
    for (int k = 0; k < EP_K; ++k) {
        if (k < K && ki[k] != 0xFFFFFFFFu) {
            const uint32_t rgba = ndc[ki[k]].rgba;
            float fr = (float)(rgba & 0xFFu), fg = (float)((rgba >> 8) & 0xFFu), fb = (float)((rgba >> 16) & 0xFFu);
            if (scale255) {
                fr = fr / 255.0f;
                fg = fg / 255.0f;
                fb = fb / 255.0f;
            }
            const float alpha = 1.0f - kd[k] / rp.radius_sq_w;
            const float wgt = cum * alpha;
            cr += wgt * fr;
            cg += wgt * fg;
            cb += wgt * fb;
            cum = cum * (1.0f - alpha);
        }
    }
    const size_t pix = (size_t)yi * (size_t)g.W + (size_t)xi;
    if (out.image_f32 != nullptr) {
        out.image_f32[pix * 3 + 0] = cr;
        out.image_f32[pix * 3 + 1] = cg;
        out.image_f32[pix * 3 + 2] = cb;
    }
    if (out.image_u8 != nullptr) {
        float r = fminf(fmaxf(cr, 0.0f), 1.0f), gg = fminf(fmaxf(cg, 0.0f), 1.0f), b = fminf(fmaxf(cb, 0.0f), 1.0f);
        if (r < 1e-2f && gg < 1e-2f && b < 1e-2f) {
            r = 0.0f;
            gg = 0.0f;
            b = 0.0f;
        }
        const size_t o = rp.rotate_cw ? ((size_t)xi * (size_t)g.H + (size_t)(g.H - 1 - yi)) : pix;
        out.image_u8[o * 3 + 0] = (uint8_t)(r * 255.0f);
        out.image_u8[o * 3 + 1] = (uint8_t)(gg * 255.0f);
        out.image_u8[o * 3 + 2] = (uint8_t)(b * 255.0f);
    }
    if (out.idx != nullptr) {
#pragma unroll
        for (int k = 0; k < EP_K; ++k) {
            if (k < K) {
                const bool filled = ki[k] != 0xFFFFFFFFu;
                const size_t o = pix * (size_t)K + (size_t)k;
                out.idx[o] = filled ? (int32_t)ki[k] : -1;
                if (out.zbuf != nullptr) out.zbuf[o] = filled ? __uint_as_float(kz[k]) : -1.0f;
                if (out.dists != nullptr) out.dists[o] = filled ? kd[k] : -1.0f;
            }
        }
    }
}

}  // namespace ep

// ------------------------------------------------------------------------------------------------------------------
// host side of the C ABI for this file
// ------------------------------------------------------------------------------------------------------------------
#include "egoprior_host.h"

namespace {

ep::RasterGeom make_geom(int H, int W, float radius) {
    ep::RasterGeom g;
    g.H = H;
    g.W = W;
    g.tiles_x = (W + EP_TILE - 1) / EP_TILE;
    g.tiles_y = (H + EP_TILE - 1) / EP_TILE;
    const double range_x = (W > H) ? (2.0 * W) / H : 2.0;
    const double range_y = (H > W) ? (2.0 * H) / W : 2.0;
    g.half_range_x = (float)(range_x / 2.0);
    g.half_range_y = (float)(range_y / 2.0);
    g.inv_pitch_x = (float)(W / range_x);
    g.inv_pitch_y = (float)(H / range_y);
    // margin: float32 rounding of the centre formula is < 1e-3 px for |coordinate| < 8192 px
    g.rad_px_x = (float)((double)radius * (W / range_x) + 0.05);
    g.rad_px_y = (float)((double)radius * (H / range_y) + 0.05);
    g.radius2 = radius * radius;
    return g;
}

// upper bound on tiles one disc can overlap
int64_t tiles_per_point_bound(const ep::RasterGeom& g) {
    const int64_t bx = (int64_t)std::ceil((2.0 * g.rad_px_x + 1.0) / EP_TILE) + 1;
    const int64_t by = (int64_t)std::ceil((2.0 * g.rad_px_y + 1.0) / EP_TILE) + 1;
    const int64_t bxc = bx < g.tiles_x ? bx : g.tiles_x;
    const int64_t byc = by < g.tiles_y ? by : g.tiles_y;
    return bxc * byc;
}

struct RasterWs {
    uint32_t* tile_count;   // n_tiles
    uint32_t* tile_start;   // n_tiles + 1
    uint32_t* tile_cursor;  // n_tiles
    int32_t* total;         // 1 (+ padding)
    uint32_t* list;         // capacity
    uint64_t capacity;
    size_t header_bytes;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t raster_header_bytes(int n_tiles) { return align_up(sizeof(uint32_t) * (3 * (size_t)n_tiles + 1 + 16), 256); }

bool carve_raster_ws(void* ws, size_t bytes, int n_tiles, RasterWs& r) {
    const size_t hb = raster_header_bytes(n_tiles);
    if (ws == nullptr || bytes < hb + 256) return false;
    uint32_t* base = (uint32_t*)ws;
    r.tile_count = base;
    r.tile_start = base + n_tiles;
    r.tile_cursor = base + 2 * (size_t)n_tiles + 1;
    r.total = (int32_t*)(base + 3 * (size_t)n_tiles + 1);
    r.list = (uint32_t*)((char*)ws + hb);
    uint64_t cap = (bytes - hb) / sizeof(uint32_t);
    if (cap > 0xFFFFFFF0ull) cap = 0xFFFFFFF0ull;
    r.capacity = cap;
    r.header_bytes = hb;
    return true;
}

int grid_for(int64_t n, int block) {
    int64_t b = (n + block - 1) / block;
    if (b < 1) b = 1;
    if (b > 8192) b = 8192;
    return (int)b;
}

int check_raster_args(const egoprior_raster_params* rp, int H, int W, int64_t n) {
    if (rp == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "raster params is NULL");
    if (H <= 0 || W <= 0 || H > 16384 || W > 16384) return ep_host::fail(EGOPRIOR_E_INVALID, "image size out of range");
    if (n < 0 || n > 0x7FFFFFF0ll) return ep_host::fail(EGOPRIOR_E_INVALID, "point count out of range");
    if (rp->K < 1 || rp->K > EGOPRIOR_MAX_K) return ep_host::fail(EGOPRIOR_E_INVALID, "K must be in 1..20");
    if (!(rp->radius > 0.0f) || !(rp->radius_sq_w > 0.0f))
        return ep_host::fail(EGOPRIOR_E_INVALID, "radius must be positive");
    return 0;
}

// count (optionally with projection) must already have run into w.tile_count
int launch_scan_fill_splat(const ep::RasterGeom& g, const egoprior_raster_params* rp, const egoprior_point* d_ndc,
                           int64_t n, const int32_t* d_n, const int32_t* d_flags_a, const int32_t* d_flags_b,
                           const RasterWs& w, ep::SplatOut out, int32_t* d_status, hipStream_t s) {
    const int n_tiles = g.tiles_x * g.tiles_y;
    EP_LAUNCH("k_tile_scan", ep::k_tile_scan, dim3(1), dim3(1024), 0, s, w.tile_count, w.tile_start, w.tile_cursor, n_tiles,
                       (uint32_t)w.capacity, d_status);
    EP_LAUNCH("k_tile_fill", ep::k_tile_fill, dim3(grid_for(n, 256)), dim3(256), 0, s, g, d_ndc, n, d_n, w.tile_start,
                       w.tile_cursor, w.list, (uint32_t)w.capacity);
    EP_LAUNCH("k_splat", ep::k_splat, dim3(n_tiles), dim3(256), 0, s, g, *rp, d_ndc, w.tile_start, w.list,
                       (uint32_t)w.capacity, d_flags_a, d_flags_b, out);
    return ep_host::check_launch("rasterize");
}

}  // namespace

extern "C" size_t egoprior_raster_workspace_bytes(int32_t H, int32_t W, int64_t n, float radius) {
    if (H <= 0 || W <= 0 || n < 0 || !(radius > 0.0f)) return 0;
    const ep::RasterGeom g = make_geom(H, W, radius);
    const int n_tiles = g.tiles_x * g.tiles_y;
    const uint64_t cap = (uint64_t)(n > 0 ? n : 1) * (uint64_t)tiles_per_point_bound(g);
    return raster_header_bytes(n_tiles) + align_up((size_t)cap * sizeof(uint32_t), 256) + 256;
}

extern "C" size_t egoprior_render_workspace_bytes(int32_t H, int32_t W, int64_t n, float radius) {
    const size_t r = egoprior_raster_workspace_bytes(H, W, n, radius);
    if (r == 0) return 0;
    return r + align_up((size_t)(n > 0 ? n : 1) * sizeof(egoprior_point), 256);
}

extern "C" int egoprior_project_points(const egoprior_camera* cam, const egoprior_point* d_a, int64_t n_a,
                                       const int32_t* d_n_a, const egoprior_point* d_b, int64_t n_b,
                                       const int32_t* d_n_b, egoprior_point* d_ndc, void* stream) {
    if (cam == nullptr || d_ndc == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "camera or output is NULL");
    if (n_a < 0 || n_b < 0 || (n_a > 0 && d_a == nullptr) || (n_b > 0 && d_b == nullptr))
        return ep_host::fail(EGOPRIOR_E_INVALID, "bad cloud pointers / sizes");
    if (cam->model != EGOPRIOR_CAM_PINHOLE && cam->model != EGOPRIOR_CAM_FISHEYE)
        return ep_host::fail(EGOPRIOR_E_INVALID, "unknown camera model");
    if (n_a + n_b == 0) return 0;
    ep::RasterGeom g = make_geom(cam->H > 0 ? cam->H : 1, cam->W > 0 ? cam->W : 1, 1e-3f);
    EP_LAUNCH("k_project_count", ep::k_project_count<true>, dim3(grid_for(n_a + n_b, 256)), dim3(256), 0, (hipStream_t)stream,
                       *cam, g, d_a, n_a, d_n_a, n_b > 0 ? d_b : nullptr, n_b, d_n_b, d_ndc, (int32_t*)nullptr,
                       (uint32_t*)nullptr, 0);
    return ep_host::check_launch("project_points");
}

extern "C" int egoprior_rasterize_composite(const egoprior_raster_params* rp, int32_t H, int32_t W,
                                            const egoprior_point* d_ndc, int64_t n, const int32_t* d_n,
                                            const int32_t* d_flags_a, const int32_t* d_flags_b, uint8_t* d_image_u8,
                                            float* d_image_f32, int32_t* d_idx, float* d_zbuf, float* d_dists,
                                            int32_t* d_status, void* d_workspace, size_t workspace_bytes,
                                            void* stream) {
    if (int e = check_raster_args(rp, H, W, n)) return e;
    if (n > 0 && d_ndc == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "d_ndc is NULL");
    if ((d_zbuf != nullptr || d_dists != nullptr) && d_idx == nullptr)
        return ep_host::fail(EGOPRIOR_E_INVALID, "zbuf/dists need idx");
    hipStream_t s = (hipStream_t)stream;
    const ep::RasterGeom g = make_geom(H, W, rp->radius);
    const int n_tiles = g.tiles_x * g.tiles_y;
    RasterWs w;
    if (!carve_raster_ws(d_workspace, workspace_bytes, n_tiles, w))
        return ep_host::fail(EGOPRIOR_E_WORKSPACE, "raster workspace too small");
    if (hipMemsetAsync(w.tile_count, 0, sizeof(uint32_t) * (size_t)n_tiles, s) != hipSuccess)
        return ep_host::fail(EGOPRIOR_E_HIP, "memset failed");
    egoprior_camera cam_unused = {};
    EP_LAUNCH("k_project_count", ep::k_project_count<false>, dim3(grid_for(n, 256)), dim3(256), 0, s, cam_unused, g, d_ndc, n,
                       d_n, (const egoprior_point*)nullptr, (int64_t)0, (const int32_t*)nullptr,
                       (egoprior_point*)nullptr, w.total, w.tile_count, 1);
    ep::SplatOut out = {d_image_u8, d_image_f32, d_idx, d_zbuf, d_dists};
    return launch_scan_fill_splat(g, rp, d_ndc, n, d_n, d_flags_a, d_flags_b, w, out, d_status, s);
}

extern "C" int egoprior_render_frame(const egoprior_camera* cam, const egoprior_raster_params* rp,
                                     const egoprior_point* d_a, int64_t n_a, const int32_t* d_n_a,
                                     const int32_t* d_flags_a, const egoprior_point* d_b, int64_t n_b,
                                     const int32_t* d_n_b, const int32_t* d_flags_b, uint8_t* d_image_u8,
                                     float* d_image_f32, int32_t* d_idx, float* d_zbuf, float* d_dists,
                                     int32_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (cam == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "camera is NULL");
    if (cam->model != EGOPRIOR_CAM_PINHOLE && cam->model != EGOPRIOR_CAM_FISHEYE)
        return ep_host::fail(EGOPRIOR_E_INVALID, "unknown camera model");
    if (n_a < 0 || n_b < 0 || (n_a > 0 && d_a == nullptr) || (n_b > 0 && d_b == nullptr))
        return ep_host::fail(EGOPRIOR_E_INVALID, "bad cloud pointers / sizes");
    const int64_t n = n_a + n_b;
    if (int e = check_raster_args(rp, cam->H, cam->W, n)) return e;
    if ((d_zbuf != nullptr || d_dists != nullptr) && d_idx == nullptr)
        return ep_host::fail(EGOPRIOR_E_INVALID, "zbuf/dists need idx");
    hipStream_t s = (hipStream_t)stream;
    const int H = cam->H, W = cam->W;
    const ep::RasterGeom g = make_geom(H, W, rp->radius);
    const int n_tiles = g.tiles_x * g.tiles_y;
    const size_t ndc_bytes = align_up((size_t)(n > 0 ? n : 1) * sizeof(egoprior_point), 256);
    if (d_workspace == nullptr || workspace_bytes < ndc_bytes)
        return ep_host::fail(EGOPRIOR_E_WORKSPACE, "render workspace too small");
    egoprior_point* d_ndc = (egoprior_point*)d_workspace;
    RasterWs w;
    if (!carve_raster_ws((char*)d_workspace + ndc_bytes, workspace_bytes - ndc_bytes, n_tiles, w))
        return ep_host::fail(EGOPRIOR_E_WORKSPACE, "render workspace too small");
    if (hipMemsetAsync(w.tile_count, 0, sizeof(uint32_t) * (size_t)n_tiles, s) != hipSuccess)
        return ep_host::fail(EGOPRIOR_E_HIP, "memset failed");
    EP_LAUNCH("k_project_count", ep::k_project_count<true>, dim3(grid_for(n, 256)), dim3(256), 0, s, *cam, g, d_a, n_a, d_n_a,
                       n_b > 0 ? d_b : nullptr, n_b, d_n_b, d_ndc, w.total, w.tile_count, 1);
    ep::SplatOut out = {d_image_u8, d_image_f32, d_idx, d_zbuf, d_dists};
    return launch_scan_fill_splat(g, rp, d_ndc, n, w.total, d_flags_a, d_flags_b, w, out, d_status, s);
}
