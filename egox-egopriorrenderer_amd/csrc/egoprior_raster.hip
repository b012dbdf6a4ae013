// Projection, tile binning and the top-K splatter + compositor for gfx950 (MI355X).
//
// Pipeline per ego frame (all on one stream, no host round trip):
//   k_project_count : world point -> view -> NDC (pinhole or fisheye), conservative tile rectangle, wave-aggregated
//                     per-tile counts                                        [HBM-bound: 16 B in, 16 B out per point]
//   k_tile_scan     : exclusive scan of the tile counts (one workgroup)
//   k_tile_fill     : same rectangle, slots reserved per run of lanes, writes the tile-list entries (8 bytes: point
//                     index and depth bits; the splat kernels fetch NDC x, y from the projected records)
//   k_splat_sweep   : one wave per 8x8 tile, one lane per pixel; buckets the tile's points by key in LDS, sweeps them
//                     front to back keeping each pixel's K smallest (depth-bits, index) keys, then composites and
//                     writes the uint8 frame (and, on request, pytorch3d's idx / zbuf / dists)
//
// The per-pixel key is the 64-bit value (float bits of z) << 32 | point index.  z >= 0 here, so unsigned integer
// order on the key equals the (z, idx) order of pytorch3d's CPU rasterizer; the selected set and its order do not
// depend on the order in which points arrive, which makes the result deterministic although the tile lists are
// filled with atomics.
#include "egoprior_common.h"

namespace ep {

// A tile-list entry: point index and depth bits, together the 64-bit key (-0.0 -> +0.0, so that key order == float order).
typedef uint2 ListEntry;
__device__ __forceinline__ ListEntry make_entry(uint32_t idx, float z) {
    return make_uint2(idx, __float_as_uint(z == 0.0f ? 0.0f : z));
}

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
// per-tile counting / slot reservation
// ------------------------------------------------------------------------------------------------------------------
// Run grouping.  Neighbouring lanes hold neighbouring points of an exo frame, and those mostly cover the same tiles:
// lanes are grouped into runs of consecutive lanes with identical tile rectangles, and only the first lane of a run
// touches the counters, adding the run's length.  Without it the same-address atomics of a wave serialise.  Must be
// called by whole waves; lanes without a point pass live = false.
// 2 or 3: every rectangle of the launch fits into that many tiles per axis; 0: larger ones can occur
__device__ __forceinline__ int straight_span(const RasterGeom& g) {
    const int m = g.span_x > g.span_y ? g.span_x : g.span_y;
    return m <= 2 ? 2 : (m == 3 ? 3 : 0);
}

struct LaneRun {
    bool head;         // first lane of its run
    uint32_t len;      // run length (valid on the head)
    int head_lane;     // lane's run head (valid on live lanes)
};
__device__ __forceinline__ LaneRun lane_runs(bool live, int tx0, int tx1, int ty0, int ty1) {
    const int lane = lane_id();
    const uint32_t rx = live ? ((uint32_t)tx0 | ((uint32_t)tx1 << 16)) : 0xFFFFFFFFu;
    const uint32_t ry = (uint32_t)ty0 | ((uint32_t)ty1 << 16);
    const uint32_t prx = (uint32_t)__shfl_up((int)rx, 1), pry = (uint32_t)__shfl_up((int)ry, 1);
    LaneRun r;
    r.head = live && (lane == 0 || rx != prx || ry != pry);
    const unsigned long long heads = __ballot(r.head);
    const unsigned long long stops = heads | __ballot(!live);  // a run ends before the next head or empty lane
    const unsigned long long above = (lane == 63) ? 0ull : (stops >> (lane + 1));
    r.len = (above != 0ull) ? (uint32_t)__builtin_ctzll(above) + 1u : (uint32_t)(64 - lane);
    r.head_lane = live ? 63 - __builtin_clzll(heads & ((2ull << lane) - 1ull)) : lane;
    return r;
}

// The heads of the runs add their run's length to every tile of the rectangle.  When no rectangle of the launch is
// larger than 2 x 2 or 3 x 3 tiles (`span`, the same for all threads), the candidates are visited in straight-line
// code instead of per-lane loops.
template <int SPAN>
__device__ __forceinline__ void count_run_lds_span(const LaneRun& r, int tx0, int tx1, int ty0, int ty1, int tiles_x,
                                                   uint32_t* hist) {
    const int t00 = ty0 * tiles_x + tx0;
#pragma unroll
    for (int q = 0; q < SPAN * SPAN; ++q) {
        const int dx = q % SPAN, dy = q / SPAN;
        const int t = t00 + dy * tiles_x + dx;
        if (r.head && tx0 + dx <= tx1 && ty0 + dy <= ty1) atomicAdd(&hist[t >> 1], r.len << ((t & 1) * 16));
    }
}
// Counting into a workgroup's LDS histogram: two 16-bit counters per word (a workgroup adds at most EP_BIN_PPB to a tile).
__device__ __forceinline__ void count_run_lds(const LaneRun& r, int span, int tx0, int tx1, int ty0, int ty1,
                                              int tiles_x, uint32_t* hist) {
    if (span == 2) return count_run_lds_span<2>(r, tx0, tx1, ty0, ty1, tiles_x, hist);
    if (span == 3) return count_run_lds_span<3>(r, tx0, tx1, ty0, ty1, tiles_x, hist);
    if (!r.head) return;
    for (int ty = ty0; ty <= ty1; ++ty) {
        for (int tx = tx0; tx <= tx1; ++tx) {
            const int t = ty * tiles_x + tx;
            atomicAdd(&hist[t >> 1], r.len << ((t & 1) * 16));
        }
    }
}

// Counting straight into the global per-tile counters: one atomic per run and tile.
__device__ __forceinline__ void count_run_global(const LaneRun& r, int tx0, int tx1, int ty0, int ty1, int tiles_x,
                                                 uint32_t* __restrict__ tile_count) {
    if (!r.head) return;
    for (int ty = ty0; ty <= ty1; ++ty)
        for (int tx = tx0; tx <= tx1; ++tx) atomicAdd(&tile_count[ty * tiles_x + tx], r.len);
}

// The same run grouping straight on the global cursors (no LDS stage): one returning global atomic per run and tile.
__device__ __forceinline__ void bin_point_runs_global(bool live, int tx0, int tx1, int ty0, int ty1, int tiles_x,
                                                      uint32_t* __restrict__ cursor,
                                                      const uint32_t* __restrict__ tile_start,
                                                      ListEntry* __restrict__ list, uint32_t capacity,
                                                      ListEntry entry) {
    const int lane = lane_id();
    const LaneRun r = lane_runs(live, tx0, tx1, ty0, ty1);
    if (!live) return;
    for (int ty = ty0; ty <= ty1; ++ty) {
        for (int tx = tx0; tx <= tx1; ++tx) {
            const int t = ty * tiles_x + tx;
            uint32_t slot0 = 0u;
            if (r.head) slot0 = tile_start[t] + atomicAdd(&cursor[t], r.len);
            // all lanes of a run walk the same rectangle, so the head is in this iteration too
            const uint32_t slot = (uint32_t)__shfl((int)slot0, r.head_lane) + (uint32_t)(lane - r.head_lane);
            if (slot < capacity) list[slot] = entry;
        }
    }
}

__device__ __forceinline__ int64_t live_count(const int32_t* d_n, int64_t cap) {
    if (d_n == nullptr) return cap;
    int64_t v = (int64_t)(*d_n);
    if (v < 0) v = 0;
    return v < cap ? v : cap;
}

#define EP_BIN_PPB 2048  // points per workgroup in the binning kernels (256 threads)
#define EP_BIN_PPT 8  // ... per thread

// world cloud [a | b] -> NDC records (+ per-tile counts when do_count).  PROJECT=false: `a` already holds NDC
// records (b unused) and nothing is written to `out`.  LDS_AGG: the workgroup first counts into an LDS histogram of
// all tiles and then adds the non-zero entries to global memory: consecutive points of an exo frame land on the same
// few tiles, so this replaces thousands of same-address global atomics by a handful per workgroup.
// (Eight waves per SIMD = at most 64 VGPRs: a workgroup then fits next to three splat waves per SIMD, which leave 65.)
template <bool PROJECT, bool LDS_AGG>
__global__ void __launch_bounds__(256, 8) k_project_count(egoprior_camera cam, RasterGeom g,
                                                       const egoprior_point* __restrict__ a, int64_t cap_a,
                                                       const int32_t* __restrict__ d_n_a,
                                                       const egoprior_point* __restrict__ b, int64_t cap_b,
                                                       const int32_t* __restrict__ d_n_b,
                                                       egoprior_point* __restrict__ out, int32_t* __restrict__ d_total,
                                                       uint32_t* __restrict__ tile_count, int do_count,
                                                       uint32_t* __restrict__ wg_piece) {
    extern __shared__ uint32_t hist[];
    const int n_tiles = g.tiles_x * g.tiles_y;
    const int64_t na = live_count(d_n_a, cap_a);
    const int64_t nb = (b != nullptr) ? live_count(d_n_b, cap_b) : 0;
    const int64_t n = na + nb;
    if (blockIdx.x == 0 && threadIdx.x == 0 && d_total != nullptr) *d_total = (int32_t)n;
    if ((int64_t)blockIdx.x * EP_BIN_PPB >= n) return;  // the grid is sized for the capacity; the live count is smaller
    const int n_words = (n_tiles + 1) >> 1;
    const int span = straight_span(g);
    if (LDS_AGG && do_count) {
        for (int t = threadIdx.x; t < n_words; t += blockDim.x) hist[t] = 0u;
        __syncthreads();
    }
    // whole waves stay in the loops together (the binning uses wave-wide ballots)
    const int64_t n_round = (n + EP_WAVE - 1) / EP_WAVE * EP_WAVE;
    for (int64_t base = (int64_t)blockIdx.x * EP_BIN_PPB; base < n_round; base += (int64_t)gridDim.x * EP_BIN_PPB) {
        const int64_t lim = (base + EP_BIN_PPB < n_round) ? base + EP_BIN_PPB : n_round;
        // the record of the next trip is requested before this trip's arithmetic (indices past the end read the last record)
        auto fetch = [&](int64_t i) -> egoprior_point {
            i = i < n ? i : n - 1;
            return (i < na) ? a[i] : b[i - na];
        };
        egoprior_point nxt = fetch(base + threadIdx.x);
        for (int64_t i = base + threadIdx.x; i < lim; i += blockDim.x) {
            const bool live = i < n;
            float ox = 0.f, oy = 0.f, oz = -1.f;
            const egoprior_point p = nxt;
            if (i + blockDim.x < lim) nxt = fetch(i + blockDim.x);
            if (live) {
                if (PROJECT) {
                    project_point(cam, p.x, p.y, p.z, ox, oy, oz);
                    // The render-entry filters of the reference (finite coordinates, |p| < 1e6 with the float32 norm
                    // sqrt((x*x + y*y) + z*z), scripts/render_vipe_pointcloud.py:807-821) for clouds that did not pass
                    // through egoprior_filter_cloud: sqrtf(s) < 1e6f exactly when s < float32(1e12), and a non-finite
                    // coordinate makes s infinite or NaN.  A point that fails is never drawn (it keeps its index).
                    if (do_count && !(((p.x * p.x + p.y * p.y) + p.z * p.z) < 999999995904.0f)) oz = -1.0f;
                    egoprior_point q;
                    q.x = ox;
                    q.y = oy;
                    q.z = oz;
                    q.rgba = p.rgba;
                    if (out != nullptr) out[i] = q;  // NULL: counting only
                } else {
                    ox = p.x;
                    oy = p.y;
                    oz = p.z;
                }
            }
            if (do_count) {
                int tx0 = 0, tx1 = 0, ty0 = 0, ty1 = 0;
                const bool hit = tile_range(g, ox, oy, oz, tx0, tx1, ty0, ty1) && live;
                const LaneRun r = lane_runs(hit, tx0, tx1, ty0, ty1);
                if (LDS_AGG)
                    count_run_lds(r, span, tx0, tx1, ty0, ty1, g.tiles_x, hist);
                else
                    count_run_global(r, tx0, tx1, ty0, ty1, g.tiles_x, tile_count);
            }
        }
    }
    if (LDS_AGG && do_count) {
        __syncthreads();
        // wg_piece (optional): the value the atomic returns is where this workgroup's entries start inside the tile's list;
        // k_tile_fill_pieces, which sends the same points through the same workgroups, picks it up from there
        uint32_t* const mine = (wg_piece != nullptr) ? wg_piece + (size_t)blockIdx.x * (size_t)n_tiles : nullptr;
        // four words per thread and trip: the returning atomics of a trip are all issued before the first result is used
        for (int w0 = threadIdx.x; w0 < n_words; w0 += 4 * blockDim.x) {
            uint32_t c[4], b0[4], b1[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = w0 + j * (int)blockDim.x;
                c[j] = (w < n_words) ? hist[w] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int w = w0 + j * (int)blockDim.x;
                b0[j] = 0u;
                b1[j] = 0u;
                if ((c[j] & 0xFFFFu) != 0u) b0[j] = atomicAdd(&tile_count[2 * w], c[j] & 0xFFFFu);
                if ((c[j] >> 16) != 0u) b1[j] = atomicAdd(&tile_count[2 * w + 1], c[j] >> 16);
            }
            if (mine != nullptr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int w = w0 + j * (int)blockDim.x;
                    if ((c[j] & 0xFFFFu) != 0u) mine[2 * w] = b0[j];
                    if ((c[j] >> 16) != 0u) mine[2 * w + 1] = b1[j];
                }
            }
        }
    }
}

// exclusive scan of tile counts; also zeroes the cursors the fill pass bumps, reports list overflow, and writes the
// launch order of the splat kernel: tiles by decreasing list length (in steps of 16 entries), so that the long tiles
// start first and the short ones fill the tail of the launch
// A tile whose list is longer than 2 * EP_SLICE_LEN entries is cut into slices of about EP_SLICE_LEN entries (at most
// EP_MAX_SLICES of them), each of which a wave of its own reduces to per-pixel lists; the last of these waves to finish
// merges the lists and shades the tile.  Without this a few hundred heavy tiles of a dense cloud leave most of the GPU idle.
struct SlicePlan {
    uint32_t* slot_base;   // n_tiles: first slot of a sliced tile, EP_NO_SLOT for the others
    uint32_t* done;        // n_tiles: slices of the tile that have finished (sliced tiles only)
    uint32_t* map;         // max_slots: tile | slice << 24 of every slot
    uint32_t* total;       // [0]: slots in use this launch, [1], [2]: the tail of this launch, a range of launch positions
    unsigned long long* partial;  // (max_slots + 3 max_tail) x EP_K x 64 keys: the per-pixel lists of every slice, descending
    uint32_t max_slots;    // slots for the slices of heavy tiles; 0 with max_tail == 0: slicing is off
    uint32_t wave_slots;   // waves of the splat kernel the GPU holds at once; 0: no tail slicing
    uint32_t* tail_done;   // max_tail: slices of a tail tile that have finished
    uint32_t max_tail;     // tiles the tail of a launch holds at most
#ifdef EP_CHECK
    uint32_t nonce;        // differs from launch to launch (diagnostic build: seeds the hand-over checksums)
#endif
};
#ifndef EP_SLICE_LEN
#define EP_SLICE_LEN 16384u
#endif
#ifndef EP_MAX_SLICES
#define EP_MAX_SLICES 64u
#endif
#define EP_NO_SLOT 0xFFFFFFFFu
// Words (64-bit) of one slot of the slice area: EP_K keys for each of the 64 pixels.  The -DEP_CHECK build appends one
// checksum per pixel, which the slice wave writes after its list and the merging wave verifies (see the hand-over).
#ifdef EP_CHECK
#define EP_SLOT_U64 ((EP_K + 1) * 64)
// hand-overs verified / hand-overs whose checksum did not match what the merging wave read (diagnostic build only)
__device__ unsigned long long g_check_seen, g_check_failed;
// Position-dependent, so a permuted list fails too; seeded with the launch, the tile and the slice, so that a list left
// in the slot by an earlier launch (or by another tile's slice) fails as well.
__device__ __forceinline__ unsigned long long check_seed(uint32_t nonce, uint32_t tile, uint32_t slice_no, int lane) {
    return ((unsigned long long)nonce << 32 | (unsigned long long)tile) * 0x9E3779B97F4A7C15ull ^
           ((unsigned long long)slice_no << 40 | (unsigned long long)(uint32_t)lane << 8);
}
__device__ __forceinline__ unsigned long long check_mix(unsigned long long h, unsigned long long key, int k) {
    const unsigned long long v = (key + (unsigned long long)(k + 1)) * 0xD6E8FEB86659FD93ull;
    return h ^ ((v << (k + 1)) | (v >> (63 - k)));
}
#else
#define EP_SLOT_U64 (EP_K * 64)
#endif
// The tail of a launch.  When the tiles with real work (EP_TAIL_MIN entries or more) outnumber the wave slots of the GPU
// and the last round of them would be at most half full, that round would keep most of the GPU idle for the duration of
// a whole tile.  The launch order sorts the tiles by falling list length, so the tiles of that round are a range
// [P0, P1) of launch positions: k_tile_scan only has to publish the two numbers (and to clear one completion counter
// per tail tile).  The workgroup at position p of the range takes slice 0 of its tile, two workgroups right behind the
// tiles of the grid take slices 1 and 2, and the three hand their lists over in slot 3 (p - P0) + slice of the slice area
// like the slices of a heavy tile do.
#define EP_TAIL_MIN 384u
#define EP_TAIL_SLICES 3u
__host__ __device__ __forceinline__ uint32_t slices_for(uint32_t count) {  // heavy tiles
    if (count <= 2u * EP_SLICE_LEN) return 1u;
    const uint32_t ns = (count + EP_SLICE_LEN - 1u) / EP_SLICE_LEN;
    return ns > EP_MAX_SLICES ? EP_MAX_SLICES : ns;
}
// entries of one slice: an equal share, rounded up to whole loads of 64
__host__ __device__ __forceinline__ uint32_t slice_span(uint32_t count, uint32_t ns) {
    return (((count + ns - 1u) / ns) + 63u) & ~63u;
}

// ONCE: the grid has at most NT * PT tiles; every thread then holds the counts of its PT consecutive tiles in registers
// from one round of loads to the end of the kernel (a single workgroup waits out every memory round trip it makes: the
// looping form makes eight of them at 6 240 tiles, this one makes one).
template <int NT, int PT = 8, bool ONCE = false>
__global__ void __launch_bounds__(NT) k_tile_scan(const uint32_t* __restrict__ tile_count,
                                                    uint32_t* __restrict__ tile_start,
                                                    uint32_t* __restrict__ tile_cursor,
                                                    uint32_t* __restrict__ tile_order, int n_tiles, uint32_t capacity,
                                                    int32_t* __restrict__ d_status, SlicePlan sp) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry_s;
    __shared__ uint32_t ohist[256];
    __shared__ uint32_t want_s, p0_s, p1_s, slot_s;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    if (tid == 0) {
        carry_s = 0;
        want_s = 0;
        p0_s = 0;
        p1_s = 0;
        slot_s = 0;
    }
    if (tid < 256) ohist[tid] = 0u;
    __syncthreads();
    uint32_t want = 0;  // slots the heavy tiles among this thread's tiles would take
    uint32_t v[PT];
    for (int base = 0; base < n_tiles; base += NT * PT) {
        const int t0 = base + tid * PT;
        uint32_t sum = 0;
        if (ONCE && t0 + PT <= n_tiles) {
            // (the counters start at the 256-byte-aligned base of the workspace)
#pragma unroll
            for (int k = 0; k < PT; k += 4) {
                const uint4 q = *reinterpret_cast<const uint4*>(tile_count + t0 + k);
                v[k] = q.x;
                v[k + 1] = q.y;
                v[k + 2] = q.z;
                v[k + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < PT; ++k) v[k] = (t0 + k < n_tiles) ? tile_count[t0 + k] : 0u;
        }
#pragma unroll
        for (int k = 0; k < PT; ++k) sum += v[k];
        uint32_t incl = sum;
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
        uint32_t run = carry + woff + incl - sum;
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            if (t0 + k < n_tiles) {
                tile_start[t0 + k] = run;
                tile_cursor[t0 + k] = 0u;
                if (v[k] > 2u * EP_SLICE_LEN) want += slices_for(v[k]);
                atomicAdd(&ohist[255u - min(v[k] >> 4, 255u)], 1u);
            }
            run += v[k];
        }
        __syncthreads();
        if (tid == NT - 1) carry_s = run;
        __syncthreads();
    }
    if (tid == 0) {
        tile_start[n_tiles] = carry_s;
        if (d_status != nullptr && carry_s > capacity) *d_status = 1;  // sticky: only the caller clears it
    }
    // slots the heavy tiles want (known before the launch order is written, which hands the slots out)
    if (sp.max_slots != 0u && __ballot(want != 0u) != 0ull) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) want += __shfl_xor(want, d);
        if (lane == 0) atomicAdd(&want_s, want);
    }
    // exclusive scan of the 256 length classes (4 waves), then the launch order
    uint32_t c = 0, incl = 0;
    if (tid < 256) {
        c = ohist[tid];
        incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wave_sums[wid] = incl;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t woff = 0;
        for (int w2 = 0; w2 < wid; ++w2) woff += wave_sums[w2];
        ohist[tid] = woff + incl - c;
    }
    __syncthreads();
    // The tail (classes are 16 entries wide, class index 255 - length / 16: ohist[k] = tiles longer than class k's = the
    // launch position of class k's first tile).
    if (tid < 256 && sp.max_tail != 0u && sp.wave_slots != 0u) {
        constexpr uint32_t k_min = 255u - EP_TAIL_MIN / 16u;  // the class of the shortest tiles that count as work
        const uint32_t n_long = ohist[k_min + 1u];
        const uint32_t rounds = n_long / sp.wave_slots;
        const uint32_t first_of_tail = rounds * sp.wave_slots;  // rank, longest first, of the first tile of the last round
        const uint32_t tail = n_long - first_of_tail;
        if (rounds != 0u && tail != 0u && 2u * tail <= sp.wave_slots) {
            // the class that holds the tile of that rank (its longer members are cut too); not class 0, which holds every
            // list of 4080 entries or more
            const uint32_t mine_k = (uint32_t)tid;
            const bool here = mine_k >= 1u && mine_k <= k_min && ohist[mine_k] <= first_of_tail &&
                              first_of_tail < ohist[mine_k + 1u];
            if (here && n_long - ohist[mine_k] <= sp.max_tail) {
                p0_s = ohist[mine_k];
                p1_s = n_long;
            }
        }
    }
    __syncthreads();
    const uint32_t tail_p0 = p0_s, tail_p1 = p1_s;
    for (uint32_t q = (uint32_t)tid; q < tail_p1 - tail_p0; q += (uint32_t)NT) sp.tail_done[q] = 0u;
    // heavy tiles all or nothing
    uint32_t want_all = (sp.max_slots != 0u) ? want_s : 0u;
    if (want_all > sp.max_slots) want_all = 0u;
    const bool slicing = want_all != 0u;
    if (tid == 0 && (sp.max_slots != 0u || sp.max_tail != 0u)) {
        sp.total[0] = want_all;  // (the per-tile slot words are only read when this says that there are slices)
        sp.total[1] = tail_p0;
        sp.total[2] = tail_p1;
    }
    // The launch order, and with it the slots of the sliced tiles: which slot a slice gets does not matter, so they are
    // handed out by a counter as the tiles come by.
    // (eight counts are requested per trip: one at a time, every trip of this single workgroup waited out a memory round trip)
    for (int t0 = ONCE ? tid * PT : tid; t0 < n_tiles; t0 += NT * PT) {
        uint32_t vv[PT];
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const int t = ONCE ? t0 + k : t0 + k * NT;
            vv[k] = ONCE ? v[k] : tile_count[t < n_tiles ? t : n_tiles - 1];
        }
#pragma unroll
        for (int k = 0; k < PT; ++k) {
            const int t = ONCE ? t0 + k : t0 + k * NT;
            if (t >= n_tiles) break;
            const uint32_t vk = vv[k];
            tile_order[atomicAdd(&ohist[255u - min(vk >> 4, 255u)], 1u)] = (uint32_t)t;
            if (slicing) {
                const uint32_t ns = slices_for(vk);
                uint32_t slot = EP_NO_SLOT;
                if (ns > 1u) {
                    slot = atomicAdd(&slot_s, ns);
                    sp.done[t] = 0u;
                    for (uint32_t q = 0; q < ns && slot + q < sp.max_slots; ++q) sp.map[slot + q] = (uint32_t)t | (q << 24);
                }
                sp.slot_base[t] = slot;
            }
        }
    }
}

// Second binning pass: writes the tile lists.  Every thread fetches its EP_BIN_PPT records together; the first lane of
// a run (see lane_runs) reserves the run's slots in each tile of the rectangle with one returning global atomic and
// hands them to its lanes, so the entries of a run land in consecutive slots.
__global__ void __launch_bounds__(256) k_tile_fill(RasterGeom g, const egoprior_point* __restrict__ ndc, int64_t cap,
                                                   const int32_t* __restrict__ d_n,
                                                   const uint32_t* __restrict__ tile_start,
                                                   uint32_t* __restrict__ tile_cursor,
                                                   ListEntry* __restrict__ list, uint32_t capacity) {
    const int64_t n = live_count(d_n, cap);
    const int64_t n_round = (n + EP_WAVE - 1) / EP_WAVE * EP_WAVE;
    for (int64_t base = (int64_t)blockIdx.x * EP_BIN_PPB; base < n_round; base += (int64_t)gridDim.x * EP_BIN_PPB) {
        egoprior_point pin[EP_BIN_PPT];
#pragma unroll
        for (int k = 0; k < EP_BIN_PPT; ++k) {
            int64_t i = base + k * 256 + threadIdx.x;
            i = i < n ? i : n - 1;
            pin[k] = ndc[i];
        }
#pragma unroll
        for (int k = 0; k < EP_BIN_PPT; ++k) {
            const int64_t i = base + k * 256 + threadIdx.x;
            int tx0 = 0, tx1 = 0, ty0 = 0, ty1 = 0;
            const bool hit = (i < n) && tile_range(g, pin[k].x, pin[k].y, pin[k].z, tx0, tx1, ty0, ty1);
            const ListEntry entry = make_entry((uint32_t)i, pin[k].z);
            bin_point_runs_global(hit, tx0, tx1, ty0, ty1, g.tiles_x, tile_cursor, tile_start, list, capacity, entry);
        }
    }
}

// One point per lane into the tiles of its rectangle, which takes at most SPAN x SPAN tiles: the candidates in
// straight-line code; the heads' loads of all of them are in flight together (one branch, results used after it).
template <int SPAN>
__device__ __forceinline__ void fill_span(bool hit, const LaneRun& r, int tx0, int tx1, int ty0, int ty1, int tiles_x,
                                          uint32_t within, ListEntry entry, const uint32_t* __restrict__ tile_start,
                                          const uint32_t* __restrict__ mine, uint32_t* cur,
                                          ListEntry* __restrict__ list, uint32_t capacity) {
    constexpr int NQ = SPAN * SPAN;
    const int t00 = ty0 * tiles_x + tx0;
    bool in_q[NQ];
    int t_q[NQ];
    uint32_t slot0[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        in_q[q] = hit && tx0 + (q % SPAN) <= tx1 && ty0 + (q / SPAN) <= ty1;
        t_q[q] = in_q[q] ? t00 + (q / SPAN) * tiles_x + (q % SPAN) : t00;
        slot0[q] = 0u;
    }
    if (r.head) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) slot0[q] = tile_start[t_q[q]] + mine[t_q[q]];
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        if (in_q[q] && r.head) {
            const uint32_t sh = (uint32_t)(t_q[q] & 1) * 16u;
            slot0[q] += (atomicAdd(&cur[t_q[q] >> 1], r.len << sh) >> sh) & 0xFFFFu;
        }
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const uint32_t slot = (uint32_t)__shfl((int)slot0[q], r.head_lane) + within;
        if (in_q[q] && slot < capacity) list[slot] = entry;
    }
}

// The fill pass without global atomics.  The counting pass added every workgroup's per-tile counts to
// the global counters and kept what those atomics returned (wg_piece): that is where the workgroup's entries start
// inside each tile's list.  The same grid over the same points runs here, so a workgroup only has to hand out the slots
// inside its own pieces, which it does with returning LDS atomics on 16-bit cursors (two per word, as in the counting
// pass).  List order still depends on which workgroup's flush reached a counter first, the selection does not.
__global__ void __launch_bounds__(256) k_tile_fill_pieces(RasterGeom g, const egoprior_point* __restrict__ ndc,
                                                          int64_t cap, const int32_t* __restrict__ d_n,
                                                          const uint32_t* __restrict__ tile_start,
                                                          const uint32_t* __restrict__ wg_piece,
                                                          ListEntry* __restrict__ list, uint32_t capacity) {
    extern __shared__ uint32_t cur[];  // per tile: entries this workgroup has handed out so far, two 16-bit counters per word
    const int n_tiles = g.tiles_x * g.tiles_y;
    const int64_t n = live_count(d_n, cap);
    if ((int64_t)blockIdx.x * EP_BIN_PPB >= n) return;
    // (Cursors that already hold the final slot - tile_start[t] + piece[t], loaded in one coalesced pass - spare the loop
    // its two global loads per run and tile and make the kernel 2 us faster on its own, but take 4 bytes per tile: 25 KiB
    // at 480 x 832, which no longer fits into the 16 KiB of LDS that twelve splat waves leave of a compute unit.  With
    // frames in flight the small footprint wins: +2 % frames/s, +4 % at 1024 x 1024.)
    const uint32_t* const mine = wg_piece + (size_t)blockIdx.x * (size_t)n_tiles;
    for (int w = threadIdx.x; w < (n_tiles + 1) / 2; w += 256) cur[w] = 0u;
    __syncthreads();
    const int64_t n_round = (n + EP_WAVE - 1) / EP_WAVE * EP_WAVE;
    const int lane = lane_id();
    const int span = straight_span(g);
    for (int64_t base = (int64_t)blockIdx.x * EP_BIN_PPB; base < n_round; base += (int64_t)gridDim.x * EP_BIN_PPB) {
        egoprior_point pin[EP_BIN_PPT];
#pragma unroll
        for (int k = 0; k < EP_BIN_PPT; ++k) {
            int64_t i = base + k * 256 + threadIdx.x;
            i = i < n ? i : n - 1;
            pin[k] = ndc[i];
        }
#pragma unroll
        for (int k = 0; k < EP_BIN_PPT; ++k) {
            const int64_t i = base + k * 256 + threadIdx.x;
            int tx0 = 0, tx1 = 0, ty0 = 0, ty1 = 0;
            const bool hit = tile_range(g, pin[k].x, pin[k].y, pin[k].z, tx0, tx1, ty0, ty1) && (i < n);
            const ListEntry entry = make_entry((uint32_t)i, pin[k].z);
            const LaneRun r = lane_runs(hit, tx0, tx1, ty0, ty1);
            if (__ballot(hit) == 0ull) continue;
            const uint32_t within = (uint32_t)(lane - r.head_lane);
            if (span == 2) {
                fill_span<2>(hit, r, tx0, tx1, ty0, ty1, g.tiles_x, within, entry, tile_start, mine, cur, list, capacity);
            } else if (span == 3) {
                fill_span<3>(hit, r, tx0, tx1, ty0, ty1, g.tiles_x, within, entry, tile_start, mine, cur, list, capacity);
            } else if (hit) {
                for (int ty = ty0; ty <= ty1; ++ty) {
                    for (int tx = tx0; tx <= tx1; ++tx) {
                        const int t = ty * g.tiles_x + tx;
                        uint32_t slot0 = 0u;
                        if (r.head) {
                            const uint32_t sh = (uint32_t)(t & 1) * 16u;
                            slot0 = tile_start[t] + mine[t] + ((atomicAdd(&cur[t >> 1], r.len << sh) >> sh) & 0xFFFFu);
                        }
                        // all lanes of a run walk the same rectangle, so the head is in this iteration too
                        const uint32_t slot = (uint32_t)__shfl((int)slot0, r.head_lane) + within;
                        if (slot < capacity) list[slot] = entry;
                    }
                }
            }
        }
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

typedef unsigned long long u64;

#if defined(EP_TILE_TIMES) || defined(EP_WAVE_SPANS)
// Diagnostic builds only: when each of the first 16384 waves of the last k_splat_sweep launch started and ended (low 32
// bits of the 100 MHz real-time counter, the same on every XCD), its list entries and its chunks.  -DEP_WAVE_SPANS alone
// adds two counter reads and one 16-byte store per wave: the launch keeps its shape.
__device__ uint4 g_wave_span[16384];
__device__ uint4 g_wave_chunk_end[16384];  // ... and when its first four chunks ended
#define EP_SPAN_BEGIN()                                                     \
    const unsigned long long tt0_real = __builtin_amdgcn_s_memrealtime(); \
    uint32_t span_chunk_end[4] = {0u, 0u, 0u, 0u};
#define EP_SPAN_CHUNK()                                                                      \
    {                                                                                        \
        const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memrealtime();                    \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) if (chunk_no - 1 == i_) span_chunk_end[i_] = now_; \
    }
#define EP_SPAN_END()                                                                                              \
    if (lane == 0 && blockIdx.x < 16384u)                                                                          \
    {                                                                                                              \
        g_wave_span[blockIdx.x] = make_uint4((uint32_t)tt0_real, (uint32_t)__builtin_amdgcn_s_memrealtime(), span_records, \
                                             (uint32_t)chunk_no);                                                  \
        g_wave_chunk_end[blockIdx.x] =                                                                             \
            make_uint4(span_chunk_end[0], span_chunk_end[1], span_chunk_end[2], span_chunk_end[3]);                \
    }
#else
#define EP_SPAN_BEGIN()
#define EP_SPAN_CHUNK()
#define EP_SPAN_END()
#endif
#ifdef EP_TILE_TIMES
// Diagnostic build only: (tile, records, cycles, chunks) of every wave of the last k_splat_sweep launch.
__device__ uint32_t g_tile_times[8192 * 4];
// ... and the cycles all waves spent in each stage: gather, sort, sweep, extraction + merge, compositor; [5] waves,
// [6] chunks, [7] chunks sorted by the bitonic fallback, [8] buckets ranked by the whole wave, [9] records in those,
// [10] records sorted, [11..14] gather: cycles waiting for keys, for x, y, in the tests, in the staging, [15] batches,
// [16..19] merge stage: mask walk, key reads + network, tile threshold, band tables
__device__ unsigned long long g_stage_cycles[32];  // [20..25] sweep: blocks of the lane-per-record form, of the open-pixel form, open pixels in each, early exits, blocks offered
#define EP_STAGE(i_)                                                 \
    {                                                                \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        stage_cyc[i_] += now_ - stage_t;                             \
        stage_t = now_;                                              \
    }
#define EP_DIAG_FLUSH()                                                                         \
    {                                                                                           \
        if (lane == 0) {                                                                        \
            for (int i = 0; i < 5; ++i) atomicAdd(&g_stage_cycles[i], stage_cyc[i]);            \
            atomicAdd(&g_stage_cycles[5], 1ull);                                                \
            atomicAdd(&g_stage_cycles[6], (unsigned long long)chunk_no);                        \
            atomicAdd(&g_stage_cycles[7], (unsigned long long)dbg_bitonic);                     \
            atomicAdd(&g_stage_cycles[8], (unsigned long long)dbg_wave_buckets);                \
            atomicAdd(&g_stage_cycles[9], (unsigned long long)dbg_wave_records);                \
            atomicAdd(&g_stage_cycles[10], (unsigned long long)dbg_sorted);                     \
            for (int i = 0; i < 5; ++i) atomicAdd(&g_stage_cycles[11 + i], dbg_g[i]);           \
            for (int i = 0; i < 4; ++i) atomicAdd(&g_stage_cycles[16 + i], dbg_m[i]);           \
            for (int i = 0; i < 6; ++i) atomicAdd(&g_stage_cycles[20 + i], dbg_s[i]);           \
        }                                                                                       \
        if (lane == 0 && blockIdx.x < 8192u) {                                                  \
            g_tile_times[blockIdx.x * 4 + 0] = (uint32_t)tile | (slice_no << 24);               \
            g_tile_times[blockIdx.x * 4 + 1] = tt_records;                                      \
            g_tile_times[blockIdx.x * 4 + 2] = (uint32_t)(__builtin_readcyclecounter() - tt0);  \
            g_tile_times[blockIdx.x * 4 + 3] = (uint32_t)chunk_no;                              \
        }                                                                                       \
    }
#else
#define EP_STAGE(i_)
#define EP_DIAG_FLUSH()
#endif
// Diagnostic build only: STOP > 0 instantiations of the sweep kernel are instruction-count probes (option "splat_variant"
// 11..14) whose chunk loop ends after the load (1), sort (2) or sweep (3) stage, or that skip the compositor (4); a
// checksum keeps the skipped-over values alive.  They produce no image.
#ifdef EP_TILE_TIMES
#define EP_PROBE(n_, expr_)                                                                   \
    if (STOP == n_) {                                                                         \
        if ((uint32_t)(expr_) == 0x12345u && out.image_u8 != nullptr) out.image_u8[0] = 1;    \
        continue;                                                                             \
    }
#else
#define EP_PROBE(n_, expr_)
#endif

typedef float float2v __attribute__((ext_vector_type(2)));

#define EP_RANK_SELF 8    // buckets up to this long are ordered by their own records
#define EP_RANK_WAVE 128  // buckets up to this long are ordered by the wave; longer ones send the chunk to bitonic sort

__device__ __forceinline__ u64 make_key(uint32_t zbits, uint32_t idx) { return ((u64)zbits << 32) | (u64)idx; }

// A value that is the same in every lane, moved to scalar registers (the compiler cannot see the uniformity of a
// value that came out of a butterfly reduction or of float arithmetic on uniform operands).
__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ float uniform_f32(float v) {
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
__device__ __forceinline__ u64 uniform_u64(u64 v) {
    return ((u64)uniform_u32((uint32_t)(v >> 32)) << 32) | (u64)uniform_u32((uint32_t)v);
}

__device__ __forceinline__ u64 wave_max_u64(u64 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d);
        const u64 o = ((u64)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return uniform_u64(v);
}
__device__ __forceinline__ u64 wave_min_u64(u64 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, d);
        const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), d);
        const u64 o = ((u64)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return uniform_u64(v);
}

// Insert (nz, ni) into the per-lane list kept sorted ascending by key in registers (compare-exchange chain).
__device__ __forceinline__ void bubble_insert(uint32_t (&kz)[EP_K], uint32_t (&ki)[EP_K], uint32_t nz, uint32_t ni) {
#pragma unroll
    for (int k = 0; k < EP_K; ++k) {
        const bool sw = make_key(nz, ni) < make_key(kz[k], ki[k]);
        const uint32_t tz = kz[k], ti = ki[k];
        kz[k] = sw ? nz : tz;
        ki[k] = sw ? ni : ti;
        nz = sw ? tz : nz;
        ni = sw ? ti : ni;
    }
}

// Front-to-back compositing of a lane's sorted fragments + the reference's post-processing + optional fragment dump.
// kz / ki: the pixel's fragments in ascending (z, index) order, "nothing" = all ones.  d2 is recomputed from the point's
// NDC x, y with the same operations as the hit test, so it is the same float.
// The winners' records are gathered GB at a time, all GB gathers in flight together: the gathers are random 16-byte
// reads, and issued in small groups their round trips added up to nearly half of a splat wave's lifetime.  Groups of 5
// whose slot no pixel of the wave fills are skipped (lists are filled front to back).
// Colour bytes become floats through a 256-entry table in LDS (the byte divided by 255.0f, or the byte itself when the
// cloud's colours are not to be scaled): a correctly rounded division costs ten instructions, and a pixel has sixty of
// them.  Every lane of the wave fills four entries with that same division; `lut` must not be in use by anything else.
__device__ __forceinline__ void fill_color_lut(float* lut, const egoprior_raster_params& rp,
                                               const int32_t* __restrict__ d_flags_a,
                                               const int32_t* __restrict__ d_flags_b, int lane) {
    int scale255 = rp.colors_gt1;
    if (scale255 < 0) {
        scale255 = 0;
        if (d_flags_a != nullptr && *d_flags_a != 0) scale255 = 1;
        if (d_flags_b != nullptr && *d_flags_b != 0) scale255 = 1;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float b = (float)(lane * 4 + j);
        lut[lane * 4 + j] = scale255 ? b / 255.0f : b;
    }
    __syncthreads();
}

// a / b for a divisor that is the same for every fragment.  This is the instruction sequence of the compiler's own
// float division (reciprocal refined once, quotient refined twice with exact FMA residuals) without its range scaling
// and special-case fix-up, and with the reciprocal part computed once: the same bits whenever a, b and a / b are
// normal numbers or a is zero.  Here b is the squared radius and 0 <= a < b; a quotient too small to be normal only
// ever enters 1 - a / b, which is 1 for anything below 2^-25.
__device__ __forceinline__ float refined_rcp(float b) {
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_by_uniform(float a, float b, float y) {
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    const float q1 = __builtin_fmaf(r0, y, q0);
    const float r1 = __builtin_fmaf(-b, q1, a);
    return __builtin_fmaf(r1, y, q1);
}

template <int GB>
__device__ __forceinline__ void composite_and_store(const RasterGeom& g, const egoprior_raster_params& rp,
                                                    const egoprior_point* __restrict__ ndc,
                                                    const uint32_t (&kz)[EP_K], const uint32_t (&ki)[EP_K], int xi,
                                                    int yi, float xf, float yf, const float* lut, const SplatOut& out) {
    const int K = rp.K;
    const size_t pix = (size_t)yi * (size_t)g.W + (size_t)xi;
    float cum = 1.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
    const float rcp_r2w = refined_rcp(rp.radius_sq_w);
    constexpr int G = 5;  // fragments per skip group
    static_assert(EP_K % GB == 0 && GB % G == 0, "batch and group sizes");
#pragma unroll
    for (int h0 = 0; h0 < EP_K; h0 += GB) {
        egoprior_point p[GB];
        bool any[GB / G];
#pragma unroll
        for (int gq = 0; gq < GB / G; ++gq) {
            const int h = h0 + gq * G;
            any[gq] = __ballot(h < K && ki[h] != 0xFFFFFFFFu) != 0ull;  // wave-uniform
            if (any[gq]) {
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const bool filled = (h + k < K) && (ki[h + k] != 0xFFFFFFFFu);
                    p[gq * G + k] = ndc[filled ? ki[h + k] : 0u];  // unconditional so that the gathers overlap
                }
            }
        }
#pragma unroll
        for (int gq = 0; gq < GB / G; ++gq) {
            const int h = h0 + gq * G;
            if (any[gq]) {
#pragma unroll
                for (int k = 0; k < G; ++k) {
                    const egoprior_point q = p[gq * G + k];
                    const bool filled = (h + k < K) && (ki[h + k] != 0xFFFFFFFFu);
                    // an empty slot goes through the same arithmetic on record 0 with alpha = 0, which leaves the sums as
                    // they are (all terms are >= +0): a select instead of a divergent branch per fragment
                    const float dx = q.x - xf;
                    const float dy = q.y - yf;
                    const float dd = dx * dx + dy * dy;
                    const float d2 = filled ? dd : -1.0f;
                    const float fr = lut[q.rgba & 0xFFu], fg = lut[(q.rgba >> 8) & 0xFFu], fb = lut[(q.rgba >> 16) & 0xFFu];
                    const float alpha = filled ? 1.0f - div_by_uniform(dd, rp.radius_sq_w, rcp_r2w) : 0.0f;
                    const float wgt = cum * alpha;
                    cr += wgt * fr;
                    cg += wgt * fg;
                    cb += wgt * fb;
                    cum = cum * (1.0f - alpha);
                    if (out.idx != nullptr && h + k < K) {
                        const size_t o = pix * (size_t)K + (size_t)(h + k);
                        out.idx[o] = filled ? (int32_t)ki[h + k] : -1;
                        if (out.zbuf != nullptr) out.zbuf[o] = filled ? __uint_as_float(kz[h + k]) : -1.0f;
                        if (out.dists != nullptr) out.dists[o] = d2;
                    }
                }
            } else if (out.idx != nullptr) {
#pragma unroll
                for (int k = 0; k < G; ++k)
                    if (h + k < K) {
                        const size_t o = pix * (size_t)K + (size_t)(h + k);
                        out.idx[o] = -1;
                        if (out.zbuf != nullptr) out.zbuf[o] = -1.0f;
                        if (out.dists != nullptr) out.dists[o] = -1.0f;
                    }
            }
        }
    }
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
}

// Straightforward variant: one wave per 8x8 tile, every record of the tile list is offered to every pixel and
// inserted into the register-resident sorted list.  Kept as the simple cross-check of k_splat_sweep (option
// "splat_variant" = 1) and as its A/B partner in benchmarks.
__global__ void __launch_bounds__(64) k_splat_bubble(RasterGeom g, egoprior_raster_params rp,
                                                     const egoprior_point* __restrict__ ndc,
                                                     const uint32_t* __restrict__ tile_start,
                                                     const uint32_t* __restrict__ tile_order,
                                                     const ListEntry* __restrict__ list, uint32_t capacity,
                                                     const int32_t* __restrict__ d_flags_a,
                                                     const int32_t* __restrict__ d_flags_b, SplatOut out) {
    __shared__ float4 recs[64];
    const int tile = (int)tile_order[blockIdx.x];
    const int tyi = tile / g.tiles_x, txi = tile - tyi * g.tiles_x;
    const int lane = threadIdx.x;
    const int xi = txi * EP_TILE + (lane & 7);
    const int yi = tyi * EP_TILE + (lane >> 3);
    const bool inside = (xi < g.W) && (yi < g.H);
    const float xf = pix_to_ndc(g.W - 1 - xi, g.W, g.H);
    const float yf = pix_to_ndc(g.H - 1 - yi, g.H, g.W);
    const float r2 = g.radius2;
    uint32_t kz[EP_K], ki[EP_K];
#pragma unroll
    for (int k = 0; k < EP_K; ++k) {
        kz[k] = 0xFFFFFFFFu;
        ki[k] = 0xFFFFFFFFu;
    }
    uint32_t beg = tile_start[tile], end = tile_start[tile + 1];
    if (end > capacity) end = capacity;
    if (beg > end) beg = end;
    for (uint32_t c0 = beg; c0 < end; c0 += 64) {
        const uint32_t nrec = min(64u, end - c0);
        __syncthreads();
        if ((uint32_t)lane < nrec) {
            const ListEntry e = list[c0 + lane];
            const float2v xy = *reinterpret_cast<const float2v*>(&ndc[e.x].x);
            recs[lane] = make_float4(xy.x, xy.y, __uint_as_float(e.y), __uint_as_float(e.x));
        }
        __syncthreads();
        for (uint32_t j = 0; j < nrec; ++j) {
            const float4 r = recs[j];
            const float dx = r.x - xf;
            const float dy = r.y - yf;
            const float d2 = dx * dx + dy * dy;
            const uint32_t nz = __float_as_uint(r.z), ni = __float_as_uint(r.w);
            const bool take = inside && (d2 < r2) && (make_key(nz, ni) < make_key(kz[EP_K - 1], ki[EP_K - 1]));
            if (__ballot(take) == 0ull) continue;
            if (take) bubble_insert(kz, ki, nz, ni);
        }
    }
    fill_color_lut(reinterpret_cast<float*>(recs), rp, d_flags_a, d_flags_b, lane);
    if (!inside) return;
    composite_and_store<10>(g, rp, ndc, kz, ki, xi, yi, xf, yf, reinterpret_cast<const float*>(recs), out);
}

// Hit masks of four pixels against the 64 records held one per lane: h[i] = ballot(d[i] < r2), then lanes L .. L+3 of
// (lo, hi) receive the two halves of h[0] .. h[3].  Written as two assembly blocks because a VALU instruction that reads
// a scalar register needs two other instructions between itself and the VALU instruction that wrote the register, a
// distance the compiler keeps for its own instructions but not for the operands of inline assembly: here every lane
// write sits at least three instructions behind its compare by construction.
//
// INVARIANT (hardware hazard, not checked by any tool): on gfx950 a VALU read of an SGPR needs >= 2 wait states after
// the VALU write of that SGPR.  It holds here because
//   (1) the four compares are ONE asm statement, in the order h0, h1, h2, h3, and nothing can be scheduled into it;
//   (2) the lane writes are ONE asm statement that consumes the masks in the SAME order (h0 lo, h0 hi, h1 lo, ...), so
//       the closest pair is compare h0 -> first lane write: compares h1, h2, h3 lie between them (3 >= 2), and every
//       later pair is farther apart; whatever the compiler places between the two statements only adds distance.
// Anyone who reorders either list, drops a compare, or splits a statement must re-derive the distances (or put an
// `s_nop 1` in front of the first lane write); tests/test_gpu_parity.py::test_rasterize_* fail on stale masks.
__device__ __forceinline__ void hit_masks4(const int L, float d0, float d1, float d2, float d3, float r2, uint32_t& lo,
                                           uint32_t& hi) {
    unsigned long long h0, h1, h2, h3;
    asm("v_cmp_gt_f32_e64 %0, %4, %5\n\tv_cmp_gt_f32_e64 %1, %4, %6\n\tv_cmp_gt_f32_e64 %2, %4, %7\n\tv_cmp_gt_f32_e64 %3, %4, %8"
        : "=&s"(h0), "=&s"(h1), "=&s"(h2), "=&s"(h3)
        : "s"(r2), "v"(d0), "v"(d1), "v"(d2), "v"(d3));
    asm("v_writelane_b32 %0, %2, %10\n\tv_writelane_b32 %1, %3, %10\n\t"
        "v_writelane_b32 %0, %4, %11\n\tv_writelane_b32 %1, %5, %11\n\t"
        "v_writelane_b32 %0, %6, %12\n\tv_writelane_b32 %1, %7, %12\n\t"
        "v_writelane_b32 %0, %8, %13\n\tv_writelane_b32 %1, %9, %13"
        : "+v"(lo), "+v"(hi)
        : "s"((uint32_t)h0), "s"((uint32_t)(h0 >> 32)), "s"((uint32_t)h1), "s"((uint32_t)(h1 >> 32)), "s"((uint32_t)h2),
          "s"((uint32_t)(h2 >> 32)), "s"((uint32_t)h3), "s"((uint32_t)(h3 >> 32)), "n"(L), "n"(L + 1), "n"(L + 2), "n"(L + 3));
}

// Value of lane K of the caller's row of 16 lanes.  The compiler folds the move into the instruction that uses it.
template <int K>
__device__ __forceinline__ float row_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + K, 0xf, 0xf, true));
}

// Compare-exchange of two entries of a pixel's running list: the larger key goes to position a.  (Keys are held as
// 64-bit values throughout, so that each one lives in an aligned register pair and the compares need no re-pairing;
// both values are copied to locals first: with an array element as an arm of ?: the compiler emits a branch.)
__device__ __forceinline__ void ce_desc(u64 (&kl)[EP_K], int a, int b) {
    const u64 ka = kl[a], kb = kl[b];
    const bool sw = kb > ka;
    kl[a] = sw ? kb : ka;
    kl[b] = sw ? ka : kb;
}

__device__ __forceinline__ u64 lds_key(const uint4* rec) { return *reinterpret_cast<const u64*>(rec); }

constexpr int next_pow2(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Production variant.  One wave owns one 8x8 tile and works through the tile's point list in chunks of C list
// entries (C/64 per lane, held in registers while the chunk is organised):
//   gather  : batches of C/64 independent, fully coalesced 8-byte entry loads per lane (point index and depth bits);
//             entries whose key cannot beat the largest K-th key of the tile's pixels (T) are dropped, the NDC x, y of the
//             others are gathered from the point records, discs that overlap the tile's rectangle without reaching any of
//             its pixel centres are dropped too, and so are (from the third chunk on) keys that cannot beat the K-th
//             key of any pixel row / column band their disc reaches; survivors are compacted into LDS until the chunk
//             is full or the list ends (a batch that ends the list or fills the chunk is used straight from registers)
//   sort    : a monotone two-level bucket map sends the keys to LDS in O(n).  Level 1 cuts the chunk's key range
//             into NB equal ranges and histograms them; a range holding c records is cut again into 2^floor(log2 c)
//             sub-ranges, so buckets hold a record or two unless depths repeat.  After the scatter every record ranks
//             itself inside its bucket: short buckets by reading their neighbours' keys, medium ones (up to
//             EP_RANK_WAVE) with a wave-wide all-pairs count, and a chunk with a longer bucket is bitonic-sorted.
//             S[] ends up sorted.
//   sweep   : the sorted records are offered front to back, 64 at a time, to the 64 pixels.  With many open pixels
//             sixteen records at a time sit one per lane in every row of 16 lanes and reach the row's pixels as DPP row
//             broadcasts, one hit bit per record is shifted into the pixel's mask; with few open pixels the roles swap:
//             each lane holds one record, the open pixels' coordinates come from v_readlane and a ballot is the
//             pixel's mask.  The masks go to LDS as they are (two words per pixel and block) and only a hit count is
//             kept.  Sorted input makes the first K hits of a pixel its K nearest, so a pixel closes at K hits (and, from
//             the second chunk on, at the first record beyond its own K-th key).
//   merge   : every lane walks its mask words for the positions of its first K hits (one count-trailing-zeros step per
//             winner; extracting hits during the sweep made every lane wait for the one with the most hits, block after
//             block); the winners' keys go into the register-resident running top-K (first chunk: plain copy; later
//             chunks: a 40-exchange bitonic merge network that works in place on the running list).
// The running list is kept in DESCENDING key order: position p holds the pixel's (EP_K-1-p)-th nearest fragment, empty
// positions (all ones) come first.  With the chunk's winners B in ascending order, min(list[p], B[p]) for every p is
// then the EP_K smallest keys of both as a rising-then-falling sequence, which the network sorts back to descending
// order without any scratch registers.
// LDS per wave = C*16 (sorted records) + max(NB*4, C*8) (bucket table and hit masks share the space) + 1.1 KiB of
// threshold tables: 11.6 KiB for C = 448 (the default), 13.1 KiB for C = 512; three waves per SIMD (12 per CU) in the
// 160 KiB of a CU.
// Sorts a list that rises and then falls (the element-wise minimum of a descending list and an ascending one) back into
// descending order: a bitonic merge for a length that is not a power of two (EP_K = 16 + 4); positions 20 .. 31 would
// hold minus infinity and never move, so the exchanges with them are left out.
__device__ __forceinline__ void merge_network_desc(u64 (&kl)[EP_K]) {
    static_assert(EP_K == 20, "the merge network is written for 20 entries");
#pragma unroll
    for (int i = 0; i < 4; ++i) ce_desc(kl, i, i + 16);
#pragma unroll
    for (int j = 8; j >= 1; j >>= 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if ((i & j) == 0) ce_desc(kl, i, i | j);
    }
    ce_desc(kl, 16, 18);
    ce_desc(kl, 17, 19);
    ce_desc(kl, 16, 17);
    ce_desc(kl, 18, 19);
}

#ifndef EP_PIXMAJ
#define EP_PIXMAJ 16
#endif
template <int C, int NB, int WPE, int PIXMAJ, int STOP>
__device__ __forceinline__ void splat_sweep_wave(const RasterGeom& g, const egoprior_raster_params& rp,
                                                 const egoprior_point* __restrict__ ndc,
                                                 const uint32_t* __restrict__ tile_start,
                                                 const uint32_t* __restrict__ tile_order,
                                                 const ListEntry* __restrict__ list, uint32_t capacity,
                                                 const int32_t* __restrict__ d_flags_a,
                                                 const int32_t* __restrict__ d_flags_b, const SlicePlan& sp,
                                                 const SplatOut& out) {
    static_assert(C % 64 == 0 && NB % 64 == 0 && C <= 1024 && (NB & (NB - 1)) == 0 && NB >= C, "geometry");
    constexpr int RPL = C / 64;   // records per lane while a chunk is being organised
    constexpr int PER = NB / 64;  // buckets per lane
    __shared__ uint4 S[C];        // x = point index, y = z bits, z = NDC x bits, w = NDC y bits; sorted by (y, x)
    // The bucket counters are dead once a chunk is sorted and the hit masks (one 64-bit word per pixel and block of 64
    // records) are only alive from the sweep to the merge, so the two share storage.
    constexpr int AUX0 = (NB * 4 > C * 8) ? NB * 4 : C * 8;
    __shared__ __align__(16) unsigned char aux[AUX0];
    uint32_t* const bkt = reinterpret_cast<uint32_t*>(aux);
    uint32_t(*const masks)[64] = reinterpret_cast<uint32_t(*)[64]>(aux);  // row 2 b + h: records 32 h .. 32 h + 31 of block b

    // The first workgroups of the launch take one tile each, longest list first, and leave the sliced tiles alone; the
    // sp.max_slots workgroups behind them take the slices (the slots in use come first, the others leave at once).
    // ... and in front of those, 2 sp.max_tail workgroups for slices 1 and 2 of the launch's tail tiles (see EP_TAIL_MIN),
    // whose slice 0 is taken by the tile's own workgroup.
    int tile;
    uint32_t my_slot = EP_NO_SLOT, n_slices = 1u, slice_no = 0u;
    uint32_t tail_q = EP_NO_SLOT;  // this wave works on a slice of tail tile tail_q
    const uint32_t n_tiles_all = (uint32_t)(g.tiles_x * g.tiles_y);
    uint32_t tail_p0 = 0u, tail_p1 = 0u;
    if (sp.max_tail != 0u) {
        tail_p0 = sp.total[1];
        tail_p1 = sp.total[2];
    }
    if (blockIdx.x >= n_tiles_all + 2u * sp.max_tail) {
        my_slot = blockIdx.x - n_tiles_all - 2u * sp.max_tail;
        if (my_slot >= sp.total[0]) return;
        const uint32_t packed = sp.map[my_slot];
        tile = (int)(packed & 0xFFFFFFu);
        slice_no = packed >> 24;
    } else if (blockIdx.x >= n_tiles_all) {
        const uint32_t e = blockIdx.x - n_tiles_all;
        tail_q = e >> 1;
        if (tail_q >= tail_p1 - tail_p0) return;
        tile = (int)tile_order[tail_p0 + tail_q];
        slice_no = 1u + (e & 1u);
    } else {
        tile = (int)tile_order[blockIdx.x];
        if (sp.max_slots != 0u && sp.total[0] != 0u && sp.slot_base[tile] != EP_NO_SLOT) return;
        if (blockIdx.x >= tail_p0 && blockIdx.x < tail_p1) tail_q = blockIdx.x - tail_p0;
    }
    if (tail_q != EP_NO_SLOT) {
        n_slices = EP_TAIL_SLICES;
        my_slot = sp.max_slots + EP_TAIL_SLICES * tail_q + slice_no;
    }
    const int tyi = tile / g.tiles_x, txi = tile - tyi * g.tiles_x;
    const int lane = threadIdx.x;
    const int xi = txi * EP_TILE + (lane & 7);
    const int yi = tyi * EP_TILE + (lane >> 3);
    const bool inside = (xi < g.W) && (yi < g.H);
    const float xf = pix_to_ndc(g.W - 1 - xi, g.W, g.H);
    const float yf = pix_to_ndc(g.H - 1 - yi, g.H, g.W);
    const float r2 = g.radius2;
    const int K = rp.K;
    // The binning works on tile rectangles, so a list also holds discs that overlap the rectangle's corner tile without
    // reaching any of its pixel centres.  They are dropped at load time: [cx_lo, cx_hi] x [cy_lo, cy_hi] is the box of the
    // tile's pixel centres, a disc whose centre is farther than r from it cannot hit (the margin on r^2 is far above the
    // rounding of either evaluation, so nothing that the exact test would accept is dropped).
    const float cx_hi = uniform_f32(pix_to_ndc(g.W - 1 - txi * EP_TILE, g.W, g.H));
    const float cx_lo = uniform_f32(pix_to_ndc(g.W - 1 - (txi * EP_TILE + EP_TILE - 1), g.W, g.H));
    const float cy_hi = uniform_f32(pix_to_ndc(g.H - 1 - tyi * EP_TILE, g.H, g.W));
    const float cy_lo = uniform_f32(pix_to_ndc(g.H - 1 - (tyi * EP_TILE + EP_TILE - 1), g.H, g.W));
    const float r2_reach = uniform_f32(r2 * 1.0001f);

    u64 kl[EP_K];  // (depth bits << 32 | index), descending: position EP_K-1-a holds the a-th nearest fragment
#pragma unroll
    for (int k = 0; k < EP_K; ++k) kl[k] = ~0ull;
    // the pixel's K-th nearest key so far (all ones while its list is not full); K is the same for the whole launch
    auto kth_key = [&]() -> u64 {
        if (K == EP_K) return kl[0];
        u64 v = ~0ull;
#pragma unroll
        for (int p = 0; p < EP_K; ++p)
            if (p == EP_K - K) v = kl[p];
        return v;
    };
    u64 T = ~0ull;  // only keys below T can still enter some pixel's top K
    uint32_t pos = tile_start[tile], end = tile_start[tile + 1];
    if (my_slot != EP_NO_SLOT) {  // this wave's share of the list
        const uint32_t count = end - pos;
        if (tail_q == EP_NO_SLOT) n_slices = slices_for(count);
        const uint32_t span = slice_span(count, n_slices);
        pos = min(pos + slice_no * span, end);
        end = min(pos + span, end);
    }
    if (end > capacity) end = capacity;
    if (pos > end) pos = end;
    int chunk_no = 0;
#if defined(EP_TILE_TIMES) || defined(EP_WAVE_SPANS)
    const uint32_t span_records = end - pos;
#endif
    EP_SPAN_BEGIN()
#ifdef EP_TILE_TIMES
    const unsigned long long tt0 = __builtin_readcyclecounter();
    const uint32_t tt_records = end - pos;
    unsigned long long stage_cyc[5] = {0, 0, 0, 0, 0}, stage_t = tt0;
    uint32_t dbg_bitonic = 0, dbg_wave_buckets = 0, dbg_wave_records = 0, dbg_sorted = 0;
    unsigned long long dbg_g[5] = {0, 0, 0, 0, 0}, dbg_m[4] = {0, 0, 0, 0}, dbg_s[6] = {0, 0, 0, 0, 0, 0};
#endif

    // Per-record threshold tables (read while a chunk is gathered): rowband[band(a, b)] is the largest K-th key over the
    // pixel rows a..b of the tile (all ones while a pixel's list is not full), colband the same for columns a..b.  A
    // record can only enter the list of a pixel its disc reaches, and min(rowband[rows it reaches], colband[columns it
    // reaches]) is an upper bound on the K-th keys of those pixels.
    // The tables have LDS of their own, so that they outlive the chunk after the one they were built for: a pixel's K-th
    // key only ever falls, which keeps an old table a valid bound, and the tables of a long list change little from chunk
    // to chunk.  They are rebuilt after chunks 2, 3, 4, 6, 8, 12, 16, 24, 32, ... (building them is two wave-wide
    // reductions, two barriers and a short divergent loop: as much as gathering a batch of keys).
    // They are kept small: only the 36 bands with a <= b are stored, and the column table holds the depth bits of its
    // keys only (a record then passes on equal depth whatever its index: still a bound).  With that the default kernel
    // stays within 11 KiB of LDS, and twelve waves of it leave a compute unit room for two workgroups of the binning
    // kernels instead of one (LDS is handed out in steps of 512 bytes).
    __shared__ u64 rowband[36];
    __shared__ uint32_t colband[36];
    auto band = [](int a_, int b_) { return ((a_ * (17 - a_)) >> 1) + b_ - a_; };  // a_ <= b_ < 8
    const float inv_pitch = uniform_f32(0.5f * (float)min(g.W, g.H));           // pixels per NDC unit
    const float reach_px = uniform_f32(sqrtf(r2) * inv_pitch + 0.01f);          // the margin covers every rounding

    while (pos < end) {
        // ---- gather: keys that can still matter, batch by batch, until the chunk is full --------------------------
        // Survivors are staged in S (free until this chunk's scatter) and read back lane-strided, so that the sort and
        // the sweep below always work on a (nearly) full chunk however few keys of the list pass the thresholds.
        u64 key[RPL];               // depth bits << 32 | point index
        uint32_t rx[RPL], ry[RPL];  // NDC x, y bits
        bool val[RPL];
        int m = 0;
        __syncthreads();  // the previous chunk's readers of S are done; the threshold tables are written
        // A batch = the C list entries from p on, RPL per lane.  Its keys are fetched first (issue_keys), then the x, y of
        // the keys below the tile's threshold, and while those are on their way the keys of the batch after it are
        // requested, so that only the first batch of a wave waits for two round trips in a row.  All loads are
        // unconditional (clamped addresses): a load under a branch would make every later wait count conservatively.
        auto issue_keys = [&](uint32_t p, u64(&k)[RPL], bool(&ok)[RPL]) {
#pragma unroll
            for (int u = 0; u < RPL; ++u) {
                const uint32_t e = p + (uint32_t)(u * 64 + lane);
                ok[u] = e < end;
                k[u] = reinterpret_cast<const u64*>(list)[ok[u] ? e : end - 1u];  // (index, depth bits) = the key as it is
            }
        };
        // tests the batch k against the thresholds; ok[] marks the survivors, whose number is returned; requests the keys
        // of the batch at p_next into kn on the way
        auto finish_batch = [&](const u64(&k)[RPL], uint32_t(&x)[RPL], uint32_t(&y)[RPL], bool(&ok)[RPL], uint32_t p_next,
                                u64(&kn)[RPL], bool(&okn)[RPL]) -> int {
#ifdef EP_TILE_TIMES
            const unsigned long long g_t0 = __builtin_readcyclecounter();
#endif
#pragma unroll
            for (int u = 0; u < RPL; ++u) {
                const bool want = ok[u] && (k[u] < T);  // (the others read record 0)
                const float2v xy = *reinterpret_cast<const float2v*>(&ndc[want ? (uint32_t)k[u] : 0u].x);
                x[u] = __float_as_uint(xy.x);
                y[u] = __float_as_uint(xy.y);
            }
            issue_keys(p_next, kn, okn);
#ifdef EP_TILE_TIMES
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            const unsigned long long g_t2 = __builtin_readcyclecounter();
#endif
            int s = 0;
#pragma unroll
            for (int u = 0; u < RPL; ++u) {
                const float px = __uint_as_float(x[u]), py = __uint_as_float(y[u]);
                const float ox = fmaxf(fmaxf(cx_lo - px, px - cx_hi), 0.0f);
                const float oy = fmaxf(fmaxf(cy_lo - py, py - cy_hi), 0.0f);
                bool o = ok[u] && (k[u] < T) && !(ox * ox + oy * oy > r2_reach);
                if (chunk_no > 1) {
                    // pixel columns / rows of the tile within reach of the disc (column c sits at cx_hi - c * pitch)
                    const float colf = (cx_hi - px) * inv_pitch, rowf = (cy_hi - py) * inv_pitch;
                    const int xa = max((int)ceilf(colf - reach_px), 0), xb = min((int)floorf(colf + reach_px), 7);
                    const int ya = max((int)ceilf(rowf - reach_px), 0), yb = min((int)floorf(rowf + reach_px), 7);
                    o = o && xa <= xb && ya <= yb;
                    const u64 tr = rowband[o ? band(ya, yb) : 0];
                    const uint32_t tc = colband[o ? band(xa, xb) : 0];
                    o = o && k[u] < tr && (uint32_t)(k[u] >> 32) <= tc;
                }
                ok[u] = o;
                s += __popcll(__ballot(o));
            }
#ifdef EP_TILE_TIMES
            dbg_g[1] += g_t2 - g_t0;
            dbg_g[2] += __builtin_readcyclecounter() - g_t2;
            dbg_g[4] += 1;
#endif
            return s;
        };
        // Compacts the survivors of a batch into S, row by row (a row = the 64 entries of one load), for as long as whole
        // rows fit into the chunk; returns the number of rows taken.  The list position moves on by those rows only, so
        // a row that did not fit is simply fetched again for the next chunk (it is in the caches by then).
        auto stage_batch = [&](const u64(&k)[RPL], const uint32_t(&x)[RPL], const uint32_t(&y)[RPL],
                               const bool(&ok)[RPL]) -> int {
            int rows = 0;
            bool open = true;  // wave-uniform
#pragma unroll
            for (int u = 0; u < RPL; ++u) {
                const u64 bal = __ballot(ok[u]);
                const int c = __popcll(bal);
                open = open && (m + c <= C);
                if (open) {
                    if (ok[u]) {
                        const int dst = m + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(bal >> 32),
                                                                           __builtin_amdgcn_mbcnt_lo((uint32_t)bal, 0u));
                        S[dst] = make_uint4((uint32_t)k[u], (uint32_t)(k[u] >> 32), x[u], y[u]);
                    }
                    m += c;
                    ++rows;
                }
            }
            return rows;
        };
        u64 kn[RPL];  // the keys of the batch after the one at work
        bool okn[RPL];
#ifdef EP_TILE_TIMES
        const unsigned long long g_k0 = __builtin_readcyclecounter();
#endif
        issue_keys(pos, key, val);
#ifdef EP_TILE_TIMES
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg_g[0] += __builtin_readcyclecounter() - g_k0;
#endif
        const int s0 = finish_batch(key, rx, ry, val, pos + (uint32_t)C, kn, okn);
        pos += C;
        if (pos >= end || s0 > C - 64) {
            // the batch ends the list or (nearly) fills the chunk by itself: it is sorted straight from the registers
            // (the keys requested on the way are dropped: they are the next chunk's first batch, in the caches by then)
            m = s0;
        } else {
            stage_batch(key, rx, ry, val);  // s0 <= C - 64: every row fits
            // Whole batches from here on, however little room is left: a batch costs the same memory round trips
            // whatever its size, and with a few keys in a hundred passing the thresholds the chunk takes many of them.
            while (pos < end && m <= C - 64) {
                u64 k2[RPL];
                uint32_t x2[RPL], y2[RPL];
                bool ok2[RPL];
#pragma unroll
                for (int u = 0; u < RPL; ++u) {
                    k2[u] = kn[u];
                    ok2[u] = okn[u];
                }
                finish_batch(k2, x2, y2, ok2, pos + (uint32_t)C, kn, okn);
                const int rows = stage_batch(k2, x2, y2, ok2);
                pos += (uint32_t)rows * 64u;
                if (rows < RPL) break;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < RPL; ++u) {
                const int i = u * 64 + lane;
                val[u] = i < m;
                const uint4 rec = S[val[u] ? i : 0];
                key[u] = make_key(rec.y, rec.x);
                rx[u] = rec.z;
                ry[u] = rec.w;
            }
        }
        if (m == 0) {
            if (pos < end) continue;
            break;
        }
        // The key range of the chunk from its depths alone (32-bit reductions): every index fits under the largest depth
        // and above the smallest, and the map below only needs bounds.
        uint32_t zmin = 0xFFFFFFFFu, zmax = 0u;
#pragma unroll
        for (int u = 0; u < RPL; ++u) {
            const uint32_t zb = (uint32_t)(key[u] >> 32);
            zmin = min(zmin, val[u] ? zb : 0xFFFFFFFFu);
            zmax = max(zmax, val[u] ? zb : 0u);
        }
        EP_STAGE(0)
#ifdef EP_TILE_TIMES
        dbg_sorted += (uint32_t)m;
#endif
        EP_PROBE(1, (uint32_t)key[0] ^ rx[0] ^ ry[RPL - 1] ^ (uint32_t)m)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            zmin = min(zmin, (uint32_t)__shfl_xor((int)zmin, d));
            zmax = max(zmax, (uint32_t)__shfl_xor((int)zmax, d));
        }
        const u64 kmin = (u64)uniform_u32(zmin) << 32, kmax = ((u64)uniform_u32(zmax) << 32) | 0xFFFFFFFFull;
        // Two-level monotone bucket map.  Level 1: NB equal-width ranges of [kmin, kmax].  Their histogram, and then the
        // table that maps a range to its first bucket and sub-range count, live in the first bytes of S, which is not in
        // use until the scatter.  Range c, which received cc records, is cut into 2^floor(log2 cc) equal sub-ranges, so
        // buckets hold a record or two wherever the keys are locally uniform, however unevenly the ranges are populated.
        uint32_t bq[RPL];
        bool flat;          // (wave-uniform) no level-1 range holds more than EP_RANK_SELF records: level 1 is the bucket map
        {
            constexpr int LOGR = __builtin_ctz((unsigned)NB);
            static_assert(4 * NB <= 16 * C, "the level-1 table fits into S");
            const u64 range = kmax - kmin;
            int cshift = 0;
            if (range >= (u64)NB) cshift = (64 - __builtin_clzll(range)) - LOGR;
            uint32_t* const h1 = reinterpret_cast<uint32_t*>(S);
            __syncthreads();  // previous chunk's readers of S / bkt / slots are done
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                h1[q * 64 + lane] = 0u;
                bkt[q * 64 + lane] = 0u;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < RPL; ++u)
                if (val[u]) atomicAdd(&h1[(uint32_t)((key[u] - kmin) >> cshift)], 1u);
            __syncthreads();
            {
                uint32_t mx = 0u;
#pragma unroll
                for (int q = 0; q < PER; ++q) mx = max(mx, h1[lane * PER + q]);
                // Keys of one smooth surface spread evenly over the ranges (448 records in 512 ranges: a range with nine
                // is a one-in-a-million event), and then the second level -- sub-range table, second histogram -- has
                // nothing to add: the level-1 histogram is the bucket histogram.  Depth steps inside a tile take the
                // two-level map below.
                flat = __ballot(mx > (uint32_t)EP_RANK_SELF) == 0ull;
            }
            if (flat) {
#pragma unroll
                for (int u = 0; u < RPL; ++u) bq[u] = val[u] ? (uint32_t)((key[u] - kmin) >> cshift) : 0u;
            } else {
            {
                uint32_t inf[PER];  // sub-ranges | log2(sub-ranges) << 16 of this lane's PER ranges
                uint32_t sum = 0;
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    const uint32_t cc = h1[lane * PER + q];
                    int fl = (cc <= 1u) ? 0 : (31 - __builtin_clz(cc));  // floor(log2 cc)
                    fl = min(fl, cshift);                                 // cannot cut finer than one key value
                    const uint32_t nf = (cc == 0u) ? 0u : (1u << fl);
                    inf[q] = nf | ((uint32_t)fl << 16);
                    sum += nf;
                }
                uint32_t incl = sum;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = __shfl_up(incl, d);
                    if (lane >= d) incl += o;
                }
                uint32_t run = incl - sum;
#pragma unroll
                for (int q = 0; q < PER; ++q) {
                    h1[lane * PER + q] = run | (inf[q] & 0xFFFF0000u);  // first bucket | log2(sub-ranges) << 16
                    run += inf[q] & 0xFFFFu;
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < RPL; ++u) {
                const u64 d = val[u] ? key[u] - kmin : 0ull;  // filtered-out records: range 0
                const uint32_t c = (uint32_t)(d >> cshift);
                const uint32_t info = h1[c];
                const int fl = (int)(info >> 16);
                bq[u] = val[u] ? (info & 0xFFFFu) + (uint32_t)((d - ((u64)c << cshift)) >> (cshift - fl)) : 0u;
            }
            }  // two-level map
        }

        // ---- bucket: histogram, exclusive scan, scatter into S -----------------------------------------------------
        if (!flat) {
#pragma unroll
            for (int u = 0; u < RPL; ++u)
                if (val[u]) atomicAdd(&bkt[bq[u]], 1u);
            __syncthreads();
        }
        uint32_t bigmask = 0u;  // which of this lane's PER buckets hold more than EP_RANK_SELF records
        {
            uint32_t loc[PER];
            uint32_t sum = 0;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                // (flat: the level-1 counts are still where the histogram left them, in the first bytes of S)
                loc[q] = flat ? reinterpret_cast<const uint32_t*>(S)[lane * PER + q] : bkt[lane * PER + q];
                sum += loc[q];
                bigmask |= (loc[q] > EP_RANK_SELF) ? (1u << q) : 0u;
            }
            uint32_t incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            uint32_t run = incl - sum;
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                bkt[lane * PER + q] = run;
                run += loc[q];
            }
        }
        __syncthreads();
        // From here on only the key half of a record stays in registers; the scatter position goes into posf[u].
        // posf[u] = bucket start | scatter position << 10 | bucket length << 20, sign bit = nothing to place
        static_assert(C <= 1024, "positions are packed into 10 bits");
        int posf[RPL];
#pragma unroll
        for (int u = 0; u < RPL; ++u) {
            posf[u] = 0;
            if (val[u]) {
                const uint32_t sp = atomicAdd(&bkt[bq[u]], 1u);
                S[sp] = make_uint4((uint32_t)key[u], (uint32_t)(key[u] >> 32), rx[u], ry[u]);
                posf[u] = (int)(sp << 10);
            }
        }
        __syncthreads();
        // the sweep reads whole blocks of 64: pad the last block with records that never hit and sort last
        for (int i = m + lane; i < ((m + 63) & ~63); i += 64)
            S[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0x7F800000u, 0x7F800000u);
        // bkt[b] is now the end of bucket b and S holds the records in bucket order.  Order inside a bucket:
        //  - up to EP_RANK_SELF records: every record (still in this lane's registers) counts the keys of its own bucket
        //    that are smaller than its key; keys are unique, so start + count is its final position;
        //  - up to EP_RANK_WAVE records: the wave ranks the bucket together afterwards (all pairs: each lane holds up to
        //    two of its records and every key of the bucket is read once, as a broadcast);
        //  - anything longer (a long run of equal depths): bitonic sort of the whole chunk.
        int big = 0;
#pragma unroll
        for (int u = 0; u < RPL; ++u) {
            if (val[u]) {
                const int bs = (bq[u] == 0u) ? 0 : (int)bkt[bq[u] - 1u];
                const int be = (int)bkt[bq[u]];
                posf[u] |= bs | ((be - bs) << 20);
                big = max(big, be - bs);
            } else {
                posf[u] = (int)0x80100000u;
            }
        }
        if (__ballot(big > EP_RANK_WAVE) == 0ull) {
            // The first 4 keys of the bucket are fetched unconditionally (clamped), two records at a time, so that 8
            // independent LDS reads are in flight together; only buckets with more than 4 records loop.
#pragma unroll
            for (int u0 = 0; u0 < RPL; u0 += 2) {
                u64 kk[2][4];
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2) {
                    const int u = (u0 + a2 < RPL) ? u0 + a2 : u0;
                    const int bs = posf[u] & 1023, bsz = (posf[u] >> 20) & 2047;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        kk[a2][j] = lds_key(&S[bs + min(j, bsz - 1)]);
                }
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2) {
                    if (u0 + a2 >= RPL) continue;
                    const int u = u0 + a2;
                    const int bs = posf[u] & 1023, bsz = (posf[u] >> 20) & 2047;
                    int fin = -1;  // negative: nothing to place, or left to the wave-wide pass below
                    if (posf[u] >= 0 && bsz <= EP_RANK_SELF) {
                        int r = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) r += (j < bsz && kk[a2][j] < key[u]) ? 1 : 0;
                        for (int j = 4; j < bsz; ++j) r += (lds_key(&S[bs + j]) < key[u]) ? 1 : 0;
                        fin = bs + r;
                    }
                    posf[u] = (fin < 0) ? -1 : (fin | (posf[u] & (1023 << 10)));
                }
            }
            // every rank is computed: fetch the other half of each record from where the scatter left it, then move
            uint2 zw[RPL];
#pragma unroll
            for (int u = 0; u < RPL; ++u) zw[u] = *reinterpret_cast<const uint2*>(&S[(posf[u] >> 10) & 1023].z);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < RPL; ++u)
                if (posf[u] >= 0)
                    S[posf[u] & 1023] = make_uint4((uint32_t)key[u], (uint32_t)(key[u] >> 32), zw[u].x, zw[u].y);
            // wave-wide pass over the long buckets, one after the other (lane L owns buckets L*PER .. L*PER+PER-1)
            u64 has = __ballot(bigmask != 0u);
            while (has != 0ull) {
                const int L = __builtin_ctzll(has);
                has &= has - 1ull;
                uint32_t mk = (uint32_t)__builtin_amdgcn_readlane((int)bigmask, L);
                while (mk != 0u) {
                    const int b = L * PER + __builtin_ctz(mk);
                    mk &= mk - 1u;
                    const int bs = (b == 0) ? 0 : (int)bkt[b - 1];
                    const int n = (int)bkt[b] - bs;  // EP_RANK_SELF < n <= EP_RANK_WAVE
#ifdef EP_TILE_TIMES
                    ++dbg_wave_buckets;
                    dbg_wave_records += (uint32_t)n;
#endif
                    __syncthreads();
                    uint4 mine[EP_RANK_WAVE / 64];
                    int rank[EP_RANK_WAVE / 64];
#pragma unroll
                    for (int r = 0; r < EP_RANK_WAVE / 64; ++r) {
                        mine[r] = S[bs + min(r * 64 + lane, n - 1)];
                        rank[r] = 0;
                    }
                    for (int i = 0; i < n; ++i) {
                        const u64 ko = lds_key(&S[bs + i]);
#pragma unroll
                        for (int r = 0; r < EP_RANK_WAVE / 64; ++r) rank[r] += (ko < make_key(mine[r].y, mine[r].x)) ? 1 : 0;
                    }
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < EP_RANK_WAVE / 64; ++r)
                        if (r * 64 + lane < n) S[bs + rank[r]] = mine[r];
                }
            }
        } else {
            // Degenerate key distribution (long runs of equal depth): bitonic sort of the whole chunk, written with
            // ascending compare-exchanges only (each merge starts with a flip stage), so that the positions from C up to
            // the next power of two can stay virtual: they would hold "nothing" records, which never move.
            constexpr int P2 = next_pow2(C);
#ifdef EP_TILE_TIMES
            ++dbg_bitonic;
#endif
            for (int i = m + lane; i < C; i += 64) S[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0x7F800000u, 0x7F800000u);
            __syncthreads();
            for (int k2 = 2; k2 <= P2; k2 <<= 1) {
                for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                    for (int t = lane; t < P2 / 2; t += 64) {
                        int lo, hi;
                        if (j2 == (k2 >> 1)) {  // flip: position i of a block of k2 against position k2-1-i
                            const int blk = t / j2, i = t - blk * j2;
                            lo = blk * k2 + i;
                            hi = blk * k2 + k2 - 1 - i;
                        } else {
                            lo = ((t & ~(j2 - 1)) << 1) | (t & (j2 - 1));
                            hi = lo | j2;
                        }
                        if (hi < C) {
                            const uint4 ra = S[lo], rb = S[hi];
                            if (make_key(ra.y, ra.x) > make_key(rb.y, rb.x)) {
                                S[lo] = rb;
                                S[hi] = ra;
                            }
                        }
                    }
                    __syncthreads();
                }
            }
        }
        __syncthreads();
        EP_STAGE(1)
        EP_PROBE(2, S[lane].x ^ S[lane].w)

        // ---- sweep front to back ---------------------------------------------------------------------------------
        int cnt = inside ? 0 : K;  // pixels outside the image never take part
        // Later chunks: a pixel whose running list is full only cares about records with a key below its K-th key,
        // and S is sorted, so those are a prefix S[0, jlim) found by binary search.
        int jlim = m;
        if (chunk_no > 0) {
            const u64 mine = kth_key();
            int lo = 0, hi = m;  // first position whose key is >= mine
            while (__ballot(lo < hi) != 0ull) {
                if (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (lds_key(&S[mid]) < mine)
                        lo = mid + 1;
                    else
                        hi = mid;
                }
            }
            jlim = inside ? lo : 0;
        }
        int m_sweep = jlim;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m_sweep = max(m_sweep, __shfl_xor(m_sweep, d));
        m_sweep = (int)uniform_u32((uint32_t)m_sweep);
        int n_blocks = 0;  // blocks of 64 records that the sweep has offered to the pixels
        static_assert(2 * (C / 64) <= 32, "one bit per mask word");
        uint32_t nz_words = 0u;  // bit w: mask word w of this lane (32 records each) is not empty
        // the tile's eight pixel columns and rows as wave-uniform values (lanes 0..7 hold the columns, every eighth lane a row)
        float xcol[8], yrow[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            xcol[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xf), i));
            yrow[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yf), 8 * i));
        }
        for (int j0 = 0; j0 < m_sweep; j0 += 64) {
            const u64 act = __ballot(cnt < K && j0 < jlim);  // pixels that can still take a record of this block
#ifdef EP_TILE_TIMES
            if (j0 == 0) dbg_s[5] += (unsigned long long)((m_sweep + 63) / 64);
            if (act == 0ull) dbg_s[4] += 1;
            else if (PIXMAJ && __popcll(act) <= PIXMAJ) { dbg_s[1] += 1; dbg_s[2] += (unsigned long long)__popcll(act); }
            else { dbg_s[0] += 1; dbg_s[3] += (unsigned long long)__popcll(act); }
#endif
            if (act == 0ull) break;  // every pixel already holds its K nearest of this chunk
            // Offer records j0 .. j0+63 to every pixel.  Slots past m hold infinite coordinates (written by the sort
            // stage), which never hit.
            uint32_t hm_lo = 0u, hm_hi = 0u;
            if (PIXMAJ && __popcll(act) <= PIXMAJ) {
                // Few pixels are still open: turn the test around.  Every lane holds one record of the block and the
                // open pixels are visited one by one (their coordinates travel through scalar registers); the
                // wave-wide compare result is that pixel's hit mask as it stands and is dropped into its lane.
                const float2v rxy = *reinterpret_cast<const float2v*>(&S[j0 + lane].z);
                u64 a = act;
                while (a != 0ull) {  // two pixels per trip: two independent chains of scalar and vector work
                    const int p = __builtin_ctzll(a);
                    a &= a - 1ull;
                    const int p2 = (a != 0ull) ? __builtin_ctzll(a) : p;  // an odd count visits the last pixel twice
                    a &= a - 1ull;
                    float2v pp, pq;
                    pp.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xf), p));
                    pp.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yf), p));
                    pq.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xf), p2));
                    pq.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yf), p2));
                    const float2v dxy = rxy - pp, exy = rxy - pq;
                    const float2v q = dxy * dxy, q2 = exy * exy;
                    const float d2 = q.x + q.y, e2 = q2.x + q2.y;
                    const u64 hit = __ballot(d2 < r2), hit2 = __ballot(e2 < r2);
                    hm_lo = (lane == p) ? (uint32_t)hit : hm_lo;
                    hm_hi = (lane == p) ? (uint32_t)(hit >> 32) : hm_hi;
                    hm_lo = (lane == p2) ? (uint32_t)hit2 : hm_lo;
                    hm_hi = (lane == p2) ? (uint32_t)(hit2 >> 32) : hm_hi;
                }
            } else {
                // Every lane holds one record of the block.  The squared distance splits into a column term and a row
                // term, (x - xf[c])^2 and (y - yf[r])^2, eight of each per record and rounded exactly as in the direct
                // evaluation; a pixel's test is then one addition and one compare whose wave-wide result is that pixel's
                // hit mask over the 64 records, which two lane writes drop into the pixel's lane.
                const float2v rxy = *reinterpret_cast<const float2v*>(&S[j0 + lane].z);
                float2v dx2[4];
                float dy2[8];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float2v d;
                    d.x = rxy.x - xcol[2 * c];
                    d.y = rxy.x - xcol[2 * c + 1];
                    dx2[c] = d * d;
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float d = rxy.y - yrow[r];
                    dy2[r] = d * d;
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    // a row of pixels is skipped when none of them is open or when no record of the block comes within
                    // the radius of the row (the row term alone is already a lower bound on the squared distance)
                    if (((act >> (8 * r)) & 0xFFull) == 0ull || __ballot(dy2[r] < r2) == 0ull) continue;
                    float2v rowv;
                    rowv.x = dy2[r];
                    rowv.y = dy2[r];
                    const float2v a = dx2[0] + rowv, b = dx2[1] + rowv, c = dx2[2] + rowv, d = dx2[3] + rowv;
                    hit_masks4(r * 8, a.x, a.y, b.x, b.y, r2, hm_lo, hm_hi);  // L must fold to a constant: unrolled
                    hit_masks4(r * 8 + 4, c.x, c.y, d.x, d.y, r2, hm_lo, hm_hi);
                }
            }
            // Hits beyond a pixel's first K are harmless (only the first K set bits of its masks are ever read), and in the
            // first chunk every pixel's prefix is the whole chunk, whose padding never hits: only later chunks have to cut
            // the masks at the end of the pixel's prefix.
            if (chunk_no == 0) {
            } else if (j0 >= jlim) {
                hm_lo = 0u;
                hm_hi = 0u;
            } else if (jlim - j0 < 64) {  // this lane's prefix ends inside the block
                const int nb = jlim - j0;
                if (nb <= 32) {
                    hm_hi = 0u;
                    hm_lo &= (nb == 32) ? 0xFFFFFFFFu : ((1u << nb) - 1u);
                } else {
                    hm_hi &= (1u << (nb - 32)) - 1u;
                }
            }
            // The pixel's hits of this block stay a bit mask: its first K hits overall are its winners, and the merge stage
            // below walks the masks for them (extracting them here, hit by hit with every lane waiting for the one with the
            // most hits, was a third of the kernel's instructions).
            masks[2 * n_blocks][lane] = hm_lo;
            masks[2 * n_blocks + 1][lane] = hm_hi;
            // which of the lane's mask words hold a hit at all: the walk below goes from one such word to the next
            nz_words |= ((hm_lo != 0u) ? 1u : 0u) << (2 * n_blocks);
            nz_words |= ((hm_hi != 0u) ? 2u : 0u) << (2 * n_blocks);
            cnt = min(cnt + (int)(__popc(hm_lo) + __popc(hm_hi)), K);
            ++n_blocks;
        }
        __syncthreads();
        EP_STAGE(2)
        EP_PROBE(3, (uint32_t)cnt + masks[0][lane])

        // ---- merge the chunk's winners into the running list -------------------------------------------------------
        const int won = inside ? cnt : 0;  // at most K, and never more than the lane's hits in the masks
#ifdef EP_TILE_TIMES
        unsigned long long m_t = __builtin_readcyclecounter();
#define EP_MSTAGE(i_)                                                 \
    {                                                                 \
        const unsigned long long now_ = __builtin_readcyclecounter(); \
        dbg_m[i_] += now_ - m_t;                                      \
        m_t = now_;                                                   \
    }
#else
#define EP_MSTAGE(i_)
#endif
        // positions in S of the lane's winners, in ascending key order: the first `won` set bits of its masks
        uint32_t widx[EP_K];
        {
            // Words without a hit are never read: with a few winners spread over a long chunk (the later chunks of a
            // heavy tile) the walk used to spend its time on dependent LDS reads of empty words.
            int row = 0;  // 32 records per mask word
            uint32_t word = 0u, todo = nz_words;
#pragma unroll
            for (int p = 0; p < EP_K; ++p) {
                widx[p] = 0u;
                if (p < won) {
                    if (word == 0u) {  // next word that holds a hit (there is one: the lane has more than p hits in all)
                        row = __builtin_ctz(todo);
                        todo &= todo - 1u;
                        word = masks[row][lane];
                    }
                    widx[p] = (uint32_t)(row * 32 + __builtin_ctz(word));
                    word &= word - 1u;
                }
            }
        }
        EP_MSTAGE(0)
        if (chunk_no == 0) {
#pragma unroll
            for (int k = 0; k < EP_K; ++k) {
                if (k < won) kl[EP_K - 1 - k] = lds_key(&S[widx[k]]);
            }
        } else if (__ballot(won > 0) != 0ull) {
            // list[p] = min(list[p], B[p]), B = this chunk's winners in ascending order: the EP_K smallest keys of both,
            // rising then falling
#pragma unroll
            for (int p0 = 0; p0 < EP_K; p0 += 10) {  // ten at a time: twenty pending reads would not fit the registers
#pragma unroll
                for (int p = p0; p < p0 + 10; ++p) {
                    const u64 w = lds_key(&S[widx[p]]);
                    const u64 kb = (p < won) ? w : ~0ull, ka = kl[p];
                    kl[p] = kb < ka ? kb : ka;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            merge_network_desc(kl);
        }
        ++chunk_no;
        EP_MSTAGE(1)
        if (pos < end) {
            const u64 mine = inside ? kth_key() : 0ull;  // outside pixels never raise the threshold
            T = wave_max_u64(mine);
        }
        EP_MSTAGE(2)
        const int c3 = chunk_no / 3;
        const bool bands_due = chunk_no <= 4 || (chunk_no & (chunk_no - 1)) == 0 || (c3 * 3 == chunk_no && (c3 & (c3 - 1)) == 0);
        if (pos < end && chunk_no > 1 && bands_due) {
            // threshold tables for the next gathers: row / column maxima of the K-th keys, then every band of them
            const u64 mine = inside ? kth_key() : 0ull;
            u64 rmax = mine, cmax = mine;
#pragma unroll
            for (int d = 1; d < 8; d <<= 1) {
                const u64 o = __shfl_xor(rmax, d);
                rmax = o > rmax ? o : rmax;
            }
#pragma unroll
            for (int d = 8; d < 64; d <<= 1) {
                const u64 o = __shfl_xor(cmax, d);
                cmax = o > cmax ? o : cmax;
            }
            u64* const line = reinterpret_cast<u64*>(aux);  // 8 row maxima, 8 column maxima (the hit masks are dead)
            __syncthreads();                                // (the tables' last readers, the previous gather, are long done)
            if ((lane & 7) == 0) line[lane >> 3] = rmax;
            if (lane < 8) line[8 + lane] = cmax;
            __syncthreads();
            u64 rb = 0ull, cb = 0ull;
            for (int i = lane >> 3; i <= (lane & 7); ++i) {
                const u64 a = line[i], b = line[8 + i];
                rb = a > rb ? a : rb;
                cb = b > cb ? b : cb;
            }
            if ((lane >> 3) <= (lane & 7)) {
                rowband[band(lane >> 3, lane & 7)] = rb;
                colband[band(lane >> 3, lane & 7)] = (uint32_t)(cb >> 32);
            }
        }
        EP_MSTAGE(3)
#undef EP_MSTAGE
        EP_STAGE(3)
        EP_SPAN_CHUNK()
    }
#ifdef EP_TILE_TIMES
    if (STOP >= 1 && STOP <= 4) {
        if (STOP == 4 && (uint32_t)(kl[0] ^ kl[19]) == 0x12345u && out.image_u8 != nullptr) out.image_u8[0] = 1;
        return;
    }
#endif
    if (my_slot != EP_NO_SLOT) {
        // Hand this slice's lists over; the wave that finds all the others already there merges them and goes on.  The
        // GPU's eight L2 caches are not coherent with each other, and a device-scope release fence writes a whole L2 back:
        // the lists are stored write-through instead (agent-scope relaxed atomic stores), the wave waits for its stores
        // to drain and only then counts itself in; the merging wave reads every byte of the other slices' lists with
        // agent-scope loads, which pass its own L1.  No release fence; one acquire, in the merging wave only.
        u64* const mine_p = sp.partial + (size_t)my_slot * EP_SLOT_U64 + lane;
#pragma unroll
        for (int k = 0; k < EP_K; ++k) __hip_atomic_store(&mine_p[k * 64], kl[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef EP_CHECK
        {
            u64 h = check_seed(sp.nonce, (uint32_t)tile, slice_no, lane);
#pragma unroll
            for (int k = 0; k < EP_K; ++k) h = check_mix(h, kl[k], k);
            __hip_atomic_store(&mine_p[EP_K * 64], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint32_t before = 0u;
        uint32_t* const done = (tail_q != EP_NO_SLOT) ? &sp.tail_done[tail_q] : &sp.done[tile];
        if (lane == 0) before = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        before = uniform_u32(before);
        if (before != n_slices - 1u) {
            EP_DIAG_FLUSH()
            EP_SPAN_END()
            return;
        }
        // The merging wave alone pays an acquire (an invalidate of its compute unit's L1, nothing on the L2): the loads
        // below pass the L1 anyway, the fence keeps them behind the count and covers placements of several workgroups on a
        // compute unit that the write-through hand-over has not been measured for.
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t first = (tail_q != EP_NO_SLOT) ? sp.max_slots + EP_TAIL_SLICES * tail_q : sp.slot_base[tile];
        for (uint32_t o = 0; o < n_slices; ++o) {
            if (first + o == my_slot) continue;
            const u64* const theirs = sp.partial + (size_t)(first + o) * EP_SLOT_U64 + lane;
#ifdef EP_CHECK
            u64 chk = check_seed(sp.nonce, (uint32_t)tile, o, lane);
#endif
            // their list is descending like ours: read back to front it is the ascending sequence the merge wants
#pragma unroll
            for (int p0 = 0; p0 < EP_K; p0 += 10) {
#pragma unroll
                for (int p = p0; p < p0 + 10; ++p) {
                    const u64 kb = __hip_atomic_load(&theirs[(EP_K - 1 - p) * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef EP_CHECK
                    chk = check_mix(chk, kb, EP_K - 1 - p);  // the very values that enter the merge
#endif
                    const u64 ka = kl[p];
                    kl[p] = kb < ka ? kb : ka;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef EP_CHECK
            {
                const u64 stored = __hip_atomic_load(&theirs[EP_K * 64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const u64 bad = __ballot(stored != chk);
                if (lane == 0) {
                    atomicAdd(&g_check_seen, 1ull);
                    if (bad != 0ull) atomicAdd(&g_check_failed, 1ull);
                }
            }
#endif
            merge_network_desc(kl);
        }
    }
    fill_color_lut(reinterpret_cast<float*>(S), rp, d_flags_a, d_flags_b, lane);  // S is free now
    if (inside) {
        uint32_t az[EP_K], ai[EP_K];  // ascending order for the compositor (a renaming, no data moves)
#pragma unroll
        for (int k = 0; k < EP_K; ++k) {
            az[k] = (uint32_t)(kl[EP_K - 1 - k] >> 32);
            ai[k] = (uint32_t)kl[EP_K - 1 - k];
        }
        // the pixel's position is worked out again from an opaque copy of the lane number, so that the index arithmetic
        // of the output addresses does not stay in registers across the whole chunk loop
        int lane2 = lane;
        asm volatile("" : "+v"(lane2));
        const int xi2 = txi * EP_TILE + (lane2 & 7), yi2 = tyi * EP_TILE + (lane2 >> 3);
        composite_and_store<(WPE >= 4 ? 10 : 20)>(g, rp, ndc, az, ai, xi2, yi2, xf, yf, reinterpret_cast<const float*>(S),
                                                  out);
    }
#ifdef EP_TILE_TIMES
    EP_STAGE(4)
    EP_DIAG_FLUSH()
#endif
    EP_SPAN_END()
}

#define EP_SPLAT_ARGS                                                                                                \
    RasterGeom g, egoprior_raster_params rp, const egoprior_point *__restrict__ ndc,                                 \
        const uint32_t *__restrict__ tile_start, const uint32_t *__restrict__ tile_order,                            \
        const ListEntry *__restrict__ list, uint32_t capacity, const int32_t *__restrict__ d_flags_a,                \
        const int32_t *__restrict__ d_flags_b, SlicePlan sp, SplatOut out
template <int C, int NB, int WPE, int PIXMAJ = EP_PIXMAJ, int STOP = 0>
__global__ void __launch_bounds__(64, WPE) k_splat_sweep(EP_SPLAT_ARGS) {
    splat_sweep_wave<C, NB, WPE, PIXMAJ, STOP>(g, rp, ndc, tile_start, tile_order, list, capacity, d_flags_a, d_flags_b, sp,
                                               out);
}
// The default geometry, 448-entry chunks, is compiled for 144 registers (the attribute counts in units of two on this
// target) instead of the 168 that three waves per SIMD would allow, and takes 11.6 KiB of LDS per wave.  Twelve splat waves
// then leave a compute unit 80 registers per SIMD and 20 KiB of LDS: room for a workgroup of any other kernel of the
// frames in flight (40 to 64 registers per wave, up to 12.5 KiB of LDS), which otherwise waits for splat waves to retire.
// Nothing spills at seven records per lane.
template <>
__global__ void __launch_bounds__(64, 3) __attribute__((amdgpu_num_vgpr(72)))
k_splat_sweep<448, 512, 3, EP_PIXMAJ, 0>(EP_SPLAT_ARGS) {
    splat_sweep_wave<448, 512, 3, EP_PIXMAJ, 0>(g, rp, ndc, tile_start, tile_order, list, capacity, d_flags_a, d_flags_b, sp,
                                                out);
}
// 512-entry chunks (12 KiB of LDS per wave, three values spilled) with the same register budget: option "splat_variant" 5,
// the default before the 448-entry one.
template <>
__global__ void __launch_bounds__(64, 3) __attribute__((amdgpu_num_vgpr(72)))
k_splat_sweep<512, 512, 3, EP_PIXMAJ, 0>(EP_SPLAT_ARGS) {
    splat_sweep_wave<512, 512, 3, EP_PIXMAJ, 0>(g, rp, ndc, tile_start, tile_order, list, capacity, d_flags_a, d_flags_b, sp,
                                                out);
}
#undef EP_SPLAT_ARGS

}  // namespace ep

// ------------------------------------------------------------------------------------------------------------------
// host side of the C ABI for this file
// ------------------------------------------------------------------------------------------------------------------
#include <algorithm>
#include <atomic>

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
    // a disc's candidate pixels on one axis are the integers of an interval 2 * rad_px long
    const double bx = std::ceil((2.0 * g.rad_px_x + 1.0) / EP_TILE) + 1.0;
    const double by = std::ceil((2.0 * g.rad_px_y + 1.0) / EP_TILE) + 1.0;
    g.span_x = (int)std::min<double>(bx, g.tiles_x);
    g.span_y = (int)std::min<double>(by, g.tiles_y);
    return g;
}

// upper bound on tiles one disc can overlap
int64_t tiles_per_point_bound(const ep::RasterGeom& g) { return (int64_t)g.span_x * (int64_t)g.span_y; }

// list positions are 32-bit: refuse a cloud whose discs could need more entries than that
int check_list_size(const ep::RasterGeom& g, int64_t n) {
    if ((uint64_t)n * (uint64_t)tiles_per_point_bound(g) > 0xFFFFFFF0ull)
        return ep_host::fail(EGOPRIOR_E_INVALID, "too many point-tile pairs for one call (points x tiles per disc > 2^32)");
    return 0;
}

// The tuning switches a call works with, read once when the call starts (egoprior_set_option may change them from
// another thread at any time; a call in progress keeps the values it started with).
struct Tuning {
    int splat_variant, fill_pieces, bin_lds, slices, scan_threads, scan_once, tail_slots;
};
Tuning read_tuning() {
    Tuning t;
    t.splat_variant = ep_host::option("splat_variant");
    t.fill_pieces = ep_host::option("fill_pieces");
    t.bin_lds = ep_host::option("bin_lds");
    t.slices = ep_host::option("slices");
    t.scan_threads = ep_host::option("scan_threads");
    t.scan_once = ep_host::option("scan_once");
    t.tail_slots = ep_host::option("tail_slots");
    return t;
}

// Workspace layout: header (per-tile words) | pieces table (optional) | tile lists (8-byte entries) | slice area
// (per-tile slot words, slot map, per-slice pixel lists).  Lists and slice area share what the header and the pieces
// table leave: the lists get the largest capacity whose slice area (enough slots for any distribution of that many
// entries) still fits.
struct RasterWs {
    uint32_t* tile_count;   // n_tiles
    uint32_t* tile_start;   // n_tiles + 1
    uint32_t* tile_cursor;  // n_tiles
    uint32_t* tile_order;   // n_tiles
    int32_t* total;         // 1 (+ padding)
    uint32_t* pieces;       // workgroups x tiles words: where each counting workgroup's entries start inside each tile's
                            // list (k_tile_fill_pieces); NULL when that pass is not used
    ep::ListEntry* list;    // capacity entries
    uint64_t capacity;
    ep::SlicePlan slices;   // max_slots == 0: the workspace has no room for it
};

// Slots that `entries` list entries can need at most: a sliced tile has more than 2 * EP_SLICE_LEN entries and takes
// ceil(entries of the tile / EP_SLICE_LEN) slots, fewer than 1.5 slots per EP_SLICE_LEN entries.
constexpr uint64_t kWaveSlots = 256 * 12;  // MI355X: 256 compute units, three splat waves on each of their four SIMDs
uint64_t heavy_slots_for(uint64_t entries) { return entries * 3 / (2 * (uint64_t)EP_SLICE_LEN) + 2; }
// ... and the tail of a launch: at most half of the wave slots' worth of tiles, each of which holds EP_TAIL_MIN entries or more
uint64_t tail_tiles_for(uint64_t entries) { return std::min<uint64_t>(kWaveSlots / 2, entries / EP_TAIL_MIN); }
constexpr size_t kSlotBytes = (size_t)EP_SLOT_U64 * sizeof(unsigned long long);
size_t slice_area_bytes(int n_tiles, uint64_t entries) {
    const uint64_t heavy = heavy_slots_for(entries), tail = tail_tiles_for(entries);
    return (sizeof(uint32_t) * (2 * (size_t)n_tiles + 64) + 255) / 256 * 256 + (sizeof(uint32_t) * heavy + 255) / 256 * 256 +
           (sizeof(uint32_t) * (tail + 1) + 255) / 256 * 256 + kSlotBytes * (heavy + EP_TAIL_SLICES * tail);
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

size_t raster_header_bytes(int n_tiles) { return align_up(sizeof(uint32_t) * (4 * (size_t)n_tiles + 1 + 16), 256); }

// the counting pass aggregates per workgroup in LDS (two 16-bit counters per word) unless option "bin_lds" is 0
size_t bin_lds_bytes(int n_tiles) { return sizeof(uint32_t) * (size_t)((n_tiles + 1) >> 1); }
// The LDS histogram has 16-bit counters: it is only used while no workgroup can see 65536 points (the grid stops
// growing at 4096 workgroups, which then loop over several blocks of EP_BIN_PPB points each).
bool lds_agg_possible(int n_tiles, int64_t n) {
    return bin_lds_bytes(n_tiles) <= 64 * 1024 &&
           (n + EP_BIN_PPB - 1) / EP_BIN_PPB <= (int64_t)4096 * (65535 / EP_BIN_PPB);
}
static_assert(EP_BIN_PPB < 65536, "a workgroup's per-tile count must fit into 16 bits");

int bin_grid(int64_t n) {
    int64_t b = (n + EP_BIN_PPB - 1) / EP_BIN_PPB;
    if (b < 1) b = 1;
    if (b > 4096) b = 4096;
    return (int)b;
}

// the fill pass that works from the pieces table keeps one 16-bit cursor per tile in LDS
size_t fill_lds_bytes(int n_tiles) { return sizeof(uint32_t) * (size_t)((n_tiles + 1) / 2); }

// bytes of the pieces table for n points (0 when the counting pass cannot aggregate in LDS for this size, or the fill
// pass cannot hold its cursors there)
size_t pieces_bytes(int n_tiles, int64_t n) {
    if (!lds_agg_possible(n_tiles, n) || fill_lds_bytes(n_tiles) > 64 * 1024) return 0;
    return align_up((size_t)bin_grid(n) * (size_t)n_tiles * sizeof(uint32_t), 256);
}

size_t raster_ws_bytes_for(int n_tiles, int64_t n, uint64_t entries) {
    return raster_header_bytes(n_tiles) + pieces_bytes(n_tiles, n) + align_up((size_t)entries * sizeof(ep::ListEntry), 256) +
           slice_area_bytes(n_tiles, entries) + 256;
}

bool carve_raster_ws(void* ws, size_t bytes, int n_tiles, int64_t n, const Tuning& tn, RasterWs& r) {
    const size_t hb = raster_header_bytes(n_tiles);
    if (ws == nullptr || bytes < hb + 256) return false;
    uint32_t* base = (uint32_t*)ws;
    r.tile_count = base;
    r.tile_start = base + n_tiles;
    r.tile_cursor = base + 2 * (size_t)n_tiles + 1;
    r.tile_order = base + 3 * (size_t)n_tiles + 1;
    r.total = (int32_t*)(base + 4 * (size_t)n_tiles + 1);
    size_t off = hb;
    r.pieces = nullptr;
    const size_t pb = pieces_bytes(n_tiles, n);
    // the table is used when it is wanted and leaves at least as much room for the lists as it takes itself
    if (pb != 0 && tn.fill_pieces != 0 && tn.bin_lds != 0 && bytes >= hb + 2 * pb + 256) {
        r.pieces = (uint32_t*)((char*)ws + off);
        off += pb;
    }
    r.list = (ep::ListEntry*)((char*)ws + off);
    const size_t rest = bytes - off;
    uint64_t cap = rest / sizeof(ep::ListEntry);
    if (cap > 0xFFFFFFF0ull) cap = 0xFFFFFFF0ull;
    r.slices = ep::SlicePlan{};
    if (tn.slices != 0 && tn.splat_variant != 1) {
        // largest capacity whose lists and slice area both fit (the need grows with the capacity: bisection)
        auto need = [&](uint64_t c) { return align_up((size_t)c * sizeof(ep::ListEntry), 256) + slice_area_bytes(n_tiles, c); };
        if (need(64) <= rest) {
            uint64_t lo = 64, hi = cap;  // need(lo) fits
            while (lo < hi) {
                const uint64_t mid = lo + (hi - lo + 1) / 2;
                if (need(mid) <= rest)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            cap = lo;
            const uint64_t slots = heavy_slots_for(cap), tail = tail_tiles_for(cap);
            char* q = (char*)ws + off + align_up((size_t)cap * sizeof(ep::ListEntry), 256);
            r.slices.slot_base = (uint32_t*)q;
            r.slices.done = r.slices.slot_base + n_tiles;
            r.slices.total = r.slices.done + n_tiles;
            q += align_up(sizeof(uint32_t) * (2 * (size_t)n_tiles + 64), 256);
            r.slices.map = (uint32_t*)q;
            q += align_up(sizeof(uint32_t) * slots, 256);
            r.slices.tail_done = (uint32_t*)q;
            q += align_up(sizeof(uint32_t) * (tail + 1), 256);
            r.slices.partial = (unsigned long long*)q;
            r.slices.max_slots = (uint32_t)(slots > 0x00FFFFFFull ? 0x00FFFFFFull : slots);
            r.slices.wave_slots = tn.tail_slots > 0 ? (uint32_t)tn.tail_slots : 0u;
            r.slices.max_tail = r.slices.wave_slots != 0u ? (uint32_t)tail : 0u;
#ifdef EP_CHECK
            static std::atomic<uint32_t> launch_nonce{1u};
            r.slices.nonce = launch_nonce.fetch_add(1u);
#endif
        }
    }
    r.capacity = cap;
    return true;
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

__global__ void k_fill_f32(float* __restrict__ p, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// the number of list entries a frame needs, widened for the caller (tile_start[n_tiles] after k_tile_scan)
__global__ void k_copy_total(const uint32_t* __restrict__ total, int64_t* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *out = (int64_t)(*total);
}

// nothing to draw: the frame is the background and every fragment slot is empty
int clear_outputs(int H, int W, int K, const ep::SplatOut& out, hipStream_t s) {
    const size_t npix = (size_t)H * (size_t)W;
    bool ok = true;
    if (out.image_u8 != nullptr) ok = ok && hipMemsetAsync(out.image_u8, 0, npix * 3, s) == hipSuccess;
    if (out.image_f32 != nullptr) ok = ok && hipMemsetAsync(out.image_f32, 0, npix * 3 * sizeof(float), s) == hipSuccess;
    if (out.idx != nullptr) ok = ok && hipMemsetAsync(out.idx, 0xFF, npix * K * sizeof(int32_t), s) == hipSuccess;
    if (!ok) return ep_host::fail(EGOPRIOR_E_HIP, "memset failed");
    const size_t nk = npix * (size_t)K;
    if (out.zbuf != nullptr)
        EP_LAUNCH("k_fill_f32", k_fill_f32, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, out.zbuf, nk, -1.0f);
    if (out.dists != nullptr)
        EP_LAUNCH("k_fill_f32", k_fill_f32, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, out.dists, nk, -1.0f);
    return ep_host::check_launch("clear_outputs");
}

// Projection (cam != NULL: [a | b] are world points and the NDC records go to d_ndc; cam == NULL: `a` already holds NDC
// records) and the per-tile counts.  Zeroes the counters first.
int launch_count(const egoprior_camera* cam, const ep::RasterGeom& g, const egoprior_point* d_a, int64_t n_a,
                 const int32_t* d_n_a, const egoprior_point* d_b, int64_t n_b, const int32_t* d_n_b,
                 egoprior_point* d_ndc, const RasterWs& w, const Tuning& tn, hipStream_t s) {
    const int n_tiles = g.tiles_x * g.tiles_y;
    const int64_t n = n_a + n_b;
    if (hipMemsetAsync(w.tile_count, 0, sizeof(uint32_t) * (size_t)n_tiles, s) != hipSuccess)
        return ep_host::fail(EGOPRIOR_E_HIP, "memset failed");
    const bool lds = tn.bin_lds != 0 && lds_agg_possible(n_tiles, n);
    const size_t shm = lds ? bin_lds_bytes(n_tiles) : 0;
    uint32_t* const pieces = lds ? w.pieces : nullptr;
    egoprior_camera cam_unused = {};
    if (cam != nullptr) {
        if (lds)
            EP_LAUNCH("k_project_count", (ep::k_project_count<true, true>), dim3(bin_grid(n)), dim3(256), shm, s, *cam, g,
                      d_a, n_a, d_n_a, n_b > 0 ? d_b : nullptr, n_b, d_n_b, d_ndc, w.total, w.tile_count, 1, pieces);
        else
            EP_LAUNCH("k_project_count", (ep::k_project_count<true, false>), dim3(bin_grid(n)), dim3(256), 0, s, *cam, g,
                      d_a, n_a, d_n_a, n_b > 0 ? d_b : nullptr, n_b, d_n_b, d_ndc, w.total, w.tile_count, 1, pieces);
    } else {
        if (lds)
            EP_LAUNCH("k_project_count", (ep::k_project_count<false, true>), dim3(bin_grid(n)), dim3(256), shm, s,
                      cam_unused, g, d_a, n_a, d_n_a, (const egoprior_point*)nullptr, (int64_t)0,
                      (const int32_t*)nullptr, (egoprior_point*)nullptr, w.total, w.tile_count, 1, pieces);
        else
            EP_LAUNCH("k_project_count", (ep::k_project_count<false, false>), dim3(bin_grid(n)), dim3(256), 0, s,
                      cam_unused, g, d_a, n_a, d_n_a, (const egoprior_point*)nullptr, (int64_t)0,
                      (const int32_t*)nullptr, (egoprior_point*)nullptr, w.total, w.tile_count, 1, pieces);
    }
    return ep_host::check_launch("project_count");
}

// count (optionally with projection) must already have run into w.tile_count
int launch_scan_fill_splat(const ep::RasterGeom& g, const egoprior_raster_params* rp, const egoprior_point* d_ndc,
                           int64_t n, const int32_t* d_n, const int32_t* d_flags_a, const int32_t* d_flags_b,
                           const RasterWs& w, const Tuning& tn, ep::SplatOut out, int32_t* d_status, hipStream_t s) {
    const int n_tiles = g.tiles_x * g.tiles_y;
    // 256 threads (one wave per SIMD, 62 VGPRs) find room on a compute unit whose SIMDs are full of the splat waves of the
    // frames in flight; 1024 threads are 3 us faster on an idle GPU (option "scan_threads")
    if (tn.scan_threads != 1024 && n_tiles <= 256 * 16 && tn.scan_once != 0)
        EP_LAUNCH("k_tile_scan", (ep::k_tile_scan<256, 16, true>), dim3(1), dim3(256), 0, s, w.tile_count, w.tile_start,
                  w.tile_cursor, w.tile_order, n_tiles, (uint32_t)w.capacity, d_status, w.slices);
    else if (tn.scan_threads != 1024 && n_tiles <= 256 * 32 && tn.scan_once != 0)
        EP_LAUNCH("k_tile_scan", (ep::k_tile_scan<256, 32, true>), dim3(1), dim3(256), 0, s, w.tile_count, w.tile_start,
                  w.tile_cursor, w.tile_order, n_tiles, (uint32_t)w.capacity, d_status, w.slices);
    else if (tn.scan_threads != 1024)
        EP_LAUNCH("k_tile_scan", ep::k_tile_scan<256>, dim3(1), dim3(256), 0, s, w.tile_count, w.tile_start,
                  w.tile_cursor, w.tile_order, n_tiles, (uint32_t)w.capacity, d_status, w.slices);
    else
        EP_LAUNCH("k_tile_scan", ep::k_tile_scan<1024>, dim3(1), dim3(1024), 0, s, w.tile_count, w.tile_start,
                  w.tile_cursor, w.tile_order, n_tiles, (uint32_t)w.capacity, d_status, w.slices);
    const bool lds = tn.bin_lds != 0 && lds_agg_possible(n_tiles, n);
    if (lds && w.pieces != nullptr)
        EP_LAUNCH("k_tile_fill_pieces", ep::k_tile_fill_pieces, dim3(bin_grid(n)), dim3(256), fill_lds_bytes(n_tiles), s, g,
                  d_ndc, n, d_n, w.tile_start, w.pieces, w.list, (uint32_t)w.capacity);
    else
        EP_LAUNCH("k_tile_fill", ep::k_tile_fill, dim3(bin_grid(n)), dim3(256), 0, s, g, d_ndc, n, d_n, w.tile_start,
                  w.tile_cursor, w.list, (uint32_t)w.capacity);
#define EP_SWEEP(C_, NB_, WPE_)                                                                                \
    EP_LAUNCH("k_splat_sweep", (ep::k_splat_sweep<C_, NB_, WPE_>), dim3(n_tiles + 2 * w.slices.max_tail + w.slices.max_slots), dim3(64), 0, s, g, \
              *rp, d_ndc, w.tile_start, w.tile_order, w.list, (uint32_t)w.capacity, d_flags_a, d_flags_b, w.slices, out)
    switch (tn.splat_variant) {
        case 1:  // plain register insertion: the simple cross-check
            EP_LAUNCH("k_splat_bubble", ep::k_splat_bubble, dim3(n_tiles), dim3(64), 0, s, g, *rp, d_ndc, w.tile_start,
                      w.tile_order, w.list, (uint32_t)w.capacity, d_flags_a, d_flags_b, out);
            break;
        case 2:  // tiny chunks: forces the multi-chunk path on small inputs (tests)
            EP_SWEEP(64, 64, 2);
            break;
        case 3:  // 1024-record chunks, two waves per SIMD
            EP_SWEEP(1024, 1024, 2);
            break;
        case 4:  // 384-record chunks, 8.6 KiB of LDS per wave, four waves per SIMD
            EP_SWEEP(384, 512, 4);
            break;
        case 5:  // 512-record chunks, 12 KiB of LDS per wave, three waves per SIMD (the default before the 448-record one)
            EP_SWEEP(512, 512, 3);
            break;
#ifdef EP_TILE_TIMES
#define EP_PROBE_CASE(v_, stop_)                                                                                  \
    case v_:                                                                                                       \
        EP_LAUNCH("k_splat_sweep", (ep::k_splat_sweep<448, 512, 3, EP_PIXMAJ, stop_>), dim3(n_tiles + 2 * w.slices.max_tail + w.slices.max_slots),    \
                  dim3(64), 0, s, g, *rp, d_ndc, w.tile_start, w.tile_order, w.list, (uint32_t)w.capacity, d_flags_a,  \
                  d_flags_b, w.slices, out);                                                                         \
        break;
            EP_PROBE_CASE(11, 1)
            EP_PROBE_CASE(12, 2)
            EP_PROBE_CASE(13, 3)
            EP_PROBE_CASE(14, 4)
#undef EP_PROBE_CASE
#endif
        default:  // 448-record chunks, 11.6 KiB of LDS per wave, three waves per SIMD
            EP_SWEEP(448, 512, 3);
    }
#undef EP_SWEEP
    return ep_host::check_launch("rasterize");
}

}  // namespace

extern "C" size_t egoprior_raster_workspace_bytes(int32_t H, int32_t W, int64_t n, float radius) {
    if (H <= 0 || W <= 0 || n < 0 || !(radius > 0.0f)) return 0;
    const ep::RasterGeom g = make_geom(H, W, radius);
    const int n_tiles = g.tiles_x * g.tiles_y;
    const uint64_t cap = (uint64_t)(n > 0 ? n : 1) * (uint64_t)tiles_per_point_bound(g);
    return raster_ws_bytes_for(n_tiles, n > 0 ? n : 1, cap);
}

extern "C" size_t egoprior_raster_workspace_bytes_entries(int32_t H, int32_t W, int64_t n, int64_t entries) {
    if (H <= 0 || W <= 0 || n < 0 || entries < 0) return 0;
    const int n_tiles = ((W + EP_TILE - 1) / EP_TILE) * ((H + EP_TILE - 1) / EP_TILE);
    return raster_ws_bytes_for(n_tiles, n > 0 ? n : 1, (uint64_t)entries);
}

extern "C" size_t egoprior_render_workspace_bytes(int32_t H, int32_t W, int64_t n, float radius) {
    const size_t r = egoprior_raster_workspace_bytes(H, W, n, radius);
    if (r == 0) return 0;
    return r + align_up((size_t)(n > 0 ? n : 1) * sizeof(egoprior_point), 256);
}

extern "C" size_t egoprior_render_workspace_bytes_entries(int32_t H, int32_t W, int64_t n, int64_t entries) {
    const size_t r = egoprior_raster_workspace_bytes_entries(H, W, n, entries);
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
    EP_LAUNCH("k_project_count", (ep::k_project_count<true, false>), dim3(bin_grid(n_a + n_b)), dim3(256), 0,
              (hipStream_t)stream, *cam, g, d_a, n_a, d_n_a, n_b > 0 ? d_b : nullptr, n_b, d_n_b, d_ndc,
              (int32_t*)nullptr, (uint32_t*)nullptr, 0, (uint32_t*)nullptr);
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
    if (n == 0) return clear_outputs(H, W, rp->K, ep::SplatOut{d_image_u8, d_image_f32, d_idx, d_zbuf, d_dists}, s);
    const ep::RasterGeom g = make_geom(H, W, rp->radius);
    const int n_tiles = g.tiles_x * g.tiles_y;
    if (int e = check_list_size(g, n)) return e;
    const Tuning tn = read_tuning();
    RasterWs w;
    if (!carve_raster_ws(d_workspace, workspace_bytes, n_tiles, n, tn, w))
        return ep_host::fail(EGOPRIOR_E_WORKSPACE, "raster workspace too small");
    if (int e = launch_count(nullptr, g, d_ndc, n, d_n, nullptr, 0, nullptr, nullptr, w, tn, s)) return e;
    ep::SplatOut out = {d_image_u8, d_image_f32, d_idx, d_zbuf, d_dists};
    return launch_scan_fill_splat(g, rp, d_ndc, n, d_n, d_flags_a, d_flags_b, w, tn, out, d_status, s);
}

namespace {
int check_render_args(const egoprior_camera* cam, const egoprior_raster_params* rp, const egoprior_point* d_a,
                      int64_t n_a, const egoprior_point* d_b, int64_t n_b) {
    if (cam == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "camera is NULL");
    if (cam->model != EGOPRIOR_CAM_PINHOLE && cam->model != EGOPRIOR_CAM_FISHEYE)
        return ep_host::fail(EGOPRIOR_E_INVALID, "unknown camera model");
    if (n_a < 0 || n_b < 0 || (n_a > 0 && d_a == nullptr) || (n_b > 0 && d_b == nullptr))
        return ep_host::fail(EGOPRIOR_E_INVALID, "bad cloud pointers / sizes");
    return check_raster_args(rp, cam->H, cam->W, n_a + n_b);
}
}  // namespace

extern "C" int egoprior_render_frame(const egoprior_camera* cam, const egoprior_raster_params* rp,
                                     const egoprior_point* d_a, int64_t n_a, const int32_t* d_n_a,
                                     const int32_t* d_flags_a, const egoprior_point* d_b, int64_t n_b,
                                     const int32_t* d_n_b, const int32_t* d_flags_b, uint8_t* d_image_u8,
                                     float* d_image_f32, int32_t* d_idx, float* d_zbuf, float* d_dists,
                                     int32_t* d_status, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (int e = check_render_args(cam, rp, d_a, n_a, d_b, n_b)) return e;
    const int64_t n = n_a + n_b;
    if ((d_zbuf != nullptr || d_dists != nullptr) && d_idx == nullptr)
        return ep_host::fail(EGOPRIOR_E_INVALID, "zbuf/dists need idx");
    hipStream_t s = (hipStream_t)stream;
    const int H = cam->H, W = cam->W;
    if (n == 0) return clear_outputs(H, W, rp->K, ep::SplatOut{d_image_u8, d_image_f32, d_idx, d_zbuf, d_dists}, s);
    const ep::RasterGeom g = make_geom(H, W, rp->radius);
    const int n_tiles = g.tiles_x * g.tiles_y;
    if (int e = check_list_size(g, n)) return e;
    const size_t ndc_bytes = align_up((size_t)(n > 0 ? n : 1) * sizeof(egoprior_point), 256);
    if (d_workspace == nullptr || workspace_bytes < ndc_bytes)
        return ep_host::fail(EGOPRIOR_E_WORKSPACE, "render workspace too small");
    egoprior_point* d_ndc = (egoprior_point*)d_workspace;
    const Tuning tn = read_tuning();
    RasterWs w;
    if (!carve_raster_ws((char*)d_workspace + ndc_bytes, workspace_bytes - ndc_bytes, n_tiles, n, tn, w))
        return ep_host::fail(EGOPRIOR_E_WORKSPACE, "render workspace too small");
    if (int e = launch_count(cam, g, d_a, n_a, d_n_a, d_b, n_b, d_n_b, d_ndc, w, tn, s)) return e;
    ep::SplatOut out = {d_image_u8, d_image_f32, d_idx, d_zbuf, d_dists};
    return launch_scan_fill_splat(g, rp, d_ndc, n, w.total, d_flags_a, d_flags_b, w, tn, out, d_status, s);
}

extern "C" int egoprior_count_list_entries(const egoprior_camera* cam, const egoprior_raster_params* rp, int32_t H,
                                           int32_t W, const egoprior_point* d_a, int64_t n_a, const int32_t* d_n_a,
                                           const egoprior_point* d_b, int64_t n_b, const int32_t* d_n_b,
                                           int64_t* d_entries, void* d_workspace, size_t workspace_bytes,
                                           void* stream) {
    if (d_entries == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "d_entries is NULL");
    if (cam != nullptr) {
        if (int e = check_render_args(cam, rp, d_a, n_a, d_b, n_b)) return e;
        H = cam->H;
        W = cam->W;
    } else {
        if (n_b != 0 || d_b != nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "points in NDC come as one part");
        if (int e = check_raster_args(rp, H, W, n_a)) return e;
        if (n_a > 0 && d_a == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "d_a is NULL");
    }
    const int64_t n = n_a + n_b;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) {
        if (hipMemsetAsync(d_entries, 0, sizeof(int64_t), s) != hipSuccess)
            return ep_host::fail(EGOPRIOR_E_HIP, "memset failed");
        return 0;
    }
    const ep::RasterGeom g = make_geom(H, W, rp->radius);
    const int n_tiles = g.tiles_x * g.tiles_y;
    if (int e = check_list_size(g, n)) return e;
    Tuning tn = read_tuning();
    tn.fill_pieces = 0;  // nothing is filled: the header is all this call needs
    tn.slices = 0;
    RasterWs w;
    if (!carve_raster_ws(d_workspace, workspace_bytes, n_tiles, n, tn, w))
        return ep_host::fail(EGOPRIOR_E_WORKSPACE, "workspace too small (egoprior_raster_workspace_bytes_entries(H, W, n, 0))");
    if (int e = launch_count(cam, g, d_a, n_a, d_n_a, d_b, n_b, d_n_b, nullptr, w, tn, s)) return e;
    EP_LAUNCH("k_tile_scan", ep::k_tile_scan<1024>, dim3(1), dim3(1024), 0, s, w.tile_count, w.tile_start, w.tile_cursor,
              w.tile_order, n_tiles, 0xFFFFFFFFu, (int32_t*)nullptr,
              ep::SlicePlan{});
    EP_LAUNCH("k_copy_total", k_copy_total, dim3(1), dim3(64), 0, s, w.tile_start + n_tiles, d_entries);
    return ep_host::check_launch("count_list_entries");
}

#ifdef EP_CHECK
// Diagnostic build only: slice hand-overs whose checksum the merging wave verified, and how many of them did not match.
extern "C" int egoprior_debug_check_counts(unsigned long long* seen, unsigned long long* failed, int reset) {
    if (hipMemcpyFromSymbol(seen, HIP_SYMBOL(ep::g_check_seen), sizeof(*seen)) != hipSuccess) return -3;
    if (hipMemcpyFromSymbol(failed, HIP_SYMBOL(ep::g_check_failed), sizeof(*failed)) != hipSuccess) return -3;
    if (reset) {
        const unsigned long long z = 0ull;
        if (hipMemcpyToSymbol(HIP_SYMBOL(ep::g_check_seen), &z, sizeof(z)) != hipSuccess) return -3;
        if (hipMemcpyToSymbol(HIP_SYMBOL(ep::g_check_failed), &z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
#endif
#ifdef EP_TILE_TIMES
extern "C" int egoprior_debug_stage_cycles(unsigned long long* host32, int reset) {
    if (hipMemcpyFromSymbol(host32, HIP_SYMBOL(ep::g_stage_cycles), 32 * sizeof(unsigned long long)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(ep::g_stage_cycles), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
extern "C" int egoprior_debug_tile_times(uint32_t* host, int n_words) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ep::g_tile_times), sizeof(uint32_t) * (size_t)n_words) == hipSuccess ? 0 : -3;
}
#endif
#if defined(EP_TILE_TIMES) || defined(EP_WAVE_SPANS)
// (start, end, list entries, chunks) of the first n_waves <= 16384 waves of the last splat launch; reset = 1 clears them
extern "C" int egoprior_debug_wave_chunk_ends(uint32_t* host, int n_waves) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(ep::g_wave_chunk_end), sizeof(uint4) * (size_t)n_waves) == hipSuccess ? 0 : -3;
}
extern "C" int egoprior_debug_wave_spans(uint32_t* host, int n_waves, int reset) {
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(ep::g_wave_span), sizeof(uint4) * (size_t)n_waves) != hipSuccess) return -3;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(ep::g_wave_span)) != hipSuccess) return -3;
        if (hipMemset(p, 0, sizeof(uint4) * 16384) != hipSuccess) return -3;
    }
    return 0;
}
#endif
