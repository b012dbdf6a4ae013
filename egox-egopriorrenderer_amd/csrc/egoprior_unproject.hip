// Exo-frame side of the path for gfx950: reliability mask, order-preserving compaction, fused unprojection
// (rays * depth -> exo c2w -> float64 exo->world), render-entry filters and the mean-background reduction.
//
// All of these stream their inputs once: they are HBM-bound.  The 5x5 reliability stencil goes through an LDS tile
// with a 2-pixel halo so each depth value is fetched from memory once per tile; everything else is a coalesced
// element-wise pass.  Compaction keeps row-major order (block counts -> one-workgroup scan -> scatter) because the
// reference's point indices -- the tie-breaker of the rasterizer -- are positions in the compacted arrays.
#include "egoprior_common.h"
#include "egoprior_host.h"

#include <hip/hip_fp16.h>

#include <mutex>

namespace ep {

// Depth maps come as float32 or, as the EXR files store them (vipe/utils/io.py:274-277), as float16; the conversion to
// float32 is exact, so both give the same bits downstream.
__device__ __forceinline__ float ldd(const float* p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldd(const __half* p, size_t i) { return __half2float(p[i]); }

// ------------------------------------------------------------------------------------------------------------------
// 5x5 reliability test on an LDS tile (vipe/utils/depth.py:385-422)
// ------------------------------------------------------------------------------------------------------------------
// tile: clamped-coordinate copy of the depth around the block, row stride `ts`; (lx, ly) = position of the window's
// top-left tap in the tile; (u, v) = pixel coordinates of the window centre.  Border taps outside the image are
// ignored by max/min and count as zero in the 25-tap sum, which is always divided by 25.  The sum adds the taps one by
// one, rows first, like ATen's CPU avg_pool2d.
__device__ __forceinline__ bool reliable_at(const float* __restrict__ tile, int ts, int lx, int ly, int u, int v, int H,
                                            int W) {
    const bool interior = (u >= 2) && (v >= 2) && (u + 2 < W) && (v + 2 < H);
    float mx = -INFINITY, mn = INFINITY, acc = 0.0f;
    bool any_nan = false;
    if (interior) {
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                const float d = tile[(ly + kh) * ts + lx + kw];
                any_nan |= (d != d);
                mx = fmaxf(mx, d);
                mn = fminf(mn, d);
                acc = acc + d;
            }
        }
    } else {
#pragma unroll
        for (int kh = 0; kh < 5; ++kh) {
#pragma unroll
            for (int kw = 0; kw < 5; ++kw) {
                const int yy = v + kh - 2, xx = u + kw - 2;
                if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
                    const float d = tile[(ly + kh) * ts + lx + kw];
                    any_nan |= (d != d);
                    mx = fmaxf(mx, d);
                    mn = fminf(mn, d);
                    acc = acc + d;
                }
            }
        }
    }
    if (any_nan) return false;  // torch's max pool propagates NaN, so does the sum: ratio is NaN
    const float centre = tile[(ly + 2) * ts + lx + 2];
    const float mean = acc / 25.0f;
    const float ratio = (mx - mn) / (mean + 1e-6f);
    return (ratio < 0.1f) && (centre > 0.0f);
}

// cooperative clamped load of a (fw x fh) footprint whose top-left pixel is (x0, y0)
template <typename DT>
__device__ __forceinline__ void load_footprint(float* __restrict__ tile, const DT* __restrict__ depth, int H, int W,
                                               int x0, int y0, int fw, int fh) {
    for (int e = threadIdx.x; e < fw * fh; e += blockDim.x) {
        const int ly = e / fw, lx = e - ly * fw;
        int yy = y0 + ly, xx = x0 + lx;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        tile[e] = ldd(depth, (size_t)yy * W + xx);
    }
}

#define EP_MASK_SX 32
#define EP_MASK_SY 8

// flags[j*Ws + i] = reliable(depth)[s*j, s*i] && select(mask[s*j, s*i]); one thread per sample, 32x8 samples per
// block.  Dynamic LDS: (31*s+5)*(7*s+5) floats.
template <typename DT>
__global__ void __launch_bounds__(256) k_reliable_flags(const DT* __restrict__ depth,
                                                        const uint8_t* __restrict__ mask, int H, int W, int s, int Hs,
                                                        int Ws, int select, uint8_t* __restrict__ flags) {
    extern __shared__ float tile[];
    const int fw = (EP_MASK_SX - 1) * s + 5, fh = (EP_MASK_SY - 1) * s + 5;
    const int i0 = blockIdx.x * EP_MASK_SX, j0 = blockIdx.y * EP_MASK_SY;
    load_footprint(tile, depth, H, W, i0 * s - 2, j0 * s - 2, fw, fh);
    __syncthreads();
    const int li = threadIdx.x & (EP_MASK_SX - 1), lj = threadIdx.x / EP_MASK_SX;
    const int i = i0 + li, j = j0 + lj;
    if (i >= Ws || j >= Hs) return;
    const int u = i * s, v = j * s;
    bool ok = reliable_at(tile, fw, li * s, lj * s, u, v, H, W);
    if (select != EGOPRIOR_SELECT_ALL) {
        if (mask != nullptr) {
            const uint8_t m = mask[(size_t)v * W + u];
            ok = ok && ((select == EGOPRIOR_SELECT_STATIC) ? (m == 0) : (m != 0));
        } else if (select == EGOPRIOR_SELECT_DYNAMIC) {
            ok = false;
        }
    }
    flags[(size_t)j * Ws + i] = ok ? 1 : 0;
}

// Raw hardware max / min.  fmaxf()/fminf() first quiet every operand that might be a signalling NaN (one extra
// instruction per tap); here a NaN tap makes the 25-tap sum NaN and the test fail whatever max and min return, so the
// plain instructions do.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float min3_raw(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// The 5x5 test for four horizontally adjacent samples of one thread, out of an LDS tile with row stride TS whose origin
// is the top-left tap of the block's first sample.  (tx, ty): the thread's column group and row in the block; (i, v):
// sample column index of its first sample and image row.  Returns bit k set if sample k is reliable; centre[k] = its depth.
template <int S, int TS>
__device__ __forceinline__ uint32_t reliable4(const float* __restrict__ tile, int tx, int ty, int i, int v, int H, int W,
                                              float (&centre)[4]) {
    constexpr int RW = 3 * S + 5;     // floats of one window row that the four samples touch
    constexpr int RV = (RW + 3) / 4;  // ... in 16-byte reads
    float mx[4], mn[4], acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mx[k] = -INFINITY;
        mn[k] = INFINITY;
        acc[k] = 0.0f;
        centre[k] = 0.0f;
    }
    // all four windows inside the image?  (wave-uniform decision: one code path per wave)
    const bool interior = (v >= 2) && (v + 2 < H) && (i * S >= 2) && ((i + 3) * S + 2 < W);
    const bool fast = __ballot(!interior) == 0ull;
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
        float row[4 * RV];
        const float4* src = reinterpret_cast<const float4*>(&tile[(ty * S + kh) * TS + 4 * tx * S]);
#pragma unroll
        for (int q = 0; q < RV; ++q) {
            const float4 t = src[q];
            row[4 * q + 0] = t.x;
            row[4 * q + 1] = t.y;
            row[4 * q + 2] = t.z;
            row[4 * q + 3] = t.w;
        }
        if (fast) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float* r = &row[k * S];
                mx[k] = max3_raw(max3_raw(mx[k], r[0], r[1]), max3_raw(r[2], r[3], r[4]), r[2]);
                mn[k] = min3_raw(min3_raw(mn[k], r[0], r[1]), min3_raw(r[2], r[3], r[4]), r[2]);
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) acc[k] = acc[k] + r[kw];
            }
        } else {
            const int yy = v + kh - 2;
            const bool rowin = (yy >= 0) && (yy < H);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int kw = 0; kw < 5; ++kw) {
                    const int xx = (i + k) * S + kw - 2;
                    if (rowin && xx >= 0 && xx < W) {  // taps outside the image: skipped by max/min, zero in the sum
                        const float d = row[k * S + kw];
                        mx[k] = fmaxf(mx[k], d);
                        mn[k] = fminf(mn[k], d);
                        acc[k] = acc[k] + d;
                    }
                }
            }
        }
        if (kh == 2) {
#pragma unroll
            for (int k = 0; k < 4; ++k) centre[k] = row[k * S + 2];
        }
    }
    // ratio = (max - min) / (sum / 25 + 1e-6) < 0.1 with two correctly rounded divisions.  Both are first replaced by
    // reciprocal multiplies (a few 1e-7 of relative error); only a ratio within 1e-5 of the threshold, or a
    // denominator that lost digits to cancellation, sends the wave to the exact divisions.
    uint32_t okbits = 0u, unsure = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float mean_a = acc[k] * 0.04f;
        const float den_a = mean_a + 1e-6f;
        const float q = (mx[k] - mn[k]) * __builtin_amdgcn_rcpf(den_a);
        const bool trust = fabsf(den_a) >= 0.125f * fabsf(mean_a);
        const bool yes = trust && (q < 0.099999f), no = trust && (q > 0.100001f);
        okbits |= yes ? (1u << k) : 0u;
        unsure |= (yes || no) ? 0u : (1u << k);  // also every NaN
    }
    if (__ballot(unsure != 0u) != 0ull) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float mean = acc[k] / 25.0f;
            const float ratio = (mx[k] - mn[k]) / (mean + 1e-6f);
            if ((unsure >> k) & 1u) okbits = (okbits & ~(1u << k)) | ((ratio < 0.1f) ? (1u << k) : 0u);
        }
    }
    // A NaN tap makes the sum NaN (torch's max pool propagates it too): the ratio is NaN and the test fails.  Opposite
    // infinities also give a NaN sum, and fail the test as well (ratio = inf / NaN).
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (!((acc[k] == acc[k]) && (centre[k] > 0.0f))) okbits &= ~(1u << k);
    return okbits;
}

// Subsample factors 1 and 2: 64 x 16 samples per block, four horizontally adjacent samples per thread.  A thread pulls
// each of its five window rows out of the LDS tile once, as 16-byte reads, and feeds all four windows from registers, so a
// sample costs 2.5 (S=1) or 3.75 (S=2) LDS reads instead of 25.  Same arithmetic, tap by tap, as reliable_at().
// (Built for 40 registers - the attribute counts in units of two on this target - although it would take 53: two of its
// waves then fit into the 80 registers that three splat waves leave of a SIMD.  A few values spill and the kernel is 1 us
// slower on its own; with frames in flight the frame rate is 3 % higher.  The same goes for k_unproject_scatter.)
template <int S, typename DT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(20))) k_reliable_flags_row4(const DT* __restrict__ depth,
                                                             const uint8_t* __restrict__ mask, int H, int W, int Hs,
                                                             int Ws, int select, uint8_t* __restrict__ flags) {
    constexpr int BX = 64, BY = 16;
    constexpr int FW = (BX - 1) * S + 5, FH = (BY - 1) * S + 5;
    constexpr int TS = (FW + 3) & ~3;   // row stride in floats, a multiple of 4 so that every thread's reads are aligned
    static_assert(4 * (BX / 4 - 1) * S + 4 * ((3 * S + 5 + 3) / 4) <= TS, "row reads stay inside the tile row");
    __shared__ __align__(16) float tile[TS * FH];
    const int i0 = blockIdx.x * BX, j0 = blockIdx.y * BY;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int i = i0 + 4 * tx, j = j0 + ty;
    const int v = j * S;  // image row of the samples; sample k sits at image column (i + k) * S
    // The instance rule first: which of the thread's samples can be selected at all.  A workgroup none of whose samples
    // can (typical for the dynamic selection: moving objects cover a small part of the frame) writes zero flags
    // and skips the depth footprint and the stencil altogether.
    uint32_t want = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        bool w = (j < Hs) && (i + k < Ws);
        if (w && select != EGOPRIOR_SELECT_ALL) {
            if (mask != nullptr) {
                const uint8_t m = mask[(size_t)v * W + (size_t)(i + k) * S];
                w = (select == EGOPRIOR_SELECT_STATIC) ? (m == 0) : (m != 0);
            } else {
                w = select != EGOPRIOR_SELECT_DYNAMIC;
            }
        }
        want |= w ? (1u << k) : 0u;
    }
    if (__syncthreads_or(want != 0u) == 0) {
        if (j < Hs) {
            const size_t o = (size_t)j * Ws + i;
            if (i + 3 < Ws && (o & 3) == 0) {
                *reinterpret_cast<uint32_t*>(&flags[o]) = 0u;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (i + k < Ws) flags[o + k] = 0;
            }
        }
        return;
    }
    {
        // all of a thread's footprint loads are issued before the first of them is stored to LDS
        constexpr int NL = (TS * FH + 255) / 256;
        const int x0 = i0 * S - 2, y0 = j0 * S - 2;
        float tmp[NL];
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int e = min((int)threadIdx.x + q * 256, TS * FH - 1);
            const int ly = e / TS, lx = e - ly * TS;
            int yy = y0 + ly, xx = x0 + lx;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            tmp[q] = ldd(depth, (size_t)yy * W + xx);
        }
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int e = (int)threadIdx.x + q * 256;
            if (e < TS * FH) tile[e] = tmp[q];
        }
    }
    __syncthreads();
    float centre[4];
    const uint32_t okbits = reliable4<S, TS>(tile, tx, ty, i, v, H, W, centre) & want;
    if (j >= Hs) return;
    uint32_t packed = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) packed |= ((okbits >> k) & 1u) << (8 * k);
    const size_t o = (size_t)j * Ws + i;
    if (i + 3 < Ws && (o & 3) == 0) {
        *reinterpret_cast<uint32_t*>(&flags[o]) = packed;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k < Ws) flags[o + k] = (uint8_t)((packed >> (8 * k)) & 1u);
    }
}

// window taps straight from global memory (large subsample factors, where windows do not overlap)
template <typename DT>
__global__ void __launch_bounds__(256) k_reliable_flags_direct(const DT* __restrict__ depth,
                                                               const uint8_t* __restrict__ mask, int H, int W, int s,
                                                               int Hs, int Ws, int select,
                                                               uint8_t* __restrict__ flags) {
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)Hs * Ws) return;
    const int j = (int)(e / Ws), i = (int)(e - (int64_t)j * Ws);
    const int u = i * s, v = j * s;
    float win[25];
#pragma unroll
    for (int kh = 0; kh < 5; ++kh) {
#pragma unroll
        for (int kw = 0; kw < 5; ++kw) {
            int yy = v + kh - 2, xx = u + kw - 2;
            yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            win[kh * 5 + kw] = ldd(depth, (size_t)yy * W + xx);
        }
    }
    bool ok = reliable_at(win, 5, 0, 0, u, v, H, W);
    if (select != EGOPRIOR_SELECT_ALL) {
        if (mask != nullptr) {
            const uint8_t m = mask[(size_t)v * W + u];
            ok = ok && ((select == EGOPRIOR_SELECT_STATIC) ? (m == 0) : (m != 0));
        } else if (select == EGOPRIOR_SELECT_DYNAMIC) {
            ok = false;
        }
    }
    flags[e] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// order-preserving compaction helpers: 1024 elements per block, 4 consecutive elements per thread
// ------------------------------------------------------------------------------------------------------------------
#define EP_CB 1024

// the four flag bytes of elements base .. base+3 as one word (byte k = element base+k; zero past n)
__device__ __forceinline__ uint32_t load_flags4(const uint8_t* __restrict__ flags, int64_t base, int64_t n) {
    if (base + 3 < n) return *reinterpret_cast<const uint32_t*>(flags + base);  // base is a multiple of 4
    uint32_t w = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (base + k < n) w |= (uint32_t)flags[base + k] << (8 * k);
    return w;
}

// d_flags (may be NULL): cleared here when no k_block_scan runs in between (see block_offset_of)
__global__ void __launch_bounds__(256) k_flag_block_count(const uint8_t* __restrict__ flags, int64_t n,
                                                          uint32_t* __restrict__ block_count,
                                                          int32_t* __restrict__ d_flags) {
    __shared__ uint32_t ws[4];
    if (blockIdx.x == 0 && threadIdx.x == 0 && d_flags != nullptr) *d_flags = 0;
    const int64_t base = (int64_t)blockIdx.x * EP_CB + (int64_t)threadIdx.x * 4;
    const uint32_t w4 = load_flags4(flags, base, n);
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) c += ((w4 >> (8 * k)) & 255u) ? 1u : 0u;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) c += __shfl_xor(c, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}

// exclusive scan of block counts (one workgroup); writes the total to d_count and clears d_flags
__global__ void __launch_bounds__(1024) k_block_scan(const uint32_t* __restrict__ block_count,
                                                     uint32_t* __restrict__ block_offset, int n_blocks,
                                                     int32_t* __restrict__ d_count, int32_t* __restrict__ d_flags) {
    __shared__ uint32_t wave_sums[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024) {
        const int t = base + tid;
        const uint32_t v = (t < n_blocks) ? block_count[t] : 0u;
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
        if (t < n_blocks) block_offset[t] = carry + woff + incl - v;
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) {
        if (d_count != nullptr) *d_count = (int32_t)carry_s;
        if (d_flags != nullptr) *d_flags = 0;
    }
}

// rank of each of this thread's 4 elements inside the block (exclusive), given its 4 flags
__device__ __forceinline__ uint32_t block_exclusive_base(uint32_t thread_total, uint32_t* ws /*[4] shared*/) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t incl = thread_total;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
    }
    if (lane == 63) ws[wid] = incl;
    __syncthreads();
    uint32_t woff = 0;
    for (int w2 = 0; w2 < wid; ++w2) woff += ws[w2];
    return woff + incl - thread_total;
}

#define EP_FUSED_SCAN_MAX 4096  // up to this many blocks every scatter workgroup sums the counts before it by itself

// Offset of this workgroup's first output record.  With block_offset == NULL (no k_block_scan ran) the workgroup adds up
// the counts of the blocks before it -- a few KiB that sit in the L2 -- and the last workgroup also publishes the total;
// one launch less on a chain of four short kernels.  Must be reached by all 256 threads.
__device__ __forceinline__ uint32_t block_offset_of(const uint32_t* __restrict__ block_count,
                                                    const uint32_t* __restrict__ block_offset,
                                                    int32_t* __restrict__ d_count, uint32_t* ws /*[4] shared*/) {
    if (block_offset != nullptr) return block_offset[blockIdx.x];
    uint32_t s = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += 256) s += block_count[i];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    const uint32_t before = ws[0] + ws[1] + ws[2] + ws[3];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0 && d_count != nullptr)
        *d_count = (int32_t)(before + block_count[blockIdx.x]);
    return before;
}

// Must be reached by all threads of the workgroup.  One atomic per workgroup at most, and none once the flag is up:
// thousands of same-address atomics would serialise in the L2.
__device__ __forceinline__ void note_color_gt1(bool any_gt1, int32_t* __restrict__ d_flags) {
    if (d_flags == nullptr) return;
    if (__syncthreads_or(any_gt1 ? 1 : 0) != 0 && threadIdx.x == 0 &&
        __atomic_load_n(d_flags, __ATOMIC_RELAXED) == 0)
        atomicOr(d_flags, 1);
}

// flags + block offsets -> compacted records: rays * depth, c2w (float32), exo->world (float64), colour bytes
template <typename DT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(20))) k_unproject_scatter(egoprior_unproject_params p, int Hs, int Ws,
                                                           const DT* __restrict__ depth,
                                                           const float* __restrict__ rays,
                                                           const uint8_t* __restrict__ rgb,
                                                           const uint8_t* __restrict__ flags,
                                                           const uint32_t* __restrict__ block_count,
                                                           const uint32_t* __restrict__ block_offset,
                                                           egoprior_point* __restrict__ out,
                                                           double* __restrict__ out64, int32_t* __restrict__ d_count,
                                                           int32_t* __restrict__ d_flags) {
    __shared__ uint32_t ws[4];
    __shared__ uint32_t ws_off[4];
    const int64_t n = (int64_t)Hs * Ws;
    const int64_t base = (int64_t)blockIdx.x * EP_CB + (int64_t)threadIdx.x * 4;
    // nothing selected in this block (uniform for the workgroup); the last one may still have the total to publish
    if (block_count[blockIdx.x] == 0u && !(block_offset == nullptr && blockIdx.x == gridDim.x - 1)) return;
    const uint32_t w4 = load_flags4(flags, base, n);
    bool f[4];
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[k] = ((w4 >> (8 * k)) & 255u) != 0u;
        c += f[k] ? 1u : 0u;
    }
    // The depth and colour of the selected elements are fetched together (unselected ones read pixel 0) and before the
    // block-wide scan, so that their latency overlaps it; one division gives the grid position of all four elements.
    int jj[4], ii[4];
    {
        const int64_t eb = base < n ? base : n - 1;
        jj[0] = (int)(eb / Ws);
        ii[0] = (int)(eb - (int64_t)jj[0] * Ws);
#pragma unroll
        for (int k = 1; k < 4; ++k) {
            const bool wrap = ii[k - 1] + 1 == Ws;
            ii[k] = wrap ? 0 : ii[k - 1] + 1;
            jj[k] = jj[k - 1] + (wrap ? 1 : 0);
        }
    }
    float dv[4], rx[4], ry[4], rz[4];
    uint32_t cr[4], cg[4], cb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const size_t pix = f[k] ? (size_t)(jj[k] * p.subsample) * p.W + (size_t)(ii[k] * p.subsample) : 0;
        dv[k] = ldd(depth, pix);
        if (rays != nullptr) {  // ray table of a non-pinhole camera, one ray per sample of the [::s, ::s] grid
            const size_t e = f[k] ? (size_t)jj[k] * Ws + ii[k] : 0;
            rx[k] = rays[e * 3 + 0];
            ry[k] = rays[e * 3 + 1];
            rz[k] = rays[e * 3 + 2];
        }
        cr[k] = rgb[pix * 3 + 0];
        cg[k] = rgb[pix * 3 + 1];
        cb[k] = rgb[pix * 3 + 2];
    }
    uint32_t slot = block_offset_of(block_count, block_offset, d_count, ws_off) + block_exclusive_base(c, ws);
    bool gt1 = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!f[k]) continue;
        const int u = ii[k] * p.subsample, v = jj[k] * p.subsample;
        const float d = dv[k];
        const float X = (rays != nullptr) ? rx[k] : ((float)u - p.cx) / p.fx;
        const float Y = (rays != nullptr) ? ry[k] : ((float)v - p.cy) / p.fy;
        const float Z = (rays != nullptr) ? rz[k] : 1.0f;
        const float pcx = X * d, pcy = Y * d, pcz = Z * d;
        const float vx = ((p.c2w[0] * pcx + p.c2w[1] * pcy) + p.c2w[2] * pcz) + p.c2w[3];
        const float vy = ((p.c2w[4] * pcx + p.c2w[5] * pcy) + p.c2w[6] * pcz) + p.c2w[7];
        const float vz = ((p.c2w[8] * pcx + p.c2w[9] * pcy) + p.c2w[10] * pcz) + p.c2w[11];
        const double wx = ((p.T[0] * (double)vx + p.T[1] * (double)vy) + p.T[2] * (double)vz) + p.T[3];
        const double wy = ((p.T[4] * (double)vx + p.T[5] * (double)vy) + p.T[6] * (double)vz) + p.T[7];
        const double wz = ((p.T[8] * (double)vx + p.T[9] * (double)vy) + p.T[10] * (double)vz) + p.T[11];
        const uint32_t r = cr[k], g = cg[k], b = cb[k];
        gt1 |= (r > 1u) || (g > 1u) || (b > 1u);
        egoprior_point q;
        q.x = (float)wx;
        q.y = (float)wy;
        q.z = (float)wz;
        q.rgba = r | (g << 8) | (b << 16);
        out[slot] = q;
        if (out64 != nullptr) {
            out64[(size_t)slot * 3 + 0] = wx;
            out64[(size_t)slot * 3 + 1] = wy;
            out64[(size_t)slot * 3 + 2] = wz;
        }
        ++slot;
    }
    note_color_gt1(gt1, d_flags);
}

// render-entry filters: finite coordinates and |p| < 1e6 (float32 norm, ((x*x + y*y) + z*z) then sqrt)
__global__ void __launch_bounds__(256) k_filter_flags(const float* __restrict__ xyz, int64_t n,
                                                      uint8_t* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = xyz[i * 3 + 0], y = xyz[i * 3 + 1], z = xyz[i * 3 + 2];
    const bool finite = isfinite(x) && isfinite(y) && isfinite(z);
    const float mag = sqrtf((x * x + y * y) + z * z);
    flags[i] = (finite && (mag < 1e6f)) ? 1 : 0;
}

__global__ void __launch_bounds__(256) k_filter_scatter(const float* __restrict__ xyz, const uint8_t* __restrict__ rgb,
                                                        int64_t n, const uint8_t* __restrict__ flags,
                                                        const uint32_t* __restrict__ block_count,
                                                        const uint32_t* __restrict__ block_offset,
                                                        egoprior_point* __restrict__ out, int32_t* __restrict__ d_count,
                                                        int32_t* __restrict__ d_flags) {
    __shared__ uint32_t ws[4];
    __shared__ uint32_t ws_off[4];
    const int64_t base = (int64_t)blockIdx.x * EP_CB + (int64_t)threadIdx.x * 4;
    // nothing selected in this block (uniform for the workgroup); the last one may still have the total to publish
    if (block_count[blockIdx.x] == 0u && !(block_offset == nullptr && blockIdx.x == gridDim.x - 1)) return;
    const uint32_t w4 = load_flags4(flags, base, n);
    bool f[4];
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f[k] = ((w4 >> (8 * k)) & 255u) != 0u;
        c += f[k] ? 1u : 0u;
    }
    uint32_t slot = block_offset_of(block_count, block_offset, d_count, ws_off) + block_exclusive_base(c, ws);
    bool gt1 = false;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (!f[k]) continue;
        const int64_t e = base + k;
        const uint32_t r = rgb[e * 3 + 0], g = rgb[e * 3 + 1], b = rgb[e * 3 + 2];
        gt1 |= (r > 1u) || (g > 1u) || (b > 1u);
        egoprior_point q;
        q.x = xyz[e * 3 + 0];
        q.y = xyz[e * 3 + 1];
        q.z = xyz[e * 3 + 2];
        q.rgba = r | (g << 8) | (b << 16);
        out[slot++] = q;
    }
    note_color_gt1(gt1, d_flags);
}

// ------------------------------------------------------------------------------------------------------------------
// mean background
// ------------------------------------------------------------------------------------------------------------------
// state planes: [(level*4 + q) * HW + pix], level 0..3, q = depth, r, g, b; plane 16 = int32 count.
// One thread owns one pixel and walks the frames; per frame the block stages the 5x5 footprint of its 32x8 pixels in
// LDS.  Sums follow ATen's CPU cascade order (see oracle.cascade_nansum): level 0 takes the terms one by one and is
// flushed upward after every 16th term.
template <typename DT>
__global__ void __launch_bounds__(256) k_meanbg_accumulate(float* __restrict__ state, const DT* __restrict__ depth,
                                                           const uint8_t* __restrict__ rgb,
                                                           const uint8_t* __restrict__ mask, int F,
                                                           long long first_index, int H, int W) {
    __shared__ float tile[(EP_MASK_SX + 4) * (EP_MASK_SY + 4)];
    __shared__ float lut[256];
    const int fw = EP_MASK_SX + 4, fh = EP_MASK_SY + 4;
    const int x0 = blockIdx.x * EP_MASK_SX, y0 = blockIdx.y * EP_MASK_SY;
    const int lx = threadIdx.x & (EP_MASK_SX - 1), ly = threadIdx.x / EP_MASK_SX;
    const int u = x0 + lx, v = y0 + ly;
    const bool inside = (u < W) && (v < H);
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)v * W + u;
    lut[threadIdx.x] = (float)threadIdx.x / 255.0f;  // rgb = u8 / 255.0 (vipe/utils/io.py:250)

    float acc[4][4];
    int cnt = 0;
#pragma unroll
    for (int l = 0; l < 4; ++l)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[l][q] = inside ? state[(size_t)(l * 4 + q) * HW + pix] : 0.0f;
    if (inside) cnt = ((const int*)state)[16 * HW + pix];

    for (int f = 0; f < F; ++f) {
        const DT* dfr = depth + (size_t)f * HW;
        __syncthreads();
        load_footprint(tile, dfr, H, W, x0 - 2, y0 - 2, fw, fh);
        __syncthreads();
        if (inside) {
            bool ok = reliable_at(tile, fw, lx, ly, u, v, H, W);
            if (mask != nullptr) ok = ok && (mask[(size_t)f * HW + pix] == 0);
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f, t3 = 0.0f;
            if (ok) {
                const uint8_t* c = rgb + ((size_t)f * HW + pix) * 3;
                t0 = tile[(ly + 2) * fw + lx + 2];
                t1 = lut[c[0]];
                t2 = lut[c[1]];
                t3 = lut[c[2]];
                ++cnt;
            }
            acc[0][0] = acc[0][0] + t0;
            acc[0][1] = acc[0][1] + t1;
            acc[0][2] = acc[0][2] + t2;
            acc[0][3] = acc[0][3] + t3;
            const long long i1 = first_index + f + 1;
            if ((i1 & 15) == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[1][q] = acc[1][q] + acc[0][q];
                    acc[0][q] = 0.0f;
                }
                if ((i1 & 0xF0) == 0) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[2][q] = acc[2][q] + acc[1][q];
                        acc[1][q] = 0.0f;
                    }
                    if ((i1 & 0xF00) == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[3][q] = acc[3][q] + acc[2][q];
                            acc[2][q] = 0.0f;
                        }
                    }
                }
            }
        }
    }
    if (inside) {
#pragma unroll
        for (int l = 0; l < 4; ++l)
#pragma unroll
            for (int q = 0; q < 4; ++q) state[(size_t)(l * 4 + q) * HW + pix] = acc[l][q];
        ((int*)state)[16 * HW + pix] = cnt;
    }
}

// Same reduction, 64 x 16 pixels per workgroup and four horizontally adjacent pixels per thread (W % 4 == 0).  The next
// frame's depth footprint, mask bytes and colour bytes are fetched into registers while the current frame is worked on,
// so a frame costs its arithmetic and not two exposed memory latencies.  Per pixel the terms are added in the same
// order as in the one-pixel-per-thread kernel.
template <typename DT>
__global__ void __launch_bounds__(256) k_meanbg_accumulate_row4(float* __restrict__ state,
                                                                const DT* __restrict__ depth,
                                                                const uint8_t* __restrict__ rgb,
                                                                const uint8_t* __restrict__ mask, int F,
                                                                long long first_index, int H, int W) {
    constexpr int BX = 64, BY = 16, FW = BX + 4, FH = BY + 4, TS = FW;  // 68 floats per tile row: a multiple of 4
    constexpr int NL = (TS * FH + 255) / 256;
    __shared__ __align__(16) float tile[TS * FH];
    __shared__ float lut[256];
    const int x0 = blockIdx.x * BX, y0 = blockIdx.y * BY;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int u = x0 + 4 * tx, v = y0 + ty;  // first of the thread's four pixels
    const bool inside = (u < W) && (v < H);  // W % 4 == 0: all four or none
    const size_t HW = (size_t)H * W;
    const size_t pix = inside ? (size_t)v * W + u : 0;
    lut[threadIdx.x] = (float)threadIdx.x / 255.0f;  // rgb = u8 / 255.0 (vipe/utils/io.py:250)

    // footprint element e of this thread's q-th load -> clamped image offset (the same for every frame)
    uint32_t foff[NL];
#pragma unroll
    for (int q = 0; q < NL; ++q) {
        const int e = min((int)threadIdx.x + q * 256, TS * FH - 1);
        const int ly = e / TS, lx = e - ly * TS;
        int yy = y0 - 2 + ly, xx = x0 - 2 + lx;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
        xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
        foff[q] = (uint32_t)yy * (uint32_t)W + (uint32_t)xx;
    }

    float acc[4][4][4];  // [pixel][level][quantity]
    int cnt[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int l = 0; l < 4; ++l)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[k][l][q] = inside ? state[(size_t)(l * 4 + q) * HW + pix + k] : 0.0f;
        cnt[k] = inside ? ((const int*)state)[16 * HW + pix + k] : 0;
    }

    // registers holding the frame that is worked on next
    float nfoot[NL];
    uint32_t nmask = 0u, nrgb[3] = {0u, 0u, 0u};
    auto fetch = [&](int f) {
        const DT* dfr = depth + (size_t)f * HW;
#pragma unroll
        for (int q = 0; q < NL; ++q) nfoot[q] = ldd(dfr, foff[q]);
        if (mask != nullptr) nmask = *reinterpret_cast<const uint32_t*>(mask + (size_t)f * HW + pix);
        const uint32_t* c = reinterpret_cast<const uint32_t*>(rgb + ((size_t)f * HW + pix) * 3);  // 12 bytes, 4-aligned
        nrgb[0] = c[0];
        nrgb[1] = c[1];
        nrgb[2] = c[2];
    };
    if (F > 0) fetch(0);
    for (int f = 0; f < F; ++f) {
        __syncthreads();  // the previous frame's readers of the tile are done
#pragma unroll
        for (int q = 0; q < NL; ++q) {
            const int e = (int)threadIdx.x + q * 256;
            if (e < TS * FH) tile[e] = nfoot[q];
        }
        const uint32_t cmask = nmask, c0 = nrgb[0], c1 = nrgb[1], c2 = nrgb[2];
        if (f + 1 < F) fetch(f + 1);
        __syncthreads();
        float centre[4];
        uint32_t ok = reliable4<1, TS>(tile, tx, ty, u, v, H, W, centre);
        // colour bytes of the four pixels: r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3
        const uint32_t cr[4] = {c0 & 255u, c0 >> 24, (c1 >> 16) & 255u, (c2 >> 8) & 255u};
        const uint32_t cg[4] = {(c0 >> 8) & 255u, c1 & 255u, c1 >> 24, (c2 >> 16) & 255u};
        const uint32_t cb[4] = {(c0 >> 16) & 255u, (c1 >> 8) & 255u, c2 & 255u, c2 >> 24};
        const long long i1 = first_index + f + 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bool take = inside && ((ok >> k) & 1u) != 0u && ((cmask >> (8 * k)) & 255u) == 0u;
            const float t0 = take ? centre[k] : 0.0f;
            const float t1 = take ? lut[cr[k]] : 0.0f;
            const float t2 = take ? lut[cg[k]] : 0.0f;
            const float t3 = take ? lut[cb[k]] : 0.0f;
            cnt[k] += take ? 1 : 0;
            acc[k][0][0] = acc[k][0][0] + t0;
            acc[k][0][1] = acc[k][0][1] + t1;
            acc[k][0][2] = acc[k][0][2] + t2;
            acc[k][0][3] = acc[k][0][3] + t3;
        }
        if ((i1 & 15) == 0) {  // frame-index conditions: the same for every thread
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[k][1][q] = acc[k][1][q] + acc[k][0][q];
                    acc[k][0][q] = 0.0f;
                }
            if ((i1 & 0xF0) == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc[k][2][q] = acc[k][2][q] + acc[k][1][q];
                        acc[k][1][q] = 0.0f;
                    }
                if ((i1 & 0xF00) == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[k][3][q] = acc[k][3][q] + acc[k][2][q];
                            acc[k][2][q] = 0.0f;
                        }
                }
            }
        }
    }
    if (inside) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#pragma unroll
            for (int l = 0; l < 4; ++l)
#pragma unroll
                for (int q = 0; q < 4; ++q) state[(size_t)(l * 4 + q) * HW + pix + k] = acc[k][l][q];
            ((int*)state)[16 * HW + pix + k] = cnt[k];
        }
    }
}

__device__ __forceinline__ float folded_sum(const float* __restrict__ state, size_t HW, size_t pix, int q) {
    float s = state[(size_t)q * HW + pix];
    s = s + state[(size_t)(4 + q) * HW + pix];
    s = s + state[(size_t)(8 + q) * HW + pix];
    s = s + state[(size_t)(12 + q) * HW + pix];
    return s;
}

__device__ __forceinline__ float nan_to_num0(float x) {
    if (x != x) return 0.0f;
    if (isinf(x)) return x > 0.0f ? 3.4028234663852886e38f : -3.4028234663852886e38f;
    return x;
}

__global__ void __launch_bounds__(256) k_meanbg_finalize(const float* __restrict__ state, int H, int W,
                                                         float* __restrict__ mean_depth,
                                                         uint8_t* __restrict__ mean_rgb) {
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
    const float cnt = (float)((const int*)state)[16 * HW + pix];
    mean_depth[pix] = nan_to_num0(folded_sum(state, HW, pix, 0) / cnt);
#pragma unroll
    for (int q = 1; q < 4; ++q) {
        const float m = nan_to_num0(folded_sum(state, HW, pix, q) / cnt);
        const float c = fminf(fmaxf(m * 255.0f, 0.0f), 255.0f);
        mean_rgb[pix * 3 + (q - 1)] = (uint8_t)c;
    }
}

// The viewer's flavour of the finalisation (vipe/utils/viser.py:60-74, 189-192): the means of the pixels of the
// [::s, ::s] grid only, colours left as float32 means (no quantisation).  A pixel's terms were added in frame order by
// the accumulate kernels, which is the order torch.nanmean uses whether the stack is subsampled before or after.
__global__ void __launch_bounds__(256) k_meanbg_finalize_f32(const float* __restrict__ state, int H, int W, int s, int Hs,
                                                             int Ws, float* __restrict__ mean_depth,
                                                             float* __restrict__ mean_rgb) {
    const size_t HW = (size_t)H * W;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)Hs * Ws) return;
    const size_t j = e / (size_t)Ws, i = e - j * (size_t)Ws;
    const size_t pix = (j * (size_t)s) * (size_t)W + i * (size_t)s;
    const float cnt = (float)((const int*)state)[16 * HW + pix];
    mean_depth[e] = nan_to_num0(folded_sum(state, HW, pix, 0) / cnt);
#pragma unroll
    for (int q = 1; q < 4; ++q) mean_rgb[e * 3 + (q - 1)] = nan_to_num0(folded_sum(state, HW, pix, q) / cnt);
}

__global__ void __launch_bounds__(256) k_meanbg_export(const float* __restrict__ state, int H, int W,
                                                       float* __restrict__ sums4, int32_t* __restrict__ count) {
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) sums4[(size_t)q * HW + pix] = folded_sum(state, HW, pix, q);
    count[pix] = ((const int*)state)[16 * HW + pix];
}

__global__ void __launch_bounds__(256) k_meanbg_import(float* __restrict__ state, int H, int W,
                                                       const float* __restrict__ sums4,
                                                       const int32_t* __restrict__ count) {
    const size_t HW = (size_t)H * W;
    const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        state[(size_t)q * HW + pix] = sums4[(size_t)q * HW + pix];
        state[(size_t)(4 + q) * HW + pix] = 0.0f;
        state[(size_t)(8 + q) * HW + pix] = 0.0f;
        state[(size_t)(12 + q) * HW + pix] = 0.0f;
    }
    ((int*)state)[16 * HW + pix] = count[pix];
}

}  // namespace ep

// ------------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------------
namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct CompactWs {
    uint8_t* flags;
    uint32_t* block_count;
    uint32_t* block_offset;
    int n_blocks;
};

size_t compact_ws_bytes(int64_t n) {
    const int64_t nb = (n + EP_CB - 1) / EP_CB + 1;
    return align_up((size_t)n + 16, 256) + 2 * align_up((size_t)nb * sizeof(uint32_t), 256);
}

bool carve_compact_ws(void* ws, size_t bytes, int64_t n, CompactWs& c) {
    if (ws == nullptr || bytes < compact_ws_bytes(n)) return false;
    const int64_t nb = (n + EP_CB - 1) / EP_CB + 1;
    char* p = (char*)ws;
    c.flags = (uint8_t*)p;
    p += align_up((size_t)n + 16, 256);
    c.block_count = (uint32_t*)p;
    p += align_up((size_t)nb * sizeof(uint32_t), 256);
    c.block_offset = (uint32_t*)p;
    c.n_blocks = (int)((n + EP_CB - 1) / EP_CB);
    return true;
}

template <typename DT>
int launch_flags(const DT* d_depth, const uint8_t* d_mask, int H, int W, int s, int select, uint8_t* flags,
                 hipStream_t st) {
    const int Hs = (H + s - 1) / s, Ws = (W + s - 1) / s;
    if (s == 1) {
        EP_LAUNCH("k_reliable_flags", (ep::k_reliable_flags_row4<1, DT>), dim3((Ws + 63) / 64, (Hs + 15) / 16), dim3(256), 0, st,
                  d_depth, d_mask, H, W, Hs, Ws, select, flags);
    } else if (s == 2) {
        EP_LAUNCH("k_reliable_flags", (ep::k_reliable_flags_row4<2, DT>), dim3((Ws + 63) / 64, (Hs + 15) / 16), dim3(256), 0, st,
                  d_depth, d_mask, H, W, Hs, Ws, select, flags);
    } else if (s <= 4) {
        const int fw = (EP_MASK_SX - 1) * s + 5, fh = (EP_MASK_SY - 1) * s + 5;
        const dim3 grid((Ws + EP_MASK_SX - 1) / EP_MASK_SX, (Hs + EP_MASK_SY - 1) / EP_MASK_SY);
        EP_LAUNCH("k_reliable_flags", ep::k_reliable_flags<DT>, grid, dim3(256), sizeof(float) * (size_t)fw * fh, st, d_depth, d_mask,
                           H, W, s, Hs, Ws, select, flags);
    } else {
        const int64_t n = (int64_t)Hs * Ws;
        EP_LAUNCH("k_reliable_flags_direct", ep::k_reliable_flags_direct<DT>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_depth,
                           d_mask, H, W, s, Hs, Ws, select, flags);
    }
    return ep_host::check_launch("reliable_flags");
}

template <typename DT>
int reliable_depth_mask_t(const DT* d_depth, int32_t H, int32_t W, uint8_t* d_mask_out, void* stream) {
    if (d_depth == nullptr || d_mask_out == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    if (H <= 0 || W <= 0) return ep_host::fail(EGOPRIOR_E_INVALID, "bad image size");
    return launch_flags(d_depth, nullptr, H, W, 1, EGOPRIOR_SELECT_ALL, d_mask_out, (hipStream_t)stream);
}

}  // namespace

extern "C" int egoprior_reliable_depth_mask(const float* d_depth, int32_t H, int32_t W, uint8_t* d_mask_out,
                                            void* stream) {
    return reliable_depth_mask_t(d_depth, H, W, d_mask_out, stream);
}
extern "C" int egoprior_reliable_depth_mask_f16(const uint16_t* d_depth, int32_t H, int32_t W, uint8_t* d_mask_out,
                                                void* stream) {
    return reliable_depth_mask_t(reinterpret_cast<const __half*>(d_depth), H, W, d_mask_out, stream);
}

extern "C" size_t egoprior_unproject_workspace_bytes(int32_t H, int32_t W, int32_t subsample) {
    if (H <= 0 || W <= 0 || subsample <= 0) return 0;
    const int64_t Hs = (H + subsample - 1) / subsample, Ws = (W + subsample - 1) / subsample;
    return compact_ws_bytes(Hs * Ws);
}

namespace {
template <typename DT>
int unproject_frame_t(const egoprior_unproject_params* p, const DT* d_depth, const float* d_rays, const uint8_t* d_rgb,
                      const uint8_t* d_mask, egoprior_point* d_points, double* d_xyz64, int32_t* d_count,
                      int32_t* d_flags, void* d_workspace, size_t workspace_bytes, void* stream) {
    if (p == nullptr || d_depth == nullptr || d_rgb == nullptr || d_points == nullptr || d_count == nullptr)
        return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    if (p->H <= 0 || p->W <= 0 || p->subsample <= 0) return ep_host::fail(EGOPRIOR_E_INVALID, "bad frame geometry");
    if (p->select < EGOPRIOR_SELECT_STATIC || p->select > EGOPRIOR_SELECT_ALL)
        return ep_host::fail(EGOPRIOR_E_INVALID, "unknown select rule");
    const int s = p->subsample;
    const int Hs = (p->H + s - 1) / s, Ws = (p->W + s - 1) / s;
    const int64_t n = (int64_t)Hs * Ws;
    CompactWs c;
    if (!carve_compact_ws(d_workspace, workspace_bytes, n, c))
        return ep_host::fail(EGOPRIOR_E_WORKSPACE, "unproject workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (int e = launch_flags(d_depth, d_mask, p->H, p->W, s, p->select, c.flags, st)) return e;
    const bool fused = c.n_blocks <= EP_FUSED_SCAN_MAX;  // the scatter kernel sums the block counts itself
    EP_LAUNCH("k_flag_block_count", ep::k_flag_block_count, dim3(c.n_blocks), dim3(256), 0, st, c.flags, n, c.block_count,
              fused ? d_flags : (int32_t*)nullptr);
    if (!fused)
        EP_LAUNCH("k_block_scan", ep::k_block_scan, dim3(1), dim3(1024), 0, st, c.block_count, c.block_offset, c.n_blocks,
                  d_count, d_flags);
    EP_LAUNCH("k_unproject_scatter", ep::k_unproject_scatter<DT>, dim3(c.n_blocks), dim3(256), 0, st, *p, Hs, Ws, d_depth,
              d_rays, d_rgb, c.flags, c.block_count, fused ? (const uint32_t*)nullptr : c.block_offset, d_points, d_xyz64,
              d_count, d_flags);
    return ep_host::check_launch("unproject_frame");
}
}  // namespace

extern "C" int egoprior_unproject_frame(const egoprior_unproject_params* p, const float* d_depth, const uint8_t* d_rgb,
                                        const uint8_t* d_mask, egoprior_point* d_points, double* d_xyz64,
                                        int32_t* d_count, int32_t* d_flags, void* d_workspace, size_t workspace_bytes,
                                        void* stream) {
    return unproject_frame_t(p, d_depth, (const float*)nullptr, d_rgb, d_mask, d_points, d_xyz64, d_count, d_flags,
                             d_workspace, workspace_bytes, stream);
}
extern "C" int egoprior_unproject_frame_f16(const egoprior_unproject_params* p, const uint16_t* d_depth,
                                            const uint8_t* d_rgb, const uint8_t* d_mask, egoprior_point* d_points,
                                            double* d_xyz64, int32_t* d_count, int32_t* d_flags, void* d_workspace,
                                            size_t workspace_bytes, void* stream) {
    return unproject_frame_t(p, reinterpret_cast<const __half*>(d_depth), (const float*)nullptr, d_rgb, d_mask,
                             d_points, d_xyz64, d_count, d_flags, d_workspace, workspace_bytes, stream);
}
extern "C" int egoprior_unproject_frame_rays(const egoprior_unproject_params* p, const void* d_depth,
                                             int32_t depth_is_f16, const float* d_rays, const uint8_t* d_rgb,
                                             const uint8_t* d_mask, egoprior_point* d_points, double* d_xyz64,
                                             int32_t* d_count, int32_t* d_flags, void* d_workspace,
                                             size_t workspace_bytes, void* stream) {
    if (d_rays == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "d_rays is NULL");
    if (depth_is_f16)
        return unproject_frame_t(p, reinterpret_cast<const __half*>(d_depth), d_rays, d_rgb, d_mask, d_points, d_xyz64,
                                 d_count, d_flags, d_workspace, workspace_bytes, stream);
    return unproject_frame_t(p, reinterpret_cast<const float*>(d_depth), d_rays, d_rgb, d_mask, d_points, d_xyz64,
                             d_count, d_flags, d_workspace, workspace_bytes, stream);
}

// ---- the entry point that allocates nothing ---------------------------------------------------------------------
// Scratch of egoprior_unproject_frame_noalloc: the selection flags (one byte per sample) and the per-1024 block counts
// and offsets of the compaction.  A file-scope device array: every GPU the library runs on holds its own copy, at its
// own address, until the process ends.  In host code the array's name is only a stand-in for HIP_SYMBOL(); the memory
// is reached through the pointer hipGetSymbolAddress returns for the current device.
#define EP_NOALLOC_SCRATCH_BYTES ((size_t)EGOPRIOR_NOALLOC_MAX_SAMPLES + 16 + 256 + 2 * (((size_t)EGOPRIOR_NOALLOC_MAX_SAMPLES / EP_CB + 2) * 4 + 256))
static __device__ __align__(256) unsigned char scratch[EP_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
void* g_scratch_ptr[kMaxDevices] = {nullptr};  // per device; written once under g_scratch_mu
std::mutex g_scratch_mu;

// Address of `scratch` on the current device (resolved on the first call on that device), or NULL with the error text
// set.  The kernels run on `stream`: its device must be the current one, or they would be handed another GPU's array.
void* scratch_on_current_device(hipStream_t stream) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        ep_host::fail(EGOPRIOR_E_HIP, "hipGetDevice failed (or more than 64 devices)");
        return nullptr;
    }
    if (stream != nullptr) {  // (the null stream always belongs to the current device)
        hipDevice_t sdev = -1;
        if (hipStreamGetDevice(stream, &sdev) != hipSuccess) {
            ep_host::fail(EGOPRIOR_E_HIP, "hipStreamGetDevice failed");
            return nullptr;
        }
        if ((int)sdev != dev) {
            std::snprintf(ep_host::error_buffer(), 512,
                          "egoprior_unproject_frame_noalloc: the stream belongs to device %d but device %d is current; "
                          "make the stream's device current (hipSetDevice) before the call",
                          (int)sdev, dev);
            return nullptr;
        }
    }
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        void* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || ptr == nullptr) {
            std::snprintf(ep_host::error_buffer(), 512, "hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
            return nullptr;
        }
        g_scratch_ptr[dev] = ptr;
    }
    return g_scratch_ptr[dev];
}
}  // namespace

extern "C" int egoprior_unproject_frame_noalloc(const egoprior_unproject_params* p, const float* d_depth,
                                                const uint8_t* d_rgb, const uint8_t* d_mask,
                                                egoprior_point* d_points, double* d_xyz64, int32_t* d_count,
                                                int32_t* d_flags, void* stream) {
    if (p == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    if (p->H <= 0 || p->W <= 0 || p->subsample <= 0) return ep_host::fail(EGOPRIOR_E_INVALID, "bad frame geometry");
    const int64_t Hs = (p->H + p->subsample - 1) / p->subsample, Ws = (p->W + p->subsample - 1) / p->subsample;
    if (Hs * Ws > (int64_t)EGOPRIOR_NOALLOC_MAX_SAMPLES) {
        std::snprintf(ep_host::error_buffer(), 512,
                      "egoprior_unproject_frame_noalloc: %lld samples exceed the static scratch (at most %lld); use "
                      "egoprior_unproject_frame with a caller-owned workspace",
                      (long long)(Hs * Ws), (long long)EGOPRIOR_NOALLOC_MAX_SAMPLES);
        return EGOPRIOR_E_UNSUPPORTED;
    }
    void* const ptr = scratch_on_current_device((hipStream_t)stream);
    if (ptr == nullptr) return EGOPRIOR_E_INVALID;
    // Nothing in the scratch has to start at zero: the flags kernel writes every flag byte of this frame, the count
    // kernel every block count, the scan (when it runs) every block offset, and nothing beyond this frame's extent is
    // read -- so a call after a larger one is not disturbed by what that one left behind.
    return unproject_frame_t(p, d_depth, (const float*)nullptr, d_rgb, d_mask, d_points, d_xyz64, d_count, d_flags,
                             ptr, EP_NOALLOC_SCRATCH_BYTES, stream);
}

extern "C" size_t egoprior_filter_workspace_bytes(int64_t N) { return N < 0 ? 0 : compact_ws_bytes(N > 0 ? N : 1); }

extern "C" int egoprior_filter_cloud(const float* d_xyz, const uint8_t* d_rgb, int64_t N, egoprior_point* d_points,
                                     int32_t* d_count, int32_t* d_flags, void* d_workspace, size_t workspace_bytes,
                                     void* stream) {
    if (N < 0 || N > 0x7FFFFFF0ll) return ep_host::fail(EGOPRIOR_E_INVALID, "point count out of range");
    if (d_count == nullptr) return ep_host::fail(EGOPRIOR_E_INVALID, "d_count is NULL");
    if (N > 0 && (d_xyz == nullptr || d_rgb == nullptr || d_points == nullptr))
        return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    CompactWs c;
    if (!carve_compact_ws(d_workspace, workspace_bytes, N > 0 ? N : 1, c))
        return ep_host::fail(EGOPRIOR_E_WORKSPACE, "filter workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (N == 0) {
        EP_LAUNCH("k_block_scan", ep::k_block_scan, dim3(1), dim3(1024), 0, st, c.block_count, c.block_offset, 0, d_count,
                           d_flags);
        return ep_host::check_launch("filter_cloud");
    }
    EP_LAUNCH("k_filter_flags", ep::k_filter_flags, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, st, d_xyz, N, c.flags);
    const bool fused = c.n_blocks <= EP_FUSED_SCAN_MAX;
    EP_LAUNCH("k_flag_block_count", ep::k_flag_block_count, dim3(c.n_blocks), dim3(256), 0, st, c.flags, N, c.block_count,
              fused ? d_flags : (int32_t*)nullptr);
    if (!fused)
        EP_LAUNCH("k_block_scan", ep::k_block_scan, dim3(1), dim3(1024), 0, st, c.block_count, c.block_offset, c.n_blocks,
                  d_count, d_flags);
    EP_LAUNCH("k_filter_scatter", ep::k_filter_scatter, dim3(c.n_blocks), dim3(256), 0, st, d_xyz, d_rgb, N, c.flags,
              c.block_count, fused ? (const uint32_t*)nullptr : c.block_offset, d_points, d_count, d_flags);
    return ep_host::check_launch("filter_cloud");
}

extern "C" size_t egoprior_meanbg_state_bytes(int32_t H, int32_t W) {
    if (H <= 0 || W <= 0) return 0;
    return (size_t)17 * (size_t)H * (size_t)W * sizeof(float);
}

namespace {
template <typename DT>
int meanbg_accumulate_t(float* d_state, const DT* d_depth, const uint8_t* d_rgb, const uint8_t* d_mask, int32_t F,
                        int64_t first_index, int32_t H, int32_t W, void* stream) {
    if (d_state == nullptr || d_depth == nullptr || d_rgb == nullptr)
        return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    if (H <= 0 || W <= 0 || F < 0 || first_index < 0) return ep_host::fail(EGOPRIOR_E_INVALID, "bad sizes");
    if (first_index + F >= (1ll << 19))
        return ep_host::fail(EGOPRIOR_E_UNSUPPORTED, "more than 2^19 frames changes the cascade run length");
    if (F == 0) return 0;
    const bool aligned4 = (((uintptr_t)d_rgb | (uintptr_t)d_mask) & 3u) == 0;  // the fast kernel reads 32-bit words
    if (W % 4 == 0 && aligned4 && (int64_t)H * W < (1ll << 31)) {
        EP_LAUNCH("k_meanbg_accumulate", ep::k_meanbg_accumulate_row4<DT>, dim3((W + 63) / 64, (H + 15) / 16), dim3(256), 0,
                  (hipStream_t)stream, d_state, d_depth, d_rgb, d_mask, F, (long long)first_index, H, W);
        return ep_host::check_launch("meanbg_accumulate");
    }
    const dim3 grid((W + EP_MASK_SX - 1) / EP_MASK_SX, (H + EP_MASK_SY - 1) / EP_MASK_SY);
    EP_LAUNCH("k_meanbg_accumulate", ep::k_meanbg_accumulate<DT>, grid, dim3(256), 0, (hipStream_t)stream, d_state, d_depth, d_rgb, d_mask,
                       F, (long long)first_index, H, W);
    return ep_host::check_launch("meanbg_accumulate");
}
}  // namespace

extern "C" int egoprior_meanbg_accumulate(float* d_state, const float* d_depth, const uint8_t* d_rgb,
                                          const uint8_t* d_mask, int32_t F, int64_t first_index, int32_t H, int32_t W,
                                          void* stream) {
    return meanbg_accumulate_t(d_state, d_depth, d_rgb, d_mask, F, first_index, H, W, stream);
}
extern "C" int egoprior_meanbg_accumulate_f16(float* d_state, const uint16_t* d_depth, const uint8_t* d_rgb,
                                              const uint8_t* d_mask, int32_t F, int64_t first_index, int32_t H,
                                              int32_t W, void* stream) {
    return meanbg_accumulate_t(d_state, reinterpret_cast<const __half*>(d_depth), d_rgb, d_mask, F, first_index, H, W,
                               stream);
}

extern "C" int egoprior_meanbg_finalize(const float* d_state, int32_t H, int32_t W, float* d_mean_depth,
                                        uint8_t* d_mean_rgb, void* stream) {
    if (d_state == nullptr || d_mean_depth == nullptr || d_mean_rgb == nullptr)
        return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    if (H <= 0 || W <= 0) return ep_host::fail(EGOPRIOR_E_INVALID, "bad sizes");
    const size_t HW = (size_t)H * W;
    EP_LAUNCH("k_meanbg_finalize", ep::k_meanbg_finalize, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_state, H, W, d_mean_depth, d_mean_rgb);
    return ep_host::check_launch("meanbg_finalize");
}

extern "C" int egoprior_meanbg_finalize_f32(const float* d_state, int32_t H, int32_t W, int32_t subsample,
                                            float* d_mean_depth, float* d_mean_rgb, void* stream) {
    if (d_state == nullptr || d_mean_depth == nullptr || d_mean_rgb == nullptr)
        return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    if (H <= 0 || W <= 0 || subsample <= 0) return ep_host::fail(EGOPRIOR_E_INVALID, "bad sizes");
    const int Hs = (H + subsample - 1) / subsample, Ws = (W + subsample - 1) / subsample;
    const size_t n = (size_t)Hs * Ws;
    EP_LAUNCH("k_meanbg_finalize_f32", ep::k_meanbg_finalize_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
              (hipStream_t)stream, d_state, H, W, subsample, Hs, Ws, d_mean_depth, d_mean_rgb);
    return ep_host::check_launch("meanbg_finalize_f32");
}

extern "C" int egoprior_meanbg_export(const float* d_state, int32_t H, int32_t W, float* d_sums4, int32_t* d_count,
                                      void* stream) {
    if (d_state == nullptr || d_sums4 == nullptr || d_count == nullptr)
        return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    if (H <= 0 || W <= 0) return ep_host::fail(EGOPRIOR_E_INVALID, "bad sizes");
    const size_t HW = (size_t)H * W;
    EP_LAUNCH("k_meanbg_export", ep::k_meanbg_export, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_state, H, W, d_sums4, d_count);
    return ep_host::check_launch("meanbg_export");
}

extern "C" int egoprior_meanbg_import(float* d_state, int32_t H, int32_t W, const float* d_sums4,
                                      const int32_t* d_count, void* stream) {
    if (d_state == nullptr || d_sums4 == nullptr || d_count == nullptr)
        return ep_host::fail(EGOPRIOR_E_INVALID, "NULL pointer");
    if (H <= 0 || W <= 0) return ep_host::fail(EGOPRIOR_E_INVALID, "bad sizes");
    const size_t HW = (size_t)H * W;
    EP_LAUNCH("k_meanbg_import", ep::k_meanbg_import, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       d_state, H, W, d_sums4, d_count);
    return ep_host::check_launch("meanbg_import");
}
