// Check of the lane-write transposition used by the splat sweep: hit masks built with v_cmp + v_writelane against a direct evaluation.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned long long u64;
__device__ __forceinline__ uint32_t write_lane(uint32_t v, uint32_t s, int L) {
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(s), "n"(L));
    return v;
}
__global__ void k(const float2* recs, uint2* out_new, uint2* out_ref, float r2) {
    const int lane = threadIdx.x;
    const float xf = 0.1f * (float)(lane & 7), yf = 0.1f * (float)(lane >> 3);
    const float2 r = recs[lane];
    float2v rxy; rxy.x = r.x; rxy.y = r.y;
    uint32_t hm_lo = 0, hm_hi = 0;
    float2v dx2[4]; float dy2[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float2v d;
        d.x = rxy.x - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xf), 2 * c));
        d.y = rxy.x - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(xf), 2 * c + 1));
        dx2[c] = d * d;
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const float d = rxy.y - __int_as_float(__builtin_amdgcn_readlane(__float_as_int(yf), 8 * rr));
        dy2[rr] = d * d;
    }
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        float2v rowv; rowv.x = dy2[rr]; rowv.y = dy2[rr];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float2v d2 = dx2[c] + rowv;
            const u64 h0 = __ballot(d2.x < r2), h1 = __ballot(d2.y < r2);
            hm_lo = write_lane(hm_lo, (uint32_t)h0, rr * 8 + 2 * c);
            hm_hi = write_lane(hm_hi, (uint32_t)(h0 >> 32), rr * 8 + 2 * c);
            hm_lo = write_lane(hm_lo, (uint32_t)h1, rr * 8 + 2 * c + 1);
            hm_hi = write_lane(hm_hi, (uint32_t)(h1 >> 32), rr * 8 + 2 * c + 1);
        }
    }
    out_new[lane] = make_uint2(hm_lo, hm_hi);
    // reference: this lane = pixel, loop over records
    u64 m = 0;
    for (int j = 0; j < 64; ++j) {
        const float px = __shfl(r.x, j), py = __shfl(r.y, j);
        const float dx = px - xf, dy = py - yf;
        const float d2 = dx * dx + dy * dy;
        if (d2 < r2) m |= 1ull << j;
    }
    out_ref[lane] = make_uint2((uint32_t)m, (uint32_t)(m >> 32));
}
int main() {
    std::vector<float2> h(64);
    srand(1);
    for (auto& v : h) { v.x = (rand() % 1000) / 1000.0f; v.y = (rand() % 1000) / 1000.0f; }
    float2* d; uint2 *a, *b;
    hipMalloc(&d, 64 * sizeof(float2)); hipMalloc(&a, 64 * sizeof(uint2)); hipMalloc(&b, 64 * sizeof(uint2));
    hipMemcpy(d, h.data(), 64 * sizeof(float2), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, b, 0.09f);
    std::vector<uint2> ha(64), hb(64);
    hipMemcpy(ha.data(), a, 64 * sizeof(uint2), hipMemcpyDeviceToHost);
    hipMemcpy(hb.data(), b, 64 * sizeof(uint2), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) if (ha[i].x != hb[i].x || ha[i].y != hb[i].y) { if (bad < 8) printf("lane %d: new %08x %08x ref %08x %08x\n", i, ha[i].x, ha[i].y, hb[i].x, hb[i].y); ++bad; }
    printf("mismatching lanes: %d\n", bad);
    return bad != 0;
}
