// Micro-benchmark: how many cycles a SIMD of gfx950 needs per wave64 instruction of the kinds the splat kernel is made
// of, with 1..4 waves resident per SIMD.  Every wave runs ITER iterations of a block of 32 independent instructions of
// one kind and reports its own elapsed shader cycles (s_memtime); cycles per instruction per SIMD = elapsed / (waves
// per SIMD x instructions).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define ITER 2000

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

template <int KIND>
__global__ void __launch_bounds__(256) k_bench(unsigned long long* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    typedef float float2v __attribute__((ext_vector_type(2)));
    float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    unsigned long long q0 = (unsigned long long)threadIdx.x * 977u + 1, q1 = q0 * 3, q2 = q0 * 5, q3 = q0 * 7;
    unsigned int u0 = threadIdx.x, u1 = u0 * 3 + 1, u2 = u0 * 5 + 2, u3 = u0 * 7 + 3;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {  // v_fma_f32, 8 independent chains
            REP8(asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %6, %6, %6, %7"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 1) {  // v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_mul_f32 %1, %1, %2\n v_pk_mul_f32 %2, %2, %3\n v_pk_mul_f32 %3, %3, %0"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
        } else if (KIND == 2) {  // v_sub_f32 with a DPP row_share operand
            REP8(asm volatile("v_sub_f32_dpp %0, %4, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_sub_f32_dpp %1, %5, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_sub_f32_dpp %2, %6, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_sub_f32_dpp %3, %7, %0 row_newbcast:9 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
        } else if (KIND == 3) {  // 64-bit compare + four selects (one compare-exchange of two 64-bit keys)
            REP8(asm volatile("v_cmp_lt_u64_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 %4, %5, %6, s[20:21]\n v_cndmask_b32_e64 %5, %6, %7, s[20:21]\n"
                              "v_cndmask_b32_e64 %6, %7, %4, s[20:21]\n v_cndmask_b32_e64 %7, %4, %5, s[20:21]"
                              : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : "s20", "s21");)
        } else if (KIND == 4) {  // v_alignbit_b32
            REP8(asm volatile("v_alignbit_b32 %0, %0, %1, 31\n v_alignbit_b32 %1, %1, %2, 31\n v_alignbit_b32 %2, %2, %3, 31\n v_alignbit_b32 %3, %3, %0, 31"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
        } else if (KIND == 5) {  // v_cmp_lt_u64 alone (to VCC)
            REP8(asm volatile("v_cmp_lt_u64_e32 vcc, %0, %1\n v_cmp_lt_u64_e32 vcc, %1, %2\n v_cmp_lt_u64_e32 vcc, %2, %3\n v_cmp_lt_u64_e32 vcc, %3, %0"
                              : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3) : : "vcc");)
        } else if (KIND == 6) {  // v_lshl_add_u64
            REP8(asm volatile("v_lshl_add_u64 %0, %1, 3, %0\n v_lshl_add_u64 %1, %2, 3, %1\n v_lshl_add_u64 %2, %3, 3, %2\n v_lshl_add_u64 %3, %0, 3, %3"
                              : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));)
        } else if (KIND == 7) {  // v_mul_f32 + v_add_f32 (plain 32-bit)
            REP8(asm volatile("v_mul_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_add_f32 %3, %3, %0"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));)
        } else if (KIND == 8) {  // v_cndmask_b32 e32 (VCC)
            REP8(asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc\n v_cndmask_b32_e32 %1, %1, %2, vcc\n v_cndmask_b32_e32 %2, %2, %3, vcc\n v_cndmask_b32_e32 %3, %3, %0, vcc"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : );)
        } else if (KIND == 9) {  // v_pk_add_f32
            REP8(asm volatile("v_pk_add_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %2, %2, %3\n v_pk_add_f32 %3, %3, %0"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));)
        } else if (KIND == 10) {  // v_sub_u32 / v_add_u32 / v_and / v_bfe mix
            REP8(asm volatile("v_add_u32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_lshlrev_b32 %2, 1, %3\n v_xor_b32 %3, %3, %0"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
        } else if (KIND == 11) {  // v_cmp_gt_f32 into SGPR pairs
            REP8(asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1\n v_cmp_gt_f32_e64 s[22:23], %1, %2\n v_cmp_gt_f32_e64 s[24:25], %2, %3\n v_cmp_gt_f32_e64 s[26:27], %3, %0"
                              : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if (KIND == 12) {  // v_writelane_b32 from SGPRs, immediate lane
            REP8(asm volatile("v_writelane_b32 %0, s20, 3\n v_writelane_b32 %1, s21, 17\n v_writelane_b32 %2, s22, 40\n v_writelane_b32 %3, s23, 63"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : : );)
        } else if (KIND == 13) {  // hit test of two pixels against 64 records held one per lane: packed add, two compares, four lane writes
            REP8(asm volatile("v_pk_add_f32 %4, %5, %6\n v_cmp_gt_f32_e64 s[20:21], %7, %8\n v_cmp_gt_f32_e64 s[22:23], %8, %7\n"
                              "v_writelane_b32 %0, s20, 3\n v_writelane_b32 %1, s21, 3\n v_writelane_b32 %0, s22, 4\n v_writelane_b32 %1, s23, 4"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(p0) : "v"(p1), "v"(p2), "v"(a7), "v"(a6) : "s20", "s21", "s22", "s23");)
        } else if (KIND == 14) {  // hit test of two records against the 64 pixels held one per lane: four DPP subtractions, four packed operations, two alignbits
            REP8(asm volatile("v_sub_f32_dpp %4, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_sub_f32_dpp %5, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_sub_f32_dpp %6, %10, %11 row_newbcast:3 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_sub_f32_dpp %7, %10, %11 row_newbcast:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_add_f32 %1, %1, %2\n v_pk_add_f32 %1, %1, %3\n"
                              "v_alignbit_b32 %0, %0, %4, 31\n v_alignbit_b32 %0, %0, %5, 31"
                              : "+v"(u0), "+v"(p0), "+v"(p1), "+v"(p2), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(a4), "v"(a5), "v"(a6), "v"(a7));)
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y;
    unsigned long long qq = q0 ^ q1 ^ q2 ^ q3 ^ u0 ^ u1 ^ u2 ^ u3;
    if (s == 12345.678f && qq == 42) out[0] = 1;  // keeps the chains alive
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int insts_per_iter, unsigned long long* d_out) {
    for (int wps = 1; wps <= 4; ++wps) {  // workgroups of 256 threads (one wave per SIMD) per CU
        const int blocks = 256 * wps;
        hipMemset(d_out, 0, sizeof(unsigned long long) * (1 + 4 * 1024));
        hipLaunchKernelGGL(k_bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 1.0f);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(1 + 4 * 1024);
        hipMemcpy(h.data(), d_out, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        std::vector<unsigned long long> v(h.begin() + 1, h.begin() + 1 + 4 * blocks);
        std::sort(v.begin(), v.end());
        const double med = (double)v[v.size() / 2];
        const double per_inst_wave = med / ((double)ITER * insts_per_iter);
        printf("%-34s waves/SIMD %d: %.2f cycles per instruction per wave, %.2f per SIMD\n", name, wps, per_inst_wave,
               per_inst_wave / wps);
    }
}

int main() {
    unsigned long long* d_out;
    hipMalloc(&d_out, sizeof(unsigned long long) * (1 + 4 * 1024));
    run<0>("v_fma_f32", 32, d_out);
    run<7>("v_mul_f32 / v_add_f32", 32, d_out);
    run<10>("32-bit integer mix", 32, d_out);
    run<1>("v_pk_mul_f32", 32, d_out);
    run<9>("v_pk_add_f32", 32, d_out);
    run<2>("v_sub_f32_dpp row_newbcast", 32, d_out);
    run<4>("v_alignbit_b32", 32, d_out);
    run<8>("v_cndmask_b32 (vcc)", 32, d_out);
    run<5>("v_cmp_lt_u64", 32, d_out);
    run<6>("v_lshl_add_u64", 32, d_out);
    run<3>("v_cmp_lt_u64 + 4 v_cndmask (sgpr)", 40, d_out);
    run<11>("v_cmp_gt_f32 (sgpr pair)", 32, d_out);
    run<12>("v_writelane_b32", 32, d_out);
    run<13>("2 pixels x 64 records (7 inst)", 56, d_out);
    run<14>("2 records x 64 pixels (10 inst)", 80, d_out);
    hipFree(d_out);
    return 0;
}
