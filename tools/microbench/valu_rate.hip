// valu_rate.hip -- how many cycles a wave64 integer VALU instruction of each encoding costs a gfx950 SIMD.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate tools/microbench/valu_rate.hip && /tmp/valu_rate
//
// Each test issues N independent instructions of one kind per loop iteration from W waves per SIMD on every CU and
// reports SIMD cycles per wave-instruction = elapsed shader cycles (s_memtime) * waves per SIMD ... / instructions.
// kp_sw_kernel's inner loop is priced with these figures in DESIGN.md (the MI355X guide's "2 cycles per wave64 VALU" is
// measured on v_fma_f32; this asks the same of the integer / VOP3 / DPP / carry forms the DP cell is made of).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(unsigned *out, int iters, unsigned long long *cycles) {
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = blockIdx.x, d = 7, e = 11, f = 13, g = 17, h = 19;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // VOP2 e32: v_add_u32
            REP8(asm volatile("v_add_u32_e32 %0, %0, %4\n v_add_u32_e32 %1, %1, %4\n v_add_u32_e32 %2, %2, %4\n v_add_u32_e32 %3, %3, %4\n"
                              "v_add_u32_e32 %5, %5, %4\n v_add_u32_e32 %6, %6, %4\n v_add_u32_e32 %7, %7, %4\n v_add_u32_e32 %8, %8, %4"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f), "v"(g), "v"(h), "v"(e));)
        } else if (KIND == 1) {  // VOP3: v_add3_u32
            REP8(asm volatile("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5\n"
                              "v_add3_u32 %0, %0, %5, %4\n v_add3_u32 %1, %1, %5, %4\n v_add3_u32 %2, %2, %5, %4\n v_add3_u32 %3, %3, %5, %4"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (KIND == 2) {  // VOP3 v_max3_i32
            REP8(asm volatile("v_max3_i32 %0, %0, %4, %5\n v_max3_i32 %1, %1, %4, %5\n v_max3_i32 %2, %2, %4, %5\n v_max3_i32 %3, %3, %4, %5\n"
                              "v_max3_i32 %0, %0, %5, %4\n v_max3_i32 %1, %1, %5, %4\n v_max3_i32 %2, %2, %5, %4\n v_max3_i32 %3, %3, %5, %4"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (KIND == 3) {  // v_cmp e64 into an SGPR pair (result unused) -- 8 compares
            REP8(asm volatile("v_cmp_ge_i32_e64 s[20:21], %0, %1\n v_cmp_ge_i32_e64 s[22:23], %1, %2\n v_cmp_ge_i32_e64 s[24:25], %2, %3\n"
                              "v_cmp_ge_i32_e64 s[26:27], %3, %0\n v_cmp_ge_i32_e64 s[20:21], %0, %2\n v_cmp_ge_i32_e64 s[22:23], %1, %3\n"
                              "v_cmp_ge_i32_e64 s[24:25], %2, %0\n v_cmp_ge_i32_e64 s[26:27], %3, %1"
                              : : "v"(a), "v"(b), "v"(c), "v"(d) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");)
        } else if (KIND == 4) {  // v_cmp e32 (VCC) + v_addc e32 (VCC in/out): 4 pairs = 8 instructions
            REP8(asm volatile("v_cmp_ge_i32_e32 vcc, %4, %5\n v_addc_co_u32_e32 %0, vcc, %0, %0, vcc\n"
                              "v_cmp_ge_i32_e32 vcc, %5, %4\n v_addc_co_u32_e32 %1, vcc, %1, %1, vcc\n"
                              "v_cmp_ge_i32_e32 vcc, %4, %5\n v_addc_co_u32_e32 %2, vcc, %2, %2, vcc\n"
                              "v_cmp_ge_i32_e32 vcc, %5, %4\n v_addc_co_u32_e32 %3, vcc, %3, %3, vcc"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");)
        } else if (KIND == 5) {  // v_cmp e64 + VOP3 v_addc with SGPR carry: 4 pairs = 8 instructions (the kernel's form)
            REP8(asm volatile("v_cmp_ge_i32_e64 s[20:21], %4, %5\n v_addc_co_u32_e64 %0, s[28:29], %0, %0, s[20:21]\n"
                              "v_cmp_ge_i32_e64 s[22:23], %5, %4\n v_addc_co_u32_e64 %1, s[28:29], %1, %1, s[22:23]\n"
                              "v_cmp_ge_i32_e64 s[24:25], %4, %5\n v_addc_co_u32_e64 %2, s[28:29], %2, %2, s[24:25]\n"
                              "v_cmp_ge_i32_e64 s[26:27], %5, %4\n v_addc_co_u32_e64 %3, s[28:29], %3, %3, s[26:27]"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f)
                              : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "s28", "s29");)
        } else if (KIND == 6) {  // DPP moves
            REP8(asm volatile("v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %2, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %3, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %0, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %1, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %2, %5 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                              "v_mov_b32_dpp %3, %4 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (KIND == 7) {  // v_max_i32 e32
            REP8(asm volatile("v_max_i32_e32 %0, %0, %4\n v_max_i32_e32 %1, %1, %4\n v_max_i32_e32 %2, %2, %4\n v_max_i32_e32 %3, %3, %4\n"
                              "v_max_i32_e32 %0, %0, %5\n v_max_i32_e32 %1, %1, %5\n v_max_i32_e32 %2, %2, %5\n v_max_i32_e32 %3, %3, %5"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (KIND == 8) {  // v_bfe_i32 (VOP3)
            REP8(asm volatile("v_bfe_i32 %0, %4, %5, 6\n v_bfe_i32 %1, %4, %5, 6\n v_bfe_i32 %2, %4, %5, 6\n v_bfe_i32 %3, %4, %5, 6\n"
                              "v_bfe_i32 %0, %5, %4, 6\n v_bfe_i32 %1, %5, %4, 6\n v_bfe_i32 %2, %5, %4, 6\n v_bfe_i32 %3, %5, %4, 6"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        } else if (KIND == 9) {  // v_cndmask e64 with an SGPR mask
            REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[30:31]\n v_cndmask_b32_e64 %1, %1, %4, s[30:31]\n"
                              "v_cndmask_b32_e64 %2, %2, %4, s[30:31]\n v_cndmask_b32_e64 %3, %3, %4, s[30:31]\n"
                              "v_cndmask_b32_e64 %0, %0, %5, s[30:31]\n v_cndmask_b32_e64 %1, %1, %5, s[30:31]\n"
                              "v_cndmask_b32_e64 %2, %2, %5, s[30:31]\n v_cndmask_b32_e64 %3, %3, %5, s[30:31]"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "s30", "s31");)
        } else if (KIND == 10) {  // v_lshl_or_b32 (VOP3)
            REP8(asm volatile("v_lshl_or_b32 %0, %0, 1, %4\n v_lshl_or_b32 %1, %1, 1, %4\n v_lshl_or_b32 %2, %2, 1, %4\n v_lshl_or_b32 %3, %3, 1, %4\n"
                              "v_lshl_or_b32 %0, %0, 1, %5\n v_lshl_or_b32 %1, %1, 1, %5\n v_lshl_or_b32 %2, %2, 1, %5\n v_lshl_or_b32 %3, %3, 1, %5"
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND>
void run(const char *name, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 2000;  // 256-thread blocks = 4 waves = one per SIMD of a CU
    unsigned *out; unsigned long long *cyc;
    hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, (size_t)blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, (size_t)blocks * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
    const double instr_per_wave = (double)iters * 64.0;  // 8 x 8 instructions per iteration
    // s_memtime ticks at 100 MHz on this family; use wall time and the measured clock instead: cycles per SIMD =
    // ms * f_clk; every SIMD ran waves_per_simd waves
    printf("%-44s waves/SIMD %d  %8.3f ms  -> %6.2f ns per wave-instruction per SIMD (x f_clk = cycles)   [memtime ticks/instr %.3f]\n", name,
           waves_per_simd, ms, ms * 1e6 / (instr_per_wave * waves_per_simd), mean / instr_per_wave);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 4, 8}) {
        run<0>("v_add_u32_e32 (VOP2)", w);
        run<7>("v_max_i32_e32 (VOP2)", w);
        run<1>("v_add3_u32 (VOP3)", w);
        run<2>("v_max3_i32 (VOP3)", w);
        run<8>("v_bfe_i32 (VOP3)", w);
        run<10>("v_lshl_or_b32 (VOP3)", w);
        run<9>("v_cndmask_b32_e64 (SGPR mask)", w);
        run<3>("v_cmp_ge_i32_e64 -> SGPR pair", w);
        run<4>("v_cmp_e32 + v_addc_co_u32_e32 (VCC)", w);
        run<5>("v_cmp_e64 + v_addc_co_u32_e64 (SGPR carry)", w);
        run<6>("v_mov_b32_dpp row_shr/shl:1", w);
    }
    return 0;
}
