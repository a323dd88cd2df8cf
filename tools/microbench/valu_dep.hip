// valu_dep.hip -- how often one wave can issue VALU instructions that depend on each other (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_dep tools/microbench/valu_dep.hip && /tmp/valu_dep
// CHAINS independent dependency chains per wave (1 = every instruction needs the previous one's result), W waves per
// SIMD; v_add_u32 (a 2-cycle instruction) and v_pk_max_u16 (4-cycle).  Reported: ns per wave-instruction per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x

template <int CHAINS, bool SLOW>
__global__ __launch_bounds__(256) void dep_kernel(unsigned *out, int iters) {
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = blockIdx.x, d = 7, e = 11 + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        if (!SLOW) {
            if (CHAINS == 1) { REP8(asm volatile("v_add_u32_e32 %0, %0, %1\n v_add_u32_e32 %0, %0, %1\n v_add_u32_e32 %0, %0, %1\n v_add_u32_e32 %0, %0, %1\n"
                                                 "v_add_u32_e32 %0, %0, %1\n v_add_u32_e32 %0, %0, %1\n v_add_u32_e32 %0, %0, %1\n v_add_u32_e32 %0, %0, %1" : "+v"(a) : "v"(e));) }
            if (CHAINS == 2) { REP8(asm volatile("v_add_u32_e32 %0, %0, %2\n v_add_u32_e32 %1, %1, %2\n v_add_u32_e32 %0, %0, %2\n v_add_u32_e32 %1, %1, %2\n"
                                                 "v_add_u32_e32 %0, %0, %2\n v_add_u32_e32 %1, %1, %2\n v_add_u32_e32 %0, %0, %2\n v_add_u32_e32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(e));) }
            if (CHAINS == 4) { REP8(asm volatile("v_add_u32_e32 %0, %0, %4\n v_add_u32_e32 %1, %1, %4\n v_add_u32_e32 %2, %2, %4\n v_add_u32_e32 %3, %3, %4\n"
                                                 "v_add_u32_e32 %0, %0, %4\n v_add_u32_e32 %1, %1, %4\n v_add_u32_e32 %2, %2, %4\n v_add_u32_e32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        } else {
            if (CHAINS == 1) { REP8(asm volatile("v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %0, %0, %1\n"
                                                 "v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %0, %0, %1\n v_pk_max_u16 %0, %0, %1" : "+v"(a) : "v"(e));) }
            if (CHAINS == 2) { REP8(asm volatile("v_pk_max_u16 %0, %0, %2\n v_pk_max_u16 %1, %1, %2\n v_pk_max_u16 %0, %0, %2\n v_pk_max_u16 %1, %1, %2\n"
                                                 "v_pk_max_u16 %0, %0, %2\n v_pk_max_u16 %1, %1, %2\n v_pk_max_u16 %0, %0, %2\n v_pk_max_u16 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(e));) }
            if (CHAINS == 4) { REP8(asm volatile("v_pk_max_u16 %0, %0, %4\n v_pk_max_u16 %1, %1, %4\n v_pk_max_u16 %2, %2, %4\n v_pk_max_u16 %3, %3, %4\n"
                                                 "v_pk_max_u16 %0, %0, %4\n v_pk_max_u16 %1, %1, %4\n v_pk_max_u16 %2, %2, %4\n v_pk_max_u16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <int CHAINS, bool SLOW>
void run(int waves) {
    const int blocks = 256 * waves, iters = 4000;
    unsigned *out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((dep_kernel<CHAINS, SLOW>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((dep_kernel<CHAINS, SLOW>), dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-14s chains %d  waves/SIMD %d  %7.3f ns per wave-instruction per SIMD  (%.2f ns between a wave's instructions)\n",
           SLOW ? "v_pk_max_u16" : "v_add_u32", CHAINS, waves, ms * 1e6 / ((double)iters * 64.0 * waves), ms * 1e6 / ((double)iters * 64.0));
    (void)hipFree(out);
}

int main() {
    for (int w : {1, 2, 3, 4, 5, 6, 8}) {
        run<1, false>(w); run<2, false>(w); run<4, false>(w);
        run<1, true>(w); run<2, true>(w); run<4, true>(w);
    }
    return 0;
}
