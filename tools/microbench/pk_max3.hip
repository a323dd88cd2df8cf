// pk_max3.hip -- can v_pk_maximum3_f16 (new in gfx950) stand in for two v_pk_max_u16 on the fill kernel's biased scores?
// Positive half-precision bit patterns below 0x7C00 (infinity) order like unsigned integers, so a three-way float
// maximum of such patterns is the three-way integer maximum -- provided the hardware returns the winning operand's bits
// unchanged (denormal patterns 0x0001..0x03FF included) in the default mode of a HIP kernel.  This program checks that
// exhaustively for pairs (every a, b in [0, 0x7C00) against a third operand sweep) and measures the issue rate next to
// v_pk_max_u16 (method of valu_rate.hip).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pk_max3 tools/microbench/pk_max3.hip && /tmp/pk_max3
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned max3(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm volatile("v_pk_maximum3_f16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// every (a, b) with a in the block's range, b = all; c cycles through a few patterns; low and high halves carry different values
__global__ void check(unsigned long long *bad, unsigned *first_bad) {
    const unsigned LIM = 0x7C00u;
    const unsigned a = blockIdx.x;  // 0 .. LIM-1
    for (unsigned b = threadIdx.x; b < LIM; b += blockDim.x) {
        const unsigned cs[6] = {0u, 1u, 0x3FFu, 0x400u, (a + b) % LIM, LIM - 1u};
        for (int i = 0; i < 6; ++i) {
            const unsigned c = cs[i];
            const unsigned a2 = (LIM - 1u - a), b2 = (b * 7u + 3u) % LIM, c2 = (c * 5u + 11u) % LIM;
            const unsigned got = max3(a | (a2 << 16), b | (b2 << 16), c | (c2 << 16));
            const unsigned w1 = a > b ? (a > c ? a : c) : (b > c ? b : c), w2 = a2 > b2 ? (a2 > c2 ? a2 : c2) : (b2 > c2 ? b2 : c2);
            if (got != (w1 | (w2 << 16))) {
                if (atomicAdd(bad, 1ull) == 0ull) { first_bad[0] = a | (a2 << 16); first_bad[1] = b | (b2 << 16); first_bad[2] = c | (c2 << 16); first_bad[3] = got; }
            }
        }
    }
}

#define REP8(x) x x x x x x x x
__global__ __launch_bounds__(256) void rate_max3(unsigned *out, int iters) {
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = blockIdx.x, d = 7, e = 11 + threadIdx.x, f = 13;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_pk_maximum3_f16 %0, %0, %4, %5\nv_pk_maximum3_f16 %1, %1, %4, %5\nv_pk_maximum3_f16 %2, %2, %4, %5\n"
                          "v_pk_maximum3_f16 %3, %3, %4, %5\nv_pk_maximum3_f16 %0, %0, %5, %4\nv_pk_maximum3_f16 %1, %1, %5, %4\n"
                          "v_pk_maximum3_f16 %2, %2, %5, %4\nv_pk_maximum3_f16 %3, %3, %5, %4"
                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}
__global__ __launch_bounds__(256) void rate_max2(unsigned *out, int iters) {
    unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = blockIdx.x, d = 7, e = 11 + threadIdx.x, f = 13;
    for (int i = 0; i < iters; ++i) {
        REP8(asm volatile("v_pk_max_u16 %0, %0, %4\nv_pk_max_u16 %1, %1, %4\nv_pk_max_u16 %2, %2, %4\nv_pk_max_u16 %3, %3, %4\n"
                          "v_pk_max_u16 %0, %0, %5\nv_pk_max_u16 %1, %1, %5\nv_pk_max_u16 %2, %2, %5\nv_pk_max_u16 %3, %3, %5"
                          : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;
}

template <typename K>
static double ns_per_instr(K kernel, unsigned *out) {
    const int blocks = 256 * 4, iters = 2000;  // 4 blocks of 4 waves per CU: 4 waves per SIMD
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, 10);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x iters x 64 instructions
    return ms * 1e6 / (4.0 * iters * 64.0);
}

int main() {
    unsigned long long *bad;
    unsigned *first_bad, *out;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&first_bad, 16); (void)hipMalloc(&out, 256 * 4 * 256 * 4);
    (void)hipMemset(bad, 0, 8);
    hipLaunchKernelGGL(check, dim3(0x7C00), dim3(256), 0, 0, bad, first_bad);
    unsigned long long h_bad = 0;
    unsigned h_first[4] = {0, 0, 0, 0};
    (void)hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h_first, first_bad, 16, hipMemcpyDeviceToHost);
    printf("v_pk_maximum3_f16 as a three-way unsigned maximum of half-words in [0, 0x7C00): %llu mismatches in %llu triples\n", h_bad,
           (unsigned long long)0x7C00 * 0x7C00 * 6);
    if (h_bad) printf("  first: a=%08x b=%08x c=%08x -> %08x\n", h_first[0], h_first[1], h_first[2], h_first[3]);
    printf("issue, ns per wave64 instruction and SIMD (4 waves per SIMD):  v_pk_max_u16 %.3f   v_pk_maximum3_f16 %.3f\n",
           ns_per_instr(rate_max2, out), ns_per_instr(rate_max3, out));
    return 0;
}
