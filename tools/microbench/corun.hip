// corun.hip -- do a VALU-bound kernel and an L2-request-bound kernel share the MI355X, or take turns?
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/corun tools/microbench/corun.hip && /tmp/corun
// V: single-wave blocks that keep 128 VGPRs live and issue dependent integer VALU work (the shape of kp_sw_kernel), N per CU.
// G: 256-thread blocks with 32 KB of LDS doing independent random 8-byte reads out of a 2 MB table (the shape of
//    kp_scan_kernel).  Each is timed alone, then both are launched on two streams at once.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(64, 4) void valu_kernel(unsigned *out, int iters) {
    __shared__ unsigned pad[2048];  // 8 KB, as the fill kernel
    unsigned r[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) r[i] = threadIdx.x * (i + 1) + blockIdx.x;
    pad[threadIdx.x] = r[0];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 96; ++i) r[i] = max(r[i] + r[(i + 1) % 96], r[(i + 7) % 96]) ^ (r[(i + 13) % 96] >> 3);
    }
    unsigned acc = pad[(threadIdx.x + 1) & 63];
#pragma unroll
    for (int i = 0; i < 96; ++i) acc ^= r[i];
    out[blockIdx.x * 64 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void gather_kernel(const uint64_t *__restrict__ table, uint32_t mask, int rounds, uint64_t *out) {
    __shared__ uint64_t lds[4096];  // 32 KB, as the scan kernel
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint64_t acc = 0;
    lds[threadIdx.x] = x;
    for (int r = 0; r < rounds; ++r) {
        uint64_t v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; v[i] = table[x & mask]; }
#pragma unroll
        for (int i = 0; i < 8; ++i) acc += v[i];
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc + lds[(threadIdx.x + 1) & 255];
}

int main() {
    const size_t entries = (2u << 20) / 8;
    uint64_t *table, *gout; unsigned *vout;
    (void)hipMalloc(&table, entries * 8); (void)hipMemset(table, 1, entries * 8);
    (void)hipMalloc(&gout, (size_t)256 * 64 * 256 * 8); (void)hipMalloc(&vout, (size_t)256 * 64 * 64 * 4);
    hipStream_t sv, sg; (void)hipStreamCreateWithFlags(&sv, hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&sg, hipStreamNonBlocking);
    hipEvent_t v0, v1, g0, g1; (void)hipEventCreate(&v0); (void)hipEventCreate(&v1); (void)hipEventCreate(&g0); (void)hipEventCreate(&g1);
    const int g_blocks = 256 * 5, g_rounds = 3000;  // ~5 ms alone
    for (int per_cu : {8, 12, 16}) {
        const int v_blocks = 256 * per_cu, v_iters = 16 * 1200 / per_cu;  // same total work whatever the occupancy
        float tv = 0, tg = 0, tv2 = 0, tg2 = 0;
        for (int rep = 0; rep < 2; ++rep) {  // (first repetition warms up)
            (void)hipEventRecord(v0, sv); hipLaunchKernelGGL(valu_kernel, dim3(v_blocks), dim3(64), 0, sv, vout, v_iters); (void)hipEventRecord(v1, sv);
            (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&tv, v0, v1);
            (void)hipEventRecord(g0, sg); hipLaunchKernelGGL(gather_kernel, dim3(g_blocks), dim3(256), 0, sg, table, (uint32_t)(entries - 1), g_rounds, gout); (void)hipEventRecord(g1, sg);
            (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&tg, g0, g1);
            // both at once: V first (it is resident when G arrives), G repeated so that it overlaps all of V
            (void)hipEventRecord(v0, sv); hipLaunchKernelGGL(valu_kernel, dim3(v_blocks), dim3(64), 0, sv, vout, v_iters); (void)hipEventRecord(v1, sv);
            (void)hipEventRecord(g0, sg);
            for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(gather_kernel, dim3(g_blocks), dim3(256), 0, sg, table, (uint32_t)(entries - 1), g_rounds, gout);
            (void)hipEventRecord(g1, sg);
            (void)hipDeviceSynchronize(); (void)hipEventElapsedTime(&tv2, v0, v1); (void)hipEventElapsedTime(&tg2, g0, g1);
        }
        printf("V at %2d blocks/CU: alone V %.2f ms, G %.2f ms;  together: V %.2f ms, two G launches %.2f ms (alone they would take %.2f)\n",
               per_cu, tv, tg, tv2, tg2, 2 * tg);
    }
    return 0;
}
