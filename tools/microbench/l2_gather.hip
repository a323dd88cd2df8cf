// l2_gather.hip -- the rate at which an MI355X serves independent random 8-byte reads out of a small table: the roof of
// kp_scan_kernel's presence-filter probes (one 8-byte block of a 2 MB blocked Bloom filter per selected k-mer).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/l2_gather tools/microbench/l2_gather.hip && /tmp/l2_gather
//
// Every lane keeps `INFLIGHT` independent loads in the air (the scan keeps 8), addresses come from a per-lane xorshift,
// nothing else is computed.  Tables: 32 KB (fits the CU's vector L1), 2 MB and 4 MB (fit every XCD's 4 MB L2 -- each XCD
// caches its own copy), 64 MB (falls out of the L2s into the 256 MB MALL) and 1 GB (HBM).  Reported: lane-reads per
// second; a wave instruction of 64 random addresses is 64 requests to the L2 unless the L1 catches them.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <vector>

template <int INFLIGHT>
__global__ __launch_bounds__(256) void gather_kernel(const uint64_t *__restrict__ table, uint32_t mask, int rounds,
                                                    uint64_t *__restrict__ out) {
    uint32_t x = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    uint64_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        uint64_t v[INFLIGHT];
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) {
            x ^= x << 13; x ^= x >> 17; x ^= x << 5;
            v[i] = table[x & mask];
        }
#pragma unroll
        for (int i = 0; i < INFLIGHT; ++i) acc += v[i];
    }
    out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int INFLIGHT>
double run(const uint64_t *table, size_t entries, int waves_per_simd, uint64_t *out) {
    const int blocks = 256 * waves_per_simd, rounds = 2048 / INFLIGHT * 8;  // 16384 reads per lane
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(gather_kernel<INFLIGHT>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(entries - 1), 16, out);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(gather_kernel<INFLIGHT>, dim3(blocks), dim3(256), 0, 0, table, (uint32_t)(entries - 1), rounds, out);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)blocks * 256.0 * rounds * INFLIGHT / (ms * 1e-3) / 1e9;  // G lane-reads per second
}

int main() {
    const size_t max_entries = (size_t)1 << 27;  // 1 GB
    uint64_t *table, *out;
    (void)hipMalloc(&table, max_entries * 8);
    (void)hipMemset(table, 1, max_entries * 8);
    (void)hipMalloc(&out, (size_t)256 * 16 * 256 * 8);
    printf("%-10s %-12s %10s %10s %10s\n", "table", "waves/SIMD", "4 in air", "8 in air", "16 in air");
    for (size_t bytes : {(size_t)32 << 10, (size_t)2 << 20, (size_t)4 << 20, (size_t)64 << 20, (size_t)1 << 30}) {
        for (int w : {2, 4, 8}) {
            const size_t entries = bytes / 8;
            const double a = run<4>(table, entries, w, out), b = run<8>(table, entries, w, out), c = run<16>(table, entries, w, out);
            char name[32];
            snprintf(name, sizeof name, bytes >= (1 << 20) ? "%zu MB" : "%zu KB", bytes >= (1 << 20) ? bytes >> 20 : bytes >> 10);
            printf("%-10s %-12d %8.1f G %8.1f G %8.1f G   lane-reads/s\n", name, w, a, b, c);
        }
    }
    return 0;
}
