// valu_rate2.hip -- issue cost of candidate VALU instructions on gfx950 (companion of valu_rate.hip; same method:
// 4 waves per SIMD on every CU, 64 independent-ish instructions per loop iteration, wall time / instructions).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate2 tools/microbench/valu_rate2.hip && /tmp/valu_rate2
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x

#define K2(NAME, OP)                                                                                                       \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters) {                                                \
        unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = blockIdx.x, d = 7, e = 11 + threadIdx.x, f = 13;            \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            REP8(asm volatile(OP " %0, %0, %4\n" OP " %1, %1, %4\n" OP " %2, %2, %4\n" OP " %3, %3, %4\n" OP               \
                                 " %0, %0, %5\n" OP " %1, %1, %5\n" OP " %2, %2, %5\n" OP " %3, %3, %5"                    \
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                                         \
                              : "v"(e), "v"(f));)                                                                          \
        }                                                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;                                                        \
    }
#define K3(NAME, OP)                                                                                                       \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters) {                                                \
        unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = blockIdx.x, d = 7, e = 11 + threadIdx.x, f = 13;            \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            REP8(asm volatile(OP " %0, %0, %4, %5\n" OP " %1, %1, %4, %5\n" OP " %2, %2, %4, %5\n" OP " %3, %3, %4, %5\n" OP \
                                 " %0, %0, %5, %4\n" OP " %1, %1, %5, %4\n" OP " %2, %2, %5, %4\n" OP " %3, %3, %5, %4"    \
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                                         \
                              : "v"(e), "v"(f));)                                                                          \
        }                                                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;                                                        \
    }
// 64-bit (register pair) packed forms
#define K2W(NAME, OP)                                                                                                      \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters) {                                                \
        unsigned long long a = threadIdx.x, b = threadIdx.x * 3 + 1, c = blockIdx.x, d = 7, e = 11 + threadIdx.x, f = 13;  \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            REP8(asm volatile(OP " %0, %0, %4\n" OP " %1, %1, %4\n" OP " %2, %2, %4\n" OP " %3, %3, %4\n" OP               \
                                 " %0, %0, %5\n" OP " %1, %1, %5\n" OP " %2, %2, %5\n" OP " %3, %3, %5"                    \
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                                         \
                              : "v"(e), "v"(f));)                                                                          \
        }                                                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a ^ b ^ c ^ d);                                            \
    }
#define K3W(NAME, OP)                                                                                                      \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters) {                                                \
        unsigned long long a = threadIdx.x, b = threadIdx.x * 3 + 1, c = blockIdx.x, d = 7, e = 11 + threadIdx.x, f = 13;  \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            REP8(asm volatile(OP " %0, %0, %4, %5\n" OP " %1, %1, %4, %5\n" OP " %2, %2, %4, %5\n" OP " %3, %3, %4, %5\n" OP \
                                 " %0, %0, %5, %4\n" OP " %1, %1, %5, %4\n" OP " %2, %2, %5, %4\n" OP " %3, %3, %5, %4"    \
                              : "+v"(a), "+v"(b), "+v"(c), "+v"(d)                                                         \
                              : "v"(e), "v"(f));)                                                                          \
        }                                                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (unsigned)(a ^ b ^ c ^ d);                                            \
    }
// compare into VCC (result unused)
#define KC(NAME, OP)                                                                                                       \
    __global__ __launch_bounds__(256) void NAME(unsigned *out, int iters) {                                                \
        unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = blockIdx.x, d = 7;                                          \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            REP8(asm volatile(OP " vcc, %0, %1\n" OP " vcc, %1, %2\n" OP " vcc, %2, %3\n" OP " vcc, %3, %0\n" OP           \
                                 " vcc, %0, %2\n" OP " vcc, %1, %3\n" OP " vcc, %2, %0\n" OP " vcc, %3, %1"                \
                              :                                                                                            \
                              : "v"(a), "v"(b), "v"(c), "v"(d)                                                             \
                              : "vcc");)                                                                                   \
        }                                                                                                                  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d;                                                        \
    }

K2(k_add_u32, "v_add_u32_e32")
K2(k_sub_u32, "v_sub_u32_e32")
K2(k_and, "v_and_b32_e32")
K2(k_or, "v_or_b32_e32")
K2(k_xor, "v_xor_b32_e32")
K2(k_lshl, "v_lshlrev_b32_e32")
K2(k_lshr, "v_lshrrev_b32_e32")
K2(k_max_i32, "v_max_i32_e32")
K2(k_max_u32, "v_max_u32_e32")
K2(k_min_u32, "v_min_u32_e32")
K2(k_max_f32, "v_max_f32_e32")
K2(k_add_f32, "v_add_f32_e32")
K2(k_mul_f32, "v_mul_f32_e32")
K2(k_max_i16, "v_max_i16_e32")
K2(k_add_u16, "v_add_u16_e32")
K2(k_pk_max_i16, "v_pk_max_i16")
K2(k_pk_add_i16, "v_pk_add_i16")
K2(k_pk_add_u16, "v_pk_add_u16")
K2(k_pk_max_f16, "v_pk_max_f16")
K2(k_pk_add_f16, "v_pk_add_f16")
K2(k_mul_u24, "v_mul_u32_u24_e32")
K2(k_mov, "v_mov_b32_e32 %0, %4 ;")
K3(k_fma_f32, "v_fma_f32")
K3(k_mad_u24, "v_mad_u32_u24")
K3(k_mad_i24, "v_mad_i32_i24")
K3(k_max3_f32, "v_max3_f32")
K3(k_med3_i32, "v_med3_i32")
K3(k_perm, "v_perm_b32")
K3(k_alignbit, "v_alignbit_b32")
K3(k_and_or, "v_and_or_b32")
K3(k_or3, "v_or3_b32")
K3(k_add_lshl, "v_add_lshl_u32")
K3(k_lshl_add, "v_lshl_add_u32")
K3(k_xad, "v_xad_u32")
K3(k_pk_mad_i16, "v_pk_mad_i16")
K3(k_pk_fma_f16, "v_pk_fma_f16")
K3(k_max3_i16, "v_max3_i16")
K3(k_pk_max3_f16, "v_pk_maximum3_f16")
K3(k_maximum3_f32, "v_maximum3_f32")
K2(k_ashr, "v_ashrrev_i32_e32")
K2(k_max_u16, "v_max_u16_e32")
K2(k_min_i16, "v_min_i16_e32")
K2(k_sub_u16, "v_sub_u16_e32")
K2(k_lshl_b16, "v_lshlrev_b16_e32")
K2(k_lshr_b16, "v_lshrrev_b16_e32")
K2(k_mul_lo_u16, "v_mul_lo_u16_e32")
K2(k_pk_lshr_b16, "v_pk_lshrrev_b16")
K2(k_pk_ashr_i16, "v_pk_ashrrev_i16")
K2(k_pk_max_u16, "v_pk_max_u16")
K2(k_pk_sub_u16, "v_pk_sub_u16")
K2(k_pk_sub_i16, "v_pk_sub_i16")
K2(k_add_dpp, "v_add_u32_dpp %0, %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1 ;")
K2(k_add_lit, "v_add_u32_e32 %0, 0x12345, %0 ;")
K2(k_add_sgpr, "v_add_u32_e32 %0, s4, %0 ;")
K2(k_and_lit, "v_and_b32_e32 %0, 0x80008000, %0 ;")
K2(k_not, "v_not_b32_e32 %0, %4 ;")
K2(k_add_co, "v_add_co_u32_e32 %0, vcc, %0, %4 ;")
K2(k_addc, "v_addc_co_u32_e32 %0, vcc, %0, %4, vcc ;")
K2(k_cndmask_vcc, "v_cndmask_b32_e32 %0, %0, %4, vcc ;")
K2(k_mov_sdwa, "v_mov_b32_sdwa %0, %4 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 ;")
K2(k_add_sdwa, "v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1 ;")
K3(k_bfi, "v_bfi_b32")
K2(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %4 ;")
K2(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %4 ;")
K2(k_ffbl, "v_ffbl_b32_e32 %0, %4 ;")
K2(k_bcnt, "v_bcnt_u32_b32 %0, %0, %4 ;")
K3(k_sad_u16, "v_sad_u16")
K3(k_add3, "v_add3_u32")
K3(k_fmac, "v_fma_f32")
K2W(k_pk_add_f32, "v_pk_add_f32")
K2W(k_pk_mul_f32, "v_pk_mul_f32")
K3W(k_pk_fma_f32, "v_pk_fma_f32")
K2W(k_lshl_b64, "v_lshlrev_b64 %0, 1, %4 ;")
KC(k_cmp_gt_i32, "v_cmp_gt_i32_e32")
KC(k_cmp_gt_f32, "v_cmp_gt_f32_e32")
KC(k_cmp_eq_u32, "v_cmp_eq_u32_e32")
KC(k_cmp_gt_i16, "v_cmp_gt_i16_e32")

template <typename F>
void run(const char *name, F kernel) {
    const int waves = 4, blocks = 256 * waves, iters = 4000;
    unsigned *out;
    (void)hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters);  // warm (also spins the clock up)
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %8.3f ms  %6.3f ns per wave-instruction per SIMD\n", name, ms, ms * 1e6 / ((double)iters * 64.0 * waves));
    (void)hipFree(out);
}
#define RUN(k) run(#k, k)

int main() {
    RUN(k_add_u32); RUN(k_add_u32); RUN(k_sub_u32); RUN(k_and); RUN(k_or); RUN(k_xor); RUN(k_lshl); RUN(k_lshr);
    RUN(k_max_i32); RUN(k_max_u32); RUN(k_min_u32); RUN(k_max_f32); RUN(k_add_f32); RUN(k_mul_f32); RUN(k_max_i16);
    RUN(k_add_u16); RUN(k_pk_max_i16); RUN(k_pk_add_i16); RUN(k_pk_add_u16); RUN(k_pk_max_f16); RUN(k_pk_add_f16);
    RUN(k_mul_u24); RUN(k_mov); RUN(k_fma_f32); RUN(k_mad_u24); RUN(k_mad_i24); RUN(k_max3_f32); RUN(k_med3_i32);
    RUN(k_perm); RUN(k_alignbit); RUN(k_and_or); RUN(k_or3); RUN(k_add_lshl); RUN(k_lshl_add); RUN(k_xad);
    RUN(k_pk_mad_i16); RUN(k_pk_fma_f16); RUN(k_max3_i16); RUN(k_pk_max3_f16); RUN(k_maximum3_f32);
    RUN(k_pk_add_f32); RUN(k_pk_mul_f32); RUN(k_pk_fma_f32); RUN(k_lshl_b64);
    RUN(k_cmp_gt_i32); RUN(k_cmp_gt_f32); RUN(k_cmp_eq_u32); RUN(k_cmp_gt_i16);
    RUN(k_ashr); RUN(k_max_u16); RUN(k_min_i16); RUN(k_sub_u16); RUN(k_lshl_b16); RUN(k_lshr_b16); RUN(k_mul_lo_u16);
    RUN(k_pk_lshr_b16); RUN(k_pk_ashr_i16); RUN(k_pk_max_u16); RUN(k_pk_sub_u16); RUN(k_pk_sub_i16); RUN(k_add_dpp); RUN(k_add_lit);
    RUN(k_add_sgpr); RUN(k_and_lit); RUN(k_not); RUN(k_add_co); RUN(k_addc); RUN(k_cndmask_vcc); RUN(k_mov_sdwa); RUN(k_add_sdwa);
    RUN(k_bfi); RUN(k_mul_lo_u32); RUN(k_mul_hi_u32); RUN(k_ffbl); RUN(k_bcnt); RUN(k_sad_u16); RUN(k_add3);
    return 0;
}
