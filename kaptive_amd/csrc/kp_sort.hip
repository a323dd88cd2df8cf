// kp_sort.hip -- per-assembly sort of the anchor regions.  Plain library sort (rocPRIM segmented radix sort, 64-bit
// keys, one segment per assembly); keys are unique, so any correct sort gives the same order.  `end_bit` = one past the
// highest key bit any anchor of this database can set (the gene/strand field sits on top).
#include <string.h>

#include <cstring>

#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "kp_internal.h"

// rocPRIM has no tuned configuration for this key type on gfx950 and falls back to 6-bit digits with 128-thread
// blocks; an assembly's segment is tens of thousands of keys, so wider digits (fewer passes) and larger blocks pay
#ifndef KP_SORT_RADIX_BITS
#define KP_SORT_RADIX_BITS 8
#define KP_SORT_BLOCK 256
#define KP_SORT_ITEMS 16
#endif
using SortConfig = rocprim::segmented_radix_sort_config<KP_SORT_RADIX_BITS, rocprim::kernel_config<KP_SORT_BLOCK, KP_SORT_ITEMS>,
                                                        rocprim::WarpSortConfig<32, 4, 256, 3000, 32, 4, 256>, true>;

void kp_launch_segments(const uint32_t *count, uint32_t cap, int n_asm, uint32_t *seg_begin, uint32_t *seg_end,
                        hipStream_t stream);

int kp_sort_anchors(kp_ctx *ctx, uint64_t *keys_in, uint64_t *keys_out, const uint32_t *d_count, uint32_t cap,
                    int32_t n_asm, void **temp, size_t *temp_bytes, uint32_t *d_seg_begin, uint32_t *d_seg_end,
                    int end_bit, hipStream_t stream) {
    if (n_asm == 0) return KP_OK;
    kp_launch_segments(d_count, cap, n_asm, d_seg_begin, d_seg_end, stream);
    const unsigned int size = (unsigned int)((size_t)n_asm * cap);
    size_t need = 0;
    KP_HIP_CHECK(ctx, rocprim::segmented_radix_sort_keys<SortConfig>(nullptr, need, keys_in, keys_out, size, (unsigned)n_asm,
                                                         d_seg_begin, d_seg_end, 0, end_bit, stream));
    if (need > *temp_bytes) {
        if (*temp) KP_HIP_CHECK(ctx, hipFree(*temp));
        *temp = nullptr;
        KP_HIP_CHECK(ctx, hipMalloc(temp, need));
        *temp_bytes = need;
    }
    KP_HIP_CHECK(ctx, rocprim::segmented_radix_sort_keys<SortConfig>(*temp, need, keys_in, keys_out, size, (unsigned)n_asm,
                                                         d_seg_begin, d_seg_end, 0, end_bit, stream));
    return KP_OK;
}
