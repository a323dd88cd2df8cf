// kp_sort.hip -- per-assembly sort of the anchor regions.  Plain library sort (rocPRIM segmented radix sort, 64-bit
// keys, one segment per assembly); keys are unique, so any correct sort gives the same order.  `end_bit` = one past the
// highest key bit any anchor of this database can set (the gene/strand field sits on top).
#include <string.h>

#include <cstring>

#include <rocprim/device/device_segmented_radix_sort.hpp>

#include "kp_internal.h"

void kp_launch_segments(const uint32_t *count, uint32_t cap, int n_asm, uint32_t *seg_begin, uint32_t *seg_end,
                        hipStream_t stream);

int kp_sort_anchors(kp_ctx *ctx, uint64_t *keys_in, uint64_t *keys_out, const uint32_t *d_count, uint32_t cap,
                    int32_t n_asm, void **temp, size_t *temp_bytes, uint32_t *d_seg_begin, uint32_t *d_seg_end,
                    int end_bit, hipStream_t stream) {
    if (n_asm == 0) return KP_OK;
    kp_launch_segments(d_count, cap, n_asm, d_seg_begin, d_seg_end, stream);
    const unsigned int size = (unsigned int)((size_t)n_asm * cap);
    size_t need = 0;
    KP_HIP_CHECK(ctx, rocprim::segmented_radix_sort_keys(nullptr, need, keys_in, keys_out, size, (unsigned)n_asm,
                                                         d_seg_begin, d_seg_end, 0, end_bit, stream));
    if (need > *temp_bytes) {
        if (*temp) KP_HIP_CHECK(ctx, hipFree(*temp));
        *temp = nullptr;
        KP_HIP_CHECK(ctx, hipMalloc(temp, need));
        *temp_bytes = need;
    }
    KP_HIP_CHECK(ctx, rocprim::segmented_radix_sort_keys(*temp, need, keys_in, keys_out, size, (unsigned)n_asm,
                                                         d_seg_begin, d_seg_end, 0, end_bit, stream));
    return KP_OK;
}
