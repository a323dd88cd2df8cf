// kp_chain.hip -- sorted anchors -> band tasks (the "chaining" step of the aligner, include/kp_spec.h).
//
// Stands in for the chaining stage inside rammappy's map_batch (reference call site
// src/kaptive/serotyping/core.py:154).  Anchors of an assembly arrive sorted by (gene*2+strand, diagonal, query pos).
// Pass 1 labels every anchor with its contig.  Pass 2 runs one thread per anchor; the threads that sit on a hard
// break (first anchor, new gene/strand, new contig, diagonal jump > KP_DIAG_GAP) walk their run forward, cut it
// whenever it would span more than KP_MAX_SPREAD diagonals, and append one task per surviving cluster to the list of
// its band-width class.  Runs are short (tens of anchors), so the sequential walk is not a bottleneck.
#include "kp_internal.h"

namespace {

constexpr int WALK = 8;

__global__ __launch_bounds__(256) void kp_anchor_contig_kernel(KpBatchView b, const uint64_t *__restrict__ keys,
                                                               const uint32_t *__restrict__ count, uint32_t cap,
                                                               int32_t *__restrict__ contig) {
    const int a = blockIdx.y;
    uint32_t n = count[a];
    if (n > cap) n = cap;
    const int c0 = b.asm_first_ctg[a], nc = b.asm_first_ctg[a + 1] - c0;
    const int32_t *starts = b.ctg_start + c0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t k = keys[(size_t)a * cap + i];
        const int32_t t = (int32_t)KP_KEY_DIAG(k) - KP_DIAG_BIAS + (int32_t)KP_KEY_QPOS(k);
        int lo = 0, hi = nc;  // last contig starting at or before t
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (starts[mid] <= t) lo = mid + 1; else hi = mid;
        }
        contig[(size_t)a * cap + i] = lo - 1;
    }
}

// Tasks are staged per block in LDS and appended to the global lists with one atomic per block and class: a batch
// produces ~10^6 tasks for three counters, which would otherwise serialise on those three words.
constexpr int STAGE0 = 192, STAGE_REST = 32;  // staged tasks per block for the narrowest class / each wider class

struct TaskStage {
    KpTask t0[STAGE0], rest[KP_N_CLASSES - 1][STAGE_REST];
    uint32_t n[KP_N_CLASSES], base[KP_N_CLASSES];
    __device__ KpTask *list(int cls) { return cls == 0 ? t0 : rest[cls - 1]; }
    __device__ static uint32_t room(int cls) { return cls == 0 ? STAGE0 : STAGE_REST; }
};

__device__ __forceinline__ void flush_cluster(int a, uint32_t gs, int ctg, uint32_t d0, uint32_t dmax, uint32_t qmin,
                                              uint32_t qmax, int cnt, KpTask *tasks, uint32_t *task_count,
                                              uint32_t task_cap, TaskStage &st) {
    if (cnt < KP_MIN_ANCHORS || (int)(qmax - qmin) + KP_K < KP_MIN_SEED_SPAN) return;
    int margin = KP_BAND_MARGIN_NARROW, need = (int)(dmax - d0) + 1 + 2 * KP_BAND_MARGIN_NARROW, w = 16, cls = 0;
    if (need > 16) {
        margin = KP_BAND_MARGIN;
        need = (int)(dmax - d0) + 1 + 2 * KP_BAND_MARGIN;
        w = need <= 32 ? 32 : (need <= 64 ? 64 : 128);
        cls = w == 32 ? 1 : (w == 64 ? 2 : 3);
    }
    KpTask t;
    t.asm_id = a; t.gs = (int32_t)gs; t.contig = ctg; t.width = w; t.n_anchors = cnt;
    t.lo = (int32_t)d0 - KP_DIAG_BIAS - margin - (w - need) / 2;
    t.qmin = (int32_t)qmin; t.qmax = (int32_t)qmax;
    const uint32_t s = atomicAdd(&st.n[cls], 1u);
    if (s < TaskStage::room(cls)) {
        st.list(cls)[s] = t;
        return;
    }
    const uint32_t slot = atomicAdd(&task_count[cls], 1u);  // stage full: append directly
    if (slot < task_cap) tasks[(size_t)cls * task_cap + slot] = t;  // beyond cap: counted, not stored (host retries)
}

__global__ __launch_bounds__(256) void kp_chain_kernel(const uint64_t *__restrict__ keys,
                                                       const int32_t *__restrict__ contig,
                                                       const uint32_t *__restrict__ count, uint32_t cap,
                                                       KpTask *__restrict__ tasks, uint32_t *__restrict__ task_count,
                                                       uint32_t task_cap) {
    __shared__ TaskStage st;
    const int a = blockIdx.y;
    uint32_t n = count[a];
    if (n > cap) n = cap;
    const uint64_t *k = keys + (size_t)a * cap;
    const int32_t *c = contig + (size_t)a * cap;
    if (threadIdx.x < KP_N_CLASSES) st.n[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t gs = KP_KEY_GS(k[i]);
        const int ctg = c[i];
        if (i > 0 && KP_KEY_GS(k[i - 1]) == gs && c[i - 1] == ctg &&
            KP_KEY_DIAG(k[i]) - KP_KEY_DIAG(k[i - 1]) <= KP_DIAG_GAP)
            continue;  // not the head of a run
        uint32_t d0 = KP_KEY_DIAG(k[i]), dprev = d0, q = KP_KEY_QPOS(k[i]);
        uint32_t qmin = q, qmax = q;
        int cnt = 1;
        bool open = true;
        for (uint32_t j0 = i + 1; j0 < n && open; j0 += WALK) {  // fetch WALK anchors at a time: the walk is latency-bound
            uint64_t kk[WALK];
            int32_t cc[WALK];
#pragma unroll
            for (int u = 0; u < WALK; ++u) {
                const uint32_t j = j0 + u < n ? j0 + u : n - 1;
                kk[u] = k[j];
                cc[u] = c[j];
            }
#pragma unroll
            for (int u = 0; u < WALK; ++u) {
                if (!open || j0 + u >= n) { open = open && j0 + u < n; continue; }
                const uint32_t d = KP_KEY_DIAG(kk[u]);
                if (KP_KEY_GS(kk[u]) != gs || cc[u] != ctg || d - dprev > KP_DIAG_GAP) { open = false; continue; }
                q = KP_KEY_QPOS(kk[u]);
                if (d - d0 > KP_MAX_SPREAD) {  // soft cut: close the cluster, open the next one here
                    flush_cluster(a, gs, ctg, d0, dprev, qmin, qmax, cnt, tasks, task_count, task_cap, st);
                    d0 = d; qmin = qmax = q; cnt = 0;
                }
                dprev = d;
                cnt++;
                qmin = min(qmin, q);
                qmax = max(qmax, q);
            }
        }
        flush_cluster(a, gs, ctg, d0, dprev, qmin, qmax, cnt, tasks, task_count, task_cap, st);
    }
    __syncthreads();
    if (threadIdx.x < KP_N_CLASSES) {
        const uint32_t room = TaskStage::room(threadIdx.x);
        const uint32_t m = st.n[threadIdx.x] < room ? st.n[threadIdx.x] : room;
        st.n[threadIdx.x] = m;
        st.base[threadIdx.x] = m ? atomicAdd(&task_count[threadIdx.x], m) : 0u;
    }
    __syncthreads();
    for (int cls = 0; cls < KP_N_CLASSES; ++cls) {
        const KpTask *src = st.list(cls);
        for (uint32_t i = threadIdx.x; i < st.n[cls]; i += blockDim.x) {
            const uint32_t slot = st.base[cls] + i;
            if (slot < task_cap) tasks[(size_t)cls * task_cap + slot] = src[i];
        }
    }
}

__global__ void kp_segments_kernel(const uint32_t *__restrict__ count, uint32_t cap, int n_asm,
                                   uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_asm) return;
    const uint32_t n = count[a] > cap ? cap : count[a];
    seg_begin[a] = (uint32_t)a * cap;
    seg_end[a] = (uint32_t)a * cap + n;
}

// ---- task order: counting sort of each width class by query length, longest first ------------------------------------------
// Tasks of one wave run in lock step for as many steps as the longest of them needs, so neighbours in the processing
// order should have similar lengths; starting with the long ones also keeps the tail of the launch short.
__device__ __forceinline__ int length_bucket(int qlen) {
    const int b = qlen >> 5;
    return 63 - (b > 63 ? 63 : b);
}

__global__ __launch_bounds__(256) void kp_task_hist_kernel(KpGenes genes, const KpTask *__restrict__ tasks,
                                                           const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                           uint32_t *__restrict__ hist) {
    __shared__ uint32_t s_h[64];
    const int cls = blockIdx.y;
    uint32_t n = task_count[cls];
    if (n > task_cap) n = task_cap;
    if (threadIdx.x < 64) s_h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        atomicAdd(&s_h[length_bucket(genes.len[tasks[(size_t)cls * task_cap + i].gs >> 1])], 1u);
    __syncthreads();
    if (threadIdx.x < 64 && s_h[threadIdx.x]) atomicAdd(&hist[cls * 64 + threadIdx.x], s_h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void kp_task_scatter_kernel(KpGenes genes, const KpTask *__restrict__ tasks,
                                                              const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                              uint32_t *__restrict__ hist, uint32_t *__restrict__ order) {
    __shared__ uint32_t s_start[64], s_h[64], s_base[64];
    const int cls = blockIdx.y;
    uint32_t n = task_count[cls];
    if (n > task_cap) n = task_cap;
    if (threadIdx.x == 0) {  // bucket starts from the finished histogram (64 entries: not worth a scan)
        uint32_t acc = 0;
        for (int k = 0; k < 64; ++k) { s_start[k] = acc; acc += hist[cls * 64 + k]; }
    }
    if (threadIdx.x < 64) s_h[threadIdx.x] = 0;
    __syncthreads();
    // each block handles one contiguous chunk so that it can reserve its slots with one atomic per bucket
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per, hi = min(n, lo + per);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x)
        atomicAdd(&s_h[length_bucket(genes.len[tasks[(size_t)cls * task_cap + i].gs >> 1])], 1u);
    __syncthreads();
    if (threadIdx.x < 64) {
        s_base[threadIdx.x] = s_h[threadIdx.x] ? atomicAdd(&hist[KP_N_CLASSES * 64 + cls * 64 + threadIdx.x], s_h[threadIdx.x]) : 0u;
        s_h[threadIdx.x] = 0;
    }
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int k = length_bucket(genes.len[tasks[(size_t)cls * task_cap + i].gs >> 1]);
        order[(size_t)cls * task_cap + s_start[k] + s_base[k] + atomicAdd(&s_h[k], 1u)] = i;
    }
}

}  // namespace

void kp_launch_task_order(const KpGenes &genes, const KpTask *tasks, const uint32_t *task_count, uint32_t task_cap,
                          uint32_t *hist, uint32_t *order, hipStream_t stream) {
    const dim3 grid(128, KP_N_CLASSES), block(256);
    hipLaunchKernelGGL(kp_task_hist_kernel, grid, block, 0, stream, genes, tasks, task_count, task_cap, hist);
    hipLaunchKernelGGL(kp_task_scatter_kernel, grid, block, 0, stream, genes, tasks, task_count, task_cap, hist, order);
}

void kp_launch_segments(const uint32_t *count, uint32_t cap, int n_asm, uint32_t *seg_begin, uint32_t *seg_end,
                        hipStream_t stream) {
    if (n_asm == 0) return;
    hipLaunchKernelGGL(kp_segments_kernel, dim3((n_asm + 255) / 256), dim3(256), 0, stream, count, cap, n_asm, seg_begin,
                       seg_end);
}

void kp_launch_chain(const KpBatchView &b, const uint64_t *sorted_anchors, const uint32_t *anchor_count, uint32_t cap,
                     int32_t *anchor_contig, KpTask *tasks, uint32_t *task_count, uint32_t task_cap,
                     hipStream_t stream) {
    if (b.n_asm == 0) return;
    const dim3 grid(32, b.n_asm), block(256);
    hipLaunchKernelGGL(kp_anchor_contig_kernel, grid, block, 0, stream, b, sorted_anchors, anchor_count, cap,
                       anchor_contig);
    hipLaunchKernelGGL(kp_chain_kernel, grid, block, 0, stream, sorted_anchors, anchor_contig, anchor_count, cap, tasks,
                       task_count, task_cap);
}
