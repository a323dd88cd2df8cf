// kp_capi.hip -- the C ABI of libkaptive_amd.so (include/kaptive_amd.h): context, resident database, batches,
// orchestration of the alignment kernels on the context's stream, and host-side finalisation of the hit table.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>

#include "kp_internal.h"
#include <cctype>
#include <sys/mman.h>
#include "kp_sketch.h"
#include "kp_reduce_core.h"

// kp_reduce.hip
void kp_launch_hit_finalise(const KpBatchView &b, const int32_t *gene_len, const KpTask *tasks, const KpSwResult *results, const uint8_t *task_drop,
                            const uint32_t *task_count, uint32_t task_cap, kp_hit *raw, uint32_t *n_raw, uint32_t hit_cap,
                            uint64_t *keys, kp_hit *hits, uint32_t *n_hits, unsigned long long *cells, const float *ln_half,
                            const float *ln_int, const KpJoin *joins, const uint32_t *join_count, uint32_t join_cap, hipStream_t stream);
void kp_launch_score(const KpBatchView &b, const kp_hit *hits, const uint32_t *n_hits, uint32_t hit_cap,
                     const KpTypingDb &db, double min_cov, double *scores, int32_t *counts, hipStream_t stream);
void kp_launch_reduce(const KpBatchView &b, const kp_hit *hits, const uint32_t *n_hits, uint32_t hit_cap,
                      const KpTypingDb &db, const KpTypingParams &prm, const int32_t *best, uint64_t *keys,
                      uint32_t *order, uint8_t *kept_flag, KpKept *kept, int kept_cap, KpPiece *pieces, int piece_cap,
                      KpAsmSummary *summary, uint8_t *prot, int prot_cap, int32_t *pair_q_off, int32_t *pair_q_len,
                      int32_t *pair_t_off, int32_t *pair_t_len, int32_t *n_pairs, int32_t *pair_base,
                      hipStream_t stream);
void kp_launch_states(const KpBatchView &b, const KpTypingDb &db, const KpTypingParams &prm, KpKept *kept, int kept_cap,
                      KpAsmSummary *summary, const int32_t *dp8, const int32_t *pair_base, hipStream_t stream);

namespace {

constexpr size_t ORDER_HEAD = KP_ORDER_HEAD;  // task-order histogram, cursors and per-class counts (kp_chain.hip)

std::mutex g_err_mutex;
std::string g_global_error = "";

std::atomic<long long> g_dev_allocs{0};  // re-allocations of device buffers since the process started

template <class T>
struct DevBuf {  // growable device allocation
    T *p = nullptr;
    size_t n = 0;
    hipError_t reserve(size_t want) {
        if (want <= n) return hipSuccess;
        // hipFree / hipMalloc wait for the whole device: a buffer that grows while passes are in flight stalls the pipeline for
        // as long as those passes take (kp_device_allocations counts them, so that a caller can show a stream of batches does none)
        if (p) {
            g_dev_allocs.fetch_add(1, std::memory_order_relaxed);
            if (std::getenv("KAPTIVE_AMD_DEBUG_ALLOC"))
                std::fprintf(stderr, "[DevBuf] re-allocation: %zu -> %zu items of %zu bytes\n", n, want, sizeof(T));
            want += want / 8;  // a buffer that had to grow once gets head-room: sizes that creep by a per cent from batch to
                               // batch (the fullest assembly of a batch decides several of them) must not re-allocate each time
        }
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(want, 1) * sizeof(T));
        if (e == hipSuccess) n = want;
        return e;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

// Typing tables of one database.  Its genes are the contiguous range [gene_lo, gene_hi) of the context's genes; all
// tables use gene indices relative to gene_lo.
struct KpTypingGroup {
    int32_t gene_lo = 0, gene_hi = 0;
    DevBuf<uint16_t> d_gene_locus, d_gene_pos;
    DevBuf<uint8_t> d_gene_extra, d_prot_db;
    DevBuf<int8_t> d_gene_strand;
    DevBuf<int32_t> d_locus_off, d_locus_len, d_prot_db_off, d_prot_db_len;
    KpTypingDb typing{};
    int max_db_prot_len = 0;
    void release() {
        d_gene_locus.release(); d_gene_pos.release(); d_gene_extra.release(); d_prot_db.release(); d_gene_strand.release();
        d_locus_off.release(); d_locus_len.release(); d_prot_db_off.release(); d_prot_db_len.release();
    }
};

// Reduction state of one (batch, typing group): the group's hits (copied out of the batch's hit table with gene indices
// rebased) and everything score / reduce / typing produce for it.
struct KpTypingRun {
    // every group works on streams of its own (highest priority), so the reductions of several databases over one
    // batch overlap: they are chains of short, low-occupancy kernels.  The streams belong to the context (one pair per
    // group, shared by the work sets: a context's streams should not outnumber the runtime's hardware queues)
    hipStream_t stream = nullptr, aux = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    bool split = false;  // hits / hit_n belong to the work set's most recent alignment pass
    const kp_hit *hits = nullptr;    // the group's hit rows: d_hits, or the work set's table itself when the group
    const uint32_t *hit_n = nullptr;  // spans every gene of the context (no copy)
    DevBuf<kp_hit> d_hits;
    DevBuf<uint32_t> d_hit_n;
    DevBuf<uint64_t> d_keys;   // cull keys
    DevBuf<uint32_t> d_order;
    DevBuf<uint8_t> d_flag;
    DevBuf<int32_t> d_dp_scratch;
    int kept_cap = 0, piece_cap = 0, prot_cap = 0;
    DevBuf<uint32_t> d_pack;  // kept / piece rows cut to the strides the caller asked for (kp_batch_typing)
    DevBuf<uint8_t> d_prot;
    DevBuf<double> d_scores;
    DevBuf<int32_t> d_lcounts, d_best, d_pairs, d_dp;
    DevBuf<KpKept> d_kept;
    DevBuf<KpPiece> d_pieces;
    DevBuf<KpAsmSummary> d_summary;
    KpTypingParams prm{};
    bool scored = false, reduced = false;
    bool sums_valid = false;  // h_sums / max_kept / max_pieces belong to the most recent reduction
    std::vector<KpAsmSummary> h_sums;
    int32_t max_kept = 1, max_pieces = 1;
    void release() {
        if (stream) { (void)hipStreamSynchronize(stream); stream = nullptr; }
        if (aux) { (void)hipStreamSynchronize(aux); aux = nullptr; }
        if (ev_fork) { (void)hipEventDestroy(ev_fork); ev_fork = nullptr; }
        if (ev_join) { (void)hipEventDestroy(ev_join); ev_join = nullptr; }
        d_keys.release(); d_order.release(); d_flag.release(); d_dp_scratch.release();
        d_hits.release(); d_hit_n.release(); d_pack.release(); d_prot.release(); d_scores.release(); d_lcounts.release();
        d_best.release(); d_pairs.release(); d_dp.release(); d_kept.release(); d_pieces.release(); d_summary.release();
    }
};

struct kp_batch;

// Options of a context: defaults come from the environment once, at kp_ctx_create; kp_ctx_set_option changes them.
struct KpOptions {
    uint32_t anchor_cap = 1u << 17, tasks_per_asm = 4096, hit_cap = 4096;
    uint32_t trace_kb_per_asm = 2048;  // first guess for the DP trace buffer (a 5 Mbp K-locus assembly needs ~12 MB)
    uint32_t kept_cap = 256, piece_cap = 32, prot_cap = 32768;
    int scan_mode = 0;           // KAPTIVE_AMD_SCAN_ABLATE (tools/scan_ablate.py)
    int library_sort = 0;        // anchors through kp_anchor_compact + rocPRIM's segmented radix sort instead of kp_bsort.hip
    uint32_t upload_piece_mb = 4096;  // H2D copies of a batch's words are enqueued in pieces of this size (batch_make)
    int readback_copy_engine = 0;   // results read back with hipMemcpyAsync instead of the read-back kernel (see Fetch)
    int spin_wait = 0;              // host waits spin on the stream (the runtime's default) instead of blocking on an interrupt
};

// Page-locked host memory the library holds (kp_host_alloc and the batches' table staging), for kp_host_pinned_bytes.
// Two kinds: small blocks straight from hipHostMalloc; large ones (the callers' shard buffers) as anonymous memory
// advised to use huge pages and then registered -- locking 0.8 GB of 4 KB pages costs 139 ms and 84 ms to give back,
// of 2 MB pages 54 ms (51 of them the first touch, which a caller that fills the block before it locks it spreads over
// its own threads: kp_host_reserve / kp_host_lock) and 31 ms; a process that ends holding 3.2 GB leaves the kernel
// 410 ms of work against 168 (tools/microbench/pin_thp.cpp).
static std::mutex g_pin_mutex;
struct PinBlock { size_t bytes; bool mapped, locked; };
static std::unordered_map<void *, PinBlock> g_pin_blocks;
static size_t g_pin_bytes = 0;
constexpr size_t HUGE_PAGE = (size_t)2 << 20;
static hipError_t pinned_alloc(void **out, size_t bytes) {
    // portable: usable by every device's context whichever thread (and current device) allocates it -- a reader thread of
    // the CLI takes page-locked buffers while the driving thread holds the context
    const hipError_t e = hipHostMalloc(out, bytes, hipHostMallocPortable);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pin_mutex);
        g_pin_blocks[*out] = PinBlock{bytes, false, true};
        g_pin_bytes += bytes;
    }
    return e;
}
static void *mapped_alloc(size_t bytes) {  // 2 MB-aligned anonymous memory, huge pages where the system grants them on advice
    bytes = (bytes + HUGE_PAGE - 1) & ~(HUGE_PAGE - 1);
    char *raw = (char *)mmap(nullptr, bytes + HUGE_PAGE, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (raw == MAP_FAILED) return nullptr;
    char *p = (char *)(((uintptr_t)raw + HUGE_PAGE - 1) & ~(uintptr_t)(HUGE_PAGE - 1));
    if (p > raw) munmap(raw, (size_t)(p - raw));
    if (raw + HUGE_PAGE > p) munmap(p + bytes, (size_t)(raw + HUGE_PAGE - p));
    (void)madvise(p, bytes, MADV_HUGEPAGE);  // (refused where transparent huge pages are off: plain pages then)
    std::lock_guard<std::mutex> lk(g_pin_mutex);
    g_pin_blocks[p] = PinBlock{bytes, true, false};
    return p;
}
static hipError_t mapped_lock(void *p) {
    size_t bytes;
    {
        std::lock_guard<std::mutex> lk(g_pin_mutex);
        auto it = g_pin_blocks.find(p);
        if (it == g_pin_blocks.end() || !it->second.mapped) return hipErrorInvalidValue;
        if (it->second.locked) return hipSuccess;
        bytes = it->second.bytes;
    }
    const hipError_t e = hipHostRegister(p, bytes, hipHostRegisterPortable);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pin_mutex);
        g_pin_blocks[p].locked = true;
        g_pin_bytes += bytes;
    }
    return e;
}
static void pinned_free(void *p) {
    if (!p) return;
    PinBlock b{0, false, true};
    {
        std::lock_guard<std::mutex> lk(g_pin_mutex);
        auto it = g_pin_blocks.find(p);
        if (it != g_pin_blocks.end()) {
            b = it->second;
            if (b.locked) g_pin_bytes -= b.bytes;
            g_pin_blocks.erase(it);
        }
    }
    if (b.mapped) {
        if (b.locked) (void)hipHostUnregister(p);
        munmap(p, b.bytes);
    } else {
        (void)hipHostFree(p);
    }
}

// Device copy of one batch's input (packed words + tables).  Recycled through the context (hipFree synchronises the
// device, so a stream of batches must not free anything).
struct KpInput {
    DevBuf<uint32_t> d_words;
    DevBuf<int64_t> d_asm_word_off;
    DevBuf<int32_t> d_ctg_start, d_ctg_len, d_asm_first_ctg, d_n_runs, d_asm_first_nrun;
    uint8_t *h_stage = nullptr;  // pinned staging of the tables (the caller's copies may be freed on return)
    size_t h_stage_bytes = 0;
    hipEvent_t ready = nullptr;  // recorded on the copy stream after the last H2D copy of the batch
    void release() {
        d_words.release(); d_asm_word_off.release(); d_ctg_start.release(); d_ctg_len.release();
        d_asm_first_ctg.release(); d_n_runs.release(); d_asm_first_nrun.release();
        pinned_free(h_stage);
        h_stage = nullptr; h_stage_bytes = 0;
        if (ready) (void)hipEventDestroy(ready);
        ready = nullptr;
    }
};

// Work set: every device buffer an alignment pass and the reductions after it write, and the results they leave.  A
// context owns KP_WORK_SLOTS of them and hands them to batches round-robin at kp_batch_align, so a stream of batches
// allocates nothing after the first few and keeps what it learnt about buffer sizes (the caps live in the context).
struct KpWork {
    kp_batch *owner = nullptr;
    KpKeyBits key_bits{16, 30};  // compact anchor keys of the most recent alignment pass
    uint32_t anchor_cap = 0, task_cap = 0, hit_cap = 0;  // what the buffers of the most recent pass were sized for
    uint64_t cand_cap = 0;
    DevBuf<uint64_t> d_anchors_a, d_anchors_b;
    DevBuf<uint32_t> d_counts;  // [n_asm] anchor counts, [KP_N_CLASSES] task counts, [n_asm] largest sub-slice demand
    DevBuf<uint32_t> d_sub_counts;  // [n_asm * KP_ANCHOR_SUBS]
    DevBuf<uint64_t> d_cand;        // candidates of the scan (kp_cand_pack); d_cand_count[0] = how many
    DevBuf<unsigned long long> d_cand_count;
    DevBuf<uint32_t> d_seg;     // [2 * n_asm]
    DevBuf<KpTask> d_tasks;
    DevBuf<KpSwResult> d_results;
    DevBuf<uint8_t> d_task_drop, d_jscratch;
    // counting tables of the occurrence cut's quantile (kp_chain.hip: block_mid_occ): occ_slots tables of 2^occ_log2 entries
    DevBuf<uint32_t> d_occ_keys, d_occ_cnts, d_occ_state;
    uint32_t occ_slots = 0, occ_log2 = 0;  // per task slot: a chain consumed the cluster, its band task reports no hit (kp_join.hip)
    DevBuf<KpSwEnd> d_ends;
    DevBuf<unsigned long long> d_trace_top;
    uint64_t trace_cap = 0;  // 16-byte units the trace buffer was sized for in the most recent pass
    // A work set's alignment pass runs on the set's own stream with its own trace buffer and sort scratch: the passes of
    // consecutive batches overlap on the device (the seed scan and the sort of one wait on the L2 and on HBM while the
    // fill kernel of the other keeps the vector ALUs busy)
    hipStream_t astream = nullptr;
    // ... and the joined fill of its joins on a second one, beside the band tasks' fill (kp_join.hip: a few waves, each a long
    // chain of dependent steps: 3 ms that the pass would otherwise wait for)
    hipStream_t jstream = nullptr;
    hipEvent_t ev_jfork = nullptr, ev_jdone = nullptr;
    DevBuf<uint4> d_trace;  // direction bits of the banded Smith-Waterman: written by the fill kernel, read by the traceback
    void *sort_temp = nullptr;
    size_t sort_temp_bytes = 0;
    DevBuf<uint32_t> d_task_order;  // [ORDER_HEAD] histogram + cursors, then [KP_N_CLASSES * task_cap] permutation
    // kp-align v4 (kp_join.hip): groups of provisional clusters, joins per band class, their counts ([0] groups, [1 + c] joins)
    DevBuf<KpGroup> d_groups;
    DevBuf<KpJoin> d_joins;
    DevBuf<uint32_t> d_join_counts;
    uint32_t group_cap = 0, join_cap = 0;
    uint32_t h_join_counts[1 + KP_N_CLASSES] = {};
    std::vector<KpJoin> h_joins;  // fetched on first use (kp_batch_joins: stage tests only)
    // device-side hit tables (per-assembly regions of hit_cap rows)
    DevBuf<kp_hit> d_hits_raw, d_hits;
    DevBuf<uint32_t> d_hit_counts;  // [n_asm] raw, then [n_asm] final
    DevBuf<uint64_t> d_keys;        // 3 per hit row
    DevBuf<unsigned long long> d_cells;
    // reduction: one run per typing group, created on first use
    std::vector<std::unique_ptr<KpTypingRun>> runs;
    // results
    bool aligned = false, finalised = false;
    std::vector<uint32_t> h_counts, h_hit_counts;
    std::vector<KpTask> h_tasks[KP_N_CLASSES];
    std::vector<int64_t> hit_off;
    int64_t stats[5] = {0, 0, 0, 0, 0};
    hipEvent_t ev[4 + KP_N_CLASSES] = {};  // stage boundaries of the most recent alignment pass; the last one marks its end
    bool have_events = false;
    void release() {
        d_anchors_a.release(); d_anchors_b.release(); d_counts.release(); d_sub_counts.release(); d_cand.release();
        d_cand_count.release(); d_seg.release(); d_tasks.release(); d_results.release(); d_task_drop.release(); d_jscratch.release(); d_occ_keys.release(); d_occ_cnts.release(); d_occ_state.release(); d_task_order.release();
        d_ends.release(); d_trace_top.release(); d_trace.release();
        d_groups.release(); d_joins.release(); d_join_counts.release();
        if (sort_temp) { (void)hipFree(sort_temp); sort_temp = nullptr; sort_temp_bytes = 0; }
        if (astream) { (void)hipStreamSynchronize(astream); (void)hipStreamDestroy(astream); astream = nullptr; }
        if (jstream) { (void)hipStreamSynchronize(jstream); (void)hipStreamDestroy(jstream); jstream = nullptr; }
        if (ev_jfork) { (void)hipEventDestroy(ev_jfork); ev_jfork = nullptr; }
        if (ev_jdone) { (void)hipEventDestroy(ev_jdone); ev_jdone = nullptr; }
        d_hits_raw.release(); d_hits.release(); d_hit_counts.release(); d_keys.release(); d_cells.release();
        for (auto &r : runs)
            if (r) r->release();
        runs.clear();
        if (have_events)
            for (auto &e : ev) (void)hipEventDestroy(e);
        have_events = false;
    }
};

#define KP_INPUT_POOL 16  /* recycled device copies of batch inputs: uploads run several shards ahead of the passes that read them */

struct kp_ctx {
    int device = 0;
    uint8_t *bounce = nullptr;     // page-locked landing area of result read-backs (Fetch)
    size_t bounce_bytes = 0;
    int gs_bits = 18;              // bits of the gene/strand field of an anchor key this database can set
    int max_gene_len = 0;
    hipStream_t stream = nullptr;  // database uploads, stand-alone protein alignments (alignment passes: KpWork::astream)
    hipStream_t post = nullptr;    // everything after a batch's alignment pass (waits on that batch's event)
    hipStream_t aux = nullptr;     // forked off `post` for kernels that only fill a few CUs (wide-band proteins)
    hipStream_t copy = nullptr;    // H2D copies of batch inputs (overlap with the passes of earlier batches)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    std::string error;
    KpOptions opt;
    // learnt buffer sizes (0 = not yet sized: first use takes the option's value); they only grow
    uint32_t anchor_cap = 0, hit_cap = 0;
    uint32_t tasks_per_asm = 0;  // task_cap of a pass = n_asm * tasks_per_asm
    double cand_frac = 0.0;      // cand_cap of a pass = total selected positions * cand_frac
    int64_t words_hw = 0;        // most packed words any batch of this context held: candidate lists are sized for that, so a
                                 // work set that meets a slightly larger batch than before does not re-allocate (and stall)
    uint64_t trace_units_per_asm = 0;  // trace buffer of a pass = n_asm * this many 16-byte units
    uint32_t group_cap = 0, join_cap = 0;  // group / join lists of a pass (entries; joins per band class)
    uint32_t occ_slots = 2;                // counting tables of the occurrence cut's quantile a pass may use (learnt like the list sizes)
    // resident database
    bool has_db = false;
    int32_t n_genes = 0;
    int64_t n_postings = 0;
    std::vector<int32_t> gene_len;  // host copy (finalisation flips reverse-strand coordinates)
    DevBuf<uint2> d_slots;
    DevBuf<uint64_t> d_filter, d_filter2;
    DevBuf<uint64_t> d_postings;
    DevBuf<uint32_t> d_nib;
    DevBuf<int32_t> d_nib_off, d_gene_len;
    DevBuf<uint4> d_gene_prof;   // row profiles of the genes for the fill kernel (KpGenes::prof)
    DevBuf<uint8_t> d_gene_has_n;
    KpSeedIndex index{};
    KpGenes genes{};
    // protein stage
    DevBuf<int8_t> d_blosum;
    DevBuf<float> d_ln;  // logarithm tables of the mapping quality (kp_mapq.h): ln(i / 2), then ln(i), from kp_mapq_ln
    DevBuf<uint8_t> d_pq, d_pt;
    DevBuf<int32_t> d_pmeta, d_pout, d_pscratch;
    // typing tables (kp_db_load_typing / kp_db_load_typing_group): one set per database whose genes are in the index
    std::vector<std::unique_ptr<KpTypingGroup>> groups;
    // per typing group: learnt sizes of the reduction buffers
    struct RunCaps { int kept_cap = 0, piece_cap = 0, prot_cap = 0; size_t pack_items = 0; /* most words kp_batch_typing packed */ };
    std::vector<RunCaps> run_caps;
    struct GroupStreams { hipStream_t stream = nullptr, aux = nullptr; };
    std::vector<GroupStreams> group_streams;  // reduction streams, per typing group
    // work sets and recycled inputs
    KpWork work[KP_WORK_SLOTS];
    uint32_t next_slot = 0;
    std::vector<KpInput *> free_inputs;
    std::vector<kp_batch *> batches;  // live batches (a context destroyed first detaches them)
};

struct kp_batch {
    kp_ctx *ctx = nullptr;
    int32_t n_asm = 0;
    KpInput *in = nullptr;       // device copy of the input (returned to the context's pool by kp_batch_destroy)
    const uint32_t *d_words = nullptr;  // in->d_words.p, or the caller's device pointer (kp_batch_create_device)
    KpBatchView view{};
    int64_t max_asm_bases = 0;  // longest assembly of the batch (padded)
    int32_t n_ctg_total = 0;    // contigs of all its assemblies (one thread each in the edge kernel)
    KpWork *w = nullptr;        // the work set holding this batch's alignment results, while it still does
    KpWork *last_w = nullptr;   // the work set of its most recent pass (for completion waits; may have a new owner)
    kp_batch *after = nullptr;  // its words are another batch's device copy: passes wait for that batch's upload
    int32_t group = 0;          // the typing group score / reduce / typing calls address
};

int kp_fail(kp_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->error = msg;
    else {
        std::lock_guard<std::mutex> lk(g_err_mutex);
        g_global_error = msg;
    }
    return code;
}

namespace {
// measurement aid (KAPTIVE_AMD_DUMMY_LAUNCHES=n: n empty launches per alignment pass): what a kernel boundary costs the kernels
// of the other passes in flight -- every launch begins and ends with cache maintenance on the L2s
__global__ void kp_noop_kernel() {}

// Streams of the short, low-occupancy kernels that follow an alignment pass get the highest priority the device offers:
// when another batch's alignment pass fills the chip, their waves are scheduled as soon as any slot frees up.
hipError_t create_priority_stream(hipStream_t *stream) {
    int least = 0, greatest = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) return e;
    return hipStreamCreateWithPriority(stream, hipStreamDefault, greatest);
}

uint32_t env_u32(const char *name, uint32_t dflt) {
    const char *v = std::getenv(name);
    if (!v || !*v) return dflt;
    const long long x = std::atoll(v);
    return x > 0 ? (uint32_t)x : dflt;
}

// the environment is read here, once per context, and nowhere else
void options_from_env(KpOptions &o) {
    o.anchor_cap = env_u32("KAPTIVE_AMD_ANCHOR_CAP", o.anchor_cap);
    o.tasks_per_asm = env_u32("KAPTIVE_AMD_TASKS_PER_ASM", o.tasks_per_asm);
    o.hit_cap = env_u32("KAPTIVE_AMD_HIT_CAP", o.hit_cap);
    o.trace_kb_per_asm = env_u32("KAPTIVE_AMD_TRACE_KB_PER_ASM", o.trace_kb_per_asm);
    o.kept_cap = env_u32("KAPTIVE_AMD_KEPT_CAP", o.kept_cap);
    o.piece_cap = env_u32("KAPTIVE_AMD_PIECE_CAP", o.piece_cap);
    o.prot_cap = env_u32("KAPTIVE_AMD_PROT_CAP", o.prot_cap);
    o.scan_mode = (int)env_u32("KAPTIVE_AMD_SCAN_ABLATE", 0);
    o.library_sort = (int)env_u32("KAPTIVE_AMD_LIBRARY_SORT", 0);
    o.upload_piece_mb = std::max<uint32_t>(1, env_u32("KAPTIVE_AMD_UPLOAD_PIECE_MB", 4096));
    { const char *rb = getenv("KAPTIVE_AMD_READBACK"); o.readback_copy_engine = rb && std::string(rb) == "copy"; }
    o.spin_wait = (int)env_u32("KAPTIVE_AMD_SPIN_WAIT", 0);
}

// BLOSUM62 as the reference lays it out: 256x256 bytes, -128 outside ARNDCQEGHILKMFPSTWYVBJZX*
// (src/kaptive/core/pairwise.py:343-391)
void fill_blosum(int8_t *m) {
    static const int8_t b[25][25] = {
        {4, -1, -2, -2, 0, -1, -1, 0, -2, -1, -1, -1, -1, -2, -1, 1, 0, -3, -2, 0, -2, -1, -1, -1, -4},
        {-1, 5, 0, -2, -3, 1, 0, -2, 0, -3, -2, 2, -1, -3, -2, -1, -1, -3, -2, -3, -1, -2, 0, -1, -4},
        {-2, 0, 6, 1, -3, 0, 0, 0, 1, -3, -3, 0, -2, -3, -2, 1, 0, -4, -2, -3, 4, -3, 0, -1, -4},
        {-2, -2, 1, 6, -3, 0, 2, -1, -1, -3, -4, -1, -3, -3, -1, 0, -1, -4, -3, -3, 4, -3, 1, -1, -4},
        {0, -3, -3, -3, 9, -3, -4, -3, -3, -1, -1, -3, -1, -2, -3, -1, -1, -2, -2, -1, -3, -1, -3, -1, -4},
        {-1, 1, 0, 0, -3, 5, 2, -2, 0, -3, -2, 1, 0, -3, -1, 0, -1, -2, -1, -2, 0, -2, 4, -1, -4},
        {-1, 0, 0, 2, -4, 2, 5, -2, 0, -3, -3, 1, -2, -3, -1, 0, -1, -3, -2, -2, 1, -3, 4, -1, -4},
        {0, -2, 0, -1, -3, -2, -2, 6, -2, -4, -4, -2, -3, -3, -2, 0, -2, -2, -3, -3, -1, -4, -2, -1, -4},
        {-2, 0, 1, -1, -3, 0, 0, -2, 8, -3, -3, -1, -2, -1, -2, -1, -2, -2, 2, -3, 0, -3, 0, -1, -4},
        {-1, -3, -3, -3, -1, -3, -3, -4, -3, 4, 2, -3, 1, 0, -3, -2, -1, -3, -1, 3, -3, 3, -3, -1, -4},
        {-1, -2, -3, -4, -1, -2, -3, -4, -3, 2, 4, -2, 2, 0, -3, -2, -1, -2, -1, 1, -4, 3, -3, -1, -4},
        {-1, 2, 0, -1, -3, 1, 1, -2, -1, -3, -2, 5, -1, -3, -1, 0, -1, -3, -2, -2, 0, -3, 1, -1, -4},
        {-1, -1, -2, -3, -1, 0, -2, -3, -2, 1, 2, -1, 5, 0, -2, -1, -1, -1, -1, 1, -3, 2, -1, -1, -4},
        {-2, -3, -3, -3, -2, -3, -3, -3, -1, 0, 0, -3, 0, 6, -4, -2, -2, 1, 3, -1, -3, 0, -3, -1, -4},
        {-1, -2, -2, -1, -3, -1, -1, -2, -2, -3, -3, -1, -2, -4, 7, -1, -1, -4, -3, -2, -2, -3, -1, -1, -4},
        {1, -1, 1, 0, -1, 0, 0, 0, -1, -2, -2, 0, -1, -2, -1, 4, 1, -3, -2, -2, 0, -2, 0, -1, -4},
        {0, -1, 0, -1, -1, -1, -1, -2, -2, -1, -1, -1, -1, -2, -1, 1, 5, -2, -2, 0, -1, -1, -1, -1, -4},
        {-3, -3, -4, -4, -2, -2, -3, -2, -2, -3, -2, -3, -1, 1, -4, -3, -2, 11, 2, -3, -4, -2, -2, -1, -4},
        {-2, -2, -2, -3, -2, -1, -2, -3, 2, -1, -1, -2, -1, 3, -3, -2, -2, 2, 7, -1, -3, -1, -2, -1, -4},
        {0, -3, -3, -3, -1, -2, -2, -3, -3, 3, 1, -2, 1, -1, -2, -2, 0, -3, -1, 4, -3, 2, -2, -1, -4},
        {-2, -1, 4, 4, -3, 0, 1, -1, 0, -3, -4, 0, -3, -3, -2, 0, -1, -4, -3, -3, 4, -3, 0, -1, -4},
        {-1, -2, -3, -3, -1, -2, -3, -4, -3, 3, 3, -3, 2, 0, -3, -2, -1, -2, -1, 2, -3, 3, -3, -1, -4},
        {-1, 0, 0, 1, -3, 4, 4, -2, 0, -3, -3, 1, -1, -3, -1, 0, -1, -2, -2, -2, 0, -3, 4, -1, -4},
        {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -4},
        {-4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, -4, 1},
    };
    static const char alphabet[] = "ARNDCQEGHILKMFPSTWYVBJZX*";
    std::memset(m, KP_PROT_FILL, 256 * 256);
    for (int x = 0; x < 25; ++x)
        for (int y = 0; y < 25; ++y) m[(uint8_t)alphabet[x] * 256 + (uint8_t)alphabet[y]] = b[x][y];
}

struct HostPosting { uint32_t key, gene, pos, z; };  // a gene seed: x, gene, first base on the forward strand, strand bit

// Results come back to the host through the shader, not through a copy engine: kernels of the stream write them into a
// page-locked landing area, the host waits for the stream and copies them out.  A read-back handed to the copy engines
// (hipMemcpyAsync) queued up behind the shard that was being uploaded -- 1.25 GB, 22 ms -- so that every kp_batch_score
// of a host-fed stream returned only when the upload in flight had landed (tools/experiments/h2d_interference.py: 20 ms
// per batch alone, 140 ms beside a continuous upload; profiles/r4_h2d_timeline.md).  KAPTIVE_AMD_READBACK=copy restores
// the copy-engine path for comparison.
struct Fetch {
    kp_ctx *ctx;
    hipStream_t stream;
    struct Item { void *dst; size_t off, bytes; };
    std::vector<Item> items;
    size_t used = 0;
    bool by_copy_engine;
    Fetch(kp_ctx *c, hipStream_t s) : ctx(c), stream(s), by_copy_engine(c->opt.readback_copy_engine != 0) {}
    // A Fetch that is dropped half-way (an error between add() and finish()) leaves kernels writing into the landing
    // area: wait for them, so that the next begin() never frees or reuses memory that is still being written.
    ~Fetch() { if (!items.empty()) (void)hipStreamSynchronize(stream); }
    // total bytes of everything that will be added before finish()
    int begin(size_t total) {
        if (by_copy_engine) return KP_OK;
        total += 64 * 8;
        if (total > ctx->bounce_bytes) {
            // (calls on a context are serialised and every Fetch waits for its stream before it goes away, so nothing
            // is in flight towards the old area here)
            if (ctx->bounce) pinned_free(ctx->bounce);
            ctx->bounce = nullptr; ctx->bounce_bytes = 0;
            const size_t want = total + total / 4 + (1u << 20);
            KP_HIP_CHECK(ctx, pinned_alloc((void **)&ctx->bounce, want));
            ctx->bounce_bytes = want;
        }
        return KP_OK;
    }
    int add(void *dst, const void *src, size_t bytes) {
        if (!bytes) return KP_OK;
        if (by_copy_engine || (bytes & 3u)) { KP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, stream)); return KP_OK; }
        if (used + bytes > ctx->bounce_bytes) return kp_fail(ctx, KP_ESTATE, "read-back larger than announced");
        kp_launch_read_back(src, ctx->bounce + used, bytes, stream);
        items.push_back(Item{dst, used, bytes});
        used = (used + bytes + 63) & ~(size_t)63;
        return KP_OK;
    }
    int finish() {
        const hipError_t e = hipStreamSynchronize(stream);
        if (e != hipSuccess) { items.clear(); used = 0; }
        KP_HIP_CHECK(ctx, e);
        for (const Item &it : items) std::memcpy(it.dst, ctx->bounce + it.off, it.bytes);
        items.clear(); used = 0;
        return KP_OK;
    }
};

template <class T>
int upload(kp_ctx *ctx, DevBuf<T> &buf, const T *src, size_t n, hipStream_t stream = nullptr) {
    KP_HIP_CHECK(ctx, buf.reserve(n));
    if (n) KP_HIP_CHECK(ctx, hipMemcpyAsync(buf.p, src, n * sizeof(T), hipMemcpyHostToDevice, stream ? stream : ctx->stream));
    return KP_OK;
}

// ---- batch inputs ---------------------------------------------------------------------------------------------------------
KpInput *acquire_input(kp_ctx *ctx) {
    if (!ctx->free_inputs.empty()) {
        KpInput *in = ctx->free_inputs.back();
        ctx->free_inputs.pop_back();
        return in;
    }
    KpInput *in = new (std::nothrow) KpInput();
    if (in && hipEventCreateWithFlags(&in->ready, hipEventDisableTiming) != hipSuccess) { delete in; in = nullptr; }
    return in;
}

void recycle_input(kp_ctx *ctx, KpInput *in) {
    if (!in) return;
    if (ctx && ctx->free_inputs.size() < KP_INPUT_POOL) { ctx->free_inputs.push_back(in); return; }
    in->release();
    delete in;
}

// Validates the tables, stages them in the input's pinned buffer and enqueues their H2D copies on the copy stream (the
// caller's arrays may be freed as soon as this returns).
int batch_tables(kp_ctx *ctx, kp_batch *b, int32_t n_asm, const int64_t *asm_word_off, const int32_t *ctg_start,
                 const int32_t *ctg_len, const int32_t *asm_first_ctg, const int32_t *n_runs,
                 const int32_t *asm_first_nrun) {
    b->max_asm_bases = 0;
    for (int a = 0; a < n_asm; ++a) {
        const int64_t words = asm_word_off[a + 1] - asm_word_off[a];
        b->max_asm_bases = std::max<int64_t>(b->max_asm_bases, words * 16);
        if (words < 0 || (words * 16) % KP_ASM_ALIGN != 0 || (uint64_t)words * 16 > KP_MAX_ASM_LEN)
            return kp_fail(ctx, KP_EINVAL, "assembly length must be a multiple of KP_ASM_ALIGN and <= KP_MAX_ASM_LEN");
        if (asm_first_ctg[a + 1] < asm_first_ctg[a] || asm_first_nrun[a + 1] < asm_first_nrun[a])
            return kp_fail(ctx, KP_EINVAL, "offset tables must be non-decreasing");
        for (int c = asm_first_ctg[a]; c < asm_first_ctg[a + 1]; ++c) {
            if (ctg_start[c] % (int)KP_CONTIG_ALIGN != 0 || ctg_len[c] < 0 ||
                (int64_t)ctg_start[c] + ctg_len[c] > words * 16 ||
                (c > asm_first_ctg[a] && ctg_start[c] < ctg_start[c - 1] + ctg_len[c - 1]))
                return kp_fail(ctx, KP_EINVAL, "contig table violates the packed layout (kp_spec.h)");
        }
    }
    KpInput &in = *b->in;
    const size_t n_ctg = (size_t)asm_first_ctg[n_asm], n_run = (size_t)asm_first_nrun[n_asm];
    const size_t na1 = (size_t)n_asm + 1;
    // staging layout: asm_word_off (8-byte entries first), then the int32 tables
    const size_t bytes = na1 * 8 + (2 * n_ctg + 2 * na1 + 2 * n_run) * 4;
    if (bytes > in.h_stage_bytes) {
        pinned_free(in.h_stage);
        in.h_stage = nullptr; in.h_stage_bytes = 0;
        const size_t want = bytes + bytes / 4 + 4096;
        KP_HIP_CHECK(ctx, pinned_alloc((void **)&in.h_stage, want));
        in.h_stage_bytes = want;
    }
    uint8_t *p = in.h_stage;
    auto stage = [&](auto &buf, const auto *src, size_t n) -> int {
        using T = std::remove_cv_t<std::remove_pointer_t<decltype(src)>>;
        if (n) std::memcpy(p, src, n * sizeof(T));
        const int rc = upload(ctx, buf, reinterpret_cast<const T *>(p), n, ctx->copy);
        p += n * sizeof(T);
        return rc;
    };
    int rc;
    if ((rc = stage(in.d_asm_word_off, asm_word_off, na1))) return rc;
    if ((rc = stage(in.d_ctg_start, ctg_start, n_ctg))) return rc;
    if ((rc = stage(in.d_ctg_len, ctg_len, n_ctg))) return rc;
    if ((rc = stage(in.d_asm_first_ctg, asm_first_ctg, na1))) return rc;
    if ((rc = stage(in.d_n_runs, n_runs, 2 * n_run))) return rc;
    if ((rc = stage(in.d_asm_first_nrun, asm_first_nrun, na1))) return rc;
    b->view.words = b->d_words;
    b->view.asm_word_off = in.d_asm_word_off.p;
    b->view.ctg_start = in.d_ctg_start.p;
    b->view.ctg_len = in.d_ctg_len.p;
    b->view.asm_first_ctg = in.d_asm_first_ctg.p;
    b->view.n_runs = in.d_n_runs.p;
    b->view.asm_first_nrun = in.d_asm_first_nrun.p;
    b->view.n_asm = n_asm;
    b->view.total_words = asm_word_off[n_asm];
    b->n_ctg_total = (int32_t)n_ctg;
    KP_HIP_CHECK(ctx, hipEventRecord(in.ready, ctx->copy));
    return KP_OK;
}

void detach_batch(kp_ctx *ctx, kp_batch *b) {
    auto &v = ctx->batches;
    v.erase(std::remove(v.begin(), v.end(), b), v.end());
}

// everything this batch enqueued has finished (its pass, its hit finalisation, its reductions)
void quiesce_batch(kp_batch *b) {
    KpWork *w = b->last_w;
    if (!w) return;
    if (w->have_events) (void)hipEventSynchronize(w->ev[3 + KP_N_CLASSES]);  // end of the slot's most recent pass
    if (b->ctx) (void)hipStreamSynchronize(b->ctx->post);
    for (auto &r : w->runs)
        if (r && r->stream) { (void)hipStreamSynchronize(r->stream); if (r->aux) (void)hipStreamSynchronize(r->aux); }
}

int batch_make(kp_ctx *ctx, int32_t n_asm, const uint32_t *words, bool words_on_device, const int64_t *asm_word_off,
               const int32_t *ctg_start, const int32_t *ctg_len, const int32_t *asm_first_ctg, const int32_t *n_runs,
               const int32_t *asm_first_nrun, kp_batch **out) {
    if (!ctx) return kp_fail(nullptr, KP_EINVAL, "null context");
    if (!out || n_asm < 0 || !asm_word_off) return kp_fail(ctx, KP_EINVAL, "bad batch arguments");
    *out = nullptr;
    if (asm_word_off[0] != 0) return kp_fail(ctx, KP_EINVAL, "asm_word_off[0] must be 0");
    if (!asm_first_ctg || !asm_first_nrun) return kp_fail(ctx, KP_EINVAL, "null offset table");
    if (asm_word_off[n_asm] > 0 && !words) return kp_fail(ctx, KP_EINVAL, "null words");
    if (words_on_device && ((uintptr_t)words & 15u) != 0) return kp_fail(ctx, KP_EINVAL, "device words must be 16-byte aligned");
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    kp_batch *b = new (std::nothrow) kp_batch();
    if (!b) return kp_fail(ctx, KP_ENOMEM, "out of host memory");
    b->ctx = ctx; b->n_asm = n_asm;
    b->in = acquire_input(ctx);
    if (!b->in) { delete b; return kp_fail(ctx, KP_ENOMEM, "out of memory (batch input)"); }
    ctx->batches.push_back(b);
    if (words_on_device) {
        b->d_words = words;
    } else {
        const size_t nw = (size_t)asm_word_off[n_asm];
        hipError_t e = b->in->d_words.reserve(std::max<size_t>(nw, 4));
        if (e != hipSuccess) { kp_batch_destroy(b); return kp_fail(ctx, KP_ENOMEM, std::string("hipMalloc(words): ") + hipGetErrorString(e)); }
        // (optionally in pieces -- `upload_piece_mb` --: tried so that result read-backs could slip in between them on the
        // copy engines; they did not, see Fetch)
        const size_t piece = (size_t)ctx->opt.upload_piece_mb << 18;  // words
        for (size_t at = 0; at < nw && e == hipSuccess; at += piece)
            e = hipMemcpyAsync(b->in->d_words.p + at, words + at, std::min(piece, nw - at) * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->copy);
        if (e != hipSuccess) { kp_batch_destroy(b); return kp_fail(ctx, KP_EHIP, std::string("H2D words: ") + hipGetErrorString(e)); }
        b->d_words = b->in->d_words.p;
    }
    const int rc = batch_tables(ctx, b, n_asm, asm_word_off, ctg_start, ctg_len, asm_first_ctg, n_runs, asm_first_nrun);
    if (rc) { const std::string msg = ctx->error; kp_batch_destroy(b); ctx->error = msg; return rc; }
    *out = b;
    return KP_OK;
}

KpWork *work_of(kp_ctx *ctx, kp_batch *b) {  // null: never aligned, or its results have been displaced
    (void)ctx;
    return (b->w && b->w->owner == b) ? b->w : nullptr;
}

const char *const NO_RESULTS = "this batch has no resident alignment results (not aligned yet, or displaced: a context keeps "
                               "the results of its KP_WORK_SLOTS most recently aligned batches)";

}  // namespace

extern "C" {

int kp_device_count(void) {
    int n_dev = 0;
    const hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess) return kp_fail(nullptr, KP_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n_dev;
}

int kp_device_numa_node(int device_id) {
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device_id) != hipSuccess) return -1;
    for (char *c = bus; *c; ++c) *c = (char)std::tolower((unsigned char)*c);  // sysfs names are lower case
    const std::string path = std::string("/sys/bus/pci/devices/") + bus + "/numa_node";
    std::FILE *f = std::fopen(path.c_str(), "r");
    if (!f) return -1;
    int node = -1;
    if (std::fscanf(f, "%d", &node) != 1) node = -1;
    std::fclose(f);
    return node;
}

int kp_ctx_create(int device_id, kp_ctx **out) {
    if (!out) return kp_fail(nullptr, KP_EINVAL, "out is null");
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev == 0)
        return kp_fail(nullptr, KP_EHIP, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= n_dev) return kp_fail(nullptr, KP_EINVAL, "device_id out of range");
    kp_ctx *ctx = new (std::nothrow) kp_ctx();
    if (!ctx) return kp_fail(nullptr, KP_ENOMEM, "out of host memory");
    ctx->device = device_id;
    options_from_env(ctx->opt);
    // The driving thread spends most of its time waiting for the device (scores, records): blocked on an interrupt it
    // leaves its core to the readers that feed the next shard (eight ranks and their ingest share one host); the
    // runtime's default spins.  (A device-wide flag: it has to be set before the device's streams exist.)
    if ((e = hipSetDevice(device_id)) == hipSuccess && !ctx->opt.spin_wait) (void)hipSetDeviceFlags(hipDeviceScheduleBlockingSync);
    if (e != hipSuccess || (e = hipStreamCreate(&ctx->stream)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->copy, hipStreamNonBlocking)) != hipSuccess ||
        (e = create_priority_stream(&ctx->post)) != hipSuccess || (e = create_priority_stream(&ctx->aux)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming)) != hipSuccess) {
        delete ctx;
        return kp_fail(nullptr, KP_EHIP, std::string("device setup failed: ") + hipGetErrorString(e));
    }
    std::vector<int8_t> m(256 * 256);
    fill_blosum(m.data());
    std::vector<float> ln(KP_MAPQ_LN_HALF_SIZE + KP_MAPQ_LN_INT_SIZE, 0.0f);
    for (int i = 1; i < KP_MAPQ_LN_HALF_SIZE; ++i) ln[(size_t)i] = kp_mapq_ln((double)i / 2.0);
    for (int i = 1; i < KP_MAPQ_LN_INT_SIZE; ++i) ln[(size_t)KP_MAPQ_LN_HALF_SIZE + i] = kp_mapq_ln((double)i);
    if (upload(ctx, ctx->d_blosum, m.data(), m.size()) != KP_OK || upload(ctx, ctx->d_ln, ln.data(), ln.size()) != KP_OK ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        std::string msg = ctx->error;
        kp_ctx_destroy(ctx);
        return kp_fail(nullptr, KP_EHIP, "substitution table upload failed: " + msg);
    }
    *out = ctx;
    return KP_OK;
}

void kp_ctx_destroy(kp_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (kp_batch *b : ctx->batches) {  // batches that outlive their context keep nothing on the device
        if (b->in) { b->in->release(); delete b->in; b->in = nullptr; }
        b->ctx = nullptr; b->w = nullptr; b->last_w = nullptr;
    }
    ctx->batches.clear();
    for (KpInput *in : ctx->free_inputs) { in->release(); delete in; }
    ctx->free_inputs.clear();
    for (auto &w : ctx->work) w.release();
    if (ctx->bounce) { pinned_free(ctx->bounce); ctx->bounce = nullptr; ctx->bounce_bytes = 0; }
    ctx->d_slots.release(); ctx->d_filter.release(); ctx->d_filter2.release(); ctx->d_postings.release(); ctx->d_nib.release(); ctx->d_nib_off.release(); ctx->d_gene_prof.release(); ctx->d_gene_has_n.release();
    ctx->d_gene_len.release(); ctx->d_blosum.release(); ctx->d_pq.release(); ctx->d_pt.release();
    ctx->d_pmeta.release(); ctx->d_pout.release(); ctx->d_pscratch.release(); ctx->d_ln.release();
    for (auto &g : ctx->groups)
        if (g) g->release();
    ctx->groups.clear();
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->copy) (void)hipStreamDestroy(ctx->copy);
    for (auto &gs : ctx->group_streams) {
        if (gs.stream) (void)hipStreamDestroy(gs.stream);
        if (gs.aux) (void)hipStreamDestroy(gs.aux);
    }
    if (ctx->post) (void)hipStreamDestroy(ctx->post);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    delete ctx;
}

const char *kp_last_error(const kp_ctx *ctx) {
    if (ctx) return ctx->error.c_str();
    std::lock_guard<std::mutex> lk(g_err_mutex);
    static thread_local std::string copy;
    copy = g_global_error;
    return copy.c_str();
}

void *kp_ctx_stream(kp_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int kp_ctx_set_option(kp_ctx *ctx, const char *name, int64_t value) {
    if (!ctx) return kp_fail(nullptr, KP_EINVAL, "null context");
    if (!name || value < 0) return kp_fail(ctx, KP_EINVAL, "bad option");
    const std::string n(name);
    KpOptions &o = ctx->opt;
    // buffer-size options also reset what the context has learnt, so the next pass starts from the new value
    if (n == "anchor_cap") { o.anchor_cap = (uint32_t)std::max<int64_t>(value, 1); ctx->anchor_cap = 0; }
    else if (n == "tasks_per_asm") { o.tasks_per_asm = (uint32_t)std::max<int64_t>(value, 1); ctx->tasks_per_asm = 0; }
    else if (n == "hit_cap") { o.hit_cap = (uint32_t)std::max<int64_t>(value, 1); ctx->hit_cap = 0; }
    else if (n == "trace_kb_per_asm") { o.trace_kb_per_asm = (uint32_t)std::max<int64_t>(value, 1); ctx->trace_units_per_asm = 0; }
    else if (n == "kept_cap") { o.kept_cap = (uint32_t)std::max<int64_t>(value, 1); for (auto &c : ctx->run_caps) c.kept_cap = 0; }
    else if (n == "piece_cap") { o.piece_cap = (uint32_t)std::max<int64_t>(value, 1); for (auto &c : ctx->run_caps) c.piece_cap = 0; }
    else if (n == "prot_cap") { o.prot_cap = (uint32_t)std::max<int64_t>(value, 1); for (auto &c : ctx->run_caps) c.prot_cap = 0; }
    else if (n == "scan_mode") o.scan_mode = (int)value;
    else if (n == "library_sort") o.library_sort = value != 0;
    else if (n == "upload_piece_mb") o.upload_piece_mb = (uint32_t)std::max<int64_t>(1, std::min<int64_t>(value, 4096));
    else return kp_fail(ctx, KP_EINVAL, "unknown option: " + n);
    return KP_OK;
}

int kp_host_alloc(size_t bytes, void **out) {
    if (!out) return kp_fail(nullptr, KP_EINVAL, "out is null");
    *out = nullptr;
    if (bytes >= 4 * HUGE_PAGE) {
        int rc = kp_host_reserve(bytes, out);
        if (rc == KP_OK && (rc = kp_host_lock(*out)) == KP_OK) return KP_OK;
        if (*out) { pinned_free(*out); *out = nullptr; }  // (a host that will not register mapped memory: the runtime's own allocation)
    }
    const hipError_t e = pinned_alloc(out, std::max<size_t>(bytes, 1));
    if (e != hipSuccess) return kp_fail(nullptr, KP_ENOMEM, std::string("hipHostMalloc: ") + hipGetErrorString(e));
    return KP_OK;
}

int kp_host_reserve(size_t bytes, void **out) {
    if (!out) return kp_fail(nullptr, KP_EINVAL, "out is null");
    *out = mapped_alloc(std::max<size_t>(bytes, 1));
    if (!*out) return kp_fail(nullptr, KP_ENOMEM, "mmap failed");
    return KP_OK;
}

int kp_host_lock(void *p) {
    const hipError_t e = mapped_lock(p);
    if (e == hipErrorInvalidValue) return kp_fail(nullptr, KP_EINVAL, "not a block of kp_host_reserve");
    if (e != hipSuccess) return kp_fail(nullptr, KP_ENOMEM, std::string("hipHostRegister: ") + hipGetErrorString(e));
    return KP_OK;
}

void kp_host_free(void *p) { pinned_free(p); }

int64_t kp_device_allocations(void) { return (int64_t)g_dev_allocs.load(std::memory_order_relaxed); }

int64_t kp_host_pinned_bytes(void) {
    std::lock_guard<std::mutex> lk(g_pin_mutex);
    return (int64_t)g_pin_bytes;
}

int kp_db_load(kp_ctx *ctx, const uint8_t *gene_codes, const int32_t *gene_off, int32_t n_genes) {
    if (!ctx) return kp_fail(nullptr, KP_EINVAL, "null context");
    if (!gene_off || n_genes < 0 || (n_genes > 0 && !gene_codes)) return kp_fail(ctx, KP_EINVAL, "bad database arguments");
    if (n_genes > KP_MAX_GENES) return kp_fail(ctx, KP_EINVAL, "too many genes (KP_MAX_GENES)");
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ctx->has_db = false;
    for (auto &g : ctx->groups)
        if (g) g->release();
    ctx->groups.clear();  // typing tables index the genes that are being replaced
    ctx->run_caps.clear();
    ctx->gene_len.resize((size_t)n_genes);
    std::vector<int32_t> nib_off(2 * (size_t)n_genes);
    size_t n_words = 0;
    for (int g = 0; g < n_genes; ++g) {
        const int len = gene_off[g + 1] - gene_off[g];
        if (len < 0 || len > KP_MAX_GENE_LEN) return kp_fail(ctx, KP_EINVAL, "gene length outside [0, KP_MAX_GENE_LEN]");
        ctx->gene_len[(size_t)g] = len;
        nib_off[(size_t)g] = (int32_t)n_words;
        n_words += (size_t)(len + 7) / 8;
    }
    for (int g = 0; g < n_genes; ++g) {
        nib_off[(size_t)n_genes + g] = (int32_t)n_words;
        n_words += (size_t)(ctx->gene_len[(size_t)g] + 7) / 8;
    }
    const bool load_timing = std::getenv("KAPTIVE_AMD_LOAD_TIMING") != nullptr;
    const auto t_load0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (load_timing) std::fprintf(stderr, "[kp_db_load] %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_load0).count()); };
    std::vector<uint32_t> nib(std::max<size_t>(n_words, 1), 0x44444444u);
    std::vector<uint16_t> prof(8 * nib.size(), 0);  // eight rows per word of `nib`
    std::vector<uint8_t> has_n(std::max<size_t>((size_t)n_genes, 1), 0);
    // Packing, profiles and sketches gene by gene on a few threads (45 ms of a 120 ms load on one: a command that types one
    // genome spends a fifth of its 0.6 s here); every thread keeps the postings of its own genes, joined in gene order
    std::vector<HostPosting> post;
    {
        const int n_thr = std::max(1, std::min({4, (int)std::thread::hardware_concurrency(), n_genes / 256}));
        std::vector<std::vector<HostPosting>> part((size_t)n_thr);
        std::atomic<bool> failed{false};
        auto work = [&](int t) {
            try {
            const int g_lo = (int)((int64_t)n_genes * t / n_thr), g_hi = (int)((int64_t)n_genes * (t + 1) / n_thr);
            std::vector<uint8_t> rc;
            std::vector<HostPosting> &post = part[(size_t)t];
            for (int g = g_lo; g < g_hi; ++g) {
            const int len = ctx->gene_len[(size_t)g];
            const uint8_t *fwd = gene_codes + gene_off[g];
            rc.resize((size_t)len);
            for (int i = 0; i < len; ++i) {
                const uint8_t c = fwd[len - 1 - i];
                rc[(size_t)i] = c > 3 ? 4 : (uint8_t)(3 - c);
            }
            for (int s = 0; s < 2; ++s) {
                const uint8_t *c = s ? rc.data() : fwd;
                uint32_t *dst = nib.data() + nib_off[(size_t)(s ? n_genes + g : g)];
                for (int i = 0; i < len; ++i) {
                    const uint32_t code = c[i] > 3 ? 4u : c[i];
                    dst[i >> 3] = (dst[i >> 3] & ~(15u << (4 * (i & 7)))) | (code << (4 * (i & 7)));
                    prof[8 * (size_t)(dst - nib.data()) + (size_t)i] = (uint16_t)kp_row_profile(code);
                    if (code > 3u) has_n[(size_t)g] = 1;
                }
            }
            // the gene's seeds: minimap2 sketches a query on its forward strand (kp_spec.h; kp_sketch.h is the state machine)
            KpSketchState st;
            kp_sketch_reset(st);
            auto emit = [&](int64_t start, uint32_t z, uint32_t x) { post.push_back(HostPosting{x, (uint32_t)g, (uint32_t)start, z}); };
            for (int i = 0; i < len; ++i) kp_sketch_step(st, i, fwd[i], emit);
            if (len > 0) kp_sketch_final(st, len - 1, emit);
            }
            } catch (...) { failed.store(true); }  // (out of memory in a worker must not end the process)
        };
        // A thread that cannot be started (std::system_error under a thread limit, bad_alloc) must not let the exception
        // leave this extern "C" function past joinable threads (std::terminate): the calling thread takes the ranges that
        // got no thread of their own
        std::vector<std::thread> pool;
        int n_started = 1;
        try {
            pool.reserve((size_t)n_thr);
            for (; n_started < n_thr; ++n_started) pool.emplace_back(work, n_started);
        } catch (...) {
        }
        work(0);
        for (int t = n_started; t < n_thr; ++t) work(t);
        for (auto &th : pool) th.join();
        if (failed.load()) return kp_fail(ctx, KP_ENOMEM, "out of host memory while sketching the genes");
        size_t total = 0;
        for (const auto &v : part) total += v.size();
        post.reserve(total);
        for (const auto &v : part) post.insert(post.end(), v.begin(), v.end());
    }
    lap("genes packed and sketched");
    {   // by (key, gene, pos): three stable counting passes over the 30-bit key, then the few postings that share a key --
        // they arrive in gene order, positions nearly so -- put right by insertion (a comparison sort took 45 ms of the load)
        static_assert(KP_KMER_MASK < (1u << 30), "three passes of ten bits cover the key");
        std::vector<HostPosting> tmp(post.size());
        for (int pass = 0; pass < 3; ++pass) {
            const int sh = 10 * pass;
            size_t cnt[1025] = {0};
            for (const HostPosting &q : post) ++cnt[((q.key >> sh) & 1023u) + 1];
            for (int b = 0; b < 1024; ++b) cnt[b + 1] += cnt[b];
            for (const HostPosting &q : post) tmp[cnt[(q.key >> sh) & 1023u]++] = q;
            post.swap(tmp);
        }
        auto less = [](const HostPosting &a, const HostPosting &b) { return a.gene != b.gene ? a.gene < b.gene : a.pos < b.pos; };
        for (size_t i = 1; i < post.size(); ++i) {
            if (post[i].key != post[i - 1].key || !less(post[i], post[i - 1])) continue;
            const HostPosting q = post[i];
            size_t j = i;
            for (; j > 0 && post[j - 1].key == q.key && less(q, post[j - 1]); --j) post[j] = post[j - 1];
            post[j] = q;
        }
    }
    lap("postings sorted");
    size_t n_unique = 0;
    for (size_t i = 0; i < post.size(); ++i) n_unique += (i == 0 || post[i].key != post[i - 1].key);
    uint32_t log_slots = 10;
    while (((size_t)1 << log_slots) < 2 * n_unique + 1) ++log_slots;
    if (log_slots > 30) return kp_fail(ctx, KP_EINVAL, "seed index too large");
    const uint32_t n_slots = 1u << log_slots, mask = n_slots - 1, shift = 32 - log_slots;
    std::vector<uint2> slots(n_slots, make_uint2(0xFFFFFFFFu, 0u));
    std::vector<uint64_t> filter((size_t)1 << (KP_FILTER_LOG2 - 6), 0ull), filter2((size_t)1 << (KP_FILTER2_LOG2 - 6), 0ull);
    std::vector<uint64_t> flat;
    flat.reserve(2 * post.size() + n_unique + 1);
    for (size_t i = 0; i < post.size();) {
        size_t j = i;
        while (j < post.size() && post[j].key == post[i].key) ++j;
        uint32_t slot = (post[i].key * 2654435769u) >> shift;
        while (slots[slot].x != 0xFFFFFFFFu) slot = (slot + 1) & mask;
        slots[slot] = make_uint2(post[i].key, (uint32_t)flat.size());
        filter[kp_filter_block(post[i].key)] |= kp_filter_mask(post[i].key);
        {
            const uint2 m2 = kp_filter2_mask2(post[i].key);
            filter2[kp_filter2_block(post[i].key)] |= ((uint64_t)m2.y << 32) | m2.x;
        }
        flat.push_back((uint64_t)(j - i));
        for (uint32_t zt = 0; zt < 2; ++zt)  // the anchors a contig seed with strand bit zt makes with these gene seeds
            for (size_t x = i; x < j; ++x) {
                const uint32_t rev = post[x].z != zt ? 1u : 0u;
                const uint32_t qpos = rev ? (uint32_t)(ctx->gene_len[post[x].gene] - KP_K) - post[x].pos : post[x].pos;
                flat.push_back(((uint64_t)(2u * post[x].gene + rev) << 46) | ((uint64_t)(KP_DIAG_BIAS - qpos) << 16) | qpos);
            }
        i = j;
    }
    if (flat.size() > 0xFFFFFFFFull) return kp_fail(ctx, KP_EINVAL, "seed index too large");
    if (flat.empty()) flat.push_back(0);
    lap("table, filters and anchor lists built");
    int rcode;
    if ((rcode = upload(ctx, ctx->d_slots, slots.data(), slots.size()))) return rcode;
    if ((rcode = upload(ctx, ctx->d_filter, filter.data(), filter.size()))) return rcode;
    if ((rcode = upload(ctx, ctx->d_filter2, filter2.data(), filter2.size()))) return rcode;
    if ((rcode = upload(ctx, ctx->d_postings, flat.data(), flat.size()))) return rcode;
    if ((rcode = upload(ctx, ctx->d_nib, nib.data(), nib.size()))) return rcode;
    if ((rcode = upload(ctx, ctx->d_nib_off, nib_off.data(), nib_off.size()))) return rcode;
    if ((rcode = upload(ctx, ctx->d_gene_prof, reinterpret_cast<const uint4 *>(prof.data()), nib.size()))) return rcode;
    if ((rcode = upload(ctx, ctx->d_gene_has_n, has_n.data(), has_n.size()))) return rcode;
    if ((rcode = upload(ctx, ctx->d_gene_len, ctx->gene_len.data(), ctx->gene_len.size()))) return rcode;
    KP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    lap("uploaded");
    ctx->index = KpSeedIndex{ctx->d_filter.p, ctx->d_filter2.p, ctx->d_slots.p,
                             ctx->d_postings.p, mask, shift};
    ctx->genes = KpGenes{ctx->d_nib.p, ctx->d_nib_off.p, ctx->d_gene_len.p, n_genes, ctx->d_gene_prof.p, ctx->d_gene_has_n.p};
    ctx->n_genes = n_genes;
    ctx->gs_bits = 1;
    while (ctx->gs_bits < 18 && (2ull * (uint64_t)n_genes) >> ctx->gs_bits) ++ctx->gs_bits;
    ctx->max_gene_len = 0;
    for (int len : ctx->gene_len) ctx->max_gene_len = std::max(ctx->max_gene_len, len);
    ctx->n_postings = (int64_t)post.size();
    ctx->has_db = true;
    return KP_OK;
}

int64_t kp_db_n_postings(const kp_ctx *ctx) { return ctx && ctx->has_db ? ctx->n_postings : 0; }

int kp_batch_create_async(kp_ctx *ctx, int32_t n_asm, const uint32_t *words, const int64_t *asm_word_off,
                          const int32_t *ctg_start, const int32_t *ctg_len, const int32_t *asm_first_ctg,
                          const int32_t *n_runs, const int32_t *asm_first_nrun, kp_batch **out) {
    return batch_make(ctx, n_asm, words, false, asm_word_off, ctg_start, ctg_len, asm_first_ctg, n_runs, asm_first_nrun, out);
}

int kp_batch_create(kp_ctx *ctx, int32_t n_asm, const uint32_t *words, const int64_t *asm_word_off,
                    const int32_t *ctg_start, const int32_t *ctg_len, const int32_t *asm_first_ctg,
                    const int32_t *n_runs, const int32_t *asm_first_nrun, kp_batch **out) {
    const int rc = batch_make(ctx, n_asm, words, false, asm_word_off, ctg_start, ctg_len, asm_first_ctg, n_runs, asm_first_nrun, out);
    if (rc) return rc;
    // the caller may free `words` on return
    if (hipStreamSynchronize(ctx->copy) != hipSuccess) {
        kp_batch_destroy(*out);
        *out = nullptr;
        return kp_fail(ctx, KP_EHIP, "H2D copy of the batch failed");
    }
    return KP_OK;
}

int kp_batch_create_device(kp_ctx *ctx, int32_t n_asm, const uint32_t *d_words, const int64_t *asm_word_off,
                           const int32_t *ctg_start, const int32_t *ctg_len, const int32_t *asm_first_ctg,
                           const int32_t *n_runs, const int32_t *asm_first_nrun, kp_batch **out) {
    return batch_make(ctx, n_asm, d_words, true, asm_word_off, ctg_start, ctg_len, asm_first_ctg, n_runs, asm_first_nrun, out);
}

int kp_batch_depends_on(kp_ctx *ctx, kp_batch *b, kp_batch *other) {
    if (!ctx || !b || b->ctx != ctx || !other || !other->ctx) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    if (other->ctx->device != ctx->device) return kp_fail(ctx, KP_EINVAL, "batches live on different devices");
    b->after = other;
    return KP_OK;
}

int kp_batch_upload_wait(kp_ctx *ctx, kp_batch *b) {
    if (!ctx || !b || b->ctx != ctx) return kp_fail(ctx, KP_EINVAL, "bad context/batch");
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KP_HIP_CHECK(ctx, hipEventSynchronize(b->in->ready));
    return KP_OK;
}

const void *kp_batch_device_words(const kp_batch *b) { return b ? (const void *)b->d_words : nullptr; }

void kp_batch_destroy(kp_batch *b) {
    if (!b) return;
    kp_ctx *ctx = b->ctx;
    if (ctx) {
        (void)hipSetDevice(ctx->device);
        if (b->in && b->in->ready) (void)hipEventSynchronize(b->in->ready);  // an upload still in flight
        quiesce_batch(b);
        if (b->w && b->w->owner == b) { b->w->owner = nullptr; b->w->aligned = false; b->w->finalised = false; }
        recycle_input(ctx, b->in);
        detach_batch(ctx, b);
    }
    delete b;
}

static int enqueue_align(kp_ctx *ctx, kp_batch *b, KpWork *w) {
    const size_t n_asm = (size_t)b->n_asm;
    if (!w->have_events) {
        for (auto &e : w->ev) KP_HIP_CHECK(ctx, hipEventCreate(&e));
        w->have_events = true;
    }
    if (!w->astream) KP_HIP_CHECK(ctx, hipStreamCreateWithFlags(&w->astream, hipStreamNonBlocking));
    if (!w->jstream) {
        // (KAPTIVE_AMD_JOIN_STREAM_PRIO=1: experiment -- the join kernels' blocks need 128-176 VGPRs, a SIMD with three band-fill waves
        // has 56 free; from a queue of the highest priority they take the place of a fill wave that retires)
        static const bool jprio = [] { const char *e = std::getenv("KAPTIVE_AMD_JOIN_STREAM_PRIO"); return e && std::atoi(e) != 0; }();
        if (jprio) KP_HIP_CHECK(ctx, create_priority_stream(&w->jstream));
        else KP_HIP_CHECK(ctx, hipStreamCreateWithFlags(&w->jstream, hipStreamNonBlocking));
        KP_HIP_CHECK(ctx, hipEventCreateWithFlags(&w->ev_jfork, hipEventDisableTiming));
        KP_HIP_CHECK(ctx, hipEventCreateWithFlags(&w->ev_jdone, hipEventDisableTiming));
    }
    hipStream_t stream = w->astream;
    hipEvent_t *ev = w->ev;
    if ((uint64_t)n_asm * w->anchor_cap > 0xFFFFFFF0ull)
        return kp_fail(ctx, KP_EOVERFLOW, "anchor buffer would exceed 2^32 entries; use smaller batches");
    if (((uint64_t)b->view.total_words << 4) >> KP_CAND_POS_BITS)
        return kp_fail(ctx, KP_EOVERFLOW, "a batch holds at most 2^33 bases (candidate positions); use smaller batches");
    KP_HIP_CHECK(ctx, w->d_anchors_a.reserve(n_asm * w->anchor_cap));
    KP_HIP_CHECK(ctx, w->d_anchors_b.reserve(n_asm * w->anchor_cap));
    KP_HIP_CHECK(ctx, w->d_counts.reserve(2 * n_asm + KP_N_CLASSES));
    KP_HIP_CHECK(ctx, w->d_sub_counts.reserve(n_asm * KP_ANCHOR_SUBS));
    KP_HIP_CHECK(ctx, w->d_seg.reserve(2 * n_asm));
    KP_HIP_CHECK(ctx, w->d_tasks.reserve(KP_N_CLASSES * (size_t)w->task_cap));
    KP_HIP_CHECK(ctx, w->d_results.reserve(KP_N_CLASSES * (size_t)w->task_cap));
    KP_HIP_CHECK(ctx, w->d_task_drop.reserve(KP_N_CLASSES * (size_t)w->task_cap));
    KP_HIP_CHECK(ctx, w->d_jscratch.reserve(kp_join_chain_scratch_bytes()));
    {   // a table holds every distinct minimizer of the longest assembly (2 / 11 of its bases) at a load of at most a half
        uint32_t lg = 12;
        while (((uint64_t)1 << lg) < (uint64_t)b->max_asm_bases * 2 / 5 + 1 && lg < 31) ++lg;
        w->occ_log2 = lg;
        w->occ_slots = std::max<uint32_t>(ctx->occ_slots, 2u);
        KP_HIP_CHECK(ctx, w->d_occ_keys.reserve((size_t)w->occ_slots << lg));
        KP_HIP_CHECK(ctx, w->d_occ_cnts.reserve((size_t)w->occ_slots << lg));
        KP_HIP_CHECK(ctx, w->d_occ_state.reserve(kp_occ_state_words(n_asm, w->occ_slots)));
    }
    KP_HIP_CHECK(ctx, w->d_task_order.reserve(ORDER_HEAD + KP_N_CLASSES * (size_t)w->task_cap));
    KP_HIP_CHECK(ctx, w->d_cand.reserve(w->cand_cap));
    KP_HIP_CHECK(ctx, w->d_cand_count.reserve(2));  // [0] the streaming kernel's candidates (front), [1] the edge kernel's (back)
    KP_HIP_CHECK(ctx, w->d_ends.reserve(KP_N_CLASSES * (size_t)w->task_cap));
    KP_HIP_CHECK(ctx, w->d_trace_top.reserve(4));  // [0] trace units handed out, [1..2] the fill kernel's quad counters (four 32-bit words)
    KP_HIP_CHECK(ctx, w->d_trace.reserve(w->trace_cap));
    KP_HIP_CHECK(ctx, w->d_groups.reserve(w->group_cap));
    KP_HIP_CHECK(ctx, w->d_joins.reserve(KP_N_CLASSES * (size_t)w->join_cap));
    KP_HIP_CHECK(ctx, w->d_join_counts.reserve(1 + KP_N_CLASSES));
    KP_HIP_CHECK(ctx, hipStreamWaitEvent(stream, b->in->ready, 0));  // the batch's H2D copies
    if (b->after && b->after->in) KP_HIP_CHECK(ctx, hipStreamWaitEvent(stream, b->after->in->ready, 0));
    KP_HIP_CHECK(ctx, hipMemsetAsync(w->d_counts.p, 0, (2 * n_asm + KP_N_CLASSES) * sizeof(uint32_t), stream));
    KP_HIP_CHECK(ctx, hipMemsetAsync(w->d_sub_counts.p, 0, n_asm * KP_ANCHOR_SUBS * sizeof(uint32_t), stream));
    KP_HIP_CHECK(ctx, hipMemsetAsync(w->d_cand_count.p, 0, 2 * sizeof(unsigned long long), stream));
    KP_HIP_CHECK(ctx, hipMemsetAsync(w->d_task_order.p, 0, ORDER_HEAD * sizeof(uint32_t), stream));
    KP_HIP_CHECK(ctx, hipMemsetAsync(w->d_trace_top.p, 0, 4 * sizeof(unsigned long long), stream));
    KP_HIP_CHECK(ctx, hipMemsetAsync(w->d_join_counts.p, 0, (1 + KP_N_CLASSES) * sizeof(uint32_t), stream));
    uint32_t *d_task_count = w->d_counts.p + n_asm;
    const uint32_t sub_cap = w->anchor_cap / KP_ANCHOR_SUBS;
    // compact anchor keys: as many bits per field as this batch and database can set
    auto bits_for = [](uint64_t max_value) { uint32_t n = 1; while (n < 63 && (max_value >> n)) ++n; return n; };
    w->key_bits.qb = std::min<uint32_t>(16, bits_for((uint64_t)std::max(ctx->max_gene_len, 1)));
    w->key_bits.db = std::min<uint32_t>(30, bits_for((uint64_t)b->max_asm_bases + KP_DIAG_BIAS));
    KP_HIP_CHECK(ctx, hipEventRecord(ev[0], stream));
    kp_launch_scan(b->view, ctx->index, w->d_cand.p, w->d_cand_count.p, w->cand_cap, w->d_anchors_a.p, w->d_sub_counts.p,
                   sub_cap, w->key_bits, ctx->opt.scan_mode, b->n_ctg_total, stream, ev[1]);
    if (!ctx->opt.library_sort && kp_bsort_fits(2u * (uint32_t)ctx->n_genes)) {
        // buckets of the gene/strand field, each sorted on its own (kp_bsort.hip); sorted keys end up where the chaining reads them
        kp_launch_anchor_bsort(b->view, w->d_anchors_a.p, w->d_sub_counts.p, sub_cap, w->d_anchors_b.p, w->d_anchors_a.p,
                               w->d_counts.p, w->d_counts.p + n_asm + KP_N_CLASSES, 2u * (uint32_t)ctx->n_genes, w->key_bits, stream);
    } else {  // `library_sort`: compaction + the library's segmented radix sort (tests compare the two)
        kp_launch_anchor_compact(b->view, w->d_anchors_a.p, w->d_sub_counts.p, sub_cap, w->d_anchors_b.p, w->d_counts.p,
                                 w->d_counts.p + n_asm + KP_N_CLASSES, stream);
        int rc = kp_sort_anchors(ctx, w->d_anchors_b.p, w->d_anchors_a.p, w->d_counts.p, w->anchor_cap, b->n_asm,
                                 &w->sort_temp, &w->sort_temp_bytes, w->d_seg.p, w->d_seg.p + n_asm,
                                 (int)(w->key_bits.qb + w->key_bits.db) + ctx->gs_bits,
                                 stream);
        if (rc) return rc;
    }
    KP_HIP_CHECK(ctx, hipEventRecord(ev[2], stream));
    kp_launch_occ_cut(b->view, ctx->d_gene_len.p, w->d_anchors_a.p, w->d_counts.p, w->anchor_cap, w->key_bits, w->d_occ_keys.p, w->d_occ_cnts.p,
                      w->d_occ_state.p, w->d_trace_top.p + 3, w->occ_slots, w->occ_log2, stream);  // (words 1-2 of trace_top are the fill kernel's quad counters)
    {
        static const int n_dummy = [] { const char *e = std::getenv("KAPTIVE_AMD_DUMMY_LAUNCHES"); return e ? std::atoi(e) : 0; }();
        for (int i = 0; i < n_dummy; ++i) hipLaunchKernelGGL(kp_noop_kernel, dim3(1), dim3(64), 0, stream);
    }
    kp_launch_chain(b->view, w->d_anchors_a.p, w->d_counts.p, w->anchor_cap, w->key_bits, w->d_tasks.p,
                    d_task_count, w->task_cap, w->d_groups.p, w->d_join_counts.p, w->group_cap, stream);
    // kp-align v5: the chains of a group's anchors, their joined fill and walk-back need the groups and the sorted anchors only:
    // they fork off here and run on the work set's second stream beside the band tasks' order, fill and traceback
    KP_HIP_CHECK(ctx, hipMemsetAsync(w->d_task_drop.p, 0, KP_N_CLASSES * (size_t)w->task_cap, stream));
    KP_HIP_CHECK(ctx, hipEventRecord(w->ev_jfork, stream));
    KP_HIP_CHECK(ctx, hipStreamWaitEvent(w->jstream, w->ev_jfork, 0));
    kp_launch_join_chain(b->view, ctx->genes, w->d_anchors_a.p, w->anchor_cap, w->key_bits, w->d_tasks.p, w->task_cap, w->d_groups.p,
                         w->d_join_counts.p, w->group_cap, w->d_joins.p, w->d_join_counts.p + 1, w->join_cap, w->d_jscratch.p, w->jstream);
    static const bool join_after_fill = [] { const char *e = std::getenv("KAPTIVE_AMD_JOIN_AFTER_FILL"); return e && std::atoi(e) != 0; }();
    auto join_fill_and_walk = [&]() -> int {
        kp_launch_join_fill(b->view, ctx->genes, w->d_joins.p, w->d_join_counts.p + 1, w->join_cap, w->d_trace.p, w->d_trace_top.p, w->trace_cap,
                            w->jstream);
        kp_launch_join_trace(b->view, ctx->genes, w->d_joins.p, w->d_join_counts.p + 1, w->join_cap, w->task_cap, w->d_trace.p, w->d_task_drop.p,
                             w->jstream);
        KP_HIP_CHECK(ctx, hipEventRecord(w->ev_jdone, w->jstream));
        return KP_OK;
    };
    if (!join_after_fill) { if (int rc = join_fill_and_walk()) return rc; }
    kp_launch_task_order(b->view, ctx->genes, w->d_anchors_a.p, w->anchor_cap, w->key_bits, w->d_tasks.p, d_task_count, w->task_cap,
                         w->d_results.p, w->d_task_order.p, w->d_task_order.p + ORDER_HEAD, stream);
    KP_HIP_CHECK(ctx, hipEventRecord(ev[3], stream));
    // all four band classes in one fill launch, then the traceback (kp_sw.hip): ev[3]..ev[4] is the fill, ev[4]..ev[5]
    // the traceback; the remaining event slots stay in the layout and read 0
    kp_launch_sw(b->view, ctx->genes, w->d_tasks.p, w->d_task_order.p + KP_ORDER_COUNTS, w->task_cap, w->d_task_order.p + ORDER_HEAD,
                 w->d_ends.p, w->d_trace.p, w->d_trace_top.p, w->trace_cap, w->d_results.p, ctx->max_gene_len > KP_FILL16_MAX_GENE_LEN,
                 stream, ev[4]);
    if (join_after_fill) {  // (experiment: the joined fill dispatched once the band fill is through, beside the traceback)
        KP_HIP_CHECK(ctx, hipStreamWaitEvent(w->jstream, ev[4], 0));
        if (int rc = join_fill_and_walk()) return rc;
    }
    // ev[5]..ev[6]: what is left of the join kernels once the band tasks are through (the "sw64" slot of kp_batch_profile; the
    // last slot reads 0)
    KP_HIP_CHECK(ctx, hipEventRecord(ev[5], stream));
    KP_HIP_CHECK(ctx, hipStreamWaitEvent(stream, w->ev_jdone, 0));
    for (int c = 2; c < KP_N_CLASSES; ++c) KP_HIP_CHECK(ctx, hipEventRecord(ev[4 + c], stream));
    KP_HIP_CHECK(ctx, hipGetLastError());
    return KP_OK;
}

// sizes of the pass's buffers from what the context has learnt so far (first use: the options)
static void size_work(kp_ctx *ctx, const kp_batch *b, KpWork *w) {
    if (ctx->anchor_cap == 0) ctx->anchor_cap = ctx->opt.anchor_cap;
    ctx->anchor_cap = std::max<uint32_t>((ctx->anchor_cap + KP_ANCHOR_SUBS - 1) / KP_ANCHOR_SUBS, 16u) * KP_ANCHOR_SUBS;
    if (ctx->tasks_per_asm == 0) ctx->tasks_per_asm = ctx->opt.tasks_per_asm;
    if (ctx->cand_frac <= 0.0) ctx->cand_frac = 0.004;  // 2 / 11 of the positions are seeds; ~1 % of those pass both filters
    if (ctx->hit_cap == 0) ctx->hit_cap = ctx->opt.hit_cap;
    w->anchor_cap = ctx->anchor_cap;
    w->task_cap = (uint32_t)std::min<uint64_t>((uint64_t)std::max(b->n_asm, 1) * ctx->tasks_per_asm, 1u << 28);
    ctx->words_hw = std::max(ctx->words_hw, b->view.total_words + b->view.total_words / 64);  // (batches of one stream differ by a per cent or so)
    w->cand_cap = std::max<uint64_t>(1 << 16, (uint64_t)((double)ctx->words_hw * 4.0 * ctx->cand_frac));
    w->hit_cap = ctx->hit_cap;
    if (ctx->trace_units_per_asm == 0) ctx->trace_units_per_asm = (uint64_t)ctx->opt.trace_kb_per_asm * 64;
    w->trace_cap = std::max<uint64_t>(4096, (uint64_t)std::max(b->n_asm, 1) * ctx->trace_units_per_asm);
    if (ctx->group_cap == 0) ctx->group_cap = 1024;
    if (ctx->join_cap == 0) ctx->join_cap = 1024;
    w->group_cap = ctx->group_cap; w->join_cap = ctx->join_cap;
}

int kp_batch_align(kp_ctx *ctx, kp_batch *b) {
    if (!ctx || !b || b->ctx != ctx) return kp_fail(ctx, KP_EINVAL, "bad context/batch");
    if (!ctx->has_db) return kp_fail(ctx, KP_ESTATE, "no database loaded");
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    KpWork *w = work_of(ctx, b);
    if (!w) {  // next work set, round-robin; whoever held it loses its results
        w = &ctx->work[ctx->next_slot++ % KP_WORK_SLOTS];
        if (w->owner) {
            // its reductions may still be reading the hit tables this pass's finalisation will rewrite
            KP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->post));
            for (auto &r : w->runs)
                if (r && r->stream) { KP_HIP_CHECK(ctx, hipStreamSynchronize(r->stream)); if (r->aux) KP_HIP_CHECK(ctx, hipStreamSynchronize(r->aux)); }
            w->owner->w = nullptr;
        }
        w->owner = b;
        b->w = w; b->last_w = w;
    }
    size_work(ctx, b, w);
    w->aligned = false; w->finalised = false;
    for (auto &v : w->h_tasks) v.clear();
    for (auto &r : w->runs)
        if (r) { r->split = false; r->scored = false; r->reduced = false; r->sums_valid = false; }
    w->stats[4] = 0;
    int rc = enqueue_align(ctx, b, w);
    if (rc) return rc;
    w->aligned = true;
    return KP_OK;
}

// hit-table finalisation on the device: compaction of the band-task results into per-assembly lists, emission order,
// duplicates, mapq.  Grows hit_cap and repeats if an assembly produced more hits than its region holds.
static int finalise_hits_on_device(kp_ctx *ctx, kp_batch *b, KpWork *w) {
    const size_t n_asm = (size_t)b->n_asm;
    for (int attempt = 0;; ++attempt) {
        KP_HIP_CHECK(ctx, w->d_hits_raw.reserve(n_asm * w->hit_cap));
        KP_HIP_CHECK(ctx, w->d_hits.reserve(n_asm * w->hit_cap));
        KP_HIP_CHECK(ctx, w->d_keys.reserve(n_asm * w->hit_cap * 3));
        KP_HIP_CHECK(ctx, w->d_hit_counts.reserve(2 * n_asm));
        KP_HIP_CHECK(ctx, w->d_cells.reserve(1));
        KP_HIP_CHECK(ctx, hipMemsetAsync(w->d_hit_counts.p, 0, 2 * n_asm * sizeof(uint32_t), ctx->post));
        KP_HIP_CHECK(ctx, hipMemsetAsync(w->d_cells.p, 0, sizeof(unsigned long long), ctx->post));
        kp_launch_hit_finalise(b->view, ctx->d_gene_len.p, w->d_tasks.p, w->d_results.p, w->d_task_drop.p, w->d_counts.p + n_asm,
                               w->task_cap, w->d_hits_raw.p, w->d_hit_counts.p, w->hit_cap, w->d_keys.p, w->d_hits.p,
                               w->d_hit_counts.p + n_asm, w->d_cells.p, ctx->d_ln.p, ctx->d_ln.p + KP_MAPQ_LN_HALF_SIZE, w->d_joins.p,
                               w->d_join_counts.p + 1, w->join_cap, ctx->post);
        KP_HIP_CHECK(ctx, hipGetLastError());
        w->h_hit_counts.resize(2 * n_asm);
        unsigned long long cells = 0;
        {
            Fetch f(ctx, ctx->post);
            int frc;
            if ((frc = f.begin(2 * n_asm * sizeof(uint32_t) + sizeof cells)) || (frc = f.add(w->h_hit_counts.data(), w->d_hit_counts.p, 2 * n_asm * sizeof(uint32_t))) ||
                (frc = f.add(&cells, w->d_cells.p, sizeof cells)) || (frc = f.finish()))
                return frc;
        }
        uint32_t max_raw = 0;
        for (size_t a = 0; a < n_asm; ++a) max_raw = std::max(max_raw, w->h_hit_counts[a]);
        if (max_raw <= w->hit_cap) { w->stats[2] = (int64_t)cells; break; }
        if (attempt >= 2) return kp_fail(ctx, KP_EOVERFLOW, "hit buffers overflowed repeatedly");
        w->hit_cap = (max_raw + max_raw / 4 + 255u) & ~255u;  // a quarter of headroom: later batches differ a little
        ctx->hit_cap = std::max(ctx->hit_cap, w->hit_cap);
        w->stats[4] += 1;
    }
    w->hit_off.assign(n_asm + 1, 0);
    for (size_t a = 0; a < n_asm; ++a) w->hit_off[a + 1] = w->hit_off[a] + (int64_t)w->h_hit_counts[n_asm + a];
    return KP_OK;
}

int kp_batch_wait(kp_ctx *ctx, kp_batch *b) {
    if (!ctx || !b || b->ctx != ctx) return kp_fail(ctx, KP_EINVAL, "bad context/batch");
    KpWork *w = work_of(ctx, b);
    if (!w || !w->aligned) return kp_fail(ctx, KP_ESTATE, w ? "kp_batch_align has not been called" : NO_RESULTS);
    if (w->finalised) return KP_OK;
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t n_asm = (size_t)b->n_asm;
    for (int attempt = 0;; ++attempt) {
        // the post stream picks up where this batch's alignment pass ends; later passes on ctx->stream are not waited for
        KP_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->post, w->ev[3 + KP_N_CLASSES], 0));
        w->h_counts.resize(2 * n_asm + KP_N_CLASSES);
        unsigned long long n_cand2[2] = {0, 0}, trace_top2[4] = {0, 0, 0, 0};
        {
            Fetch f(ctx, ctx->post);
            int frc;
            if ((frc = f.begin((2 * n_asm + KP_N_CLASSES) * sizeof(uint32_t) + sizeof n_cand2 + sizeof trace_top2 + sizeof w->h_join_counts)) ||
                (frc = f.add(w->h_counts.data(), w->d_counts.p, (2 * n_asm + KP_N_CLASSES) * sizeof(uint32_t))) ||
                (frc = f.add(n_cand2, w->d_cand_count.p, sizeof n_cand2)) || (frc = f.add(trace_top2, w->d_trace_top.p, sizeof trace_top2)) ||
                (frc = f.add(w->h_join_counts, w->d_join_counts.p, sizeof w->h_join_counts)) || (frc = f.finish()))
                return frc;
        }
        const unsigned long long trace_need = trace_top2[0], occ_need = trace_top2[3];  // (assemblies that needed their mid_occ)
        uint32_t max_join = 0;
        for (int c = 0; c < KP_N_CLASSES; ++c) max_join = std::max(max_join, w->h_join_counts[1 + c]);
        const uint32_t n_group = w->h_join_counts[0];
        if (std::getenv("KAPTIVE_AMD_JOIN_STATS"))
            std::fprintf(stderr, "[kp_batch_wait] %zu assemblies: %u groups, joins per band class %u %u %u %u, %llu assemblies needed their mid_occ (%u tables)\n", n_asm, n_group,
                         w->h_join_counts[1], w->h_join_counts[2], w->h_join_counts[3], w->h_join_counts[4], occ_need, w->occ_slots);
        const unsigned long long n_cand = n_cand2[0] + n_cand2[1];
        uint32_t max_slice = 0, max_task = 0;
        for (size_t a = 0; a < n_asm; ++a) max_slice = std::max(max_slice, w->h_counts[n_asm + KP_N_CLASSES + a]);
        for (int c = 0; c < KP_N_CLASSES; ++c) max_task = std::max(max_task, w->h_counts[n_asm + c]);
        const uint32_t sub_cap = w->anchor_cap / KP_ANCHOR_SUBS;
        if (max_slice <= sub_cap && max_task <= w->task_cap && n_cand <= w->cand_cap && trace_need <= w->trace_cap &&
            n_group <= w->group_cap && max_join <= w->join_cap && occ_need <= w->occ_slots) {
            // Everything fitted.  What came close makes room for the passes after this one: the sub-slice an anchor
            // lands in depends on the order in which the scan's waves flushed, so the fullest slice varies from pass to
            // pass on the same input, and a rerun costs a whole pass.
            if (max_slice + max_slice / 4 > sub_cap)
                ctx->anchor_cap = std::max(ctx->anchor_cap, ((max_slice + max_slice / 2 + 15u) & ~15u) * KP_ANCHOR_SUBS);
            if (trace_need + trace_need / 16 > w->trace_cap && trace_need + trace_need / 4 <= (1ull << 32))
                ctx->trace_units_per_asm = std::max<uint64_t>(ctx->trace_units_per_asm,
                                                              (trace_need + trace_need / 4 + n_asm - 1) / std::max<size_t>(n_asm, 1));
            if ((uint64_t)max_task + max_task / 16 > w->task_cap)
                ctx->tasks_per_asm = std::max<uint32_t>(ctx->tasks_per_asm,
                                                        (uint32_t)(((uint64_t)max_task + max_task / 4 + n_asm - 1) / std::max<size_t>(n_asm, 1)));
            if (n_cand + n_cand / 16 > w->cand_cap)
                ctx->cand_frac = std::max(ctx->cand_frac, (double)(n_cand + n_cand / 4) / ((double)b->view.total_words * 4.0));
            if (n_group + n_group / 4 > w->group_cap) ctx->group_cap = std::max(ctx->group_cap, 2 * n_group);
            if (max_join + max_join / 4 > w->join_cap) ctx->join_cap = std::max(ctx->join_cap, 2 * max_join);
            break;
        }
        if (attempt >= 4) return kp_fail(ctx, KP_EOVERFLOW, "anchor/task buffers overflowed repeatedly");
        // a region overflowed: counts kept counting, so they say how much room a clean rerun needs.  The context
        // remembers it (with some headroom, later batches differ a little) for every later pass.
        if (n_cand > w->cand_cap) {
            w->cand_cap = n_cand + n_cand / 8;
            ctx->cand_frac = std::max(ctx->cand_frac, (double)w->cand_cap / ((double)b->view.total_words * 4.0) * 1.0001);
        }
        if (max_slice > sub_cap) {
            w->anchor_cap = ((max_slice + max_slice / 2 + 15u) & ~15u) * KP_ANCHOR_SUBS;
            ctx->anchor_cap = std::max(ctx->anchor_cap, w->anchor_cap);
        }
        if (trace_need > w->trace_cap) {  // (a pass cut short by another overflow reports less than it will need)
            if (trace_need > (1ull << 32)) return kp_fail(ctx, KP_EOVERFLOW, "DP trace would exceed 64 GB; use smaller batches");
            w->trace_cap = std::min<uint64_t>(trace_need + trace_need / 4, 1ull << 32);  // later batches differ by a few per cent
            ctx->trace_units_per_asm = std::max<uint64_t>(ctx->trace_units_per_asm, (w->trace_cap + n_asm - 1) / std::max<size_t>(n_asm, 1));
        }
        if (n_group > w->group_cap) { w->group_cap = n_group + n_group / 4 + 64; ctx->group_cap = std::max(ctx->group_cap, w->group_cap); }
        if (occ_need > w->occ_slots) ctx->occ_slots = std::max<uint32_t>(ctx->occ_slots, (uint32_t)std::min<unsigned long long>(occ_need + occ_need / 4 + 1, 1u << 16));
        if (max_join > w->join_cap) { w->join_cap = max_join + max_join / 4 + 64; ctx->join_cap = std::max(ctx->join_cap, w->join_cap); }
        if (max_task > w->task_cap) {
            w->task_cap = (max_task + max_task / 8 + 1023u) & ~1023u;
            ctx->tasks_per_asm = std::max<uint32_t>(ctx->tasks_per_asm, (uint32_t)((w->task_cap + n_asm - 1) / std::max<size_t>(n_asm, 1)));
        }
        w->stats[4] += 1;
        int rc = enqueue_align(ctx, b, w);
        if (rc) return rc;
    }
    int64_t n_anchor = 0, n_task = 0;
    for (size_t a = 0; a < n_asm; ++a) n_anchor += w->h_counts[a];
    for (int c = 0; c < KP_N_CLASSES; ++c) n_task += w->h_counts[n_asm + c];
    for (auto &v : w->h_tasks) v.clear();
    w->h_joins.clear();
    int rc = finalise_hits_on_device(ctx, b, w);
    if (rc) return rc;
    w->stats[0] = n_anchor; w->stats[1] = n_task; w->stats[3] = w->hit_off[n_asm];
    w->finalised = true;
    return KP_OK;
}

// the batch's work set with finalised hit tables, or null after recording the error
static KpWork *finalised_work(kp_ctx *ctx, kp_batch *b) {
    KpWork *w = work_of(ctx, b);
    if (!w) { kp_fail(ctx, KP_ESTATE, NO_RESULTS); return nullptr; }
    if (!w->finalised) { kp_fail(ctx, KP_ESTATE, "kp_batch_wait has not completed"); return nullptr; }
    return w;
}

int kp_batch_hit_offsets(kp_ctx *ctx, kp_batch *b, int64_t *hit_off) {
    if (!ctx || !b || b->ctx != ctx || !hit_off) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = finalised_work(ctx, b);
    if (!w) return KP_ESTATE;
    std::memcpy(hit_off, w->hit_off.data(), w->hit_off.size() * sizeof(int64_t));
    return KP_OK;
}

int kp_batch_hits(kp_ctx *ctx, kp_batch *b, kp_hit *out, int64_t cap) {
    if (!ctx || !b || b->ctx != ctx || (!out && cap > 0)) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = finalised_work(ctx, b);
    if (!w) return KP_ESTATE;
    const size_t n_asm = (size_t)b->n_asm;
    const int64_t total = w->hit_off[n_asm];
    if (cap < total) return kp_fail(ctx, KP_EINVAL, "hit buffer too small");
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (size_t a = 0; a < n_asm; ++a) {  // regions are contiguous per assembly; copy each used prefix
        const int64_t n = w->hit_off[a + 1] - w->hit_off[a];
        if (n > 0)
            KP_HIP_CHECK(ctx, hipMemcpyAsync(out + w->hit_off[a], w->d_hits.p + a * (size_t)w->hit_cap,
                                             (size_t)n * sizeof(kp_hit), hipMemcpyDeviceToHost, ctx->post));
    }
    KP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->post));
    return KP_OK;
}

int kp_batch_set_hits(kp_ctx *ctx, kp_batch *b, const kp_hit *hits, const int64_t *hit_off) {
    if (!ctx || !b || b->ctx != ctx || !hit_off) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = finalised_work(ctx, b);
    if (!w) return KP_ESTATE;
    const size_t n_asm = (size_t)b->n_asm;
    int64_t max_n = 0;
    for (size_t a = 0; a < n_asm; ++a) {
        const int64_t n = hit_off[a + 1] - hit_off[a];
        if (n < 0 || (n > 0 && !hits)) return kp_fail(ctx, KP_EINVAL, "hit offsets must ascend");
        max_n = std::max(max_n, n);
    }
    if (max_n > (1 << 24)) return kp_fail(ctx, KP_EOVERFLOW, "too many hits for one assembly");
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // nothing may still be reading the table that is about to be replaced
    KP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->post));
    for (auto &r : w->runs)
        if (r && r->stream) { KP_HIP_CHECK(ctx, hipStreamSynchronize(r->stream)); if (r->aux) KP_HIP_CHECK(ctx, hipStreamSynchronize(r->aux)); }
    if ((uint64_t)max_n > w->hit_cap) {
        w->hit_cap = ((uint32_t)max_n + 255u) & ~255u;
        ctx->hit_cap = std::max(ctx->hit_cap, w->hit_cap);
    }
    KP_HIP_CHECK(ctx, w->d_hits.reserve(n_asm * w->hit_cap));
    KP_HIP_CHECK(ctx, w->d_hit_counts.reserve(2 * n_asm));
    w->h_hit_counts.resize(2 * n_asm);
    w->hit_off.assign(n_asm + 1, 0);
    for (size_t a = 0; a < n_asm; ++a) {
        const int64_t n = hit_off[a + 1] - hit_off[a];
        if (n > 0)
            KP_HIP_CHECK(ctx, hipMemcpy(w->d_hits.p + a * (size_t)w->hit_cap, hits + hit_off[a], (size_t)n * sizeof(kp_hit), hipMemcpyHostToDevice));
        w->h_hit_counts[n_asm + a] = (uint32_t)n;
        w->hit_off[a + 1] = w->hit_off[a] + n;
    }
    if (n_asm)
        KP_HIP_CHECK(ctx, hipMemcpy(w->d_hit_counts.p + n_asm, w->h_hit_counts.data() + n_asm, n_asm * sizeof(uint32_t), hipMemcpyHostToDevice));
    w->stats[3] = w->hit_off[n_asm];
    for (auto &r : w->runs)
        if (r) { r->split = false; r->scored = false; r->reduced = false; r->sums_valid = false; }
    return KP_OK;
}

int kp_batch_stats(kp_ctx *ctx, kp_batch *b, int64_t *stats5) {
    if (!ctx || !b || b->ctx != ctx || !stats5) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = finalised_work(ctx, b);
    if (!w) return KP_ESTATE;
    std::memcpy(stats5, w->stats, sizeof w->stats);
    return KP_OK;
}

int kp_batch_profile(kp_ctx *ctx, kp_batch *b, float *ms7, int64_t *bytes_scanned) {
    if (!ctx || !b || b->ctx != ctx || !ms7) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = finalised_work(ctx, b);
    if (!w) return KP_ESTATE;
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (int i = 0; i < 3 + KP_N_CLASSES; ++i)
        if (hipEventElapsedTime(&ms7[i], w->ev[i], w->ev[i + 1]) != hipSuccess)
            return kp_fail(ctx, KP_EHIP, "event timing failed");
    if (bytes_scanned) *bytes_scanned = 4 * b->view.total_words;
    return KP_OK;
}

int64_t kp_batch_anchors(kp_ctx *ctx, kp_batch *b, int32_t a, uint64_t *out, int64_t cap) {
    if (!ctx || !b || b->ctx != ctx || a < 0 || a >= b->n_asm) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = finalised_work(ctx, b);
    if (!w) return KP_ESTATE;
    const int64_t n = w->h_counts[(size_t)a];
    const int64_t m = std::min(n, cap);
    if (out && m > 0) {
        if (hipMemcpy(out, w->d_anchors_a.p + (size_t)a * w->anchor_cap, (size_t)m * sizeof(uint64_t),
                      hipMemcpyDeviceToHost) != hipSuccess)
            return kp_fail(ctx, KP_EHIP, "D2H anchors failed");
        for (int64_t i = 0; i < m; ++i) out[i] = kp_key_unpack(out[i], w->key_bits);  // callers see the spec's layout
    }
    return n;
}

int64_t kp_batch_tasks(kp_ctx *ctx, kp_batch *b, int32_t a, int32_t *out8, int64_t cap) {
    if (!ctx || !b || b->ctx != ctx || a < 0 || a >= b->n_asm) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = finalised_work(ctx, b);
    if (!w) return KP_ESTATE;
    for (int c = 0; c < KP_N_CLASSES; ++c) {  // fetched on first use: only the stage tests look at tasks
        const size_t nt = w->h_counts[(size_t)b->n_asm + c];
        if (w->h_tasks[c].size() == nt) continue;
        w->h_tasks[c].resize(nt);
        if (nt && hipMemcpy(w->h_tasks[c].data(), w->d_tasks.p + (size_t)c * w->task_cap, nt * sizeof(KpTask),
                            hipMemcpyDeviceToHost) != hipSuccess)
            return kp_fail(ctx, KP_EHIP, "D2H tasks failed");
    }
    int64_t n = 0;
    for (int c = 0; c < KP_N_CLASSES; ++c)
        for (const KpTask &t : w->h_tasks[c]) {
            if (t.asm_id != a || t.n_anchors == 0) continue;  // (n_anchors == 0: a cluster the chaining rejected)
            if (out8 && n < cap) {
                int32_t *o = out8 + 8 * n;
                o[0] = t.gs; o[1] = t.contig; o[2] = t.lo; o[3] = t.width; o[4] = t.n_anchors; o[5] = (int32_t)(t.qspan & 0xFFFFu); o[6] = (int32_t)(t.qspan >> 16); o[7] = t.chain_score;
            }
            ++n;
        }
    return n;
}

int64_t kp_batch_task_results(kp_ctx *ctx, kp_batch *b, int32_t a, int32_t *out7, int64_t cap) {
    if (!ctx || !b || b->ctx != ctx || a < 0 || a >= b->n_asm) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    const int64_t n_tasks = kp_batch_tasks(ctx, b, a, nullptr, 0);  // (also fetches the task lists)
    if (n_tasks < 0) return n_tasks;
    KpWork *w = finalised_work(ctx, b);
    if (!w) return KP_ESTATE;
    int64_t n = 0;
    std::vector<KpSwResult> res;
    for (int c = 0; c < KP_N_CLASSES; ++c) {
        const size_t nt = w->h_tasks[c].size();
        res.resize(nt);
        if (nt && hipMemcpy(res.data(), w->d_results.p + (size_t)c * w->task_cap, nt * sizeof(KpSwResult), hipMemcpyDeviceToHost) != hipSuccess)
            return kp_fail(ctx, KP_EHIP, "D2H task results failed");
        for (size_t i = 0; i < nt; ++i) {
            if (w->h_tasks[c][i].asm_id != a || w->h_tasks[c][i].n_anchors == 0) continue;
            if (out7 && n < cap) {
                const KpSwResult &r = res[i];
                int32_t *o = out7 + 7 * n;
                o[0] = r.score; o[1] = r.q_start; o[2] = r.q_end; o[3] = r.t_start; o[4] = r.t_end; o[5] = r.matches; o[6] = r.block_len;
            }
            ++n;
        }
    }
    return n;
}

int64_t kp_batch_joins(kp_ctx *ctx, kp_batch *b, int32_t a, int32_t *out, int64_t cap) {
    if (!ctx || !b || b->ctx != ctx || a < 0 || a >= b->n_asm) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = finalised_work(ctx, b);
    if (!w) return KP_ESTATE;
    size_t total = 0;
    for (int c = 0; c < KP_N_CLASSES; ++c) total += w->h_join_counts[1 + c];
    if (w->h_joins.size() != total) {  // fetched on first use: only the stage tests look at joins
        w->h_joins.resize(total);
        size_t at = 0;
        for (int c = 0; c < KP_N_CLASSES; ++c) {
            const size_t nj = w->h_join_counts[1 + c];
            if (nj && hipMemcpy(w->h_joins.data() + at, w->d_joins.p + (size_t)c * w->join_cap, nj * sizeof(KpJoin), hipMemcpyDeviceToHost) != hipSuccess)
                return kp_fail(ctx, KP_EHIP, "D2H joins failed");
            at += nj;
        }
    }
    int64_t n = 0;
    for (const KpJoin &J : w->h_joins) {
        if (J.asm_id != a) continue;
        if (out && n < cap) {
            int32_t *o = out + KP_JOIN_ROW_INTS * n;
            std::memset(o, 0, KP_JOIN_ROW_INTS * sizeof(int32_t));
            o[0] = J.gs; o[1] = J.contig; o[2] = J.n_pieces; o[3] = J.n_anchors; o[4] = J.chain_score; o[5] = J.width;
            for (int k = 0; k < J.n_pieces; ++k) {
                o[6 + k] = J.lo[k];
                int32_t *pr = o + 6 + KP_JOIN_MAX_PIECES + 11 * k;
                pr[0] = J.state[k]; pr[1] = J.visited[k];
                if (J.state[k] == 1) std::memcpy(pr + 2, J.res[k], 9 * sizeof(int32_t));
            }
        }
        ++n;
    }
    return n;
}

// ---- batched typing ---------------------------------------------------------------------------------------------------
int kp_db_load_typing(kp_ctx *ctx, const kp_typing_tables *t) {
    if (!ctx) return kp_fail(nullptr, KP_EINVAL, "null context");
    return kp_db_load_typing_group(ctx, 0, 0, ctx->n_genes, t);
}

int kp_db_load_typing_group(kp_ctx *ctx, int32_t group, int32_t gene_lo, int32_t gene_hi, const kp_typing_tables *t) {
    if (!ctx) return kp_fail(nullptr, KP_EINVAL, "null context");
    if (!ctx->has_db) return kp_fail(ctx, KP_ESTATE, "kp_db_load must come first");
    if (group < 0 || group >= KP_MAX_TYPING_GROUPS || gene_lo < 0 || gene_hi < gene_lo || gene_hi > ctx->n_genes)
        return kp_fail(ctx, KP_EINVAL, "bad typing group or gene range");
    if (!t || t->n_loci <= 0 || !t->gene_locus || !t->gene_extra || !t->gene_pos || !t->gene_strand || !t->locus_gene_off ||
        !t->locus_gene_len || !t->prot_off || !t->prot_len)
        return kp_fail(ctx, KP_EINVAL, "bad typing tables");
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t G = (size_t)(gene_hi - gene_lo), L = (size_t)t->n_loci;
    size_t prot_bytes = 0;
    int max_len = 0;
    for (size_t g = 0; g < G; ++g) {
        if (t->gene_locus[g] >= L) return kp_fail(ctx, KP_EINVAL, "gene_locus out of range");
        if (t->prot_len[g] < 0 || t->prot_off[g] < 0 || t->prot_len[g] > 65535) return kp_fail(ctx, KP_EINVAL, "bad protein table");
        prot_bytes = std::max(prot_bytes, (size_t)t->prot_off[g] + (size_t)t->prot_len[g]);
        max_len = std::max(max_len, t->prot_len[g]);
    }
    for (size_t l = 0; l < L; ++l)
    {
        if (t->locus_gene_off[l] < 0 || t->locus_gene_len[l] < 0 || (size_t)t->locus_gene_off[l] + (size_t)t->locus_gene_len[l] > G)
            return kp_fail(ctx, KP_EINVAL, "locus gene range out of bounds");
        if (t->locus_gene_len[l] > KP_MAX_LOCUS_GENES)
            return kp_fail(ctx, KP_EINVAL, "a locus has more genes than KP_MAX_LOCUS_GENES (width of the missing-gene mask)");
    }
    if (prot_bytes && !t->prot) return kp_fail(ctx, KP_EINVAL, "null protein data");
    if (ctx->groups.size() <= (size_t)group) ctx->groups.resize((size_t)group + 1);
    if (!ctx->groups[(size_t)group]) ctx->groups[(size_t)group].reset(new KpTypingGroup());
    KpTypingGroup &T = *ctx->groups[(size_t)group];
    int rc;
    if ((rc = upload(ctx, T.d_gene_locus, t->gene_locus, G))) return rc;
    if ((rc = upload(ctx, T.d_gene_extra, t->gene_extra, G))) return rc;
    if ((rc = upload(ctx, T.d_gene_pos, t->gene_pos, G))) return rc;
    if ((rc = upload(ctx, T.d_gene_strand, t->gene_strand, G))) return rc;
    if ((rc = upload(ctx, T.d_locus_off, t->locus_gene_off, L))) return rc;
    if ((rc = upload(ctx, T.d_locus_len, t->locus_gene_len, L))) return rc;
    if ((rc = upload(ctx, T.d_prot_db, t->prot, prot_bytes))) return rc;
    if ((rc = upload(ctx, T.d_prot_db_off, t->prot_off, G))) return rc;
    if ((rc = upload(ctx, T.d_prot_db_len, t->prot_len, G))) return rc;
    KP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    T.typing = KpTypingDb{T.d_gene_locus.p, T.d_gene_extra.p, T.d_gene_pos.p, T.d_gene_strand.p,
                          ctx->d_gene_len.p + gene_lo, T.d_locus_off.p, T.d_locus_len.p, T.d_prot_db.p,
                          T.d_prot_db_off.p, T.d_prot_db_len.p, (int32_t)G, (int32_t)t->n_loci};
    T.max_db_prot_len = max_len;
    T.gene_lo = gene_lo; T.gene_hi = gene_hi;
    return KP_OK;
}

// the typing group a batch currently addresses and the run of the batch's work set for it (created on first use)
static KpTypingGroup *typing_group(kp_ctx *ctx, const kp_batch *b) {
    return (size_t)b->group < ctx->groups.size() ? ctx->groups[(size_t)b->group].get() : nullptr;
}
static KpTypingRun &typing_run(KpWork *w, int32_t group) {
    if (w->runs.size() <= (size_t)group) w->runs.resize((size_t)group + 1);
    if (!w->runs[(size_t)group]) w->runs[(size_t)group].reset(new KpTypingRun());
    return *w->runs[(size_t)group];
}
static kp_ctx::RunCaps &run_caps(kp_ctx *ctx, int32_t group) {
    if (ctx->run_caps.size() <= (size_t)group) ctx->run_caps.resize((size_t)group + 1);
    kp_ctx::RunCaps &c = ctx->run_caps[(size_t)group];
    if (c.kept_cap == 0) c.kept_cap = (int)ctx->opt.kept_cap;
    if (c.piece_cap == 0) c.piece_cap = (int)ctx->opt.piece_cap;
    if (c.prot_cap == 0) c.prot_cap = (int)ctx->opt.prot_cap;
    return c;
}
static int ensure_run_streams(kp_ctx *ctx, KpTypingRun &R, int32_t group) {
    if (R.stream) return KP_OK;
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->group_streams.size() <= (size_t)group) ctx->group_streams.resize((size_t)group + 1);
    kp_ctx::GroupStreams &gs = ctx->group_streams[(size_t)group];
    if (!gs.stream) {
        KP_HIP_CHECK(ctx, create_priority_stream(&gs.stream));
        KP_HIP_CHECK(ctx, create_priority_stream(&gs.aux));
    }
    R.stream = gs.stream; R.aux = gs.aux;
    KP_HIP_CHECK(ctx, hipEventCreateWithFlags(&R.ev_fork, hipEventDisableTiming));
    KP_HIP_CHECK(ctx, hipEventCreateWithFlags(&R.ev_join, hipEventDisableTiming));
    return KP_OK;
}

int kp_batch_use_group(kp_ctx *ctx, kp_batch *b, int32_t group) {
    if (!ctx || !b || b->ctx != ctx) return kp_fail(ctx, KP_EINVAL, "bad context/batch");
    if (group < 0 || (size_t)group >= ctx->groups.size() || !ctx->groups[(size_t)group])
        return kp_fail(ctx, KP_EINVAL, "no typing tables loaded for this group");
    b->group = group;
    return KP_OK;
}

// the group's hits out of the batch's finalised hit table (sorted by gene, so they are one run per assembly), with gene
// indices made relative to the group's first gene.  A group that spans every gene of the context reads the table in place.
static int split_hits(kp_ctx *ctx, kp_batch *b, KpWork *w, const KpTypingGroup &T, KpTypingRun &R) {
    if (R.split) return KP_OK;
    const size_t n_asm = (size_t)b->n_asm;
    if (T.gene_lo == 0 && T.gene_hi == ctx->n_genes) {
        R.hits = w->d_hits.p;
        R.hit_n = w->d_hit_counts.p + n_asm;
    } else {
        KP_HIP_CHECK(ctx, R.d_hits.reserve(n_asm * w->hit_cap));
        KP_HIP_CHECK(ctx, R.d_hit_n.reserve(n_asm));
        kp_launch_hit_split(w->d_hits.p, w->d_hit_counts.p + n_asm, w->hit_cap, T.gene_lo, T.gene_hi, R.d_hits.p, R.d_hit_n.p,
                            b->n_asm, R.stream);
        KP_HIP_CHECK(ctx, hipGetLastError());
        R.hits = R.d_hits.p;
        R.hit_n = R.d_hit_n.p;
    }
    R.split = true;
    return KP_OK;
}

int kp_batch_score(kp_ctx *ctx, kp_batch *b, double min_gene_coverage, double *locus_scores, int32_t *locus_counts) {
    if (!ctx || !b || b->ctx != ctx || !locus_scores || !locus_counts) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpTypingGroup *Tp = typing_group(ctx, b);
    if (!Tp) return kp_fail(ctx, KP_ESTATE, "kp_db_load_typing has not been called");
    KpTypingGroup &T = *Tp;
    int rc = kp_batch_wait(ctx, b);  // hit tables final (and the post stream idle) when this returns
    if (rc) return rc;
    KpWork *w = work_of(ctx, b);
    KpTypingRun &R = typing_run(w, b->group);
    if ((rc = ensure_run_streams(ctx, R, b->group))) return rc;
    if ((rc = split_hits(ctx, b, w, T, R))) return rc;
    const size_t n = (size_t)b->n_asm * (size_t)T.typing.n_loci;
    KP_HIP_CHECK(ctx, R.d_scores.reserve(n));
    KP_HIP_CHECK(ctx, R.d_lcounts.reserve(n));
    kp_launch_score(b->view, R.hits, R.hit_n, w->hit_cap, T.typing, min_gene_coverage,
                    R.d_scores.p, R.d_lcounts.p, R.stream);
    KP_HIP_CHECK(ctx, hipGetLastError());
    {
        Fetch f(ctx, R.stream);
        int frc;
        if ((frc = f.begin(n * (sizeof(double) + sizeof(int32_t)))) || (frc = f.add(locus_scores, R.d_scores.p, n * sizeof(double))) ||
            (frc = f.add(locus_counts, R.d_lcounts.p, n * sizeof(int32_t))) || (frc = f.finish()))
            return frc;
    }
    R.prm.min_gene_coverage = min_gene_coverage;
    R.scored = true;
    return KP_OK;
}

static int enqueue_reduce(kp_ctx *ctx, kp_batch *b, KpWork *w) {
    KpTypingGroup &T = *typing_group(ctx, b);
    KpTypingRun &R = typing_run(w, b->group);
    const kp_ctx::RunCaps caps = run_caps(ctx, b->group);
    R.kept_cap = caps.kept_cap; R.piece_cap = caps.piece_cap; R.prot_cap = caps.prot_cap;
    const size_t n_asm = (size_t)b->n_asm;
    if ((uint64_t)n_asm * (uint64_t)R.prot_cap > 0x7FFFFFFFull)
        return kp_fail(ctx, KP_EOVERFLOW, "protein buffer would exceed 2^31 bytes; use smaller batches");
    const size_t slots = n_asm * (size_t)R.kept_cap;
    KP_HIP_CHECK(ctx, R.d_keys.reserve(n_asm * w->hit_cap));
    KP_HIP_CHECK(ctx, R.d_order.reserve(n_asm * w->hit_cap));
    KP_HIP_CHECK(ctx, R.d_flag.reserve(n_asm * w->hit_cap));
    KP_HIP_CHECK(ctx, R.d_kept.reserve(slots));
    KP_HIP_CHECK(ctx, R.d_pieces.reserve(n_asm * (size_t)R.piece_cap));
    KP_HIP_CHECK(ctx, R.d_summary.reserve(n_asm));
    KP_HIP_CHECK(ctx, R.d_prot.reserve(n_asm * (size_t)R.prot_cap));
    KP_HIP_CHECK(ctx, R.d_pairs.reserve(4 * slots + n_asm + 1));
    KP_HIP_CHECK(ctx, R.d_dp.reserve(8 * slots));
    int32_t *q_off = R.d_pairs.p, *q_len = q_off + slots, *t_off = q_len + slots, *t_len = t_off + slots;
    int32_t *pair_base = t_len + slots, *n_pairs = pair_base + n_asm;
    KP_HIP_CHECK(ctx, hipMemsetAsync(n_pairs, 0, sizeof(int32_t), R.stream));
    kp_launch_reduce(b->view, R.hits, R.hit_n, w->hit_cap, T.typing, R.prm, R.d_best.p,
                     R.d_keys.p, R.d_order.p, R.d_flag.p, R.d_kept.p, R.kept_cap, R.d_pieces.p, R.piece_cap,
                     R.d_summary.p, R.d_prot.p, R.prot_cap, q_off, q_len, t_off, t_len, n_pairs, pair_base, R.stream);
    // protein DP of every kept hit against its database protein (pair list is compact; its length lives on the device)
    const int n_blocks = (int)std::min<size_t>(std::max<size_t>(slots, 1), 256 * 24);
    // row buffer of the strip kernel: KP_PROT_ROWBUF_FIELDS ints per column of the database protein, one region per
    // block, and 64 ints for its work counter (kp_prot.hip)
    const size_t scratch_per_block = (size_t)KP_PROT_ROWBUF_FIELDS * ((size_t)T.max_db_prot_len + 1);
    KP_HIP_CHECK(ctx, R.d_dp_scratch.reserve(scratch_per_block * (size_t)n_blocks + 64));
    kp_launch_protein(R.d_prot.p, q_off, q_len, T.d_prot_db.p, t_off, t_len, (int32_t)slots, n_pairs, ctx->d_blosum.p,
                      R.d_dp.p, R.d_dp_scratch.p, scratch_per_block, n_blocks, R.stream, R.aux, R.ev_fork,
                      R.ev_join);
    kp_launch_states(b->view, T.typing, R.prm, R.d_kept.p, R.kept_cap, R.d_summary.p, R.d_dp.p, pair_base,
                     R.stream);
    KP_HIP_CHECK(ctx, hipGetLastError());
    return KP_OK;
}

int kp_batch_reduce(kp_ctx *ctx, kp_batch *b, const int32_t *best_locus, const kp_typing_params *prm) {
    if (!ctx || !b || b->ctx != ctx || !prm || (b->n_asm > 0 && !best_locus)) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpTypingGroup *Tp = typing_group(ctx, b);
    if (!Tp) return kp_fail(ctx, KP_ESTATE, "kp_db_load_typing has not been called");
    KpWork *w = finalised_work(ctx, b);
    if (!w) return KP_ESTATE;
    KpTypingRun &R = typing_run(w, b->group);
    if (!R.scored) return kp_fail(ctx, KP_ESTATE, "kp_batch_score has not been called");
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (int a = 0; a < b->n_asm; ++a)
        if (best_locus[a] < 0 || best_locus[a] >= Tp->typing.n_loci) return kp_fail(ctx, KP_EINVAL, "best_locus out of range");
    R.prm = *prm;
    int rc;
    if (ctx->opt.readback_copy_engine) {
        rc = upload(ctx, R.d_best, best_locus, (size_t)b->n_asm, R.stream);
    } else {  // through the landing area and a kernel, like the read-backs: a copy-engine upload queues behind the shard in flight
        Fetch f(ctx, R.stream);
        const size_t bytes = (size_t)b->n_asm * sizeof(int32_t);
        if ((rc = f.begin(bytes))) return rc;
        KP_HIP_CHECK(ctx, R.d_best.reserve((size_t)b->n_asm));
        if (bytes) {
            std::memcpy(ctx->bounce, best_locus, bytes);
            kp_launch_read_back(ctx->bounce, R.d_best.p, bytes, R.stream);
        }
    }
    if (rc == KP_OK && hipStreamSynchronize(R.stream) != hipSuccess) rc = kp_fail(ctx, KP_EHIP, "H2D best loci failed");
    if (rc) return rc;
    rc = enqueue_reduce(ctx, b, w);
    if (rc) return rc;
    R.reduced = true;
    R.sums_valid = false;
    return KP_OK;
}

// waits for the reduction, re-runs it with larger buffers while any assembly overflowed one, and keeps the summaries
static int fetch_summaries(kp_ctx *ctx, kp_batch *b, KpWork *w) {
    KpTypingRun &R = typing_run(w, b->group);
    if (R.sums_valid) return KP_OK;
    const size_t n_asm = (size_t)b->n_asm;
    R.h_sums.resize(n_asm);
    for (int attempt = 0;; ++attempt) {
        {
            Fetch f(ctx, R.stream);
            int frc;
            if ((frc = f.begin(n_asm * sizeof(KpAsmSummary))) || (frc = f.add(R.h_sums.data(), R.d_summary.p, n_asm * sizeof(KpAsmSummary))) ||
                (frc = f.finish()))
                return frc;
        }
        int flags = 0;
        for (const auto &s : R.h_sums) flags |= s.overflow;
        if (flags & 4) return kp_fail(ctx, KP_EINVAL, "a locus has more genes than KP_MAX_LOCUS_GENES");
        if (!(flags & (1 | 2 | 8))) break;
        if (attempt >= 8) return kp_fail(ctx, KP_EOVERFLOW, "reduction buffers overflowed repeatedly");
        kp_ctx::RunCaps &caps = run_caps(ctx, b->group);
        if (flags & 1) {
            if (caps.kept_cap >= 2048) return kp_fail(ctx, KP_EOVERFLOW, "more than 2048 non-overlapping hits in one assembly");
            caps.kept_cap = std::min(caps.kept_cap * 4, 2048);
        }
        if (flags & 2) caps.piece_cap *= 4;
        if (flags & 8) caps.prot_cap *= 4;
        w->stats[4] += 1;
        int rc = enqueue_reduce(ctx, b, w);
        if (rc) return rc;
    }
    R.max_kept = 1; R.max_pieces = 1;
    for (const auto &s : R.h_sums) {
        R.max_kept = std::max(R.max_kept, s.n_kept);
        R.max_pieces = std::max(R.max_pieces, s.n_pieces);
    }
    R.sums_valid = true;
    return KP_OK;
}

// the run of the batch's current group after kp_batch_reduce, or null after recording the error
static KpTypingRun *reduced_run(kp_ctx *ctx, kp_batch *b, KpWork **w_out) {
    KpWork *w = finalised_work(ctx, b);
    if (!w) return nullptr;
    KpTypingRun &R = typing_run(w, b->group);
    if (!R.reduced) { kp_fail(ctx, KP_ESTATE, "kp_batch_reduce has not been called"); return nullptr; }
    *w_out = w;
    return &R;
}

int kp_batch_typing(kp_ctx *ctx, kp_batch *b, kp_asm_summary *summaries, kp_kept *kept, int32_t kept_stride,
                    kp_piece *pieces, int32_t piece_stride) {
    if (!ctx || !b || b->ctx != ctx || (b->n_asm > 0 && (!summaries || !kept || !pieces)))
        return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = nullptr;
    KpTypingRun *Rp = reduced_run(ctx, b, &w);
    if (!Rp) return KP_ESTATE;
    KpTypingRun &R = *Rp;
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const size_t n_asm = (size_t)b->n_asm;
    int rc = fetch_summaries(ctx, b, w);
    if (rc) return rc;
    if (n_asm == 0) return KP_OK;
    if (kept_stride < R.max_kept || piece_stride < R.max_pieces)
        return kp_fail(ctx, KP_EINVAL, "output strides too small (see kp_batch_typing_caps)");
    std::memcpy(summaries, R.h_sums.data(), n_asm * sizeof(KpAsmSummary));
    // rows of the device buffers are kept_cap / piece_cap records long; only the first `stride` records of each are
    // wanted (a batch keeps a few dozen hits per assembly, the buffers leave room for hundreds): packed on the device,
    // then one linear copy each
    const size_t kw = (size_t)std::min(kept_stride, R.kept_cap) * sizeof(KpKept) / 4;
    const size_t pw = (size_t)std::min(piece_stride, R.piece_cap) * sizeof(KpPiece) / 4;
    {   // sized from the most any batch of this typing group has needed (plus a quarter), whichever work set it ran on: the
        // fullest assembly of a batch decides the strides, and three work sets learning that one by one re-allocate for steps
        size_t &hw = run_caps(ctx, b->group).pack_items;
        const size_t need = n_asm * ((size_t)kept_stride * sizeof(KpKept) + (size_t)piece_stride * sizeof(KpPiece)) / 4;
        if (need > hw) hw = need + need / 4;
        KP_HIP_CHECK(ctx, R.d_pack.reserve(hw));
    }
    uint32_t *pk = R.d_pack.p, *pp = pk + n_asm * (size_t)kept_stride * sizeof(KpKept) / 4;
    kp_launch_pack_rows(reinterpret_cast<const uint32_t *>(R.d_kept.p), (size_t)R.kept_cap * sizeof(KpKept) / 4, pk,
                        (size_t)kept_stride * sizeof(KpKept) / 4, kw, (int)n_asm, R.stream);
    kp_launch_pack_rows(reinterpret_cast<const uint32_t *>(R.d_pieces.p), (size_t)R.piece_cap * sizeof(KpPiece) / 4, pp,
                        (size_t)piece_stride * sizeof(KpPiece) / 4, pw, (int)n_asm, R.stream);
    {
        Fetch f(ctx, R.stream);
        int frc;
        const size_t kb = n_asm * (size_t)kept_stride * sizeof(KpKept), pb = n_asm * (size_t)piece_stride * sizeof(KpPiece);
        if ((frc = f.begin(kb + pb)) || (frc = f.add(kept, pk, kb)) || (frc = f.add(pieces, pp, pb)) || (frc = f.finish())) return frc;
    }
    // identity sums use numpy's float32 association; a few dozen adds per assembly, done here on the copied rows
    std::vector<float> vals;
    for (size_t a = 0; a < n_asm; ++a) {
        vals.clear();
        const KpKept *k = kept + a * (size_t)kept_stride;
        for (int i = 0; i < summaries[a].n_kept; ++i)
            if (!(k[i].flags & KP_F_SPURIOUS) && k[i].state == KP_STATE_NORMAL) vals.push_back(k[i].pident);
        summaries[a].n_normal = (int32_t)vals.size();
        summaries[a].ident_sum = kp_np_sum_f32(vals.data(), (int)vals.size());
    }
    return KP_OK;
}

int kp_batch_typing_caps(kp_ctx *ctx, kp_batch *b, int32_t *kept_cap, int32_t *piece_cap) {
    if (!ctx || !b || b->ctx != ctx || !kept_cap || !piece_cap) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = nullptr;
    KpTypingRun *Rp = reduced_run(ctx, b, &w);
    if (!Rp) return KP_ESTATE;
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int rc = fetch_summaries(ctx, b, w);
    if (rc) return rc;
    *kept_cap = Rp->max_kept;
    *piece_cap = Rp->max_pieces;
    return KP_OK;
}

int kp_batch_proteins(kp_ctx *ctx, kp_batch *b, int32_t asm_index, uint8_t *out, int64_t cap) {
    if (!ctx || !b || b->ctx != ctx || asm_index < 0 || asm_index >= b->n_asm || (!out && cap > 0))
        return kp_fail(ctx, KP_EINVAL, "bad arguments");
    KpWork *w = nullptr;
    KpTypingRun *Rp = reduced_run(ctx, b, &w);
    if (!Rp) return KP_ESTATE;
    KpTypingRun &R = *Rp;
    const int64_t n = std::min<int64_t>(cap, R.prot_cap);
    if (n > 0 && hipMemcpy(out, R.d_prot.p + (size_t)asm_index * (size_t)R.prot_cap, (size_t)n, hipMemcpyDeviceToHost) != hipSuccess)
        return kp_fail(ctx, KP_EHIP, "D2H proteins failed");
    return (int)n;
}

static int protein_align(kp_ctx *ctx, const uint8_t *q, const int32_t *q_off, const int32_t *q_len, const uint8_t *t,
                         const int32_t *t_off, const int32_t *t_len, int32_t n, const int32_t *seed_off, int32_t seed_k,
                         int32_t *out8) {
    if (!ctx) return kp_fail(nullptr, KP_EINVAL, "null context");
    if (n < 0 || (n > 0 && (!q_off || !q_len || !t_off || !t_len || !out8))) return kp_fail(ctx, KP_EINVAL, "bad arguments");
    if (n == 0) return KP_OK;
    KP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    size_t q_bytes = 0, t_bytes = 0;
    int max_t_len = 0;
    for (int i = 0; i < n; ++i) {
        if (q_len[i] < 0 || t_len[i] < 0 || q_off[i] < 0 || t_off[i] < 0 || q_len[i] > 65535 || t_len[i] > 65535)
            return kp_fail(ctx, KP_EINVAL, "protein lengths must be within [0, 65535]");
        q_bytes = std::max(q_bytes, (size_t)q_off[i] + (size_t)q_len[i]);
        t_bytes = std::max(t_bytes, (size_t)t_off[i] + (size_t)t_len[i]);
        max_t_len = std::max(max_t_len, t_len[i]);
    }
    if ((q_bytes && !q) || (t_bytes && !t)) return kp_fail(ctx, KP_EINVAL, "null sequence data");
    const int n_blocks = std::max(1, std::min(n, 256 * 16));
    const size_t scratch_per_block = (size_t)KP_PROT_ROWBUF_FIELDS * ((size_t)max_t_len + 1);  // see kp_prot.hip
    std::vector<int32_t> meta(5 * (size_t)n, 0);
    std::memcpy(meta.data(), q_off, (size_t)n * 4);
    std::memcpy(meta.data() + n, q_len, (size_t)n * 4);
    std::memcpy(meta.data() + 2 * (size_t)n, t_off, (size_t)n * 4);
    std::memcpy(meta.data() + 3 * (size_t)n, t_len, (size_t)n * 4);
    if (seed_off) std::memcpy(meta.data() + 4 * (size_t)n, seed_off, (size_t)n * 4);
    int rc;
    if ((rc = upload(ctx, ctx->d_pq, q, q_bytes))) return rc;
    if ((rc = upload(ctx, ctx->d_pt, t, t_bytes))) return rc;
    if ((rc = upload(ctx, ctx->d_pmeta, meta.data(), meta.size()))) return rc;
    KP_HIP_CHECK(ctx, ctx->d_pout.reserve(8 * (size_t)n));
    KP_HIP_CHECK(ctx, ctx->d_pscratch.reserve(scratch_per_block * (size_t)n_blocks + 64));
    kp_launch_protein(ctx->d_pq.p, ctx->d_pmeta.p, ctx->d_pmeta.p + n, ctx->d_pt.p, ctx->d_pmeta.p + 2 * (size_t)n,
                      ctx->d_pmeta.p + 3 * (size_t)n, n, nullptr, ctx->d_blosum.p, ctx->d_pout.p, ctx->d_pscratch.p,
                      scratch_per_block, n_blocks, ctx->stream, nullptr, nullptr, nullptr,
                      seed_off ? ctx->d_pmeta.p + 4 * (size_t)n : nullptr, seed_k);
    KP_HIP_CHECK(ctx, hipGetLastError());
    KP_HIP_CHECK(ctx, hipMemcpyAsync(out8, ctx->d_pout.p, 8 * (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost,
                                     ctx->stream));
    KP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return KP_OK;
}

int kp_protein_align(kp_ctx *ctx, const uint8_t *q, const int32_t *q_off, const int32_t *q_len, const uint8_t *t,
                     const int32_t *t_off, const int32_t *t_len, int32_t n, int32_t *out8) {
    return protein_align(ctx, q, q_off, q_len, t, t_off, t_len, n, nullptr, 0, out8);
}

int kp_protein_align_seeded(kp_ctx *ctx, const uint8_t *q, const int32_t *q_off, const int32_t *q_len, const uint8_t *t,
                            const int32_t *t_off, const int32_t *t_len, int32_t n, const int32_t *diagonal_offsets,
                            int32_t k, int32_t *out8) {
    if (ctx && n > 0 && (!diagonal_offsets || k < 0 || k > KP_MAX_GENE_LEN)) return kp_fail(ctx, KP_EINVAL, "bad seed arguments");
    return protein_align(ctx, q, q_off, q_len, t, t_off, t_len, n, diagonal_offsets, k, out8);
}

}  // extern "C"
