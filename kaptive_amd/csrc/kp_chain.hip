// kp_chain.hip -- sorted anchors -> band tasks (the "chaining" step of the aligner, include/kp_spec.h).
//
// Stands in for the chaining stage inside rammappy's map_batch (reference call site
// src/kaptive/serotyping/core.py:154).  Anchors of an assembly arrive sorted by (gene*2+strand, diagonal, query pos),
// as compact keys (KpKeyBits, kp_internal.h).
// A run is a maximal stretch without a hard break (new gene/strand, new contig, diagonal jump > KP_DIAG_GAP); a run is
// cut greedily into clusters whenever it would span more than KP_MAX_SPREAD diagonals; every cluster with enough
// anchors becomes one task in the list of its band-width class.
//
// One wave per slice of an assembly's anchor list, 64 anchors per round: every lane takes one anchor (contig by binary
// search, hard-break flag against its predecessor), the breaks are a ballot, and the pieces between breaks are merged
// into the cluster the wave carries in uniform registers (first / last diagonal by readlane -- diagonals ascend within
// a run --, query extent by a wave reduction).  Only a piece that would stretch the cluster past KP_MAX_SPREAD is walked
// anchor by anchor.  A wave owns the gene/strand groups that START inside its slice: it skips the rest of the group its
// slice starts in and keeps going past its own end until its last group is over, because the clusters of a gene/strand
// have to meet in one place for the joins of kp-align v4 (open sequences of clusters within KP_JOIN_BW diagonals,
// JoinWave below; kp_join.hip chains and aligns them).
#include "kp_internal.h"
#include "kp_sketch.h"

namespace {

#ifndef KP_CHAIN_SLICES
#define KP_CHAIN_SLICES 16
#endif
#ifndef KP_CHAIN_WAVES
#define KP_CHAIN_WAVES 4
#endif
constexpr int CHAIN_SLICES = KP_CHAIN_SLICES;  // waves per assembly
constexpr int CHAIN_WAVES = KP_CHAIN_WAVES;  // ... of which a block holds this many: they share one task stage, so that its 22 KB of LDS
                                  // do not keep the CUs at a fraction of the waves they could hold (the kernel is a chain of memory trips)
static_assert(CHAIN_SLICES % CHAIN_WAVES == 0, "whole blocks");

// Tasks are staged per block in LDS and appended to the global lists with one atomic per block and class: a batch
// produces ~10^6 tasks for four counters, which would otherwise serialise on those words.  One stage for all classes
// (an entry remembers its rank within its class): with a stage per class the wide classes had room for 32 tasks a block,
// which a workload rich in wide bands -- diverged relatives with indels, `bench.py --background paralog` -- overflowed
// into 0.8 M single appends per batch (kp_chain_kernel 4.2 ms instead of 0.7).
#ifndef KP_CHAIN_STAGE
#define KP_CHAIN_STAGE 640
#endif
constexpr int STAGE = KP_CHAIN_STAGE;  // staged tasks per block (an assembly's ~1200-2100 tasks come from four blocks)

struct TaskStage {
    KpTask t[STAGE];
    uint16_t rank[STAGE];  // ... of the entry among the staged tasks of its class
    uint32_t total;        // entries asked for (may exceed STAGE: the excess was appended directly)
    uint32_t n[KP_N_CLASSES], base[KP_N_CLASSES];
};

__constant__ uint8_t c_chain_pen[KP_CHAIN_PEN_SIZE] = KP_CHAIN_PEN_TABLE;

// minimap2's chaining of a small cluster (kp_spec.h), by one lane: keys[0..n) are the cluster's anchors (compact keys,
// sorted by diagonal then query position), n <= KP_CHAIN_DP_MAX.  Returns the chain's score, *cnt = its anchor count.
// The lane's working arrays live in LDS, slot-major (`at(i)` = slot i of this lane: consecutive lanes hit consecutive
// banks): with private arrays every dynamically indexed access was a trip through scratch memory (1.44 ms per 1000
// assemblies for what is a few dozen operations per cluster).
constexpr int CS_THREADS = 64;
struct ChainScratch {
    int32_t t[KP_CHAIN_DP_MAX][CS_THREADS], q[KP_CHAIN_DP_MAX][CS_THREADS];
    int16_t f[KP_CHAIN_DP_MAX][CS_THREADS];
    int8_t p[KP_CHAIN_DP_MAX][CS_THREADS];
};

__device__ __forceinline__ int chain_small(const uint64_t *__restrict__ keys, int n, KpKeyBits kb, ChainScratch &cs, int me, int *cnt) {
    for (int i = 0; i < n; ++i) {  // (independent loads: all of the cluster's keys are in flight together)
        const uint64_t key = keys[i];
        const int32_t qi = (int32_t)kp_ckey_qpos(key, kb);
        cs.q[i][me] = qi;
        cs.t[i][me] = (int32_t)kp_ckey_diag(key, kb) - KP_DIAG_BIAS + qi;
    }
    for (int i = 1; i < n; ++i) {  // insertion sort by (target, query); a cluster on one diagonal arrives sorted
        const int32_t ti = cs.t[i][me], qi = cs.q[i][me];
        int j = i;
        while (j > 0) {
            const int32_t tp = cs.t[j - 1][me], qp = cs.q[j - 1][me];
            if (!(tp > ti || (tp == ti && qp > qi))) break;
            cs.t[j][me] = tp; cs.q[j][me] = qp;
            --j;
        }
        if (j != i) { cs.t[j][me] = ti; cs.q[j][me] = qi; }
    }
    int best = 0, f_best = 0;
    for (int i = 0; i < n; ++i) {
        const int32_t ti = cs.t[i][me], qi = cs.q[i][me];
        int max_f = KP_K, max_j = -1;
        for (int j = i - 1; j >= 0; --j) {
            const int dq = qi - cs.q[j][me], dr = ti - cs.t[j][me];
            if (dq <= 0 || dq > KP_CHAIN_MAX_DIST || dr == 0) continue;
            const int dd = dr > dq ? dr - dq : dq - dr, dg = dr < dq ? dr : dq;
            int sc = dg < KP_K ? dg : KP_K;
            if (dd || dg > KP_K) sc -= c_chain_pen[dd < KP_CHAIN_PEN_SIZE ? dd : KP_CHAIN_PEN_SIZE - 1];
            sc += cs.f[j][me];
            if (sc > max_f) { max_f = sc; max_j = j; }
        }
        cs.f[i][me] = (int16_t)max_f; cs.p[i][me] = (int8_t)max_j;
        if (max_f >= f_best) { best = i; f_best = max_f; }  // the largest f, the later anchor on ties
    }
    int i = best, max_s = 0, steps = 0, cut = 0;
    do {  // walk back; the chain is cut where the score counted from its end peaks
        i = cs.p[i][me];
        ++steps;
        const int sc = i < 0 ? f_best : f_best - cs.f[i][me];
        if (sc > max_s) { max_s = sc; cut = steps; }
    } while (i >= 0);
    *cnt = cut;
    return max_s;
}

// ---- kp-align v4: groups of provisional clusters (kp_spec.h) --------------------------------------------------------------------
// Lane 0 of a wave meets its clusters in the order of the sorted anchors, so it can tell which provisional clusters of one
// gene/strand and contig follow each other within KP_JOIN_BW diagonals.  The last cluster of every OPEN sequence (one per
// contig, at most KP_JOIN_OPEN per gene/strand) is held back until a later cluster shows whether the sequence goes on: a lone
// cluster then goes to the block's stage as before, a member of a group is appended to the global list at once -- its slot is
// what the group record refers to -- and the finished group record (tasks, anchor ranges) goes to the group list.  Nearly
// always one sequence is open: its entry lives in lane 0's registers (a chain of LDS round trips per cluster cost 0.2 ms per
// pass); further ones -- the clusters of a gene that a contig boundary cuts interleave by diagonal -- wait in LDS.
struct PendEntry {
    KpTask t;
    uint32_t first, cnt, dmax;
    int in_group;
    int weak;  // (v5) too few anchors / query bases for a band task: a member of its group all the same, never a task
};
struct JoinLds {
    PendEntry e[KP_JOIN_OPEN];
    KpGroup g[KP_JOIN_OPEN];
};
struct JoinWave {
    PendEntry A;  // the open sequence touched last; its group record is L->g[a_slot]
    int a_valid, a_slot;
    uint32_t others;  // slots of L->e that hold further open sequences
    JoinLds *L;
};
struct GroupOut {
    KpGroup *groups;
    uint32_t *count;
    uint32_t cap;
};

__device__ __forceinline__ int class_of_width(int w) { return w == 16 ? 0 : (w == 32 ? 1 : (w == 64 ? 2 : 3)); }

__device__ __forceinline__ void emit_staged(const KpTask &t, KpTask *tasks, uint32_t *task_count, uint32_t task_cap, TaskStage &st) {
    const int cls = class_of_width(t.width);
    const uint32_t s = atomicAdd(&st.total, 1u);  // (the block's waves share the stage)
    if (s < (uint32_t)STAGE) {
        st.t[s] = t;
        st.rank[s] = (uint16_t)atomicAdd(&st.n[cls], 1u);
        return;
    }
    const uint32_t slot = atomicAdd(&task_count[cls], 1u);  // stage full: append directly
    if (slot < task_cap) tasks[(size_t)cls * task_cap + slot] = t;  // beyond cap: counted, not stored (host retries)
}

__device__ __forceinline__ void entry_to_group(const PendEntry &E, KpGroup &G, KpTask *tasks, uint32_t *task_count, uint32_t task_cap) {
    uint32_t ref = KP_REF_NONE;
    if (!E.weak) {
        const int cls = class_of_width(E.t.width);
        const uint32_t slot = atomicAdd(&task_count[cls], 1u);
        if (slot < task_cap) tasks[(size_t)cls * task_cap + slot] = E.t;
        ref = KP_TASK_REF(cls, slot);
    }
    const int n = G.n;
    G.task[n] = ref; G.first[n] = E.first; G.cnt[n] = E.cnt;
    G.n = n + 1;
}

// the sequence ends with E
__device__ __forceinline__ void close_entry(const PendEntry &E, KpGroup &G, KpTask *tasks, uint32_t *task_count, uint32_t task_cap,
                                            TaskStage &st, const GroupOut &go) {
    if (E.in_group) {
        entry_to_group(E, G, tasks, task_count, task_cap);
        bool any = false;  // a group needs a provisional cluster (kp_spec.h): weak clusters alone chain to nothing that is reported
        uint32_t total = 0;
        for (int i = 0; i < G.n; ++i) { any = any || G.task[i] != KP_REF_NONE; total += G.cnt[i]; }
        G.total = total;
        if (any) {
            const uint32_t g = atomicAdd(go.count, 1u);
            if (g < go.cap) go.groups[g] = G;  // beyond cap: counted, not stored (host retries)
        }
    } else if (!E.weak) {
        emit_staged(E.t, tasks, task_count, task_cap, st);
    }
}

__device__ __forceinline__ void pending_flush(JoinWave &J, KpTask *tasks, uint32_t *task_count, uint32_t task_cap, TaskStage &st,
                                              const GroupOut &go) {
    if (J.a_valid) close_entry(J.A, J.L->g[J.a_slot], tasks, task_count, task_cap, st, go);
    J.a_valid = 0;
    while (J.others) {
        const int s = __builtin_ctz(J.others);
        J.others &= J.others - 1;
        close_entry(J.L->e[s], J.L->g[s], tasks, task_count, task_cap, st, go);
    }
}

// called by one lane.  A cluster whose anchors cover fewer than KP_MIN_SEED_SPAN query bases cannot chain to
// KP_MIN_CHAIN_SCORE (a chain scores at most its query extent) and is dropped here; the rest become PROVISIONAL tasks
// -- n_anchors = the cluster's anchor count, chain_score = the index of its first anchor in the assembly's sorted list --
// which kp_chain_score_kernel settles (kp_spec.h: chain score and anchor count, or rejection).
__device__ __forceinline__ void flush_cluster(int a, uint32_t gs, int ctg, uint32_t d0, uint32_t dmax, uint32_t qmin,
                                              uint32_t qmax, int cnt, uint32_t first, KpTask *tasks, uint32_t *task_count,
                                              uint32_t task_cap, TaskStage &st, JoinWave &J, const GroupOut &go) {
    const int weak = cnt < KP_MIN_ANCHORS || (int)(qmax - qmin) + KP_K < KP_MIN_SEED_SPAN;
    int margin = KP_BAND_MARGIN_NARROW, need = (int)(dmax - d0) + 1 + 2 * KP_BAND_MARGIN_NARROW, w = 16;
    if (need > 16) {
        margin = KP_BAND_MARGIN;
        need = (int)(dmax - d0) + 1 + 2 * KP_BAND_MARGIN;
        w = need <= 32 ? 32 : (need <= 64 ? 64 : 128);
    }
    PendEntry c;
    c.t.asm_id = a; c.t.gs = (int32_t)gs; c.t.contig = ctg; c.t.width = w; c.t.n_anchors = cnt; c.t.chain_score = (int32_t)first;
    c.t.lo = (int32_t)d0 - KP_DIAG_BIAS - margin - (w - need) / 2;
    c.t.qspan = qmin | (qmax << 16);
    c.first = first; c.cnt = (uint32_t)cnt; c.dmax = dmax; c.in_group = 0; c.weak = weak;
    // (kp_spec.h, GROUPS) all open sequences belong to one gene/strand
    if (J.a_valid && J.A.t.gs != c.t.gs) pending_flush(J, tasks, task_count, task_cap, st, go);
    bool found = J.a_valid && J.A.t.contig == ctg;
    if (!found && J.others) {  // (rare: the gene has open sequences on other contigs)
        for (uint32_t m = J.others; m; m &= m - 1) {
            const int s = __builtin_ctz(m);
            if (J.L->e[s].t.contig != ctg) continue;
            J.L->e[J.a_slot] = J.A;  // A waits in its own slot, the contig's sequence comes to the registers
            J.others = (J.others | (1u << J.a_slot)) & ~(1u << s);
            J.A = J.L->e[s];
            J.a_slot = s;
            found = true;
            break;
        }
    }
    if (found) {
        KpGroup &G = J.L->g[J.a_slot];
        if (d0 - J.A.dmax <= (uint32_t)KP_JOIN_BW && (J.A.in_group ? G.n + 1 : 1) < KP_JOIN_GROUP_MAX) {
            if (!J.A.in_group) { G.n = 0; G.asm_id = a; G.gs = (int32_t)gs; G.contig = ctg; }
            entry_to_group(J.A, G, tasks, task_count, task_cap);
            c.in_group = 1;
        } else {
            close_entry(J.A, G, tasks, task_count, task_cap, st, go);
        }
    } else if (J.a_valid) {  // a further contig: A waits in LDS; when all slots are taken, the sequence that ends lowest closes
        J.L->e[J.a_slot] = J.A;
        J.others |= 1u << J.a_slot;
        if (J.others == (1u << KP_JOIN_OPEN) - 1u) {
            int ev = -1;
            for (uint32_t m = J.others; m; m &= m - 1) {
                const int s = __builtin_ctz(m);
                if (ev < 0 || J.L->e[s].dmax < J.L->e[ev].dmax || (J.L->e[s].dmax == J.L->e[ev].dmax && J.L->e[s].t.contig < J.L->e[ev].t.contig)) ev = s;
            }
            close_entry(J.L->e[ev], J.L->g[ev], tasks, task_count, task_cap, st, go);
            J.others &= ~(1u << ev);
        }
        J.a_slot = __builtin_ctz(~J.others);
    }
    J.A = c;
    J.a_valid = 1;
}

struct Cluster {  // wave-uniform
    bool open;
    uint32_t gs, d0, dprev, qmin, qmax;
    uint32_t first;  // index of its first anchor: a cluster is a contiguous piece of the sorted list
    int ctg, cnt;
};

__global__ __launch_bounds__(64 * CHAIN_WAVES) void kp_chain_kernel(KpBatchView b, const uint64_t *__restrict__ keys,
                                                      const uint32_t *__restrict__ count, uint32_t cap, KpKeyBits kb,
                                                      KpTask *__restrict__ tasks, uint32_t *__restrict__ task_count,
                                                      uint32_t task_cap, GroupOut go) {
    __shared__ TaskStage st;
    __shared__ JoinLds s_join[CHAIN_WAVES];
    JoinWave jw;
    jw.L = &s_join[threadIdx.x >> 6];
    const int a = blockIdx.y, lane = threadIdx.x & 63;
    uint32_t n = count[a];
    if (n > cap) n = cap;
    const uint32_t per = (((n + CHAIN_SLICES - 1) / CHAIN_SLICES) + 63u) & ~63u;  // whole rounds of 64 anchors
    const uint32_t lo = (blockIdx.x * CHAIN_WAVES + (threadIdx.x >> 6)) * per, hi = min(n, lo + per);
    const bool idle = lo >= n;  // (a slice past the list's end: the wave only takes part in the block's barriers)
    const uint64_t *k = keys + (size_t)a * cap;
    const int c0 = b.asm_first_ctg[a], nc = b.asm_first_ctg[a + 1] - c0;
    const int32_t *starts = b.ctg_start + c0;
    auto contig_of = [&](uint64_t key) {  // last contig starting at or before the anchor's target position
        const int32_t t = (int32_t)kp_ckey_diag(key, kb) - KP_DIAG_BIAS + (int32_t)kp_ckey_qpos(key, kb);
        int l = 0, h = nc;
        while (l < h) {
            const int mid = (l + h) >> 1;
            if (starts[mid] <= t) l = mid + 1; else h = mid;
        }
        return l - 1;
    };
    if (threadIdx.x < KP_N_CLASSES) st.n[threadIdx.x] = 0;
    if (threadIdx.x == 0) st.total = 0;
    jw.a_valid = 0; jw.a_slot = 0; jw.others = 0;
    jw.A.first = jw.A.cnt = jw.A.dmax = 0; jw.A.in_group = 0; jw.A.weak = 0;
    jw.A.t.asm_id = 0; jw.A.t.gs = 0; jw.A.t.contig = 0; jw.A.t.lo = 0; jw.A.t.width = 16; jw.A.t.n_anchors = 0; jw.A.t.qspan = 0; jw.A.t.chain_score = 0;
    __syncthreads();
    // A wave owns the gene/strand GROUPS OF ANCHORS that start in its slice (kp-align v4: the clusters of one gene/strand must
    // pass through one lane in order, see JoinWave): a slice that starts inside such a group leaves it to the wave before it
    // -- it skips ahead to the first anchor of another gene/strand -- and a wave keeps going past its slice's end until the
    // gene/strand changes.
    uint32_t start = lo;
    if (!idle && lo > 0) {
        const uint32_t g_prev = kp_ckey_gs(k[lo - 1], kb);
        for (;;) {
            const uint32_t i = start + lane;
            const unsigned long long other = __ballot(i < n && kp_ckey_gs(k[i], kb) != g_prev);
            if (other) { start += (uint32_t)__builtin_ctzll(other); break; }
            start += 64;
            if (start >= n) break;
        }
    }
    if (!idle && start < hi) {
    const uint32_t g_hi = kp_ckey_gs(k[hi - 1], kb);  // anchors from `hi` on are this wave's while they carry it
    uint64_t prev_key = 0;  // the anchor before the round's first one
    int prev_ctg = -1;
    bool have_prev = start > 0;
    if (have_prev) { prev_key = k[start - 1]; prev_ctg = contig_of(prev_key); }
    Cluster cur;
    cur.open = false; cur.gs = cur.d0 = cur.dprev = cur.qmin = cur.qmax = cur.first = 0; cur.ctg = 0; cur.cnt = 0;
    auto flush = [&]() {
        if (cur.open && lane == 0)
            flush_cluster(a, cur.gs, cur.ctg, cur.d0, cur.dprev, cur.qmin, cur.qmax, cur.cnt, cur.first, tasks, task_count, task_cap, st, jw, go);
        cur.open = false;
    };

    // [from, to) of the round joins the open cluster (same run: no hard break inside)
    auto merge = [&](int from, int to, uint32_t d, uint32_t q, uint32_t round_base) {
        const uint32_t d_last = __shfl(d, to - 1);
        if (d_last - cur.d0 <= KP_MAX_SPREAD) {  // the whole piece joins the cluster
            const bool mine = lane >= from && lane < to;
            uint32_t mn = mine ? q : 0xFFFFFFFFu, mx = mine ? q : 0u;
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                mn = min(mn, (uint32_t)__shfl_xor(mn, o));
                mx = max(mx, (uint32_t)__shfl_xor(mx, o));
            }
            cur.qmin = min(cur.qmin, mn); cur.qmax = max(cur.qmax, mx);
            cur.dprev = d_last;
            cur.cnt += to - from;
        } else {  // a soft cut falls inside the piece: anchor by anchor, exactly the greedy rule
            for (int u = from; u < to; ++u) {
                const uint32_t du = __shfl(d, u), qu = __shfl(q, u);
                if (du - cur.d0 > KP_MAX_SPREAD) {
                    flush();
                    cur.open = true;
                    cur.d0 = du; cur.qmin = cur.qmax = qu; cur.cnt = 0; cur.first = round_base + (uint32_t)u;
                }
                cur.dprev = du;
                cur.cnt++;
                cur.qmin = min(cur.qmin, qu); cur.qmax = max(cur.qmax, qu);
            }
        }
    };

    uint64_t next_key = start + lane < n ? k[start + lane] : 0ull;  // the following round's anchors are fetched a round ahead
    for (uint32_t w = start; w < n; w += 64) {
        const uint32_t i = w + lane;
        const uint64_t key = next_key;
        next_key = i + 64 < n ? k[i + 64] : 0ull;
        const uint32_t gs = kp_ckey_gs(key, kb), d = kp_ckey_diag(key, kb), q = kp_ckey_qpos(key, kb);
        // the round's anchors that are this wave's: a prefix (the list is sorted by gene/strand first)
        const unsigned long long foreign = __ballot(!(i < n && (i < hi || gs == g_hi)));
        const int n_valid = foreign ? (int)__builtin_ctzll(foreign) : 64;
        if (n_valid == 0) break;
        const bool valid = lane < n_valid;
        const int ctg = valid ? contig_of(key) : -1;
        uint64_t pk = ((uint64_t)__shfl_up((uint32_t)(key >> 32), 1) << 32) | __shfl_up((uint32_t)key, 1);
        int pc = __shfl_up(ctg, 1);
        bool has_pred = true;
        if (lane == 0) { pk = prev_key; pc = prev_ctg; has_pred = have_prev; }
        const bool head = valid && (!has_pred || kp_ckey_gs(pk, kb) != gs || pc != ctg || d - kp_ckey_diag(pk, kb) > KP_DIAG_GAP);
        const unsigned long long heads = __ballot(head);
        prev_key = __shfl(key, n_valid - 1);
        prev_ctg = __shfl(ctg, n_valid - 1);
        have_prev = true;
        const int first = heads ? (int)__builtin_ctzll(heads) : n_valid;  // anchors before it continue the open run
        if (first > 0 && cur.open) merge(0, first, d, q, w);
        if (heads) {
            flush();  // the first break ends whatever was open
            // Most runs are stray seeds (fewer than KP_MIN_ANCHORS anchors) that can never become a task: every head lane
            // measures its own run, and only runs that are long enough -- or reach the end of the round and may go on --
            // are visited.
            const unsigned long long later = lane < 63 ? heads & (~0ull << (lane + 1)) : 0ull;
            const int my_end = later ? min(n_valid, (int)__builtin_ctzll(later)) : n_valid;
            // (v5) ... or lie within KP_JOIN_BW diagonals of the run before or after them: weak clusters are members of groups.  A
            // stray run further than that from both neighbours can neither join a sequence nor keep one open.
            const unsigned long long linked = __ballot(head && has_pred && kp_ckey_gs(pk, kb) == gs && d - kp_ckey_diag(pk, kb) <= (uint32_t)KP_JOIN_BW);
            const bool near = ((linked >> lane) & 1ull) || (my_end < n_valid && ((linked >> my_end) & 1ull));
            unsigned long long todo = __ballot(head && (my_end - lane >= KP_MIN_ANCHORS || my_end == n_valid || near));
            while (todo) {
                const int pos = (int)__builtin_ctzll(todo);
                todo &= todo - 1;
                const int end = __shfl(my_end, pos);
                cur.open = true;
                cur.gs = __shfl(gs, pos); cur.ctg = __shfl(ctg, pos);
                cur.d0 = cur.dprev = __shfl(d, pos);
                cur.qmin = cur.qmax = __shfl(q, pos);
                cur.cnt = 1; cur.first = w + (uint32_t)pos;
                if (pos + 1 < end) merge(pos + 1, end, d, q, w);
                if (end < n_valid) flush();  // the run ends inside the round; otherwise it stays open for the next one
            }
        }
        if (n_valid < 64) break;  // the wave's last anchors
    }
    flush();
    if (lane == 0) pending_flush(jw, tasks, task_count, task_cap, st, go);
    }
    __syncthreads();
    if (threadIdx.x < KP_N_CLASSES) st.base[threadIdx.x] = st.n[threadIdx.x] ? atomicAdd(&task_count[threadIdx.x], st.n[threadIdx.x]) : 0u;
    __syncthreads();
    const uint32_t staged = min(st.total, (uint32_t)STAGE);
    for (uint32_t i = threadIdx.x; i < staged; i += 64 * CHAIN_WAVES) {
        const int cls = class_of_width(st.t[i].width);
        const uint32_t slot = st.base[cls] + st.rank[i];
        if (slot < task_cap) tasks[(size_t)cls * task_cap + slot] = st.t[i];
    }
}

// ---- occurrence cut (kp_spec.h, KP_MID_OCC): a gene seed with more than ten anchors in an assembly loses them all ----------------
// The anchors of gene g -- both strands: gs = 2g and 2g + 1 -- are one stretch of the assembly's sorted list, and the anchors of
// one of its seeds are those with the same position on the gene's forward strand.  One block per assembly, a wave per slice
// of the list (slices begin and end where the gene changes); stretches that pass the certificate below -- nearly all on the
// headline workload -- are only measured; the others are counted in LDS, a counter per position of the gene's forward strand,
// a window of OCC_BINS positions at a time (one or two windows for a Kaptive-sized gene), and the counters they touched are
// cleared again.  Every wave has a counter table of its own: with one table per block behind a lock (round 5's first version)
// the exact counts of a block took turns, and an assembly with diverged relatives of database genes -- their anchors lie on
// more than ten diagonals per gene, the certificate fails for thousands of genes -- ran this kernel in 0.75 ms instead of 0.13.
// Anchors to drop become tombstones and the list is compacted at the end, which almost never happens.
// (Eight waves a block, 33 KB of tables: four blocks a CU as two of sixteen waves were, but a block now finds room beside the
// band fill's blocks of the passes in flight -- twelve a CU at 9 KB each leave 52 KB --: +0.6 % on the overlapped step.)
#ifndef KP_OCC_WAVES
#define KP_OCC_WAVES 8
#endif
#ifndef KP_OCC_BINS
#define KP_OCC_BINS 1024
#endif
constexpr int OCC_WAVES = KP_OCC_WAVES, OCC_BINS = KP_OCC_BINS;
constexpr uint64_t OCC_TOMB = ~0ull;

// ---- minimap2's mid_occ of the assemblies that need it (kp_spec.h, OCCURRENCE CUT) ------------------------------------------------
// The assembly's minimizers are not kept anywhere (the scan probes them against the genes' filter and forgets them), so an
// assembly in which a gene has KP_MIN_ANCHORS seeds beyond the floor is sketched once more: the first launch of
// kp_occ_cut_kernel (phase 0) claims it one of `n_slots` counting tables in global scratch and clears the table;
// kp_occ_sketch_kernel -- OCC_PARTS blocks per table -- runs kp_spec.h's state machine (kp_sketch.h) over chunks of OCC_CHUNK
// positions per thread, every chunk after a warm-up from a fresh state as kp_edge_kernel's right flanks, and counts every seed
// value in the open-addressing table; kp_occ_quantile_kernel builds the histogram of the counts and reads the quantile off it;
// the second launch of kp_occ_cut_kernel (phase 1) cuts at it.  Assemblies that need this are rare (a stretch of a gene in
// more than ten copies); the launches find nothing to do otherwise.  When more assemblies of a batch ask than there are tables
// the demand is counted and the host grows the scratch and reruns the pass.
struct OccScratch {
    uint32_t *keys;       // [n_slots << log2_size] seed values, 0xFFFFFFFF = empty
    uint32_t *cnts;       // [n_slots << log2_size]
    uint32_t *state;      // [n_asm] bit 0: a seed beyond the floor, bit 1: its own mid_occ decides; [n_asm] mid_occ; [n_slots] the table's assembly;
                          // [n_slots][KP_MID_OCC_HIST + 2] the quantile kernel's histogram, distinct values and blocks done (kp_occ_state_words)
    unsigned long long *demand;  // assemblies of the pass that asked for a table
    uint32_t n_slots, log2_size;
    int32_t n_asm;
};
constexpr int OCC_CHUNK = 256;  // (5 Mbp = 19 500 chunks for the 32 768 threads of a table's blocks: with 512 and 64 blocks a thread ran 640 steps, 1.4 ms)
constexpr int OCC_WARM = 48 + 2 * (KP_K + KP_W);
constexpr int OCC_PARTS = 128;

__global__ __launch_bounds__(256) void kp_occ_sketch_kernel(KpBatchView b, OccScratch sc) {
    const uint32_t slot = blockIdx.y;
    unsigned long long want = *sc.demand;
    if (slot >= want || slot >= sc.n_slots) return;
    const int a = (int)sc.state[2 * (size_t)sc.n_asm + slot];
    const uint32_t mask = (1u << sc.log2_size) - 1u;
    uint32_t *tk = sc.keys + ((size_t)slot << sc.log2_size), *tc = sc.cnts + ((size_t)slot << sc.log2_size);
    const uint32_t *aw = b.words + b.asm_word_off[a];
    const int c0 = b.asm_first_ctg[a], nc = b.asm_first_ctg[a + 1] - c0;
    const int r0 = b.asm_first_nrun[a], nr = b.asm_first_nrun[a + 1] - r0;
    const int32_t *runs = b.n_runs + 2 * (size_t)r0;
    // chunks of OCC_CHUNK positions, numbered through the assembly's padded space (contigs start on 32-base boundaries, the
    // padding between them is not sketched): every thread of the table's blocks takes every n-th chunk, whatever contig it lies in
    const int64_t total = (int64_t)(b.asm_word_off[a + 1] - b.asm_word_off[a]) << 4;
    const int32_t *cstarts = b.ctg_start + c0;
    const int64_t nthr = (int64_t)gridDim.x * blockDim.x, tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int64_t s_pad = tid * OCC_CHUNK; s_pad < total; s_pad += nthr * OCC_CHUNK) {
        int lo = 0, hi = nc;  // the contigs that overlap [s_pad, s_pad + OCC_CHUNK): from the last one that starts at or before s_pad
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cstarts[mid] <= s_pad) lo = mid + 1; else hi = mid; }
        for (int c = max(lo - 1, 0); c < nc && cstarts[c] < s_pad + OCC_CHUNK; ++c) {
            const int64_t cs = cstarts[c], ce = cs + b.ctg_len[c0 + c];
            const int64_t s0 = max(cs, s_pad), e0 = min(ce, s_pad + OCC_CHUNK);  // seeds whose 15-mer starts in [s0, e0) are this chunk's
            if (s0 >= e0) continue;
            const int64_t from = max(cs, s0 - OCC_WARM), to = min(ce, e0 + KP_K + KP_W + 1);  // (a seed is emitted at most K + W steps after its start)
            int ri = 0;  // first N run that ends after `from`
            {
                int l2 = 0, h2 = nr;
                while (l2 < h2) { const int mid = (l2 + h2) >> 1; if (runs[2 * mid + 1] <= from) l2 = mid + 1; else h2 = mid; }
                ri = l2;
            }
            auto emit = [&](int64_t t, uint32_t z, uint32_t x) {
                (void)z;
                if (t < s0 || t >= e0) return;
                uint32_t at = (x * 2654435769u) >> (32 - sc.log2_size);
                for (;;) {
                    const uint32_t old = atomicCAS(&tk[at], 0xFFFFFFFFu, x);
                    if (old == 0xFFFFFFFFu || old == x) { atomicAdd(&tc[at], 1u); break; }
                    at = (at + 1) & mask;
                }
            };
            KpSketchState st;
            kp_sketch_reset(st);
            uint32_t word = aw[from >> 4];
            for (int64_t i = from; i < to; ++i) {
                if ((i & 15) == 0) word = aw[i >> 4];  // (one load per sixteen steps)
                while (ri < nr && runs[2 * ri + 1] <= i) ++ri;
                const bool is_n = ri < nr && runs[2 * ri] <= i;
                const uint32_t code = is_n ? 4u : ((word >> (2 * (i & 15))) & 3u);
                kp_sketch_step(st, i, code, emit);
            }
            if (to == ce) kp_sketch_final(st, ce - 1, emit);
        }
    }
}

constexpr int OCC_QPARTS = 32;  // blocks per table in the quantile kernel
// words per table behind the state arrays: the histogram the table's blocks add up, the number of distinct values, blocks done
constexpr size_t OCC_QWORDS = (size_t)KP_MID_OCC_HIST + 2;
__device__ __forceinline__ uint32_t *occ_qwords(const OccScratch &sc, uint32_t slot) {
    return sc.state + 2 * (size_t)sc.n_asm + sc.n_slots + (size_t)slot * OCC_QWORDS;
}

__global__ __launch_bounds__(1024) void kp_occ_quantile_kernel(OccScratch sc) {
    __shared__ uint32_t s_hist[KP_MID_OCC_HIST];
    __shared__ uint32_t s_distinct, s_last;
    const uint32_t slot = blockIdx.y;
    unsigned long long want = *sc.demand;
    if (slot >= want || slot >= sc.n_slots) return;
    const int a = (int)sc.state[2 * (size_t)sc.n_asm + slot];
    const uint32_t size = 1u << sc.log2_size;
    const uint32_t *tc = sc.cnts + ((size_t)slot << sc.log2_size);
    uint32_t *gq = occ_qwords(sc, slot);  // (zeroed by the block that claimed the table)
    for (int i = threadIdx.x; i < KP_MID_OCC_HIST; i += blockDim.x) s_hist[i] = 0;
    if (threadIdx.x == 0) { s_distinct = 0; s_last = 0; }
    __syncthreads();
    // (an empty slot has the count 0: the claiming block cleared both arrays.  Nearly every minimizer of an assembly occurs once
    // or twice: those are counted in registers -- with an LDS atomic per slot a block's 1024 threads queued up at two bins, 2.0 ms
    // a table in one block --, the counts come four to a load, and OCC_QPARTS blocks share a table: 0.1 ms.)
    uint32_t mine = 0, n1 = 0, n2 = 0;
    auto take = [&](uint32_t c) {
        if (c == 0) return;
        ++mine;
        if (c == 1) ++n1;
        else if (c == 2) ++n2;
        else atomicAdd(&s_hist[c < (uint32_t)KP_MID_OCC_HIST - 1u ? c : (uint32_t)KP_MID_OCC_HIST - 1u], 1u);
    };
    if (size >= 4) {
        const uint4 *tc4 = reinterpret_cast<const uint4 *>(tc);  // (tables are 2^log2_size entries apart: 16-byte aligned)
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < size / 4; i += gridDim.x * blockDim.x) {
            const uint4 v = tc4[i];
            take(v.x); take(v.y); take(v.z); take(v.w);
        }
    } else if (blockIdx.x == 0) {
        for (uint32_t i = threadIdx.x; i < size; i += blockDim.x) take(tc[i]);
    }
    if (n1) atomicAdd(&s_hist[1], n1);
    if (n2) atomicAdd(&s_hist[2], n2);
    if (mine) atomicAdd(&s_distinct, mine);
    __syncthreads();
    for (int i = threadIdx.x; i < KP_MID_OCC_HIST; i += blockDim.x)
        if (s_hist[i]) atomicAdd(&gq[i], s_hist[i]);
    if (threadIdx.x == 0 && s_distinct) atomicAdd(&gq[KP_MID_OCC_HIST], s_distinct);
    __threadfence();  // (device scope: the table's blocks run on any XCD)
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&gq[KP_MID_OCC_HIST + 1], 1u) == gridDim.x - 1 ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (int i = threadIdx.x; i < KP_MID_OCC_HIST; i += blockDim.x) s_hist[i] = __hip_atomic_load(&gq[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n = __hip_atomic_load(&gq[KP_MID_OCC_HIST], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t mid = KP_MID_OCC;
        if (n > 0) {
            uint32_t kth = (uint32_t)((1. - (double)KP_MID_OCC_FRAC) * (double)n);
            if (kth >= n) kth = n - 1;
            uint32_t cum = 0, q = 0;
            for (uint32_t c = 0; c < (uint32_t)KP_MID_OCC_HIST; ++c) {
                cum += s_hist[c];
                if (cum > kth) { q = c; break; }
            }
            mid = max(mid, q + 1u);
        }
        sc.state[(size_t)sc.n_asm + a] = mid;
    }
}

__global__ __launch_bounds__(64 * OCC_WAVES) void kp_occ_cut_kernel(KpBatchView b, const int32_t *__restrict__ gene_len, uint64_t *__restrict__ keys,
                                                                  uint32_t *__restrict__ count, uint32_t cap, KpKeyBits kb, OccScratch sc,
                                                                  int phase) {
    __shared__ uint32_t s_cnt[OCC_WAVES][OCC_BINS];
    __shared__ uint32_t s_dropped, s_base, s_wave_n[OCC_WAVES];
    __shared__ uint32_t s_over, s_any, s_mid, s_slot;
    const int a = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t n = count[a];
    if (n > cap) n = cap;
    uint64_t *k = keys + (size_t)a * cap;
    uint32_t *cnt = s_cnt[wave];
    for (int i = lane; i < OCC_BINS; i += 64) cnt[i] = 0;
    if (threadIdx.x == 0) { s_dropped = 0; s_base = 0; s_over = 0; s_any = 0; s_mid = KP_MID_OCC; }
    __syncthreads();
    // Two launches (kp_spec.h, OCCURRENCE CUT).  Phase 0 only looks for seeds with more than KP_MID_OCC anchors; nearly every
    // assembly has none.  Where a GENE has KP_MIN_ANCHORS such seeds the block claims a counting table for kp_occ_sketch_kernel /
    // kp_occ_quantile_kernel, which work out minimap2's mid_occ of the assembly between the two launches.  Phase 1 returns at
    // once for an assembly without a seed beyond the floor, and drops what exceeds the assembly's mid_occ -- or the floor -- in
    // the others.
    const int pass = phase;
    if (phase == 1) {
        const uint32_t st = sc.state[a];
        if (!(st & 1u)) return;
        if (threadIdx.x == 0) s_mid = (st & 2u) ? sc.state[(size_t)sc.n_asm + a] : (uint32_t)KP_MID_OCC;
        __syncthreads();
    }
    {
    const uint32_t thr = pass == 0 ? (uint32_t)KP_MID_OCC : s_mid;
    const bool apply = pass == 1;
    const uint32_t per = (((n + OCC_WAVES - 1) / OCC_WAVES) + 63u) & ~63u;
    const uint32_t lo = (uint32_t)wave * per, hi = min(n, lo + per);
    uint32_t cur = lo;
    if (lo > 0 && lo < n) {  // a slice that starts inside a gene's stretch leaves it to the wave before
        const uint32_t g_prev = kp_ckey_gs(k[lo - 1], kb) >> 1;
        for (;;) {
            const uint32_t i = cur + lane;
            const unsigned long long other = __ballot(i < n && (kp_ckey_gs(k[i], kb) >> 1) != g_prev);
            if (other) { cur += (uint32_t)__builtin_ctzll(other); break; }
            cur += 64;
            if (cur >= n) break;
        }
    }
    // The exact count of one gene's stretch [cur, end): rare (see the certificate below), so it may reload what it needs.
    auto exact_cut = [&](uint32_t cur, uint32_t end) {
        const uint32_t gene = kp_ckey_gs(k[cur], kb) >> 1;
        const int glen = gene_len[gene];
        auto qf_of = [&](uint64_t key) { const int q = (int)kp_ckey_qpos(key, kb); return (kp_ckey_gs(key, kb) & 1u) ? glen - KP_K - q : q; };
        auto wave_sync = [&]() {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        // positions are counted a window of OCC_BINS at a time, exactly
        uint32_t n_over = 0;  // (first pass) seeds of this gene beyond the floor
        for (int w0 = 0; w0 < glen; w0 += OCC_BINS) {
            for (uint32_t i = cur + lane; i < end; i += 64) {
                const uint64_t key = k[i];
                if (key == OCC_TOMB) continue;
                const int qf = qf_of(key) - w0;
                if (qf >= 0 && qf < OCC_BINS) atomicAdd(&cnt[qf], 1u);
            }
            wave_sync();
            bool any_drop = false;
            for (uint32_t i0 = cur; i0 < end; i0 += 64) {
                const uint32_t i = i0 + lane;
                const uint64_t key = i < end ? k[i] : OCC_TOMB;
                const int qf = key != OCC_TOMB ? qf_of(key) - w0 : -1;
                const bool drop = qf >= 0 && qf < OCC_BINS && cnt[qf] > thr;
                if (drop && apply) k[i] = OCC_TOMB;  // (the counters are read, not changed: every anchor of the seed sees the same count)
                any_drop = any_drop || __any(drop);
            }
            wave_sync();
            if (any_drop) {  // rare: the dropped anchors no longer tell which counters they touched
                if (!apply)  // how many of the gene's seeds are beyond the floor (the counters of this window hold them)
                    for (int i = lane; i < OCC_BINS; i += 64) n_over += (uint32_t)__builtin_popcountll(__ballot(cnt[i] > thr));
                if (lane == 0) { if (apply) s_dropped = 1; else s_any = 1; }
                for (int i = lane; i < OCC_BINS; i += 64) cnt[i] = 0;
            } else {
                for (uint32_t i = cur + lane; i < end; i += 64) {
                    const uint64_t key = k[i];
                    if (key == OCC_TOMB) continue;
                    const int qf = qf_of(key) - w0;
                    if (qf >= 0 && qf < OCC_BINS) cnt[qf] = 0;
                }
            }
            wave_sync();
        }
        if (!apply && n_over >= (uint32_t)KP_MIN_ANCHORS && lane == 0) s_over = 1;  // (kp_spec.h: enough of them to chain on their own)
    };
    // CERTIFICATE.  The anchors of one seed lie on different (gene/strand, diagonal) pairs, so a stretch with anchors on ten or
    // fewer of them cannot hold a seed with more than ten anchors -- and a gene's anchors in an assembly sit on one or two
    // diagonals unless the assembly really repeats it.  Rounds of 64 anchors: gene changes and diagonal changes are two
    // ballots, every stretch that begins in the round counts its own diagonals from them (the one that reaches into the next
    // round is carried); only a stretch that fails the certificate is counted exactly.  No memory is touched beyond the keys.
    uint32_t open_start = cur, open_diags = 0;  // the stretch that reaches the current round from before it
    bool open = false;
    uint64_t next_key = cur + lane < n ? k[cur + lane] : 0ull;
    uint64_t prev_key = cur > 0 ? k[cur - 1] : 0ull;
    const uint32_t g_hi = hi > 0 && hi <= n ? kp_ckey_gs(k[hi - 1], kb) >> 1 : 0xFFFFFFFFu;
    for (uint32_t w = cur; w < n; w += 64) {
        const uint32_t i = w + lane;
        const uint64_t key = next_key;
        next_key = i + 64 < n ? k[i + 64] : 0ull;
        const uint32_t gsd_hi = (uint32_t)(key >> 32), gene = kp_ckey_gs(key, kb) >> 1;
        // the round's anchors that are this wave's: up to the slice's end, and beyond it while the gene stays the same
        const unsigned long long foreign = __ballot(!(i < n && (i < hi || gene == g_hi)));
        const int n_valid = foreign ? (int)__builtin_ctzll(foreign) : 64;
        if (n_valid == 0) break;
        const bool valid = lane < n_valid;
        uint64_t pk = ((uint64_t)__shfl_up(gsd_hi, 1) << 32) | __shfl_up((uint32_t)key, 1);
        if (lane == 0) pk = prev_key;
        const bool first_ever = w == 0 && lane == 0;
        const bool head = valid && (first_ever || (kp_ckey_gs(pk, kb) >> 1) != gene || (w == cur && lane == 0 && !open));
        const bool newdiag = valid && (first_ever || (pk >> kb.qb) != (key >> kb.qb));  // another gene/strand or diagonal than the anchor before
        const unsigned long long heads = __ballot(head), diags = __ballot(newdiag);
        prev_key = __shfl(key, n_valid - 1);
        const int first_head = heads ? (int)__builtin_ctzll(heads) : n_valid;
        const unsigned long long valid_mask = n_valid == 64 ? ~0ull : ((1ull << n_valid) - 1ull);
        if (open) {  // the carried stretch goes on for `first_head` anchors of this round
            open_diags += (uint32_t)__builtin_popcountll(diags & (first_head == 64 ? ~0ull : ((1ull << first_head) - 1ull)));
            if (first_head < n_valid || n_valid < 64) {  // ... and ends here
                if (open_diags > thr) exact_cut(open_start, w + (uint32_t)first_head);
                open = false;
            }
        }
        if (heads) {
            // every head lane measures its own stretch: [lane, my_end) and its diagonals (its own anchor opens the first)
            const unsigned long long later = lane < 63 ? heads & (~0ull << (lane + 1)) : 0ull;
            const int my_end = later ? (int)__builtin_ctzll(later) : n_valid;
            const unsigned long long range = (my_end == 64 ? ~0ull : ((1ull << my_end) - 1ull)) & ~((1ull << lane) - 1ull) & valid_mask;
            const uint32_t my_diags = (uint32_t)__builtin_popcountll((diags | (1ull << lane)) & range);
            const bool closed = head && (later != 0ull || n_valid < 64);  // ends inside the round (or with the wave's anchors)
            unsigned long long todo = __ballot(closed && my_diags > thr);
            while (todo) {  // (almost never)
                const int pos = (int)__builtin_ctzll(todo);
                todo &= todo - 1;
                exact_cut(w + (uint32_t)pos, w + (uint32_t)__shfl(my_end, pos));
            }
            const int last_head = 63 - (int)__builtin_clzll(heads);
            if (n_valid == 64) {  // the last stretch of the round reaches the next one
                open = true;
                open_start = w + (uint32_t)last_head;
                open_diags = (uint32_t)__shfl((int)my_diags, last_head);
            }
        }
        if (n_valid < 64) break;
    }
    if (open && open_diags > thr) {  // (the list ended with the carried stretch)
        uint32_t end = open_start;
        const uint32_t gene = kp_ckey_gs(k[open_start], kb) >> 1;
        for (;;) {
            const uint32_t i = end + lane;
            const unsigned long long other = __ballot(!(i < n && (kp_ckey_gs(k[i], kb) >> 1) == gene));
            if (other) { end += (uint32_t)__builtin_ctzll(other); break; }
            end += 64;
        }
        exact_cut(open_start, end);
    }
    __syncthreads();
    }
    if (phase == 0) {
        uint32_t st = (s_any ? 1u : 0u), slot = 0;
        if (s_over) {  // some gene has KP_MIN_ANCHORS seeds beyond the floor: the assembly's own mid_occ decides (rare)
            if (threadIdx.x == 0) s_slot = (uint32_t)atomicAdd(sc.demand, 1ull);
            __syncthreads();
            slot = s_slot;
            if (slot < sc.n_slots) {  // (no table left: the floor for now; the host sees the demand and reruns the pass)
                st |= 2u;
                uint32_t *tk = sc.keys + ((size_t)slot << sc.log2_size), *tc = sc.cnts + ((size_t)slot << sc.log2_size);
                for (uint32_t i = threadIdx.x; i < (1u << sc.log2_size); i += blockDim.x) { tk[i] = 0xFFFFFFFFu; tc[i] = 0u; }
                for (uint32_t i = threadIdx.x; i < (uint32_t)OCC_QWORDS; i += blockDim.x) occ_qwords(sc, slot)[i] = 0u;
                if (threadIdx.x == 0) { sc.state[2 * (size_t)sc.n_asm + slot] = (uint32_t)a; sc.state[(size_t)sc.n_asm + a] = KP_MID_OCC; }
            }
        }
        if (threadIdx.x == 0) sc.state[a] = st;
        return;
    }
    __syncthreads();
    if (!s_dropped) return;
    // compaction of the whole list, in order, a block-wide chunk at a time (reads of a chunk finish before its writes: the
    // write position never passes the read position)
    for (uint32_t c0 = 0; c0 < n; c0 += 64 * OCC_WAVES) {
        const uint32_t i = c0 + threadIdx.x;
        const uint64_t key = i < n ? k[i] : OCC_TOMB;
        const bool keep = key != OCC_TOMB;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave_n[wave] = (uint32_t)__builtin_popcountll(m);
        __syncthreads();
        uint32_t off = s_base;
        for (int w = 0; w < wave; ++w) off += s_wave_n[w];
        if (keep) k[off + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull))] = key;
        __syncthreads();
        if (threadIdx.x == 0) { uint32_t t = 0; for (int w = 0; w < OCC_WAVES; ++w) t += s_wave_n[w]; s_base += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) count[a] = s_base;
}

__global__ void kp_segments_kernel(const uint32_t *__restrict__ count, uint32_t cap, int n_asm,
                                   uint32_t *__restrict__ seg_begin, uint32_t *__restrict__ seg_end) {
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= n_asm) return;
    const uint32_t n = count[a] > cap ? cap : count[a];
    seg_begin[a] = (uint32_t)a * cap;
    seg_end[a] = (uint32_t)a * cap + n;
}

// ---- chain scores: one lane per provisional task (kp_spec.h: which clusters become tasks) -----------------------------------
// Small clusters are chained as minimap2 chains them (chain_small); a cluster that does not reach KP_MIN_CHAIN_SCORE is
// REJECTED: n_anchors = 0, its result row reads score 0 (nothing downstream fills or traces it: the order below leaves it
// out).  Larger clusters keep their anchor count and get the score of a co-linear chain.
// A lane's work grows with the square of its cluster's anchor count and a wave runs at the pace of its largest cluster, so a
// block does not take its tasks as they lie in the list (appended by thousands of waves: sizes mixed at random): it takes a
// tile of CS_TILE of them, ranks the tile by anchor count (counting sort over the 26 possible sizes, largest first, in LDS)
// and hands each round of 64 lanes neighbours in that order.
#ifndef KP_CS_TILE
#define KP_CS_TILE 512
#endif
constexpr int CS_TILE = KP_CS_TILE;
__global__ __launch_bounds__(CS_THREADS) void kp_chain_score_kernel(const uint64_t *__restrict__ keys, uint32_t cap, KpKeyBits kb,
                                                             KpTask *__restrict__ tasks, const uint32_t *__restrict__ task_count,
                                                             uint32_t task_cap, KpSwResult *__restrict__ results) {
    __shared__ ChainScratch cs;
    __shared__ uint32_t s_bin[KP_CHAIN_DP_MAX + 2];
    __shared__ uint16_t s_order[CS_TILE];
    static_assert(CS_TILE % CS_THREADS == 0 && CS_TILE <= 65536, "whole rounds, 16-bit tile indices");
    const int cls = blockIdx.y, lane = (int)threadIdx.x;
    uint32_t n = task_count[cls];
    if (n > task_cap) n = task_cap;
    KpTask *list = tasks + (size_t)cls * task_cap;
    for (uint32_t tile0 = blockIdx.x * (uint32_t)CS_TILE; tile0 < n; tile0 += gridDim.x * (uint32_t)CS_TILE) {
        const uint32_t m = min((uint32_t)CS_TILE, n - tile0);
        if (lane < KP_CHAIN_DP_MAX + 2) s_bin[lane] = 0;
        __syncthreads();
        int bin[CS_TILE / CS_THREADS];
        uint32_t rank[CS_TILE / CS_THREADS];
#pragma unroll
        for (int r = 0; r < CS_TILE / CS_THREADS; ++r) {
            const uint32_t i = (uint32_t)(r * CS_THREADS + lane);
            bin[r] = 0; rank[r] = 0;
            if (i < m) {
                const int c = list[tile0 + i].n_anchors;
                bin[r] = c > KP_CHAIN_DP_MAX ? 0 : KP_CHAIN_DP_MAX + 1 - c;  // bin 0: nothing to chain; then 24, 23, ... anchors
                rank[r] = atomicAdd(&s_bin[bin[r]], 1u);
            }
        }
        __syncthreads();
        if (lane == 0) {
            uint32_t acc = 0;
            for (int b = 0; b < KP_CHAIN_DP_MAX + 2; ++b) { const uint32_t c = s_bin[b]; s_bin[b] = acc; acc += c; }
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CS_TILE / CS_THREADS; ++r) {
            const uint32_t i = (uint32_t)(r * CS_THREADS + lane);
            if (i < m) s_order[s_bin[bin[r]] + rank[r]] = (uint16_t)i;
        }
        __syncthreads();
        for (uint32_t k = (uint32_t)lane; k < m; k += CS_THREADS) {
            const uint32_t i = tile0 + s_order[k];
            KpTask &t = list[i];
            const int cnt = t.n_anchors;
            const uint32_t first = (uint32_t)t.chain_score;
            int chain_cnt = cnt, chain_sc;
            if (cnt <= KP_CHAIN_DP_MAX) {
                chain_sc = chain_small(keys + (size_t)t.asm_id * cap + first, cnt, kb, cs, lane, &chain_cnt);
                if (chain_sc < KP_MIN_CHAIN_SCORE) {
                    chain_cnt = 0;
                    results[(size_t)cls * task_cap + i].score = 0;
                }
            } else {
                chain_sc = min(KP_K * cnt, (int)(t.qspan >> 16) - (int)(t.qspan & 0xFFFFu) + KP_K);
            }
            t.n_anchors = chain_cnt;
            t.chain_score = chain_sc;
        }
        __syncthreads();
    }
}

// ---- task order: counting sort of each width class by the rows the fill kernel will compute, most first ---------------------
// (rejected tasks, n_anchors == 0, are left out; ordered[cls] = how many the order holds)
// Tasks of one wave run in lock step for as many steps as the longest of them needs, so neighbours in the processing
// order should have similar lengths; starting with the long ones also keeps the tail of the launch short.
__device__ __forceinline__ int length_bucket(int rows) {
    const int b = rows >> 5;
    return 63 - (b > 63 ? 63 : b);
}
constexpr int NB = KP_ORDER_BUCKETS;  // 64 length buckets + the bucket of long genes' tasks (last in the order)
__device__ __forceinline__ int task_bucket(const KpBatchView &b, const KpGenes &genes, const KpTask &t);
// rows of the gene the task's band can reach inside its contig (kp_task_rows: a gene at a contig end is not filled beyond it)
__device__ __forceinline__ int task_rows(const KpBatchView &b, const KpGenes &genes, const KpTask &t) {
    const int c_abs = b.asm_first_ctg[t.asm_id] + t.contig;
    const int cstart = b.ctg_start[c_abs];
    int r_lo, r_hi;
    kp_task_rows(t.lo, t.width, cstart, cstart + b.ctg_len[c_abs], genes.len[t.gs >> 1], &r_lo, &r_hi);
    return r_hi - r_lo;
}

__device__ __forceinline__ int task_bucket(const KpBatchView &b, const KpGenes &genes, const KpTask &t) {
    return genes.len[t.gs >> 1] > KP_FILL16_MAX_GENE_LEN ? NB - 1 : length_bucket(task_rows(b, genes, t));
}

__global__ __launch_bounds__(256) void kp_task_hist_kernel(KpBatchView b, KpGenes genes, const KpTask *__restrict__ tasks,
                                                           const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                           uint32_t *__restrict__ hist) {
    __shared__ uint32_t s_h[NB];
    const int cls = blockIdx.y;
    uint32_t n = task_count[cls];
    if (n > task_cap) n = task_cap;
    if (threadIdx.x < NB) s_h[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const KpTask &t = tasks[(size_t)cls * task_cap + i];
        if (t.n_anchors) atomicAdd(&s_h[task_bucket(b, genes, t)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < NB && s_h[threadIdx.x]) atomicAdd(&hist[cls * NB + threadIdx.x], s_h[threadIdx.x]);
}

__global__ __launch_bounds__(256) void kp_task_scatter_kernel(KpBatchView b, KpGenes genes, const KpTask *__restrict__ tasks,
                                                              const uint32_t *__restrict__ task_count, uint32_t task_cap,
                                                              uint32_t *__restrict__ hist, uint32_t *__restrict__ order,
                                                              uint32_t *__restrict__ ordered) {
    __shared__ uint32_t s_start[NB], s_h[NB], s_base[NB];
    const int cls = blockIdx.y;
    uint32_t n = task_count[cls];
    if (n > task_cap) n = task_cap;
    if (threadIdx.x == 0) {  // bucket starts from the finished histogram (65 entries: not worth a scan)
        uint32_t acc = 0;
        for (int k = 0; k < NB; ++k) { s_start[k] = acc; acc += hist[cls * NB + k]; }
        if (blockIdx.x == 0) {
            ordered[cls] = s_start[NB - 1];                    // ordinary tasks: the packed fill kernel's
            ordered[KP_N_CLASSES + cls] = acc - s_start[NB - 1];  // tasks of long genes, behind them
        }
    }
    if (threadIdx.x < NB) s_h[threadIdx.x] = 0;
    __syncthreads();
    // each block handles one contiguous chunk so that it can reserve its slots with one atomic per bucket
    const uint32_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint32_t lo = blockIdx.x * per, hi = min(n, lo + per);
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const KpTask &t = tasks[(size_t)cls * task_cap + i];
        if (t.n_anchors) atomicAdd(&s_h[task_bucket(b, genes, t)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < NB) {
        s_base[threadIdx.x] = s_h[threadIdx.x] ? atomicAdd(&hist[KP_N_CLASSES * NB + cls * NB + threadIdx.x], s_h[threadIdx.x]) : 0u;
        s_h[threadIdx.x] = 0;
    }
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const KpTask &t = tasks[(size_t)cls * task_cap + i];
        if (!t.n_anchors) continue;
        const int k = task_bucket(b, genes, t);
        order[(size_t)cls * task_cap + s_start[k] + s_base[k] + atomicAdd(&s_h[k], 1u)] = i;
    }
}

}  // namespace

void kp_launch_task_order(const KpBatchView &b, const KpGenes &genes, const uint64_t *sorted_anchors, uint32_t cap, KpKeyBits key_bits,
                          KpTask *tasks, const uint32_t *task_count, uint32_t task_cap, KpSwResult *results, uint32_t *hist,
                          uint32_t *order, hipStream_t stream) {
    const dim3 grid(128, KP_N_CLASSES), block(256);
    hipLaunchKernelGGL(kp_chain_score_kernel, dim3(2048, KP_N_CLASSES), dim3(CS_THREADS), 0, stream, sorted_anchors, cap, key_bits, tasks,
                       task_count, task_cap, results);
    hipLaunchKernelGGL(kp_task_hist_kernel, grid, block, 0, stream, b, genes, tasks, task_count, task_cap, hist);
    hipLaunchKernelGGL(kp_task_scatter_kernel, grid, block, 0, stream, b, genes, tasks, task_count, task_cap, hist, order,
                       hist + KP_ORDER_COUNTS);
}

void kp_launch_segments(const uint32_t *count, uint32_t cap, int n_asm, uint32_t *seg_begin, uint32_t *seg_end,
                        hipStream_t stream) {
    if (n_asm == 0) return;
    hipLaunchKernelGGL(kp_segments_kernel, dim3((n_asm + 255) / 256), dim3(256), 0, stream, count, cap, n_asm, seg_begin,
                       seg_end);
}

size_t kp_occ_state_words(size_t n_asm, uint32_t occ_slots) { return 2 * n_asm + occ_slots + (size_t)occ_slots * OCC_QWORDS; }

void kp_launch_occ_cut(const KpBatchView &b, const int32_t *gene_len, uint64_t *sorted_anchors, uint32_t *anchor_count, uint32_t cap,
                       KpKeyBits key_bits, uint32_t *occ_keys, uint32_t *occ_cnts, uint32_t *occ_state, unsigned long long *occ_demand,
                       uint32_t occ_slots, uint32_t occ_log2_size, hipStream_t stream) {
    if (b.n_asm == 0) return;
    OccScratch sc;
    sc.keys = occ_keys; sc.cnts = occ_cnts; sc.state = occ_state; sc.demand = occ_demand; sc.n_slots = occ_slots; sc.log2_size = occ_log2_size;
    sc.n_asm = b.n_asm;
    hipLaunchKernelGGL(kp_occ_cut_kernel, dim3(b.n_asm), dim3(64 * OCC_WAVES), 0, stream, b, gene_len, sorted_anchors, anchor_count, cap, key_bits, sc, 0);
    hipLaunchKernelGGL(kp_occ_sketch_kernel, dim3(OCC_PARTS, occ_slots), dim3(256), 0, stream, b, sc);
    hipLaunchKernelGGL(kp_occ_quantile_kernel, dim3(OCC_QPARTS, occ_slots), dim3(1024), 0, stream, sc);
    hipLaunchKernelGGL(kp_occ_cut_kernel, dim3(b.n_asm), dim3(64 * OCC_WAVES), 0, stream, b, gene_len, sorted_anchors, anchor_count, cap, key_bits, sc, 1);
}

void kp_launch_chain(const KpBatchView &b, const uint64_t *sorted_anchors, const uint32_t *anchor_count, uint32_t cap,
                     KpKeyBits key_bits, KpTask *tasks, uint32_t *task_count, uint32_t task_cap, KpGroup *groups,
                     uint32_t *group_count, uint32_t group_cap, hipStream_t stream) {
    if (b.n_asm == 0) return;
    GroupOut go;
    go.groups = groups; go.count = group_count; go.cap = group_cap;
    hipLaunchKernelGGL(kp_chain_kernel, dim3(CHAIN_SLICES / CHAIN_WAVES, b.n_asm), dim3(64 * CHAIN_WAVES), 0, stream, b, sorted_anchors, anchor_count,
                       cap, key_bits, tasks, task_count, task_cap, go);
}
