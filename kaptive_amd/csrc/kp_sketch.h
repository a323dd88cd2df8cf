// kp_sketch.h -- the seed state machine of include/kp_spec.h (minimap2's mm_sketch for w = 10, k = 15), one step at a time.
//
// Two users: kp_db_load sketches every gene with it on the host, and kp_edge_kernel (kp_scan.hip) runs it on the device
// over the few dozen bases next to every contig end and N run, where the streaming kernel's window rule does not apply.
// The window is kept oldest-first in ten named slots that shift every step, so that on the device every access has a
// static index and the state lives in registers.  An entry is (x << 1) | z; KP_SK_INF marks a step without a 15-mer.
#pragma once

#include "kp_internal.h"

#define KP_SK_INF 0xFFFFFFFFu

struct KpSketchState {
    uint32_t win[KP_W];  // the last KP_W steps, oldest first
    uint32_t fwd, rev;   // the 15-mer ending at the current base and its reverse complement (kp_spec.h)
    uint32_t min_v;      // the tracked minimum (an entry), KP_SK_INF = none
    int32_t min_age;     // steps since it was written
    int32_t run;         // unambiguous bases ending at the current base, saturated (only compared with <= KP_W + KP_K)
};

__host__ __device__ inline void kp_sketch_reset(KpSketchState &s) {
#pragma unroll
    for (int j = 0; j < KP_W; ++j) s.win[j] = KP_SK_INF;
    s.fwd = s.rev = 0;
    s.min_v = KP_SK_INF;
    s.min_age = 0;
    s.run = 0;
}

// One base (code 0..3, anything else ambiguous) at index i of its sequence.  emit(start, z, x) is called for every seed
// the step settles; `start` is the index of the seed's first base.
template <class Emit>
__host__ __device__ inline void kp_sketch_step(KpSketchState &s, int64_t i, uint32_t code, Emit &&emit) {
    uint32_t nv = KP_SK_INF;
    if (code < 4u) {
        s.fwd = ((s.fwd << 2) | code) & KP_KMER_MASK;
        s.rev = (s.rev >> 2) | ((3u - code) << (2 * (KP_K - 1)));
        if (s.run < 64) ++s.run;
        if (s.run >= KP_K) {
            const uint32_t z = s.fwd < s.rev ? 0u : 1u;
            nv = (kp_hash30(z ? s.rev : s.fwd) << 1) | z;
        }
    } else {
        s.run = 0;
    }
#pragma unroll
    for (int j = 0; j + 1 < KP_W; ++j) s.win[j] = s.win[j + 1];
    s.win[KP_W - 1] = nv;  // the entry of age a sits in win[KP_W - 1 - a]
    ++s.min_age;
    const int64_t newest = i - (KP_K - 1);  // first base of the 15-mer that ends at i
    if (s.run == KP_W + KP_K - 1 && s.min_v != KP_SK_INF) {  // ties of the first full window, oldest first
#pragma unroll
        for (int a = KP_W - 1; a >= 1; --a) {
            const uint32_t e = s.win[KP_W - 1 - a];
            if ((e >> 1) == (s.min_v >> 1) && a != s.min_age) emit(newest - a, e & 1u, e >> 1);
        }
    }
    if ((nv >> 1) <= (s.min_v >> 1)) {  // a new minimum takes over
        if (s.run >= KP_W + KP_K && s.min_v != KP_SK_INF) emit(newest - s.min_age, s.min_v & 1u, s.min_v >> 1);
        s.min_v = nv;
        s.min_age = 0;
    } else if (s.min_age == KP_W) {  // the minimum has left the window
        if (s.run >= KP_W + KP_K - 1) emit(newest - s.min_age, s.min_v & 1u, s.min_v >> 1);
        s.min_v = KP_SK_INF;
#pragma unroll
        for (int a = KP_W - 1; a >= 0; --a) {  // oldest to newest: the last smallest wins
            const uint32_t e = s.win[KP_W - 1 - a];
            if ((e >> 1) <= (s.min_v >> 1)) { s.min_v = e; s.min_age = a; }
        }
        if (s.run >= KP_W + KP_K - 1 && s.min_v != KP_SK_INF) {
#pragma unroll
            for (int a = KP_W - 1; a >= 0; --a) {
                const uint32_t e = s.win[KP_W - 1 - a];
                if ((e >> 1) == (s.min_v >> 1) && a != s.min_age) emit(newest - a, e & 1u, e >> 1);
            }
        }
    }
}

// after the sequence's last base (index i_last): the tracked minimum is a seed
template <class Emit>
__host__ __device__ inline void kp_sketch_final(const KpSketchState &s, int64_t i_last, Emit &&emit) {
    if (s.min_v != KP_SK_INF) emit(i_last - (KP_K - 1) - s.min_age, s.min_v & 1u, s.min_v >> 1);
}
