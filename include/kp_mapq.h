/* kp_mapq.h -- mapping quality of a primary hit (kp-align v3), shared by the HIP kernels and the CPU oracle.
 *
 * minimap2's mm_set_mapq (the published source of the aligner the reference's rammappy wheel is described as following:
 * docs/serotyping/method.md:23-28 "minimap2-based") with the quantities it reads there: the chain's score and anchor
 * count (kp_spec.h: the cluster's chain), the hit's alignment score for dp_max, and from the secondaries that
 * mm_set_parent hangs on the hit (kp_spec.h, hit record) the best chain score `subsc`, the best alignment score
 * `dp_max2` and their count `n_sub`.  It cannot be checked against rammappy here (parity unpinned).
 * The reference reads mapq only as the third cull key (src/kaptive/core/alignment.py:669-675).
 *
 *   pen_s1 = chain > 100 ? 1 : 0.01 * chain            pen_cm = seeds > 10 ? 1 : 0.1 * seeds;  pen_cm = min(pen_s1, pen_cm)
 *   identity = matches / block_len                     subsc = max(subsc, KP_MIN_CHAIN_SCORE)
 *   dp_max2 > 0:  x = dp_max2 * subsc / score / chain;  q = identity * pen_cm * 40 * (1 - x * x) * ln(score / match)
 *                 q = min(q, int(6.02 * identity^2 * (score - dp_max2) / match + 0.499))
 *   else:         x = subsc / chain;                    q = identity * pen_cm * 40 * (1 - x) * ln(score / match)
 *   q -= int(4.343 * ln(n_sub + 1) + 0.499);  clamp to [0, 60];  q == 0 and score > dp_max2 -> 1
 *
 * All arithmetic is float32 in exactly this order (no fused multiply-add: the function switches contraction off for
 * clang; gcc in ISO C mode does not contract).  The two logarithms come from tables the caller fills on the host with
 * kp_mapq_ln below (ln_half[i] = kp_mapq_ln(i / 2), ln_int[i] = kp_mapq_ln(i); entry 0 is 0): a fixed sequence of IEEE
 * double operations, so the tables -- and with them every mapping quality -- are the same on every host, whatever its libm.
 */
#ifndef KP_MAPQ_H
#define KP_MAPQ_H

#include <stdint.h>

#include "kp_spec.h"

#ifndef KP_MAPQ_FN
#define KP_MAPQ_FN static inline
#endif

/* Natural logarithm of x > 0 as float, from IEEE double +, -, *, / only: x = m * 2^e with m in [sqrt(1/2), sqrt(2)),
 * ln m = 2 atanh s with s = (m - 1) / (m + 1) summed as an odd series to s^21 (|s| < 0.172: the truncation is below
 * 1e-18), ln x = e * ln 2 + ln m, rounded once to float. */
static inline float kp_mapq_ln(double x) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    union {
        double d;
        uint64_t u;
    } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7FF) - 1023;
    v.u = (v.u & 0x000FFFFFFFFFFFFFULL) | 0x3FF0000000000000ULL; /* m in [1, 2) */
    double m = v.d;
    if (m > 1.4142135623730951) m = m * 0.5, e += 1;
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    double term = s, sum = 0.0;
    for (int k = 1; k <= 21; k += 2) {
        sum = sum + term / (double)k;
        term = term * s2;
    }
    return (float)((double)e * 0.6931471805599453 + 2.0 * sum);
}

#define KP_MAPQ_LN_HALF_SIZE 131072 /* scores are at most 2 * KP_MAX_GENE_LEN plus the two-piece credit of long gaps */
#define KP_MAPQ_LN_INT_SIZE 4096

KP_MAPQ_FN int kp_mapq_value(int score, int chain, int seeds, int matches, int block_len, int subsc_in, int dp_max2, int n_sub,
                             const float *ln_half, const float *ln_int) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
    const float pen_s1 = chain > 100 ? 1.0f : 0.01f * (float)chain;
    float pen_cm = seeds > 10 ? 1.0f : 0.1f * (float)seeds;
    pen_cm = pen_s1 < pen_cm ? pen_s1 : pen_cm;
    const float identity = (float)matches / (float)block_len;
    const int subsc = subsc_in > KP_MIN_CHAIN_SCORE ? subsc_in : KP_MIN_CHAIN_SCORE;
    const float lg = ln_half[score < KP_MAPQ_LN_HALF_SIZE ? score : KP_MAPQ_LN_HALF_SIZE - 1];
    int q;
    if (dp_max2 > 0 && score > 0) {
        float x = (float)dp_max2 * (float)subsc;
        x = x / (float)score;
        x = x / (float)chain;
        float v = identity * pen_cm;
        v = v * 40.0f;
        v = v * (1.0f - x * x);
        v = v * lg;
        q = (int)v;
        float alt = 6.02f * identity;
        alt = alt * identity;
        alt = alt * (float)(score - dp_max2);
        alt = alt / (float)KP_SC_MATCH;
        alt = alt + 0.499f;
        const int q_alt = (int)alt;
        q = q < q_alt ? q : q_alt;
    } else {
        const float x = (float)subsc / (float)chain;
        float v = identity * pen_cm;
        v = v * 40.0f;
        v = v * (1.0f - x);
        v = v * lg;
        q = (int)v;
    }
    {
        float pen = 4.343f * ln_int[(n_sub + 1) < KP_MAPQ_LN_INT_SIZE ? (n_sub + 1) : KP_MAPQ_LN_INT_SIZE - 1];
        pen = pen + 0.499f;
        q -= (int)pen;
    }
    q = q > 0 ? q : 0;
    q = q < 60 ? q : 60;
    if (q == 0 && score > dp_max2) q = 1;
    return q;
}

#endif /* KP_MAPQ_H */
