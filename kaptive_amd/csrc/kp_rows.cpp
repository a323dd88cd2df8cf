// kp_rows.cpp -- KaptiveRow TSV bytes for a whole batch, straight from the records of the batched reduction (host only).
//
// Stands in for KaptiveRow.from_result + __bytes__ (src/kaptive/serotyping/io.py:191-296, 37-43) called once per genome
// through a SerotypingResult object: SURVEY.md section 8 row a15.  The reference builds 22 byte strings per genome in a
// Python loop over the kept hits; here one call formats every assembly of a batch from the arrays kp_batch_typing
// returned plus the per-assembly decisions the host finished column-wise (kaptive_amd/serotyping/batch.py).  Number
// formatting is C's "%.2f" on the same doubles Python's "%.2f" gets (float32 identities / coverages widen exactly), so
// the bytes are identical; tests compare them with the reference's golden rows.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/kaptive_amd.h"

namespace {

struct Out {
    char *p;
    int64_t cap, n = 0;  // n keeps counting past cap: the caller learns how much room the rows need
    inline void put(const char *s, int64_t len) {
        if (n + len <= cap) std::memcpy(p + n, s, (size_t)len);
        n += len;
    }
    inline void put(char c) {
        if (n < cap) p[n] = c;
        ++n;
    }
    inline void lit(const char *s) { put(s, (int64_t)std::strlen(s)); }
    inline void pct(double v) {  // "%.2f%%"
        char buf[64];
        const int len = std::snprintf(buf, sizeof buf, "%.2f%%", v);
        put(buf, len);
    }
    // "%.2f%%" of a float32 value in [0, 2^20): the product with 100 is exact in double (24 + 7 significant bits), so
    // rounding it to an integer with ties to even is what printf does with the exact decimal expansion
    inline void pct_f32(float v) {
        if (!(v >= 0.0f && v < 1048576.0f)) { pct((double)v); return; }
        long long cents = std::llrint((double)v * 100.0);
        char buf[24];
        int at = 24;
        buf[--at] = '%';
        buf[--at] = (char)('0' + cents % 10); cents /= 10;
        buf[--at] = (char)('0' + cents % 10); cents /= 10;
        buf[--at] = '.';
        do { buf[--at] = (char)('0' + cents % 10); cents /= 10; } while (cents);
        put(buf + at, 24 - at);
    }
    inline void num(long long v) {
        char buf[32];
        const int len = std::snprintf(buf, sizeof buf, "%lld", v);
        put(buf, len);
    }
};

inline bool alive(const kp_kept &k) { return (k.flags & KP_F_SPURIOUS) == 0; }

// how many distinct genes the selected hits have (np.unique(gene_indices[mask]) in the reference)
template <class Pred>
int distinct_genes(const kp_kept *k, int n, Pred pred) {
    int count = 0;
    for (int i = 0; i < n; ++i) {
        if (!alive(k[i]) || !pred(k[i])) continue;
        bool seen = false;
        for (int j = 0; j < i && !seen; ++j) seen = alive(k[j]) && pred(k[j]) && k[j].gene == k[i].gene;
        count += seen ? 0 : 1;
    }
    return count;
}

// "id,ident%,cov%[,state]" per selected hit, ';'-joined (io.py _gene_details)
template <class Pred>
void details(Out &o, const kp_row_tables *t, const kp_kept *k, int n, Pred pred) {
    bool first = true;
    for (int i = 0; i < n; ++i) {
        if (!alive(k[i]) || !pred(k[i])) continue;
        if (!first) o.put(';');
        first = false;
        o.put(t->gene_ids + t->gene_id_off[k[i].gene], t->gene_id_off[k[i].gene + 1] - t->gene_id_off[k[i].gene]);
        o.put(',');
        o.pct_f32(k[i].pident);
        o.put(',');
        o.pct_f32(k[i].coverage);
        if (k[i].state == KP_STATE_PARTIAL) o.lit(",partial");
        else if (k[i].state == KP_STATE_TRUNCATED) o.lit(",truncated");
        else if (k[i].state == KP_STATE_NOVEL) o.lit(",below_id_threshold");
    }
}

void share(Out &o, int n, int total) {  // "%d / %d (%.2f%%)"
    if (!total) { o.lit("0 / 0 (0.00%)"); return; }
    o.num(n); o.lit(" / "); o.num(total); o.lit(" (");
    o.pct((double)n / (double)total * 100.0);
    o.put(')');
}

}  // namespace

extern "C" int64_t kp_format_rows(const kp_row_tables *t, int32_t n_asm, const kp_asm_summary *sums, const kp_kept *kept,
                                  int32_t kept_stride, const kp_row_columns *c, char *out, int64_t cap) {
    if (!t || !c || n_asm < 0 || (n_asm > 0 && (!sums || !kept)) || cap < 0 || (cap > 0 && !out)) return KP_EINVAL;
    Out o{out, cap};
    for (int a = 0; a < n_asm; ++a) {
        const kp_asm_summary &s = sums[a];
        const kp_kept *k = kept + (size_t)a * (size_t)kept_stride;
        const int n = s.n_kept;
        const int best = c->best_locus[a];
        o.put(t->prefix, t->prefix_len);  // Kaptive version, database name, database version (tab-terminated)
        o.put(c->asm_ids + c->asm_id_off[a], c->asm_id_off[a + 1] - c->asm_id_off[a]); o.put('\t');
        o.put(t->locus_names + t->locus_name_off[best], t->locus_name_off[best + 1] - t->locus_name_off[best]); o.put('\t');
        o.put(c->phenotypes + c->phenotype_off[a], c->phenotype_off[a + 1] - c->phenotype_off[a]); o.put('\t');
        o.lit(c->typeable[a] ? "Typeable" : "Untypeable"); o.put('\t');
        static const char symbols[] = "?+-*!";  // SerotypingProblem.to_symbols (models.py:82-92)
        for (int bit = 0; bit < 5; ++bit)
            if (c->problems[a] & (1 << bit)) o.put(symbols[bit]);
        o.put('\t');
        o.pct(c->identity[a]); o.put('\t');
        o.pct(c->coverage[a]); o.put('\t');
        if (std::isnan(c->length_discrepancy[a])) o.lit("n/a");
        else o.num((long long)c->length_discrepancy[a]);
        o.put('\t');
        auto in_exp = [](const kp_kept &h) { return (h.flags & KP_F_INSIDE) && (h.flags & KP_F_EXPECTED); };
        auto out_exp = [](const kp_kept &h) { return !(h.flags & KP_F_INSIDE) && (h.flags & KP_F_EXPECTED); };
        auto in_other = [](const kp_kept &h) { return (h.flags & KP_F_INSIDE) && !(h.flags & (KP_F_EXPECTED | KP_F_EXTRA)); };
        auto out_other = [](const kp_kept &h) { return !(h.flags & KP_F_INSIDE) && !(h.flags & (KP_F_EXPECTED | KP_F_EXTRA)); };
        const int n_in = distinct_genes(k, n, in_exp), n_out = distinct_genes(k, n, out_exp);
        const int total = n_in + n_out + s.n_missing;
        share(o, n_in, total); o.put('\t');
        details(o, t, k, n, in_exp); o.put('\t');
        {  // missing expected genes, in database order
            const int g0 = t->locus_gene_off[best], ng = t->locus_gene_len[best] < KP_MAX_LOCUS_GENES ? t->locus_gene_len[best] : KP_MAX_LOCUS_GENES;
            bool first = true;
            for (int j = 0; j < ng; ++j) {
                if (!((s.missing_mask[j >> 6] >> (j & 63)) & 1ull)) continue;
                if (!first) o.put(';');
                first = false;
                o.put(t->gene_ids + t->gene_id_off[g0 + j], t->gene_id_off[g0 + j + 1] - t->gene_id_off[g0 + j]);
            }
        }
        o.put('\t');
        o.num(distinct_genes(k, n, in_other)); o.put('\t');
        details(o, t, k, n, in_other); o.put('\t');
        share(o, n_out, total); o.put('\t');
        details(o, t, k, n, out_exp); o.put('\t');
        o.num(distinct_genes(k, n, out_other)); o.put('\t');
        details(o, t, k, n, out_other); o.put('\t');
        details(o, t, k, n, [](const kp_kept &h) { return h.state == KP_STATE_TRUNCATED || h.state == KP_STATE_PARTIAL; });
        o.put('\t');
        details(o, t, k, n, [](const kp_kept &h) { return (h.flags & KP_F_EXTRA) != 0; });
        o.put('\n');
    }
    return o.n;
}
