// kp_json.cpp -- the JSON lines of `kaptive assembly -j` for a whole batch, straight from the records of the batched reduction
// and the contigs' text (host only).
//
// Stands in for orjson.dumps(SerotypingResult.to_dict(), OPT_SERIALIZE_NUMPY | OPT_APPEND_NEWLINE) called once per genome
// (src/kaptive/serotyping/cli.py:67-76 on src/kaptive/serotyping/models.py:629-654) -- and for what has to exist before it can be
// called: a SerotypingResult with its GeneHits columns (src/kaptive/serotyping/core.py:303-329), the three Sequences
// (extract / translate, src/kaptive/core/seq.py:327-408) and their ids.  kaptive_amd/serotyping/jsonl.py restates orjson's
// output conventions in Python (the orjson wheel is absent from the build image); this file writes the same bytes for
// every assembly of a batch from the arrays kp_batch_typing returned plus the columns the host finished
// (kaptive_amd/serotyping/batch.py): tests compare the two byte for byte on every golden case.
//   * numbers: a float64 as the shortest digits that read back (std::to_chars), laid out as Ryu's format64 does; the two
//     float32 columns (coverages, protein identities) as the shortest digits of the float32, format32's layout; NaN = null
//   * strings arrive as ready JSON literals where they come from the database or the batch (the Python side escapes them
//     once per database / batch with the encoder jsonl.py uses); ids built here ("<contig>_<start>_<end>_<strand>") take the
//     escaped contig name as it is
#include <charconv>
#include <cmath>
#include <cstring>

#include "../../include/kaptive_amd.h"

namespace {

struct Out {
    char *p;
    int64_t cap, n = 0;  // n keeps counting past cap: the caller learns how much room the lines need
    inline void put(const char *s, int64_t len) {
        if (n + len <= cap) std::memcpy(p + n, s, (size_t)len);
        n += len;
    }
    inline void put(char c) {
        if (n < cap) p[n] = c;
        ++n;
    }
    inline void lit(const char *s) { put(s, (int64_t)std::strlen(s)); }
    inline void num(long long v) {
        char buf[32];
        auto r = std::to_chars(buf, buf + sizeof buf, v);
        put(buf, r.ptr - buf);
    }
    // digits of a shortest scientific representation "d[.ddd]e[+-]XX" laid out as ryu::raw::format64 / format32 lay them out
    // (kaptive_amd/serotyping/jsonl.py::_ryu_layout)
    void layout(const char *sci, const char *end, int point_max, int zeros_max) {
        char ds[40];
        int nd = 0;
        const char *q = sci;
        if (*q == '-') { put('-'); ++q; }
        for (; q < end && *q != 'e'; ++q)
            if (*q != '.') ds[nd++] = *q;
        int e10 = 0;
        if (q < end) {  // 'e', sign, digits
            ++q;
            const bool neg = *q == '-';
            if (*q == '-' || *q == '+') ++q;
            for (; q < end; ++q) e10 = e10 * 10 + (*q - '0');
            if (neg) e10 = -e10;
        }
        while (nd > 1 && ds[nd - 1] == '0') --nd;
        const int kk = e10 + 1, k = kk - nd;  // value = ds * 10^k; kk = position of the decimal point counted from the first digit
        if (k >= 0 && kk <= point_max) {
            put(ds, nd);
            for (int i = 0; i < k; ++i) put('0');
            lit(".0");
        } else if (kk > 0 && kk <= point_max) {
            put(ds, kk); put('.'); put(ds + kk, nd - kk);
        } else if (kk > -zeros_max && kk <= 0) {
            lit("0.");
            for (int i = 0; i < -kk; ++i) put('0');
            put(ds, nd);
        } else {
            put(ds[0]);
            if (nd > 1) { put('.'); put(ds + 1, nd - 1); }
            put('e');
            num(kk - 1);
        }
    }
    void f64(double v) {
        if (std::isnan(v) || std::isinf(v)) { lit("null"); return; }
        if (v == 0.0) { lit(std::signbit(v) ? "-0.0" : "0.0"); return; }
        char buf[48];
        auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
        layout(buf, r.ptr, 16, 5);
    }
    void f32(float v) {
        if (std::isnan(v) || std::isinf(v)) { lit("null"); return; }
        if (v == 0.0f) { lit(std::signbit(v) ? "-0.0" : "0.0"); return; }
        char buf[48];
        auto r = std::to_chars(buf, buf + sizeof buf, v, std::chars_format::scientific);
        layout(buf, r.ptr, 13, 6);
    }
};

inline bool alive(const kp_kept &k) { return (k.flags & KP_F_SPURIOUS) == 0; }

template <class F>
void int_array(Out &o, const char *key, const kp_kept *k, int n, F f) {
    o.lit(key);
    o.put('[');
    bool first = true;
    for (int i = 0; i < n; ++i) {
        if (!alive(k[i])) continue;
        if (!first) o.put(',');
        first = false;
        o.num(f(k[i]));
    }
    o.put(']');
}
template <class F>
void bool_array(Out &o, const char *key, const kp_kept *k, int n, F f) {
    o.lit(key);
    o.put('[');
    bool first = true;
    for (int i = 0; i < n; ++i) {
        if (!alive(k[i])) continue;
        if (!first) o.put(',');
        first = false;
        o.lit(f(k[i]) ? "true" : "false");
    }
    o.put(']');
}
void text_array(Out &o, const char *key, const kp_kept *k, int n, const char *blob, const int64_t *off) {
    o.lit(key);
    o.put('[');
    bool first = true;
    for (int i = 0; i < n; ++i) {
        if (!alive(k[i])) continue;
        if (!first) o.put(',');
        first = false;
        o.put(blob + off[k[i].gene], off[k[i].gene + 1] - off[k[i].gene]);
    }
    o.put(']');
}

// [start, end) of a contig's text, reverse-complemented for strand < 0 (Sequences.extract, core/seq.py): straight into the line
void put_extract(Out &o, const uint8_t *seqs, int64_t base, int start, int end, int strand, const uint8_t *comp) {
    const int len = end - start;
    if (len <= 0) return;
    if (o.n + len <= o.cap) {
        char *dst = o.p + o.n;
        if (strand >= 0) std::memcpy(dst, seqs + base + start, (size_t)len);
        else
            for (int i = 0; i < len; ++i) dst[i] = (char)comp[seqs[base + end - 1 - i]];
    }
    o.n += len;
}

// Everything the formatters index with values that come out of the records (a corrupt or untyped record must be an error
// code, not a read past a table): the best locus, its gene range, every live hit's gene / contig / interval, every piece.
bool records_in_range(const kp_json_tables *t, const kp_json_columns *c, int a, const kp_asm_summary &s, const kp_kept *k,
                      int kept_stride, const kp_piece *pc, int piece_stride, const int32_t *order, bool need_locus) {
    const int best = c->best_locus[a];
    if (need_locus) {
        if (best < 0 || best >= t->n_loci) return false;
        const int64_t g0 = t->locus_gene_off[best], ng = t->locus_gene_len[best];
        if (g0 < 0 || ng < 0 || g0 + ng > t->n_genes) return false;
    }
    if (s.n_kept < 0 || s.n_kept > kept_stride || s.n_pieces < 0 || s.n_pieces > piece_stride) return false;
    const int n_ctg = c->n_ctg[a];
    const int64_t text = c->ctg_text_len[a];
    const int32_t *off = c->ctg_off[a];
    auto span_ok = [&](int ctg, int64_t start, int64_t end) {
        return ctg >= 0 && ctg < n_ctg && start >= 0 && start <= end && off[ctg] >= 0 && (int64_t)off[ctg] + end <= text;
    };
    int n_alive = 0;
    for (int i = 0; i < s.n_kept; ++i) {
        if (!alive(k[i])) continue;
        ++n_alive;
        if (k[i].gene < 0 || k[i].gene >= t->n_genes || !span_ok(k[i].contig, k[i].t_start, k[i].t_end)) return false;
    }
    if (n_alive > 4096) return false;  // (the translation lengths of a line are kept on the stack)
    for (int i = 0; i < s.n_pieces; ++i) {
        if (order[i] < 0 || order[i] >= piece_stride) return false;
        const kp_piece &q = pc[order[i]];
        if (!span_ok(q.contig, q.start, q.end)) return false;
    }
    return true;
}

}  // namespace

extern "C" int64_t kp_format_json(const kp_json_tables *t, int32_t n_asm, const kp_asm_summary *sums, const kp_kept *kept,
                                  int32_t kept_stride, const kp_piece *pieces, int32_t piece_stride, const kp_json_columns *c,
                                  char *out, int64_t cap) {
    if (!t || !c || n_asm < 0 || (n_asm > 0 && (!sums || !kept || !pieces || !c->n_ctg || !c->ctg_text_len)) || cap < 0 || (cap > 0 && !out)) return KP_EINVAL;
    Out o{out, cap};
    for (int a = 0; a < n_asm; ++a) {
        const kp_asm_summary &s = sums[a];
        const kp_kept *k = kept + (size_t)a * (size_t)kept_stride;
        const kp_piece *pc = pieces + (size_t)a * (size_t)piece_stride;
        const int32_t *order = c->piece_order + (size_t)a * (size_t)piece_stride;
        const int n = s.n_kept, np = s.n_pieces, best = c->best_locus[a];
        const uint8_t *seqs = c->ctg_seqs[a];
        const int32_t *ctg_off = c->ctg_off[a];
        if (!records_in_range(t, c, a, s, k, kept_stride, pc, piece_stride, order, true)) return KP_EINVAL;
        o.put(t->head, t->head_len);  // {"kaptive_version":...,"database_taxon":N,"genome":
        o.put(c->asm_ids + c->asm_id_off[a], c->asm_id_off[a + 1] - c->asm_id_off[a]);
        o.lit(",\"best_locus_idx\":"); o.num(best);
        o.lit(",\"best_locus_name\":"); o.put(t->locus_names + t->locus_name_off[best], t->locus_name_off[best + 1] - t->locus_name_off[best]);
        o.lit(",\"best_locus_score\":"); o.f64(c->best_score[a]);
        o.lit(",\"best_locus_completeness\":"); o.f64(c->completeness[a]);
        o.lit(",\"length_discrepancy\":"); o.f64(c->length_discrepancy[a]);
        o.lit(",\"percent_identity\":"); o.f64(c->identity[a]);
        o.lit(",\"percent_coverage\":"); o.f64(c->coverage[a]);
        o.lit(",\"phenotype\":"); o.put(c->phenotypes + c->phenotype_off[a], c->phenotype_off[a + 1] - c->phenotype_off[a]);
        o.lit(",\"typeable\":"); o.lit(c->typeable[a] ? "true" : "false");
        o.lit(",\"missing_expected_genes\":[");
        {
            const int g0 = t->locus_gene_off[best], ng = t->locus_gene_len[best] < KP_MAX_LOCUS_GENES ? t->locus_gene_len[best] : KP_MAX_LOCUS_GENES;
            bool first = true;
            for (int j = 0; j < ng; ++j) {
                if (!((s.missing_mask[j >> 6] >> (j & 63)) & 1ull)) continue;
                if (!first) o.put(',');
                first = false;
                o.put(t->gene_names + t->gene_name_off[g0 + j], t->gene_name_off[g0 + j + 1] - t->gene_name_off[g0 + j]);
            }
        }
        o.lit("],\"problems\":"); o.num(c->problems[a]);
        // locus pieces, in the order of their mean expected positions (numpy's argsort: the caller's)
        o.lit(",\"locus_pieces\":{\"ctg_indices\":[");
        for (int i = 0; i < np; ++i) { if (i) o.put(','); o.num(pc[order[i]].contig); }
        o.lit("],\"starts\":[");
        for (int i = 0; i < np; ++i) { if (i) o.put(','); o.num(pc[order[i]].start); }
        o.lit("],\"ends\":[");
        for (int i = 0; i < np; ++i) { if (i) o.put(','); o.num(pc[order[i]].end); }
        o.lit("],\"strands\":[");
        for (int i = 0; i < np; ++i) { if (i) o.put(','); o.num(pc[order[i]].strand); }
        o.lit("]},");
        // gene hits (GeneHits.to_dict: the numeric columns in declaration order, then the three text columns)
        int_array(o, "\"gene_hits\":{\"gene_indices\":", k, n, [](const kp_kept &h) { return h.gene; });
        int_array(o, ",\"q_starts\":", k, n, [](const kp_kept &h) { return h.q_start; });
        int_array(o, ",\"q_ends\":", k, n, [](const kp_kept &h) { return h.q_end; });
        int_array(o, ",\"t_indices\":", k, n, [](const kp_kept &h) { return h.contig; });
        int_array(o, ",\"t_starts\":", k, n, [](const kp_kept &h) { return h.t_start; });
        int_array(o, ",\"t_ends\":", k, n, [](const kp_kept &h) { return h.t_end; });
        int_array(o, ",\"strands\":", k, n, [](const kp_kept &h) { return (int)h.strand; });
        bool_array(o, ",\"is_expected\":", k, n, [](const kp_kept &h) { return (h.flags & KP_F_EXPECTED) != 0; });
        bool_array(o, ",\"is_inside\":", k, n, [](const kp_kept &h) { return (h.flags & KP_F_INSIDE) != 0; });
        bool_array(o, ",\"is_extra\":", k, n, [](const kp_kept &h) { return (h.flags & KP_F_EXTRA) != 0; });
        int_array(o, ",\"expected_positions\":", k, n, [t](const kp_kept &h) { return t->gene_position[h.gene]; });
        int_array(o, ",\"expected_strands\":", k, n, [t](const kp_kept &h) { return (int)t->gene_strand[h.gene]; });
        o.lit(",\"coverages\":[");
        {
            bool first = true;
            for (int i = 0; i < n; ++i) {
                if (!alive(k[i])) continue;
                if (!first) o.put(',');
                first = false;
                o.f32(k[i].coverage);
            }
        }
        o.put(']');
        text_array(o, ",\"gene_ids\":", k, n, t->gene_ids, t->gene_id_off);
        text_array(o, ",\"cluster_names\":", k, n, t->cluster_names, t->cluster_name_off);
        text_array(o, ",\"product_descriptions\":", k, n, t->products, t->product_off);
        int_array(o, "},\"gene_states\":", k, n, [](const kp_kept &h) { return (int)h.state; });
        o.lit(",\"protein_identities\":[");
        {
            bool first = true;
            for (int i = 0; i < n; ++i) {
                if (!alive(k[i])) continue;
                if (!first) o.put(',');
                first = false;
                o.f32(k[i].pident);
            }
        }
        o.put(']');
        // locus sequences: ids "<contig>_<start>_<end>_<strand>", text, offsets, lengths
        o.lit(",\"locus_seqs\":{\"ids\":[");
        for (int i = 0; i < np; ++i) {
            const kp_piece &q = pc[order[i]];
            const int64_t id = (int64_t)a * piece_stride + order[i];
            if (i) o.put(',');
            o.put('"');
            o.put(c->piece_ctg_names + c->piece_ctg_name_off[id], c->piece_ctg_name_off[id + 1] - c->piece_ctg_name_off[id]);
            o.put('_'); o.num(q.start); o.put('_'); o.num(q.end); o.put('_'); o.num(q.strand);
            o.put('"');
        }
        o.lit("],\"seqs\":\"");
        for (int i = 0; i < np; ++i) {
            const kp_piece &q = pc[order[i]];
            put_extract(o, seqs, ctg_off[q.contig], q.start, q.end, q.strand, t->comp_map);
        }
        o.lit("\",\"offsets\":[");
        { long long at = 0; for (int i = 0; i < np; ++i) { if (i) o.put(','); o.num(at); at += pc[order[i]].end - pc[order[i]].start; } }
        o.lit("],\"lengths\":[");
        for (int i = 0; i < np; ++i) { if (i) o.put(','); o.num(pc[order[i]].end - pc[order[i]].start); }
        o.lit("]}");
        // gene sequences and their translations (frame (-q_start) mod 3, up to the first stop: Sequences.translate)
        text_array(o, ",\"gene_seqs\":{\"ids\":", k, n, t->gene_names, t->gene_name_off);
        o.lit(",\"seqs\":\"");
        for (int i = 0; i < n; ++i)
            if (alive(k[i])) put_extract(o, seqs, ctg_off[k[i].contig], k[i].t_start, k[i].t_end, k[i].strand, t->comp_map);
        o.lit("\",\"offsets\":[");
        {
            long long at = 0;
            bool first = true;
            for (int i = 0; i < n; ++i) {
                if (!alive(k[i])) continue;
                if (!first) o.put(',');
                first = false;
                o.num(at);
                at += k[i].t_end - k[i].t_start;
            }
        }
        int_array(o, "],\"lengths\":", k, n, [](const kp_kept &h) { return h.t_end - h.t_start; });
        text_array(o, "},\"translations\":{\"ids\":", k, n, t->gene_names, t->gene_name_off);
        o.lit(",\"seqs\":\"");
        int32_t plen[4096];
        int n_alive = 0;
        for (int i = 0; i < n; ++i) {
            if (!alive(k[i])) continue;
            const kp_kept &h = k[i];
            const int len = h.t_end - h.t_start, frame = ((-h.q_start) % 3 + 3) % 3;
            const int64_t base = ctg_off[h.contig];
            int count = 0;
            for (int p = frame; p + 3 <= len; p += 3) {
                uint8_t b[3];
                for (int z = 0; z < 3; ++z)
                    b[z] = h.strand >= 0 ? seqs[base + h.t_start + p + z] : t->comp_map[seqs[base + h.t_end - 1 - (p + z)]];
                const uint8_t aa = t->codon_map[t->char_map[b[0]] * 25 + t->char_map[b[1]] * 5 + t->char_map[b[2]]];
                if (aa == 42) break;  // '*': the stop is not part of the protein
                o.put((char)aa);
                ++count;
            }
            if (n_alive < 4096) plen[n_alive] = count;
            ++n_alive;
        }
        o.lit("\",\"offsets\":[");
        { long long at = 0; for (int i = 0; i < n_alive; ++i) { if (i) o.put(','); o.num(at); at += plen[i]; } }
        o.lit("],\"lengths\":[");
        for (int i = 0; i < n_alive; ++i) { if (i) o.put(','); o.num(plen[i]); }
        o.lit("]}}\n");
    }
    return o.n;
}

// The per-assembly FASTA outputs of `kaptive assembly` (-l / -g / -p; reference: src/kaptive/serotyping/cli.py:78-114 writes
// result.locus_seqs / gene_seqs / translations .to_fasta() to a file per assembly): the records of every assembly of the batch,
// ">id\nsequence\n" each (Sequences.to_fasta, core/seq.py), back to back; asm_end[a] = where assembly a's records end.
// kind 0: the locus pieces in the order of their mean positions, ids "<contig>_<start>_<end>_<strand>" (names[a * piece_stride
// + p] = the piece's contig); 1: the kept genes' sequences; 2: their translations (frame (-q_start) mod 3, up to the first stop)
// -- ids = names[gene] for both.  Names are raw bytes here (not JSON strings).
extern "C" int64_t kp_format_fasta(const kp_json_tables *t, int32_t n_asm, const kp_asm_summary *sums, const kp_kept *kept,
                                   int32_t kept_stride, const kp_piece *pieces, int32_t piece_stride, const kp_json_columns *c,
                                   int32_t kind, const char *names, const int64_t *name_off, char *out, int64_t cap, int64_t *asm_end) {
    if (!t || !c || !names || !name_off || !asm_end || kind < 0 || kind > 2 || n_asm < 0 ||
        (n_asm > 0 && (!sums || !kept || !pieces || !c->n_ctg || !c->ctg_text_len)) || cap < 0 || (cap > 0 && !out))
        return KP_EINVAL;
    Out o{out, cap};
    for (int a = 0; a < n_asm; ++a) {
        const kp_asm_summary &s = sums[a];
        const kp_kept *k = kept + (size_t)a * (size_t)kept_stride;
        const kp_piece *pc = pieces + (size_t)a * (size_t)piece_stride;
        const int32_t *order = c->piece_order + (size_t)a * (size_t)piece_stride;
        const uint8_t *seqs = c->ctg_seqs[a];
        const int32_t *ctg_off = c->ctg_off[a];
        if (!records_in_range(t, c, a, s, k, kept_stride, pc, piece_stride, order, false)) return KP_EINVAL;
        if (kind == 0) {
            for (int i = 0; i < s.n_pieces; ++i) {
                const kp_piece &q = pc[order[i]];
                const int64_t id = (int64_t)a * piece_stride + order[i];
                o.put('>');
                o.put(names + name_off[id], name_off[id + 1] - name_off[id]);
                o.put('_'); o.num(q.start); o.put('_'); o.num(q.end); o.put('_'); o.num(q.strand);
                o.put('\n');
                put_extract(o, seqs, ctg_off[q.contig], q.start, q.end, q.strand, t->comp_map);
                o.put('\n');
            }
        } else {
            for (int i = 0; i < s.n_kept; ++i) {
                if (!alive(k[i])) continue;
                const kp_kept &h = k[i];
                o.put('>');
                o.put(names + name_off[h.gene], name_off[h.gene + 1] - name_off[h.gene]);
                o.put('\n');
                if (kind == 1) {
                    put_extract(o, seqs, ctg_off[h.contig], h.t_start, h.t_end, h.strand, t->comp_map);
                } else {
                    const int len = h.t_end - h.t_start, frame = ((-h.q_start) % 3 + 3) % 3;
                    const int64_t base = ctg_off[h.contig];
                    for (int p = frame; p + 3 <= len; p += 3) {
                        uint8_t b[3];
                        for (int z = 0; z < 3; ++z)
                            b[z] = h.strand >= 0 ? seqs[base + h.t_start + p + z] : t->comp_map[seqs[base + h.t_end - 1 - (p + z)]];
                        const uint8_t aa = t->codon_map[t->char_map[b[0]] * 25 + t->char_map[b[1]] * 5 + t->char_map[b[2]]];
                        if (aa == 42) break;
                        o.put((char)aa);
                    }
                }
                o.put('\n');
            }
        }
        asm_end[a] = o.n;
    }
    return o.n;
}

