// kp_fasta.cpp -- FASTA text -> the packed assembly layout of include/kp_spec.h, in one pass, on the host.
//
// Stands in for rammappy.fasta.parse_fasta_bytes + Sequences.from_records + the per-contig byte copies the reference
// makes to feed its aligner (src/kaptive/core/genome.py:35-46,188; src/kaptive/core/seq.py:281-325): SURVEY.md section 8
// row (f1).  Record name = first word of the header line; sequence = every following line up to the next '>' with
// whitespace removed; A C G T/U (either case) -> 0..3, anything else is an N run.  No GPU involved; ctypes releases the
// GIL, so callers pack many files from a thread pool.
#include <zlib.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/kaptive_amd.h"

namespace {

struct Tables {
    uint8_t code[256];  // 0..3 = A C G T/U (either case), 4 = any other symbol (part of an N run), 8 = whitespace
    Tables() {
        std::memset(code, 4, sizeof code);
        const char *acgt = "ACGT";
        for (int i = 0; i < 4; ++i) { code[(uint8_t)acgt[i]] = (uint8_t)i; code[(uint8_t)acgt[i] + 32] = (uint8_t)i; }
        code['U'] = code['u'] = 3;
        for (int c : {9, 10, 11, 12, 13, 32}) code[c] = 8;
    }
};
const Tables T;

// 2-bit writer: bases go through a 64-bit accumulator, whole words leave it
struct Packer {
    uint32_t *wp;
    uint64_t acc = 0;
    int nb = 0;       // bits waiting in acc (even, < 32)
    int64_t pos = 0;  // bases written (padded space)
    explicit Packer(uint32_t *w) : wp(w) {}
    inline void put(uint32_t code) {
        acc |= (uint64_t)code << nb;
        nb += 2;
        if (nb == 32) { *wp++ = (uint32_t)acc; acc = 0; nb = 0; }
        ++pos;
    }
    inline void put16(uint32_t sixteen) {  // 16 bases at once
        acc |= (uint64_t)sixteen << nb;
        *wp++ = (uint32_t)acc;
        acc >>= 32;
        pos += 16;
    }
    inline void pad_to(int64_t align) {
        while (pos % align) put(0);
    }
};

// gzip / zlib stream(s) -> bytes (concatenated gzip members are read through, as gzip.open does)
int inflate_all(const uint8_t *data, int64_t n, std::vector<uint8_t> &out) {
    out.clear();
    out.reserve((size_t)n * 4 + 1024);
    int64_t at = 0;
    while (at < n) {
        z_stream z;
        std::memset(&z, 0, sizeof z);
        if (inflateInit2(&z, 15 + 32) != Z_OK) return KP_ENOMEM;  // + 32: gzip or zlib header, detected
        z.next_in = const_cast<Bytef *>(data + at);
        z.avail_in = (uInt)std::min<int64_t>(n - at, 1 << 30);
        int rc = Z_OK;
        while (rc != Z_STREAM_END) {
            const size_t have = out.size();
            out.resize(have + std::max<size_t>(1 << 20, have / 2));
            z.next_out = out.data() + have;
            z.avail_out = (uInt)std::min<size_t>(out.size() - have, 1u << 30);
            const size_t room = z.avail_out;
            rc = inflate(&z, Z_NO_FLUSH);
            out.resize(have + (room - z.avail_out));
            if (rc != Z_OK && rc != Z_STREAM_END && !(rc == Z_BUF_ERROR && z.avail_in > 0)) { inflateEnd(&z); return KP_EINVAL; }
            if (rc == Z_BUF_ERROR && z.avail_in == 0) { inflateEnd(&z); return KP_EINVAL; }  // truncated input
        }
        at = (int64_t)(z.next_in - data);
        inflateEnd(&z);
        while (at < n && data[at] == 0) ++at;  // zero padding between / after members
    }
    return KP_OK;
}

int pack_text(const uint8_t *data, int64_t n, bool keep_text, kp_packed_fasta **out);

}  // namespace

extern "C" {

int kp_fasta_pack(const uint8_t *data, int64_t n, kp_packed_fasta **out) {
    if (!out || (n > 0 && !data) || n < 0) return KP_EINVAL;
    *out = nullptr;
    return pack_text(data, n, false, out);
}

int kp_fasta_ingest(const uint8_t *data, int64_t n, int32_t flags, kp_packed_fasta **out) {
    if (!out || (n > 0 && !data) || n < 0) return KP_EINVAL;
    *out = nullptr;
    if (flags & KP_FASTA_GZIP) {
        std::vector<uint8_t> text;
        const int rc = inflate_all(data, n, text);
        if (rc) return rc;
        return pack_text(text.data(), (int64_t)text.size(), (flags & KP_FASTA_KEEP_TEXT) != 0, out);
    }
    return pack_text(data, n, (flags & KP_FASTA_KEEP_TEXT) != 0, out);
}

}  // extern "C"

namespace {

int pack_text(const uint8_t *data, int64_t n, bool keep_text, kp_packed_fasta **out) {
    // every '>' may open a contig (32-base alignment: at most two more words each); sequence bytes give <= n bases
    size_t marks = 0;
    for (const uint8_t *p = data, *e = data + n; p < e && (p = (const uint8_t *)std::memchr(p, '>', (size_t)(e - p))); ++p) ++marks;
    std::vector<uint32_t> words((size_t)n / 16 + 2 * marks + 8, 0u);
    std::vector<int32_t> ctg_start, ctg_len, runs, name_off;
    std::string names;
    std::vector<uint8_t> dense;  // keep_text: the contigs' symbols as written (whitespace removed), back to back
    if (keep_text) dense.reserve((size_t)n);
    Packer pk(words.data());
    int64_t i = 0;
    while (i < n && data[i] != '>') {  // text before the first header is ignored
        const uint8_t *nl = (const uint8_t *)std::memchr(data + i, '\n', (size_t)(n - i));
        i = nl ? (nl - data) + 1 : n;
    }
    bool in_run = false;
    auto slow = [&](const uint8_t *p, const uint8_t *e) {  // symbol by symbol: whitespace dropped, N runs recorded
        for (; p < e; ++p) {
            const uint8_t c = T.code[*p];
            if (c == 8) continue;
            if (keep_text) dense.push_back(*p);
            if (c == 4) {
                if (!in_run) { runs.push_back((int32_t)pk.pos); runs.push_back((int32_t)pk.pos); in_run = true; }
                runs.back() = (int32_t)pk.pos + 1;
                pk.put(0);
            } else {
                in_run = false;
                pk.put(c);
            }
        }
    };
    while (i < n) {
        // header line: the record's name is its first word
        int64_t j = i + 1;
        while (j < n && T.code[data[j]] != 8) ++j;
        name_off.push_back((int32_t)names.size());
        names.append((const char *)data + i + 1, (size_t)(j - i - 1));
        const uint8_t *nl = (const uint8_t *)std::memchr(data + j, '\n', (size_t)(n - j));
        i = nl ? (nl - data) + 1 : n;
        // sequence lines up to the next line that starts with '>'
        pk.pad_to(KP_CONTIG_ALIGN);
        if (pk.pos > (int64_t)KP_MAX_ASM_LEN) return KP_EINVAL;
        const int64_t start = pk.pos;
        in_run = false;
        while (i < n && data[i] != '>') {
            nl = (const uint8_t *)std::memchr(data + i, '\n', (size_t)(n - i));
            const uint8_t *p = data + i, *e = nl ? nl : data + n;
            i = nl ? (nl - data) + 1 : n;
            for (; e - p >= 16; p += 16) {  // 16 symbols at a time while they are plain bases
                uint32_t w = 0, seen = 0;
#pragma GCC unroll 16
                for (int k = 0; k < 16; ++k) {
                    const uint32_t c = T.code[p[k]];
                    seen |= c;
                    w |= (c & 3u) << (2 * k);
                }
                if (seen & 12u) slow(p, p + 16);
                else {
                    in_run = false;
                    pk.put16(w);
                    if (keep_text) dense.insert(dense.end(), p, p + 16);
                }
            }
            slow(p, e);
            if (pk.pos > (int64_t)KP_MAX_ASM_LEN) return KP_EINVAL;
        }
        ctg_start.push_back((int32_t)start);
        ctg_len.push_back((int32_t)(pk.pos - start));
    }
    name_off.push_back((int32_t)names.size());
    pk.pad_to(KP_ASM_ALIGN);
    const int64_t pos = pk.pos;
    words.resize((size_t)(pos / 16));

    kp_packed_fasta *r = new (std::nothrow) kp_packed_fasta();
    if (!r) return KP_ENOMEM;
    auto dup = [](const void *src, size_t bytes) -> void * {
        void *p = std::malloc(bytes ? bytes : 1);
        if (p && bytes) std::memcpy(p, src, bytes);
        return p;
    };
    r->padded_len = pos;
    r->n_contigs = (int32_t)ctg_start.size();
    r->n_runs = (int32_t)(runs.size() / 2);
    r->words = (uint32_t *)dup(words.data(), words.size() * 4);
    r->ctg_start = (int32_t *)dup(ctg_start.data(), ctg_start.size() * 4);
    r->ctg_len = (int32_t *)dup(ctg_len.data(), ctg_len.size() * 4);
    r->n_run_pairs = (int32_t *)dup(runs.data(), runs.size() * 4);
    r->names = (char *)dup(names.data(), names.size());
    r->name_off = (int32_t *)dup(name_off.data(), name_off.size() * 4);
    r->seqs = keep_text ? (uint8_t *)dup(dense.data(), dense.size()) : nullptr;
    r->n_seq_bytes = keep_text ? (int64_t)dense.size() : 0;
    if (!r->words || !r->ctg_start || !r->ctg_len || !r->n_run_pairs || !r->names || !r->name_off || (keep_text && !r->seqs)) {
        kp_fasta_free(r);
        return KP_ENOMEM;
    }
    *out = r;
    return KP_OK;
}

}  // namespace

extern "C" {

// Contigs that are already in memory (one byte per base, back to back) -> the same packed layout; names stay with the
// caller.  Every byte is a base here: whatever is not A C G T/U belongs to an N run.
int kp_pack_contigs(const uint8_t *seqs, const int64_t *offsets, const int32_t *lengths, int32_t n_contigs,
                    kp_packed_fasta **out) {
    if (!out || n_contigs < 0 || (n_contigs > 0 && (!offsets || !lengths))) return KP_EINVAL;
    *out = nullptr;
    size_t total = 0;
    for (int32_t c = 0; c < n_contigs; ++c) {
        if (lengths[c] < 0 || offsets[c] < 0 || (lengths[c] > 0 && !seqs)) return KP_EINVAL;
        total += (size_t)lengths[c];
    }
    std::vector<uint32_t> words(total / 16 + 2 * (size_t)n_contigs + 8, 0u);
    std::vector<int32_t> ctg_start((size_t)n_contigs), ctg_len((size_t)n_contigs), runs;
    Packer pk(words.data());
    for (int32_t c = 0; c < n_contigs; ++c) {
        pk.pad_to(KP_CONTIG_ALIGN);
        if (pk.pos + lengths[c] > (int64_t)KP_MAX_ASM_LEN) return KP_EINVAL;
        ctg_start[(size_t)c] = (int32_t)pk.pos;
        ctg_len[(size_t)c] = lengths[c];
        bool in_run = false;
        const uint8_t *p = seqs + offsets[c], *e = p + lengths[c];
        auto slow = [&](const uint8_t *q, const uint8_t *qe) {
            for (; q < qe; ++q) {
                const uint8_t code = T.code[*q];
                if (code > 3) {
                    if (!in_run) { runs.push_back((int32_t)pk.pos); runs.push_back((int32_t)pk.pos); in_run = true; }
                    runs.back() = (int32_t)pk.pos + 1;
                    pk.put(0);
                } else {
                    in_run = false;
                    pk.put(code);
                }
            }
        };
        for (; e - p >= 16; p += 16) {
            uint32_t w = 0, seen = 0;
#pragma GCC unroll 16
            for (int k = 0; k < 16; ++k) {
                const uint32_t code = T.code[p[k]];
                seen |= code;
                w |= (code & 3u) << (2 * k);
            }
            if (seen & 12u) slow(p, p + 16);
            else { in_run = false; pk.put16(w); }
        }
        slow(p, e);
    }
    pk.pad_to(KP_ASM_ALIGN);
    words.resize((size_t)(pk.pos / 16));
    kp_packed_fasta *r = new (std::nothrow) kp_packed_fasta();
    if (!r) return KP_ENOMEM;
    auto dup = [](const void *src, size_t bytes) -> void * {
        void *q = std::malloc(bytes ? bytes : 1);
        if (q && bytes) std::memcpy(q, src, bytes);
        return q;
    };
    r->padded_len = pk.pos;
    r->n_contigs = n_contigs;
    r->n_runs = (int32_t)(runs.size() / 2);
    r->words = (uint32_t *)dup(words.data(), words.size() * 4);
    r->ctg_start = (int32_t *)dup(ctg_start.data(), ctg_start.size() * 4);
    r->ctg_len = (int32_t *)dup(ctg_len.data(), ctg_len.size() * 4);
    r->n_run_pairs = (int32_t *)dup(runs.data(), runs.size() * 4);
    r->names = (char *)dup("", 0);
    r->seqs = nullptr;
    r->n_seq_bytes = 0;
    const int32_t zero = 0;
    r->name_off = (int32_t *)dup(&zero, 4);
    if (!r->words || !r->ctg_start || !r->ctg_len || !r->n_run_pairs || !r->names || !r->name_off) {
        kp_fasta_free(r);
        return KP_ENOMEM;
    }
    *out = r;
    return KP_OK;
}

void kp_fasta_free(kp_packed_fasta *p) {
    if (!p) return;
    std::free(p->words); std::free(p->ctg_start); std::free(p->ctg_len); std::free(p->n_run_pairs);
    std::free(p->names); std::free(p->name_off); std::free(p->seqs);
    delete p;
}

}  // extern "C"
