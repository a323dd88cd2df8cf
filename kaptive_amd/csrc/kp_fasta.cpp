// kp_fasta.cpp -- FASTA text -> the packed assembly layout of include/kp_spec.h, in one pass, on the host.
//
// Stands in for rammappy.fasta.parse_fasta_bytes + Sequences.from_records + the per-contig byte copies the reference
// makes to feed its aligner (src/kaptive/core/genome.py:35-46,188; src/kaptive/core/seq.py:281-325): SURVEY.md section 8
// row (f1).  Record name = first word of the header line; sequence = every following line up to the next '>' with
// whitespace removed; A C G T/U (either case) -> 0..3, anything else is an N run.  No GPU involved; ctypes releases the
// GIL, so callers pack many files from a thread pool.
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/kaptive_amd.h"

namespace {

struct Tables {
    uint8_t code[256];
    bool space[256];
    Tables() {
        std::memset(code, 4, sizeof code);
        const char *acgt = "ACGT";
        for (int i = 0; i < 4; ++i) { code[(uint8_t)acgt[i]] = (uint8_t)i; code[(uint8_t)acgt[i] + 32] = (uint8_t)i; }
        code['U'] = code['u'] = 3;
        std::memset(space, 0, sizeof space);
        for (int c : {9, 10, 11, 12, 13, 32}) space[c] = true;
    }
};
const Tables T;

}  // namespace

extern "C" {

int kp_fasta_pack(const uint8_t *data, int64_t n, kp_packed_fasta **out) {
    if (!out || (n > 0 && !data) || n < 0) return KP_EINVAL;
    *out = nullptr;
    std::vector<uint32_t> words;
    std::vector<int32_t> ctg_start, ctg_len, runs, name_off;
    std::string names;
    words.reserve((size_t)n / 16 + 64);
    int64_t pos = 0;  // position in the padded space
    auto put = [&](uint32_t code) {
        if ((pos & 15) == 0) words.push_back(0u);
        words.back() |= code << (2 * (pos & 15));
        ++pos;
    };
    int64_t i = 0;
    while (i < n && data[i] != '>') {  // text before the first header is ignored
        while (i < n && data[i] != '\n') ++i;
        ++i;
    }
    while (i < n) {
        // header line
        int64_t j = i + 1;
        while (j < n && data[j] != '\n' && !T.space[data[j]]) ++j;
        name_off.push_back((int32_t)names.size());
        names.append((const char *)data + i + 1, (size_t)(j - i - 1));
        while (j < n && data[j] != '\n') ++j;
        i = j + 1;
        // sequence lines
        while (pos % KP_CONTIG_ALIGN) put(0);
        if (pos > (int64_t)KP_MAX_ASM_LEN) return KP_EINVAL;
        const int64_t start = pos;
        bool in_run = false;
        bool line_start = true;
        while (i < n) {
            const uint8_t c = data[i];
            if (line_start && c == '>') break;
            line_start = c == '\n';
            ++i;
            if (T.space[c]) continue;
            const uint8_t code = T.code[c];
            if (code > 3) {
                if (!in_run) { runs.push_back((int32_t)pos); runs.push_back((int32_t)pos); in_run = true; }
                runs.back() = (int32_t)pos + 1;
                put(0);
            } else {
                in_run = false;
                put(code);
            }
            if (pos > (int64_t)KP_MAX_ASM_LEN) return KP_EINVAL;
        }
        ctg_start.push_back((int32_t)start);
        ctg_len.push_back((int32_t)(pos - start));
    }
    name_off.push_back((int32_t)names.size());
    while (pos % KP_ASM_ALIGN) put(0);

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
    std::free(p->names); std::free(p->name_off);
    delete p;
}

}  // extern "C"
