// kp_fasta.cpp -- FASTA text -> the packed assembly layout of include/kp_spec.h, in one pass, on the host.
//
// Stands in for rammappy.fasta.parse_fasta_bytes + Sequences.from_records + the per-contig byte copies the reference
// makes to feed its aligner (src/kaptive/core/genome.py:35-46,188; src/kaptive/core/seq.py:281-325): SURVEY.md section 8
// row (f1).  Record name = first word of the header line; sequence = every following line up to the next '>' with
// whitespace removed; A C G T/U (either case) -> 0..3, anything else is an N run.  No GPU involved; ctypes releases the
// GIL, so callers pack many files from a thread pool.
#include <dlfcn.h>
#include <fcntl.h>
#include <immintrin.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <functional>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
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

// The two large arrays of a result (packed words, sequence text) come from a small pool of recycled blocks: a thread pool
// that ingests file after file would otherwise map, fault in and unmap ~12 MB per 5 Mbp assembly, and with many threads
// those page faults serialise in the kernel (measured: 8 threads together slower than one).
struct BlockPool {
    struct Block { void *p; size_t cap; };
    std::mutex mu;
    std::vector<Block> free_blocks;
    size_t held = 0;
    static constexpr size_t MAX_HELD = (size_t)2 << 30, MAX_BLOCKS = 512;
    void *take(size_t bytes, size_t *cap) {
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = free_blocks.size();
            for (size_t i = 0; i < free_blocks.size(); ++i)
                if (free_blocks[i].cap >= bytes && (best == free_blocks.size() || free_blocks[i].cap < free_blocks[best].cap)) best = i;
            if (best < free_blocks.size() && free_blocks[best].cap <= 2 * bytes + (1 << 20)) {
                Block b = free_blocks[best];
                free_blocks[best] = free_blocks.back();
                free_blocks.pop_back();
                held -= b.cap;
                *cap = b.cap;
                return b.p;
            }
        }
        const size_t want = bytes + bytes / 8 + 4096;  // the next file is a little longer or shorter
        *cap = want;
        return std::malloc(want);
    }
    void give(void *p, size_t cap) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (cap && held + cap <= MAX_HELD && free_blocks.size() < MAX_BLOCKS) {
                free_blocks.push_back({p, cap});
                held += cap;
                return;
            }
        }
        std::free(p);
    }
};
BlockPool g_pool;

struct Owned : kp_packed_fasta {  // what kp_fasta_free gets back: the public record plus the capacities of its pooled blocks
    size_t words_cap = 0, seqs_cap = 0;
};

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
            if (z.avail_in == 0) {  // a member with more than 1 GiB of compressed bytes: hand zlib the next slice
                const int64_t left = n - (int64_t)(z.next_in - data);
                z.avail_in = (uInt)std::min<int64_t>(left, 1 << 30);
            }
            rc = inflate(&z, Z_NO_FLUSH);
            out.resize(have + (room - z.avail_out));
            if (rc != Z_OK && rc != Z_STREAM_END && !(rc == Z_BUF_ERROR && z.avail_in > 0)) { inflateEnd(&z); return KP_EINVAL; }
            if (rc == Z_BUF_ERROR && z.avail_in == 0 && z.next_in >= data + n) { inflateEnd(&z); return KP_EINVAL; }  // truncated
        }
        at = (int64_t)(z.next_in - data);
        inflateEnd(&z);
        while (at < n && data[at] == 0) ++at;  // zero padding between / after members
    }
    return KP_OK;
}

// ---- bzip2 and xz: the image ships the shared libraries (libbz2.so.1.0, liblzma.so.5) without their headers, so the two
// stream structures are declared here from the libraries' stable, documented ABI and the entry points are looked up at
// first use.  A host without the libraries gets KP_ENOTSUP and the Python side falls back to its own bz2 / lzma modules.
struct BzStream {  // bz_stream of bzlib.h (1.0.x)
    char *next_in; unsigned avail_in, total_in_lo32, total_in_hi32;
    char *next_out; unsigned avail_out, total_out_lo32, total_out_hi32;
    void *state; void *(*bzalloc)(void *, int, int); void (*bzfree)(void *, void *); void *opaque;
};
struct LzmaStream {  // lzma_stream of lzma/base.h (5.x)
    const uint8_t *next_in; size_t avail_in; uint64_t total_in;
    uint8_t *next_out; size_t avail_out; uint64_t total_out;
    const void *allocator; void *internal;
    void *reserved_ptr1, *reserved_ptr2, *reserved_ptr3, *reserved_ptr4;
    uint64_t reserved_int1, reserved_int2; size_t reserved_int3, reserved_int4;
    int reserved_enum1, reserved_enum2;
};
struct Codecs {
    int (*bz_init)(BzStream *, int, int) = nullptr;
    int (*bz_run)(BzStream *) = nullptr;
    int (*bz_end)(BzStream *) = nullptr;
    int (*xz_decoder)(LzmaStream *, uint64_t, uint32_t) = nullptr;
    int (*xz_code)(LzmaStream *, int) = nullptr;
    void (*xz_end)(LzmaStream *) = nullptr;
    // libdeflate (whole-buffer inflate, 2.8 x zlib's pace on FASTA text): optional, zlib takes over without it
    void *(*ld_alloc)() = nullptr;
    void (*ld_free)(void *) = nullptr;
    int (*ld_gzip)(void *, const void *, size_t, void *, size_t, size_t *, size_t *) = nullptr;
    Codecs() {
        for (const char *name : {"libdeflate.so.0", "libdeflate.so"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
                ld_alloc = reinterpret_cast<decltype(ld_alloc)>(dlsym(h, "libdeflate_alloc_decompressor"));
                ld_free = reinterpret_cast<decltype(ld_free)>(dlsym(h, "libdeflate_free_decompressor"));
                ld_gzip = reinterpret_cast<decltype(ld_gzip)>(dlsym(h, "libdeflate_gzip_decompress_ex"));
                if (ld_alloc && ld_free && ld_gzip) break;
                ld_alloc = nullptr;
            }
        }
        for (const char *name : {"libbz2.so.1.0", "libbz2.so.1", "libbz2.so"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
                bz_init = reinterpret_cast<decltype(bz_init)>(dlsym(h, "BZ2_bzDecompressInit"));
                bz_run = reinterpret_cast<decltype(bz_run)>(dlsym(h, "BZ2_bzDecompress"));
                bz_end = reinterpret_cast<decltype(bz_end)>(dlsym(h, "BZ2_bzDecompressEnd"));
                if (bz_init && bz_run && bz_end) break;
                bz_init = nullptr;
            }
        }
        for (const char *name : {"liblzma.so.5", "liblzma.so"}) {
            if (void *h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
                xz_decoder = reinterpret_cast<decltype(xz_decoder)>(dlsym(h, "lzma_stream_decoder"));
                xz_code = reinterpret_cast<decltype(xz_code)>(dlsym(h, "lzma_code"));
                xz_end = reinterpret_cast<decltype(xz_end)>(dlsym(h, "lzma_end"));
                if (xz_decoder && xz_code && xz_end) break;
                xz_decoder = nullptr;
            }
        }
    }
};
const Codecs &codecs() {
    static const Codecs c;
    return c;
}

// gzip member(s) -> bytes through libdeflate when the host has it: the whole buffer at once, so the output size has to be
// guessed -- the last member's ISIZE trailer, which is the answer for the usual one-member file -- and the call repeated
// with more room when it was not enough.  Anything it does not take (zlib framing, damaged data, no library) is left to
// inflate_all, which also decides what is an error.
bool inflate_fast(const uint8_t *data, int64_t n, std::vector<uint8_t> &out) {
    const Codecs &c = codecs();
    if (!c.ld_alloc || n < 18 || data[0] != 0x1f || data[1] != 0x8b) return false;
    void *d = c.ld_alloc();
    if (!d) return false;
    const uint32_t isize = (uint32_t)data[n - 4] | (uint32_t)data[n - 3] << 8 | (uint32_t)data[n - 2] << 16 | (uint32_t)data[n - 1] << 24;
    size_t room = std::max<size_t>((size_t)isize + 64, (size_t)n * 2), have = 0;
    int64_t at = 0;
    bool ok = true;
    out.clear();
    while (ok && at < n) {
        if (out.size() < have + room) out.resize(have + room);
        size_t used = 0, made = 0;
        const int rc = c.ld_gzip(d, data + at, (size_t)(n - at), out.data() + have, out.size() - have, &used, &made);
        if (rc == 3 /* LIBDEFLATE_INSUFFICIENT_SPACE */) {
            room = std::max<size_t>(2 * room, (size_t)1 << 20);
            if (room > ((size_t)1 << 33)) ok = false;
            continue;
        }
        if (rc != 0 || used == 0) { ok = false; break; }
        have += made;
        at += (int64_t)used;
        while (at < n && data[at] == 0) ++at;  // zero padding between / after members
        room = (size_t)(n - at) * 4 + 64;
    }
    c.ld_free(d);
    if (ok) out.resize(have);
    return ok;
}

// bzip2 stream(s) -> bytes (concatenated streams are read through, as bz2.open does)
int bunzip_all(const uint8_t *data, int64_t n, std::vector<uint8_t> &out) {
    const Codecs &c = codecs();
    if (!c.bz_init) return KP_ENOTSUP;
    out.clear();
    int64_t at = 0;
    while (at < n) {
        BzStream z;
        std::memset(&z, 0, sizeof z);
        if (c.bz_init(&z, 0, 0) != 0) return KP_ENOMEM;
        z.next_in = reinterpret_cast<char *>(const_cast<uint8_t *>(data + at));
        int rc = 0;
        while (rc != 4 /* BZ_STREAM_END */) {
            if (z.avail_in == 0) z.avail_in = (unsigned)std::min<int64_t>(n - (reinterpret_cast<uint8_t *>(z.next_in) - data), 1 << 30);
            const size_t have = out.size();
            out.resize(have + std::max<size_t>(1 << 20, have / 2));
            z.next_out = reinterpret_cast<char *>(out.data() + have);
            z.avail_out = (unsigned)std::min<size_t>(out.size() - have, 1u << 30);
            const size_t room = z.avail_out;
            const unsigned in_before = z.avail_in;
            rc = c.bz_run(&z);
            out.resize(have + (room - z.avail_out));
            if (rc != 0 /* BZ_OK */ && rc != 4) { c.bz_end(&z); return KP_EINVAL; }
            if (rc == 0 && z.avail_in == 0 && in_before == 0 && room == z.avail_out) { c.bz_end(&z); return KP_EINVAL; }  // truncated
        }
        at = (int64_t)(reinterpret_cast<uint8_t *>(z.next_in) - data);
        c.bz_end(&z);
    }
    return KP_OK;
}

// .xz stream(s) -> bytes (LZMA_CONCATENATED, as lzma.open does)
int unxz_all(const uint8_t *data, int64_t n, std::vector<uint8_t> &out) {
    const Codecs &c = codecs();
    if (!c.xz_decoder) return KP_ENOTSUP;
    out.clear();
    if (n == 0) return KP_OK;
    LzmaStream z;
    std::memset(&z, 0, sizeof z);
    if (c.xz_decoder(&z, UINT64_MAX, 0x08u /* LZMA_CONCATENATED */) != 0) return KP_ENOMEM;
    z.next_in = data;
    z.avail_in = (size_t)n;
    int rc = 0;
    while (rc != 1 /* LZMA_STREAM_END */) {
        const size_t have = out.size();
        out.resize(have + std::max<size_t>(1 << 20, have / 2));
        z.next_out = out.data() + have;
        z.avail_out = out.size() - have;
        const size_t room = z.avail_out;
        rc = c.xz_code(&z, z.avail_in == 0 ? 3 /* LZMA_FINISH */ : 0 /* LZMA_RUN */);
        out.resize(have + (room - z.avail_out));
        if (rc != 0 && rc != 1) { c.xz_end(&z); return KP_EINVAL; }  // (LZMA_BUF_ERROR = 10 on truncated input)
    }
    c.xz_end(&z);
    return KP_OK;
}

int pack_text(const uint8_t *data, int64_t n, bool keep_text, kp_packed_fasta **out);
int detect_simd();
int simd_level();
extern std::atomic<int> g_simd;

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
    if (flags & (KP_FASTA_GZIP | KP_FASTA_BZ2 | KP_FASTA_XZ)) {
        std::vector<uint8_t> text;
        const int rc = (flags & KP_FASTA_GZIP) ? (inflate_fast(data, n, text) ? KP_OK : inflate_all(data, n, text))
                       : (flags & KP_FASTA_BZ2) ? bunzip_all(data, n, text) : unxz_all(data, n, text);
        if (rc) return rc;
        return pack_text(text.data(), (int64_t)text.size(), (flags & KP_FASTA_KEEP_TEXT) != 0, out);
    }
    return pack_text(data, n, (flags & KP_FASTA_KEEP_TEXT) != 0, out);
}

// The file itself, read into a recycled block of the library's pool: a Python caller would read it into a freshly
// allocated bytes object (as much time as the parse takes on the vector paths, page faults included).  Mapping the file
// instead was measured on the 2 x 64-core GPU host and does not scale: 3.6 / 14 / 20 / 17.5 / 15.8 GB/s on 1 / 4 / 8 / 16 /
// 64 threads (mmap and munmap take the process's mapping lock, every munmap interrupts the other threads' cores to flush
// their TLBs), against reads into per-call buffers that no other thread's address-space changes touch.
int kp_fasta_ingest_file(const char *path, int32_t flags, kp_packed_fasta **out) {
    if (!out || !path) return KP_EINVAL;
    *out = nullptr;
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return KP_EIO;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode)) {  // (a pipe or a device: the caller reads it and passes the bytes)
        close(fd);
        return KP_EIO;
    }
    size_t cap = 0;
    uint8_t *buf = (uint8_t *)g_pool.take((size_t)st.st_size + 64, &cap);
    if (!buf) {
        close(fd);
        return KP_ENOMEM;
    }
    int64_t got = 0;
    int rc = KP_OK;
    while (got < (int64_t)st.st_size) {
        const ssize_t k = read(fd, buf + got, (size_t)((int64_t)st.st_size - got));
        if (k < 0 && errno == EINTR) continue;
        if (k < 0) { rc = KP_EIO; break; }
        if (k == 0) break;  // (shorter than fstat said: take what there is)
        got += k;
    }
    close(fd);
    if (rc == KP_OK) rc = kp_fasta_ingest(buf, got, flags, out);
    g_pool.give(buf, cap);
    return rc;
}

// ---- a chunk of files -> the tables of one batch ---------------------------------------------------------------------
// What a reader that feeds a GPU does per chunk, in two calls that never hold the interpreter lock: parse the files on
// the library's threads and lay their tables out as kp_batch_create* wants them; then copy the packed words of all of
// them, back to back, into the (page-locked) buffer the caller chose once it knew the size.
struct Shard : kp_packed_shard {
    std::vector<kp_packed_fasta *> parts;
};

static void run_on_threads(int32_t n_items, int32_t threads, const std::function<void(int32_t)> &item) {
    int t = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    t = std::max(1, std::min(t, (int)n_items));
    std::atomic<int32_t> next{0};
    auto work = [&]() {
        for (int32_t i = next.fetch_add(1); i < n_items; i = next.fetch_add(1)) item(i);
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)t - 1);
    for (int k = 1; k < t; ++k) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
}

int kp_fasta_ingest_shard(const char *const *paths, const int32_t *flags, int32_t n_files, int32_t threads, kp_packed_shard **out) {
    if (!out || n_files < 0 || (n_files > 0 && (!paths || !flags))) return KP_EINVAL;
    *out = nullptr;
    Shard *sh = new (std::nothrow) Shard();
    if (!sh) return KP_ENOMEM;
    std::memset(static_cast<kp_packed_shard *>(sh), 0, sizeof(kp_packed_shard));
    sh->parts.assign((size_t)n_files, nullptr);
    sh->n_asm = n_files;
    sh->rc = (int32_t *)std::calloc((size_t)n_files + 1, 4);
    sh->asm_word_off = (int64_t *)std::calloc((size_t)n_files + 1, 8);
    sh->asm_first_ctg = (int32_t *)std::calloc((size_t)n_files + 1, 4);
    sh->asm_first_nrun = (int32_t *)std::calloc((size_t)n_files + 1, 4);
    if (!sh->rc || !sh->asm_word_off || !sh->asm_first_ctg || !sh->asm_first_nrun) { kp_shard_free(sh); return KP_ENOMEM; }
    run_on_threads(n_files, threads, [&](int32_t i) {
        sh->rc[i] = kp_fasta_ingest_file(paths[i], flags[i] & ~KP_FASTA_KEEP_TEXT, &sh->parts[(size_t)i]);
    });
    for (int32_t i = 0; i < n_files; ++i) {
        const kp_packed_fasta *f = sh->parts[(size_t)i];
        if (sh->rc[i] != KP_OK && sh->n_failed++ == 0) sh->first_failed = i;
        sh->asm_word_off[i + 1] = sh->asm_word_off[i] + (f ? f->padded_len / 16 : 0);
        sh->asm_first_ctg[i + 1] = sh->asm_first_ctg[i] + (f ? f->n_contigs : 0);
        sh->asm_first_nrun[i + 1] = sh->asm_first_nrun[i] + (f ? f->n_runs : 0);
    }
    sh->total_words = sh->asm_word_off[n_files];
    const size_t nc = (size_t)sh->asm_first_ctg[n_files], nr = (size_t)sh->asm_first_nrun[n_files];
    sh->ctg_start = (int32_t *)std::malloc(4 * nc + 4);
    sh->ctg_len = (int32_t *)std::malloc(4 * nc + 4);
    sh->n_runs = (int32_t *)std::malloc(8 * nr + 4);
    if (!sh->ctg_start || !sh->ctg_len || !sh->n_runs) { kp_shard_free(sh); return KP_ENOMEM; }
    for (int32_t i = 0; i < n_files; ++i) {
        const kp_packed_fasta *f = sh->parts[(size_t)i];
        if (!f) continue;
        std::memcpy(sh->ctg_start + sh->asm_first_ctg[i], f->ctg_start, 4 * (size_t)f->n_contigs);
        std::memcpy(sh->ctg_len + sh->asm_first_ctg[i], f->ctg_len, 4 * (size_t)f->n_contigs);
        std::memcpy(sh->n_runs + 2 * (size_t)sh->asm_first_nrun[i], f->n_run_pairs, 8 * (size_t)f->n_runs);
    }
    *out = sh;
    return KP_OK;
}

int kp_shard_words_into(kp_packed_shard *shard, uint32_t *dst, int64_t dst_words, int32_t threads) {
    if (!shard || (shard->total_words > 0 && !dst) || dst_words < shard->total_words) return KP_EINVAL;
    Shard *sh = static_cast<Shard *>(shard);
    if (sh->parts.empty() && sh->n_asm > 0) return KP_ESTATE;  // the words were handed over already
    run_on_threads(sh->n_asm, threads, [&](int32_t i) {
        kp_packed_fasta *&f = sh->parts[(size_t)i];
        if (!f) return;
        std::memcpy(dst + sh->asm_word_off[i], f->words, 4 * (size_t)(sh->asm_word_off[i + 1] - sh->asm_word_off[i]));
        kp_fasta_free(f);  // (its blocks go back to the pool for the next chunk's files)
        f = nullptr;
    });
    sh->parts.clear();
    return KP_OK;
}

void kp_shard_free(kp_packed_shard *shard) {
    if (!shard) return;
    Shard *sh = static_cast<Shard *>(shard);
    for (kp_packed_fasta *f : sh->parts) kp_fasta_free(f);
    std::free(sh->rc); std::free(sh->asm_word_off); std::free(sh->asm_first_ctg); std::free(sh->asm_first_nrun);
    std::free(sh->ctg_start); std::free(sh->ctg_len); std::free(sh->n_runs);
    delete sh;
}

int kp_fasta_simd(int32_t cap) {
    const int have = detect_simd();
    if (cap >= 0) g_simd.store(std::min(have, (int)cap), std::memory_order_relaxed);
    return simd_level();
}

int kp_fasta_ingest_many(const uint8_t *const *data, const int64_t *n, const int32_t *flags, int32_t n_files, int32_t threads,
                         kp_packed_fasta **out, int32_t *rc) {
    if (n_files < 0 || (n_files > 0 && (!data || !n || !flags || !out || !rc))) return KP_EINVAL;
    int t = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    t = std::max(1, std::min(t, (int)n_files));
    std::atomic<int32_t> next{0};
    auto work = [&]() {
        for (int32_t i = next.fetch_add(1); i < n_files; i = next.fetch_add(1)) rc[i] = kp_fasta_ingest(data[i], n[i], flags[i], &out[i]);
    };
    std::vector<std::thread> pool;
    pool.reserve((size_t)t - 1);
    for (int k = 1; k < t; ++k) pool.emplace_back(work);
    work();
    for (auto &th : pool) th.join();
    return KP_OK;
}

}  // extern "C"

namespace {

// ---- the sequence lines of one record ----------------------------------------------------------------------------------
// 64 bytes of text at a time: every byte is classed at once -- base (A C G T U, either case), whitespace, anything else --
// and the 2-bit codes of the bases come out of the bytes' own bits: code = (bit 1 ^ bit 2) + 2 * (bit 2 ^ bit 3) of the
// ASCII value ('A' 0x41 -> 0, 'C' 0x43 -> 1, 'G' 0x47 -> 2, 'T' 0x54 / 'U' 0x55 -> 3; bit 5, the case, takes no part).  The
// two bit planes leave the vector as 64-bit masks; whitespace is squeezed out of the MASKS (pext), not out of the bytes,
// and the planes are interleaved into packed words (pdep).  Line ends are not looked for at all: a chunk stops at the
// first byte that is neither base nor whitespace, which is looked at on its own (an N-run symbol, or the '>' of the next
// header when the byte before it ended a line).  One core turns ~9 GB/s of plain FASTA into words this way (scalar table
// look-ups: 2.9); which path runs is settled once from cpuid (kp_fasta_simd).
struct Masks { uint64_t base, ws, b0, b1; };  // per byte of the chunk: is a base / whitespace; code bit 0 / 1

#define KP_T512 __attribute__((target("avx512f,avx512bw,avx512vl,avx512vbmi2,avx2,bmi,bmi2,popcnt,lzcnt")))
#define KP_T256 __attribute__((target("avx2,bmi,bmi2,popcnt,lzcnt")))

alignas(16) const uint8_t BASE_BY_NIBBLE[16] = {0xFF, 'A', 0xFF, 'C', 'T', 'U', 0xFF, 'G', 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};

KP_T512 inline Masks classify512(__m512i v) {
    const __m512i lut = _mm512_broadcast_i32x4(_mm_load_si128((const __m128i *)BASE_BY_NIBBLE));
    // the upper-case letter a base with this low nibble would be (bytes >= 0x80 look up 0): equal to the byte's own
    // upper case exactly for the ten base symbols
    const __m512i up = _mm512_and_si512(v, _mm512_set1_epi8((char)0xDF));
    const __m512i t = _mm512_xor_si512(v, _mm512_srli_epi16(v, 1));  // bits 1 and 2: the code (bit 7 takes a neighbour's bit: unused)
    Masks m;
    m.base = _mm512_cmpeq_epi8_mask(_mm512_shuffle_epi8(lut, v), up);
    m.ws = _mm512_cmple_epu8_mask(_mm512_sub_epi8(v, _mm512_set1_epi8(9)), _mm512_set1_epi8(4)) | _mm512_cmpeq_epi8_mask(v, _mm512_set1_epi8(32));
    m.b0 = _mm512_test_epi8_mask(t, _mm512_set1_epi8(2));
    m.b1 = _mm512_test_epi8_mask(t, _mm512_set1_epi8(4));
    return m;
}

KP_T256 inline void classify256(__m256i v, uint32_t &base, uint32_t &ws, uint32_t &b0, uint32_t &b1) {
    const __m256i lut = _mm256_broadcastsi128_si256(_mm_load_si128((const __m128i *)BASE_BY_NIBBLE));
    const __m256i up = _mm256_and_si256(v, _mm256_set1_epi8((char)0xDF));
    const __m256i t = _mm256_xor_si256(v, _mm256_srli_epi16(v, 1));
    const __m256i d = _mm256_sub_epi8(v, _mm256_set1_epi8(9));  // 9..13 -> 0..4: (d min 4) == d
    base = (uint32_t)_mm256_movemask_epi8(_mm256_cmpeq_epi8(_mm256_shuffle_epi8(lut, v), up));
    ws = (uint32_t)_mm256_movemask_epi8(_mm256_or_si256(_mm256_cmpeq_epi8(_mm256_min_epu8(d, _mm256_set1_epi8(4)), d),
                                                        _mm256_cmpeq_epi8(v, _mm256_set1_epi8(32))));
    b0 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(t, 6));  // bit 1 of every byte -> its bit 7
    b1 = (uint32_t)_mm256_movemask_epi8(_mm256_slli_epi16(t, 5));
}

// 2-bit writer for the vector paths: up to 32 bases per call through a 64-bit accumulator, 64 bits leave it at a time
struct Packer64 {
    uint32_t *wp;
    uint64_t acc = 0;
    int nb = 0;       // bits waiting in acc (even, < 64)
    int64_t pos = 0;  // bases written (padded space)
    explicit Packer64(uint32_t *w) : wp(w) {}
    inline void put_n(uint64_t bits, int n_bases) {  // bits: 2 * n_bases valid bits, zero above; n_bases <= 32
        acc |= bits << nb;
        int tot = nb + 2 * n_bases;
        if (tot >= 64) {
            std::memcpy(wp, &acc, 8);
            wp += 2;
            acc = nb ? bits >> (64 - nb) : 0;
            tot -= 64;
        }
        nb = tot;
        pos += n_bases;
    }
    inline void put(uint32_t code) { put_n(code, 1); }
    inline void pad_to(int64_t align) {
        for (int64_t k = (align - pos % align) % align; k > 0;) {
            const int t = k > 32 ? 32 : (int)k;
            put_n(0, t);
            k -= t;
        }
    }
    inline void flush() {  // (pos is a multiple of 16 here)
        if (nb) { std::memcpy(wp, &acc, 4); wp += 1; acc = 0; nb = 0; }
    }
};

struct Body {  // what the body loops share with pack_text
    Packer64 &pk;
    std::vector<int32_t> &runs;
    uint8_t *&dp;  // sequence text kept (nullptr: not kept)
    bool in_run = false;
};

// one symbol that is not a base (nor whitespace): part of an N run
inline void put_other(Body &b, uint8_t c) {
    if (!b.in_run) { b.runs.push_back((int32_t)b.pk.pos); b.runs.push_back((int32_t)b.pk.pos); b.in_run = true; }
    b.runs.back() = (int32_t)b.pk.pos + 1;
    b.pk.put(0);
    if (b.dp) *b.dp++ = c;
}

// the first `take` bytes of a classified chunk (all of them bases or whitespace) -> packed words (+ text)
KP_T256 inline void emit_bits(Body &b, const Masks &m, int take) {
    const uint64_t keep = m.base & (take >= 64 ? ~0ull : ((1ull << take) - 1ull));
    const int n_bases = (int)_mm_popcnt_u64(keep);
    if (!n_bases) return;
    b.in_run = false;
    const uint64_t c0 = _pext_u64(m.b0, keep), c1 = _pext_u64(m.b1, keep);
    const uint64_t lo = _pdep_u64(c0, 0x5555555555555555ull) | _pdep_u64(c1, 0xAAAAAAAAAAAAAAAAull);
    if (n_bases <= 32) {
        b.pk.put_n(lo, n_bases);
    } else {
        b.pk.put_n(lo, 32);
        b.pk.put_n(_pdep_u64(c0 >> 32, 0x5555555555555555ull) | _pdep_u64(c1 >> 32, 0xAAAAAAAAAAAAAAAAull), n_bases - 32);
    }
}

// Sequence lines from data[i] (a line start) up to the next line that starts with '>' or the end of the text; returns
// the index it stopped at.  LEVEL 2: AVX-512 (BW, VBMI2), 1: AVX2; both need BMI2.
// symbols that are not bases, and the whitespace between them, one at a time: until a base or a header turns up.  A '>' opens
// a header iff it is the first byte of a line: the body's first byte (`from`), or the byte after a '\n'.
inline int64_t scalar_stretch(Body &b, const uint8_t *data, int64_t i, int64_t n, int64_t from, bool &header) {
    header = false;
    while (i < n) {
        const uint8_t c = data[i], code = T.code[c];
        if (code == 8) { ++i; continue; }
        if (c == '>' && (i == from || data[i - 1] == '\n')) { header = true; break; }
        if (code < 4) break;
        put_other(b, c);
        ++i;
    }
    return i;
}

KP_T512 int64_t body512(Body &b, const uint8_t *data, int64_t i, int64_t n) {
    const int64_t from = i;
    bool header = false;
    while (i < n) {
        if (b.pk.pos > (int64_t)KP_MAX_ASM_LEN) return -1;
        const int64_t left = n - i;
        const __m512i v = left >= 64 ? _mm512_loadu_si512((const void *)(data + i))
                                     : _mm512_maskz_loadu_epi8((1ull << left) - 1ull, data + i);
        const Masks m = classify512(v);
        const uint64_t bad = ~(m.base | m.ws);  // (bytes past the end of the text read as 0: bad)
        const int take = bad ? (int)_tzcnt_u64(bad) : 64;
        if (take) {
            if (b.dp) {
                const uint64_t keep = m.base & (take >= 64 ? ~0ull : ((1ull << take) - 1ull));
                _mm512_storeu_si512((void *)b.dp, _mm512_maskz_compress_epi8(keep, v));  // (the block has 64 bytes of slack)
                b.dp += _mm_popcnt_u64(keep);
            }
            emit_bits(b, m, take);
            i += take;
        }
        if (take < 64 && i < n) {
            i = scalar_stretch(b, data, i, n, from, header);
            if (header) break;
        }
    }
    return i;
}

KP_T256 int64_t body256(Body &b, const uint8_t *data, int64_t i, int64_t n) {
    const int64_t from = i;
    bool header = false;
    while (i < n) {
        if (b.pk.pos > (int64_t)KP_MAX_ASM_LEN) return -1;
        int take = 0;
        if (n - i >= 64) {
            Masks m;
            uint32_t a[4], c[4];
            classify256(_mm256_loadu_si256((const __m256i *)(data + i)), a[0], a[1], a[2], a[3]);
            classify256(_mm256_loadu_si256((const __m256i *)(data + i + 32)), c[0], c[1], c[2], c[3]);
            m.base = a[0] | (uint64_t)c[0] << 32; m.ws = a[1] | (uint64_t)c[1] << 32;
            m.b0 = a[2] | (uint64_t)c[2] << 32; m.b1 = a[3] | (uint64_t)c[3] << 32;
            const uint64_t bad = ~(m.base | m.ws);
            take = bad ? (int)_tzcnt_u64(bad) : 64;
            if (take) {
                if (b.dp) {  // the text without its whitespace: runs of bases copied as they lie
                    uint64_t keep = m.base & (take >= 64 ? ~0ull : ((1ull << take) - 1ull));
                    while (keep) {
                        const int s = (int)_tzcnt_u64(keep);
                        const uint64_t rest = ~(keep >> s);
                        const int len = rest ? (int)_tzcnt_u64(rest) : 64 - s;
                        std::memcpy(b.dp, data + i + s, (size_t)len);
                        b.dp += len;
                        keep = s + len >= 64 ? 0 : keep & ~((1ull << (s + len)) - 1ull);
                    }
                }
                emit_bits(b, m, take);
                i += take;
            }
            if (take == 64) continue;
        }
        if (i < n) {
            // the chunk's first odd symbol, or the last bytes of the text: a base goes through the packer on its own
            const uint8_t ch = data[i], code = T.code[ch];
            if (code < 4) {
                b.in_run = false;
                b.pk.put(code);
                if (b.dp) *b.dp++ = ch;
                ++i;
                continue;
            }
            i = scalar_stretch(b, data, i, n, from, header);
            if (header) break;
        }
    }
    return i;
}

int64_t body_scalar(Body &b, const uint8_t *data, int64_t i, int64_t n) {
    auto slow = [&](const uint8_t *p, const uint8_t *e) {  // symbol by symbol: whitespace dropped, N runs recorded
        for (; p < e; ++p) {
            const uint8_t c = T.code[*p];
            if (c == 8) continue;
            if (c == 4) { put_other(b, *p); continue; }
            b.in_run = false;
            b.pk.put(c);
            if (b.dp) *b.dp++ = *p;
        }
    };
    while (i < n && data[i] != '>') {
        const uint8_t *nl = (const uint8_t *)std::memchr(data + i, '\n', (size_t)(n - i));
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
                b.in_run = false;
                b.pk.put_n(w, 16);
                if (b.dp) { std::memcpy(b.dp, p, 16); b.dp += 16; }
            }
        }
        slow(p, e);
        if (b.pk.pos > (int64_t)KP_MAX_ASM_LEN) return -1;
    }
    return i;
}

int detect_simd() {
    __builtin_cpu_init();
    if (!__builtin_cpu_supports("bmi2") || !__builtin_cpu_supports("avx2") || !__builtin_cpu_supports("popcnt")) return 0;
    if (__builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl") &&
        __builtin_cpu_supports("avx512vbmi2"))
        return 2;
    return 1;
}
std::atomic<int> g_simd{-1};
int simd_level() {
    int l = g_simd.load(std::memory_order_relaxed);
    if (l < 0) {
        l = detect_simd();
        if (const char *e = std::getenv("KAPTIVE_AMD_FASTA_SIMD")) l = std::min(l, std::max(0, std::atoi(e)));
        g_simd.store(l, std::memory_order_relaxed);
    }
    return l;
}

int pack_text(const uint8_t *data, int64_t n, bool keep_text, kp_packed_fasta **out) {
    // Room for the words: every symbol is at most one base and every contig starts on a 32-base boundary (two words at
    // most).  The contigs are not counted ahead (that would be a pass over the text of its own): the block starts with
    // room for one contig per 2 KB and, should a file have more, the '>' of the rest are counted then and the block grows.
    int64_t marks_left = -1;  // '>' from the current header on; -1: not counted yet
    size_t n_words_max = (size_t)n / 16 + 2 * (256 + (size_t)n / 2048) + 16;
    size_t words_cap = 0, seqs_cap = 0;
    uint32_t *words = (uint32_t *)g_pool.take(n_words_max * 4, &words_cap);
    uint8_t *dense = keep_text ? (uint8_t *)g_pool.take((size_t)n + 80, &seqs_cap) : nullptr;  // the contigs' symbols as written
    uint8_t *dp = dense;                                                                          // (whitespace removed), back to back
    struct Guard {  // blocks go back to the pool on every early return
        uint32_t *&w; size_t &wc; uint8_t *&d; size_t &dc; bool armed = true;
        ~Guard() { if (armed) { g_pool.give(w, wc); g_pool.give(d, dc); } }
    } guard{words, words_cap, dense, seqs_cap};
    if (!words || (keep_text && !dense)) return KP_ENOMEM;
    std::vector<int32_t> ctg_start, ctg_len, runs, name_off;
    std::string names;
    Packer64 pk(words);
    Body body{pk, runs, dp};
    const int level = simd_level();
    int64_t i = 0;
    while (i < n && data[i] != '>') {  // text before the first header is ignored
        const uint8_t *nl = (const uint8_t *)std::memchr(data + i, '\n', (size_t)(n - i));
        i = nl ? (nl - data) + 1 : n;
    }
    while (i < n) {
        // header line: the record's name is its first word
        int64_t j = i + 1;
        while (j < n && T.code[data[j]] != 8) ++j;
        name_off.push_back((int32_t)names.size());
        names.append((const char *)data + i + 1, (size_t)(j - i - 1));
        const uint8_t *nl = (const uint8_t *)std::memchr(data + j, '\n', (size_t)(n - j));
        const int64_t header_at = i;
        i = nl ? (nl - data) + 1 : n;
        // sequence lines up to the next line that starts with '>'
        pk.pad_to(KP_CONTIG_ALIGN);
        if (pk.pos > (int64_t)KP_MAX_ASM_LEN) return KP_EINVAL;
        if (marks_left > 0) --marks_left;
        const size_t need = (size_t)(pk.pos + (n - i) + 64) / 16 + 8 + (marks_left > 0 ? 2 * (size_t)marks_left : 0);
        if (need > n_words_max) {
            if (marks_left < 0) {
                marks_left = 0;
                for (const uint8_t *p = data + header_at + 1, *e = data + n; p < e && (p = (const uint8_t *)std::memchr(p, '>', (size_t)(e - p))); ++p) ++marks_left;
            }
            const size_t grown = need + 2 * (size_t)marks_left + 64;
            size_t cap2 = 0;
            uint32_t *w2 = (uint32_t *)g_pool.take(grown * 4, &cap2);
            if (!w2) return KP_ENOMEM;
            const size_t done = (size_t)(pk.wp - words);
            std::memcpy(w2, words, done * 4);
            g_pool.give(words, words_cap);
            words = w2; words_cap = cap2; n_words_max = grown;
            pk.wp = words + done;
        }
        const int64_t start = pk.pos;
        body.in_run = false;
        i = level == 2 ? body512(body, data, i, n) : level == 1 ? body256(body, data, i, n) : body_scalar(body, data, i, n);
        if (i < 0 || pk.pos > (int64_t)KP_MAX_ASM_LEN) return KP_EINVAL;
        ctg_start.push_back((int32_t)start);
        ctg_len.push_back((int32_t)(pk.pos - start));
    }
    name_off.push_back((int32_t)names.size());
    pk.pad_to(KP_ASM_ALIGN);
    pk.flush();
    const int64_t pos = pk.pos;
    std::memset(pk.wp, 0, 16);  // (nothing reads past padded_len / 16 words; a zeroed line's worth for good measure)

    Owned *r = new (std::nothrow) Owned();
    if (!r) return KP_ENOMEM;
    auto dup = [](const void *src, size_t bytes) -> void * {
        void *p = std::malloc(bytes ? bytes : 1);
        if (p && bytes) std::memcpy(p, src, bytes);
        return p;
    };
    r->padded_len = pos;
    r->n_contigs = (int32_t)ctg_start.size();
    r->n_runs = (int32_t)(runs.size() / 2);
    r->words = words; r->words_cap = words_cap;
    r->seqs = dense; r->seqs_cap = seqs_cap;
    r->n_seq_bytes = keep_text ? (int64_t)(dp - dense) : 0;
    guard.armed = false;  // the record owns the blocks now
    r->ctg_start = (int32_t *)dup(ctg_start.data(), ctg_start.size() * 4);
    r->ctg_len = (int32_t *)dup(ctg_len.data(), ctg_len.size() * 4);
    r->n_run_pairs = (int32_t *)dup(runs.data(), runs.size() * 4);
    r->names = (char *)dup(names.data(), names.size());
    r->name_off = (int32_t *)dup(name_off.data(), name_off.size() * 4);
    if (!r->ctg_start || !r->ctg_len || !r->n_run_pairs || !r->names || !r->name_off) {
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
    Owned *r = new (std::nothrow) Owned();
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
    Owned *o = static_cast<Owned *>(p);  // every record this library hands out is one
    g_pool.give(o->words, o->words_cap);  // (capacity 0: a plain malloc, freed)
    g_pool.give(o->seqs, o->seqs_cap);
    std::free(p->ctg_start); std::free(p->ctg_len); std::free(p->n_run_pairs);
    std::free(p->names); std::free(p->name_off);
    delete o;
}

}  // extern "C"
