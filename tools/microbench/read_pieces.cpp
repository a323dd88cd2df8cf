// What does getting a file's bytes cost next to parsing them?  T threads read the same F files (tmpfs / page cache) over and
// over: whole file into a recycled buffer of its size, or in pieces into a buffer that stays in the cache; optionally each
// byte is then looked at once (a sum with AVX loads stands in for the parser's read of the text).
//   g++ -O2 -mavx2 -pthread -o read_pieces.bin read_pieces.cpp && ./read_pieces.bin DIR [threads=16] [seconds=2]
#include <dirent.h>
#include <fcntl.h>
#include <immintrin.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static uint64_t touch(const uint8_t *p, size_t n) {
    __m256i acc = _mm256_setzero_si256();
    size_t i = 0;
    for (; i + 32 <= n; i += 32) acc = _mm256_add_epi64(acc, _mm256_loadu_si256((const __m256i *)(p + i)));
    uint64_t lanes[4];
    _mm256_storeu_si256((__m256i *)lanes, acc);
    return lanes[0] + lanes[1] + lanes[2] + lanes[3];
}

int main(int argc, char **argv) {
    if (argc < 2) return 1;
    const int T = argc > 2 ? atoi(argv[2]) : 16;
    const double secs = argc > 3 ? atof(argv[3]) : 2.0;
    std::vector<std::string> files;
    if (DIR *d = opendir(argv[1])) {
        while (dirent *e = readdir(d))
            if (strstr(e->d_name, ".fasta")) files.push_back(std::string(argv[1]) + "/" + e->d_name);
        closedir(d);
    }
    if (files.empty()) { printf("no files\n"); return 1; }
    for (size_t piece : {(size_t)0, (size_t)64 << 10, (size_t)256 << 10, (size_t)1 << 20}) {
        for (int look = 0; look < 2; ++look) {
            std::atomic<uint64_t> bytes{0}, sink{0};
            std::atomic<bool> stop{false};
            std::vector<std::thread> pool;
            for (int t = 0; t < T; ++t)
                pool.emplace_back([&, t]() {
                    std::vector<uint8_t> buf(piece ? piece : (size_t)8 << 20);
                    uint64_t mine = 0, s = 0;
                    for (size_t k = t; !stop.load(std::memory_order_relaxed); k += T) {
                        const int fd = open(files[k % files.size()].c_str(), O_RDONLY | O_CLOEXEC);
                        if (fd < 0) continue;
                        if (piece == 0) {
                            size_t got = 0;
                            for (;;) {
                                const ssize_t r = read(fd, buf.data() + got, buf.size() - got);
                                if (r <= 0) break;
                                got += (size_t)r;
                            }
                            if (look) s += touch(buf.data(), got);
                            mine += got;
                        } else {
                            for (;;) {
                                const ssize_t r = read(fd, buf.data(), piece);
                                if (r <= 0) break;
                                if (look) s += touch(buf.data(), (size_t)r);
                                mine += (size_t)r;
                            }
                        }
                        close(fd);
                    }
                    bytes += mine;
                    sink += s;
                });
            const double t0 = now();
            std::this_thread::sleep_for(std::chrono::duration<double>(secs));
            stop = true;
            for (auto &th : pool) th.join();
            const double dt = now() - t0;
            printf("%2d threads, %s, %s: %.1f GB/s (%.0f files of 5 MB per second)%s\n", T, piece ? (std::to_string(piece >> 10) + " KB pieces").c_str() : "whole file",
                   look ? "read + one pass over the bytes" : "read only", bytes / dt / 1e9, bytes / dt / 5e6, sink.load() == 42 ? "." : "");
        }
    }
    return 0;
}
