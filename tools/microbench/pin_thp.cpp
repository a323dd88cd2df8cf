// Page-locked host memory two ways: hipHostMalloc against mmap + MADV_HUGEPAGE + hipHostRegister.
// Times allocation (locking), the first H2D copy, release, and what a process that holds N GB of either costs to exit.
//   hipcc -O2 -o pin_thp.bin pin_thp.cpp && ./pin_thp.bin [GB per buffer = 0.8] [buffers = 4]
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static void *huge_alloc(size_t bytes, int advise) {
    const size_t two_mb = 2u << 20;
    bytes = (bytes + two_mb - 1) & ~(two_mb - 1);
    char *raw = (char *)mmap(nullptr, bytes + two_mb, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (raw == MAP_FAILED) { perror("mmap"); exit(1); }
    char *p = (char *)(((uintptr_t)raw + two_mb - 1) & ~(uintptr_t)(two_mb - 1));
    if (p > raw) munmap(raw, p - raw);
    if (raw + two_mb > p) munmap(p + bytes, raw + two_mb - p);
    if (advise && madvise(p, bytes, MADV_HUGEPAGE) != 0) perror("madvise");
    return p;
}

int main(int argc, char **argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 0.8;
    const int n = argc > 2 ? atoi(argv[2]) : 4;
    const size_t bytes = ((size_t)(gb * (1u << 30)) + (2u << 20) - 1) & ~(size_t)((2u << 20) - 1);
    CK(hipSetDevice(0));
    if (argc > 4) {  // child of the exit test: hold n buffers, tell the parent, leave without freeing anything
        const int mode = argv[3][4] - '0', fd = atoi(argv[4]);
        for (int i = 0; i < n; ++i) {
            void *p;
            if (mode == 0) CK(hipHostMalloc(&p, bytes, hipHostMallocPortable));
            else { p = huge_alloc(bytes, 1); memset(p, 1, bytes); CK(hipHostRegister(p, bytes, hipHostRegisterPortable)); }
        }
        if (const char *dg = getenv("PIN_THP_DEVICE_GB")) {  // ... and this much device memory, written once
            const int gbs = atoi(dg);
            for (int i = 0; i < gbs; ++i) { void *d; CK(hipMalloc(&d, 1u << 30)); CK(hipMemset(d, 1, 1u << 30)); }
            CK(hipDeviceSynchronize());
        }
        if (write(fd, "x", 1) != 1) return 1;
        _exit(0);
    }
    void *dev;
    CK(hipMalloc(&dev, bytes));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    for (int mode = 0; mode < 3; ++mode) {  // 0: hipHostMalloc, 1: mmap + register, 2: mmap + MADV_HUGEPAGE + register
        std::vector<void *> bufs;
        double t_alloc = 0, t_touch = 0, t_copy = 0, t_free = 0;
        for (int i = 0; i < n; ++i) {
            double t = now();
            void *p;
            if (mode == 0) CK(hipHostMalloc(&p, bytes, hipHostMallocPortable));
            else {
                p = huge_alloc(bytes, mode == 2);
                double t1 = now();
                memset(p, 1, bytes);  // fault the pages in
                t_touch += now() - t1;
                CK(hipHostRegister(p, bytes, hipHostRegisterPortable));
            }
            t_alloc += now() - t;
            t = now();
            CK(hipMemcpyAsync(dev, p, bytes, hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            t_copy += now() - t;
            bufs.push_back(p);
        }
        double t2 = now();
        CK(hipMemcpyAsync(dev, bufs[0], bytes, hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        const double second_copy = now() - t2;
        for (void *p : bufs) {
            double t = now();
            if (mode == 0) CK(hipHostFree(p));
            else { CK(hipHostUnregister(p)); munmap(p, bytes); }
            t_free += now() - t;
        }
        printf("mode %d (%s): per %.2f GB buffer: lock %.1f ms (of which touching %.1f), first copy %.1f ms, again %.1f ms (%.1f GB/s), release %.1f ms\n", mode,
               mode == 0 ? "hipHostMalloc" : mode == 1 ? "mmap+register" : "mmap+MADV_HUGEPAGE+register", bytes / 1073741824.0, 1e3 * t_alloc / n,
               1e3 * t_touch / n, 1e3 * t_copy / n, 1e3 * second_copy, bytes / 1e9 / second_copy, 1e3 * t_free / n);
    }
    // exit cost of a process that holds n buffers
    for (int mode = 0; mode < 3; mode += 2) {
        fflush(stdout);
        int fd[2];
        if (pipe(fd)) return 1;
        const pid_t pid = fork();  // (the child makes its own runtime state: it only calls HIP after exec-less fork in a fresh state is not allowed, so exec ourselves)
        if (pid == 0) {
            char a1[32], a2[32], a3[32];
            snprintf(a1, sizeof a1, "%f", gb); snprintf(a2, sizeof a2, "%d", n); snprintf(a3, sizeof a3, "%d", fd[1]);
            const char *m = mode == 0 ? "hold0" : "hold2";
            execl("/proc/self/exe", argv[0], a1, a2, m, a3, (char *)nullptr);
            _exit(9);
        }
        close(fd[1]);
        char c;
        if (read(fd[0], &c, 1) != 1) printf("child failed\n");
        const double t = now();
        int st;
        waitpid(pid, &st, 0);
        printf("exit of a process holding %d x %.2f GB (%s): %.0f ms\n", n, gb, mode == 0 ? "hipHostMalloc" : "huge pages + register", 1e3 * (now() - t));
        close(fd[0]);
    }
    return 0;
}
