// fresh_pages -- what it costs to deliver 8 / 24 / 32 MB of results into pages the process has never touched (a fresh
// mmap, which is what malloc / Rf_allocVector hand out for a large vector): hipMemcpy D2H straight into them, the same
// after MADV_POPULATE_WRITE or after touching one byte per page with 1..8 threads, and a host memcpy from pinned staging.
// Build: hipcc -O3 --offload-arch=gfx950 -o fresh_pages fresh_pages.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <algorithm>
#include <thread>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif

static double now()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
#define OK(e)                                                                     \
    do                                                                            \
    {                                                                             \
        hipError_t r__ = (e);                                                     \
        if (r__ != hipSuccess)                                                    \
        {                                                                         \
            printf("HIP error %s line %d\n", hipGetErrorString(r__), __LINE__); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

static char *fresh(size_t bytes, bool via_malloc)
{
    if (via_malloc)
        return (char *)malloc(bytes);
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    return p == MAP_FAILED ? nullptr : (char *)p;
}
static void release(char *p, size_t bytes, bool via_malloc)
{
    if (via_malloc)
        free(p);
    else
        munmap(p, bytes);
}
static void touch(char *p, size_t bytes, int nt)
{
    auto work = [=](size_t lo, size_t hi) {
        for (size_t o = lo; o < hi; o += 4096)
            ((volatile char *)p)[o] = 0;
    };
    if (nt <= 1)
    {
        work(0, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
    for (int t = 0; t < nt; ++t)
    {
        const size_t lo = (size_t)t * per, hi = std::min(bytes, lo + per);
        if (lo < hi)
            th.emplace_back(work, lo, hi);
    }
    for (auto &t : th)
        t.join();
}

int main()
{
    const size_t MB = 1 << 20;
    {
        FILE *f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r");
        char buf[128] = "?";
        if (f)
        {
            if (!fgets(buf, sizeof buf, f))
                buf[0] = 0;
            fclose(f);
        }
        printf("transparent_hugepage: %s", buf);
    }
    for (size_t bytes : {8 * MB, 24 * MB, 32 * MB})
    {
        char *d = nullptr, *hp = nullptr;
        OK(hipMalloc(&d, bytes));
        OK(hipHostMalloc(&hp, bytes, hipHostMallocDefault));
        OK(hipMemset(d, 1, bytes));
        memset(hp, 2, bytes);
        printf("== %zu MB ==\n", bytes / MB);
        for (int via_malloc = 0; via_malloc < 2; ++via_malloc)
        {
            const char *how = via_malloc ? "malloc" : "mmap  ";
            auto run = [&](const char *what, auto prep, auto fill) {
                std::vector<double> tp, tf;
                for (int r = 0; r < 5; ++r)
                {
                    char *p = fresh(bytes, via_malloc);
                    const double t0 = now();
                    prep(p);
                    const double t1 = now();
                    fill(p);
                    const double t2 = now();
                    tp.push_back(t1 - t0);
                    tf.push_back(t2 - t1);
                    release(p, bytes, via_malloc);
                }
                std::sort(tp.begin(), tp.end());
                std::sort(tf.begin(), tf.end());
                printf("%s %-34s prepare %.3f ms  fill %.3f ms  (sum %.3f, fill %.1f GB/s)\n", how, what, 1e3 * tp[2], 1e3 * tf[2],
                       1e3 * (tp[2] + tf[2]), bytes / tf[2] / 1e9);
            };
            auto d2h = [&](char *p) { OK(hipMemcpy(p, d, bytes, hipMemcpyDeviceToHost)); };
            auto cpy = [&](char *p) { memcpy(p, hp, bytes); };
            run("hipMemcpy D2H, untouched", [](char *) {}, d2h);
            run("MADV_POPULATE_WRITE + D2H", [&](char *p) {
                const uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + bytes) & ~(uintptr_t)4095;
                if (madvise((void *)a, e - a, MADV_POPULATE_WRITE) != 0)
                    perror("madvise");
            }, d2h);
            for (int nt : {1, 2, 4, 8})
            {
                char name[64];
                snprintf(name, sizeof name, "touch %d thread(s) + D2H", nt);
                run(name, [&](char *p) { touch(p, bytes, nt); }, d2h);
            }
            run("host memcpy from pinned, untouched", [](char *) {}, cpy);
            run("touch 4 threads + host memcpy", [&](char *p) { touch(p, bytes, 4); }, cpy);
        }
        OK(hipFree(d));
        OK(hipHostFree(hp));
    }
    return 0;
}
