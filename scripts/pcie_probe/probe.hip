// pcie_probe -- what a one-shot gslnls_nls() at C2 (16 MB in, 32 MB out, PAGEABLE caller buffers on both sides, as R
// vectors are) can cost at best on this host: link rates with pinned memory, what hipMemcpy does with pageable memory,
// the price of hipHostRegister, host memcpy pageable <-> pinned with 1..8 threads, and kernels that read / write mapped
// host memory directly.  Build: hipcc -O3 --offload-arch=gfx950 -o probe probe.hip -lpthread
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <thread>
#include <vector>

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

__global__ void copy_kernel(const double2 *__restrict__ src, double2 *__restrict__ dst, size_t n2)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

static void par_memcpy(char *dst, const char *src, size_t bytes, int nt)
{
    if (nt <= 1)
    {
        memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = ((bytes / nt) + 4095) & ~(size_t)4095;
    for (int t = 0; t < nt; ++t)
    {
        const size_t lo = (size_t)t * per, hi = lo + per < bytes ? lo + per : bytes;
        if (lo < hi)
            th.emplace_back([=] { memcpy(dst + lo, src + lo, hi - lo); });
    }
    for (auto &t : th)
        t.join();
}

template <class F>
static double med(F f, int reps = 7)
{
    std::vector<double> t;
    for (int r = 0; r < reps; ++r)
    {
        const double t0 = now();
        f();
        t.push_back(now() - t0);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main()
{
    const size_t MB = 1 << 20;
    hipStream_t st;
    OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (size_t bytes : {16 * MB, 32 * MB})
    {
        char *d = nullptr, *hp = nullptr;
        OK(hipMalloc(&d, bytes));
        OK(hipHostMalloc(&hp, bytes, hipHostMallocMapped));
        char *pg = (char *)aligned_alloc(4096, bytes), *pg2 = (char *)aligned_alloc(4096, bytes);
        memset(pg, 1, bytes);
        memset(pg2, 2, bytes);
        memset(hp, 3, bytes);
        OK(hipMemcpy(d, hp, bytes, hipMemcpyHostToDevice));
        printf("== %zu MB ==\n", bytes / MB);
        double t;
        t = med([&] { OK(hipMemcpyAsync(d, hp, bytes, hipMemcpyHostToDevice, st)); OK(hipStreamSynchronize(st)); });
        printf("pinned   H2D  %.3f ms  %.1f GB/s\n", 1e3 * t, bytes / t / 1e9);
        t = med([&] { OK(hipMemcpyAsync(hp, d, bytes, hipMemcpyDeviceToHost, st)); OK(hipStreamSynchronize(st)); });
        printf("pinned   D2H  %.3f ms  %.1f GB/s\n", 1e3 * t, bytes / t / 1e9);
        t = med([&] { OK(hipMemcpy(d, pg, bytes, hipMemcpyHostToDevice)); });
        printf("pageable H2D  %.3f ms  %.1f GB/s (hipMemcpy)\n", 1e3 * t, bytes / t / 1e9);
        t = med([&] { OK(hipMemcpy(pg2, d, bytes, hipMemcpyDeviceToHost)); });
        printf("pageable D2H  %.3f ms  %.1f GB/s (hipMemcpy)\n", 1e3 * t, bytes / t / 1e9);
        // a FRESH pageable destination every time (R allocates the result vectors right before the copy): first touch
        {
            std::vector<double> tt;
            for (int r = 0; r < 5; ++r)
            {
                char *fresh = (char *)aligned_alloc(4096, bytes);
                const double t0 = now();
                OK(hipMemcpy(fresh, d, bytes, hipMemcpyDeviceToHost));
                tt.push_back(now() - t0);
                free(fresh);
            }
            std::sort(tt.begin(), tt.end());
            printf("pageable D2H into untouched malloc  %.3f ms  %.1f GB/s\n", 1e3 * tt[2], bytes / tt[2] / 1e9);
        }
        double tr = med([&] { OK(hipHostRegister(pg, bytes, hipHostRegisterDefault)); OK(hipHostUnregister(pg)); }, 5);
        printf("hipHostRegister + Unregister  %.3f ms\n", 1e3 * tr);
        {
            const double t0 = now();
            OK(hipHostRegister(pg, bytes, hipHostRegisterDefault));
            const double t1 = now();
            OK(hipMemcpyAsync(d, pg, bytes, hipMemcpyHostToDevice, st));
            OK(hipStreamSynchronize(st));
            const double t2 = now();
            OK(hipHostUnregister(pg));
            const double t3 = now();
            printf("register %.3f  copy %.3f  unregister %.3f ms\n", 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2));
        }
        for (int nt : {1, 2, 4, 8, 16})
        {
            t = med([&] { par_memcpy(hp, pg, bytes, nt); });
            const double t2 = med([&] { par_memcpy(pg2, hp, bytes, nt); });
            printf("host memcpy %2d thread(s): pageable->pinned %.3f ms %.1f GB/s | pinned->pageable %.3f ms %.1f GB/s\n", nt,
                   1e3 * t, bytes / t / 1e9, 1e3 * t2, bytes / t2 / 1e9);
        }
        // kernels on mapped host memory
        char *hp_dev = nullptr;
        OK(hipHostGetDevicePointer((void **)&hp_dev, hp, 0));
        hipEvent_t e0, e1;
        OK(hipEventCreate(&e0));
        OK(hipEventCreate(&e1));
        for (int dir = 0; dir < 2; ++dir)
            for (int grid : {64, 256, 1024})
            {
                float best = 1e9f;
                for (int r = 0; r < 5; ++r)
                {
                    OK(hipEventRecord(e0, st));
                    if (dir == 0)
                        hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, st, (const double2 *)hp_dev, (double2 *)d, bytes / 16);
                    else
                        hipLaunchKernelGGL(copy_kernel, dim3(grid), dim3(256), 0, st, (const double2 *)d, (double2 *)hp_dev, bytes / 16);
                    OK(hipEventRecord(e1, st));
                    OK(hipStreamSynchronize(st));
                    float ms;
                    OK(hipEventElapsedTime(&ms, e0, e1));
                    best = ms < best ? ms : best;
                }
                printf("kernel %s mapped host, grid %4d: %.3f ms %.1f GB/s\n", dir == 0 ? "reads " : "writes", grid, best,
                       bytes / (best * 1e-3) / 1e9);
            }
        // chunked pipeline: host memcpy pageable -> pinned ring slot, async H2D per chunk (and the reverse)
        for (size_t chunk : {1 * MB, 2 * MB, 4 * MB})
            for (int nt : {1, 4})
            {
                const int slots = 4;
                hipEvent_t ev[4];
                for (int k = 0; k < slots; ++k)
                    OK(hipEventCreate(&ev[k]));
                t = med([&] {
                    size_t off = 0;
                    int k = 0;
                    while (off < bytes)
                    {
                        const size_t len = bytes - off < chunk ? bytes - off : chunk;
                        const int s = k % slots;
                        if (k >= slots)
                            OK(hipEventSynchronize(ev[s]));
                        par_memcpy(hp + s * chunk, pg + off, len, nt);
                        OK(hipMemcpyAsync(d + off, hp + s * chunk, len, hipMemcpyHostToDevice, st));
                        OK(hipEventRecord(ev[s], st));
                        off += len;
                        ++k;
                    }
                    OK(hipStreamSynchronize(st));
                });
                const double tu = t;
                t = med([&] {
                    size_t off = 0;
                    int k = 0;
                    const int nchunks = (int)((bytes + chunk - 1) / chunk);
                    // issue up to `slots` copies ahead, drain in order
                    int issued = 0;
                    for (; issued < nchunks && issued < slots; ++issued)
                    {
                        const size_t o = (size_t)issued * chunk, len = bytes - o < chunk ? bytes - o : chunk;
                        OK(hipMemcpyAsync(hp + (issued % slots) * chunk, d + o, len, hipMemcpyDeviceToHost, st));
                        OK(hipEventRecord(ev[issued % slots], st));
                    }
                    for (k = 0; k < nchunks; ++k)
                    {
                        const int s = k % slots;
                        off = (size_t)k * chunk;
                        const size_t len = bytes - off < chunk ? bytes - off : chunk;
                        OK(hipEventSynchronize(ev[s]));
                        par_memcpy(pg2 + off, hp + s * chunk, len, nt);
                        if (issued < nchunks)
                        {
                            const size_t o = (size_t)issued * chunk, l2 = bytes - o < chunk ? bytes - o : chunk;
                            OK(hipMemcpyAsync(hp + s * chunk, d + o, l2, hipMemcpyDeviceToHost, st));
                            OK(hipEventRecord(ev[s], st));
                            ++issued;
                        }
                    }
                });
                printf("staged ring chunk %zu MB, %d copy thread(s): H2D %.3f ms %.1f GB/s | D2H %.3f ms %.1f GB/s\n", chunk / MB, nt,
                       1e3 * tu, bytes / tu / 1e9, 1e3 * t, bytes / t / 1e9);
                for (int k = 0; k < slots; ++k)
                    OK(hipEventDestroy(ev[k]));
            }
        OK(hipFree(d));
        OK(hipHostFree(hp));
        free(pg);
        free(pg2);
    }
    return 0;
}
