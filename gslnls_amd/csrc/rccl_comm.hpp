// rccl_comm.hpp -- the one collective of the path, inside the library.
//
// Multi-start shards the sample points of a batch over the ranks of a one-process-per-GPU job; the only
// exchange step is ONE all-gather of the per-point records per batch (the reference's loop
// over the points is src/nls_mstart.c:42-128 and is strictly sequential, src/nls.c:372-399).  Here that
// all-gather is an RCCL ncclAllGather over xGMI, enqueued on the library's own stream right behind the batch
// kernel: no host language, no torch, no callback in the data path.
//
// RCCL is bound at run time (dlopen), not at link time: a single-GPU host (an R session) loads
// libgslnls_hip.so without RCCL being present.  When the process already carries an RCCL (torch ships one
// as librccl.so) that copy is used, so that a process never holds two.
//
// Bootstrap (once per job, not in the data path): rank 0 creates the 128-byte unique id
// (gslnls_comm_get_unique_id), the host application hands it to the other ranks by whatever channel it has
// (MPI, a socket, torch's store -- or the shared-file helper gslnls_comm_init_file), every rank calls
// gslnls_comm_init_rank.
#pragma once
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include "mstart_driver.hpp"

namespace gslnls
{

struct RcclApi
{
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    char err[256] = {0};

    bool load()
    {
        if (handle)
            return true;
        // an RCCL the process already holds (torch's private copy has no soname: it is known as librccl.so)
        const char *names[] = {"librccl.so", "librccl.so.1"};
        for (const char *nm : names)
            if ((handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD)))
                break;
        if (!handle)
        {
            const char *paths[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
            for (const char *nm : paths)
                if ((handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL)))
                    break;
        }
        if (!handle)
        {
            snprintf(err, sizeof err, "RCCL not found: %s", dlerror());
            return false;
        }
        GetUniqueId = (decltype(GetUniqueId))dlsym(handle, "ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))dlsym(handle, "ncclCommInitRank");
        AllGather = (decltype(AllGather))dlsym(handle, "ncclAllGather");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy)
        {
            snprintf(err, sizeof err, "RCCL lacks a symbol");
            handle = nullptr;
            return false;
        }
        return true;
    }
};

struct RcclComm
{
    RcclApi api;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    double *shard = nullptr, *all = nullptr; // device buffers, grown on demand
    size_t cap_doubles = 0;                  // capacity of `shard`; `all` holds world times that
    long long n_allgathers = 0;              // collectives issued (tests, benchmark)
    // one double per rank, allocated with the communicator: the ranks agree through it on whether a growth of the
    // data buffers succeeded everywhere BEFORE any of them enters the data collective with the new size
    double *flag_shard = nullptr, *flag_all = nullptr;
    double *h_flags = nullptr; // pinned, world doubles
    // optional timing of the collective alone (gslnls_comm_set_timing): an event pair around ncclAllGather
    int timing = 0;
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool timed_pending = false;
    double allgather_ms_total = 0.0;
    long long allgather_timed = 0;
    int fail_next_ensure = 0; // test hook (GSLNLS_COMM_FAIL_ENSURE_RANK): this rank's next growth reports failure

    int get_unique_id(char *out128)
    {
        if (!api.load())
            return GSLNLS_E_UNSUPPORTED;
        ncclUniqueId id;
        if (api.GetUniqueId(&id) != ncclSuccess)
            return GSLNLS_E_NODEVICE;
        memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
        return GSLNLS_SUCCESS;
    }
    int init_rank(const char *id128, int rank_, int world_)
    {
        if (world_ < 1 || rank_ < 0 || rank_ >= world_)
            return GSLNLS_EINVAL;
        destroy();
        if (!api.load())
            return GSLNLS_E_UNSUPPORTED;
        ncclUniqueId id;
        memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
        const ncclResult_t r = api.CommInitRank(&comm, world_, id, rank_);
        if (r != ncclSuccess)
        {
            snprintf(api.err, sizeof api.err, "ncclCommInitRank: %s", api.GetErrorString ? api.GetErrorString(r) : "?");
            comm = nullptr;
            return GSLNLS_E_NODEVICE;
        }
        rank = rank_;
        world = world_;
        if (hipMalloc(&flag_shard, sizeof(double)) != hipSuccess ||
            hipMalloc(&flag_all, sizeof(double) * (size_t)world) != hipSuccess ||
            hipHostMalloc(&h_flags, sizeof(double) * (size_t)world) != hipSuccess)
        {
            snprintf(api.err, sizeof api.err, "communicator status buffers: allocation failed");
            destroy();
            return GSLNLS_E_NODEVICE;
        }
        const char *fe = getenv("GSLNLS_COMM_FAIL_ENSURE_RANK");
        fail_next_ensure = (fe && atoi(fe) == rank) ? 1 : 0;
        return GSLNLS_SUCCESS;
    }
    // bootstrap over a file every rank can see.  The file holds {magic, job nonce, id}: rank 0 removes whatever a
    // previous job left under that name, writes the new one under a temporary name and renames it (atomic); the
    // others wait for a file with this job's nonce (GSLNLS_COMM_NONCE in the environment of every rank, any string
    // the launcher makes up per job; without one: a file not older than timeout_s before this rank started).  After
    // ncclCommInitRank -- which returns only when every rank has joined, i.e. has read the file -- rank 0 removes it,
    // so a finished job leaves nothing behind for the next one to trip over.
    static unsigned long long file_nonce()
    {
        const char *s = getenv("GSLNLS_COMM_NONCE");
        if (!s || !*s)
            return 0ull;
        unsigned long long h = 1469598103934665603ull; // FNV-1a
        for (; *s; ++s)
            h = (h ^ (unsigned char)*s) * 1099511628211ull;
        return h ? h : 1ull;
    }
    int init_file(const char *path, int rank_, int world_, int timeout_s)
    {
        struct Rec
        {
            char magic[8];
            unsigned long long nonce;
            char id[NCCL_UNIQUE_ID_BYTES];
        } rec;
        const unsigned long long nonce = file_nonce();
        if (rank_ == 0)
        {
            (void)unlink(path); // a previous job's id must never be read as this job's
            int rc = get_unique_id(rec.id);
            if (rc)
                return rc;
            memcpy(rec.magic, "GSLNCCL1", 8);
            rec.nonce = nonce;
            char tmp[4096];
            snprintf(tmp, sizeof tmp, "%s.tmp.%d", path, (int)getpid());
            FILE *f = fopen(tmp, "wb");
            if (!f || fwrite(&rec, 1, sizeof rec, f) != sizeof rec)
            {
                if (f)
                    fclose(f);
                return GSLNLS_EINVAL;
            }
            fclose(f);
            if (rename(tmp, path) != 0)
                return GSLNLS_EINVAL;
        }
        else
        {
            const time_t t0w = time(nullptr);
            for (;;)
            {
                struct stat st;
                if (stat(path, &st) == 0 && st.st_size == (off_t)sizeof rec && (nonce != 0 || st.st_mtime + timeout_s >= t0w))
                {
                    FILE *f = fopen(path, "rb");
                    const bool ok = f && fread(&rec, 1, sizeof rec, f) == sizeof rec;
                    if (f)
                        fclose(f);
                    if (ok && memcmp(rec.magic, "GSLNCCL1", 8) == 0 && rec.nonce == nonce)
                        break;
                }
                if (time(nullptr) - t0w > timeout_s)
                {
                    snprintf(api.err, sizeof api.err, "no communicator id of this job under %s after %d s", path, timeout_s);
                    return GSLNLS_FAILURE;
                }
                usleep(2000);
            }
        }
        const int rc = init_rank(rec.id, rank_, world_);
        if (rank_ == 0)
            (void)unlink(path);
        return rc;
    }
    // Grow the data buffers to `shard_doubles` per rank -- on every rank or on none.  All ranks call this with the
    // same size (the shard size is ceil(count / world) x K, identical arguments on every rank), so they all reach the
    // growth branch in the same call; there each allocates, then ONE one-double all-gather carries the outcome, and
    // a failure anywhere is returned everywhere.  No rank can find itself alone in the data collective because a
    // peer ran out of memory.  `stream`: the stream the data collective will be enqueued on.
    int ensure_together(size_t shard_doubles, hipStream_t stream)
    {
        if (shard_doubles <= cap_doubles)
            return 0;
        if (!comm)
            return GSLNLS_EINVAL;
        (void)hipStreamSynchronize(stream); // nothing in flight may still read the old buffers
        hipFree(shard);
        hipFree(all);
        shard = all = nullptr;
        cap_doubles = 0;
        bool ok = hipMalloc(&shard, sizeof(double) * shard_doubles) == hipSuccess &&
                  hipMalloc(&all, sizeof(double) * shard_doubles * (size_t)world) == hipSuccess;
        if (fail_next_ensure)
        {
            fail_next_ensure = 0;
            ok = false;
        }
        if (!ok)
        {
            (void)hipGetLastError();
            hipFree(shard);
            hipFree(all);
            shard = all = nullptr;
        }
        const double mine = ok ? 1.0 : 0.0;
        bool everyone = false;
        if (hipMemcpyAsync(flag_shard, &mine, sizeof(double), hipMemcpyHostToDevice, stream) == hipSuccess &&
            api.AllGather(flag_shard, flag_all, 1, ncclFloat64, comm, stream) == ncclSuccess &&
            hipMemcpyAsync(h_flags, flag_all, sizeof(double) * (size_t)world, hipMemcpyDeviceToHost, stream) == hipSuccess &&
            hipStreamSynchronize(stream) == hipSuccess)
        {
            everyone = true;
            for (int r = 0; r < world; ++r)
                everyone = everyone && h_flags[r] == 1.0;
        }
        if (!everyone)
        {
            hipFree(shard);
            hipFree(all);
            shard = all = nullptr;
            snprintf(api.err, sizeof api.err, "communicator buffers (%zu doubles per rank) could not be allocated on every rank",
                     shard_doubles);
            return GSLNLS_E_NODEVICE;
        }
        cap_doubles = shard_doubles;
        return 0;
    }
    // ncclAllGather of `count` doubles per rank on `stream`, behind whatever was enqueued there
    int allgather(size_t count, hipStream_t stream)
    {
        if (!comm || count > cap_doubles)
            return GSLNLS_EINVAL;
        collect_timing();
        if (timing && t0 && t1)
            (void)hipEventRecord(t0, stream);
        const ncclResult_t r = api.AllGather(shard, all, count, ncclFloat64, comm, stream);
        if (timing && t0 && t1)
        {
            (void)hipEventRecord(t1, stream);
            timed_pending = true;
        }
        n_allgathers += 1;
        if (r != ncclSuccess)
            snprintf(api.err, sizeof api.err, "ncclAllGather: %s", api.GetErrorString ? api.GetErrorString(r) : "?");
        return r == ncclSuccess ? GSLNLS_SUCCESS : GSLNLS_E_NODEVICE;
    }
    // the event pair of the previous timed collective, once it has completed (callers synchronise the stream after
    // every batch, so by the next call it has)
    void collect_timing()
    {
        if (!timed_pending)
            return;
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t0, t1) == hipSuccess)
        {
            allgather_ms_total += ms;
            allgather_timed += 1;
        }
        timed_pending = false;
    }
    void set_timing(int on)
    {
        collect_timing();
        timing = on;
        if (on && !t0)
        {
            (void)hipEventCreate(&t0);
            (void)hipEventCreate(&t1);
        }
        allgather_ms_total = 0.0;
        allgather_timed = 0;
    }
    void destroy()
    {
        if (comm && api.CommDestroy)
            api.CommDestroy(comm);
        comm = nullptr;
        hipFree(shard);
        hipFree(all);
        hipFree(flag_shard);
        hipFree(flag_all);
        if (h_flags)
            (void)hipHostFree(h_flags);
        shard = all = flag_shard = flag_all = h_flags = nullptr;
        cap_doubles = 0;
        timed_pending = false;
        rank = 0;
        world = 1;
    }
};

} // namespace gslnls
