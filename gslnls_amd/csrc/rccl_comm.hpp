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
        return GSLNLS_SUCCESS;
    }
    // bootstrap over a file every rank can see: rank 0 writes the id under a temporary name and renames it
    // (atomic), the others wait for it to appear
    int init_file(const char *path, int rank_, int world_, int timeout_s)
    {
        char id[NCCL_UNIQUE_ID_BYTES];
        if (rank_ == 0)
        {
            int rc = get_unique_id(id);
            if (rc)
                return rc;
            char tmp[4096];
            snprintf(tmp, sizeof tmp, "%s.tmp.%d", path, (int)getpid());
            FILE *f = fopen(tmp, "wb");
            if (!f || fwrite(id, 1, sizeof id, f) != sizeof id)
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
            const time_t t0 = time(nullptr);
            for (;;)
            {
                struct stat st;
                if (stat(path, &st) == 0 && st.st_size == (off_t)sizeof id)
                {
                    FILE *f = fopen(path, "rb");
                    const bool ok = f && fread(id, 1, sizeof id, f) == sizeof id;
                    if (f)
                        fclose(f);
                    if (ok)
                        break;
                }
                if (time(nullptr) - t0 > timeout_s)
                    return GSLNLS_FAILURE;
                usleep(2000);
            }
        }
        return init_rank(id, rank_, world_);
    }
    int ensure(size_t shard_doubles)
    {
        if (shard_doubles <= cap_doubles)
            return 0;
        hipFree(shard);
        hipFree(all);
        shard = all = nullptr;
        cap_doubles = 0;
        if (hipMalloc(&shard, sizeof(double) * shard_doubles) != hipSuccess ||
            hipMalloc(&all, sizeof(double) * shard_doubles * (size_t)world) != hipSuccess)
            return GSLNLS_E_NODEVICE;
        cap_doubles = shard_doubles;
        return 0;
    }
    // ncclAllGather of `count` doubles per rank on `stream`, behind whatever was enqueued there
    int allgather(size_t count, hipStream_t stream)
    {
        if (!comm || count > cap_doubles)
            return GSLNLS_EINVAL;
        const ncclResult_t r = api.AllGather(shard, all, count, ncclFloat64, comm, stream);
        n_allgathers += 1;
        return r == ncclSuccess ? GSLNLS_SUCCESS : GSLNLS_E_NODEVICE;
    }
    void destroy()
    {
        if (comm && api.CommDestroy)
            api.CommDestroy(comm);
        comm = nullptr;
        hipFree(shard);
        hipFree(all);
        shard = all = nullptr;
        cap_doubles = 0;
        rank = 0;
        world = 1;
    }
};

} // namespace gslnls
