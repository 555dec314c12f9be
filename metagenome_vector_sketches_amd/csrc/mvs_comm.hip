// mvs_comm.hip -- the one exchange step of the multi-GPU path behind the C ABI: an all-gather of per-rank row
// blocks (int8 limb planes, norms) over RCCL, one process per GPU (the reference's counterpart is
// every shard process re-reading the shared vectors.bin, src/pairwise_comp_optimized.cpp:953,962).
//
// RCCL is bound at run time (dlopen of librccl.so.1): the library has no link-time dependency on it, loads on hosts
// without it, and in a process that already carries a copy (PyTorch ships one) the loader hands back that copy.
// A second kind of communicator forwards the same three collectives to caller-supplied functions: ranks that
// share one device (RCCL refuses that), tests, or an application with its own transport.
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "mvs_internal.h"

namespace mvs {
int capi_fail(int code, const char* fmt, ...);          // mvs_capi.hip: sets mvs_last_error() of this thread
hipStream_t capi_stream(mvs_ctx* c);
int capi_device(mvs_ctx* c);
}  // namespace mvs

namespace {

struct NcclId {
    char internal[MVS_COMM_ID_BYTES];
};
static_assert(sizeof(NcclId) == 128, "ncclUniqueId is 128 bytes (rccl.h: NCCL_UNIQUE_ID_BYTES)");

// rccl.h: ncclDataType_t / ncclRedOp_t values used here
constexpr int kNcclInt8 = 0, kNcclInt64 = 4, kNcclMax = 2;

struct Rccl {
    void* handle = nullptr;
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(void**, int, NcclId, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, void*, hipStream_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
};

Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.handle) break;
        }
        if (!r.handle) {
            const char* e = dlerror();
            r.error = std::string("cannot load librccl.so.1: ") + (e ? e : "unknown error");
            return;
        }
        auto sym = [&](const char* n) {
            void* p = dlsym(r.handle, n);
            if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n;
            return p;
        };
        r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
        r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
        r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
        r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
        r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(sym("ncclAllReduce"));
        r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    });
    return r;
}

}  // namespace

struct mvs_comm {
    mvs_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    void* nccl = nullptr;            // ncclComm_t, or NULL for a callback communicator
    mvs_comm_callbacks cb{};         // used when nccl == NULL
    // file transport (mvs_comm_create_files): exchange through <prefix>_<seq>_<rank> files
    std::string prefix;
    unsigned long long seq = 0;
    double timeout_s = 600.0;
};

// ---- file transport: ranks that share one device (or a test box with one GPU) exchange blocks through a directory
// every rank can reach.  A block is written under a temporary name and renamed; readers poll for it; the writer
// removes it once every reader has left an acknowledgement. ----
namespace {

bool exists(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}

bool wait_for(const std::string& p, double timeout_s) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!exists(p)) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    return true;
}

bool write_file(const std::string& p, const void* data, size_t bytes) {
    const std::string tmp = p + ".part";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    const bool ok = bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes;
    if (std::fclose(f) != 0 || !ok) return false;
    return std::rename(tmp.c_str(), p.c_str()) == 0;
}

bool read_file(const std::string& p, void* data, size_t bytes) {
    FILE* f = std::fopen(p.c_str(), "rb");
    if (!f) return false;
    const bool ok = bytes == 0 || std::fread(data, 1, bytes, f) == bytes;
    std::fclose(f);
    return ok;
}

// exchange `bytes` of HOST data per rank: mine in, all ranks' blocks out (world * bytes)
int files_exchange(mvs_comm* m, const void* mine, size_t bytes, char* all) {
    const std::string base = m->prefix + "_" + std::to_string(m->seq++) + "_";
    if (!write_file(base + std::to_string(m->rank), mine, bytes)) return 1;
    for (int r = 0; r < m->world; ++r) {
        char* dst = all + (size_t)r * bytes;
        if (r == m->rank) {
            if (bytes) std::memcpy(dst, mine, bytes);
            continue;
        }
        const std::string f = base + std::to_string(r);
        if (!wait_for(f, m->timeout_s) || !read_file(f, dst, bytes)) return 2;
        if (!write_file(f + ".ack" + std::to_string(m->rank), nullptr, 0)) return 3;
    }
    for (int r = 0; r < m->world; ++r) {   // my block has been read by everybody: remove it and the acknowledgements
        if (r == m->rank) continue;
        const std::string ack = base + std::to_string(m->rank) + ".ack" + std::to_string(r);
        if (!wait_for(ack, m->timeout_s)) return 4;
        ::unlink(ack.c_str());
    }
    ::unlink((base + std::to_string(m->rank)).c_str());
    return 0;
}

int files_allgather(void* user, void* buf, size_t bytes, int rank, int world, void* stream) {
    mvs_comm* m = static_cast<mvs_comm*>(user);
    std::vector<char> mine(bytes ? bytes : 1), all((size_t)world * bytes + 1);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(mine.data(), static_cast<char*>(buf) + (size_t)rank * bytes, bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return 10;
    const int rc = files_exchange(m, mine.data(), bytes, all.data());
    if (rc) return rc;
    if (hipMemcpyAsync(buf, all.data(), (size_t)world * bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return 11;
    return 0;
}

int files_allreduce_max(void* user, int64_t* value, int rank, int world) {
    (void)rank;
    mvs_comm* m = static_cast<mvs_comm*>(user);
    std::vector<int64_t> all((size_t)world);
    const int rc = files_exchange(m, value, sizeof(int64_t), reinterpret_cast<char*>(all.data()));
    if (rc) return rc;
    for (int64_t v : all) *value = v > *value ? v : *value;
    return 0;
}

}  // namespace

namespace {

int nccl_fail(const char* what, int rc) {
    const Rccl& r = rccl();
    return mvs::capi_fail(MVS_E_HIP, "%s: %s", what, r.GetErrorString ? r.GetErrorString(rc) : "RCCL error");
}

int check_comm(mvs_ctx* c, mvs_comm* comm) {
    if (!c || !comm) return mvs::capi_fail(MVS_E_INVALID, "NULL argument");
    if (comm->ctx != c) return mvs::capi_fail(MVS_E_INVALID, "communicator belongs to another context");
    return MVS_OK;
}

// in-place all-gather of `bytes` per rank: rank r's block sits at buf + r * bytes already
int gather_bytes(mvs_ctx* c, mvs_comm* comm, void* buf, size_t bytes) {
    if (comm->world == 1 || bytes == 0) return MVS_OK;
    if (hipSetDevice(mvs::capi_device(c)) != hipSuccess) return mvs::capi_fail(MVS_E_HIP, "hipSetDevice failed");
    if (comm->nccl) {
        const char* mine = static_cast<const char*>(buf) + (size_t)comm->rank * bytes;
        const int rc = rccl().AllGather(mine, buf, bytes, kNcclInt8, comm->nccl, mvs::capi_stream(c));
        return rc == 0 ? MVS_OK : nccl_fail("ncclAllGather", rc);
    }
    const int rc = comm->cb.allgather(comm->cb.user, buf, bytes, comm->rank, comm->world, (void*)mvs::capi_stream(c));
    return rc == 0 ? MVS_OK : mvs::capi_fail(MVS_E_HIP, "communicator callback allgather failed (%d)", rc);
}

}  // namespace

extern "C" {

int mvs_comm_unique_id(void* id) {
    if (!id) return mvs::capi_fail(MVS_E_INVALID, "id is NULL");
    Rccl& r = rccl();
    if (!r.error.empty()) return mvs::capi_fail(MVS_E_HIP, "%s", r.error.c_str());
    NcclId tmp;
    const int rc = r.GetUniqueId(&tmp);
    if (rc != 0) return nccl_fail("ncclGetUniqueId", rc);
    std::memcpy(id, &tmp, sizeof tmp);
    return MVS_OK;
}

int mvs_comm_create(mvs_ctx* c, const void* id, int rank, int world, mvs_comm** out) {
    if (!c || !id || !out) return mvs::capi_fail(MVS_E_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return mvs::capi_fail(MVS_E_INVALID, "rank %d of %d", rank, world);
    Rccl& r = rccl();
    if (!r.error.empty()) return mvs::capi_fail(MVS_E_HIP, "%s", r.error.c_str());
    if (hipSetDevice(mvs::capi_device(c)) != hipSuccess) return mvs::capi_fail(MVS_E_HIP, "hipSetDevice failed");
    mvs_comm* m = new (std::nothrow) mvs_comm();
    if (!m) return mvs::capi_fail(MVS_E_NOMEM, "out of host memory");
    m->ctx = c;
    m->rank = rank;
    m->world = world;
    NcclId nid;
    std::memcpy(&nid, id, sizeof nid);
    const int rc = r.CommInitRank(&m->nccl, world, nid, rank);
    if (rc != 0 || !m->nccl) {
        delete m;
        return nccl_fail("ncclCommInitRank", rc);
    }
    *out = m;
    return MVS_OK;
}

int mvs_comm_create_callbacks(mvs_ctx* c, const mvs_comm_callbacks* cb, int rank, int world, mvs_comm** out) {
    if (!c || !cb || !out) return mvs::capi_fail(MVS_E_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return mvs::capi_fail(MVS_E_INVALID, "rank %d of %d", rank, world);
    if (!cb->allgather || !cb->allreduce_max_i64) return mvs::capi_fail(MVS_E_INVALID, "callback table incomplete");
    mvs_comm* m = new (std::nothrow) mvs_comm();
    if (!m) return mvs::capi_fail(MVS_E_NOMEM, "out of host memory");
    m->ctx = c;
    m->rank = rank;
    m->world = world;
    m->cb = *cb;
    *out = m;
    return MVS_OK;
}

int mvs_comm_create_files(mvs_ctx* c, const char* path_prefix, int rank, int world, mvs_comm** out) {
    if (!c || !path_prefix || !out) return mvs::capi_fail(MVS_E_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return mvs::capi_fail(MVS_E_INVALID, "rank %d of %d", rank, world);
    mvs_comm* m = new (std::nothrow) mvs_comm();
    if (!m) return mvs::capi_fail(MVS_E_NOMEM, "out of host memory");
    m->ctx = c;
    m->rank = rank;
    m->world = world;
    m->prefix = path_prefix;
    m->cb.user = m;
    m->cb.allgather = files_allgather;
    m->cb.allreduce_max_i64 = files_allreduce_max;
    *out = m;
    return MVS_OK;
}

int mvs_comm_destroy(mvs_comm* m) {
    if (!m) return MVS_OK;
    if (m->nccl) {
        (void)hipSetDevice(mvs::capi_device(m->ctx));
        (void)hipStreamSynchronize(mvs::capi_stream(m->ctx));
        (void)rccl().CommDestroy(m->nccl);
    }
    delete m;
    return MVS_OK;
}

int mvs_comm_info(const mvs_comm* m, int* rank, int* world, int* is_rccl) {
    if (!m) return mvs::capi_fail(MVS_E_INVALID, "comm is NULL");
    if (rank) *rank = m->rank;
    if (world) *world = m->world;
    if (is_rccl) *is_rccl = m->nccl ? 1 : 0;
    return MVS_OK;
}

int mvs_allgather_planes(mvs_ctx* c, mvs_comm* comm, int8_t* planes, int64_t rows_per_rank, int limbs, int d_pad) {
    int rc = check_comm(c, comm);
    if (rc) return rc;
    if (!planes || rows_per_rank < 0 || !mvs::limb_code_ok(limbs) || d_pad <= 0 || d_pad % mvs::kBK != 0)
        return mvs::capi_fail(MVS_E_INVALID, "bad argument");
    return gather_bytes(c, comm, planes, (size_t)rows_per_rank * (size_t)mvs::planes_of(limbs) * (size_t)d_pad);
}

int mvs_allgather_f64(mvs_ctx* c, mvs_comm* comm, double* values, int64_t count_per_rank) {
    int rc = check_comm(c, comm);
    if (rc) return rc;
    if (!values || count_per_rank < 0) return mvs::capi_fail(MVS_E_INVALID, "bad argument");
    return gather_bytes(c, comm, values, (size_t)count_per_rank * sizeof(double));
}

int mvs_allgather_bytes(mvs_ctx* c, mvs_comm* comm, void* buf, int64_t bytes_per_rank) {
    int rc = check_comm(c, comm);
    if (rc) return rc;
    if (!buf || bytes_per_rank < 0) return mvs::capi_fail(MVS_E_INVALID, "bad argument");
    return gather_bytes(c, comm, buf, (size_t)bytes_per_rank);
}

int mvs_allreduce_max_i64(mvs_ctx* c, mvs_comm* comm, int64_t* value) {
    int rc = check_comm(c, comm);
    if (rc) return rc;
    if (!value) return mvs::capi_fail(MVS_E_INVALID, "value is NULL");
    if (comm->world == 1) return MVS_OK;
    if (hipSetDevice(mvs::capi_device(c)) != hipSuccess) return mvs::capi_fail(MVS_E_HIP, "hipSetDevice failed");
    if (!comm->nccl) {
        const int r2 = comm->cb.allreduce_max_i64(comm->cb.user, value, comm->rank, comm->world);
        return r2 == 0 ? MVS_OK : mvs::capi_fail(MVS_E_HIP, "communicator callback allreduce failed (%d)", r2);
    }
    hipStream_t st = mvs::capi_stream(c);
    int64_t* d = nullptr;
    if (hipMalloc((void**)&d, 8) != hipSuccess) return mvs::capi_fail(MVS_E_NOMEM, "hipMalloc failed");
    int result = MVS_OK;
    if (hipMemcpyAsync(d, value, 8, hipMemcpyHostToDevice, st) != hipSuccess) {
        result = mvs::capi_fail(MVS_E_HIP, "upload failed");
    } else {
        const int nrc = rccl().AllReduce(d, d, 1, kNcclInt64, kNcclMax, comm->nccl, st);
        if (nrc != 0) result = nccl_fail("ncclAllReduce", nrc);
        else if (hipMemcpyAsync(value, d, 8, hipMemcpyDeviceToHost, st) != hipSuccess ||
                 hipStreamSynchronize(st) != hipSuccess)
            result = mvs::capi_fail(MVS_E_HIP, "download failed");
    }
    (void)hipFree(d);
    return result;
}

}  // extern "C"
