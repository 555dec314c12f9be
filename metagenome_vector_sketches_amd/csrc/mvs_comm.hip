// mvs_comm.hip -- the one exchange step of the multi-GPU path behind the C ABI: an all-gather of per-rank row
// blocks (int8 limb planes, norms) over RCCL, one process per GPU (the reference's counterpart is
// every shard process re-reading the shared vectors.bin, src/pairwise_comp_optimized.cpp:953,962).
//
// RCCL is bound at run time (dlopen of librccl.so.1): the library has no link-time dependency on it, loads on hosts
// without it, and in a process that already carries a copy (PyTorch ships one) the loader hands back that copy.
// A second kind of communicator forwards the same three collectives to caller-supplied functions: ranks that
// share one device (RCCL refuses that), tests, or an application with its own transport.
#include <dirent.h>
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
const Options& capi_options(mvs_ctx* c);
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
    int (*GetVersion)(int*) = nullptr;
    std::string path;                // file the symbols came from (dladdr), "" if the loader would not say
    int version = 0;                 // ncclGetVersion
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
        // which copy did the loader hand back?  (a process that carries PyTorch has its bundled librccl mapped already)
        r.GetVersion = reinterpret_cast<decltype(r.GetVersion)>(dlsym(r.handle, "ncclGetVersion"));
        Dl_info info{};
        if (r.AllGather && dladdr(reinterpret_cast<void*>(r.AllGather), &info) && info.dli_fname) r.path = info.dli_fname;
        if (r.GetVersion) (void)r.GetVersion(&r.version);
    });
    return r;
}

}  // namespace

struct mvs_comm {
    mvs_ctx* ctx = nullptr;
    int rank = 0, world = 1;
    void* nccl = nullptr;            // ncclComm_t, or NULL for a callback communicator
    long long exchanges_done = 0;    // file transport: exchanges that completed (every peer read this rank's block)
    mvs_comm_callbacks cb{};         // used when nccl == NULL
    // file transport (mvs_comm_create_files): exchange through <prefix>_<seq>_<rank> files
    std::string prefix;
    unsigned long long seq = 0;
    unsigned long long job = 0;      // nonce all ranks of THIS job agreed on at creation; every block carries it
    double timeout_s = 600.0;
    // mvs_allgather_rows: contiguous staging of a row sub-range of every rank's block (grow-only, device)
    void* pack = nullptr;
    size_t pack_bytes = 0;
};

// ---- file transport: ranks that share one device (or a test box with one GPU) exchange blocks through a directory
// every rank can reach.  A block is written under a temporary name and renamed; readers poll for it; the writer
// removes it once every reader has left an acknowledgement.
// Leftovers of an earlier job that used the same prefix (killed, timed out, failed on one rank) must never be taken
// for this job's data: at creation a rank removes its own old files, the ranks then agree on a job nonce (below), and
// every block starts with a header {magic, job nonce, sequence number, payload bytes} the reader verifies -- a file
// that does not carry this job's header is ignored (polled past) like a file that is not there. ----
namespace {

struct BlockHeader {
    unsigned long long magic, job, seq, bytes;
};
constexpr unsigned long long kBlockMagic = 0x4b4c424d4f435653ULL;   // "SVCOMBLK"

bool exists(const std::string& p) {
    struct stat st;
    return ::stat(p.c_str(), &st) == 0;
}

double seconds_since(std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

bool wait_for(const std::string& p, double timeout_s) {
    const auto t0 = std::chrono::steady_clock::now();
    while (!exists(p)) {
        if (seconds_since(t0) > timeout_s) return false;
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    return true;
}

bool write_file(const std::string& p, const BlockHeader* h, const void* data, size_t bytes) {
    const std::string tmp = p + ".part";
    FILE* f = std::fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = !h || std::fwrite(h, sizeof *h, 1, f) == 1;
    ok = ok && (bytes == 0 || std::fwrite(data, 1, bytes, f) == bytes);
    if (std::fclose(f) != 0 || !ok) {
        ::unlink(tmp.c_str());
        return false;
    }
    return std::rename(tmp.c_str(), p.c_str()) == 0;
}

// 1: read, 0: no such file / not this job's block (keep polling), -1: this job's block but damaged
int read_block(const std::string& p, const BlockHeader& want, void* data) {
    FILE* f = std::fopen(p.c_str(), "rb");
    if (!f) return 0;
    BlockHeader h{};
    int rc = 0;
    if (std::fread(&h, sizeof h, 1, f) == 1 && h.magic == want.magic && h.job == want.job && h.seq == want.seq) {
        rc = (h.bytes == want.bytes && (want.bytes == 0 || std::fread(data, 1, want.bytes, f) == want.bytes)) ? 1 : -1;
    }
    std::fclose(f);
    return rc;
}

bool read_u64(const std::string& p, unsigned long long* v) {
    FILE* f = std::fopen(p.c_str(), "rb");
    if (!f) return false;
    const bool ok = std::fread(v, 8, 1, f) == 1;
    std::fclose(f);
    return ok;
}

// this rank's files of any job under the prefix -- exactly the names this transport creates, nothing else that happens
// to share the prefix: <prefix>_<seq>_<rank>, <prefix>_hello_<rank>, <prefix>_ready_<rank>, each optionally followed by
// ".part" (a write in progress) or ".ack<peer>"
void remove_own_files(const std::string& prefix, int rank) {
    const size_t slash = prefix.find_last_of('/');
    const std::string dir = slash == std::string::npos ? "." : prefix.substr(0, slash + 1);
    const std::string stem = (slash == std::string::npos ? prefix : prefix.substr(slash + 1)) + "_";
    const std::string mine = "_" + std::to_string(rank);
    auto all_digits = [](const std::string& t) {
        if (t.empty()) return false;
        for (char ch : t)
            if (ch < '0' || ch > '9') return false;
        return true;
    };
    DIR* d = ::opendir(dir.c_str());
    if (!d) return;
    std::vector<std::string> victims;
    while (dirent* e = ::readdir(d)) {
        const std::string name = e->d_name;
        if (name.size() <= stem.size() || name.compare(0, stem.size(), stem) != 0) continue;
        const size_t dot = name.find('.', stem.size());
        const std::string core = name.substr(stem.size(), dot == std::string::npos ? std::string::npos : dot - stem.size());
        const std::string suffix = dot == std::string::npos ? std::string() : name.substr(dot);
        if (core.size() <= mine.size() || core.compare(core.size() - mine.size(), mine.size(), mine) != 0) continue;
        const std::string kind = core.substr(0, core.size() - mine.size());          // "<seq>", "hello" or "ready"
        if (!(all_digits(kind) || kind == "hello" || kind == "ready")) continue;
        std::string sfx = suffix;
        if (sfx.size() >= 5 && sfx.compare(sfx.size() - 5, 5, ".part") == 0) sfx.erase(sfx.size() - 5);   // a write in progress
        if (!(sfx.empty() || (sfx.compare(0, 4, ".ack") == 0 && all_digits(sfx.substr(4))))) continue;
        victims.push_back((slash == std::string::npos ? std::string() : dir) + name);
    }
    ::closedir(d);
    for (const std::string& v : victims) ::unlink(v.c_str());
}

unsigned long long fresh_nonce(int rank) {
    unsigned long long x = (unsigned long long)std::chrono::system_clock::now().time_since_epoch().count();
    x ^= (unsigned long long)::getpid() << 32;
    x ^= (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() * 0x9e3779b97f4a7c15ULL;
    x += (unsigned long long)(rank + 1) * 0xbf58476d1ce4e5b9ULL;
    x ^= x >> 31;
    return x ? x : 1;
}

// Agreement on the job nonce.  Every rank publishes a fresh random number in <prefix>_hello_<rank>; the job nonce is a
// hash of all of them; every rank then publishes the nonce it computed in <prefix>_ready_<rank> and waits until all
// ranks show the same one.  A rank that picked up the hello file of an EARLIER job (its owner had not replaced it yet)
// computes a nonce nobody else has, sees the disagreement, reads the hello files again and publishes anew -- so the
// loop ends exactly when all ranks of this job have seen each other's fresh numbers.
int files_handshake(mvs_comm* m) {
    remove_own_files(m->prefix, m->rank);
    const unsigned long long mine = fresh_nonce(m->rank);
    if (!write_file(m->prefix + "_hello_" + std::to_string(m->rank), nullptr, &mine, 8)) return 1;
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long published = 0;
    for (;;) {
        unsigned long long job = 0x243f6a8885a308d3ULL;
        bool all = true;
        for (int r = 0; r < m->world && all; ++r) {
            unsigned long long v = mine;
            if (r != m->rank) all = read_u64(m->prefix + "_hello_" + std::to_string(r), &v);
            job = (job ^ v) * 0x100000001b3ULL;
            job ^= job >> 29;
        }
        if (all) {
            if (job == 0) job = 1;
            if (job != published) {
                if (!write_file(m->prefix + "_ready_" + std::to_string(m->rank), nullptr, &job, 8)) return 2;
                published = job;
            }
            bool agreed = true;
            for (int r = 0; r < m->world && agreed; ++r) {
                unsigned long long v = 0;
                agreed = r == m->rank || (read_u64(m->prefix + "_ready_" + std::to_string(r), &v) && v == job);
            }
            if (agreed) {
                m->job = job;
                return 0;
            }
        }
        if (seconds_since(t0) > m->timeout_s) return 3;
        std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
}

// exchange `bytes` of HOST data per rank: mine in, all ranks' blocks out (world * bytes)
int files_exchange(mvs_comm* m, const void* mine, size_t bytes, char* all, double timeout_s = -1.0) {
    if (timeout_s < 0) timeout_s = m->timeout_s;
    const BlockHeader h{kBlockMagic, m->job, m->seq, (unsigned long long)bytes};
    const std::string base = m->prefix + "_" + std::to_string(m->seq++) + "_";
    const std::string own = base + std::to_string(m->rank);
    if (!write_file(own, &h, mine, bytes)) return 1;
    int rc = 0;
    for (int r = 0; r < m->world && !rc; ++r) {
        char* dst = all + (size_t)r * bytes;
        if (r == m->rank) {
            if (bytes) std::memcpy(dst, mine, bytes);
            continue;
        }
        const std::string f = base + std::to_string(r);
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            const int got = read_block(f, h, dst);
            if (got == 1) break;
            if (got < 0 || seconds_since(t0) > timeout_s) {
                rc = 2;
                break;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (!rc && !write_file(f + ".ack" + std::to_string(m->rank), nullptr, nullptr, 0)) rc = 3;
    }
    for (int r = 0; r < m->world; ++r) {   // my block has been read by everybody: remove it and the acknowledgements
        if (r == m->rank) continue;
        const std::string ack = own + ".ack" + std::to_string(r);
        if (!rc && !wait_for(ack, timeout_s)) rc = 4;
        ::unlink(ack.c_str());
    }
    ::unlink(own.c_str());                 // on the error paths too: nothing of this rank stays behind
    if (!rc) ++m->exchanges_done;
    return rc;
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
    m->timeout_s = (double)mvs::capi_options(c).comm_timeout_s;
    const int rc = files_handshake(m);      // collective: all ranks of the job meet here
    if (rc) {
        remove_own_files(m->prefix, rank);
        delete m;
        return mvs::capi_fail(MVS_E_HIP, "file transport: the %d ranks did not meet under %s (%d)", world, path_prefix, rc);
    }
    *out = m;
    return MVS_OK;
}

int mvs_comm_create_rendezvous(mvs_ctx* c, const char* path_prefix, int rank, int world, mvs_comm** out) {
    if (!c || !path_prefix || !out) return mvs::capi_fail(MVS_E_INVALID, "NULL argument");
    *out = nullptr;
    // the ranks meet through the file transport's handshake (stale files of an earlier job cannot be mistaken for this
    // one's), rank 0's RCCL id travels as one verified block, the meeting point is cleaned up, then ncclCommInitRank
    mvs_comm* meet = nullptr;
    int rc = mvs_comm_create_files(c, path_prefix, rank, world, &meet);
    if (rc) return rc;
    std::vector<char> ids((size_t)world * MVS_COMM_ID_BYTES);
    char id[MVS_COMM_ID_BYTES] = {0};
    long long status = 0;
    if (rank == 0 && mvs_comm_unique_id(id) != MVS_OK) status = 1;      // the others must not wait for an id that never comes
    const int xrc = files_exchange(meet, id, sizeof id, ids.data());
    mvs_comm_destroy(meet);
    if (status) return MVS_E_HIP;                                        // mvs_last_error() is mvs_comm_unique_id's
    if (xrc) return mvs::capi_fail(MVS_E_HIP, "file transport: exchange of the RCCL id failed (%d)", xrc);
    bool any = false;
    for (int i = 0; i < MVS_COMM_ID_BYTES; ++i) any = any || ids[i] != 0;
    if (!any) return mvs::capi_fail(MVS_E_HIP, "rank 0 could not draw an RCCL id");
    return mvs_comm_create(c, ids.data(), rank, world, out);
}

int mvs_comm_destroy(mvs_comm* m) {
    if (!m) return MVS_OK;
    if (m->nccl) {
        (void)hipSetDevice(mvs::capi_device(m->ctx));
        (void)hipStreamSynchronize(mvs::capi_stream(m->ctx));
        (void)rccl().CommDestroy(m->nccl);
    }
    if (m->pack) {
        (void)hipSetDevice(mvs::capi_device(m->ctx));
        (void)hipStreamSynchronize(mvs::capi_stream(m->ctx));
        (void)hipFree(m->pack);
    }
    if (!m->prefix.empty()) {
        // The hello / ready files may go once every peer is past the creation handshake, and every block of this rank
        // has been read when its exchange returned (files_exchange waits for the acknowledgements).  A completed exchange
        // proves the former -- nobody can send a block before the handshake is through -- so only a communicator that
        // never exchanged anything needs a last (empty) exchange here; shard processes that finish minutes apart
        // (pairwise_comp_optimized with MVS_COLLECTIVE=files) no longer wait for each other at exit.
        if (m->world > 1 && m->exchanges_done == 0) {
            char none = 0;
            std::vector<char> all((size_t)m->world);
            (void)files_exchange(m, &none, 0, all.data(), m->timeout_s < 10.0 ? m->timeout_s : 10.0);
        }
        remove_own_files(m->prefix, m->rank);
    }
    delete m;
    return MVS_OK;
}

int mvs_comm_library(char* path, size_t path_len, int* version) {
    Rccl& r = rccl();
    if (!r.error.empty()) return mvs::capi_fail(MVS_E_HIP, "%s", r.error.c_str());
    if (path && path_len) {
        const size_t n = std::min(path_len - 1, r.path.size());
        memcpy(path, r.path.data(), n);
        path[n] = 0;
    }
    if (version) *version = r.version;
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

int mvs_allgather_rows(mvs_ctx* c, mvs_comm* comm, int8_t* planes, int64_t rows_per_rank, int64_t row_first,
                       int64_t row_count, int limbs, int d_pad) {
    int rc = check_comm(c, comm);
    if (rc) return rc;
    if (!planes || rows_per_rank < 0 || row_first < 0 || row_count < 0 || row_first + row_count > rows_per_rank ||
        !mvs::limb_code_ok(limbs) || d_pad <= 0 || d_pad % mvs::kBK != 0)
        return mvs::capi_fail(MVS_E_INVALID, "bad argument");
    const size_t row_bytes = (size_t)mvs::planes_of(limbs) * (size_t)d_pad;
    const size_t block = (size_t)rows_per_rank * row_bytes, part = (size_t)row_count * row_bytes;
    if (row_count == rows_per_rank) return gather_bytes(c, comm, planes, block);
    if (comm->world == 1 || part == 0) return MVS_OK;
    if (hipSetDevice(mvs::capi_device(c)) != hipSuccess) return mvs::capi_fail(MVS_E_HIP, "hipSetDevice failed");
    hipStream_t st = mvs::capi_stream(c);
    const size_t need = part * (size_t)comm->world;
    if (comm->pack_bytes < need) {
        if (comm->pack) {
            if (hipStreamSynchronize(st) != hipSuccess) return mvs::capi_fail(MVS_E_HIP, "sync failed");
            (void)hipFree(comm->pack);
            comm->pack = nullptr;
            comm->pack_bytes = 0;
        }
        if (hipMalloc(&comm->pack, need) != hipSuccess) return mvs::capi_fail(MVS_E_NOMEM, "hipMalloc of %zu bytes failed", need);
        comm->pack_bytes = need;
    }
    // the sub-range of every rank's block is `part` contiguous bytes at stride `block`: this rank's goes into its slot
    // of the contiguous staging buffer, the ordinary in-place all-gather runs there, and one strided copy puts the
    // other ranks' rows where they belong (2 x part x world bytes of device copies: microseconds)
    char* own_src = reinterpret_cast<char*>(planes) + (size_t)comm->rank * block + (size_t)row_first * row_bytes;
    if (hipMemcpyAsync(static_cast<char*>(comm->pack) + (size_t)comm->rank * part, own_src, part, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return mvs::capi_fail(MVS_E_HIP, "staging copy failed");
    rc = gather_bytes(c, comm, comm->pack, part);
    if (rc) return rc;
    if (hipMemcpy2DAsync(reinterpret_cast<char*>(planes) + (size_t)row_first * row_bytes, block, comm->pack, part, part,
                         (size_t)comm->world, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return mvs::capi_fail(MVS_E_HIP, "strided copy failed");
    return MVS_OK;
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
