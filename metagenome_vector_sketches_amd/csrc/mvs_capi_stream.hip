// mvs_capi_stream.hip -- C ABI: the comparison with its result streamed out in row blocks (mvs_pairwise_stream[_encoded])
#include "mvs_capi_internal.h"

#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

using namespace mvs_capi;

extern "C" {

// -------------------------------------------------------------------------------------------------
// streamed output
// -------------------------------------------------------------------------------------------------
}  // extern "C"

namespace mvs_capi {

int bits_for(int64_t max_value) {          // bits that hold 0 .. max_value
    int b = 1;
    while (b < 63 && (max_value >> b) != 0) ++b;
    return b;
}

// Device -> host copies on a DMA engine instead of the runtime's copy kernel (option stream_copy = 1).  hipMemcpyAsync into pinned
// memory is a blit KERNEL on this stack (__amd_rocclr_copyBuffer in the traces): it reaches the link's rate but holds compute units
// while the link drains it.  hsa_amd_memory_async_copy moves the same bytes at the same 56 GB/s with no CU involved
// (tools/microbench/sdma_copy.hip).  ROCr is the layer HIP itself sits on: same process, same address space, pointers of hipMalloc /
// hipHostMalloc are valid as they are.  The copies are not on a HIP stream any more, so their ordering is the host's: the feeder
// waits for the block's arrays (dl_ready) before the first copy, the callback thread waits for a piece's completion signal, and
// the driving thread re-uses a set of arrays once the callback thread has seen the last piece of the block that used it.
struct HsaCopy {
    hsa_agent_t gpu{}, cpu{};
    hsa_signal_t sig[2]{};
    bool ok = false;
};
struct HsaAgents {
    std::vector<hsa_agent_t> gpus;
    hsa_agent_t cpu{};
    bool have_cpu = false;
};
hsa_status_t hsa_on_agent(hsa_agent_t a, void* data) {
    HsaAgents* ag = static_cast<HsaAgents*>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_GPU) ag->gpus.push_back(a);
    if (t == HSA_DEVICE_TYPE_CPU && !ag->have_cpu) {
        ag->cpu = a;
        ag->have_cpu = true;
    }
    return HSA_STATUS_SUCCESS;
}
void hsa_copy_free(void* p) {
    HsaCopy* h = static_cast<HsaCopy*>(p);
    if (h->ok) {
        for (hsa_signal_t& s : h->sig) (void)hsa_signal_destroy(s);
        (void)hsa_shut_down();
    }
    delete h;
}
int ensure_hsa_copy(mvs_ctx* c) {
    if (c->dl_hsa) return static_cast<HsaCopy*>(c->dl_hsa)->ok ? MVS_OK : fail(MVS_E_HIP, "the DMA copy path is not available");
    HsaCopy* h = new (std::nothrow) HsaCopy();
    if (!h) return fail(MVS_E_NOMEM, "out of host memory");
    c->dl_hsa = h;
    c->dl_hsa_free = hsa_copy_free;
    if (hsa_init() != HSA_STATUS_SUCCESS) return fail(MVS_E_HIP, "hsa_init failed");
    HsaAgents ag;
    if (hsa_iterate_agents(hsa_on_agent, &ag) != HSA_STATUS_SUCCESS || ag.gpus.empty() || !ag.have_cpu) {
        (void)hsa_shut_down();
        return fail(MVS_E_HIP, "no HSA agents");
    }
    // the context's device among the GPU agents: by PCI bus / device / function, by index if that cannot be read
    size_t pick = (size_t)c->device < ag.gpus.size() ? (size_t)c->device : 0;
    char bus[32] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, c->device) == hipSuccess) {
        unsigned dom = 0, b = 0, d = 0, f = 0;
        if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) == 4)
            for (size_t k = 0; k < ag.gpus.size(); ++k) {
                uint32_t bdf = 0;
                if (hsa_agent_get_info(ag.gpus[k], (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS &&
                    ((bdf >> 8) & 0xff) == b && ((bdf >> 3) & 0x1f) == d && (bdf & 0x7) == f)
                    pick = k;
            }
    }
    h->gpu = ag.gpus[pick];
    h->cpu = ag.cpu;
    for (hsa_signal_t& sg : h->sig)
        if (hsa_signal_create(0, 0, nullptr, &sg) != HSA_STATUS_SUCCESS) {
            (void)hsa_shut_down();
            return fail(MVS_E_HIP, "hsa_signal_create failed");
        }
    h->ok = true;
    return MVS_OK;
}

// Hand-over between the thread that drives the GPU and the one that runs the caller's callback: two pinned buffers,
// a queue of filled ones.  The callback therefore runs beside the next block's kernels and downloads.
struct StreamOut {
    struct Item {
        int slot;
        int64_t row_begin, row_end, n_cells;
        std::vector<int64_t> row_ptr;      // rebased to the block's first cell
        bool wide;
        // encoded pieces: the directory of the piece's non-empty rows, the records' byte count
        std::vector<uint32_t> rows, first_col, jac_bytes;
        std::vector<uint64_t> offset;
        int64_t n_bytes = 0;
        bool dma = false;                  // the piece was copied by a DMA engine: its completion is the slot's HSA signal
        bool last_of_block = false;        // (DMA copies) the callback thread has seen a block's last piece: its arrays are free again
    };
    mvs_ctx* c;
    mvs_row_block_cb cb = nullptr;
    mvs_encoded_rows_cb ecb = nullptr;     // set instead of cb by mvs_pairwise_stream_encoded
    void* user;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Item> queue;
    bool slot_busy[2] = {false, false};
    bool closing = false;
    int cb_status = 0;                     // first non-zero return of the callback
    std::string error;
    std::thread worker;
    // The feeder: hands finished row blocks to the link piece by piece (it blocks on the two pinned buffers), so that the
    // thread that drives the device never waits for the link -- it runs at most two blocks ahead (the device-side arrays
    // of a block are double-buffered: set k & 1).
    std::thread feeder;
    std::deque<std::function<int()>> feed_queue;
    bool feed_closing = false;
    int64_t fed_blocks = 0;                // blocks whose pieces have all been queued on the download stream
    int64_t done_blocks = 0;               // (DMA copies) blocks whose last piece the callback thread has consumed
    int feed_rc = 0;

    void feed_run() {
        (void)hipSetDevice(c->device);
        for (;;) {
            std::function<int()> task;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return feed_closing || !feed_queue.empty(); });
                if (feed_queue.empty()) return;
                task = std::move(feed_queue.front());
                feed_queue.pop_front();
            }
            const int r = task();
            {
                std::lock_guard<std::mutex> lk(mu);
                if (r != 0 && feed_rc == 0) {
                    feed_rc = r;
                    if (error.empty()) error = std::string("feeding the link failed: ") + mvs_last_error();
                }
                ++fed_blocks;
            }
            cv.notify_all();
        }
    }
    void enqueue_feed(std::function<int()> task) {
        {
            std::lock_guard<std::mutex> lk(mu);
            feed_queue.push_back(std::move(task));
        }
        cv.notify_all();
    }
    void wait_fed(int64_t blocks) {         // until that many blocks have been handed to the download stream
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return fed_blocks >= blocks; });
    }
    void wait_done(int64_t blocks) {        // (DMA copies) until that many blocks have left the device entirely
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return done_blocks >= blocks || cb_status != 0 || !error.empty() || feed_rc != 0; });
    }
    void close_feeder() {
        {
            std::lock_guard<std::mutex> lk(mu);
            feed_closing = true;
        }
        cv.notify_all();
        if (feeder.joinable()) feeder.join();
    }

    void run() {
        (void)hipSetDevice(c->device);
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return closing || !queue.empty(); });
                if (queue.empty()) return;
                it = std::move(queue.front());
                queue.pop_front();
            }
            int status = 0;
            hipError_t e = hipSuccess;
            if (it.dma) {
                const HsaCopy* h = static_cast<const HsaCopy*>(c->dl_hsa);
                if (hsa_signal_wait_scacquire(h->sig[it.slot], HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED) != 0)
                    e = hipErrorUnknown;       // (a negative value: the copy failed)
            } else {
                e = hipEventSynchronize(c->dl_done[it.slot]);
            }
            if (e != hipSuccess) {
                std::lock_guard<std::mutex> lk(mu);
                if (error.empty()) error = std::string("download failed: ") + hipGetErrorString(e);
            } else {
                bool skip;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    skip = cb_status != 0 || !error.empty();
                }
                if (!skip && ecb) {
                    mvs_encoded_rows b{};
                    b.row_begin = it.row_begin;
                    b.row_end = it.row_end;
                    b.n_cells = it.n_cells;
                    b.n_rows = (int64_t)it.rows.size();
                    b.rows = it.rows.data();
                    b.first_col = it.first_col.data();
                    b.offset = it.offset.data();
                    b.jac_bytes = it.jac_bytes.data();
                    b.bytes = static_cast<const uint8_t*>(c->dl_pinned[it.slot]);
                    b.n_bytes = it.n_bytes;
                    try {
                        status = ecb(user, &b);
                    } catch (...) {
                        status = -1;
                    }
                } else if (!skip) {
                    mvs_row_block b{};
                    b.row_begin = it.row_begin;
                    b.row_end = it.row_end;
                    b.n_cells = it.n_cells;
                    b.row_ptr = it.row_ptr.data();
                    const char* base = static_cast<const char*>(c->dl_pinned[it.slot]);
                    b.col = reinterpret_cast<const int32_t*>(base);
                    const char* qbase = base + (size_t)it.n_cells * 4;
                    b.q = it.wide ? nullptr : reinterpret_cast<const uint8_t*>(qbase);
                    b.q16 = it.wide ? reinterpret_cast<const uint16_t*>(qbase) : nullptr;
                    try {
                        status = cb(user, &b);
                    } catch (...) {            // a C++ callback that throws: no exception crosses the C boundary
                        status = -1;
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (status != 0 && cb_status == 0) cb_status = status;
                slot_busy[it.slot] = false;
                if (it.last_of_block) ++done_blocks;
            }
            cv.notify_all();
        }
    }
    int acquire_slot() {                    // blocks until one of the two pinned buffers is free
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !slot_busy[0] || !slot_busy[1]; });
        const int sl = slot_busy[0] ? 1 : 0;
        slot_busy[sl] = true;
        return sl;
    }
    void release_slot(int sl) {
        {
            std::lock_guard<std::mutex> lk(mu);
            slot_busy[sl] = false;
        }
        cv.notify_all();
    }
    void push(Item&& it) {
        {
            std::lock_guard<std::mutex> lk(mu);
            queue.push_back(std::move(it));
        }
        cv.notify_all();
    }
    bool failed() {
        std::lock_guard<std::mutex> lk(mu);
        return cb_status != 0 || !error.empty();
    }
    void close() {
        {
            std::lock_guard<std::mutex> lk(mu);
            closing = true;
        }
        cv.notify_all();
        if (worker.joinable()) worker.join();
    }
    ~StreamOut() {
        close_feeder();
        close();
    }
};

int ensure_download_side(mvs_ctx* c) {
    if (!c->dl_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&c->dl_done[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->dl_block[i], hipEventDisableTiming));
        }
        for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&c->dl_ready[i], hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&c->post_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->cmp_done, hipEventDisableTiming));
    }
    return MVS_OK;
}

// pinned buffer `slot` holds at least `bytes`; called by the producer while it owns the slot (nobody reads it)
int ensure_pinned_slot(mvs_ctx* c, int slot, size_t bytes) {
    if (c->dl_bytes[slot] >= bytes) return MVS_OK;
    if (c->dl_pinned[slot]) HIP_TRY(hipHostFree(c->dl_pinned[slot]));
    c->dl_pinned[slot] = nullptr;
    c->dl_bytes[slot] = 0;
    HIP_TRY(hipHostMalloc(&c->dl_pinned[slot], bytes, hipHostMallocDefault));
    c->dl_bytes[slot] = bytes;
    return MVS_OK;
}

// A row block on its way out: its CSR arrays sit in set `set` of the context (device), row_ptr is on the host.
struct BlockCsr {
    int64_t rb = 0, re = 0, n = 0;
    std::vector<int64_t> row_ptr;      // re - rb + 1 entries
    bool wide = false;                 // q is 16 bits wide in this block
    int set = 0;
    // rows encoded on the device: byte offset of every row's record (rows + 1 entries), directory values per row
    bool sizes_ready = false;          // the encoder's per-row sizes (en_size / en_jac / en_first / en_par) are on the device already
    bool encoded = false;
    std::vector<uint64_t> enc_off;
    std::vector<uint32_t> enc_jac, enc_first;
};

// The CSR arrays of `b` (set b.set) -> the rows' shard records in c->st_enc[b.set], directory on the host; on stream `ps`
// (the context's stream, or the side stream on which a dense block is post-processed beside the next comparison)
int encode_block(mvs_ctx* c, BlockCsr& b, hipStream_t ps) {
    const int64_t rows = b.re - b.rb;
    b.encoded = true;
    b.enc_off.assign((size_t)rows + 1, 0);
    b.enc_jac.assign((size_t)rows, 0);
    b.enc_first.assign((size_t)rows, 0);
    if (b.n == 0 || rows == 0) {
        HIP_TRY(hipEventRecord(c->dl_ready[b.set], ps));
        return MVS_OK;
    }
    int rc = ensure_buf(c, &c->en_size, &c->en_size_bytes, (size_t)(rows + 1) * 8);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_off, &c->en_off_bytes, (size_t)(rows + 1) * 8);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_jac, &c->en_jac_bytes, (size_t)rows * 4);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_first, &c->en_first_bytes, (size_t)rows * 4);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_par, &c->en_par_bytes, (size_t)rows * sizeof(mvs::EncRow));
    if (rc) return rc;
    const int qb = b.wide ? 2 : 1;
    HIP_TRY(hipMemsetAsync((char*)c->en_size + (size_t)rows * 8, 0, 8, ps));
    if (!b.sizes_ready) {                  // (a dense block's fill pass has computed them already)
        mvs::launch_encode_sizes(ps, (const long long*)c->st_rowptr, (const int32_t*)c->st_col[b.set], c->st_q[b.set], qb, rows,
                                 (unsigned long long*)c->en_size, (unsigned int*)c->en_jac, (unsigned int*)c->en_first,
                                 (mvs::EncRow*)c->en_par);
        rc = check_kernel("k_enc_size");
        if (rc) return rc;
    }
    size_t need = 0;
    rc = mvs::encode_offsets(ps, (unsigned long long*)c->en_size, (unsigned long long*)c->en_off, rows, nullptr, 0, &need);
    if (rc) return fail(rc, "scan sizing failed");
    rc = ensure_buf(c, &c->pw_sort, &c->pw_sort_bytes, need);
    if (rc) return rc;
    rc = mvs::encode_offsets(ps, (unsigned long long*)c->en_size, (unsigned long long*)c->en_off, rows, c->pw_sort,
                             c->pw_sort_bytes, nullptr);
    if (rc) return fail(rc, "scan of the record sizes failed");
    rc = read_back(c, ps, {{b.enc_off.data(), c->en_off, (size_t)(rows + 1) * 8},
                           {b.enc_jac.data(), c->en_jac, (size_t)rows * 4},
                           {b.enc_first.data(), c->en_first, (size_t)rows * 4}});
    if (rc) return rc;
    const size_t total = (size_t)b.enc_off[(size_t)rows];
    rc = ensure_buf(c, &c->st_enc[b.set], &c->st_enc_bytes[b.set], std::max<size_t>(total, 8));
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->st_enc[b.set], 0, total, ps));       // the unary parts are OR-ed into zeroed words
    mvs::launch_encode_fill(ps, (const long long*)c->st_rowptr, (const int32_t*)c->st_col[b.set], c->st_q[b.set], qb, rows,
                            (const unsigned long long*)c->en_off, (const mvs::EncRow*)c->en_par, (unsigned char*)c->st_enc[b.set],
                            c->opt.encode_stage_words);
    rc = check_kernel("k_enc_fill");
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->dl_ready[b.set], ps));
    return MVS_OK;
}

// before the CSR arrays of set `set` are rewritten: the downloads of the block that used them last (two blocks ago) are through
int claim_csr_set(mvs_ctx* c, int set, int64_t block_index, int64_t n, bool wide, hipStream_t ps) {
    // (with DMA copies the driving thread has waited for the callback thread to see that block's last piece: StreamOut::wait_done)
    if (block_index >= 2 && c->opt.stream_copy == 0) HIP_TRY(hipStreamWaitEvent(ps, c->dl_block[set], 0));
    int rc = ensure_buf(c, &c->st_col[set], &c->st_col_bytes[set], (size_t)std::max<int64_t>(n, 1) * 4);
    if (rc) return rc;
    return ensure_buf(c, &c->st_q[set], &c->st_q_bytes[set], (size_t)std::max<int64_t>(n, 1) * (wide ? 2 : 1));
}

// n packed cells of rows [rb, re) sit in c->st_raw: radix sort on the (row, col) bits, then row_ptr / col / q
int csr_from_packed(mvs_ctx* c, int64_t rb, int64_t re, int64_t n, int shift, int col_bits, int64_t block_index, BlockCsr& out) {
    const int64_t rows = re - rb;
    const int row_bits = bits_for(std::max<int64_t>(rows - 1, 1));
    out.rb = rb;
    out.re = re;
    out.n = n;
    out.wide = false;
    out.set = (int)(block_index & 1);
    out.row_ptr.assign((size_t)rows + 1, 0);
    if (n == 0) {
        HIP_TRY(hipEventRecord(c->dl_ready[out.set], c->stream));
        return MVS_OK;
    }
    int rc = ensure_buf(c, &c->st_rowptr, &c->st_rowptr_bytes, (size_t)(rows + 1) * 8);
    if (rc) return rc;
    rc = ensure_buf(c, &c->st_sorted, &c->st_sorted_bytes, (size_t)n * 8);
    if (rc) return rc;
    size_t need = 0;
    rc = mvs::sort_packed(c->stream, (unsigned long long*)c->st_raw, (unsigned long long*)c->st_sorted, n, 16, shift + row_bits,
                          nullptr, 0, &need);
    if (rc) return fail(rc, "sort sizing failed");
    rc = ensure_buf(c, &c->pw_sort, &c->pw_sort_bytes, need);
    if (rc) return rc;
    rc = mvs::sort_packed(c->stream, (unsigned long long*)c->st_raw, (unsigned long long*)c->st_sorted, n, 16, shift + row_bits,
                          c->pw_sort, c->pw_sort_bytes, nullptr);
    if (rc) return fail(rc, "sort of the kept cells failed");
    rc = claim_csr_set(c, out.set, block_index, n, false, c->stream);
    if (rc) return rc;
    unsigned int* d_wide = reinterpret_cast<unsigned int*>(c->d_counter + 3);
    HIP_TRY(hipMemsetAsync(d_wide, 0, 4, c->stream));
    const unsigned long long col_mask = (1ULL << col_bits) - 1ULL;
    mvs::launch_packed_csr(c->stream, (const unsigned long long*)c->st_sorted, n, shift, rows, col_mask, (long long*)c->st_rowptr,
                           (int32_t*)c->st_col[out.set], (uint8_t*)c->st_q[out.set], nullptr, d_wide);
    rc = check_kernel("k_packed_csr");
    if (rc) return rc;
    unsigned int h_wide = 0;
    HIP_TRY(hipMemcpyAsync(out.row_ptr.data(), c->st_rowptr, (size_t)(rows + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&h_wide, d_wide, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h_wide) {                      // some q needs 16 bits (norms that do not belong to the vectors): redo the q array
        out.wide = true;
        rc = ensure_buf(c, &c->st_q[out.set], &c->st_q_bytes[out.set], (size_t)n * 2);
        if (rc) return rc;
        mvs::launch_packed_csr(c->stream, (const unsigned long long*)c->st_sorted, n, shift, rows, col_mask, nullptr,
                               (int32_t*)c->st_col[out.set], nullptr, (uint16_t*)c->st_q[out.set], nullptr);
        rc = check_kernel("k_packed_csr(16-bit q)");
        if (rc) return rc;
    }
    if (out.row_ptr[(size_t)rows] != n) return fail(MVS_E_HIP, "internal: row index of the sorted cells is inconsistent");
    HIP_TRY(hipEventRecord(c->dl_ready[out.set], c->stream));          // the downloads of this block wait for exactly this point
    return MVS_OK;
}

// rows [rb, re) of the dense byte matrix (first row dense_row0, leading dimension ld) are final: count, scan, fill.
// *odd: some kept cell of the launches so far has a q the byte cannot hold -- the caller redoes the block as a list.
int csr_from_dense(mvs_ctx* c, int64_t rb, int64_t re, int64_t n_cols, int64_t dense_row0, int64_t ld, int64_t block_index,
                   BlockCsr& out, bool* odd, hipStream_t ps, mvs::DenseActive active, bool want_sizes) {
    active.row_rel0 = rb - dense_row0;
    const int64_t rows = re - rb;
    out.rb = rb;
    out.re = re;
    out.wide = false;
    out.set = (int)(block_index & 1);
    out.row_ptr.assign((size_t)rows + 1, 0);
    int rc = ensure_buf(c, &c->st_rowptr, &c->st_rowptr_bytes, (size_t)(rows + 1) * 8);
    if (rc) return rc;
    rc = ensure_buf(c, &c->st_counts, &c->st_counts_bytes, (size_t)(rows + 1) * 8);
    if (rc) return rc;
    // the active tiles of the block's tile rows, every row's first / last kept column
    int tr0 = 0, n_trows = 0, n_tc = 0;
    mvs::dense_tile_rows(active, rows, n_cols, &tr0, &n_trows, &n_tc);
    rc = ensure_buf(c, &c->st_tlist, &c->st_tlist_bytes, std::max<size_t>((size_t)n_trows * (size_t)n_tc * 4, 4));
    if (rc == MVS_OK) rc = ensure_buf(c, &c->st_tlist_n, &c->st_tlist_n_bytes, std::max<size_t>((size_t)n_trows * 4, 4));
    if (rc == MVS_OK) rc = ensure_buf(c, &c->st_ends, &c->st_ends_bytes, std::max<size_t>((size_t)rows * sizeof(int2), 8));
    if (rc) return rc;
    const uint8_t* first = (const uint8_t*)c->st_dense + (size_t)(rb - dense_row0) * (size_t)ld;
    HIP_TRY(hipMemsetAsync((char*)c->st_counts + (size_t)rows * 8, 0, 8, ps));
    mvs::launch_dense_count(ps, first, ld, n_cols, rows, (long long*)c->st_counts, (int2*)c->st_ends, active, (int*)c->st_tlist,
                            (int*)c->st_tlist_n);
    rc = check_kernel("k_dense_count");
    if (rc) return rc;
    size_t need = 0;
    rc = mvs::dense_row_ptr(ps, (long long*)c->st_counts, (long long*)c->st_rowptr, rows, nullptr, 0, &need);
    if (rc) return fail(rc, "scan sizing failed");
    rc = ensure_buf(c, &c->pw_sort, &c->pw_sort_bytes, need);
    if (rc) return rc;
    rc = mvs::dense_row_ptr(ps, (long long*)c->st_counts, (long long*)c->st_rowptr, rows, c->pw_sort, c->pw_sort_bytes, nullptr);
    if (rc) return fail(rc, "scan of the row counts failed");
    unsigned int h_odd = 0;
    rc = read_back(c, ps, {{out.row_ptr.data(), c->st_rowptr, (size_t)(rows + 1) * 8}, {&h_odd, c->d_counter + 4, 4}});
    if (rc) return rc;
    *odd = h_odd != 0;
    if (*odd) return MVS_OK;
    out.n = out.row_ptr[(size_t)rows];
    rc = claim_csr_set(c, out.set, block_index, out.n, false, ps);
    if (rc) return rc;
    // rows that will be encoded on the device: the record sizes come out of the fill pass (k_enc_size would read the CSR
    // arrays this pass is writing once more)
    if (want_sizes && rows > 0) {
        rc = ensure_buf(c, &c->en_size, &c->en_size_bytes, (size_t)(rows + 1) * 8);
        if (rc == MVS_OK) rc = ensure_buf(c, &c->en_jac, &c->en_jac_bytes, (size_t)rows * 4);
        if (rc == MVS_OK) rc = ensure_buf(c, &c->en_first, &c->en_first_bytes, (size_t)rows * 4);
        if (rc == MVS_OK) rc = ensure_buf(c, &c->en_par, &c->en_par_bytes, (size_t)rows * sizeof(mvs::EncRow));
        if (rc) return rc;
    }
    const bool sizes = want_sizes && rows > 0 && out.n > 0;
    mvs::launch_dense_fill(ps, first, ld, n_cols, rows, (const long long*)c->st_rowptr, (int32_t*)c->st_col[out.set],
                           (uint8_t*)c->st_q[out.set], active, (const int*)c->st_tlist, (const int*)c->st_tlist_n,
                           (const int2*)c->st_ends, sizes ? (unsigned long long*)c->en_size : nullptr, (unsigned int*)c->en_jac,
                           (unsigned int*)c->en_first, (mvs::EncRow*)c->en_par);
    rc = check_kernel("k_dense_fill");
    if (rc) return rc;
    out.sizes_ready = sizes;
    HIP_TRY(hipEventRecord(c->dl_ready[out.set], ps));
    return MVS_OK;
}

// The row passes of a dense block WITHOUT a host round trip in their middle (rows delivered encoded): csr_from_dense reads the
// block's cell count back before it sizes the CSR arrays, encode_block the record sizes before it sizes the record buffer -- two
// host round trips per block, each of them a bubble in front of kernels that then queue behind the link's copy kernels.  Here the
// arrays are sized from what the blocks before this one held (cap_cells, cap_bytes: the caller's estimate with head room), all
// passes are queued in one go -- the kernels drop what would not fit --, and ONE read-back at the end brings the row index, the
// record directory and the totals: *held says whether the sizes held (if not, nothing of this block has been delivered and the
// caller does it again the careful way).
int rows_from_dense_spec(mvs_ctx* c, int64_t rb, int64_t re, int64_t n_cols, int64_t dense_row0, int64_t ld, int64_t block_index,
                         BlockCsr& out, bool* odd, hipStream_t ps, mvs::DenseActive active, int64_t cap_cells, size_t cap_bytes,
                         int stage_words, bool* held) {
    *held = false;
    *odd = false;
    active.row_rel0 = rb - dense_row0;
    const int64_t rows = re - rb;
    out.rb = rb;
    out.re = re;
    out.wide = false;
    out.set = (int)(block_index & 1);
    out.row_ptr.assign((size_t)rows + 1, 0);
    int rc = ensure_buf(c, &c->st_rowptr, &c->st_rowptr_bytes, (size_t)(rows + 1) * 8);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->st_counts, &c->st_counts_bytes, (size_t)(rows + 1) * 8);
    if (rc) return rc;
    int tr0 = 0, n_trows = 0, n_tc = 0;
    mvs::dense_tile_rows(active, rows, n_cols, &tr0, &n_trows, &n_tc);
    rc = ensure_buf(c, &c->st_tlist, &c->st_tlist_bytes, std::max<size_t>((size_t)n_trows * (size_t)n_tc * 4, 4));
    if (rc == MVS_OK) rc = ensure_buf(c, &c->st_tlist_n, &c->st_tlist_n_bytes, std::max<size_t>((size_t)n_trows * 4, 4));
    if (rc == MVS_OK) rc = ensure_buf(c, &c->st_ends, &c->st_ends_bytes, std::max<size_t>((size_t)rows * sizeof(int2), 8));
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_size, &c->en_size_bytes, (size_t)(rows + 1) * 8);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_off, &c->en_off_bytes, (size_t)(rows + 1) * 8);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_jac, &c->en_jac_bytes, (size_t)rows * 4);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_first, &c->en_first_bytes, (size_t)rows * 4);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_par, &c->en_par_bytes, (size_t)rows * sizeof(mvs::EncRow));
    if (rc == MVS_OK) rc = claim_csr_set(c, out.set, block_index, cap_cells, false, ps);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->st_enc[out.set], &c->st_enc_bytes[out.set], std::max<size_t>(cap_bytes, 8));
    if (rc) return rc;
    size_t need = 0, need2 = 0;
    rc = mvs::dense_row_ptr(ps, (long long*)c->st_counts, (long long*)c->st_rowptr, rows, nullptr, 0, &need);
    if (rc == MVS_OK) rc = mvs::encode_offsets(ps, (unsigned long long*)c->en_size, (unsigned long long*)c->en_off, rows, nullptr, 0, &need2);
    if (rc) return fail(rc, "scan sizing failed");
    rc = ensure_buf(c, &c->pw_sort, &c->pw_sort_bytes, std::max(need, need2));
    if (rc) return rc;
    const uint8_t* first = (const uint8_t*)c->st_dense + (size_t)(rb - dense_row0) * (size_t)ld;
    HIP_TRY(hipMemsetAsync((char*)c->st_counts + (size_t)rows * 8, 0, 8, ps));
    mvs::launch_dense_count(ps, first, ld, n_cols, rows, (long long*)c->st_counts, (int2*)c->st_ends, active, (int*)c->st_tlist,
                            (int*)c->st_tlist_n);
    rc = check_kernel("k_dense_count");
    if (rc) return rc;
    rc = mvs::dense_row_ptr(ps, (long long*)c->st_counts, (long long*)c->st_rowptr, rows, c->pw_sort, c->pw_sort_bytes, nullptr);
    if (rc) return fail(rc, "scan of the row counts failed");
    HIP_TRY(hipMemsetAsync((char*)c->en_size + (size_t)rows * 8, 0, 8, ps));
    mvs::launch_dense_fill(ps, first, ld, n_cols, rows, (const long long*)c->st_rowptr, (int32_t*)c->st_col[out.set],
                           (uint8_t*)c->st_q[out.set], active, (const int*)c->st_tlist, (const int*)c->st_tlist_n,
                           (const int2*)c->st_ends, (unsigned long long*)c->en_size, (unsigned int*)c->en_jac, (unsigned int*)c->en_first,
                           (mvs::EncRow*)c->en_par, (long long)cap_cells);
    rc = check_kernel("k_dense_fill");
    if (rc) return rc;
    rc = mvs::encode_offsets(ps, (unsigned long long*)c->en_size, (unsigned long long*)c->en_off, rows, c->pw_sort, c->pw_sort_bytes, nullptr);
    if (rc) return fail(rc, "scan of the record sizes failed");
    HIP_TRY(hipMemsetAsync(c->st_enc[out.set], 0, cap_bytes, ps));        // the unary parts are OR-ed into zeroed words
    mvs::launch_encode_fill(ps, (const long long*)c->st_rowptr, (const int32_t*)c->st_col[out.set], c->st_q[out.set], 1, rows,
                            (const unsigned long long*)c->en_off, (const mvs::EncRow*)c->en_par, (unsigned char*)c->st_enc[out.set],
                            stage_words, (unsigned long long)cap_cells, (unsigned long long)cap_bytes);
    rc = check_kernel("k_enc_fill");
    if (rc) return rc;
    out.enc_off.assign((size_t)rows + 1, 0);
    out.enc_jac.assign((size_t)rows, 0);
    out.enc_first.assign((size_t)rows, 0);
    unsigned int h_odd = 0;
    rc = read_back(c, ps, {{out.row_ptr.data(), c->st_rowptr, (size_t)(rows + 1) * 8},
                           {out.enc_off.data(), c->en_off, (size_t)(rows + 1) * 8},
                           {out.enc_jac.data(), c->en_jac, (size_t)rows * 4},
                           {out.enc_first.data(), c->en_first, (size_t)rows * 4},
                           {&h_odd, c->d_counter + 4, 4}});
    if (rc) return rc;
    *odd = h_odd != 0;
    out.n = out.row_ptr[(size_t)rows];
    if (*odd || out.n > cap_cells || out.enc_off[(size_t)rows] > (uint64_t)cap_bytes) return MVS_OK;     // (*held stays false)
    out.sizes_ready = true;
    out.encoded = true;
    *held = true;
    HIP_TRY(hipEventRecord(c->dl_ready[out.set], ps));
    return MVS_OK;
}

// the block's CSR arrays out through the two pinned buffers, in pieces of whole rows; the host blocks here only on the
// pinned buffers (the device is free to run the next block's comparison meanwhile)
int feed_block(mvs_ctx* c, StreamOut& out, const BlockCsr& b, size_t piece_bytes) {
    const int64_t rows = b.re - b.rb, n = b.n;
    const std::vector<int64_t>& row_ptr = b.row_ptr;
    const bool wide = b.wide;
    int rc = MVS_OK;
    const size_t cell_bytes = wide ? 6 : 5;
    const int64_t piece_cells = std::max<int64_t>(1, (int64_t)(piece_bytes / cell_bytes));
    // a piece = as many whole rows as fit piece_bytes; one row alone may exceed that
    auto piece_end = [&](int64_t r0) {
        int64_t r1 = r0 + 1;
        const int64_t c0 = row_ptr[(size_t)r0];
        if (row_ptr[(size_t)r1] - c0 <= piece_cells) {
            const int64_t* end = std::upper_bound(row_ptr.data() + r1, row_ptr.data() + rows + 1, c0 + piece_cells);
            r1 = std::max<int64_t>(r1, (end - row_ptr.data()) - 1);
        }
        return r1;
    };
    // a pinned buffer is sized for the block's largest piece when the producer takes it (it is idle then)
    size_t need_bytes = std::min<size_t>(piece_bytes, std::max<size_t>((size_t)n * cell_bytes, 1u << 16));
    for (int64_t r0 = 0; r0 < rows;) {
        const int64_t r1 = piece_end(r0);
        need_bytes = std::max(need_bytes, (size_t)(row_ptr[(size_t)r1] - row_ptr[(size_t)r0]) * cell_bytes);
        r0 = r1;
    }
    const bool dma = c->opt.stream_copy != 0;
    if (dma) {
        rc = ensure_hsa_copy(c);
        if (rc) return rc;
        HIP_TRY(hipEventSynchronize(c->dl_ready[b.set]));      // the block's arrays are final (this thread only feeds)
    }
    for (int64_t r0 = 0; r0 < rows;) {
        const int64_t r1 = piece_end(r0);
        const int64_t c0 = row_ptr[(size_t)r0];
        const int64_t cells = row_ptr[(size_t)r1] - c0;
        if (out.failed()) return MVS_OK;                        // the caller reports the callback's status
        const int sl = out.acquire_slot();
        rc = ensure_pinned_slot(c, sl, need_bytes);
        if (rc) {
            out.release_slot(sl);
            return rc;
        }
        StreamOut::Item it;
        it.slot = sl;
        it.row_begin = b.rb + r0;
        it.row_end = b.rb + r1;
        it.n_cells = cells;
        it.wide = wide;
        it.row_ptr.resize((size_t)(r1 - r0) + 1);
        for (int64_t r = r0; r <= r1; ++r) it.row_ptr[(size_t)(r - r0)] = row_ptr[(size_t)r] - c0;
        it.dma = dma;
        it.last_of_block = dma && r1 == rows;
        char* dst = static_cast<char*>(c->dl_pinned[sl]);
        if (dma) {
            const HsaCopy* h = static_cast<const HsaCopy*>(c->dl_hsa);
            hsa_signal_store_relaxed(h->sig[sl], cells > 0 ? 2 : 0);       // two copies (columns, q) count it down
            hsa_status_t hs = HSA_STATUS_SUCCESS;
            if (cells > 0) {
                hs = hsa_amd_memory_async_copy(dst, h->cpu, (const char*)c->st_col[b.set] + (size_t)c0 * 4, h->gpu, (size_t)cells * 4, 0, nullptr,
                                               h->sig[sl]);
                if (hs == HSA_STATUS_SUCCESS)
                    hs = hsa_amd_memory_async_copy(dst + (size_t)cells * 4, h->cpu, (const char*)c->st_q[b.set] + (size_t)c0 * (wide ? 2 : 1),
                                                   h->gpu, (size_t)cells * (wide ? 2 : 1), 0, nullptr, h->sig[sl]);
            }
            if (hs != HSA_STATUS_SUCCESS) {
                hsa_signal_store_relaxed(h->sig[sl], 0);
                out.release_slot(sl);
                return fail(MVS_E_HIP, "DMA download of a row block failed (hsa status %d)", (int)hs);
            }
            out.push(std::move(it));
            ++c->st_pieces;
            c->st_bytes += (long long)((size_t)cells * cell_bytes);
            r0 = r1;
            continue;
        }
        hipError_t e = hipStreamWaitEvent(c->dl_stream, c->dl_ready[b.set], 0);
        if (e == hipSuccess && cells > 0) {
            e = hipMemcpyAsync(dst, (const char*)c->st_col[b.set] + (size_t)c0 * 4, (size_t)cells * 4, hipMemcpyDeviceToHost,
                               c->dl_stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(dst + (size_t)cells * 4, (const char*)c->st_q[b.set] + (size_t)c0 * (wide ? 2 : 1),
                                   (size_t)cells * (wide ? 2 : 1), hipMemcpyDeviceToHost, c->dl_stream);
        }
        if (e == hipSuccess) e = hipEventRecord(c->dl_done[sl], c->dl_stream);
        if (e != hipSuccess) {
            out.release_slot(sl);
            return fail(MVS_E_HIP, "download of a row block: %s", hipGetErrorString(e));
        }
        out.push(std::move(it));
        ++c->st_pieces;
        c->st_bytes += (long long)((size_t)cells * cell_bytes);
        r0 = r1;
    }
    if (!dma) HIP_TRY(hipEventRecord(c->dl_block[b.set], c->dl_stream));
    return MVS_OK;
}

// the block's encoded records out through the pinned buffers, in pieces of whole rows of at most piece_bytes
int feed_encoded(mvs_ctx* c, StreamOut& out, const BlockCsr& b, size_t piece_bytes) {
    const int64_t rows = b.re - b.rb;
    const std::vector<uint64_t>& off = b.enc_off;
    auto piece_end = [&](int64_t r0) {
        int64_t r1 = r0 + 1;
        const uint64_t o0 = off[(size_t)r0];
        if (off[(size_t)r1] - o0 <= piece_bytes) {
            const uint64_t* end = std::upper_bound(off.data() + r1, off.data() + rows + 1, o0 + (uint64_t)piece_bytes);
            r1 = std::max<int64_t>(r1, (end - off.data()) - 1);
        }
        return r1;
    };
    size_t need_bytes = std::min<size_t>(piece_bytes, std::max<size_t>((size_t)off[(size_t)rows], 1u << 16));
    for (int64_t r0 = 0; r0 < rows;) {
        const int64_t r1 = piece_end(r0);
        need_bytes = std::max(need_bytes, (size_t)(off[(size_t)r1] - off[(size_t)r0]));
        r0 = r1;
    }
    int rc = MVS_OK;
    const bool dma = c->opt.stream_copy != 0;
    if (dma) {
        rc = ensure_hsa_copy(c);
        if (rc) return rc;
        HIP_TRY(hipEventSynchronize(c->dl_ready[b.set]));
    }
    for (int64_t r0 = 0; r0 < rows;) {
        const int64_t r1 = piece_end(r0);
        const uint64_t o0 = off[(size_t)r0], bytes = off[(size_t)r1] - o0;
        if (out.failed()) return MVS_OK;
        const int sl = out.acquire_slot();
        rc = ensure_pinned_slot(c, sl, need_bytes);
        if (rc) {
            out.release_slot(sl);
            return rc;
        }
        StreamOut::Item it;
        it.slot = sl;
        it.row_begin = b.rb + r0;
        it.row_end = b.rb + r1;
        it.n_cells = b.row_ptr[(size_t)r1] - b.row_ptr[(size_t)r0];
        it.wide = b.wide;
        it.n_bytes = (int64_t)bytes;
        for (int64_t r = r0; r < r1; ++r)
            if (b.row_ptr[(size_t)r + 1] > b.row_ptr[(size_t)r]) {
                it.rows.push_back((uint32_t)(b.rb + r));
                it.first_col.push_back(b.enc_first[(size_t)r]);
                it.jac_bytes.push_back(b.enc_jac[(size_t)r]);
                it.offset.push_back(off[(size_t)r] - o0);
            }
        it.dma = dma;
        it.last_of_block = dma && r1 == rows;
        if (dma) {
            const HsaCopy* h = static_cast<const HsaCopy*>(c->dl_hsa);
            hsa_signal_store_relaxed(h->sig[sl], bytes > 0 ? 1 : 0);
            const hsa_status_t hs = bytes > 0 ? hsa_amd_memory_async_copy(c->dl_pinned[sl], h->cpu, (const char*)c->st_enc[b.set] + o0, h->gpu,
                                                                          (size_t)bytes, 0, nullptr, h->sig[sl])
                                              : HSA_STATUS_SUCCESS;
            if (hs != HSA_STATUS_SUCCESS) {
                hsa_signal_store_relaxed(h->sig[sl], 0);
                out.release_slot(sl);
                return fail(MVS_E_HIP, "DMA download of encoded rows failed (hsa status %d)", (int)hs);
            }
            out.push(std::move(it));
            ++c->st_pieces;
            c->st_bytes += (long long)bytes;
            r0 = r1;
            continue;
        }
        hipError_t e = hipStreamWaitEvent(c->dl_stream, c->dl_ready[b.set], 0);
        if (e == hipSuccess && bytes > 0)
            e = hipMemcpyAsync(c->dl_pinned[sl], (const char*)c->st_enc[b.set] + o0, (size_t)bytes, hipMemcpyDeviceToHost, c->dl_stream);
        if (e == hipSuccess) e = hipEventRecord(c->dl_done[sl], c->dl_stream);
        if (e != hipSuccess) {
            out.release_slot(sl);
            return fail(MVS_E_HIP, "download of encoded rows: %s", hipGetErrorString(e));
        }
        out.push(std::move(it));
        ++c->st_pieces;
        c->st_bytes += (long long)bytes;
        r0 = r1;
    }
    if (!dma) HIP_TRY(hipEventRecord(c->dl_block[b.set], c->dl_stream));
    return MVS_OK;
}

}  // namespace mvs_capi

namespace mvs_capi {
int pairwise_stream_impl(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int mem_norms, int keep_mode,
                         int64_t row_begin, int64_t row_end, size_t device_budget_bytes, mvs_row_block_cb cb,
                         mvs_encoded_rows_cb ecb, void* user, int64_t* n_cells);
}

extern "C" {

int mvs_pairwise_stream(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int mem_norms, int keep_mode,
                        int64_t row_begin, int64_t row_end, size_t device_budget_bytes, mvs_row_block_cb cb, void* user,
                        int64_t* n_cells) {
    try {       // the host side keeps per-row directories in std::vector: no exception may cross the C boundary
        return pairwise_stream_impl(c, s, norms_sq, mem_norms, keep_mode, row_begin, row_end, device_budget_bytes, cb, nullptr, user,
                                    n_cells);
    } catch (const std::bad_alloc&) {
        return fail(MVS_E_NOMEM, "out of host memory while streaming the comparison result");
    } catch (const std::exception& e) {
        return fail(MVS_E_HIP, "mvs_pairwise_stream: %s", e.what());
    }
}

int mvs_pairwise_stream_encoded(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int mem_norms, int keep_mode,
                                int64_t row_begin, int64_t row_end, size_t device_budget_bytes, mvs_encoded_rows_cb cb, void* user,
                                int64_t* n_cells) {
    try {
        return pairwise_stream_impl(c, s, norms_sq, mem_norms, keep_mode, row_begin, row_end, device_budget_bytes, nullptr, cb, user,
                                    n_cells);
    } catch (const std::bad_alloc&) {
        return fail(MVS_E_NOMEM, "out of host memory while streaming the comparison result");
    } catch (const std::exception& e) {
        return fail(MVS_E_HIP, "mvs_pairwise_stream_encoded: %s", e.what());
    }
}

}  // extern "C"

namespace mvs_capi {
int pairwise_stream_impl(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int mem_norms, int keep_mode,
                         int64_t row_begin, int64_t row_end, size_t device_budget_bytes, mvs_row_block_cb cb,
                         mvs_encoded_rows_cb ecb, void* user, int64_t* n_cells) {
    if (!c || !s || (!cb && !ecb)) return fail(MVS_E_INVALID, "NULL argument");
    const Range range(c, "mvs_pairwise_stream");
    if (n_cells) *n_cells = 0;
    if (!mem_ok(mem_norms) || (keep_mode != MVS_KEEP_INT32 && keep_mode != MVS_KEEP_INT16)) return fail(MVS_E_INVALID, "bad argument");
    if (row_begin < 0 || row_end > s->n || row_begin > row_end)
        return fail(MVS_E_INVALID, "row range [%lld,%lld) outside [0,%lld)", (long long)row_begin, (long long)row_end, (long long)s->n);
    if (row_begin == row_end || s->n == 0) return MVS_OK;
    if (!norms_sq) return fail(MVS_E_INVALID, "norms_sq is NULL");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf dn;
    const double* d_n2 = norms_sq;
    if (mem_norms == MVS_MEM_HOST) {
        HIP_TRY(dn.alloc((size_t)s->n * 8));
        HIP_TRY(hipMemcpyAsync(dn.p, norms_sq, (size_t)s->n * 8, hipMemcpyHostToDevice, c->stream));
        d_n2 = (const double*)dn.p;
    }
    // Device budget for the kept cells of one row block (raw + sorted words, CSR arrays: 21-22 bytes per cell): a quarter of
    // what is free now unless the caller says otherwise.  Only a block that goes through the exact kernel is planned
    // against it (worst case: every cell kept); the two-stage comparison's output is sized from its candidate count.
    size_t budget = device_budget_bytes;
    if (budget == 0) {
        // Default: a quarter of what is free, but no more than 2^30 worst-case cells per block (8 GiB of packed words):
        // where the exact kernel runs the result is dense and the link, not the kernel, sets the pace -- blocks of that
        // size keep the head of the pipeline (first block computed, nothing to download yet) short.
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        budget = std::min<size_t>(free_b / 4, (size_t)22 << 30);
    }
    const int64_t budget_cells = std::max<int64_t>(1 << 16, (int64_t)(budget / 22));
    // the dense byte matrix (one byte per cell of a row block) may take more: a third of what is free unless the caller set a budget
    size_t dense_budget = device_budget_bytes;
    if (dense_budget == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        dense_budget = free_b / 3;
    }
    const size_t piece_bytes = (size_t)c->opt.stream_piece_mib << 20;   // pinned buffer size (32 MiB): pinning costs ~0.3 ms per MiB
    const int col_bits = bits_for(std::max<int64_t>(s->n - 1, 1));
    const int shift = 16 + col_bits;
    int rc = ensure_download_side(c);
    if (rc) return rc;
    c->st_kernel_ms = 0.0;
    c->st_bytes = c->st_blocks = c->st_pieces = c->st_two_stage = 0;
    bool tiles_phase = false;   // the launches being timed are runs of flagged tiles (ev[6] .. ev[3]), not whole comparisons
    auto add_kernel_ms = [&]() {
        float ms = 0.0f;
        if (c->timing && c->ev_valid[1] && hipEventSynchronize(c->ev[3]) == hipSuccess &&
            hipEventElapsedTime(&ms, tiles_phase ? c->ev[6] : c->ev[2], c->ev[3]) == hipSuccess)
            c->st_kernel_ms += ms;
    };
    StreamOut out;
    out.c = c;
    out.cb = cb;
    out.ecb = ecb;
    out.user = user;
    // a block's way out, in two steps so that the next block's comparison can be queued between them: prepare = the
    // device-side work that is left (encoding the rows, where the caller asked for that), deliver = pieces to the link
    hipStream_t ps = c->stream;                                // where a block is turned into CSR / encoded rows (see `side`)
    bool side = false;
    auto prepare = [&](BlockCsr& blk) -> int { return (ecb && !blk.encoded) ? encode_block(c, blk, ps) : MVS_OK; };
    // what the dense blocks so far held per row (cells, record bytes): sizes the next block's buffers when its row passes are
    // queued without reading its own totals first (rows_from_dense_spec, option stream_spec)
    double spec_cells_per_row = -1.0, spec_bytes_per_row = -1.0;
    auto deliver = [&](BlockCsr& blk) -> int {
        auto sp = std::make_shared<BlockCsr>(std::move(blk));
        const bool enc = ecb != nullptr;
        StreamOut* o = &out;
        out.enqueue_feed([c, o, sp, enc, piece_bytes]() -> int {
            return enc ? feed_encoded(c, *o, *sp, piece_bytes) : feed_block(c, *o, *sp, piece_bytes);
        });
        return MVS_OK;
    };
    // the arrays of block k live in set k & 1: before block k is built the feeder must be through with block k - 2
    auto wait_for_set = [&](int64_t k) {
        if (k >= 2) out.wait_fed(k - 1);
        if (k >= 2 && c->opt.stream_copy != 0) out.wait_done(k - 1);      // (DMA copies: block k - 2 has left the device)
    };
    out.worker = std::thread([&out] { out.run(); });
    out.feeder = std::thread([&out] { out.feed_run(); });
    int64_t total = 0;
    // option stream_trace: where the host is when (ms since the call started)
    const auto t_call = std::chrono::steady_clock::now();
    std::vector<std::pair<std::string, double>> trace;
    auto mark = [&](const char* what, long k) {
        if (!c->opt.stream_trace) return;
        char buf[64];
        snprintf(buf, sizeof buf, "%s[%ld]", what, k);
        trace.emplace_back(buf, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    };
    auto finish = [&](int status) {
        mark("close", -1);
        out.close_feeder();                                     // every delivered block has been handed to the link
        out.close();                                            // ... and consumed
        (void)hipStreamSynchronize(c->dl_stream);
        if (side) (void)hipStreamSynchronize(c->post_stream);
        mark("done", -1);
        if (c->opt.stream_trace) {
            std::string line = "[mvs stream trace]";
            for (auto& t : trace) {
                char buf[96];
                snprintf(buf, sizeof buf, " %s %.2f", t.first.c_str(), t.second);
                line += buf;
            }
            fprintf(stderr, "%s\n", line.c_str());
        }
        if (n_cells) *n_cells = total;
        if (status != MVS_OK) return status;
        if (!out.error.empty()) return fail(MVS_E_HIP, "%s", out.error.c_str());
        if (out.cb_status != 0) return fail(MVS_E_ABORTED, "the row-block callback returned %d", out.cb_status);
        return MVS_OK;
    };
    const int64_t rows_all = row_end - row_begin;
    // ---------------------------------------------------------------------------------------------------------------
    // How the kept cells leave the device is decided by how dense the result is, which only the filter can tell:
    //  A. sparse: ONE filter pass over the whole row range, candidates re-checked, the few flagged tiles computed, all kept
    //     cells in ONE packed list that is sorted on the device (needs the row field to fit the packed word).
    //  M. dense regions: the dense byte matrix -- one byte per cell, rows -> CSR / encoded rows by count / scan / fill passes
    //     that read only the tiles that can hold something (flagged by the filter, mirror images of those, touched by the
    //     re-check's kept cells: nothing else of the matrix is ever cleared or read) -- in row blocks, so that the link is
    //     fed while the comparison goes on.  Two ways to get there:
    //       M1 (pipeline): the filter itself runs block by block (first block one tile row: its flagged share tells sparse
    //          from dense, and costs 1 % of a whole pass when the answer is "sparse"), so a block's rows are final -- and
    //          on the link -- a millisecond after the call started instead of after the whole filter pass;
    //       M2: plan A's whole filter pass found too many cells for a list: its flags and candidates feed the matrix, the
    //          flagged tiles are computed block by block.
    //  B. the filter does not apply or gave up (nearly every tile dense): the exact kernel in row blocks (dense matrix with
    //     every tile active, or packed lists), as up to round 3.
    // ---------------------------------------------------------------------------------------------------------------
    const bool fits_word = shift + bits_for(std::max<int64_t>(rows_all - 1, 1)) <= 64;
    const int64_t ld = (s->n + 127) / 128 * 128;
    mvs::PairwiseArgs probe{};
    probe.limbs = s->limbs;
    probe.d_pad = s->d_pad;
    // M1 pins the filter kernel the whole range would get, the exact-kernel blocks switch the filter off: both in a COPY of
    // the context's options that the comparison stages below are handed (the context itself is never written: a host that
    // drives several contexts from several threads must not find another call's forced variant in one of them)
    mvs::Options lopt = c->opt;
    const int saved_variant = lopt.filter_variant;
    const bool dense_ok = mvs::exact_kernel_writes_dense(probe, lopt) && lopt.stream_dense != 0;
    const bool matrix_fits = (size_t)rows_all * (size_t)ld <= dense_budget;
    const bool applies = two_stage_applies(c, s, row_begin, row_end, 0, s->n, 0.05, true, &lopt);
    mvs::PairwiseArgs wa{};                                          // the whole row range as one symmetric block
    fill_args(c, s, d_n2, keep_mode, row_begin, row_end, 0, s->n, true, false, 0.05, wa, &lopt);
    // the matrix flows need the tile grids to line up with the matrix's rows and the packed word to hold a row
    const bool can_matrix = applies && dense_ok && matrix_fits && fits_word && row_begin % 256 == 0 && mvs::filter_flags_tiles(wa, lopt);
    int n_tr_all = 0, n_tc_all = 0;
    mvs::filter_tile_grid(wa, &n_tr_all, &n_tc_all);
    const int tile_o = (int)(row_begin / 256);
    enum { kNone, kM1, kM2 } matrix_mode = kNone;
    TwoStage ts;                                                     // M2: the whole pass; M1: the current block's pass
    mvs::DenseActive active{};                                       // flags == NULL: every tile (plan B)
    auto matrix_setup = [&]() -> int {                               // matrix, touch map, list of newly touched tiles
        const void* before = c->st_dense;
        int r = ensure_buf(c, &c->st_dense, &c->st_dense_bytes, (size_t)rows_all * (size_t)ld);
        if (r) return r;
        if (c->st_dense != before) c->st_dense_zero = 0;
        const size_t n_tiles = (size_t)n_tr_all * (size_t)n_tc_all;
        r = ensure_buf(c, &c->pw_ttouch, &c->pw_ttouch_bytes, n_tiles * 4);
        if (r) return r;
        r = ensure_buf(c, &c->pw_tnew, &c->pw_tnew_bytes, (n_tiles + 1) * 4);
        if (r) return r;
        HIP_TRY(hipMemsetAsync(c->pw_ttouch, 0, n_tiles * 4, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_counter + 4, 0, 8, c->stream));      // the "q beyond a byte" flag
        c->st_dense_zero = 0;
        active.touch = (const unsigned int*)c->pw_ttouch;
        active.n_tr = n_tr_all;
        active.n_tc = n_tc_all;
        active.o = tile_o;
        active.sym = (lopt.pairwise_symmetric != 0) ? 1 : 0;
        return MVS_OK;
    };
    // re-check of t's candidates with the kept cells going into the matrix: a packed list first (the re-check decides
    // which candidates are kept), then mark / clear / scatter (mvs_internal.h: launch_packed_to_dense)
    auto recheck_into_matrix = [&](TwoStage& t) -> int {
        int r = ensure_buf(c, &c->st_raw, &c->st_raw_bytes, (size_t)(2 * t.n_cand + 64) * 8);
        if (r) return r;
        t.a.dense = nullptr;
        t.a.packed = (unsigned long long*)c->st_raw;
        t.a.capacity = c->st_raw_bytes / 8;
        t.a.pack_row0 = row_begin;
        t.a.pack_shift = shift;
        r = two_stage_recheck(c, t);
        if (r) return r;
        HIP_TRY(hipMemsetAsync(c->d_counter + 10, 0, 8, c->stream));     // count of newly touched tiles
        mvs::launch_packed_to_dense(c->stream, (const unsigned long long*)c->st_raw, c->d_counter, shift,
                                    (1ULL << col_bits) - 1ULL, (uint8_t*)c->st_dense, ld, rows_all, (unsigned int*)c->pw_ttouch, n_tc_all,
                                    (int*)c->pw_tnew, reinterpret_cast<unsigned int*>(c->d_counter + 10),
                                    reinterpret_cast<unsigned int*>(c->d_counter + 4));
        r = check_kernel("k_packed_touch / k_clear_tiles / k_packed_scatter");
        if (r) return r;
        // from here on t.a describes the exact kernel's launches on the flagged tiles: bytes of whole tiles into the matrix
        t.a.packed = nullptr;
        t.a.dense = (uint8_t*)c->st_dense;
        t.a.dense_row0 = row_begin;
        t.a.dense_ld = ld;
        t.a.dense_flag = reinterpret_cast<unsigned int*>(c->d_counter + 4);
        return MVS_OK;
    };
    // M1, one block: filter its rows (symmetric square = the whole row range), re-check into the matrix, its flagged tiles
    auto pipeline_filter = [&](int64_t rb, int64_t re, TwoStage& t) -> int {
        mvs::PairwiseArgs fa{};
        fill_args(c, s, d_n2, keep_mode, rb, re, 0, s->n, true, false, 0.05, fa, &lopt);
        fa.sym_begin = row_begin;
        fa.sym_end = row_end;
        t = TwoStage();
        t.ext_flags = (unsigned int*)c->pw_tflag + (size_t)((rb - row_begin) / 256) * (size_t)n_tc_all;
        return two_stage_filter(c, s, d_n2, 0.05, 0, true, 0, fa, t, &lopt);
    };
    if (applies) {
        bool whole_pass = true;
        if (can_matrix && rows_all > 512 && lopt.stream_pipeline != 0) {
            // M1's first block doubles as the probe: one tile row of the filter
            rc = ensure_buf(c, &c->pw_tflag, &c->pw_tflag_bytes, (size_t)n_tr_all * (size_t)n_tc_all * 4);
            if (rc) return finish(rc);
            if (lopt.filter_variant < 0) lopt.filter_variant = 8;    // what the whole range gets (filter_flags_tiles said so)
            rc = pipeline_filter(row_begin, row_begin + 256, ts);
            if (rc != MVS_OK && rc != kNeedExact) return finish(rc);
            // share of flagged tiles in this row of tiles, extrapolated to the tiles of the whole pass, as list cells
            const double tiles_all = std::max(1.0, (double)n_tr_all * (double)n_tc_all - 0.5 * (double)n_tr_all * (double)(n_tr_all - 1));
            const bool probe_gave_up = rc == kNeedExact;                 // the pass stopped: dense everywhere (never M1 then)
            const double est = probe_gave_up ? 1e30
                                             : ((double)ts.n_flagged * 131072.0 + 2.0 * (double)ts.n_cand) / (double)n_tc_all * tiles_all;
            if (probe_gave_up) {
                // more than 70 % of the first tile row is dense: its cluster alone covers half of the matrix -- no further
                // filter pass, the exact kernel does the shard (plan B; the set is marked, two_stage_filter did that)
                whole_pass = false;
                lopt.filter_variant = saved_variant;
            } else if (est > (double)lopt.stream_list_cells) {
                matrix_mode = kM1;
                whole_pass = false;
            } else {
                lopt.filter_variant = saved_variant;
            }
        }
        if (whole_pass && two_stage_applies(c, s, row_begin, row_end, 0, s->n, 0.05, true, &lopt)) {
            ts = TwoStage();
            rc = two_stage_filter(c, s, d_n2, 0.05, 0, true, 0, wa, ts, &lopt);
            if (rc != MVS_OK && rc != kNeedExact) return finish(rc);
            if (rc == MVS_OK) {
                const unsigned long long bound = 2 * ts.n_cand + (unsigned long long)ts.n_flagged * 131072ULL;
                size_t free_b = 0, total_b = 0;
                HIP_TRY(hipMemGetInfo(&free_b, &total_b));
                const bool list_fits = fits_word && (double)bound * 22.0 <= (double)free_b * 0.5;
                // stream_list_cells (2^26): below that the list (8 B per cell written, a radix sort over the key bits) is
                // cheaper than counting and filling a matrix of rows x n bytes
                const bool as_list = list_fits && (bound <= (unsigned long long)lopt.stream_list_cells || !can_matrix || ts.n_flagged == 0);
                if (as_list) {
                    rc = ensure_buf(c, &c->st_raw, &c->st_raw_bytes, (size_t)(bound + 64) * 8);
                    if (rc) return finish(rc);
                    ts.a.packed = (unsigned long long*)c->st_raw;
                    ts.a.capacity = c->st_raw_bytes / 8;
                    ts.a.pack_row0 = row_begin;
                    ts.a.pack_shift = shift;
                    rc = two_stage_recheck(c, ts);
                    if (rc == MVS_OK) rc = two_stage_tiles(c, ts, 0, ts.n_flagged, true);
                    if (rc) return finish(rc);
                    unsigned long long got = 0;
                    hipError_t e = hipMemcpyAsync(&got, c->d_counter, 8, hipMemcpyDeviceToHost, c->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                    if (e != hipSuccess) return finish(fail(MVS_E_HIP, "reading the cell count: %s", hipGetErrorString(e)));
                    if ((size_t)got * 8 > c->st_raw_bytes) return finish(fail(MVS_E_HIP, "internal: kept cells beyond the sized output"));
                    total = (int64_t)got;
                    add_kernel_ms();
                    c->st_blocks = 1;
                    c->st_two_stage = 1;
                    BlockCsr blk;
                    rc = csr_from_packed(c, row_begin, row_end, (int64_t)got, shift, col_bits, 0, blk);
                    if (rc == MVS_OK) rc = prepare(blk);
                    if (rc == MVS_OK) rc = deliver(blk);
                    return finish(rc);
                }
                if (can_matrix) matrix_mode = kM2;
                // neither a list nor the matrix fits: plan B (the filter pass was in vain)
            }
        }
    }
    // ---- row blocks ----
    // Plan B proper: the exact kernel, software-pipelined -- block k+1 is launched before block k's pieces are fed to the
    // link.  Two ways for a block's cells to leave the kernel:
    //  * dense (two limbs on the ping-pong kernel): one byte per cell in a row-major matrix.  If the matrix of ALL the rows
    //    fits the budget the blocks share it and the symmetric schedule spans the whole square: a block's launch computes its
    //    tiles on and above the diagonal and writes the mirror images into later blocks' rows, so block k is final when
    //    launch k is.  Otherwise the matrix holds one block at a time and the symmetric schedule works inside each block's
    //    own square only;
    //  * packed list (any other kernel): blocks whose worst case -- every cell kept -- fits the budget.
    // The matrix flows M1 / M2 use the same loop with the shared matrix; only what a block's "launch" is differs.
    bool dense = dense_ok, whole = false;
    int64_t block_rows = 0;
    const bool aligned = row_begin % 128 == 0;                      // the symmetric schedule needs the tile grids to line up
    if (dense) {
        whole = aligned && matrix_fits;                              // (the matrix flows imply both)
        if (whole) {
            block_rows = std::max<int64_t>(2048, (rows_all / 16 + 255) / 256 * 256);   // (1/12 .. 1/6 of the rows measure the same or worse)
        } else {
            block_rows = (int64_t)(dense_budget / (size_t)ld) / 256 * 256;
            if (block_rows < 256) dense = false;                     // not even 256 rows of bytes: list blocks instead
        }
        if (dense && lopt.stream_block_rows > 0)                   // tests: many small blocks on small inputs
            block_rows = std::min<int64_t>(block_rows, std::max<int64_t>(256, (int64_t)lopt.stream_block_rows / 256 * 256));
    }
    if (!dense) {
        block_rows = std::max<int64_t>(256, budget_cells / std::max<int64_t>(s->n, 1) / 256 * 256);
        while (shift + bits_for(std::max<int64_t>(block_rows - 1, 1)) > 64 && block_rows > 256) block_rows /= 2;
    }
    std::vector<std::pair<int64_t, int64_t>> blocks;
    // blocks of one shared matrix start small (M1: one tile row, the probe; otherwise 1024 rows) and double: the link has
    // nothing to do until the first block has been compared, counted, filled and encoded
    int64_t ramp = block_rows;
    if (dense && whole && lopt.stream_block_rows == 0 && block_rows > 1024) ramp = 1024;
    if (matrix_mode == kM1) ramp = 256;
    for (int64_t rb = row_begin; rb < row_end;) {
        const int64_t re = std::min(row_end, (rb / 256) * 256 + std::min(ramp, block_rows));
        blocks.emplace_back(rb, re);
        rb = re;
        ramp = std::min(block_rows, ramp * 2);
    }
    if (matrix_mode != kNone) {
        rc = matrix_setup();
        if (rc) return finish(rc);
        active.flags = matrix_mode == kM1 ? (const unsigned int*)c->pw_tflag : (const unsigned int*)ts.a.tile_flag;
        if (matrix_mode == kM1) {
            // flags of blocks not yet filtered read as "not flagged"; block 0 has been filtered already (the probe)
            const size_t done = (size_t)n_tc_all;
            HIP_TRY(hipMemsetAsync((unsigned int*)c->pw_tflag + done, 0, ((size_t)n_tr_all * (size_t)n_tc_all - done) * 4, c->stream));
        }
        rc = recheck_into_matrix(ts);                                // M2: all candidates; M1: block 0's
        if (rc) return finish(rc);
        add_kernel_ms();                                             // filter + re-check
        c->st_two_stage = matrix_mode == kM1 ? 3 : 2;
    } else if (dense) {
        const size_t bytes = (size_t)(whole ? rows_all : std::min(block_rows + 256, rows_all)) * (size_t)ld;
        const void* before = c->st_dense;
        rc = ensure_buf(c, &c->st_dense, &c->st_dense_bytes, bytes);
        if (rc) return finish(rc);
        (void)before;
        c->st_dense_zero = 0;
        hipError_t e = hipMemsetAsync(c->d_counter + 4, 0, 8, c->stream);      // the "q beyond a byte" flag
        if (e != hipSuccess) return finish(fail(MVS_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e)));
    }
    // M1: which blocks open a filter segment.  Segment 0 is the probe (one tile row); the others end where 4 %, 16 % and 45 %
    // of the rows are done -- by the tiles of the symmetric square that is 8 %, 22 %, 40 % and 30 % of the filter's work --
    // option stream_block_rows (tests) makes every block a segment of its own.
    std::vector<char> seg_first(blocks.size(), 0);
    int64_t seg_row0 = row_begin;
    bool seg_exact = false;
    if (matrix_mode == kM1) {
        const double marks[3] = {0.04, 0.16, 0.45};   // (0.05 / 0.3, 0.03 / 0.12 / 0.3, 0.1 / 0.4 measure the same within 2 %)
        int next_mark = 0;
        for (size_t k = 0; k < blocks.size(); ++k) {
            const double done = (double)(blocks[k].first - row_begin) / (double)rows_all;
            bool opens = k <= 1 || lopt.stream_block_rows > 0;
            while (next_mark < 3 && done >= marks[next_mark]) {
                opens = true;
                ++next_mark;
            }
            seg_first[k] = opens ? 1 : 0;
        }
    }
    mvs::Options exact_opt;                                          // (launch_exact: lopt with the filter switched off)
    // the exact kernel on every tile of rows [rb, re) (plan B; also a block of M1 whose filter pass gave up)
    auto launch_exact = [&](int64_t rb, int64_t re, bool as_dense) -> int {
        unsigned long long got = 0;
        exact_opt = lopt;
        exact_opt.pairwise_filter = 0;
        int r;
        if (as_dense) {
            DenseOut dno{(uint8_t*)c->st_dense, whole ? row_begin : rb, ld, whole ? row_begin : rb, whole ? row_end : re,
                         reinterpret_cast<unsigned int*>(c->d_counter + 4)};
            r = pairwise_launch(c, s, d_n2, keep_mode, rb, re, 0, s->n, true, false, nullptr, 0, 0, &got, 0.05, nullptr, &dno, &exact_opt);
        } else {
            const int64_t worst = (re - rb) * s->n;
            r = ensure_buf(c, &c->st_raw, &c->st_raw_bytes, (size_t)worst * 8);
            if (r == MVS_OK) {
                PackedOut po{&c->st_raw, &c->st_raw_bytes, rb, shift, false};
                r = pairwise_launch(c, s, d_n2, keep_mode, rb, re, 0, s->n, true, false, nullptr, 0, 0, &got, 0.05, &po, nullptr, &exact_opt);
            }
        }
        return r;
    };
    auto launch = [&](size_t k, bool as_dense) -> int {
        const int64_t rb = blocks[k].first, re = blocks[k].second;
        if (matrix_mode == kM2 && as_dense) {                        // this block's share of the whole pass's flagged tiles
            tiles_phase = true;
            const int t0 = (int)((rb - row_begin) / 256), t1 = (int)std::min<int64_t>(ts.n_tr, (re - row_begin + 255) / 256);
            return two_stage_tiles(c, ts, ts.row_first[(size_t)t0], ts.row_first[(size_t)t1] - ts.row_first[(size_t)t0], true);
        }
        if (matrix_mode == kM1 && as_dense) {
            // The filter runs per SEGMENT of consecutive row blocks (seg_first: the blocks that open one; the probe's tile
            // row is segment 0): few passes -- each costs a launch over the whole column range and a host round trip -- yet
            // the first rows are final, and on the link, a millisecond after the call started.
            int r = MVS_OK;
            const bool opens = k < seg_first.size() && seg_first[k];
            if (k == 0) {
                seg_row0 = rb;                                       // the probe's tile row: filtered and re-checked already
                seg_exact = false;
            } else if (opens) {
                size_t last = k;
                while (last + 1 < blocks.size() && !seg_first[last + 1]) ++last;
                r = pipeline_filter(rb, blocks[last].second, ts);
                seg_row0 = rb;
                seg_exact = r == kNeedExact;
                if (r == MVS_OK) r = recheck_into_matrix(ts);
            } else if (seg_exact) {
                r = kNeedExact;
            }
            tiles_phase = !(opens && k > 0) && !seg_exact;           // a block that opens a segment is timed ev[2] .. ev[3]
            if (r == kNeedExact) {
                // nearly every tile of these rows is dense: the exact kernel on all of them.  Flag the tiles it writes itself
                // -- outside the square, on and above its diagonal -- so that the row passes read them and their mirror
                // images; the tiles below the diagonal stay what earlier blocks made of them
                const int t0 = (int)((rb - row_begin) / 256), t1 = (int)((re - row_begin + 255) / 256);
                for (int t = t0; t < t1; ++t) {
                    unsigned int* rowf = (unsigned int*)c->pw_tflag + (size_t)t * (size_t)n_tc_all;
                    hipError_t e = hipSuccess;
                    if (tile_o > 0) e = hipMemsetD32Async((hipDeviceptr_t)rowf, 1, (size_t)tile_o, c->stream);
                    if (e == hipSuccess && t + tile_o < n_tc_all)
                        e = hipMemsetD32Async((hipDeviceptr_t)(rowf + t + tile_o), 1, (size_t)(n_tc_all - t - tile_o), c->stream);
                    if (e != hipSuccess) return fail(MVS_E_HIP, "hipMemsetD32Async: %s", hipGetErrorString(e));
                }
                c->filter_off_id = 0;                                // a verdict on these rows, not on the set
                return launch_exact(rb, re, true);
            }
            if (r) return r;
            // this block's share of the segment's flagged tiles (tile rows relative to the segment's first row)
            const int t0 = (int)((rb - seg_row0) / 256), t1 = (int)std::min<int64_t>(ts.n_tr, (re - seg_row0 + 255) / 256);
            return two_stage_tiles(c, ts, ts.row_first[(size_t)t0], ts.row_first[(size_t)t1] - ts.row_first[(size_t)t0], true);
        }
        return launch_exact(rb, re, as_dense);
    };
    auto packed_count = [&](size_t k, int64_t* n) -> int {
        unsigned long long got = 0;
        hipError_t e = hipMemcpyAsync(&got, c->d_counter, 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(MVS_E_HIP, "reading the cell count: %s", hipGetErrorString(e));
        if ((int64_t)got > (blocks[k].second - blocks[k].first) * s->n) return fail(MVS_E_HIP, "internal: more kept cells than cells");
        *n = (int64_t)got;
        return MVS_OK;
    };
    // Blocks of ONE shared matrix: block k's rows are final when launch k is and launch k + 1 never touches them (its
    // mirror images land in later blocks' rows), so block k is counted / scanned / filled / encoded on a SIDE stream while
    // launch k + 1 already runs on the context's stream -- memory-bound passes beside a matrix-core-bound kernel instead
    // of between two of them.  (stream_dense = 2: everything on the context's stream, one after the other.)
    side = dense && whole && blocks.size() > 1 &&
           (lopt.stream_dense >= 3 || (lopt.stream_dense == 1 && matrix_mode != kM2));
    // stream_dense = 4 (experiment): the row passes on the DOWNLOAD stream itself -- between two pieces' copies instead of beside
    // one (a device-to-host copy is a blit kernel that fills the card; a kernel that starts beside it ends with it)
    // stream_dense = 5 (experiment): ONE stream, but launch k + 1 is queued IN FRONT of block k's row passes (as with the side stream):
    // exact(k + 1), row passes(k), exact(k + 2), ... run one after the other at their un-contended speed and the host's read-backs of
    // block k fall into the shadow of a launch that is already queued -- the row passes beside an exact kernel wait for its
    // workgroups to leave the CUs one kernel after the other (a 49-us count took 659 us there)
    if (side) ps = lopt.stream_dense == 4 ? c->dl_stream : (lopt.stream_dense == 5 ? c->stream : c->post_stream);
    mark("setup", -1);
    if (!blocks.empty()) {
        rc = launch(0, dense);
        if (rc) return finish(rc);
    }
    mark("launched", 0);
    for (size_t k = 0; k < blocks.size(); ++k) {
        const int64_t rb = blocks[k].first, re = blocks[k].second;
        BlockCsr blk;
        bool next_launched = false;
        if (side) {
            hipError_t e = hipEventRecord(c->cmp_done, c->stream);              // launch k is the last thing queued there
            if (e == hipSuccess) e = hipStreamWaitEvent(ps, c->cmp_done, 0);
            if (e != hipSuccess) return finish(fail(MVS_E_HIP, "ordering the side stream: %s", hipGetErrorString(e)));
            if (c->timing && c->ev_valid[1]) add_kernel_ms();                    // launch k's time, before its events are reused
            if (k + 1 < blocks.size() && !out.failed()) {
                rc = launch(k + 1, true);
                if (rc) return finish(rc);
                next_launched = true;
                mark("launched", (long)k + 1);
            }
        }
        wait_for_set((int64_t)k);
        if (dense) {
            bool odd = false, held = false;
            if (ecb && lopt.stream_spec != 0 && spec_cells_per_row >= 0.0) {
                const double rows_k = (double)(re - rb);
                const int64_t cap_cells = (int64_t)(rows_k * spec_cells_per_row * 1.3) + 65536;
                const size_t cap_bytes = ((size_t)(rows_k * spec_bytes_per_row * 1.3) + ((size_t)1 << 20) + 7) & ~(size_t)7;
                rc = rows_from_dense_spec(c, rb, re, s->n, whole ? row_begin : rb, ld, (int64_t)k, blk, &odd, ps, active, cap_cells, cap_bytes,
                                          lopt.encode_stage_words, &held);
                if (rc) return finish(rc);
                if (!held) blk = BlockCsr();          // the sizes did not hold (or a 16-bit q): the careful way below
            }
            if (!held) rc = csr_from_dense(c, rb, re, s->n, whole ? row_begin : rb, ld, (int64_t)k, blk, &odd, ps, active, ecb != nullptr);
            if (rc) return finish(rc);
            mark("csr", (long)k);
            if (!side) add_kernel_ms();
            if (odd) {
                if (side) {                 // back to one stream; a launch already queued for block k + 1 is wasted, not wrong
                    (void)hipStreamSynchronize(c->stream);
                    side = false;
                    ps = c->stream;
                }
                // a kept cell whose q a byte cannot hold (norms that do not belong to the vectors): this block and the
                // rest go through the packed list, each block inside its own square -- the one case where a block is
                // compared a second time
                dense = false;
                matrix_mode = kNone;
                tiles_phase = false;
                lopt.filter_variant = saved_variant;
                int64_t br = std::max<int64_t>(256, budget_cells / std::max<int64_t>(s->n, 1) / 256 * 256);
                while (shift + bits_for(std::max<int64_t>(br - 1, 1)) > 64 && br > 256) br /= 2;
                std::vector<std::pair<int64_t, int64_t>> rest(blocks.begin(), blocks.begin() + (long)k);
                for (int64_t b0 = rb; b0 < row_end;) {
                    const int64_t b1 = std::min(row_end, (b0 / 256) * 256 + br);
                    rest.emplace_back(b0, b1);
                    b0 = b1;
                }
                blocks.swap(rest);
                rc = launch(k, false);
                if (rc) return finish(rc);
                --k;                                                   // take the block again, as a list this time
                continue;
            }
        } else {
            int64_t n = 0;
            rc = packed_count(k, &n);
            if (rc) return finish(rc);
            add_kernel_ms();
            rc = csr_from_packed(c, rb, re, n, shift, col_bits, (int64_t)k, blk);
            if (rc) return finish(rc);
        }
        total += blk.n;
        ++c->st_blocks;
        rc = prepare(blk);
        if (rc) return finish(rc);
        if (dense && ecb && blk.encoded && re > rb) {
            spec_cells_per_row = std::max(spec_cells_per_row, (double)blk.n / (double)(re - rb));
            spec_bytes_per_row = std::max(spec_bytes_per_row, (double)blk.enc_off[(size_t)(re - rb)] / (double)(re - rb));
        }
        mark("enc", (long)k);
        if (!next_launched && k + 1 < blocks.size() && !out.failed()) {   // the next block computes while this one is fed to the link
            rc = launch(k + 1, dense);
            if (rc) return finish(rc);
            next_launched = true;
            mark("launched", (long)k + 1);
        }
        rc = deliver(blk);
        if (rc) return finish(rc);
        mark("fed", (long)k);
        if (out.failed()) break;
        (void)next_launched;
    }
    return finish(MVS_OK);
}
}  // namespace mvs_capi

extern "C" {

int mvs_ctx_stream_stats(const mvs_ctx* c, double* kernel_ms, int64_t* bytes_out, int64_t* row_blocks, int64_t* pieces,
                         int* two_stage) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if (kernel_ms) *kernel_ms = c->st_kernel_ms;
    if (bytes_out) *bytes_out = c->st_bytes;
    if (row_blocks) *row_blocks = c->st_blocks;
    if (pieces) *pieces = c->st_pieces;
    if (two_stage) *two_stage = (int)c->st_two_stage;
    return MVS_OK;
}


}  // extern "C"

// -------------------------------------------------------------------------------------------------
// a shard's sorted cell list -> the same pieces (mvs_cells_stream / mvs_cells_stream_encoded)
// -------------------------------------------------------------------------------------------------
namespace mvs_capi {
int cells_stream_impl(mvs_ctx* c, const mvs_cell* cells, int64_t n, int64_t rb, int64_t re, mvs_row_block_cb cb, mvs_encoded_rows_cb ecb,
                      void* user, int64_t* n_delivered) {
    if (!c || (!cb && !ecb)) return fail(MVS_E_INVALID, "NULL argument");
    if (n_delivered) *n_delivered = 0;
    if (n < 0 || rb < 0 || re < rb || re >= (1LL << 31) || (n > 0 && !cells)) return fail(MVS_E_INVALID, "bad argument");
    if (re == rb) return MVS_OK;
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_download_side(c);
    if (rc) return rc;
    c->st_kernel_ms = 0.0;
    c->st_bytes = c->st_blocks = c->st_pieces = c->st_two_stage = 0;
    const int64_t rows = re - rb;
    const size_t piece_bytes = (size_t)c->opt.stream_piece_mib << 20;
    BlockCsr blk;
    blk.rb = rb;
    blk.re = re;
    blk.set = 0;
    blk.row_ptr.assign((size_t)rows + 1, 0);
    // the arrays of set 0 may still be on their way out of an earlier call's last block: that call drained its download stream
    // before it returned (finish / below), so nothing is in flight here
    if (n > 0) {
        rc = ensure_buf(c, &c->st_rowptr, &c->st_rowptr_bytes, (size_t)(rows + 1) * 8);
        if (rc == MVS_OK) rc = ensure_buf(c, &c->st_counts, &c->st_counts_bytes, (size_t)(rows + 2) * 8);   // index into the list + the "wide q" flag
        if (rc == MVS_OK) rc = claim_csr_set(c, 0, 0, n, false, c->stream);
        if (rc) return rc;
        unsigned int* d_wide = reinterpret_cast<unsigned int*>((long long*)c->st_counts + rows + 1);
        HIP_TRY(hipMemsetAsync(d_wide, 0, 8, c->stream));
        mvs::launch_cells_rowptr(c->stream, cells, n, rb, rows, (long long*)c->st_counts);
        rc = check_kernel("k_cells_rowptr");
        if (rc) return rc;
        mvs::launch_cells_split(c->stream, cells, (const long long*)c->st_counts, rows, n, (long long*)c->st_rowptr, (int32_t*)c->st_col[0],
                                c->st_q[0], 1, d_wide);
        rc = check_kernel("k_cells_split");
        if (rc) return rc;
        unsigned int h_wide = 0;
        rc = read_back(c, c->stream, {{blk.row_ptr.data(), c->st_rowptr, (size_t)(rows + 1) * 8}, {&h_wide, d_wide, 4}});
        if (rc) return rc;
        blk.n = blk.row_ptr[(size_t)rows];
        if (blk.n < 0 || blk.n > n) return fail(MVS_E_INVALID, "the cell list is not ordered by row");
        if (h_wide) {                      // some q needs 16 bits (norms that do not belong to the vectors): the q array again
            blk.wide = true;
            rc = ensure_buf(c, &c->st_q[0], &c->st_q_bytes[0], (size_t)n * 2);
            if (rc) return rc;
            mvs::launch_cells_split(c->stream, cells, (const long long*)c->st_counts, rows, n, nullptr, (int32_t*)c->st_col[0], c->st_q[0], 2,
                                    nullptr);
            rc = check_kernel("k_cells_split(16-bit q)");
            if (rc) return rc;
        }
    }
    HIP_TRY(hipEventRecord(c->dl_ready[0], c->stream));
    c->st_blocks = 1;
    StreamOut out;
    out.c = c;
    out.cb = cb;
    out.ecb = ecb;
    out.user = user;
    out.worker = std::thread([&out] { out.run(); });
    if (ecb) rc = encode_block(c, blk, c->stream);
    if (rc == MVS_OK) rc = ecb ? feed_encoded(c, out, blk, piece_bytes) : feed_block(c, out, blk, piece_bytes);
    out.close();
    (void)hipStreamSynchronize(c->dl_stream);
    if (rc) return rc;
    if (!out.error.empty()) return fail(MVS_E_HIP, "%s", out.error.c_str());
    if (out.cb_status != 0) return fail(MVS_E_ABORTED, "the row-block callback returned %d", out.cb_status);
    if (n_delivered) *n_delivered = blk.n;
    return MVS_OK;
}
}  // namespace mvs_capi

extern "C" {

int mvs_cells_stream(mvs_ctx* c, const mvs_cell* cells, int64_t n_cells, int64_t row_begin, int64_t row_end, mvs_row_block_cb cb,
                     void* user, int64_t* n_delivered) {
    try {
        return cells_stream_impl(c, cells, n_cells, row_begin, row_end, cb, nullptr, user, n_delivered);
    } catch (const std::bad_alloc&) {
        return fail(MVS_E_NOMEM, "out of host memory while streaming a shard's cells");
    } catch (const std::exception& e) {
        return fail(MVS_E_HIP, "mvs_cells_stream: %s", e.what());
    }
}

int mvs_cells_stream_encoded(mvs_ctx* c, const mvs_cell* cells, int64_t n_cells, int64_t row_begin, int64_t row_end,
                             mvs_encoded_rows_cb cb, void* user, int64_t* n_delivered) {
    try {
        return cells_stream_impl(c, cells, n_cells, row_begin, row_end, nullptr, cb, user, n_delivered);
    } catch (const std::bad_alloc&) {
        return fail(MVS_E_NOMEM, "out of host memory while streaming a shard's cells");
    } catch (const std::exception& e) {
        return fail(MVS_E_HIP, "mvs_cells_stream_encoded: %s", e.what());
    }
}

}  // extern "C"
