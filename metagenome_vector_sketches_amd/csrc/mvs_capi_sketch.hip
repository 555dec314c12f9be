// mvs_capi_sketch.hip -- C ABI: projection, sketch statistics, limb planes, sketch sets (include/mvs_hip.h)
#include "mvs_capi_internal.h"

using namespace mvs_capi;

extern "C" {

// -------------------------------------------------------------------------------------------------
// projection
// -------------------------------------------------------------------------------------------------
int mvs_project_csr(mvs_ctx* c, const uint64_t* hashes, int mem_hashes, const int64_t* offsets,
                    int64_t n_samples, int d, int32_t* out, int mem_out) {
    return mvs_project_csr_stats(c, hashes, mem_hashes, offsets, n_samples, d, out, mem_out, nullptr, nullptr);
}

int mvs_project_csr_stats(mvs_ctx* c, const uint64_t* hashes, int mem_hashes, const int64_t* offsets,
                          int64_t n_samples, int d, int32_t* out, int mem_out, int64_t* sumsq, int64_t* max_abs) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    const Range range(c, "mvs_project_csr");
    if ((sumsq == nullptr) != (max_abs == nullptr)) return fail(MVS_E_INVALID, "sumsq and max_abs go together");
    // sumsq lives where the sketches live: device array for device sketches, host array for host sketches
    int64_t* const sumsq_user = sumsq;
    DevBuf dsum;
    if (max_abs) *max_abs = 0;
    if (n_samples < 0 || d <= 0) return fail(MVS_E_INVALID, "n_samples=%lld d=%d", (long long)n_samples, d);
    if (!mem_ok(mem_hashes) || !mem_ok(mem_out)) return fail(MVS_E_INVALID, "bad mem flag");
    if (n_samples == 0) return MVS_OK;
    if (!offsets || !out) return fail(MVS_E_INVALID, "offsets/out is NULL");
    HIP_TRY(hipSetDevice(c->device));

    // units: runs of <= kProjUnitMax hashes, written straight into pinned memory
    if (n_samples >= (1LL << 31)) return fail(MVS_E_RANGE, "too many samples");
    size_t n_units = 0;
    bool all_single = true;
    for (int64_t s = 0; s < n_samples; ++s) {
        const int64_t b = offsets[s], e = offsets[s + 1];
        if (e < b) return fail(MVS_E_INVALID, "offsets not monotone at sample %lld", (long long)s);
        if (e - b >= (1LL << 31)) return fail(MVS_E_RANGE, "sample %lld has >= 2^31 hashes", (long long)s);
        // an empty sample gets one unit of zero hashes: the kernel then stores its row of zeros itself
        n_units += e == b ? 1 : (size_t)((e - b + mvs::kProjUnitMax - 1) / mvs::kProjUnitMax);
        all_single = all_single && (e - b) <= mvs::kProjUnitMax;
    }
    int rc = acquire_pinned(c, std::max<size_t>(n_units * sizeof(mvs::ProjUnit), 256));
    if (rc) return rc;
    mvs::ProjUnit* units = (mvs::ProjUnit*)c->pinned;
    {
        size_t w = 0;
        for (int64_t s = 0; s < n_samples; ++s) {
            const int64_t b = offsets[s], e = offsets[s + 1];
            const bool single = (e - b) <= mvs::kProjUnitMax;
            if (e == b) units[w++] = mvs::ProjUnit{b, 0, (int32_t)s, 1, 0};
            for (int64_t p = b; p < e; p += mvs::kProjUnitMax) {
                mvs::ProjUnit u;
                u.begin = p;
                u.count = (int32_t)std::min<int64_t>(mvs::kProjUnitMax, e - p);
                u.sample = (int32_t)s;
                u.single = single ? 1 : 0;
                u.pad = 0;
                units[w++] = u;
            }
        }
    }
    const int64_t total = offsets[n_samples];
    if (total > 0 && !hashes) return fail(MVS_E_INVALID, "hashes is NULL");

    DevBuf dh, dout;
    const uint64_t* d_hashes = hashes;
    // Host hash lists larger than one staging piece go up through the two-buffer pipeline: while piece k is on the
    // link, piece k+1 is being copied into pinned memory and the samples that piece k-1 completed are being projected.
    const bool pipelined = mem_hashes == MVS_MEM_HOST && (size_t)total * 8 > kUploadPiece;
    if (mem_hashes == MVS_MEM_HOST) {
        HIP_TRY(dh.alloc((size_t)total * 8));
        if (!pipelined) HIP_TRY(hipMemcpyAsync(dh.p, hashes, (size_t)total * 8, hipMemcpyHostToDevice, c->stream));
        d_hashes = (const uint64_t*)dh.p;
    }
    int32_t* d_out = out;
    const size_t out_bytes = (size_t)n_samples * (size_t)d * 4;
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(dout.alloc(out_bytes));
        d_out = (int32_t*)dout.p;
        if (sumsq) {
            HIP_TRY(dsum.alloc((size_t)n_samples * 8));
            sumsq = (int64_t*)dsum.p;
        }
    }
    const size_t ubytes = n_units * sizeof(mvs::ProjUnit);
    rc = ensure_scratch(c, std::max<size_t>(ubytes, 256));
    if (rc) return rc;
    if (n_units) {
        HIP_TRY(hipMemcpyAsync(c->scratch, units, ubytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->pinned_ev, c->stream));
        c->pinned_busy = true;
    }

    // samples cut into several units are combined with atomics and start from zero; single units store
    if (!all_single) HIP_TRY(hipMemsetAsync(d_out, 0, out_bytes, c->stream));
    const bool fused = sumsq != nullptr && all_single;         // statistics inside the projection kernel
    if (fused) {
        HIP_TRY(hipMemsetAsync(sumsq, 0, (size_t)n_samples * 8, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    }
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    const int nblk = (d + 63) / 64;
    // kernel variant (launch_project): four blocks per wave sharing the first splitmix64 round where the dimension
    // fills them (8.97 vs 9.44 ms on 10k x 50k hashes, d = 2048), else two or one block per wave; option
    // project_variant forces one
    int bpw = (nblk % 4 == 0 && nblk >= 8) ? 14 : (nblk >= 2 ? 2 : 1);
    if (c->opt.project_variant == 14 && nblk >= 4) bpw = 14;
    if (c->opt.project_variant == 12 && nblk >= 2) bpw = 12;
    if (c->opt.project_variant == 2 && nblk >= 2) bpw = 2;
    if (c->opt.project_variant == 1) bpw = 1;
    if (!pipelined) {
        mvs::launch_project(c->stream, d_hashes, (const mvs::ProjUnit*)c->scratch, (int64_t)n_units, d, d_out, bpw,
                            fused ? (unsigned long long*)sumsq : nullptr, fused ? c->d_counter : nullptr);
        rc = check_kernel("k_project");
        if (rc) return rc;
    } else {
        rc = ensure_upload_pipeline(c);
        if (rc) return rc;
        // the unit list is in hash order: units [u_done, u_next) are those whose hashes the pieces sent so far cover
        size_t u_done = 0;
        const size_t total_bytes = (size_t)total * 8;
        int piece = 0;
        for (size_t off = 0; off < total_bytes; off += kUploadPiece, ++piece) {
            const int b = piece & 1;
            const size_t len = std::min(kUploadPiece, total_bytes - off);
            if (piece >= 2) HIP_TRY(hipEventSynchronize(c->up_done[b]));      // staging buffer b is free again
            parallel_copy(c->up_pinned[b], (const char*)hashes + off, len);
            HIP_TRY(hipMemcpyAsync((char*)dh.p + off, c->up_pinned[b], len, hipMemcpyHostToDevice, c->up_stream));
            HIP_TRY(hipEventRecord(c->up_done[b], c->up_stream));
            const int64_t covered = (int64_t)((off + len) / 8);
            size_t u_next = u_done;
            while (u_next < n_units && units[u_next].begin + units[u_next].count <= covered) ++u_next;
            if (u_next > u_done) {
                HIP_TRY(hipStreamWaitEvent(c->stream, c->up_done[b], 0));
                mvs::launch_project(c->stream, d_hashes, (const mvs::ProjUnit*)c->scratch + u_done, (int64_t)(u_next - u_done), d,
                                    d_out, bpw, fused ? (unsigned long long*)sumsq : nullptr, fused ? c->d_counter : nullptr);
                rc = check_kernel("k_project");
                if (rc) return rc;
                u_done = u_next;
            }
        }
        if (u_done != n_units) return fail(MVS_E_INVALID, "internal: %zu of %zu projection units launched", u_done, n_units);
    }
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        c->ev_valid[0] = true;
    }
    if (sumsq) {
        if (fused) {
            unsigned long long m = 0;
            {
                const int rb_rc = read_back(c, c->stream, {{&m, c->d_counter, 8}});
                if (rb_rc) return rb_rc;
            }
            *max_abs = (int64_t)m;
        } else {   // some sample spans several units: its entries are final only now
            rc = mvs_sketch_stats(c, d_out, MVS_MEM_DEVICE, n_samples, d, sumsq, MVS_MEM_DEVICE, max_abs);
            if (rc) return rc;
        }
    }
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
        if (sumsq_user) HIP_TRY(hipMemcpyAsync(sumsq_user, sumsq, (size_t)n_samples * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } else if (mem_hashes == MVS_MEM_HOST) {
        HIP_TRY(hipStreamSynchronize(c->stream));   // staging buffer is freed on return
    } else {
        // the unit list lives in ctx scratch, which stays valid; nothing to wait for
    }
    return MVS_OK;
}

int mvs_sketch_sumsq(mvs_ctx* c, const int32_t* sketches, int mem_in, int64_t n, int d, int64_t* sumsq,
                     int mem_out) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if (n < 0 || d <= 0 || !mem_ok(mem_in) || !mem_ok(mem_out)) return fail(MVS_E_INVALID, "bad argument");
    if (n == 0) return MVS_OK;
    if (!sketches || !sumsq) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf din, dout;
    const int32_t* d_in = sketches;
    if (mem_in == MVS_MEM_HOST) {
        HIP_TRY(din.alloc((size_t)n * d * 4));
        HIP_TRY(hipMemcpyAsync(din.p, sketches, (size_t)n * d * 4, hipMemcpyHostToDevice, c->stream));
        d_in = (const int32_t*)din.p;
    }
    int64_t* d_out = sumsq;
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(dout.alloc((size_t)n * 8));
        d_out = (int64_t*)dout.p;
    }
    mvs::launch_sumsq(c->stream, d_in, n, d, d_out);
    int rc = check_kernel("k_sumsq");
    if (rc) return rc;
    if (mem_out == MVS_MEM_HOST)
        HIP_TRY(hipMemcpyAsync(sumsq, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    if (mem_out == MVS_MEM_HOST || mem_in == MVS_MEM_HOST) HIP_TRY(hipStreamSynchronize(c->stream));
    return MVS_OK;
}

}  // extern "C"

namespace mvs_capi {

// "%g" keeps 6 significant digits: x -> the decimal r * 10^-j (r an integer of 6 digits, round-half-even on the exact
// binary value of x as printf does) -> the double nearest to that decimal (what strtod returns) -> squared.
// The product x * 10^j is rounded once; only when it lands exactly on k + 0.5 can the true product lie on either side,
// and the fma residual says which (rounding is monotonic, so a product off the tie is on the true side of it).  An
// exponent estimate that is off by one next to a power of ten yields the same decimal (r = 10^6 is renormalised).
__device__ double norm_sq_from_text(long long sumsq, int d) {
    if (sumsq <= 0) return 0.0;
    const double x = sqrt((double)sumsq / (double)d);
    constexpr double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    int e = 0;                                   // 10^e <= x < 10^(e+1), up to the off-by-one noted above
    if (x >= 1.0) {
        while (e < 21 && x >= p10[e + 1]) ++e;
    } else {
        double y = x;
        while (e > -16 && y < 1.0) {
            y *= 10.0;
            --e;
        }
    }
    int j = 5 - e;                               // x * 10^j has 6 digits before the point
    double m, err;
    if (j >= 0) {
        m = x * p10[j];
        err = fma(x, p10[j], -m);                // exact: true product = m + err
    } else {
        m = x / p10[-j];
        err = -fma(m, p10[-j], -x);              // sign of (true quotient - m)
    }
    double r = rint(m);                          // half-even
    const double fl = floor(m);
    if (m - fl == 0.5 && err != 0.0) r = err > 0.0 ? fl + 1.0 : fl;
    if (r >= 1e6) {
        r = 1e5;
        --j;
    }
    const double v = j >= 0 ? r / p10[j] : r * p10[-j];
    return v * v;
}

__global__ __launch_bounds__(256) void k_norms_sq_text(const int64_t* __restrict__ sumsq, int64_t n, int d,
                                                       double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = norm_sq_from_text(sumsq[i], d);
}

}  // namespace mvs_capi

extern "C" {

int mvs_norms_sq_text(mvs_ctx* c, const int64_t* sumsq, int64_t n, int d, double* out) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if (n < 0 || d <= 0) return fail(MVS_E_INVALID, "bad argument");
    if (n == 0) return MVS_OK;
    if (!sumsq || !out) return fail(MVS_E_INVALID, "NULL buffer");
    if ((n + 255) / 256 > 0x7fffffffLL) return fail(MVS_E_INVALID, "too many entries");
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_norms_sq_text, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, sumsq, n, d, out);
    return check_kernel("k_norms_sq_text");
}

int mvs_sketch_stats(mvs_ctx* c, const int32_t* sketches, int mem_in, int64_t n, int d, int64_t* sumsq, int mem_out,
                     int64_t* max_abs) {
    if (!c || !max_abs) return fail(MVS_E_INVALID, "NULL argument");
    *max_abs = 0;
    if (n < 0 || d <= 0 || !mem_ok(mem_in) || !mem_ok(mem_out)) return fail(MVS_E_INVALID, "bad argument");
    if (n == 0) return MVS_OK;
    if (!sketches || !sumsq) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf din, dout;
    const int32_t* d_in = sketches;
    if (mem_in == MVS_MEM_HOST) {
        HIP_TRY(din.alloc((size_t)n * d * 4));
        HIP_TRY(hipMemcpyAsync(din.p, sketches, (size_t)n * d * 4, hipMemcpyHostToDevice, c->stream));
        d_in = (const int32_t*)din.p;
    }
    int64_t* d_out = sumsq;
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(dout.alloc((size_t)n * 8));
        d_out = (int64_t*)dout.p;
    }
    HIP_TRY(hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    mvs::launch_stats(c->stream, d_in, n, d, d_out, c->d_counter);
    int rc = check_kernel("k_stats");
    if (rc) return rc;
    if (mem_out == MVS_MEM_HOST)
        HIP_TRY(hipMemcpyAsync(sumsq, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    unsigned long long m = 0;
    {
        const int rb_rc = read_back(c, c->stream, {{&m, c->d_counter, 8}});
        if (rb_rc) return rb_rc;
    }
    *max_abs = (int64_t)m;
    return MVS_OK;
}

int mvs_sketch_saturate_i16(mvs_ctx* c, const int32_t* sketches, int mem_in, int64_t n_elems, int16_t* out,
                            int mem_out) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if (n_elems < 0 || !mem_ok(mem_in) || !mem_ok(mem_out)) return fail(MVS_E_INVALID, "bad argument");
    if (n_elems == 0) return MVS_OK;
    if (!sketches || !out) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf din, dout;
    const int32_t* d_in = sketches;
    if (mem_in == MVS_MEM_HOST) {
        HIP_TRY(din.alloc((size_t)n_elems * 4));
        HIP_TRY(hipMemcpyAsync(din.p, sketches, (size_t)n_elems * 4, hipMemcpyHostToDevice, c->stream));
        d_in = (const int32_t*)din.p;
    }
    int16_t* d_out = out;
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(dout.alloc((size_t)n_elems * 2));
        d_out = (int16_t*)dout.p;
    }
    mvs::launch_saturate_i16(c->stream, d_in, n_elems, d_out);
    int rc = check_kernel("k_saturate_i16");
    if (rc) return rc;
    if (mem_out == MVS_MEM_HOST)
        HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)n_elems * 2, hipMemcpyDeviceToHost, c->stream));
    if (mem_out == MVS_MEM_HOST || mem_in == MVS_MEM_HOST) HIP_TRY(hipStreamSynchronize(c->stream));
    return MVS_OK;
}

// -------------------------------------------------------------------------------------------------
// pairwise
// -------------------------------------------------------------------------------------------------
int mvs_sketch_max_abs(mvs_ctx* c, const void* sketches, int elem_bytes, int mem, int64_t n_elems,
                       int64_t* max_abs) {
    if (!c || !max_abs) return fail(MVS_E_INVALID, "NULL argument");
    if ((elem_bytes != 4 && elem_bytes != 2) || !mem_ok(mem) || n_elems < 0)
        return fail(MVS_E_INVALID, "bad argument");
    *max_abs = 0;
    if (n_elems == 0) return MVS_OK;
    if (!sketches) return fail(MVS_E_INVALID, "sketches is NULL");
    HIP_TRY(hipSetDevice(c->device));
    const void* d_in = sketches;
    if (mem == MVS_MEM_HOST) {
        int rc0 = ensure_buf(c, &c->stage, &c->stage_bytes, (size_t)n_elems * elem_bytes);
        if (rc0) return rc0;
        HIP_TRY(hipMemcpyAsync(c->stage, sketches, (size_t)n_elems * elem_bytes, hipMemcpyHostToDevice, c->stream));
        d_in = c->stage;
    }
    HIP_TRY(hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    mvs::launch_max_abs(c->stream, d_in, elem_bytes, n_elems, c->d_counter);
    int rc = check_kernel("k_max_abs");
    if (rc) return rc;
    unsigned long long m = 0;
    {
        const int rb_rc = read_back(c, c->stream, {{&m, c->d_counter, 8}});
        if (rb_rc) return rb_rc;
    }
    *max_abs = (int64_t)m;
    return MVS_OK;
}

int mvs_limbs_for_max_abs(int64_t max_abs) {
    if (max_abs < 0) max_abs = -max_abs;
    if (max_abs <= 127) return 1;
    if (max_abs <= 32639) return 2;      // 127 * (1 + 256)
    if (max_abs <= 8355711) return 3;    // 127 * (1 + 256 + 65536)
    return 4;                            // exact mod 2^32 for every int32
}

int mvs_limb_geometry(int64_t n, int d, int limbs, int64_t* n_alloc, int* d_pad, size_t* bytes) {
    if (n < 0 || d <= 0 || !mvs::limb_code_ok(limbs)) return fail(MVS_E_INVALID, "bad argument");
    // tiles are up to 256 rows: pad to a multiple of 256 plus one spare tile
    const int64_t na = (n + 255) / 256 * 256 + 256;
    const int dp = (d + mvs::kBK - 1) / mvs::kBK * mvs::kBK;
    if (n_alloc) *n_alloc = na;
    if (d_pad) *d_pad = dp;
    if (bytes) *bytes = (size_t)na * (size_t)mvs::planes_of(limbs) * (size_t)dp;
    return MVS_OK;
}

int mvs_limb_split(mvs_ctx* c, const void* sketches, int elem_bytes, int mem, int64_t n_rows, int d, int limbs,
                   int8_t* planes, int d_pad, int64_t row_offset) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if ((elem_bytes != 4 && elem_bytes != 2) || !mem_ok(mem) || n_rows < 0 || d <= 0 || !mvs::limb_code_ok(limbs) ||
        d_pad < d || d_pad % mvs::kBK != 0 || row_offset < 0)
        return fail(MVS_E_INVALID, "bad argument");
    if (n_rows == 0) return MVS_OK;
    if (!sketches || !planes) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    const void* d_in = sketches;
    if (mem == MVS_MEM_HOST) {   // grow-only staging buffer of the context (no allocation per chunk)
        int rc0 = ensure_buf(c, &c->stage, &c->stage_bytes, (size_t)n_rows * d * elem_bytes);
        if (rc0) return rc0;
        HIP_TRY(hipMemcpyAsync(c->stage, sketches, (size_t)n_rows * d * elem_bytes, hipMemcpyHostToDevice, c->stream));
        d_in = c->stage;
    }
    mvs::launch_limb_split(c->stream, d_in, elem_bytes, n_rows, d, limbs, planes, d_pad, row_offset);
    int rc = check_kernel("k_limb_split");
    if (rc) return rc;
    if (mem == MVS_MEM_HOST) HIP_TRY(hipStreamSynchronize(c->stream));
    return MVS_OK;
}

int mvs_sketch_set_create(mvs_ctx* c, const void* sketches, int elem_bytes, int mem, int64_t n, int d,
                          mvs_sketch_set** out) {
    if (!c || !out) return fail(MVS_E_INVALID, "NULL argument");
    *out = nullptr;
    if ((elem_bytes != 4 && elem_bytes != 2) || !mem_ok(mem) || n < 0 || d <= 0)
        return fail(MVS_E_INVALID, "bad argument");
    if (n >= (1LL << 31) - 256) return fail(MVS_E_RANGE, "n too large for int32 row/col indices");
    if (n > 0 && !sketches) return fail(MVS_E_INVALID, "sketches is NULL");
    HIP_TRY(hipSetDevice(c->device));
    // stage once if the input is on the host
    DevBuf din;
    const void* d_in = sketches;
    if (mem == MVS_MEM_HOST && n > 0) {
        HIP_TRY(din.alloc((size_t)n * d * elem_bytes));
        HIP_TRY(hipMemcpyAsync(din.p, sketches, (size_t)n * d * elem_bytes, hipMemcpyHostToDevice, c->stream));
        d_in = din.p;
    }
    int64_t max_abs = 0;
    int rc = mvs_sketch_max_abs(c, d_in, elem_bytes, MVS_MEM_DEVICE, n * d, &max_abs);
    if (rc) return rc;
    int limbs = mvs_limbs_for_max_abs(max_abs);
    // The 3-pass Karatsuba scheme (63 * (1 + 128): digits in [-64,63], their sum in int8) is exact and tested but
    // measures 9-19 % SLOWER than two base-256 limbs on MI355X (25 % fewer MFMAs, 1.5x the LDS traffic), so it is
    // opt-in: option enable_k3.
    if (c->opt.enable_k3 && max_abs > 127 && max_abs <= 8127) limbs = MVS_LIMBS_K3;
    int64_t n_alloc = 0;
    int d_pad = 0;
    size_t bytes = 0;
    mvs_limb_geometry(n, d, limbs, &n_alloc, &d_pad, &bytes);
    mvs_sketch_set* s = new (std::nothrow) mvs_sketch_set();
    if (!s) return fail(MVS_E_NOMEM, "out of host memory");
    if (hipMalloc((void**)&s->owned, bytes) != hipSuccess) {
        delete s;
        return fail(MVS_E_NOMEM, "hipMalloc of %zu bytes of limb planes failed", bytes);
    }
    s->ctx = c;
    s->planes = s->owned;
    s->n = n;
    s->n_alloc = n_alloc;
    s->d = d;
    s->d_pad = d_pad;
    s->limbs = limbs;
    s->id = ++g_set_ids;
    hipError_t e = hipMemsetAsync(s->owned, 0, bytes, c->stream);
    if (e == hipSuccess) {
        rc = mvs_limb_split(c, d_in, elem_bytes, MVS_MEM_DEVICE, n, d, limbs, s->owned, d_pad, 0);
        if (rc == MVS_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(MVS_E_HIP, "sync failed");
    } else {
        rc = fail(MVS_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    if (rc) {
        mvs_sketch_set_destroy(s);
        return rc;
    }
    *out = s;
    return MVS_OK;
}

int mvs_sketch_set_from_planes(mvs_ctx* c, const int8_t* planes, int64_t n, int64_t n_alloc, int d, int d_pad,
                               int limbs, mvs_sketch_set** out) {
    if (!c || !out || !planes) return fail(MVS_E_INVALID, "NULL argument");
    *out = nullptr;
    int64_t need_alloc = 0;
    int need_pad = 0;
    if (mvs_limb_geometry(n, d, limbs, &need_alloc, &need_pad, nullptr)) return MVS_E_INVALID;
    if (n_alloc < need_alloc || d_pad != need_pad)
        return fail(MVS_E_INVALID, "plane buffer geometry: need n_alloc >= %lld and d_pad == %d",
                    (long long)need_alloc, need_pad);
    if (n >= (1LL << 31) - 256) return fail(MVS_E_RANGE, "n too large for int32 row/col indices");
    mvs_sketch_set* s = new (std::nothrow) mvs_sketch_set();
    if (!s) return fail(MVS_E_NOMEM, "out of host memory");
    s->ctx = c;
    s->planes = planes;
    s->n = n;
    s->n_alloc = n_alloc;
    s->d = d;
    s->d_pad = d_pad;
    s->limbs = limbs;
    s->id = ++g_set_ids;
    *out = s;
    return MVS_OK;
}

int mvs_sketch_set_alloc(mvs_ctx* c, int64_t n, int d, int limbs, mvs_sketch_set** out) {
    if (!c || !out) return fail(MVS_E_INVALID, "NULL argument");
    *out = nullptr;
    if (n < 0 || d <= 0 || !mvs::limb_code_ok(limbs)) return fail(MVS_E_INVALID, "bad argument");
    if (n >= (1LL << 31) - 256) return fail(MVS_E_RANGE, "n too large for int32 row/col indices");
    HIP_TRY(hipSetDevice(c->device));
    int64_t n_alloc = 0;
    int d_pad = 0;
    size_t bytes = 0;
    mvs_limb_geometry(n, d, limbs, &n_alloc, &d_pad, &bytes);
    mvs_sketch_set* s = new (std::nothrow) mvs_sketch_set();
    if (!s) return fail(MVS_E_NOMEM, "out of host memory");
    if (hipMalloc((void**)&s->owned, bytes) != hipSuccess) {
        delete s;
        return fail(MVS_E_NOMEM, "hipMalloc of %zu bytes of limb planes failed", bytes);
    }
    s->ctx = c;
    s->planes = s->owned;
    s->n = n;
    s->n_alloc = n_alloc;
    s->d = d;
    s->d_pad = d_pad;
    s->limbs = limbs;
    s->id = ++g_set_ids;
    if (hipMemsetAsync(s->owned, 0, bytes, c->stream) != hipSuccess) {
        mvs_sketch_set_destroy(s);
        return fail(MVS_E_HIP, "hipMemsetAsync failed");
    }
    *out = s;
    return MVS_OK;
}

}  // extern "C"

namespace mvs_capi {
// Rows [lo, hi) of an owned set are about to be rewritten.  If the context holds data derived from the set's present
// contents (coarse plane, fragment-major copies) and the range is a small part of it, the set keeps its generation and
// remembers the range: refresh_derived() re-derives just those rows before the next comparison.  A search front end
// that appends its queries behind a resident database (search.py: SearchIndex) thus keeps the database's coarse plane --
// bumping the generation made every search rebuild it (6 ms per 10^6 sketches) or fall back to the exact kernels.
void note_rows_rewritten(mvs_sketch_set* s, int64_t lo, int64_t hi) {
    mvs_ctx* c = s->ctx;
    // (the "a block of few rows went to the exact kernel for lack of a coarse plane" marker counts as well: it is what makes
    // the SECOND such block build the plane, and it must survive the upload of that block's rows)
    const bool cached = (c->coarse_id == s->id && c->coarse_gen == s->gen) || (c->planes_fm_id == s->id && c->planes_fm_gen == s->gen) ||
                        (c->few_rows_id == s->id && c->few_rows_gen == s->gen);
    const int64_t u_lo = s->dirty_hi > s->dirty_lo ? std::min(s->dirty_lo, lo) : lo;
    const int64_t u_hi = s->dirty_hi > s->dirty_lo ? std::max(s->dirty_hi, hi) : hi;
    if (cached && (u_hi - u_lo) * 8 <= s->n) {
        s->dirty_lo = u_lo;
        s->dirty_hi = u_hi;
        return;
    }
    ++s->gen;   // derived data of the old contents is stale as a whole
    s->dirty_lo = s->dirty_hi = 0;
}
}  // namespace mvs_capi

extern "C" {

int mvs_sketch_set_fill(mvs_sketch_set* s, const void* sketches, int elem_bytes, int mem, int64_t row_offset,
                        int64_t n_rows) {
    if (!s || !s->owned) return fail(MVS_E_INVALID, "set is NULL or not owned by the library");
    if (row_offset < 0 || n_rows < 0 || row_offset + n_rows > s->n)
        return fail(MVS_E_INVALID, "rows [%lld,%lld) outside the set", (long long)row_offset,
                    (long long)(row_offset + n_rows));
    if (n_rows > 0) note_rows_rewritten(s, row_offset, row_offset + n_rows);
    return mvs_limb_split(s->ctx, sketches, elem_bytes, mem, n_rows, s->d, s->limbs, s->owned, s->d_pad, row_offset);
}

int mvs_sketch_set_fill_stats(mvs_sketch_set* s, const void* sketches, int elem_bytes, int mem, int64_t row_offset,
                              int64_t n_rows, int64_t* max_abs) {
    if (!s || !s->owned) return fail(MVS_E_INVALID, "set is NULL or not owned by the library");
    if (!max_abs) return fail(MVS_E_INVALID, "max_abs is NULL");
    *max_abs = 0;
    if ((elem_bytes != 4 && elem_bytes != 2) || !mem_ok(mem)) return fail(MVS_E_INVALID, "bad argument");
    if (row_offset < 0 || n_rows < 0 || row_offset + n_rows > s->n)
        return fail(MVS_E_INVALID, "rows [%lld,%lld) outside the set", (long long)row_offset,
                    (long long)(row_offset + n_rows));
    if (n_rows == 0) return MVS_OK;
    if (!sketches) return fail(MVS_E_INVALID, "sketches is NULL");
    mvs_ctx* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)n_rows * s->d * elem_bytes;
    const void* d_in = sketches;
    if (mem == MVS_MEM_HOST) {   // one upload serves both kernels
        int rc0 = ensure_buf(c, &c->stage, &c->stage_bytes, bytes);
        if (rc0) return rc0;
        HIP_TRY(hipMemcpyAsync(c->stage, sketches, bytes, hipMemcpyHostToDevice, c->stream));
        d_in = c->stage;
    }
    note_rows_rewritten(s, row_offset, row_offset + n_rows);
    HIP_TRY(hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    mvs::launch_max_abs(c->stream, d_in, elem_bytes, n_rows * s->d, c->d_counter);
    int rc = check_kernel("k_max_abs");
    if (rc) return rc;
    mvs::launch_limb_split(c->stream, d_in, elem_bytes, n_rows, s->d, s->limbs, s->owned, s->d_pad, row_offset);
    rc = check_kernel("k_limb_split");
    if (rc) return rc;
    unsigned long long m = 0;
    {
        const int rb_rc = read_back(c, c->stream, {{&m, c->d_counter, 8}});
        if (rb_rc) return rb_rc;
    }
    *max_abs = (int64_t)m;
    return MVS_OK;
}

int mvs_sketch_set_info(const mvs_sketch_set* s, int64_t* n, int* d, int* limbs, int64_t* n_alloc, int* d_pad) {
    if (!s) return fail(MVS_E_INVALID, "set is NULL");
    if (n) *n = s->n;
    if (d) *d = s->d;
    if (limbs) *limbs = s->limbs;
    if (n_alloc) *n_alloc = s->n_alloc;
    if (d_pad) *d_pad = s->d_pad;
    return MVS_OK;
}

int mvs_sketch_set_planes(mvs_sketch_set* s, int8_t** planes) {
    if (!s || !planes) return fail(MVS_E_INVALID, "NULL argument");
    if (!s->owned) return fail(MVS_E_INVALID, "the set is a view of a caller-owned buffer");
    *planes = s->owned;
    return MVS_OK;
}

int mvs_sketch_set_touch(mvs_sketch_set* s) {
    if (!s) return fail(MVS_E_INVALID, "set is NULL");
    ++s->gen;
    s->dirty_lo = s->dirty_hi = 0;
    return MVS_OK;
}

int mvs_sketch_set_destroy(mvs_sketch_set* s) {
    if (!s) return MVS_OK;
    if (s->owned) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
        (void)hipFree(s->owned);
    }
    delete s;
    return MVS_OK;
}


}  // extern "C"
