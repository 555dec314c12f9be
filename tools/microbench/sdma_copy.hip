// sdma_copy -- device-to-host copies of 32 MiB pieces three ways, alone and beside a stream of compute kernels:
//   (a) hipMemcpyAsync into pinned memory on a second stream (what the streamed output does: a blit KERNEL on this stack),
//   (b) hsa_amd_memory_async_copy (ROCr picks the engine), (c) hsa_amd_memory_async_copy_on_engine on an SDMA engine the
//   runtime reports as free for device -> host.
// Question: do (b) / (c) move the bytes without occupying the CUs -- i.e. do kernels that run beside them keep their duration?
//   hipcc --offload-arch=gfx950 -O2 -o sdma_copy sdma_copy.hip -lhsa-runtime64
#include <hip/hip_runtime.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                       \
    do {                                                                            \
        hipError_t e_ = (x);                                                        \
        if (e_ != hipSuccess) {                                                     \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
            exit(2);                                                                \
        }                                                                           \
    } while (0)
#define HK(x)                                                     \
    do {                                                          \
        hsa_status_t s_ = (x);                                    \
        if (s_ != HSA_STATUS_SUCCESS) {                           \
            const char* m_ = nullptr;                             \
            hsa_status_string(s_, &m_);                           \
            fprintf(stderr, "%s: %s\n", #x, m_ ? m_ : "?");       \
            exit(3);                                              \
        }                                                         \
    } while (0)

// a compute kernel that fills the card for a while: every lane spins on dependent FMAs
__global__ void k_busy(float* out, int iters) {
    float a = threadIdx.x * 0.001f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) a = a * b + 0.5f;
    if (a == 12345.678f) out[0] = a;
}
// a SHORT kernel (what a row pass is): a few microseconds of work
__global__ void k_short(float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = out[i] * 1.5f + 1.0f;
}

struct Agents {
    hsa_agent_t gpu{}, cpu{};
    bool have_gpu = false, have_cpu = false;
};
static hsa_status_t on_agent(hsa_agent_t a, void* data) {
    Agents* ag = static_cast<Agents*>(data);
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !ag->have_gpu) {
        ag->gpu = a;
        ag->have_gpu = true;
    }
    if (t == HSA_DEVICE_TYPE_CPU && !ag->have_cpu) {
        ag->cpu = a;
        ag->have_cpu = true;
    }
    return HSA_STATUS_SUCCESS;
}

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    const size_t piece = 32u << 20;
    const int pieces = 32;
    CK(hipSetDevice(0));
    char* dev = nullptr;
    CK(hipMalloc((void**)&dev, piece * 4));
    CK(hipMemset(dev, 1, piece * 4));
    char* host[2];
    for (auto& h : host) CK(hipHostMalloc((void**)&h, piece, hipHostMallocDefault));
    float* scratch = nullptr;
    CK(hipMalloc((void**)&scratch, 1 << 24));
    hipStream_t cs, ds;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ds, hipStreamNonBlocking));
    HK(hsa_init());
    Agents ag;
    HK(hsa_iterate_agents(on_agent, &ag));
    if (!ag.have_gpu || !ag.have_cpu) {
        fprintf(stderr, "agents not found\n");
        return 3;
    }
    uint32_t engines = 0, preferred = 0;
    hsa_status_t es = hsa_amd_memory_copy_engine_status(ag.cpu, ag.gpu, &engines);
    printf("copy engines free for device -> host: mask 0x%x (status %d)\n", engines, (int)es);
    es = hsa_amd_memory_get_preferred_copy_engine(ag.cpu, ag.gpu, &preferred);
    printf("preferred engine mask 0x%x (status %d)\n", preferred, (int)es);
    hsa_signal_t sig[2];
    for (auto& s : sig) HK(hsa_signal_create(1, 0, nullptr, &s));

    // durations of a compute kernel series and of short kernels, with `copier` running beside them (or nothing: mode 0)
    auto run = [&](int mode, const char* name) {
        CK(hipDeviceSynchronize());
        hipEvent_t e0, e1, s0, s1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        CK(hipEventCreate(&s0));
        CK(hipEventCreate(&s1));
        const double t0 = now_ms();
        // compute: 24 busy kernels of ~1 ms each, a short kernel after each (timed as a group)
        CK(hipEventRecord(e0, cs));
        for (int k = 0; k < 24; ++k) {
            hipLaunchKernelGGL(k_busy, dim3(256 * 8), dim3(256), 0, cs, scratch, 60000);
            hipLaunchKernelGGL(k_short, dim3(4096), dim3(256), 0, cs, scratch, 1 << 20);
        }
        CK(hipEventRecord(e1, cs));
        // copies beside them
        double copy_ms = 0.0;
        const double c0 = now_ms();
        if (mode == 1) {
            for (int p = 0; p < pieces; ++p) CK(hipMemcpyAsync(host[p & 1], dev + (size_t)(p & 3) * piece, piece, hipMemcpyDeviceToHost, ds));
            CK(hipStreamSynchronize(ds));
        } else if (mode == 2 || mode == 3) {
            for (int p = 0; p < pieces; ++p) {
                hsa_signal_t& s = sig[p & 1];
                if (p >= 2) hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
                hsa_signal_store_relaxed(s, 1);
                if (mode == 2) HK(hsa_amd_memory_async_copy(host[p & 1], ag.cpu, dev + (size_t)(p & 3) * piece, ag.gpu, piece, 0, nullptr, s));
                else {
                    uint32_t eng = preferred ? preferred : engines;
                    eng = eng & (~eng + 1);                    // lowest set bit
                    HK(hsa_amd_memory_async_copy_on_engine(host[p & 1], ag.cpu, dev + (size_t)(p & 3) * piece, ag.gpu, piece, 0, nullptr, s,
                                                           (hsa_amd_sdma_engine_id_t)eng, false));
                }
            }
            for (auto& s : sig) hsa_signal_wait_scacquire(s, HSA_SIGNAL_CONDITION_LT, 1, UINT64_MAX, HSA_WAIT_STATE_BLOCKED);
        }
        copy_ms = now_ms() - c0;
        CK(hipStreamSynchronize(cs));
        const double wall = now_ms() - t0;
        float kms = 0;
        CK(hipEventElapsedTime(&kms, e0, e1));
        printf("%-46s compute series %.2f ms, copies %.2f ms (%.1f GB/s), wall %.2f ms\n", name, kms, copy_ms,
               mode ? (double)piece * pieces / copy_ms / 1e6 : 0.0, wall);
        (void)s0;
        (void)s1;
    };
    for (int rep = 0; rep < 2; ++rep) {
        run(0, "no copies");
        run(1, "hipMemcpyAsync D2H (pinned), second stream");
        run(2, "hsa_amd_memory_async_copy");
        if (engines || preferred) run(3, "hsa_amd_memory_async_copy_on_engine (SDMA)");
    }
    return 0;
}
