// exchange.hip -- nin_exchange_*: the multi-GPU all-gather of the path as direct peer-to-peer writes (VERDICT round 3,
// item 7), behind the C ABI and with no collective library in it.
//
// north_star's exchange is ONE all-gather with per-rank counts: every rank's block of CSR values (and Neumann rows) must reach every
// other rank.  RCCL's ring all-gather moves each byte over one xGMI link at a time; an MI355X has SEVEN links, one to each peer,
// point to point.  Here every rank owns a gathered buffer of `world` slots on its GPU, opens its peers' buffers through HIP IPC
// handles once, and a push is world - 1 device-to-device copies, one per peer, each on a stream of its own -- seven copies in
// flight on seven links -- straight from the kernel's output into slot `rank` of the peer's buffer: no padding to the longest
// shard travels, no staging copy.  What the library does NOT do is rendezvous: the 64-byte handles are exchanged by the caller
// (MPI, torch.distributed, a file -- whatever the host application already has), and a push has landed everywhere once every rank's
// nin_exchange_wait_sent has returned AND the ranks have passed a barrier of their own (8 bytes of control against gigabytes of
// payload).  Processes that share one GPU work too (the copies are then local): that is how the GPU test suite exercises it.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ninpol_amd.h"

namespace nin {
int abi_fail(int code, const char *fmt, ...);   // abi.hip: sets nin_last_error()
}

struct nin_exchange {
    int device = -1, rank = 0, world = 1;
    size_t slot_bytes = 0;
    char *buffer = nullptr;                 // [world][slot_bytes] on `device`: slot r = rank r's block
    std::vector<char *> remote;             // peers' buffers as mapped into this process (remote[rank] = buffer)
    std::vector<hipStream_t> streams;       // one per peer
    std::vector<hipEvent_t> done;           // the last push to peer p has been issued and completed
    hipEvent_t ready = nullptr;             // the caller's stream has produced the data
    bool connected = false;
};

#define X_TRY(expr)                                                                                         \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) return nin::abi_fail(NIN_EHIP, "%s: %s", #expr, hipGetErrorString(e_));      \
    } while (0)

extern "C" {

int nin_exchange_create(int device, int rank, int world, size_t slot_bytes, nin_exchange **out) {
    if (!out || world < 1 || rank < 0 || rank >= world || slot_bytes == 0) return nin::abi_fail(NIN_EINVAL, "bad argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return nin::abi_fail(NIN_ENODEVICE, "no HIP device %d", device);
    X_TRY(hipSetDevice(device));
    auto *x = new nin_exchange();
    x->device = device; x->rank = rank; x->world = world;
    x->slot_bytes = (slot_bytes + 255) & ~(size_t)255;
    if (hipMalloc((void **)&x->buffer, x->slot_bytes * world) != hipSuccess) {
        delete x;
        return nin::abi_fail(NIN_ENOMEM, "hipMalloc of the gathered buffer (%zu bytes)", slot_bytes * world);
    }
    x->remote.assign(world, nullptr);
    x->remote[rank] = x->buffer;
    x->streams.assign(world, nullptr);
    x->done.assign(world, nullptr);
    for (int p = 0; p < world; ++p) {
        if (hipStreamCreateWithFlags(&x->streams[p], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&x->done[p], hipEventDisableTiming) != hipSuccess) {
            nin_exchange_destroy(x);
            return nin::abi_fail(NIN_EHIP, "stream / event creation failed");
        }
    }
    if (hipEventCreateWithFlags(&x->ready, hipEventDisableTiming) != hipSuccess) {
        nin_exchange_destroy(x);
        return nin::abi_fail(NIN_EHIP, "event creation failed");
    }
    x->connected = world == 1;
    *out = x;
    return NIN_OK;
}

void nin_exchange_destroy(nin_exchange *x) {
    if (!x) return;
    if (x->device >= 0) (void)hipSetDevice(x->device);
    for (int p = 0; p < x->world; ++p) {
        if (p != x->rank && x->remote.size() > (size_t)p && x->remote[p]) (void)hipIpcCloseMemHandle(x->remote[p]);
        if (x->streams.size() > (size_t)p && x->streams[p]) { (void)hipStreamSynchronize(x->streams[p]); (void)hipStreamDestroy(x->streams[p]); }
        if (x->done.size() > (size_t)p && x->done[p]) (void)hipEventDestroy(x->done[p]);
    }
    if (x->ready) (void)hipEventDestroy(x->ready);
    if (x->buffer) (void)hipFree(x->buffer);
    delete x;
}

int nin_exchange_handle(nin_exchange *x, void *handle64) {
    if (!x || !handle64) return nin::abi_fail(NIN_EINVAL, "NULL argument");
    static_assert(sizeof(hipIpcMemHandle_t) == NIN_EXCHANGE_HANDLE_BYTES, "handle size");
    X_TRY(hipSetDevice(x->device));
    hipIpcMemHandle_t h;
    X_TRY(hipIpcGetMemHandle(&h, x->buffer));
    std::memcpy(handle64, &h, sizeof h);
    return NIN_OK;
}

int nin_exchange_connect(nin_exchange *x, const void *all_handles) {
    if (!x || (!all_handles && x->world > 1)) return nin::abi_fail(NIN_EINVAL, "NULL argument");
    if (x->connected) return NIN_OK;
    X_TRY(hipSetDevice(x->device));
    const char *hs = static_cast<const char *>(all_handles);
    for (int p = 0; p < x->world; ++p) {
        if (p == x->rank) continue;
        hipIpcMemHandle_t h;
        std::memcpy(&h, hs + (size_t)p * NIN_EXCHANGE_HANDLE_BYTES, sizeof h);
        void *ptr = nullptr;
        const hipError_t e = hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess);
        if (e != hipSuccess) return nin::abi_fail(NIN_EHIP, "hipIpcOpenMemHandle of rank %d's buffer: %s", p, hipGetErrorString(e));
        x->remote[p] = static_cast<char *>(ptr);
    }
    x->connected = true;
    return NIN_OK;
}

int nin_exchange_push(nin_exchange *x, const void *dev_src, size_t bytes, size_t offset, void *stream_) {
    if (!x || !dev_src) return nin::abi_fail(NIN_EINVAL, "NULL argument");
    if (!x->connected) return nin::abi_fail(NIN_ESTATE, "nin_exchange_connect has not been called");
    if (offset + bytes > x->slot_bytes) return nin::abi_fail(NIN_ERANGE, "%zu bytes at offset %zu do not fit the slot (%zu)", bytes, offset, x->slot_bytes);
    if (bytes == 0) return NIN_OK;
    X_TRY(hipSetDevice(x->device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    X_TRY(hipEventRecord(x->ready, stream));          // the copies wait for what `stream` has enqueued so far (the weight kernels)
    const size_t at = (size_t)x->rank * x->slot_bytes + offset;
    for (int k = 0; k < x->world; ++k) {
        const int p = (x->rank + k) % x->world;        // own slot first, then the peers, each rank starting with a different one
        X_TRY(hipStreamWaitEvent(x->streams[p], x->ready, 0));
        X_TRY(hipMemcpyAsync(x->remote[p] + at, dev_src, bytes, hipMemcpyDeviceToDevice, x->streams[p]));
        X_TRY(hipEventRecord(x->done[p], x->streams[p]));
    }
    return NIN_OK;
}

int nin_exchange_wait_sent(nin_exchange *x, void *stream_, int host_wait) {
    if (!x) return nin::abi_fail(NIN_EINVAL, "NULL argument");
    X_TRY(hipSetDevice(x->device));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    for (int p = 0; p < x->world; ++p) {
        if (host_wait) X_TRY(hipEventSynchronize(x->done[p]));
        else X_TRY(hipStreamWaitEvent(stream, x->done[p], 0));
    }
    return NIN_OK;
}

void *nin_exchange_buffer(nin_exchange *x) { return x ? x->buffer : nullptr; }
size_t nin_exchange_slot_bytes(const nin_exchange *x) { return x ? x->slot_bytes : 0; }

}  // extern "C"
