// Direct gradient exchange of the data-parallel train step: ONE launch per rank on the step's stream, between the
// weight-gradient kernel and the apply kernel.  The reference has no call site for this (scripts/training_M2.py:31-33 is single-device).
//
// Every rank owns one exchange buffer (fine-grained device memory, exported through a hipIpc handle and mapped by all its peers):
//     header { ready, done, status }   send[n_pad]   recv[n_pad]
// and a launch does, for its call number k (a host-side counter, the same on every rank):
//   A  local: sum of the rank's gradient slabs (fixed order) -> own send[]; every workgroup adds 1 to own `ready` (system scope, after a
//      system-scope release fence) -- the slab sum of the multi-GPU path, fused "on the way in";
//   B  reduce-scatter by pull: rank r owns shard r.  It waits until every rank's `ready` has reached k * G (G workgroups per launch),
//      reads shard r of every rank's send[] in rank order -- one deterministic sum, computed exactly once -- and pushes the result
//      into recv[] of EVERY rank (all-gather by push), then adds 1 to every rank's `done`;
//   C  waits until own `done` has reached k * G * world, copies recv[] to the caller's `out` (the flat gradient the apply kernel reads).
// On an 8-GPU node phase B reads 7 shards and writes 7 shards per rank, all over different xGMI links at once: 2 * 7/8 * 1.2 MB per rank
// and two flag hops, against RCCL's generic ring / tree for a 1.2 MB message.
//
// Waits: phase B waits for `ready` == k * G of EVERY rank (its own included) and phase C for own `done` == k * G * world, so a launch
// completes only once all G workgroups of every rank have run phase A and phase B.  No workgroup waits for a counter before it has added
// to it itself, so there is no wait cycle inside a launch -- but phase B of every workgroup DOES wait for its own rank's `ready` == k * G,
// i.e. for all G = 64 workgroups of the launch to have STARTED: they must all become resident within the bound (64 of 256 CUs; a launch
// that shares the chip with other streams or processes and gets fewer than 64 workgroup slots for longer than the bound spins it out and
// times out like a missing peer -- status non-zero on every rank, NaN result, sticky for the life of the communicator).  Every wait is bounded by WALL TIME (s_memrealtime, 100 MHz; dvae_comm_set_timeout_ms, default 20 s, env
// DVAE_COMM_TIMEOUT_MS): when the bound expires the launch stores a non-zero status into the header of EVERY rank, still adds to the
// `done` counters (nobody waits a second bound for it) and ends; every rank that sees a non-zero status -- its own or a peer's -- fills
// `out` with NaN, so the failure is in-band (the optimizer step that follows turns the parameters and every later loss into NaN) and
// dvae_comm_status() reports it -- a missing or late peer is an error, never a hang and never a silently stale gradient.
// Buffer reuse across calls: a rank leaves phase C only after every peer has finished reading its send[] (their `done` adds come
// after their phase-B reads), and a peer starts pushing call k + 1 into recv[] only after this rank's phase A of call k + 1.
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <unistd.h>
#include "common.hpp"
#include "../../include/dvae_train.h"

namespace dvae {
namespace comm {

constexpr int MAXW = 16;
constexpr int GRID = 64;          // workgroups per launch (also the unit of the ready / done counters)
constexpr int HDR = 256;          // header bytes

struct Header { unsigned long long ready, done; unsigned status, pad; };
constexpr unsigned long long TICKS_PER_MS = 100000ull;      // s_memrealtime: 100 MHz

struct Peers { char* p[MAXW]; };

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bounded wait: true when *p reached `want` before the wall clock passed `deadline` (at least one poll is always made)
__device__ __forceinline__ bool wait_ge(const unsigned long long* p, unsigned long long want, unsigned long long deadline) {
    for (;;) {
        if (ld_sys(p) >= want) return true;
        if (__builtin_amdgcn_s_memrealtime() >= deadline) return false;
        __builtin_amdgcn_s_sleep(16);
    }
}

__global__ __launch_bounds__(256) void allreduce_kernel(Peers peers, int rank, int world, int64_t n, int64_t n_pad, int64_t shard,
                                                        const float* slabs, int n_slabs, int64_t slab_stride,
                                                        float* out /* may alias slab 0: element i is read and written by the same thread */, unsigned long long call,
                                                        unsigned long long timeout_ticks) {
    char* const me = peers.p[rank];
    Header* const hdr = reinterpret_cast<Header*>(me);
    // all exchange traffic moves as 8-byte granules (two floats) through system-scope accesses: n_pad is a multiple of 64
    typedef unsigned long long u64;
    u64* const send = reinterpret_cast<u64*>(me + HDR);
    u64* const recv = send + n_pad / 2;
    const int64_t n2 = n_pad / 2;
    const int tid = threadIdx.x;
    const unsigned long long deadline = __builtin_amdgcn_s_memrealtime() + timeout_ticks;      // per workgroup, from its own start
    __shared__ int ok_s;
    auto fail_everywhere = [&]() {                                        // the failure is every rank's: status into every header
        for (int p = 0; p < world; ++p)
            __hip_atomic_store(&reinterpret_cast<Header*>(peers.p[p])->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    auto pack = [](float a, float b) { return (u64)__float_as_uint(a) | ((u64)__float_as_uint(b) << 32); };
    auto slab_sum = [&](int64_t i) {
        if (i >= n) return 0.f;
        float t = slabs[i];
        for (int k = 1; k < n_slabs; ++k) t += slabs[(int64_t)k * slab_stride + i];      // slab order: deterministic
        return t;
    };
    // ---- A: slab sum -> send
    for (int64_t j = (int64_t)blockIdx.x * 256 + tid; j < n2; j += (int64_t)gridDim.x * 256)
        __hip_atomic_store(send + j, pack(slab_sum(2 * j), slab_sum(2 * j + 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __threadfence_system();
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(&hdr->ready, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- B: wait for every rank's send[], reduce own shard, push it to every rank's recv[]
    if (tid == 0) {
        int ok = 1;
        for (int p = 0; p < world && ok; ++p)
            ok = wait_ge(&reinterpret_cast<const Header*>(peers.p[p])->ready, call * (unsigned long long)gridDim.x, deadline) ? 1 : 0;
        if (!ok) fail_everywhere();
        ok_s = ok;
    }
    __syncthreads();
    __threadfence_system();                                   // acquire side of the flag hand-off
    const int64_t s0 = (int64_t)rank * (shard / 2);
    int64_t s1 = s0 + shard / 2; if (s1 > n2) s1 = n2;
    if (ok_s) {
        for (int64_t j = s0 + (int64_t)blockIdx.x * 256 + tid; j < s1; j += (int64_t)gridDim.x * 256) {
            u64 v[MAXW];
            for (int p = 0; p < world; ++p)                  // all loads first (independent), then the sum in rank order
                v[p] = __hip_atomic_load(reinterpret_cast<const u64*>(peers.p[p] + HDR) + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            float a = 0.f, b = 0.f;
            for (int p = 0; p < world; ++p) { a += __uint_as_float((unsigned)v[p]); b += __uint_as_float((unsigned)(v[p] >> 32)); }
            const u64 t = pack(a, b);
            for (int p = 0; p < world; ++p)
                __hip_atomic_store(reinterpret_cast<u64*>(peers.p[p] + HDR) + n2 + j, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    __threadfence_system();
    __syncthreads();
    if (tid == 0)
        for (int p = 0; p < world; ++p)
            __hip_atomic_fetch_add(&reinterpret_cast<Header*>(peers.p[p])->done, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    // ---- C: every rank has pushed its shard into own recv[] -> out
    if (tid == 0) {
        int ok = ok_s;
        if (ok) ok = wait_ge(&hdr->done, call * (unsigned long long)gridDim.x * (unsigned long long)world, deadline) ? 1 : 0;
        if (!ok) fail_everywhere();
        // a peer that gave up in phase B has stored its status here BEFORE its `done` adds (release), so it is visible now: its shard of
        // recv[] is stale.  (The status word is sticky: once set, every later call of this communicator ends in NaN as well.)
        if (__hip_atomic_load(&hdr->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) ok = 0;
        ok_s = ok;
    }
    __syncthreads();
    __threadfence_system();
    const bool good = ok_s != 0;
    for (int64_t j = (int64_t)blockIdx.x * 256 + tid; j < n2; j += (int64_t)gridDim.x * 256) {
        const u64 t = good ? __hip_atomic_load(recv + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0x7fc000007fc00000ull;
        if (2 * j < n) out[2 * j] = __uint_as_float((unsigned)t);
        if (2 * j + 1 < n) out[2 * j + 1] = __uint_as_float((unsigned)(t >> 32));
    }
}

}  // namespace comm
}  // namespace dvae

using namespace dvae;
using namespace dvae::comm;

struct dvae_comm {
    int rank, world;
    int64_t n, n_pad, shard;
    void* local;                 // own exchange buffer (allocated here)
    void* peer[MAXW];            // mapped peer buffers (peer[rank] == local)
    bool opened[MAXW];
    unsigned long long call;
    unsigned long long timeout_ticks;
    int device;
};

// What travels in the DVAE_IPC_HANDLE_BYTES of a rank: the hipIpc handle of its exchange buffer, then who and where it is -- the owning
// process (several ranks hosted by ONE process connect through plain pointers: a process cannot open its own IPC handle) and the
// device's PCI address (dvae_comm_connect checks peer access between the two devices before the first launch needs it).
struct HandleBlob {
    hipIpcMemHandle_t ipc;
    long long pid;
    void* local;                  // valid inside process `pid` only
    int pci_domain, pci_bus, pci_device, pad;
    unsigned long long nonce;     // drawn once per process: two ranks in different PID namespaces (one container per rank) may share a pid
};

// this process's identity beyond its pid (ranks in separate PID namespaces can both be pid 1): 64 random bits drawn at first use
static unsigned long long process_nonce() {
    static unsigned long long v = [] {
        unsigned long long r = 0;
        FILE* f = fopen("/dev/urandom", "rb");
        if (f) { if (fread(&r, sizeof(r), 1, f) != 1) r = 0; fclose(f); }
        if (r == 0) r = ((unsigned long long)getpid() << 32) ^ (unsigned long long)(uintptr_t)&r ^ 0x9e3779b97f4a7c15ull;
        return r;
    }();
    return v;
}
static_assert(sizeof(HandleBlob) <= DVAE_IPC_HANDLE_BYTES, "handle blob size");

static int64_t pad_n(int64_t n, int world, int64_t* shard) {
    int64_t s = (n + world - 1) / world;
    s = (s + 63) / 64 * 64;
    if (shard) *shard = s;
    return s * world;
}

static unsigned long long default_timeout_ticks() {
    const char* ms = getenv("DVAE_COMM_TIMEOUT_MS");
    long long v = ms ? atoll(ms) : 20000;            // generous: a peer may be checkpointing or validating; a dead peer still ends the launch
    if (v < 0) v = 0;
    return (unsigned long long)v * TICKS_PER_MS;
}

extern "C" int dvae_comm_create(int rank, int world, int64_t n_floats, dvae_comm_t** out, unsigned char handle[DVAE_IPC_HANDLE_BYTES]) {
    DVAE_CHECK_ARG(out && handle && world >= 1 && world <= MAXW && rank >= 0 && rank < world && n_floats > 0, "comm_create: bad argument");
    dvae_comm* c = (dvae_comm*)calloc(1, sizeof(dvae_comm));
    DVAE_CHECK_ARG(c != nullptr, "comm_create: out of host memory");
    c->rank = rank; c->world = world; c->n = n_floats;
    c->n_pad = pad_n(n_floats, world, &c->shard);
    const size_t bytes = (size_t)HDR + 2 * (size_t)c->n_pad * sizeof(float);
    // Fine-grained memory only (coherent across devices INSIDE a running kernel).  Plain hipMalloc memory is coherent at kernel boundaries
    // only: the in-kernel flag and shard reads of a peer could then see stale data and still pass the bounded waits, so there is no
    // fallback -- a platform that cannot give fine-grained device memory gets an error here and keeps the process-group exchange.
    hipError_t e = hipGetDevice(&c->device);
    if (e == hipSuccess) e = hipExtMallocWithFlags(&c->local, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) {
        (void)hipGetLastError(); free(c);
        set_error("comm_create: no fine-grained device memory for the exchange buffer (%zu B): %s -- the direct exchange is unavailable, use the process group", bytes, hipGetErrorString(e));
        return (int)e;
    }
    e = hipMemset(c->local, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    HandleBlob hb;
    memset(&hb, 0, sizeof(hb));
    if (e == hipSuccess) e = hipIpcGetMemHandle(&hb.ipc, c->local);
    hipDeviceProp_t prop;
    if (e == hipSuccess) e = hipGetDeviceProperties(&prop, c->device);
    if (e != hipSuccess) { (void)hipFree(c->local); free(c); set_error("comm_create: %s", hipGetErrorString(e)); return (int)e; }
    hb.pid = (long long)getpid(); hb.local = c->local; hb.nonce = process_nonce();
    hb.pci_domain = prop.pciDomainID; hb.pci_bus = prop.pciBusID; hb.pci_device = prop.pciDeviceID;
    memset(handle, 0, DVAE_IPC_HANDLE_BYTES);
    memcpy(handle, &hb, sizeof(hb));
    c->peer[rank] = c->local;
    c->call = 0;
    c->timeout_ticks = default_timeout_ticks();
    *out = c;
    return 0;
}

extern "C" int dvae_comm_set_timeout_ms(dvae_comm_t* c, int64_t ms) {
    DVAE_CHECK_ARG(c && ms >= 0, "comm_set_timeout_ms: bad argument");
    c->timeout_ticks = (unsigned long long)ms * TICKS_PER_MS;
    return 0;
}

extern "C" int dvae_comm_connect(dvae_comm_t* c, const unsigned char* handles /* world x DVAE_IPC_HANDLE_BYTES */) {
    DVAE_CHECK_ARG(c && handles, "comm_connect: bad argument");
    int ndev = 0;
    DVAE_HIP(hipGetDeviceCount(&ndev));
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        HandleBlob hb;
        memcpy(&hb, handles + (size_t)p * DVAE_IPC_HANDLE_BYTES, sizeof(hb));
        // the peer's device, if this process can see it: peer access must be possible before a kernel dereferences the mapping
        int pdev = -1;
        for (int d = 0; d < ndev && pdev < 0; ++d) {
            hipDeviceProp_t pr;
            if (hipGetDeviceProperties(&pr, d) == hipSuccess && pr.pciDomainID == hb.pci_domain && pr.pciBusID == hb.pci_bus && pr.pciDeviceID == hb.pci_device) pdev = d;
        }
        if (pdev >= 0 && pdev != c->device) {
            int can = 0;
            DVAE_HIP(hipDeviceCanAccessPeer(&can, c->device, pdev));
            DVAE_CHECK_ARG(can != 0, "comm_connect: device %d (rank %d) cannot access device %d (rank %d): no peer path between them", c->device, c->rank, pdev, p);
        }
        if (hb.pid == (long long)getpid() && hb.nonce == process_nonce()) {      // a rank hosted by this same process (pid AND nonce): its pointer is valid here as it is
            if (pdev >= 0 && pdev != c->device) {
                hipError_t e = hipDeviceEnablePeerAccess(pdev, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { set_error("comm_connect: hipDeviceEnablePeerAccess(%d): %s", pdev, hipGetErrorString(e)); return (int)e; }
                (void)hipGetLastError();
            }
            c->peer[p] = hb.local; c->opened[p] = false;
            continue;
        }
        void* ptr = nullptr;
        DVAE_HIP(hipIpcOpenMemHandle(&ptr, hb.ipc, hipIpcMemLazyEnablePeerAccess));
        c->peer[p] = ptr; c->opened[p] = true;
    }
    return 0;
}

extern "C" int dvae_allreduce_flat(dvae_comm_t* c, const float* slabs, int n_slabs, int64_t slab_stride, float* out, void* stream) {
    DVAE_CHECK_ARG(c && slabs && out && n_slabs >= 1, "allreduce_flat: bad argument");
    for (int p = 0; p < c->world; ++p) DVAE_CHECK_ARG(c->peer[p] != nullptr, "allreduce_flat: peer %d not connected (dvae_comm_connect)", p);
    Peers pr;
    memset(&pr, 0, sizeof(pr));
    for (int p = 0; p < c->world; ++p) pr.p[p] = (char*)c->peer[p];
    hipLaunchKernelGGL(allreduce_kernel, dim3(GRID), dim3(256), 0, (hipStream_t)stream, pr, c->rank, c->world, c->n, c->n_pad, c->shard,
                       slabs, n_slabs, slab_stride, out, c->call + 1, c->timeout_ticks);
    DVAE_LAUNCH_OK("allreduce_kernel");
    c->call += 1;                                      // only a launch that was accepted counts: a failed one leaves the counters in step
    return 0;
}

extern "C" int dvae_comm_status(dvae_comm_t* c, int* failed) {
    DVAE_CHECK_ARG(c && failed, "comm_status: bad argument");
    Header h;
    DVAE_HIP(hipMemcpy(&h, c->local, sizeof(h), hipMemcpyDeviceToHost));     // synchronises with the device
    *failed = h.status != 0 ? 1 : 0;
    if (h.status) set_error("allreduce_flat: a bounded wait for a peer expired on some rank (seen by rank %d of %d, call %llu): the reduced gradient was filled with NaN", c->rank, c->world, c->call);
    return 0;
}

extern "C" int dvae_comm_destroy(dvae_comm_t* c) {
    if (!c) return 0;
    (void)hipDeviceSynchronize();
    for (int p = 0; p < c->world; ++p)
        if (c->opened[p] && c->peer[p]) (void)hipIpcCloseMemHandle(c->peer[p]);
    if (c->local) (void)hipFree(c->local);
    free(c);
    return 0;
}
