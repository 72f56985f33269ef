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
// No waiting on workgroups of the SAME rank anywhere (counters are only added to), so residency is not assumed.  Every wait on ANOTHER
// rank is a bounded poll (s_sleep between system-scope loads): when a bound expires the launch sets header.status, stops waiting and
// ends; the result is then undefined and dvae_comm_status() reports it -- a missing peer is an error code, never a hang.
// Buffer reuse across calls: a rank leaves phase C only after every peer has finished reading its send[] (their `done` adds come
// after their phase-B reads), and a peer starts pushing call k + 1 into recv[] only after this rank's phase A of call k + 1.
#include <stdlib.h>
#include "common.hpp"
#include "../../include/dvae_train.h"

namespace dvae {
namespace comm {

constexpr int MAXW = 16;
constexpr int GRID = 64;          // workgroups per launch (also the unit of the ready / done counters)
constexpr int HDR = 256;          // header bytes

struct Header { unsigned long long ready, done; unsigned status, pad; };

struct Peers { char* p[MAXW]; };

__device__ __forceinline__ unsigned long long ld_sys(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// bounded wait: true when *p reached `want`
__device__ __forceinline__ bool wait_ge(const unsigned long long* p, unsigned long long want, long long max_polls) {
    for (long long i = 0; i < max_polls; ++i) {
        if (ld_sys(p) >= want) return true;
        __builtin_amdgcn_s_sleep(16);
    }
    return false;
}

__global__ __launch_bounds__(256) void allreduce_kernel(Peers peers, int rank, int world, int64_t n, int64_t n_pad, int64_t shard,
                                                        const float* slabs, int n_slabs, int64_t slab_stride,
                                                        float* out /* may alias slab 0: element i is read and written by the same thread */, unsigned long long call, long long max_polls) {
    char* const me = peers.p[rank];
    Header* const hdr = reinterpret_cast<Header*>(me);
    // all exchange traffic moves as 8-byte granules (two floats) through system-scope accesses: n_pad is a multiple of 64
    typedef unsigned long long u64;
    u64* const send = reinterpret_cast<u64*>(me + HDR);
    u64* const recv = send + n_pad / 2;
    const int64_t n2 = n_pad / 2;
    const int tid = threadIdx.x;
    __shared__ int ok_s;
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
            ok = wait_ge(&reinterpret_cast<const Header*>(peers.p[p])->ready, call * (unsigned long long)gridDim.x, max_polls) ? 1 : 0;
        if (!ok) __hip_atomic_store(&hdr->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
        if (ok) ok = wait_ge(&hdr->done, call * (unsigned long long)gridDim.x * (unsigned long long)world, max_polls) ? 1 : 0;
        if (!ok) __hip_atomic_store(&hdr->status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok_s = ok;
    }
    __syncthreads();
    __threadfence_system();
    for (int64_t j = (int64_t)blockIdx.x * 256 + tid; j < n2; j += (int64_t)gridDim.x * 256) {
        const u64 t = __hip_atomic_load(recv + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
    long long max_polls;
};

static int64_t pad_n(int64_t n, int world, int64_t* shard) {
    int64_t s = (n + world - 1) / world;
    s = (s + 63) / 64 * 64;
    if (shard) *shard = s;
    return s * world;
}

extern "C" int dvae_comm_create(int rank, int world, int64_t n_floats, dvae_comm_t** out, unsigned char handle[DVAE_IPC_HANDLE_BYTES]) {
    DVAE_CHECK_ARG(out && handle && world >= 1 && world <= MAXW && rank >= 0 && rank < world && n_floats > 0, "comm_create: bad argument");
    static_assert(sizeof(hipIpcMemHandle_t) <= DVAE_IPC_HANDLE_BYTES, "IPC handle size");
    dvae_comm* c = (dvae_comm*)calloc(1, sizeof(dvae_comm));
    DVAE_CHECK_ARG(c != nullptr, "comm_create: out of host memory");
    c->rank = rank; c->world = world; c->n = n_floats;
    c->n_pad = pad_n(n_floats, world, &c->shard);
    const size_t bytes = (size_t)HDR + 2 * (size_t)c->n_pad * sizeof(float);
    // fine-grained (coherent across devices inside a running kernel); plain hipMalloc memory is only coherent at kernel boundaries
    hipError_t e = hipExtMallocWithFlags(&c->local, bytes, hipDeviceMallocFinegrained);
    if (e != hipSuccess) { (void)hipGetLastError(); e = hipMalloc(&c->local, bytes); }
    if (e != hipSuccess) { free(c); set_error("comm_create: allocation of %zu B failed: %s", bytes, hipGetErrorString(e)); return (int)e; }
    e = hipMemset(c->local, 0, bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    hipIpcMemHandle_t h;
    if (e == hipSuccess) e = hipIpcGetMemHandle(&h, c->local);
    if (e != hipSuccess) { (void)hipFree(c->local); free(c); set_error("comm_create: %s", hipGetErrorString(e)); return (int)e; }
    memset(handle, 0, DVAE_IPC_HANDLE_BYTES);
    memcpy(handle, &h, sizeof(h));
    c->peer[rank] = c->local;
    c->call = 0;
    const char* mp = getenv("DVAE_COMM_MAX_POLLS");
    c->max_polls = mp ? atoll(mp) : (1ll << 19);          // 0.5 M polls of ~1000 clocks + one uncached load each: of the order of a second
    *out = c;
    return 0;
}

extern "C" int dvae_comm_connect(dvae_comm_t* c, const unsigned char* handles /* world x DVAE_IPC_HANDLE_BYTES */) {
    DVAE_CHECK_ARG(c && handles, "comm_connect: bad argument");
    for (int p = 0; p < c->world; ++p) {
        if (p == c->rank) continue;
        hipIpcMemHandle_t h;
        memcpy(&h, handles + (size_t)p * DVAE_IPC_HANDLE_BYTES, sizeof(h));
        void* ptr = nullptr;
        DVAE_HIP(hipIpcOpenMemHandle(&ptr, h, hipIpcMemLazyEnablePeerAccess));
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
    c->call += 1;
    hipLaunchKernelGGL(allreduce_kernel, dim3(GRID), dim3(256), 0, (hipStream_t)stream, pr, c->rank, c->world, c->n, c->n_pad, c->shard,
                       slabs, n_slabs, slab_stride, out, c->call, c->max_polls);
    DVAE_LAUNCH_OK("allreduce_kernel");
    return 0;
}

extern "C" int dvae_comm_status(dvae_comm_t* c, int* failed) {
    DVAE_CHECK_ARG(c && failed, "comm_status: bad argument");
    Header h;
    DVAE_HIP(hipMemcpy(&h, c->local, sizeof(h), hipMemcpyDeviceToHost));     // synchronises with the device
    *failed = h.status != 0 ? 1 : 0;
    if (h.status) set_error("allreduce_flat: a bounded wait for a peer expired (rank %d of %d, call %llu)", c->rank, c->world, c->call);
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
