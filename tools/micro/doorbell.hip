// Closed-loop floor at the benchmark's shape (1024 workgroups x 64 lanes = one wave per SIMD), with a trivial device policy
// between two steps -- the question VERDICT r2 #2 asks: does a PERSISTENT step kernel that keeps the environments' state in
// registers and is driven by doorbells beat one launch per step?
//
//   launches : [policy kernel][step-like kernel: loads 11 state groups + action row, FMA chain, stores 11 groups + obs row]  x K
//   doorbell : one persistent kernel (state groups loaded once); per step it polls `go` (hipStreamWriteValue64 behind the
//              policy kernel), loads the action row, runs the same chain, stores the obs row write-through, drains, arrives on a
//              two-level counter; the last arriver stores `done = k`, which the policy stream waits for (hipStreamWaitValue64).
//              A slice of the chain (`pre`, the Philox block's share) runs BEFORE the poll, as the real kernel could.
//   round trip only: the same with chain = 0 and no payload -- what the signalling alone costs.
//
// Every wait in the kernel is bounded (s_memrealtime, 20 ms): on a timeout the wave stores done = ~0 so that the stream drains,
// records the step in status[0] and leaves; the host runs under `timeout`.
// hipcc --offload-arch=gfx950 -O3 -o doorbell doorbell.hip && timeout -k 5 60 ./doorbell
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int NB = 1024, G = 11, OBS = 26, ACT = 6;

__device__ __forceinline__ void wt16(float4* base, size_t bytes, unsigned voff, unsigned soff, float4 x) {
    auto r = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
    u32x4 d = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, 16);   // sc1: write-through
}

__global__ __launch_bounds__(64) void k_policy(const float* __restrict__ obs, float* __restrict__ act) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    const float2 o = *reinterpret_cast<const float2*>(obs + (size_t)i * OBS);
    float2* a = reinterpret_cast<float2*>(act + (size_t)i * ACT);
    a[0] = make_float2(o.x * 0.5f, o.y * 0.5f); a[1] = make_float2(o.x, -o.y); a[2] = make_float2(0.1f, 0.2f);
}

__device__ __forceinline__ float chain(float acc, int n) {
    for (int c = 0; c < n; ++c) acc = acc * 1.0000001f + 1e-9f;
    return acc;
}

// one launch per step: state through memory
__global__ __launch_bounds__(64) void k_step(float4* __restrict__ arena, const float* __restrict__ act, float* __restrict__ obs, int n_chain) {
    float4* A = arena + (size_t)blockIdx.x * (G * 64) + threadIdx.x;
    float4 v[G];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = A[g * 64];
    const int i = blockIdx.x * 64 + threadIdx.x;
    const float2* ap = reinterpret_cast<const float2*>(act + (size_t)i * ACT);
    const float2 a0 = ap[0], a1 = ap[1], a2 = ap[2];
    float acc = chain(v[0].x + a0.x + a1.y + a2.x, n_chain);
    v[0].x = acc;
#pragma unroll
    for (int g = 0; g < G; ++g) wt16(arena + (size_t)blockIdx.x * (G * 64), G * 1024, threadIdx.x * 16, g * 1024, v[g]);
    float* o = obs + (size_t)i * OBS;
#pragma unroll
    for (int k = 0; k < OBS; k += 2) *reinterpret_cast<float2*>(o + k) = make_float2(acc + k, acc - k);
}

__device__ __forceinline__ u64 now() { return __builtin_amdgcn_s_memrealtime(); }    // 100 MHz, one clock for the whole chip

// persistent: state in registers, one doorbell per step
__global__ __launch_bounds__(64) void k_persist(float4* __restrict__ arena, const float* __restrict__ act, float* __restrict__ obs,
                                                u64* go, u64* done, unsigned* cnt /* [16 * 32] shards, + top at 16 * 32 */, int* status,
                                                int K, int n_chain, int n_pre, int payload, u64 timeout_ticks, u64* stamps /* [K + 1][4] */) {
    float4* A = arena + (size_t)blockIdx.x * (G * 64) + threadIdx.x;
    float4 v[G];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = A[g * 64];
    const int i = blockIdx.x * 64 + threadIdx.x;
    float acc = v[0].x;
    for (int k = 1; k <= K; ++k) {
        acc = chain(acc, n_pre);                                        // what needs no action (the Philox block's share)
        const u64 t0 = now();
        bool dead = false;
        while (__hip_atomic_load(go, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (u64)k) {
            __builtin_amdgcn_s_sleep(2);
            if (now() - t0 > timeout_ticks) { dead = true; break; }
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) stamps[4 * k + 1] = now();        // go seen
        if (dead) {
            if (threadIdx.x == 0) { status[0] = k; __hip_atomic_store(done, ~0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            break;
        }
        if (payload) {
            // the action row the policy kernel has just written (its launch boundary released it): device-scope loads, never L1
            const float* ap = act + (size_t)i * ACT;
            float a[ACT];
#pragma unroll
            for (int j = 0; j < ACT; ++j) a[j] = __hip_atomic_load(ap + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            acc = chain(acc + a[0] + a[3] + a[4], n_chain - n_pre);
            float4* ob = reinterpret_cast<float4*>(obs + (size_t)blockIdx.x * 64 * OBS);
            // 64 rows x 26 floats = 416 float4 per block: lanes write 6.5 of them each (write-through)
#pragma unroll
            for (int r = 0; r < 6; ++r) wt16(ob, 64 * OBS * 4, threadIdx.x * 16, r * 1024, make_float4(acc, acc + r, acc - r, (float)k));
            if (threadIdx.x < 32) wt16(ob, 64 * OBS * 4, threadIdx.x * 16, 6144, make_float4(acc, acc, acc, (float)k));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every store of this wave has been acknowledged
        if (threadIdx.x == 0) {
            const unsigned shard = blockIdx.x & 15u;
            const unsigned old = __hip_atomic_fetch_add(cnt + shard * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((old + 1u) % (NB / 16) == 0u) {
                const unsigned old2 = __hip_atomic_fetch_add(cnt + 16 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((old2 + 1u) % 16u == 0u) { stamps[4 * k + 2] = now(); __hip_atomic_store(done, (u64)k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
            }
        }
    }
    v[0].x = acc;
#pragma unroll
    for (int g = 0; g < G; ++g) wt16(arena + (size_t)blockIdx.x * (G * 64), G * 1024, threadIdx.x * 16, g * 1024, v[g]);
}

// doorbells rung / awaited by KERNELS of the policy stream instead of stream memory operations
__global__ void k_ring(u64* go, u64 k, u64* stamps) { stamps[4 * k + 0] = now(); __hip_atomic_store(go, k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__global__ void k_wait(u64* done, u64 k, u64 timeout_ticks, int* status, u64* stamps) {
    const u64 t0 = now();
    while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < k) {
        __builtin_amdgcn_s_sleep(1);
        if (now() - t0 > timeout_ticks) { status[1] = (int)k; break; }
    }
    stamps[4 * k + 3] = now();
}

int main(int argc, char** argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 2000;
    int can = 0;
    CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    if (!can) { printf("NEGATIVE: no stream wait-value on this device / runtime\n"); return 0; }
    const bool plain_flags = argc > 2 && atoi(argv[2]) != 0;      // doorbells in hipMalloc memory instead of signal memory (kernel ring / wait only)
    float4* arena; float *act, *obs; unsigned* cnt; int* status; u64 *go, *done, *stamps, *go_sig, *done_sig, *go_dev;
    std::vector<u64> hs((size_t)4 * (K + 1));
    CK(hipMalloc(&stamps, sizeof(u64) * 4 * (K + 1)));
    CK(hipMalloc(&go_dev, 256));
    CK(hipMalloc(&arena, sizeof(float4) * 64 * G * NB));
    CK(hipMalloc(&act, sizeof(float) * 64 * NB * ACT));
    CK(hipMalloc(&obs, sizeof(float) * 64 * NB * OBS));
    CK(hipMalloc(&cnt, sizeof(unsigned) * (16 * 32 + 32)));
    CK(hipMalloc(&status, sizeof(int) * 4));
    CK(hipExtMallocWithFlags((void**)&go_sig, 8, hipMallocSignalMemory));
    CK(hipExtMallocWithFlags((void**)&done_sig, 8, hipMallocSignalMemory));
    CK(hipMemset(arena, 0, sizeof(float4) * 64 * G * NB));
    CK(hipMemset(act, 0, sizeof(float) * 64 * NB * ACT));
    CK(hipMemset(obs, 0, sizeof(float) * 64 * NB * OBS));
    hipStream_t sp, sb;
    CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 g(NB), b(64);
    float ms;

    for (int n_chain : {0, 600, 900}) {
        // ---- one launch per step
        for (int k = 0; k < 100; ++k) { k_policy<<<g, b, 0, sb>>>(obs, act); k_step<<<g, b, 0, sb>>>(arena, act, obs, n_chain); }
        CK(hipStreamSynchronize(sb));
        CK(hipEventRecord(e0, sb));
        for (int k = 0; k < K; ++k) { k_policy<<<g, b, 0, sb>>>(obs, act); k_step<<<g, b, 0, sb>>>(arena, act, obs, n_chain); }
        CK(hipEventRecord(e1, sb));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double per_launch = 1e3 * ms / K;
        // the step-like kernel alone, back to back
        CK(hipEventRecord(e0, sb));
        for (int k = 0; k < K; ++k) k_step<<<g, b, 0, sb>>>(arena, act, obs, n_chain);
        CK(hipEventRecord(e1, sb));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double alone = 1e3 * ms / K;

        // ---- persistent + doorbells
        for (int mode : {1, 0, 3, 2}) {      // bit 0: payload, bit 1: kernels ring / wait instead of hipStreamWriteValue64 / WaitValue64
            const int payload = mode & 1, by_kernels = mode >> 1;
            if (!payload && n_chain != 0) continue;
            const bool dev_flags = plain_flags && by_kernels;
            go = dev_flags ? go_dev : go_sig; done = dev_flags ? go_dev + 16 : done_sig;
            CK(hipMemset(stamps, 0, sizeof(u64) * 4 * (K + 1)));
            CK(hipMemset(cnt, 0, sizeof(unsigned) * (16 * 32 + 32)));
            CK(hipMemset(status, 0, sizeof(int) * 4));
            u64 zero = 0;
            CK(hipMemcpy(go, &zero, 8, hipMemcpyHostToDevice));
            CK(hipMemcpy(done, &zero, 8, hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            k_persist<<<g, b, 0, sp>>>(arena, act, obs, go, done, cnt, status, K, n_chain, n_chain / 5, payload, 2000000ull /* 20 ms */, stamps);
            CK(hipGetLastError());
            CK(hipEventRecord(e0, sb));
            for (int k = 1; k <= K; ++k) {
                k_policy<<<g, b, 0, sb>>>(obs, act);
                if (by_kernels) {
                    k_ring<<<1, 1, 0, sb>>>(go, (u64)k, stamps);
                    k_wait<<<1, 1, 0, sb>>>(done, (u64)k, 2000000ull, status, stamps);
                } else {
                    CK(hipStreamWriteValue64(sb, go, (u64)k, 0));
                    CK(hipStreamWaitValue64(sb, done, (u64)k, hipStreamWaitValueGte, ~0ull));
                }
            }
            CK(hipEventRecord(e1, sb));
            CK(hipEventSynchronize(e1));
            CK(hipStreamSynchronize(sp));
            CK(hipEventElapsedTime(&ms, e0, e1));
            int st[4];
            CK(hipMemcpy(st, status, sizeof st, hipMemcpyDeviceToHost));
            printf("chain %4d  %-26s launches: policy+step %6.2f us/step (step kernel alone %5.2f)   doorbell by %-32s %7.2f us/step%s\n", n_chain,
                   payload ? "payload (act in, obs out)" : "round trip only", per_launch, alone,
                   by_kernels ? "1-thread ring / wait kernels:" : "hipStreamWriteValue64/WaitValue64:", 1e3 * ms / K,
                   (st[0] || st[1]) ? "   [TIMED OUT: a wait exceeded 20 ms; figure invalid]" : "");
            if (by_kernels && !(st[0] || st[1])) {     // where the time goes (100 MHz ticks -> us), means over the steps
                CK(hipMemcpy(hs.data(), stamps, sizeof(u64) * 4 * (K + 1), hipMemcpyDeviceToHost));
                double a = 0, bq = 0, c = 0, d = 0;
                for (int k = 2; k <= K; ++k) {
                    a += (double)(long long)(hs[4 * k + 1] - hs[4 * k + 0]);          // ring -> the persistent kernel's block 0 has seen it
                    bq += (double)(long long)(hs[4 * k + 2] - hs[4 * k + 1]);         // seen -> last of the 1024 waves has arrived
                    c += (double)(long long)(hs[4 * k + 3] - hs[4 * k + 2]);          // published -> the wait kernel leaves its loop
                    d += (double)(long long)(hs[4 * k + 0] - hs[4 * (k - 1) + 3]);    // wait kernel gone -> policy kernel -> next ring
                }
                const double f = 0.01 / (K - 1);
                printf("      %s flags: ring->seen %.2f us, seen->all 1024 waves arrived %.2f us, published->wait kernel released %.2f us, released->policy->next ring %.2f us\n",
                       dev_flags ? "hipMalloc" : "signal-memory", a * f, bq * f, c * f, d * f);
            }
            if (st[0]) printf("   first timed-out step: %d\n", st[0]);
        }
    }
    return 0;
}
