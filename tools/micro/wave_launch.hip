// What does launching a wave cost?  Empty kernels (grids of 64-thread workgroups) replayed from a hipGraph so that the
// host is out of the picture; variants: plain, 200 VGPRs allocated, 6.6 KB LDS allocated, 14 kernel-argument dwords.
// Build twice: with and without -mllvm -amdgpu-kernarg-preload-count=16.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(64) void k_plain(float* a) {}
__global__ __launch_bounds__(64) void k_vgpr(float* a) { asm volatile("v_mov_b32 v199, 0" ::: "v199"); }
__global__ __launch_bounds__(64) void k_lds(float* a) {
    __shared__ float t[1664];
    if (a == nullptr) { t[threadIdx.x] = 1.f; a[0] = t[63 - threadIdx.x]; }
}
__global__ __launch_bounds__(64) void k_args(float* a, const float* b, const float* c, unsigned long long t, unsigned long long seed,
                                            long long off, int n, unsigned slots, float* o0, float* o1, unsigned char* o2, unsigned char* o3) {
    if (n == -12345) a[0] = (float)(t + seed + off + slots) + b[0] + c[0] + o0[0] + o1[0] + o2[0] + o3[0];
}
__global__ __launch_bounds__(64) void k_all(float* a, const float* b, const float* c, unsigned long long t, unsigned long long seed,
                                           long long off, int n, unsigned slots, float* o0, float* o1, unsigned char* o2, unsigned char* o3) {
    __shared__ float tl[1664];
    asm volatile("v_mov_b32 v199, 0" ::: "v199");
    if (n == -12345) { tl[threadIdx.x] = 1.f; a[0] = tl[63 - threadIdx.x] + (float)(t + seed + off + slots) + b[0] + c[0] + o0[0] + o1[0] + o2[0] + o3[0]; }
}

template <typename F>
double graph_us(F launch, int per_graph = 50, int replays = 20) {
    hipStream_t st; (void)hipStreamCreate(&st);
    hipGraph_t gr; hipGraphExec_t ex;
    (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < per_graph; ++i) launch(st);
    (void)hipStreamEndCapture(st, &gr);
    (void)hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
    (void)hipGraphLaunch(ex, st); (void)hipStreamSynchronize(st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, st);
    for (int i = 0; i < replays; ++i) (void)hipGraphLaunch(ex, st);
    (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return 1e3 * ms / (per_graph * replays);
}

int main() {
    float* a; (void)hipMalloc(&a, 1 << 20);
    unsigned char* u = reinterpret_cast<unsigned char*>(a);
    for (int g : {64, 256, 1024, 2048, 4096}) {
        dim3 grid(g), b(64);
        printf("grid %5d x 64:  plain %5.2f  vgpr200 %5.2f  lds6.6K %5.2f  args14 %5.2f  all %5.2f  us per launch\n", g,
               graph_us([&](hipStream_t s) { k_plain<<<grid, b, 0, s>>>(a); }),
               graph_us([&](hipStream_t s) { k_vgpr<<<grid, b, 0, s>>>(a); }),
               graph_us([&](hipStream_t s) { k_lds<<<grid, b, 0, s>>>(a); }),
               graph_us([&](hipStream_t s) { k_args<<<grid, b, 0, s>>>(a, a, a, 1ull, 2ull, 3ll, 4, 5u, a, a, u, u); }),
               graph_us([&](hipStream_t s) { k_all<<<grid, b, 0, s>>>(a, a, a, 1ull, 2ull, 3ll, 4, 5u, a, a, u, u); }));
    }
    return 0;
}
