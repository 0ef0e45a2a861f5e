// Floor of one launch at the benchmark's shape (1024 workgroups x 64 lanes, back-to-back dependent launches):
//   empty     : dispatch + completion only
//   copyN     : every lane loads G 16-byte words of its block and stores S of them back (no arithmetic)
//   spin      : copy + a dependent FMA chain of C iterations between load and store (pure-latency compute)
// hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip && ./launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_empty(float4* a) {}

template <int G, int S, bool WT>
__global__ __launch_bounds__(64) void k_copy(float4* __restrict__ arena, float4* __restrict__ out, int chain) {
    float4* A = arena + (size_t)blockIdx.x * (G * 64) + threadIdx.x;
    float4 v[G];
#pragma unroll
    for (int g = 0; g < G; ++g) v[g] = A[g * 64];
    float acc = v[0].x;
    for (int c = 0; c < chain; ++c) acc = acc * 1.0000001f + 1e-9f;
    v[0].x = acc;
    float4* O = out + (size_t)blockIdx.x * (S * 64) + threadIdx.x;
    if (WT) {
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        auto r = __builtin_amdgcn_make_buffer_rsrc(out + (size_t)blockIdx.x * (S * 64), 0, S * 1024, 0x00020000);
#pragma unroll
        for (int g = 0; g < S; ++g) {
            float4 x = v[g % G];
            u32x4 d = {__float_as_uint(x.x), __float_as_uint(x.y), __float_as_uint(x.z), __float_as_uint(x.w)};
            __builtin_amdgcn_raw_buffer_store_b128(d, r, threadIdx.x * 16, g * 1024, 16);
        }
    } else {
#pragma unroll
        for (int g = 0; g < S; ++g) O[g * 64] = v[g % G];
    }
}

template <typename F>
double time_us(F launch, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return 1e3 * ms / reps;
}

int main() {
    const int NB = 1024;
    float4 *arena, *out;
    CK(hipMalloc(&arena, sizeof(float4) * 64 * 32 * NB));
    CK(hipMalloc(&out, sizeof(float4) * 64 * 32 * NB));
    CK(hipMemset(arena, 0, sizeof(float4) * 64 * 32 * NB));
    dim3 g(NB), b(64);
    printf("empty                          %6.2f us\n", time_us([&] { k_empty<<<g, b>>>(arena); }, 2000));
    printf("empty 256 x 256                %6.2f us\n", time_us([&] { k_empty<<<dim3(256), dim3(256)>>>(arena); }, 2000));
    printf("empty 128 x 512                %6.2f us\n", time_us([&] { k_empty<<<dim3(128), dim3(512)>>>(arena); }, 2000));
    printf("empty 1 x 64                   %6.2f us\n", time_us([&] { k_empty<<<dim3(1), dim3(64)>>>(arena); }, 2000));
    {   // the same empty launch replayed from a hipGraph of 100 kernel nodes
        hipStream_t st; hipStreamCreate(&st);
        hipGraph_t gr; hipGraphExec_t ex;
        hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
        for (int i = 0; i < 100; ++i) k_empty<<<g, b, 0, st>>>(arena);
        hipStreamEndCapture(st, &gr);
        hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
        hipGraphLaunch(ex, st); hipStreamSynchronize(st);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, st);
        for (int i = 0; i < 20; ++i) hipGraphLaunch(ex, st);
        hipEventRecord(e1, st); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("empty 1024 x 64 via hipGraph   %6.2f us\n", 1e3 * ms / 2000);
    }
    printf("copy 1 load 1 store            %6.2f us\n", time_us([&] { k_copy<1, 1, false><<<g, b>>>(arena, arena, 0); }, 2000));
    printf("copy 15 loads 22 stores plain  %6.2f us   (15.7 MB in, 23 MB out: the step's traffic)\n", time_us([&] { k_copy<15, 22, false><<<g, b>>>(arena, out, 0); }, 2000));
    printf("copy 15 loads 22 stores sc1    %6.2f us\n", time_us([&] { k_copy<15, 22, true><<<g, b>>>(arena, out, 0); }, 2000));
    printf("copy 15 loads  1 store         %6.2f us\n", time_us([&] { k_copy<15, 1, false><<<g, b>>>(arena, out, 0); }, 2000));
    printf("copy 1 load 22 stores sc1      %6.2f us\n", time_us([&] { k_copy<1, 22, true><<<g, b>>>(arena, out, 0); }, 2000));
    return 0;
}
