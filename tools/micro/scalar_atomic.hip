// Does gfx950 execute scalar memory atomics (s_atomic_add ... glc: returns through lgkmcnt, not vmcnt)?  The assembler accepts them.
// One wave per block claims `k` slots of a counter with a scalar atomic; the claims must tile [0, total) exactly.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/scalar_atomic.hip -o /tmp/scalar_atomic && /tmp/scalar_atomic
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void claim(int* counter, int* base_out, int* k_out) {
    int k = 1 + (blockIdx.x % 5);
    int r = k;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"(counter) : "memory");
    if (threadIdx.x == 0) { base_out[blockIdx.x] = r; k_out[blockIdx.x] = k; }
}
int main() {
    const int B = 4096;
    int *c, *b, *k;
    hipMalloc(&c, 4); hipMalloc(&b, 4 * B); hipMalloc(&k, 4 * B);
    hipMemset(c, 0, 4);
    hipLaunchKernelGGL(claim, dim3(B), dim3(64), 0, 0, c, b, k);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAULT %s\n", hipGetErrorString(hipGetLastError())); return 2; }
    std::vector<int> hb(B), hk(B); int total = 0;
    hipMemcpy(hb.data(), b, 4 * B, hipMemcpyDeviceToHost); hipMemcpy(hk.data(), k, 4 * B, hipMemcpyDeviceToHost); hipMemcpy(&total, c, 4, hipMemcpyDeviceToHost);
    std::vector<std::pair<int, int>> v; long want = 0;
    for (int i = 0; i < B; ++i) { v.push_back({hb[i], hk[i]}); want += hk[i]; }
    std::sort(v.begin(), v.end());
    bool ok = total == want; int at = 0;
    for (auto& p : v) { if (p.first != at) ok = false; at += p.second; }
    printf("scalar atomics: total %d (want %ld), claims tile the range: %s\n", total, want, ok ? "yes" : "NO");
    return ok ? 0 : 1;
}
