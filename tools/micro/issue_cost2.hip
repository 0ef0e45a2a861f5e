// Issue cost (cycles per instruction, one wave alone on its SIMD) of the vector instructions the step kernel is
// made of.  16 independent destination registers per op, 64 x 16 instructions timed with s_memtime.
// hipcc --offload-arch=gfx950 -O3 -o issue_cost issue_cost.hip && ./issue_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

#define KERNEL32(NAME, ASMSTR)                                                                        \
    __global__ __launch_bounds__(512) void NAME(unsigned long long* out, unsigned seed) {              \
        unsigned r[16];                                                                                \
        for (int i = 0; i < 16; ++i) r[i] = seed * (i + 3) + threadIdx.x;                              \
        unsigned a = seed | 1u, b = seed * 7u + 5u;                                                    \
        unsigned long long t0, t1;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                      \
        for (int it = 0; it < 64; ++it) {                                                              \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASMSTR : "+v"(r[i]) : "v"(a), "v"(b)); \
        }                                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                      \
        unsigned acc = 0;                                                                              \
        for (int i = 0; i < 16; ++i) acc ^= r[i];                                                      \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = acc; }                                      \
    }

#define KERNEL64(NAME, ASMSTR)                                                                        \
    __global__ __launch_bounds__(512) void NAME(unsigned long long* out, unsigned seed) {              \
        double r[16];                                                                                  \
        for (int i = 0; i < 16; ++i) r[i] = 1.0 + 1e-3 * (seed * (i + 3) % 17 + threadIdx.x);          \
        double a = 1.0000001, b = 1e-9 * seed;                                                         \
        unsigned x = seed | 1u, y = seed * 7u + 5u;                                                    \
        float o[16];                                                                                   \
        for (int i = 0; i < 16; ++i) o[i] = (float)i;                                                  \
        unsigned long long t0, t1;                                                                     \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                      \
        for (int it = 0; it < 64; ++it) {                                                              \
            _Pragma("unroll") for (int i = 0; i < 16; ++i)                                             \
                asm volatile(ASMSTR : "+v"(r[i]), "+v"(o[i]) : "v"(a), "v"(b), "v"(x), "v"(y));        \
        }                                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                      \
        double acc = 0;                                                                                \
        for (int i = 0; i < 16; ++i) acc += r[i] + o[i];                                               \
        if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (unsigned long long)acc; }                  \
    }

KERNEL32(k_add_f32, "v_add_f32 %0, %0, %1")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_xor, "v_xor_b32 %0, %0, %1")
KERNEL32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL32(k_alignbit, "v_alignbit_b32 %0, %0, %0, 13")
KERNEL32(k_mul_lo, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi, "v_mul_hi_u32 %0, %0, %1")
KERNEL32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL32(k_sin_f32, "v_sin_f32 %0, %0")
KERNEL32(k_log_f32, "v_log_f32 %0, %0")
KERNEL32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL32(k_readlane, "v_readlane_b32 s20, %0, 3")
KERNEL32(k_mov, "v_mov_b32 %0, %1")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %2")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %2")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %2, %3")
KERNEL64(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL64(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %4, %5, %0")
KERNEL64(k_cvt_f64_f32, "v_cvt_f64_f32 %0, %1")
KERNEL64(k_cvt_f32_f64, "v_cvt_f32_f64 %1, %0")
KERNEL64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 0, %2")
KERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %2")
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %2, %3")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %2")

int main() {
    unsigned long long* out;
    hipMalloc(&out, 16);
    unsigned long long h[2];
#define RUN(NAME)                                                                        \
    do {                                                                                 \
        NAME<<<1, 64>>>(out, 12345u); NAME<<<1, 64>>>(out, 12345u);                       \
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);                                    \
        double one = (double)h[0] / 1024.0;                                              \
        NAME<<<1, 512>>>(out, 12345u); NAME<<<1, 512>>>(out, 12345u);                     \
        hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);                                    \
        printf("%-22s 1 wave/SIMD %6.2f cycles/instr | 2 waves/SIMD %6.2f per wave-instr (%.2f per SIMD-instr)\n", #NAME + 2, one, (double)h[0] / 1024.0, (double)h[0] / 2048.0); \
    } while (0)
    RUN(k_add_f32); RUN(k_fma_f32); RUN(k_xor); RUN(k_add_u32); RUN(k_alignbit); RUN(k_mov); RUN(k_cndmask); RUN(k_readlane);
    RUN(k_mul_lo); RUN(k_mul_hi); RUN(k_mad_u64_u32); RUN(k_lshl_add_u64);
    RUN(k_sqrt_f32); RUN(k_rcp_f32); RUN(k_sin_f32); RUN(k_log_f32);
    RUN(k_pk_mul_f32); RUN(k_pk_fma_f32); RUN(k_pk_add_f32);
    RUN(k_add_f64); RUN(k_mul_f64); RUN(k_fma_f64); RUN(k_rcp_f64); RUN(k_sqrt_f64); RUN(k_cvt_f64_f32); RUN(k_cvt_f32_f64);
    return 0;
}
