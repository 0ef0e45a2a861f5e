// Exhaustive / randomised checks of the short correctly-rounded float32 sqrt and division used by the step kernel
// (hlx_device.h: sqrt_rn, div_rn) against the compiler's full expansions (sqrtf, operator/).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I hlynr_intercept_amd/csrc -o /tmp/rn_check tools/micro/rn_check.hip && /tmp/rn_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../include/hlx.h"
#include "hlx_device.h"
using namespace hlx;

__global__ void sqrt_all(unsigned long long* bad, uint32_t* first) {
    // every non-negative float: 0, denormals (reported separately), normals, +inf
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= 0x7f800000ull; b += stride) {
        const float x = __uint_as_float((uint32_t)b);
        const bool denormal = b != 0 && b < 0x00800000ull;
        const float a = sqrt_rn(x), r = sqrtf(x);
        if (__float_as_uint(a) != __float_as_uint(r)) {
            atomicAdd(&bad[denormal ? 1 : 0], 1ull);
            if (!denormal) { atomicMin(first, (uint32_t)b); atomicMax(first + 1, (uint32_t)b); }
        }
    }
}

__device__ uint32_t mix(uint64_t& s) { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }

__global__ void div_small_numerator(unsigned long long* bad, int rounds, int ea_min, int ea_max) {
    // numerator exponent in [ea_min, ea_max] (biased), denominator in [0.25, 4)
    uint64_t s = 0xD1B54A32D192ED03ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int k = 0; k < rounds; ++k) {
        uint32_t ma = mix(s), mb = mix(s), e = mix(s);
        const uint32_t ea = ea_min + (e & 0xffff) % (ea_max - ea_min + 1), eb = 125 + (e >> 16) % 4;
        const float a = __uint_as_float((ma & 0x807fffffu) | (ea << 23)), b = __uint_as_float((mb & 0x807fffffu) | (eb << 23));
        if (__float_as_uint(div_rn(a, b)) != __float_as_uint(a / b)) atomicAdd(bad, 1ull);
    }
}

__global__ void div_random(unsigned long long* bad, uint32_t* ex, int rounds, int emin, int emax) {
    uint64_t s = 0x9E3779B97F4A7C15ull * ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1);
    for (int k = 0; k < rounds; ++k) {
        // random mantissas and signs, exponents within [emin, emax] (biased)
        uint32_t ma = mix(s), mb = mix(s), e = mix(s);
        const uint32_t ea = emin + (e & 0xffff) % (emax - emin + 1), eb = emin + (e >> 16) % (emax - emin + 1);
        const float a = __uint_as_float((ma & 0x807fffffu) | (ea << 23)), b = __uint_as_float((mb & 0x807fffffu) | (eb << 23));
        const float q = div_rn(a, b), r = a / b;
        if (__float_as_uint(q) != __float_as_uint(r)) {
            if (atomicAdd(bad, 1ull) == 0) { ex[0] = __float_as_uint(a); ex[1] = __float_as_uint(b); }
        }
    }
}

int main() {
    unsigned long long* bad; uint32_t* first;
    hipMalloc(&bad, 32); hipMalloc(&first, 16);
    unsigned long long h[4] = {0, 0, 0, 0}; uint32_t f[4] = {0xffffffffu, 0, 0, 0};
    hipMemcpy(bad, h, 32, hipMemcpyHostToDevice); hipMemcpy(first, f, 16, hipMemcpyHostToDevice);
    sqrt_all<<<4096, 256>>>(bad, first);
    hipDeviceSynchronize();
    hipMemcpy(h, bad, 32, hipMemcpyDeviceToHost); hipMemcpy(f, first, 16, hipMemcpyDeviceToHost);
    printf("sqrt_rn vs sqrtf over all non-negative floats: %llu mismatches on 0/normal/inf inputs (first 0x%08x, last 0x%08x), %llu on denormal inputs\n",
           h[0], f[0], f[1], h[1]);
    int rc = h[0] != 0;
    // division: operands with exponents in [2^-40, 2^40] (every quotient normal), 2^34 pairs
    h[0] = 0; hipMemcpy(bad, h, 8, hipMemcpyHostToDevice);
    div_random<<<4096, 256>>>(bad, first, 16384, 127 - 40, 127 + 40);
    hipDeviceSynchronize();
    hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(f, first, 16, hipMemcpyDeviceToHost);
    printf("div_rn vs operator/ on 2^34 random pairs, exponents within 2^+-40: %llu mismatches (example a=0x%08x b=0x%08x)\n", h[0], f[0], f[1]);
    rc |= h[0] != 0;
    for (int lo = 1; lo <= 27; lo += 13) {       // numerators 2^-126..2^-114, 2^-113..2^-101, 2^-100..2^-88
        h[0] = 0; hipMemcpy(bad, h, 8, hipMemcpyHostToDevice);
        div_small_numerator<<<4096, 256>>>(bad, 1024, lo, lo + 12);
        hipDeviceSynchronize();
        hipMemcpy(h, bad, 8, hipMemcpyDeviceToHost);
        printf("div_rn, numerator in [2^%d, 2^%d], denominator in [0.25, 4): %llu mismatches in 2^30 pairs\n", lo - 127, lo - 115 + 1, h[0]);
    }
    return rc;
}
