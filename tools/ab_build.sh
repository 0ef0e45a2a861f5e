#!/bin/bash
# tools/ab_build.sh NAME "-DFOO=1 -DBAR=0"  -> hlynr_intercept_amd/libhlx_NAME.so (A/B experiments)
# add -DHLX_AB_MINIMAL to compile only the base variant (bench.py's default workload): ~15 s instead of ~90 s
set -e
cd "$(dirname "$0")/../hlynr_intercept_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 -mllvm -amdgpu-sched-strategy=max-ilp -Wno-unused-value $2 -o ../libhlx_$1.so hlx_kernels.hip
cd ../.. && python3 tools/check_hot_words.py hlynr_intercept_amd/libhlx_$1.so   # refuse variants that spill a hot-word register
