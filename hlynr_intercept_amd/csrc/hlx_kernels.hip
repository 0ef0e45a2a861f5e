// hlx_kernels.hip -- the fused intercept-environment step for MI355X (gfx950 / CDNA4).
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off   (see hlx_device.h, ARITHMETIC DISCIPLINE)
//
// One kernel launch = one `VecEnv.step`: safety clamp -> interceptor 6-DOF integrator -> missile
// -> wind -> intercept/termination -> reward -> radar/ground-radar/datalink/fusion -> Kalman filter
// -> 26-D observation, plus the auto-reset (spawn + first observation) of finished environments.
// Reference order of operations: rl_system/environment.py:605-859.
//
// Mapping to the hardware
//   * one lane per environment, 64-lane workgroups (= one wavefront): no workgroup barrier is ever
//     needed between waves, and N/64 workgroups spread evenly over the 256 CUs / 8 XCDs.  The
//     env -> workgroup mapping is the same every launch, so the slice of the arena a workgroup
//     touches stays in the L2 of the XCD it is dispatched to (round-robin by workgroup id).
//   * state arena = struct-of-arrays of 16-byte groups: every state load/store is a 16-byte-per-
//     lane, 1 KiB-per-wave fully coalesced access (11 groups base physics, 13 with physics v2 + DR).
//   * delay rings are planes indexed by the GLOBAL vec-step clock, so ring traffic is coalesced
//     too although every environment is at a different step of its own episode.
//   * the 26-float observation row of each lane goes through a [64][26] LDS tile and leaves as
//     16-byte coalesced stores of the row-major [N][26] array the policy consumes.
//   * finished environments: wave ballot + popcount + one atomic per wave compacts their indices.
//   * no MFMA: per-environment physics has no dense contraction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/hlx.h"
#include "hlx_device.h"
#include "hlx_kargs.h"
#include "hlx_baked_gen.h"

using namespace hlx;

namespace {

#define HAS(f) ((FL & (uint32_t)(f)) != 0u)
// LDS hand-over between the lanes of ONE wave (every workgroup of this kernel is a single wavefront): the LDS pipeline
// executes a wave's DS instructions in issue order, so a later ds_read already sees the earlier ds_writes of other lanes;
// all that is needed is that the compiler keeps the program order.  __syncthreads() would add `s_waitcnt vmcnt(0)` -- on
// this part that also waits for every outstanding STORE, and behind the state stores it stalled the wave for their
// write-through acknowledgements before the observation tile could leave.
#define WAVE_LDS_SYNC()                                           \
    do {                                                          \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    \
        __builtin_amdgcn_wave_barrier();                          \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");    \
    } while (0)
// rare branches: laid out of line so that the hot path is one contiguous instruction stream (the step is ~2000 instructions
// executed once per wave and launch: instruction fetch is cold, and every taken branch over a cold block breaks the prefetch)
#define RARE(x) __builtin_expect(!!(x), 0)
// The optional info planes: in line (round 3).  Until round 2 the benchmark's launches asked for none and the block sat out of
// line behind RARE(); the headline is now the launch the drop-in really issues (hlx_step with every info plane), and there the
// block runs in every wave -- out of line it would cost each of them two taken far branches.  -DHLX_INFO_RARE=1: the old layout (A/B).
#ifndef HLX_INFO_RARE
#define HLX_INFO_RARE 0
#endif
#define INFO_WANTED(x) (HLX_INFO_RARE ? RARE(x) : (x))
// Pointers that reach the kernel through the hot block (optional outputs) or through *P are GENERIC to the compiler, and a
// generic access is a FLAT instruction: it counts in vmcnt AND lgkmcnt, may complete out of order with the global ones, and while
// one may be pending every later `s_waitcnt vmcnt(n)` the compiler emits becomes vmcnt(0) -- on every path, taken or not.  (Found
// in round 3: one flat_load for the info['fuel_used'] accumulator turned the wait for the missile groups into a wait for the
// whole second load batch, 0.45 us per launch in every form of the step; the info planes' flat stores made each later
// `lgkmcnt(0)` of the LDS tile hand-over wait for their acknowledgements.)  G(p) says what they are: global memory.
template <typename T> using gptr_t = __attribute__((address_space(1))) T*;
template <typename T> DEV gptr_t<T> G(T* p) { return (gptr_t<T>)p; }

// Diagnostic build only (-DHLX_STAMPS): lane 0 of every wave records s_memtime at a few program points into
// a.stamps[block][16].  No stamp executes in the product build, and no output is ever computed from one.
#ifdef HLX_STAMPS
#ifdef HLX_STAMP_REALTIME      // one 100 MHz counter for the whole chip (s_memtime counts per CU, unsynchronised): launch timelines
#define STAMP_CLOCK "s_memrealtime"
#else
#define STAMP_CLOCK "s_memtime"
#endif
#define STAMP_RAW(k)                                                                                       \
    do {                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        unsigned long long t_;                                                                         \
        asm volatile(STAMP_CLOCK " %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
        if (lane == 0) stamp_base[(k)] = t_;                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    } while (0)
// HLX_STAMPS=1: coarse map of the whole wave; HLX_STAMPS=2: slots 1..6 re-used for a close-up of one segment
#define STAMP(k) do { if (HLX_STAMPS == 1 || (k) == 0 || (k) >= 7) STAMP_RAW(k); } while (0)
#define STAMP2(k) do { if (HLX_STAMPS == 2) STAMP_RAW(k); } while (0)
#elif defined(HLX_MARKS)   // listing-only build: section markers in the .s, for per-section instruction counts (tools/isa_sections.py)
#define STAMP(k) asm volatile("; HLXMARK " #k)
#define STAMP2(k) do { } while (0)
#else
#define STAMP(k) do { } while (0)
#define STAMP2(k) do { } while (0)
#endif

// State / ring / observation stores are WRITE-THROUGH (`sc1`) buffer stores: nothing this launch stores is read
// again before the next launch (whose acquire drops the L2 anyway), and with plain stores the ~21 MB a launch
// writes would sit dirty in the XCD L2s until the end-of-kernel release writes it all back at once, after the
// last wave has finished; written through, the bytes leave while the waves still compute and the release finds
// the L2 clean (MI355X_MICROARCH.md, "stores of each flavour" / publish-large).
#ifndef HLX_ST_AUX
#define HLX_ST_AUX 16   // buffer-instruction cache bits: 16 = sc1 (write-through); 0 = plain (A/B builds)
#endif
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
DEV void wt16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float4 v) {
    u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, HLX_ST_AUX);
}
#ifndef HLX_INFO_AUX
#define HLX_INFO_AUX HLX_ST_AUX   // cache bits of the packed info stores (A/B: 0 = plain)
#endif
DEV void wt16i(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, float4 v) {      // hlx_info_soa.packed words
    u32x4 d = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, HLX_INFO_AUX);
}
// (pool entries: plain per-lane stores -- a fill touches a few thousand environments)
struct PoolPtrs {
    size_t blocks;          // 64-environment blocks
    float4* pool;           // [blocks][POOL_GROUPS][64]
    uint32_t *tag, *ep_cur; // [blocks * 64] each
    int32_t* rf_cnt;            // [4]: [2] = respawns computed inside step launches for want of an entry, [3] = ... in a crowded wave (diagnostics)
    unsigned long long* rf_mask;   // [blocks]: the lanes of each block that have used their entry since the last fill
};
// the pool lives behind the rings in the arena allocation (hlx_host.inc hlx_create computes the same addresses)
DEV PoolPtrs pool_ptrs(float4* arena, int n, int g_planes, int o_planes) {
    PoolPtrs p;
    p.blocks = (size_t)((n + 63) >> 6);
    p.pool = arena + p.blocks * (N_GROUPS * 64) + ((size_t)(g_planes * GROUND_RING_WORDS16) + (size_t)o_planes) * (size_t)n;
    p.tag = reinterpret_cast<uint32_t*>(p.pool + p.blocks * (POOL_GROUPS * 64));
    p.ep_cur = p.tag + p.blocks * 64;
    p.rf_cnt = reinterpret_cast<int32_t*>(p.ep_cur + p.blocks * 64);
    p.rf_mask = reinterpret_cast<unsigned long long*>(p.rf_cnt + 4);
    return p;
}
// A value whose only reader asks for it in an accumulation register (AGPR) is loaded into one and stays there: the way to keep
// ~90 prefetched dwords out of the 256 ordinary VGPRs while a section that needs most of those runs (a wave that is alone on its
// SIMD owns all 512 registers of a lane, but only 256 of them are addressable as ordinary VGPRs).
DEV float acc_read(float x) {
    float r;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(r) : "a"(x));
    return r;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
DEV float4 pool_get(const float4* PA, int g) {
    const f32x4 v = *(const __attribute__((address_space(1))) f32x4*)(PA + g * 64);
    return make_float4(v.x, v.y, v.z, v.w);
}
DEV void pool_put(float4* PA, int g, float4 v) { *(__attribute__((address_space(1))) f32x4*)(PA + g * 64) = f32x4{v.x, v.y, v.z, v.w}; }
DEV void pool_put(float4* PA, int g, double2 v) {
    pool_put(PA, g, make_float4(__int_as_float(__double2loint(v.x)), __int_as_float(__double2hiint(v.x)), __int_as_float(__double2loint(v.y)),
                                __int_as_float(__double2hiint(v.y))));
}
DEV void wt16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff, double2 v) {
    u32x4 d = {(uint32_t)__double2loint(v.x), (uint32_t)__double2hiint(v.x), (uint32_t)__double2loint(v.y), (uint32_t)__double2hiint(v.y)};
    __builtin_amdgcn_raw_buffer_store_b128(d, r, voff, soff, HLX_ST_AUX);
}

// Hot-parameter access: every lane holds two dwords of the 512-byte block (hotw0 = dword `lane`, hotw1 = dword
// 64 + `lane`); a constant is fetched where it is used with one v_readlane per dword.  (Reading the whole block
// into SGPRs once cost ~170 v_readlane plus ~200 v_writelane / v_readlane of SGPR spill traffic per wave.)
template <typename T> struct HotRd;
template <> struct HotRd<float> { static DEV float get(uint32_t w0, uint32_t w1, int off) {
    return __uint_as_float(__builtin_amdgcn_readlane(off < 256 ? w0 : w1, (off >> 2) & 63)); } };
template <> struct HotRd<int32_t> { static DEV int32_t get(uint32_t w0, uint32_t w1, int off) {
    return (int32_t)__builtin_amdgcn_readlane(off < 256 ? w0 : w1, (off >> 2) & 63); } };
template <> struct HotRd<uint32_t> { static DEV uint32_t get(uint32_t w0, uint32_t w1, int off) {
    return __builtin_amdgcn_readlane(off < 256 ? w0 : w1, (off >> 2) & 63); } };
template <> struct HotRd<double> { static DEV double get(uint32_t w0, uint32_t w1, int off) {
    return __hiloint2double((int)HotRd<uint32_t>::get(w0, w1, off + 4), (int)HotRd<uint32_t>::get(w0, w1, off)); } };
template <typename P> struct HotRd<P*> { static DEV P* get(uint32_t w0, uint32_t w1, int off) {
    const unsigned long long lo = HotRd<uint32_t>::get(w0, w1, off), hi = HotRd<uint32_t>::get(w0, w1, off + 4);
    return reinterpret_cast<P*>(lo | (hi << 32)); } };
// BAKED instantiations (template parameter BAKE = 1 + index into HLX_BAKED_TAB, hlx_baked_gen.h): the configuration
// constants (KHot::c) of a shipped scenario preset are compile-time literals -- no cross-lane fetch, no SGPR hazard
// wait states, and the optimiser folds what depends on them; curriculum scalars and optional-buffer pointers, which can
// change between launches, still come from the hot words.  hlx_create selects a baked instantiation only when the KCfg
// it derives equals the table byte for byte.
template <typename T> struct BakedRd;
template <> struct BakedRd<float> { static constexpr float get(const uint32_t* t, int off) { return __builtin_bit_cast(float, t[off >> 2]); } };
template <> struct BakedRd<int32_t> { static constexpr int32_t get(const uint32_t* t, int off) { return (int32_t)t[off >> 2]; } };
template <> struct BakedRd<uint32_t> { static constexpr uint32_t get(const uint32_t* t, int off) { return t[off >> 2]; } };
template <> struct BakedRd<double> { static constexpr double get(const uint32_t* t, int off) {
    return __builtin_bit_cast(double, (unsigned long long)t[off >> 2] | ((unsigned long long)t[(off >> 2) + 1] << 32)); } };
// -DHLX_HOT_FROM_MEMORY=1, the SAFE build: every hot constant is read from the parameter block in memory (uniform loads through
// the kernel argument `P`) instead of across lanes out of two vector registers.  Slower -- a cold scalar load is 1-2 k cycles for the
// lone wave of a SIMD, which is why the product build does not do it -- but independent of what the register allocator does with
// those two registers: it is what hlynr_intercept_amd/build.py falls back to when hotcheck.py refuses the product build (a compiler
// that spills a hot-word register; the review's "one ROCm upgrade away from build refused"), so that such a toolchain yields a
// correct, slower library instead of none.  Same arithmetic, same bits (tests/test_safe_build_gpu.py).
#ifndef HLX_HOT_FROM_MEMORY
#define HLX_HOT_FROM_MEMORY 0
#endif
template <int BAKE, typename T, int OFF> DEV T hot_get(uint32_t w0, uint32_t w1, const KParams* P) {
    if constexpr (BAKE != 0 && OFF < (int)sizeof(KCfg)) {
        constexpr T v = BakedRd<T>::get(HLX_BAKED_TAB[BAKE - 1], OFF);
        return v;
    } else if constexpr (HLX_HOT_FROM_MEMORY) {
        return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(&P->hot) + OFF);
    } else return HotRd<T>::get(w0, w1, OFF);
}
// ... and the spawn / domain-randomisation constants a respawning lane needs (KCold): literals in the baked instantiations,
// scalar loads from the parameter block otherwise (a cold s_load round trip is 1-2 k cycles for the lone wave of a SIMD,
// on the path that keeps a launch open)
template <int BAKE, int OFF> DEV double cold_get(const KParams* P) {
    if constexpr (BAKE != 0) {
        constexpr double v = BakedRd<double>::get(HLX_BAKED_COLD[BAKE - 1], OFF);
        return v;
    } else return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(&P->cold) + OFF);
}
#define COLD(path) (cold_get<BAKE, (int)offsetof(KCold, path)>(P))
#define HOT(path) \
    (hot_get<BAKE, std::remove_cv_t<std::remove_reference_t<decltype(((const KHot*)nullptr)->path)>>, (int)offsetof(KHot, path)>(hotw0, hotw1, P))

// NOISE = parity-mode instantiation that can take its random draws from caller-supplied float64 buffers;
// the production instantiation (NOISE = false) contains no trace of that path.
// PERSIST = the fused-rollout instantiation (hlx_rollout): the wave keeps its 64 environments' state in registers
// over `T` consecutive steps -- state groups are read before the first and written after the last step; per
// step only the action row and the delayed ring sample come in and observation/reward/flags/ring sample go out.
// Same arithmetic, same Philox keys (clock t + k), hence bit-identical to T single-step launches.
// LATE = instantiation for small batches (<= two 64-env waves per SIMD): the Kalman groups and the delayed ring sample are
// loaded as a second batch after the Philox block instead of at kernel entry (see below; a run-time switch was tried
// and lost both ways -- the optimiser merges the two load sites).
// MODE 2 = next-episode pool fill (see "next-episode pool" below): the respawn + first observation of the NEXT episode of
// the listed environments, computed off the step's critical path and stored where the step kernel picks it up when the
// environment finishes.
// LATE = 2: the schedule for batches that give a SIMD ONE wave (<= 65 536 environments on 256 CUs): LATE = 1, and -- the wave
// having the register file to itself -- the pool entry of a lane that has just finished is requested as soon as the step knows
// it has (right behind the termination tests) and sits in registers while the observation pass runs, instead of being fetched
// when the pass is over, a memory round trip on the tail of the launch.
template <uint32_t SPEC, int MODE /*0 = step, 1 = reset-only, 2 = pool fill*/, bool NOISE, bool PERSIST = false, int LATE = 0, int BAKE = 0>
// The parity (NOISE) and reset-only (MODE 1) instantiations never run at a size where occupancy matters: they get the
// whole register file, hence no scratch spills (see tools/check_hot_words.py for why a spilled hot word is fatal).
#ifndef HLX_WAVES_PER_EU
#define HLX_WAVES_PER_EU(noise, mode, persist, late) (((noise) || (mode) != 0 || (persist) || (late) == 2) ? 1 : 2)
#endif
// waves_per_eu(2): keep the single-step kernels within 256 VGPRs so that two waves fit on a SIMD (they are there anyway: ~205).
// The fused-rollout instantiations carry the state groups in registers on top of that and do not fit: until round 3 they were
// held to 256 as well, spilled 30-odd dwords to scratch, and ran 1.5x faster for it once a batch gave a SIMD two waves (1 M
// envs: 107 -> 72 us per step).  But WHICH registers the allocator spills changes with every edit of the kernel, and when it
// picks a hot-constant register the result is wrong (hotcheck.py refuses the library): after the next-episode pool went in,
// one fused instantiation or another failed that check whatever was rearranged.  They now get the whole register file -- no
// scratch, nothing to refuse -- and give up the second wave per SIMD above 65 536 environments per GPU, which measured again
// costs nothing: 53 us per step at 1 M environments against 72 with two spilling waves.
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(HLX_WAVES_PER_EU(NOISE, MODE, PERSIST, LATE)))) void hlx_env_kernel(
    // ---- 14 dwords preloaded into SGPRs by the dispatcher: everything needed to issue the state, action and
    //      ring loads and to run the Philox block without waiting for memory
    float4* __restrict__ arena, const KParams* __restrict__ P, const float* __restrict__ actions0,
    const unsigned long long t0, const unsigned long long seed, const long long env_offset, const int n,
    const uint32_t slots,   // bits 0-3 ground-ring read slot, 4-7 ground-ring write slot, 8-11 onboard-ring write slot,
                            // 12-15 ground-ring planes (delay+1, 0 = no ring), 16-19 onboard-ring planes (0 = no ring),
                            // bit 20: some hlx_info_soa plane is wanted; 22: observation pipeline; 23: hlx_info_soa.packed is given;
                            // 24: the next-episode pool is in use; 25: MODE 2 renews the entries the blocks' masks name (else: every one)
    // ---- ordinary kernarg tail (one scalar load, issued at entry, first needed when results are stored)
    float* __restrict__ obs_out0, float* __restrict__ reward_out0, uint8_t* __restrict__ term_out0,
    uint8_t* __restrict__ trunc_out0,
    // ---- the curriculum scalars in force (environment.py:223-234, 274-351), evaluated by the host at hlx_set_global_step: plain
    //      kernel arguments since round 4 -- the reference's trainers move them after EVERY step (train_flat_ppo.py:171-177), and
    //      while they lived in the device-resident parameter block each move was a 24-byte copy command between two step launches
    //      (+3.3 us per step through a curriculum ramp, profiles/r04_pool_curriculum_ramp.txt); in the kernarg tail they cost nothing
    const float cur_radius, const float cur_cos_half_beam /* beam test on the cosine: arccos(x) > hb <=> x < cos(hb); -2 when hb >= pi */,
    const float cur_on_rel, const float cur_g_rel,
    // ---- fused rollout only: steps in this launch; output slot of the first step; output slots (step k of the
    //      rollout writes slot k mod out_slots of the [out_slots][N][...] output arrays)
    const int T, const int out_slot0, const int out_slots) {
    __shared__ __attribute__((aligned(16))) float tile[64 * HLX_OBS_DIM];
    __shared__ __attribute__((aligned(16))) PowTab s_pow;   // glibc powf tables (hlx_device.h), staged by the wave itself
    // generic variants read the feature flags at run time (one scalar load) -- except volley mode, whose K-missile loops
    // are compiled in or out: the generic variant comes with and without them (KF_DYNAMIC [| HLX_F_VOLLEY])
    const uint32_t FL = (SPEC & KF_DYNAMIC) ? ((P->hot.c.flags & ~(uint32_t)HLX_F_VOLLEY) | (SPEC & (uint32_t)HLX_F_VOLLEY)) : SPEC;
    // The four output pointers are not among the preloaded SGPRs: one scalar load, first used far below.  With its only users
    // in one later block the compiler sinks the load there, behind the Philox block, and its latency (~0.45 us) is exposed in
    // full (tools/isa_tail_load.py; profiles/r02_ab_v2dr_gust_sqrt.txt).  A second, never-taken user at the very top of the kernel
    // (before the state loads, so that everything below stays one block: placed behind them it split the entry block and cost the
    // base kernel 0.45 us) keeps the load in the entry block -- issued first, landed long before it is needed -- at the price of
    // one scalar compare.  profiles/r02_ab_kernarg_tail_load_pinned.txt: config presets 9.45 -> 8.65 us, base and v2dr unchanged.
    if (RARE(n < 0)) asm volatile("" ::"s"(obs_out0), "s"(reward_out0), "s"(term_out0), "s"(trunc_out0), "s"(cur_radius), "s"(cur_cos_half_beam), "s"(cur_on_rel), "s"(cur_g_rel));
    const int lane = threadIdx.x;
    int g_rslot = (int)(slots & 15u), g_wslot = (int)((slots >> 4) & 15u), o_wslot = (int)((slots >> 8) & 15u);
    const int g_planes = (int)((slots >> 12) & 15u), o_planes = (int)((slots >> 16) & 15u);   // preloaded: no *P needed
    // ---- next-episode pool (round 3).  The waves that restart an episode used to keep every launch open: spawn draws, float64
    // spawn trigonometry and a second trip through the observation code for one or two lanes of 64.  All of that is a function
    // of (seed, environment id, index of the episode) and of constants -- not of the clock -- so it is computed AHEAD, by a
    // MODE-2 launch for the environments that have used their entry since the last one (one launch every few dozen steps,
    // several lanes of work per wave instead of one), and a finished lane whose entry is there copies it: state groups,
    // the ring samples of the first observation and the observation row.  An entry that is not there (the environment
    // finished twice between two fills, a fused rollout, a pool that is switched off) is computed on the spot by the code
    // that has always done it -- same keys, same arithmetic, same bits (tests/test_vec_env_gpu.py).
    // Layout: behind the rings, in the arena allocation, so that every address below derives from preloaded arguments:
    //   pool [blocks][POOL_GROUPS][64] float4 | tag [blocks * 64] u32 (episode index the entry holds, 0 = none) |
    //   episode [blocks * 64] u32 (auto-resets so far) | counters int32[4] | used-entry masks [blocks] u64
    // (Step launches form these addresses inside the rare block that uses them, from an opaque copy of `n`: formed here they
    // would be loop invariants held in a dozen SGPRs across the whole kernel, and the fused-rollout instantiations have none
    // to spare -- the spill that followed evicted a hot-constant register, which hotcheck.py refuses.)
    const PoolPtrs pp2 = MODE == 2 ? pool_ptrs(arena, n, g_planes, o_planes) : PoolPtrs{};
    // Hot parameter block: two coalesced dword loads per lane now, v_readlane per constant later (hlx_kargs.h).  (The first
    // vector loads of every instantiation: hotcheck.py identifies the block's registers that way.)
    uint32_t hotw0 = HLX_HOT_FROM_MEMORY ? 0u : reinterpret_cast<const uint32_t*>(P)[lane], hotw1 = HLX_HOT_FROM_MEMORY ? 0u : reinterpret_cast<const uint32_t*>(P)[64 + lane];
    int i_ = blockIdx.x * 64 + lane;
    bool live_ = i_ < n;
    unsigned long long fill_mask = ~0ull;     // MODE 2: the lanes of this block whose entry is to be renewed
    if (MODE == 2 && (slots & (1u << 25)) != 0u) {
        // (a scalar load: the mask was written by earlier launches -- one non-returning atomic OR per wave that had a finished
        // lane; a returning one, for a compacted list, was a memory round trip on the tail of every step launch)
        fill_mask = *((const __attribute__((address_space(4))) unsigned long long*)pp2.rf_mask + blockIdx.x);
        if (fill_mask == 0ull) return;
        if (lane == 0) G(pp2.rf_mask)[blockIdx.x] = 0ull;
    }
    const int i = i_;
    const bool live = live_;
    uint32_t pool_epn = 0;     // MODE 2: the episode index this lane's entry is computed for
    float* row = tile + lane * HLX_OBS_DIM;
    bool done = false;
    int32_t* done_idx_out = nullptr;   // optional compaction output (read from the hot block inside the live section)
    float pipe_reward = 0.f;           // this step's reward, for the observation pipeline's discounted returns (bit 22 of `slots`)
    double pipe_ret_prev = 0.;

#ifdef HLX_STAMPS
    unsigned long long* stamp_base = P->stamps + (size_t)blockIdx.x * 16;   // fetched once: a per-stamp scalar load
    asm volatile("" : "+s"(stamp_base));                                     // would charge every segment ~1k cycles
#endif
    STAMP(0);
    // State loads are issued before anything else, for every lane: the arena is padded to whole 64-env blocks,
    // its pointer arrives preloaded in SGPRs, and nothing here depends on the rest of the kernel arguments.
    // arena layout: [workgroup][group][64 lanes] of 16-byte words -> one contiguous ~11 KiB chunk per wave,
    // group offsets are compile-time constants (no per-group 64-bit address arithmetic in SGPRs)
    float4* A = arena + (size_t)blockIdx.x * (N_GROUPS * 64) + lane;
    float4* const PA2 = MODE == 2 ? pp2.pool + (size_t)blockIdx.x * (POOL_GROUPS * 64) + lane : nullptr;   // pool fill: this lane's entry
    double2* AD = reinterpret_cast<double2*>(A);
    // this wave's arena block as a buffer (stores only): lane offset in voffset, group offset in soffset
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc(arena + (size_t)blockIdx.x * (N_GROUPS * 64), 0, N_GROUPS * 64 * 16, 0x00020000);
    const uint32_t lane16 = (uint32_t)lane * 16u;
#define STG(Gr, v) do { if (MODE == 2) pool_put(PA2, (Gr), (v)); else wt16(rsA, lane16, (uint32_t)(Gr) * 1024u, (v)); } while (0)
    // ... and the dword plane G_AUX (a pool entry holds none: an episode starts with total_fuel_used = 0)
#ifndef HLX_AB_NO_FU
#define HLX_AB_NO_FU 0      // timing experiments only: 1 = total_fuel_used neither loaded nor stored
#endif
#ifndef HLX_AB_AUX_LATE
#define HLX_AB_AUX_LATE 0   // A/B: 1 = the total_fuel_used dword is stored in the final burst, not with the early state groups
#endif
#ifndef HLX_INFO_W2
#define HLX_INFO_W2 0       // where hlx_info_soa.packed word 2 leaves: 0 = behind observation pass 0, complete; 1 = with words 0 and 1, and
#endif                      // its flag dword once more at the end of the kernel, with the detection bits (A/B)
#ifndef HLX_DONE_LATE
#define HLX_DONE_LATE 0     // 1 = the done list is compacted behind the observation tile's stores instead of ahead of them (A/B)
#endif
#define STAUX(v) do { if (MODE != 2 && !HLX_AB_NO_FU) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsA, (uint32_t)lane * 4u, (uint32_t)G_AUX * 1024u, HLX_ST_AUX); } while (0)
    // ------------------------------------------------------------------ first load batch, issued at entry
    // (the integrator's state groups and the action row); the Philox draws below do not depend on them and run
    // while the loads are in flight.  The Kalman groups and the ring sample follow as a second batch after Philox.
    float4 g_ipos = A[G_IPOS * 64], g_ivel = A[G_IVEL * 64], g_quat = A[G_QUAT * 64], g_w1 = A[G_W1 * 64];
    double2 g_w0 = AD[G_W0 * 64];
    // The missile's groups are released (waited for) where the missile is integrated, a whole interceptor section later --
    // unless something reads the missile earlier (LOS-frame actions, the volley's bookkeeping, a reset-only launch).
    // (with the v2 models the second batch is long enough already: 11.40 against 11.46 us, profiles/r02_ab_missile_groups_waited_late.txt)
    constexpr bool missile_late = MODE == 0 && !((SPEC & KF_DYNAMIC) != 0) && !(SPEC & (HLX_F_OBS_LOS | HLX_F_VOLLEY | HLX_F_ATMOSPHERE));
    // ... and in the small-batch schedule they are LOADED with the second batch, at its head (until round 3 that was the compiler's
    // doing -- it sank the two loads to their first use -- and an unrelated edit undid it; now it is written down)
#ifndef HLX_MISSILE_LOADS_LATE
#define HLX_MISSILE_LOADS_LATE 1
#endif
    constexpr bool missile_loads_late = HLX_MISSILE_LOADS_LATE && missile_late && LATE != 0 && !PERSIST;
    float4 g_mpos = make_float4(0.f, 0.f, 0.f, 0.f), g_mvel = g_mpos;
    if (!missile_loads_late) { g_mpos = A[G_MPOS * 64]; g_mvel = A[G_MVEL * 64]; }
    float4 g_thr = make_float4(0.f, 0.f, 0.f, 0.f), g_misc = g_thr;
    if (HAS(HLX_F_THRUST_LAG) || HAS(HLX_F_DOMAIN_RAND)) g_thr = A[G_THRUST * 64];   // .w: a domain-randomised constant
    if (HAS(HLX_F_DOMAIN_RAND)) g_misc = A[G_MISC * 64];
    // `total_fuel_used` (environment.py:204, 566, 886; info['fuel_used']): environment state since round 4 -- one dword per lane of
    // the G_AUX plane (256 B per wave), in the arena like everything else, so that it travels through hlx_get_state / hlx_set_state
    // and advances in every form of the step.  (Round 3 kept it in the CALLER's info plane: a checkpoint lost it.)
    float* const AUX = reinterpret_cast<float*>(arena + (size_t)blockIdx.x * (N_GROUPS * 64) + G_AUX * 64) + lane;
    float g_fu = 0.f;
    if (MODE == 0 && !HLX_AB_NO_FU) g_fu = *G(AUX);
    // Everything down to the construction of `hot` runs with ALL lanes enabled: the constants are fetched with
    // cross-lane reads (v_readlane), so the lanes that hold them must have executed their loads even in a partial
    // tail block; padding lanes use the last live environment's addresses and their results are discarded.
    const int ic = live ? i : (n - 1);
    // powf tables: 512 B, lane l fetches 8 bytes (in flight with the state loads), written to LDS behind the Philox block
    const bool need_pow = HAS(HLX_F_ATMOSPHERE) || HAS(HLX_F_ENH_WIND);
    unsigned long long pow_word = 0ull;
    if (need_pow) pow_word = reinterpret_cast<const unsigned long long*>(&HLX_POW_TAB)[lane];
    float4 g_kfp = make_float4(0.f, 0.f, 0.f, 0.f);
    double2 g_kf0 = make_double2(0., 0.), g_kf1 = g_kf0, g_kf2 = g_kf0;
    // next-episode pool: the environment's episode counter and the tag of its pool entry travel with the Kalman groups (8 bytes
    // per environment and step), so that a lane that finishes knows at once whether its next episode is waiting -- fetched where
    // they are needed they were a whole memory round trip on the tail of the launch
    constexpr bool POOL_PRE = MODE == 0 && !NOISE && !PERSIST;
    constexpr bool LONE = POOL_PRE && LATE == 2;     // one wave per SIMD: a finished lane's pool entry is requested early, into registers
    uint32_t pre_ep = 0, pre_tag = 0;
    // (Staging the block in LDS and letting every use be a broadcast ds_read was measured too: 190 fewer
    // instructions, 33 fewer VGPRs, but +0.65 us/step at 65 536 envs -- the lone wave of a SIMD eats each
    // ds_read's latency at the use site, while a v_readlane result is there after its issue cycles.)
    // per-step cursors (advance only in the fused rollout)
    unsigned long long t = t0;
    const float* actions = actions0;
    float* obs_out = obs_out0;
    float* reward_out = reward_out0;
    uint8_t* term_out = term_out0;
    uint8_t* trunc_out = trunc_out0;
    int oslot = out_slot0;
    if (PERSIST) {   // the Kalman groups are loop-carried state too; constants once, before the loop
        g_kfp = A[G_KFP * 64]; g_kf0 = AD[G_KF0 * 64]; g_kf1 = AD[G_KF1 * 64]; g_kf2 = AD[G_KF2 * 64];
        asm volatile("" : "+v"(hotw0), "+v"(hotw1));
        if (obs_out0) obs_out = obs_out0 + (size_t)oslot * n * HLX_OBS_DIM;
        reward_out = reward_out0 + (size_t)oslot * n; term_out = term_out0 + (size_t)oslot * n; trunc_out = trunc_out0 + (size_t)oslot * n;
    }
#pragma unroll 1
    for (int kk = 0; kk < (PERSIST ? T : 1); ++kk) {
    {
        const size_t N = (size_t)n;
        // rings live right behind the (64-padded) arena in the same allocation: addresses need only preloaded values
        float4* const gring = arena + (size_t)((n + 63) >> 6) * (N_GROUPS * 64);     // [g_delay+1][3][N]
        float4* const oring = gring + (size_t)(g_planes * GROUND_RING_WORDS16) * N;   // [o_cap][N]
        float2 a01 = make_float2(0.f, 0.f), a23 = a01, a45 = a01;
        double2 gr0 = make_double2(0., 0.);
        float4 gr1 = make_float4(0.f, 0.f, 0.f, 0.f), gr2 = gr1;
        if (MODE == 0) {
            const float2* ap = reinterpret_cast<const float2*>(actions + (size_t)ic * HLX_ACT_DIM);
            a01 = ap[0]; a23 = ap[1]; a45 = ap[2];
        }
        // Kalman groups + delayed ground-ring sample: with the rest at entry when the batch is large (several waves per
        // SIMD: more loads in flight = more HBM bandwidth), as a second batch after Philox when it is small (below)
        constexpr bool late_loads = LATE != 0;    // instantiation chosen by the host from the batch size
        if (!late_loads) {
            if (MODE == 0 && HAS(HLX_F_GROUND)) {   // unconditional when the station exists (the host always allocates >= 1 slot)
                const float4* R = gring + ((size_t)g_rslot * GROUND_RING_WORDS16) * N + ic;   // slot (t - g_delay) mod cap
                gr0 = *reinterpret_cast<const double2*>(R); gr1 = R[N]; gr2 = R[2 * N];
            }
            if (!PERSIST) { g_kfp = A[G_KFP * 64]; g_kf0 = AD[G_KF0 * 64]; g_kf1 = AD[G_KF1 * 64]; g_kf2 = AD[G_KF2 * 64]; }
            if (POOL_PRE) { const PoolPtrs pq = pool_ptrs(arena, n, g_planes, o_planes); pre_tag = G(pq.tag)[ic]; pre_ep = G(pq.ep_cur)[ic]; }
        }

        STAMP(1);   // all loads issued
        // ... and nothing of the Philox block below may be scheduled in front of them.  (Round 3: an edit far below -- the
        // info['fuel_used'] accumulator -- made the machine scheduler hoist the first two Philox rounds, ~60 instructions, above the
        // state loads of the base kernel: the loads left ~350 cycles later and every form of the step lost 0.5 us, old
        // against new library on one box, profiles/r03_ab_loads_before_philox.txt.)
#ifndef HLX_NO_LOAD_SCHED_BARRIER
        __builtin_amdgcn_sched_barrier(0);
#endif
        const bool noise_buf = NOISE && P->hot.opt.step_noise != nullptr;    // parity instantiation only
        const bool rnoise_buf = NOISE && P->hot.opt.reset_noise != nullptr;
        const unsigned long long gid = (unsigned long long)(env_offset + i);
        Rng rng{uint2{(uint32_t)seed, (uint32_t)(seed >> 32)}, (uint32_t)gid, (uint32_t)(gid >> 32), (uint32_t)t, (uint32_t)(t >> 32)};
        const double* SN = NOISE ? P->hot.opt.step_noise + ic : nullptr;   // slot-major [slot][N] (float64: parity mode replays the reference's draws)
        const double* RN = NOISE ? P->hot.opt.reset_noise + ic : nullptr;

        // ------------------------------------------------------------------ this step's random draws
        // (five independent Philox chains in one straight-line block: instruction-level parallelism for
        // the lone wave of this SIMD, and it overlaps the state loads issued above)
        D3 z_ev = d3(0., 0., 0.), z_wind = z_ev, z_gp = z_ev, z_gv = z_ev;
        D3 z_evk[HLX_MAX_VOLLEY - 1] = {z_ev, z_ev, z_ev};   // volley missiles 1..3 (generic instantiation only)
        float u_on = 0.f, u_g = 0.f;
        double u_dl = 0., u_gust = 1.;
        if (MODE == 0) {
            if (noise_buf) {
                z_ev = d3(SN[0 * N], SN[1 * N], SN[2 * N]); z_wind = d3(SN[3 * N], SN[4 * N], SN[5 * N]); u_gust = SN[6 * N];
                u_on = (float)SN[11 * N]; u_g = (float)SN[12 * N];
                z_gp = d3(SN[13 * N], SN[14 * N], SN[15 * N]); z_gv = d3(SN[16 * N], SN[17 * N], SN[18 * N]); u_dl = SN[19 * N];
            } else {
                // 4 uniforms + 12 normals = four Philox calls (all of them, whatever the feature flags: the block
                // runs in the shadow of the cold-start memory latency, and the slot -> stream map stays fixed)
                const uint4 x = rng.raw(RS_STEP_U);
                float n0, n1, n2, n3, n4, n5, n6, n7, n8, n9, n10, n11;
                rng.normals4(RS_STEP_N0, n0, n1, n2, n3);
                rng.normals4(RS_STEP_N1, n4, n5, n6, n7);
                rng.normals4(RS_STEP_N2, n8, n9, n10, n11);
                float ua = u01(x.x), ub = u01(x.y), uc = u01(x.z), ud = u01(x.w);
                // keep the block here (the optimiser would otherwise sink each chain to its first use, behind
                // the loads it is meant to overlap)
#ifndef HLX_PHILOX_PIN
#define HLX_PHILOX_PIN 3    // which draws are pinned in front of the first use of loaded state: 3 = all sixteen (rounds 1-3); 2 = not the
#endif                      // ground-measurement normals of call N2 (first needed in the observation section); 1 = nor the four uniforms (A/B)
                asm volatile("" : "+v"(n0), "+v"(n1), "+v"(n2), "+v"(n3), "+v"(n4), "+v"(n5), "+v"(n6), "+v"(n7));
                if (HLX_PHILOX_PIN >= 3) asm volatile("" : "+v"(n8), "+v"(n9), "+v"(n10), "+v"(n11));
                if (HLX_PHILOX_PIN >= 2) asm volatile("" : "+v"(ua), "+v"(ub), "+v"(uc), "+v"(ud));
                z_ev = d3((double)n0, (double)n1, (double)n2);
                z_wind = d3((double)n3, (double)n4, (double)n5);
                z_gp = d3((double)n6, (double)n7, (double)n8);
                z_gv = d3((double)n9, (double)n10, (double)n11);
                u_on = ua; u_g = ub; u_dl = (double)uc; u_gust = (double)ud;
            }
            if (HAS(HLX_F_VOLLEY)) {   // evasion draws of the other missiles of the volley
#pragma unroll
                for (int k = 1; k < HLX_MAX_VOLLEY; ++k) {
                    if (noise_buf) z_evk[k - 1] = d3(SN[(20 + 3 * (k - 1)) * N], SN[(21 + 3 * (k - 1)) * N], SN[(22 + 3 * (k - 1)) * N]);
                    else {
                        float e0, e1, e2, e3;
                        rng.normals4(RS_STEP_V1 + (uint32_t)(k - 1), e0, e1, e2, e3);
                        z_evk[k - 1] = d3((double)e0, (double)e1, (double)e2);
                    }
                }
            }
        }

        // Loaded registers become visible to the optimiser only here: without these fake defs it hoists the first
        // cheap use of a loaded value (a flag compare) above the Philox block, and the wave would sit out the
        // cold-start latency BEFORE doing the one piece of work that needs no memory.
#define PIN4(v) asm volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z), "+v"((v).w))
#define PIN2(v) asm volatile("" : "+v"((v).x), "+v"((v).y))
        if (late_loads) {
            // Second load batch: the Kalman groups and the delayed ground-ring sample are first needed ~5k cycles from
            // here.  At one wave per SIMD the load phase is bandwidth-bound and exposed: issued at kernel entry with
            // everything else they share the cold-start bandwidth with the integrator's groups (the whole chip asks
            // for 15.7 MB in the same microsecond) and delay the physics (+0.55 us at 65 536 envs); issued here they
            // move while the physics runs.  `late` is an opaque zero that pins them below Philox.
            uint32_t late = 0;
            asm volatile("" : "+v"(late));
            if (missile_loads_late) { g_mpos = A[G_MPOS * 64 + late]; g_mvel = A[G_MVEL * 64 + late]; }
            if (MODE == 0 && HAS(HLX_F_GROUND)) {   // unconditional when the station exists (the host always allocates >= 1 slot)
                const float4* R = gring + ((size_t)g_rslot * GROUND_RING_WORDS16) * N + ic + late;   // slot (t - g_delay) mod cap
                gr0 = *reinterpret_cast<const double2*>(R); gr1 = R[N]; gr2 = R[2 * N];
            }
            if (!PERSIST) { g_kfp = A[G_KFP * 64 + late]; g_kf0 = AD[G_KF0 * 64 + late]; g_kf1 = AD[G_KF1 * 64 + late]; g_kf2 = AD[G_KF2 * 64 + late]; }
            if (POOL_PRE) { const PoolPtrs pq = pool_ptrs(arena, n, g_planes, o_planes); pre_tag = G(pq.tag)[ic + late]; pre_ep = G(pq.ep_cur)[ic + late]; }
        }
        PIN4(g_ipos); PIN4(g_ivel); PIN4(g_quat); PIN4(g_w1); PIN2(g_w0);
        if (!missile_late) { PIN4(g_mpos); PIN4(g_mvel); }
        if (HAS(HLX_F_THRUST_LAG) || HAS(HLX_F_DOMAIN_RAND)) PIN4(g_thr);
        if (HAS(HLX_F_DOMAIN_RAND)) PIN4(g_misc);
        if (MODE == 0) asm volatile("" : "+v"(g_fu));
        PIN2(a01); PIN2(a23); PIN2(a45);
        // (the Kalman / ring registers are released further down, right before the observation section)
        // the output pointers of the kernarg tail have landed by now
        asm volatile("" ::"s"(obs_out), "s"(reward_out), "s"(term_out), "s"(trunc_out));
        if (!PERSIST) asm volatile("" : "+v"(hotw0), "+v"(hotw1));
        if (need_pow && kk == 0) {
            reinterpret_cast<unsigned long long*>(&s_pow)[lane] = pow_word;
            WAVE_LDS_SYNC();   // one wave per workgroup: orders the LDS writes before the per-lane lookups below
        }
        done_idx_out = HOT(opt.done_idx);
        // (the same for the observation pipeline's discounted returns, when one rides on this launch)
        if (MODE == 0 && !PERSIST && RARE(slots & (1u << 22)) && HOT(opt.pipe_returns)) pipe_ret_prev = G(HOT(opt.pipe_returns))[ic];
        if (live) {   // ============================== per-environment work, live lanes only ==============================
        STAMP(2);   // Philox block done (loads still in flight)
        V3 ipos = v3(g_ipos.x, g_ipos.y, g_ipos.z), ivel = v3(g_ivel.x, g_ivel.y, g_ivel.z);
        Quat q = Quat{g_quat.x, g_quat.y, g_quat.z, g_quat.w};
        V3 mpos = v3(0.f, 0.f, 0.f), mvel = mpos;
        if (!missile_late) { mpos = v3(g_mpos.x, g_mpos.y, g_mpos.z); mvel = v3(g_mvel.x, g_mvel.y, g_mvel.z); }
        D3 wind = d3(g_w0.x, g_w0.y, __hiloint2double(__float_as_int(g_w1.y), __float_as_int(g_w1.x)));
        float fuel = g_ipos.w, prev_distance = g_ivel.w, min_distance = 0.f, last_distance = 0.f;
        float fuel_used = g_fu;                                                      // environment.py:204 total_fuel_used
        if (!missile_late) { min_distance = g_mpos.w; last_distance = g_mvel.w; }
        uint32_t packed = __float_as_uint(g_w1.z);
        float ep_return = g_w1.w;
        int steps = (int)(packed & 0x1FFFu), worsening = (int)((packed >> 13) & 0xFFFu);
        bool crossed = (packed >> 25) & 1u, kf_init = (packed >> 26) & 1u, kf_x64 = (packed >> 27) & 1u;
        int on_delay = (int)(packed >> 28);
        V3 thrust_act = v3(g_thr.x, g_thr.y, g_thr.z);
        // Per-episode constants of domain randomisation.  The reference keeps them as Python floats (float64): T0 accumulates
        // over episodes in float64 and is rounded to float32 where it meets the float32 altitude; base_cd and the peak
        // multiplier enter float32 expressions as F(base_cd), F(peak - 1.0) and F(base_cd * supersonic_multiplier).
        double T0 = 288.15;
        DragParams dp{HOT(c.subsonic), HOT(c.supersonic), HOT(c.mach_span), HOT(c.peak_m1), 0.3f, HOT(c.cd_super)};
        if (HAS(HLX_F_DOMAIN_RAND)) {
            T0 = __hiloint2double(__float_as_int(g_misc.y), __float_as_int(g_misc.x));
            dp.base_cd = g_misc.z; dp.peak_m1 = g_misc.w; dp.cd_super = g_thr.w;
        }

        // volley mode (environment.py:44): every missile of the volley; mpos / mvel above are `self.missile_state`,
        // the entry `prio` of this list.  These groups always travel through the arena (also in the fused rollout).
        V3 vp[HLX_MAX_VOLLEY], vv[HLX_MAX_VOLLEY];
        float vmin[HLX_MAX_VOLLEY];
        bool vact[HLX_MAX_VOLLEY];
        int prio = 0, n_int = 0;
        const int VK = HAS(HLX_F_VOLLEY) ? HOT(c.volley_k) : 0;
#pragma unroll
        for (int k = 0; k < HLX_MAX_VOLLEY; ++k) {
            vp[k] = v3(0.f, 0.f, 0.f); vv[k] = vp[k]; vmin[k] = 0.f; vact[k] = false;
            if (HAS(HLX_F_VOLLEY)) {
                const float4 a = A[(G_VPOS + k) * 64], b = A[(G_VVEL + k) * 64];
                const uint32_t w = __float_as_uint(b.w);
                vp[k] = v3(a.x, a.y, a.z); vmin[k] = a.w; vv[k] = v3(b.x, b.y, b.z); vact[k] = (w & 1u) != 0u;
                if (k == 0) { prio = (int)((w >> 8) & 3u); n_int = (int)((w >> 12) & 7u); }
            }
        }

        float reward = 0.f, distance = 0.f, range_c = 0.f;   // range_c: ||missile - interceptor|| of the state the observation will see
        bool terminated = false, truncated = false, intercepted = false, hit_target = false, fuze = false,
             clamped = false;

        if (MODE == 0) {
            // -------------------------------------------------------------- action (environment.py:605-625)
            V3 at = v3(a01.x, a01.y, a23.x), aw = v3(a23.y, a45.x, a45.y);
            const float los_a0 = at.x;
            steps += 1;                                                             // :607
            if (HAS(HLX_F_OBS_LOS)) {                                               // :618-620, :965-1063
                V3 rel0 = mpos - ipos;
                float rg = snorm3(rel0);
                V3 lu = (rg > 1e-6f) ? rel0 / rg : v3(1.f, 0.f, 0.f);
                V3 h, v;
                los_basis(lu, h, v);
                at = (lu * at.x + h * at.y) + v * at.z;                             // :1052-1056
            }
            if (RARE(fuel <= 0.f)) { at = at * 0.f; clamped = true; }                      // core.py:1080-1083
            // core.py:1086-1098: ||a|| <= sqrt(3) max|a_i|, so the exact norms (numpy's float32 norm: ~20 instructions each)
            // are formed only when a component is large enough for a limit to be within reach (28 sqrt(3) < 50, 2.8 sqrt(3) < 5);
            // with actions in [-1, 1] that is never
            if (RARE(fmaxf(fmaxf(fabsf(at.x), fabsf(at.y)), fabsf(at.z)) > 28.f)) {
                float am = snorm3(at);
                if (am > 50.f) { at = at * HLX_DIVF(50.f, am); clamped = true; }                  // core.py:1086-1090
            }
            if (RARE(fmaxf(fmaxf(fabsf(aw.x), fabsf(aw.y)), fabsf(aw.z)) > 2.8f)) {
                float gm = snorm3(aw);
                if (gm > 5.f) { aw = aw * HLX_DIVF(5.f, gm); clamped = true; }                    // core.py:1094-1098
            }

            // Is the reference's wind a float64 array at this point?  Simple wind: float32 copy of
            // base_wind after reset, float64 from the first update on (environment.py:542,1127-1129).
            const bool simple_wind = !HAS(HLX_F_ENH_WIND) && HOT(c.wind_var) > 0.0;
            const bool w64 = simple_wind && steps > 1;

            STAMP(3);   // first use of loaded state + clamp done
            // -------------------------------------------------------------- interceptor (environment.py:861-956)
            V3 thr = at * 10000.f, ang = aw * 20.f;                                 // :870-871
            if (HAS(HLX_F_THRUST_LAG)) {                                            // :874-878
                V3 err = thr - thrust_act;
                thrust_act = thrust_act + divc(err * HOT(c.dt), HOT(c.inv_tau));
                thr = thrust_act;
            }
            float fc = (divc(snorm3(thr), 1.0 / 500.0) * 0.1f) * HOT(c.dt);              // :883-884
            fuel = fuel - fc;
            fuel_used = fuel_used + fc;                                              // :886 (float32 running sum, 0 at reset :566)
            if (RARE(fuel <= 0.f)) { fuel = 0.f; thr = v3(0.f, 0.f, 0.f); thrust_act = thr; } // :888-892
            const V3 tacc = divc(thr, 1.0 / 500.0);                                 // :896
            float rho = 1.225f, sos = 343.f;
            if (HAS(HLX_F_ATMOSPHERE)) atmosphere(fmaxf(ipos.z, 0.f), (float)T0, rho, sos, &s_pow); // :899-906
            const float GRAV = -9.81f;
            // Common case first and unconditionally, the rare one as an override behind RARE(): an if/else keeps both bodies
            // in line, and the lone wave of a SIMD pays an instruction-buffer refill for every branch TAKEN -- here the jump
            // over the body that no lane needs.  (The float32 form serves the first step of an episode only.)
            V3 ivel_n = ivel;
            if (simple_wind) {                                                      // float64 air-relative velocity (never with the enhanced wind model)
                D3 va = to_d3(ivel) - wind;                                         // :910
                D3 dacc;
                if (HAS(HLX_F_MACH_DRAG) && dnorm(va) > 1e-6) dacc = mach_drag_force64(va, rho, sos, 1.0, dp) * (1.0 / 500.0);
                else {                                                              // :920-921
                    double c1 = HAS(HLX_F_ATMOSPHERE) ? (double)(-0.15f * rho) : (-0.5 * 0.3 * 1.225);
                    double c2 = c1 * dnorm(va);
                    // (c2*v)/mass in float64: the reciprocal form differs by <= 1 ulp(f64), invisible after the
                    // float32 rounding of the velocity except with probability ~1e-9
                    dacc = d3((c2 * va.x) * (1.0 / 500.0), (c2 * va.y) * (1.0 / 500.0), (c2 * va.z) * (1.0 / 500.0));
                }
                D3 acc = d3((double)tacc.x + dacc.x, (double)tacc.y + dacc.y, ((double)tacc.z + dacc.z) + (double)GRAV); // :924
                ivel_n = v3((float)((double)ivel.x + acc.x * HOT(c.dt64)), (float)((double)ivel.y + acc.y * HOT(c.dt64)),
                            (float)((double)ivel.z + acc.z * HOT(c.dt64)));              // :933
                if (HAS(HLX_F_VALIDATION) && RARE(nan_hit(acc))) {                  // :927-930, out of line: repair, then the same update
                    acc = nan_guard(acc, 50.0);
                    ivel_n = v3((float)((double)ivel.x + acc.x * HOT(c.dt64)), (float)((double)ivel.y + acc.y * HOT(c.dt64)),
                                (float)((double)ivel.z + acc.z * HOT(c.dt64)));
                    asm volatile("" : "+v"(ivel_n.x), "+v"(ivel_n.y), "+v"(ivel_n.z));   // (keeps the optimiser from merging the two updates back into one behind an if/else on `acc`)
                }
            }
            if (RARE(!w64)) {
                V3 va = ivel - to_v3(wind);
                V3 dacc = v3(0.f, 0.f, 0.f);
                if (HAS(HLX_F_MACH_DRAG)) dacc = divc(mach_drag_force(va, rho, sos, 1.0f, dp), 1.0 / 500.0);
                if (!HAS(HLX_F_MACH_DRAG) || RARE(!(snorm3(va) > 1e-6f))) {         // :920 simple drag (Mach model off, or no air speed)
                    float c1 = HAS(HLX_F_ATMOSPHERE) ? (-0.15f * rho) : (float)(-0.5 * 0.3 * 1.225);
                    float c2 = c1 * snorm3(va);
                    dacc = divc(va * c2, 1.0 / 500.0);
                }
                V3 acc = v3(tacc.x + dacc.x, tacc.y + dacc.y, (tacc.z + dacc.z) + GRAV);
                if (HAS(HLX_F_VALIDATION) && RARE(nan_hit(to_d3(acc)))) acc = to_v3(nan_guard(to_d3(acc), 50.0));
                ivel_n = ivel + acc * HOT(c.dt);
            }
            ivel = ivel_n;
            ipos = ipos + ivel * HOT(c.dt);                                              // :934
            {
                float wn = snorm3(ang);                                             // :940-956
                float angle = wn * HOT(c.dt);
                if (angle > 1e-6f) {
                    float half = angle * 0.5f;                                      // angle / 2 (exact)
                    float ch = (float)cos_small((double)half), sh = (float)sin_small((double)half);
                    Quat r = quat_mul(Quat{ch, HLX_DIVF(ang.x, wn) * sh, HLX_DIVF(ang.y, wn) * sh, HLX_DIVF(ang.z, wn) * sh}, q);
                    float nq = HLX_SQRTF((float)((((double)(r.w * r.w) + (double)(r.x * r.x)) + (double)(r.y * r.y)) + (double)(r.z * r.z)));
                    q = Quat{HLX_DIVF(r.w, nq), HLX_DIVF(r.x, nq), HLX_DIVF(r.y, nq), HLX_DIVF(r.z, nq)};
                }
            }
            STAMP(4);   // interceptor integrated
            // -------------------------------------------------------------- missile (environment.py:1069-1117)
            auto missile_step = [&](V3& mpos, V3& mvel, const D3& z_ev) {
                float mrho = 1.225f, msos = 343.f;
                if (HAS(HLX_F_ATMOSPHERE)) atmosphere(fmaxf(mpos.z, 0.f), (float)T0, mrho, msos, &s_pow);
                D3 sum = d3(0., 0., 0.);                                            // drag + gravity, before evasion
                if (simple_wind) {
                    D3 va = to_d3(mvel) - wind;
                    D3 md;
                    if (HAS(HLX_F_MACH_DRAG) && dnorm(va) > 1e-6) md = mach_drag_force64(va, mrho, msos, 2.0, dp) * ((0.3 * 1.5) / 0.3) * (1.0 / 1000.0);
                    else {
                        double c1 = HAS(HLX_F_ATMOSPHERE) ? (double)(-0.15f * mrho) : (-0.5 * 0.3 * 1.225);
                        double c2 = c1 * dnorm(va);
                        md = d3((c2 * va.x) * (1.0 / 1000.0), (c2 * va.y) * (1.0 / 1000.0), (c2 * va.z) * (1.0 / 1000.0));
                    }
                    sum = d3(md.x, md.y, md.z + (double)GRAV);
                }
                if (RARE(!w64)) {
                    V3 va = mvel - to_v3(wind);
                    V3 md = v3(0.f, 0.f, 0.f);
                    if (HAS(HLX_F_MACH_DRAG)) md = divc(mach_drag_force(va, mrho, msos, 2.0f, dp) * 1.5f, 1.0 / 1000.0);   // :1087-1096
                    if (!HAS(HLX_F_MACH_DRAG) || RARE(!(snorm3(va) > 1e-6f))) {
                        float c1 = HAS(HLX_F_ATMOSPHERE) ? (-0.15f * mrho) : (float)(-0.5 * 0.3 * 1.225);
                        float c2 = c1 * snorm3(va);
                        md = divc(va * c2, 1.0 / 1000.0);
                    }
                    sum = to_d3(v3(md.x, md.y, md.z + GRAV));
                }
                if (HAS(HLX_F_EVASION))                                             // :1103-1108 (float64)
                    sum = d3(sum.x + z_ev.x * 2.0, sum.y + z_ev.y * 2.0, sum.z + z_ev.z * 2.0);
                V3 mvel_n = v3((float)((double)mvel.x + sum.x * HOT(c.dt64)), (float)((double)mvel.y + sum.y * HOT(c.dt64)),
                               (float)((double)mvel.z + sum.z * HOT(c.dt64)));           // :1116
                asm volatile("" : "+v"(mvel_n.x), "+v"(mvel_n.y), "+v"(mvel_n.z));   // (pinned here: otherwise sunk into an else branch, in line)
                if (HAS(HLX_F_VALIDATION) && RARE(nan_hit(sum))) {                  // :1111-1113, out of line
                    sum = nan_guard(sum, 20.0);
                    mvel_n = v3((float)((double)mvel.x + sum.x * HOT(c.dt64)), (float)((double)mvel.y + sum.y * HOT(c.dt64)),
                                (float)((double)mvel.z + sum.z * HOT(c.dt64)));
                    asm volatile("" : "+v"(mvel_n.x), "+v"(mvel_n.y), "+v"(mvel_n.z));
                }
                mvel = mvel_n;
                mpos = mpos + mvel * HOT(c.dt);                                          // :1117
            };
            if (missile_late) {   // the missile's two groups: waited for here (they were loaded last)
                PIN4(g_mpos); PIN4(g_mvel);
                mpos = v3(g_mpos.x, g_mpos.y, g_mpos.z); mvel = v3(g_mvel.x, g_mvel.y, g_mvel.z);
                min_distance = g_mpos.w; last_distance = g_mvel.w;
            }
            if (HAS(HLX_F_VOLLEY)) {                                                // :631-636: every ACTIVE missile
#pragma unroll
                for (int k = 0; k < HLX_MAX_VOLLEY; ++k)
                    if (k < VK && vact[k]) missile_step(vp[k], vv[k], k == 0 ? z_ev : z_evk[k > 0 ? k - 1 : 0]);
            } else missile_step(mpos, mvel, z_ev);
            STAMP(5);   // missile integrated
            // -------------------------------------------------------------- wind (environment.py:1119-1129)
            if (HAS(HLX_F_ENH_WIND)) {                                              // physics_models.py:351-387
                float walt = fmaxf(ipos.z, 0.f);
                // boundary-layer formulas for every lane (the power law of a lane at or below 10 m is computed and discarded);
                // the surface layer as selects, the free atmosphere above the boundary layer out of line
                float prof = pow_ref(divc(fmaxf(walt, 10.f), 1.0 / 10.0), 0.143f, &s_pow);    // :319-324
                float ti = HOT(c.ti_mid) * (1.0f - (walt / HOT(c.bl_height)) * 0.7f);         // :343-346
                if (walt <= 10.f) { prof = 1.0f; ti = HOT(c.ti_low); }
                if (RARE(walt > HOT(c.bl_height))) { prof = HOT(c.bl_prof); ti = HOT(c.ti_high); }
                V3 w = v3(HOT(c.base_wind[0]) * prof, HOT(c.base_wind[1]) * prof, HOT(c.base_wind[2]) * prof);
                const D3 z = z_wind;
                const double gu = u_gust;
                if (ti > 0.f) {                                                     // :370-378
                    double scale = (double)(ti * snorm3(w));
                    w = v3((float)((double)w.x + (scale * z.x) * HOT(c.turb_lp)), (float)((double)w.y + (scale * z.y) * HOT(c.turb_lp)),
                           (float)((double)w.z + (scale * z.z) * HOT(c.turb_lp)));
                }
                if (RARE(gu < 0.001)) {                                                   // :381-385
                    D3 gd; double e;
                    if (noise_buf) { gd = d3(SN[7 * N], SN[8 * N], SN[9 * N]); e = SN[10 * N]; }
                    else { V3 g; float ef; gust_draws(rng, g, ef); gd = to_d3(g); e = (double)ef; }
                    // (the compiler's sqrt here, not dnorm's short sequence: with the short one in this rare block the v2dr kernel ran
                    // 0.55 us slower per launch -- profiles/r02_ab_v2dr_gust_sqrt.txt)
                    double gn = sqrt(ddot(gd, gd)) + 1e-6, gmag = HOT(c.gust_scale) * e;
                    w = v3((float)((double)w.x + (gd.x / gn) * gmag), (float)((double)w.y + (gd.y / gn) * gmag),
                           (float)((double)w.z + (gd.z / gn) * gmag));
                }
                wind = to_d3(w);
            } else if (simple_wind) {                                               // :1127-1129
                const D3 z = z_wind;
                auto upd = [&](double w, float base, double zz) {
                    double t1 = w64 ? 0.95 * w : (double)(0.95f * (float)w);
                    return t1 + 0.05 * ((double)base + zz * HOT(c.wind_var));
                };
                wind = d3(upd(wind.x, HOT(c.base_wind[0]), z.x), upd(wind.y, HOT(c.base_wind[1]), z.y), upd(wind.z, HOT(c.base_wind[2]), z.z));
            }
            // -------------------------------------------------------------- volley: priority missile (:236-267, :643-650)
            float vd[HLX_MAX_VOLLEY] = {0.f, 0.f, 0.f, 0.f};
            if (HAS(HLX_F_VOLLEY)) {   // closest ACTIVE missile, the first wins ties, missile 0 when none is active
                int sel = -1;
                float best = 0.f;
#pragma unroll
                for (int k = 0; k < HLX_MAX_VOLLEY; ++k) {
                    if (k < VK) {
                        vd[k] = snorm3(vp[k] - ipos);
                        if (vact[k] && (sel < 0 || vd[k] < best)) { sel = k; best = vd[k]; }
                    }
                }
                prio = sel < 0 ? 0 : sel;
                mpos = vp[0]; mvel = vv[0];
#pragma unroll
                for (int k = 1; k < HLX_MAX_VOLLEY; ++k) if (prio == k) { mpos = vp[k]; mvel = vv[k]; }
            }
            // -------------------------------------------------------------- intercept / termination (:657-814)
            V3 rel = mpos - ipos;
            distance = snorm3(rel);
            range_c = distance;
            if (HAS(HLX_F_VOLLEY)) {                                                // :661-692
                const float thr = HAS(HLX_F_PROX_FUZE) ? HOT(c.kill_radius) : cur_radius;
                intercepted = false;
#pragma unroll
                for (int k = 0; k < HLX_MAX_VOLLEY; ++k) {
                    if (k < VK && vact[k]) {
                        vmin[k] = (vd[k] < vmin[k]) ? vd[k] : vmin[k];
                        if (vd[k] < thr) { intercepted = true; n_int += 1; vact[k] = false; }
                    }
                }
                bool any = false;
                distance = 0.f;                                                     // :691: 0.0 once nothing is left
#pragma unroll
                for (int k = 0; k < HLX_MAX_VOLLEY; ++k)
                    if (k < VK && vact[k] && (!any || vd[k] < distance)) { distance = vd[k]; any = true; }
            } else if (HAS(HLX_F_PROX_FUZE)) intercepted = distance < HOT(c.kill_radius); // :700-703
            else intercepted = distance < cur_radius;
            min_distance = (distance < min_distance) ? distance : min_distance;     // :706
            if (intercepted) crossed = true;                                        // :709-710
            if (HAS(HLX_F_PROX_FUZE) && min_distance < HOT(c.kill_radius)) { fuze = true; intercepted = true; } // :715-717
            const bool ground = mpos.z <= 0.f;
            const float gdx = mpos.x - HOT(c.target[0]), gdy = mpos.y - HOT(c.target[1]);
            bool near_target = false;                                               // only a missile on the ground asks (rare)
            if (HAS(HLX_F_VOLLEY) || RARE(ground)) near_target = HLX_SQRTF((float)((double)(gdx * gdx) + (double)(gdy * gdy))) < 500.f;
            if (HAS(HLX_F_VOLLEY)) {                                                // :724-748
                bool all_inactive = true;
#pragma unroll
                for (int k = 0; k < HLX_MAX_VOLLEY; ++k) {
                    if (k < VK) {
                        if (vp[k].z <= 0.f) {                                       // on the ground: neutralised
                            vact[k] = false;
                            const float gx = vp[k].x - HOT(c.target[0]), gy = vp[k].y - HOT(c.target[1]);
                            if (HLX_SQRTF((float)((double)(gx * gx) + (double)(gy * gy))) < 500.f) hit_target = true;
                        }
                        if (vact[k]) all_inactive = false;
                    }
                }
                if (all_inactive || fuze) terminated = true;
            } else if (HAS(HLX_F_PRECISION)) {                                      // :752-767
                if (ground) { terminated = true; hit_target = near_target; }
            } else {                                                                // :769-786 (selects, not branches)
                terminated = intercepted || ground;
                hit_target = !intercepted && ground && near_target;
            }
            {                                                                       // :789-811, the elif chain as selects
                const bool crash = ipos.z < 0.f, dry = fuel <= 0.f;
                const bool late = !crash && !dry && steps > 1000;
                const int w_next = (distance > last_distance) ? min(worsening + 1, 0xFFF) : max(0, worsening - 5);
                worsening = late ? w_next : worsening;
                last_distance = late ? distance : last_distance;
                terminated = terminated || crash || dry || (late && worsening > 500 && distance > 2500.f);
            }
            truncated = steps >= HOT(c.max_steps);                                       // :813-814
            STAMP(6);   // wind + termination
            // -------------------------------------------------------------- reward (:1131-1320)
            if (HAS(HLX_F_PRECISION)) {
                if (terminated) {                                                   // :1155-1201
                    float md = min_distance;
                    if (crossed) {
                        reward = 3000.f;
                        if (md < cur_radius) reward = reward + HLX_DIVF(cur_radius - md, cur_radius) * 1000.f;
                        reward = reward + exp_np(divc(-md, 1.0 / 25.0)) * 500.f;
                        reward = reward + exp_np(divc(-md, 1.0 / 10.0)) * 1000.f;
                        reward = reward + exp_np(divc(-md, 1.0 / 3.0)) * 500.f;
                        reward = reward + (float)((double)(HOT(c.max_steps) - steps) * 0.3);
                    } else {
                        reward = fmaxf(-md * 0.5f, -2000.f);
                        if (hit_target) reward -= 1000.f;
                        else if (ipos.z < 0.f) reward -= 500.f;
                        else if (fuel <= 0.f) reward -= 300.f;
                    }
                } else {                                                            // :1203-1271
                    float delta = prev_distance - distance;
                    float cv = divc(delta, HOT(c.inv_dtf));
                    reward = clampf(divc(cv, 1.0 / 100.0), -0.5f, 2.0f) * 0.5f;
                    if (distance < 50.f) { reward = reward + delta * 5.f; reward = reward + exp_np(divc(-distance, 1.0 / 10.0)) * 1.0f; }
                    else if (distance < 150.f) reward = reward + delta * 3.f;
                    else if (distance < 500.f) reward = reward + delta * 1.5f;
                    else reward = reward + delta * 0.8f;
                    float isp = snorm3(ivel);
                    if (isp > 1.0f && distance > 10.f) reward = reward + sdot3(ivel / isp, rel / distance) * 0.3f; // :1241-1250
                    if (HAS(HLX_F_OBS_LOS)) reward = reward + los_a0 * 0.4f;         // :1256-1264
                    reward = reward - 0.2f;
                    prev_distance = distance;
                }
            } else {
                // per-step shaping (:1298-1320) for every lane; the terminal forms (:1274-1296) override it for the few that end
                float delta = prev_distance - distance;
                float cv = divc(delta, HOT(c.inv_dtf));
                reward = clampf(divc(cv, 1.0 / 100.0), -0.5f, 2.0f) * 0.3f;
                reward = reward + delta * ((distance < 200.f) ? 2.0f : (distance < 500.f) ? 1.0f : 0.5f);
                reward = reward - 0.5f;
                const bool ending = intercepted || terminated;
                if (RARE(ending)) {
                    float rt = fmaxf(-distance * 0.5f, -2000.f);                    // :1284-1296
                    rt -= hit_target ? 1000.f : (ipos.z < 0.f) ? 500.f : (fuel <= 0.f) ? 300.f : 0.f;
                    reward = intercepted ? (float)(5000.0 + (double)(HOT(c.max_steps) - steps) * 0.5) : rt;   // :1274-1282
                }
                prev_distance = ending ? prev_distance : distance;
            }
            done = terminated || truncated;
            ep_return += reward;
            pipe_reward = reward;
        } else {
            // reset-only launch: `done` marks the envs to reset
            done = MODE == 2 ? ((fill_mask >> lane) & 1ull) != 0ull      // a pool fill: the environments that used their entry
                             : (HOT(opt.reset_mask) ? (G(HOT(opt.reset_mask))[i] != 0) : true);
        }

        STAMP(7);   // reward
        // Kalman / ring registers become visible here (their loads were issued last, right behind the Philox block).  BEFORE
        // the scalar output stores below: vmcnt counts loads and stores in one in-order queue on this part, so a wait for
        // these loads placed behind the stores also waits for the stores' acknowledgements -- a store latency stall that a
        // lone wave eats in full (the late loads themselves landed thousands of cycles ago).
        PIN4(g_kfp); PIN2(g_kf0); PIN2(g_kf1); PIN2(g_kf2); PIN2(gr0); PIN4(gr1); PIN4(gr2);
        if (POOL_PRE) asm volatile("" : "+v"(pre_ep), "+v"(pre_tag));
        // One observation pass or two?  (see the observation section below)
        // ... but with the pool at hand (lone-wave schedule) the two-pass flow has no second pass left for a lane whose entry is
        // there -- it observes its terminal state with everybody else and copies the rest -- and is the faster one even when nobody
        // wants the terminal observation: 7.87 against 8.29 us (profiles/r03_ab_early_state_stores.txt).  -DHLX_LONE_NEVER_SINGLE=0: A/B.
#ifndef HLX_LONE_NEVER_SINGLE
#define HLX_LONE_NEVER_SINGLE 1
#endif
        // (A compile-time fact of the lone-wave instantiations -- whether or not the pool is switched on: without the single-pass
        // path in them at all they are 0.15-0.2 us faster in every form, profiles/r03_ab_early_state_stores.txt call 10.)
        const bool single = (HLX_LONE_NEVER_SINGLE && LONE) ? false : (MODE == 0 && HOT(opt.terminal_obs) == nullptr && !(slots & (1u << 20)));
        // lone-wave schedule: the prepared episode of a lane that has just finished is requested here -- ahead of the output stores
        // below, so that waiting for it later does not wait for them -- and is unpacked when the observation pass is over
        // (`pf` is deliberately left without an initial value: it is written and read under `hit_pf` only, and a zero on the
        // other path makes the compiler merge the two with copies behind the loads -- i.e. wait for them on the spot)
#ifndef HLX_COOP_RESPAWN
#define HLX_COOP_RESPAWN 1
#endif
        // Round 4: the WAVE copies a prepared episode, not the lane that finished.  The stamps showed where a launch's tail still came
        // from (profiles/r04_stamps_contract_base_p99.txt): the slowest 1-5 % of the waves are the ones with a finished lane, and they
        // spend 3 900 cycles at the loop exit (280 for the others) plus 1 000 in the section that requests the entry -- one lane
        // executing ~30 16-byte loads, ~90 accumulation-register reads, the unpacking of every state word and the ~25 stores of the
        // final section on behalf of its new episode, 330 wave instructions for 512 bytes.  An entry is POOL_GROUPS 16-byte words:
        // lane l of the wave now loads word l of the finished lane's entry (ONE load instruction per finished lane, up to COOP_MAX of
        // them per wave; a wave with more computes the others in place), keeps it in ONE register quad through the observation pass,
        // and stores it where that word belongs -- state group l of the finished lane's arena slot, the ring planes, the finished
        // lane's row of the observation tile -- behind the wave's own final stores.  The finished lane's registers keep the old
        // episode; its own (stale) stores to those addresses are issued first and overwritten in order.  -DHLX_COOP_RESPAWN=0: the
        // round-3 form (A/B).
        constexpr int COOP_MAX = 4;
        float4 coop[COOP_MAX];                 // (deliberately uninitialised, like `pf`: written and read under `coop_mask` only)
        const bool coop_whole = n - (int)blockIdx.x * 64 >= 64;      // a whole block: every lane of the wave is at work
        unsigned long long coop_mask = 0ull;   // the finished lanes of this wave whose prepared episode the wave copies
        unsigned long long coop_over = 0ull;   // ... and the ones beyond COOP_MAX with a good entry: computed in place, counted apart from misses
        float2 coop_beam = make_float2(2.f, -2.f);   // a finished lane's OWN copy of the beam test its entry was computed with (cosine, threshold)
        float4 pf[POOL_GROUPS];
        bool hit_pf = false;
        if (LONE && HLX_COOP_RESPAWN) {
            // (whole blocks only: the copy needs lanes 0 .. POOL_GROUPS - 1 at work; the environments of a partial tail block
            // compute their next episode in place, like any environment whose entry is not there -- same bits)
            if (RARE(__ballot(done) != 0ull) && (slots & (1u << 24)) != 0u && !single && coop_whole) {
                unsigned long long m = __ballot(done && pre_tag == pre_ep + 1u);
                const PoolPtrs pq = pool_ptrs(arena, n, g_planes, o_planes);
                const float4* const PB = pq.pool + (size_t)blockIdx.x * (POOL_GROUPS * 64);
#pragma unroll
                for (int k = 0; k < COOP_MAX; ++k) {
                    if (m != 0ull) {
                        const int d = __builtin_ctzll(m);
                        m &= m - 1ull;
                        coop_mask |= 1ull << d;
                        if (lane < POOL_GROUPS) coop[k] = pool_get(PB + d, lane);      // word `lane` of lane d's entry: PB[lane * 64 + d]
                    }
                }
                coop_over = m;
                hit_pf = ((coop_mask >> lane) & 1ull) != 0ull;
                // (the finished lane fetches the two floats that decide whether its entry is still valid for itself: a cross-lane
                // read of the quad above would be a v_readlane of a register the allocator may have parked in an accumulation
                // register in between -- the hazard hotcheck.py refuses)
                if (hit_pf) {
                    typedef float f2v_ __attribute__((ext_vector_type(2)));
                    const f2v_ bt = *(const __attribute__((address_space(1))) f2v_*)(reinterpret_cast<const float*>(PB + (POOL_GROUPS - 1) * 64 + lane) + 2);
                    coop_beam = make_float2(bt.x, bt.y);
                }
            }
        }
        if (LONE && !HLX_COOP_RESPAWN) {
            if (RARE(__ballot(done) != 0ull) && (slots & (1u << 24)) != 0u && !single) {
                hit_pf = done && pre_tag == pre_ep + 1u;
                if (hit_pf) {
                    const PoolPtrs pq = pool_ptrs(arena, n, g_planes, o_planes);
                    const float4* const PA = pq.pool + (size_t)blockIdx.x * (POOL_GROUPS * 64) + lane;
#pragma unroll
                    for (int g = 0; g < POOL_GROUPS; ++g) {
                        const bool used = g <= G_KFP || (g == G_THRUST && (HAS(HLX_F_THRUST_LAG) || HAS(HLX_F_DOMAIN_RAND))) ||
                                          (g == G_MISC && HAS(HLX_F_DOMAIN_RAND)) ||
                                          (g >= G_VPOS && g < N_GROUPS && HAS(HLX_F_VOLLEY) && ((g - G_VPOS) % HLX_MAX_VOLLEY) < VK) ||
                                          (g == PG_ON && HOT(c.o_delay) > 0) ||
                                          (g >= PG_GR && g < PG_ROW && HAS(HLX_F_GROUND) && HOT(c.g_delay) > 0) || g >= PG_ROW;
                        if (used) pf[g] = pool_get(PA, g);
                    }
                }
            }
        }
#ifndef HLX_EARLY_STATE_STORES
#define HLX_EARLY_STATE_STORES 1
#endif
#ifndef HLX_EARLY_STATE_POS
#define HLX_EARLY_STATE_POS 0     // 0: behind the info block (kept); 1: ahead of the scalar outputs (A/B)
#endif
        constexpr bool EARLY_ST = HLX_EARLY_STATE_STORES && MODE == 0 && !PERSIST;
#define HLX_EARLY_BLOCK                                                                    \
        if (EARLY_ST) {                                                                    \
            STG(G_IPOS, make_float4(ipos.x, ipos.y, ipos.z, fuel));                        \
            STG(G_IVEL, make_float4(ivel.x, ivel.y, ivel.z, prev_distance));               \
            STG(G_QUAT, make_float4(q.w, q.x, q.y, q.z));                                  \
            STG(G_MPOS, make_float4(mpos.x, mpos.y, mpos.z, min_distance));                \
            STG(G_MVEL, make_float4(mvel.x, mvel.y, mvel.z, last_distance));               \
            STG(G_W0, make_double2(wind.x, wind.y));                                       \
            if (!HLX_AB_AUX_LATE) STAUX(fuel_used);                                        \
        }
#if HLX_EARLY_STATE_POS == 1
        HLX_EARLY_BLOCK
#endif
        // ---------------------------------------------------------------------- early scalar outputs
        // The step's scalar outputs go out now, while ~5k cycles of observation math follow.  (Storing the
        // integrator's state groups here as well was measured: +0.5 us/step at 65 536 envs -- the stores
        // contend with the tail of the entry loads -- so the state groups stay in the final store section.)
        uint32_t info_word = 0;     // hlx_info_soa: bits 0-4 of `flags`, `missiles` << 8, terminated << 16, truncated << 17
        __amdgpu_buffer_rsrc_t rsI = rsA;   // hlx_info_soa.packed as a buffer (set where the first two words are stored)
        if (MODE == 0) {
            reward_out[i] = reward;
            term_out[i] = terminated ? 1 : 0;
            trunc_out[i] = truncated ? 1 : 0;
            if (INFO_WANTED(slots & (1u << 20))) {   // some hlx_info_soa plane is wanted (one SGPR test instead of nine pointer fetches)
                // info['intercepted' | 'missile_hit_target' | 'proximity_fuze_triggered' | 'clamped' | 'crossed_threshold'] (:829-857) and
                // info['missiles_*'] (:846-847) as they stand at the END OF THE STEP, before an auto-reset clears them (until round 3
                // `crossed_threshold` of a finished environment was read after its respawn, i.e. always False); the detection bits
                // join when observation pass 0 has decided them
                {
                    int remaining = intercepted ? 0 : 1, got = intercepted ? 1 : 0;
                    if (HAS(HLX_F_VOLLEY)) {
                        remaining = 0; got = n_int;
#pragma unroll
                        for (int k = 0; k < HLX_MAX_VOLLEY; ++k) remaining += (k < VK && vact[k]) ? 1 : 0;
                    }
                    info_word = (intercepted ? 1u : 0u) | (hit_target ? 2u : 0u) | (fuze ? 4u : 0u) | (clamped ? 8u : 0u) | (crossed ? 16u : 0u) |
                                ((uint32_t)(got | (remaining << 4)) << 8) | (terminated ? 1u << 16 : 0u) | (truncated ? 1u << 17 : 0u);
                }
                if (slots & (1u << 23)) {
                    // hlx_info_soa.packed, the form a VecEnv facade passes (hlx.h): the nine standard keys as three 16-byte words per
                    // environment, stored like state groups -- one coalesced 16-byte-per-lane instruction each -- where rounds 1-3 issued
                    // eleven 4-byte and four 1-byte stores (1.15 us of the 8.9 us launch for 7.6 % of its bytes).  Words 0 and 1 are
                    // final here; word 2 (missile position + the flag word) leaves behind observation pass 0.
                    rsI = __builtin_amdgcn_make_buffer_rsrc((void*)HOT(opt.info.packed), 0, 0x7FFFFFFF, 0x00020000);
                    wt16i(rsI, (uint32_t)i * 16u, 0u, make_float4(distance, min_distance, fuel, fuel_used));
                    wt16i(rsI, (uint32_t)i * 16u, (uint32_t)n * 16u, make_float4(ipos.x, ipos.y, ipos.z, __int_as_float(steps)));   // :836-838
                    if (HLX_INFO_W2 == 1) wt16i(rsI, (uint32_t)i * 16u, (uint32_t)n * 32u, make_float4(mpos.x, mpos.y, mpos.z, __uint_as_float(info_word)));
                }
                if (RARE(!(slots & (1u << 23)))) {     // separate planes (an `if` of its own behind RARE(), not an `else`: the block then sits out of line)
                    if (HOT(opt.info.distance)) G(HOT(opt.info.distance))[i] = distance;
                    if (HOT(opt.info.min_distance)) G(HOT(opt.info.min_distance))[i] = min_distance;
                    if (HOT(opt.info.fuel)) G(HOT(opt.info.fuel))[i] = fuel;
                    if (HOT(opt.info.fuel_used)) G(HOT(opt.info.fuel_used))[i] = fuel_used;    // :834, :886
                    if (HOT(opt.info.interceptor_pos)) {                                  // :836-838 (post-step, pre-respawn values)
                        auto ip = G(HOT(opt.info.interceptor_pos)) + i;
                        ip[0] = ipos.x; ip[(size_t)n] = ipos.y; ip[2 * (size_t)n] = ipos.z;
                    }
                    if (HOT(opt.info.missile_pos)) {
                        auto mp = G(HOT(opt.info.missile_pos)) + i;
                        mp[0] = mpos.x; mp[(size_t)n] = mpos.y; mp[2 * (size_t)n] = mpos.z;
                    }
                    if (HOT(opt.info.steps)) G(HOT(opt.info.steps))[i] = steps;
                    if (HOT(opt.info.missiles)) G(HOT(opt.info.missiles))[i] = (uint8_t)(info_word >> 8);      // :846-847
                }
                if (HAS(HLX_F_VOLLEY) && HOT(opt.info.missile_min_distances)) {               // :848
                    auto md = G(HOT(opt.info.missile_min_distances)) + i;
#pragma unroll
                    for (int k = 0; k < HLX_MAX_VOLLEY; ++k) if (k < VK) md[(size_t)k * (size_t)n] = vmin[k];
                }
            }
        }
#undef PIN4
#undef PIN2
        // The integrator's six state groups are final here for every lane that does not finish: they are stored now, while the
        // observation section computes and the memory system idles, instead of in the burst at the end of the wave (the store
        // phase is a fifth of a wave's life: every wave of the launch writes its 24 KB at the same moment); finished lanes,
        // whose respawn replaces the values, store them again at the end.  Rounds 1 and 2 measured early stores twice and lost
        // 0.5 us both times (plain stores, one load batch); with write-through stores and the second load batch behind Philox
        // the same idea gains 0.5 us in the two-pass forms and is neutral in the single-pass one
        // (profiles/r03_ab_early_state_stores.txt).  Storing MORE early -- the Kalman groups, the packed word and the ring
        // samples behind the filter -- loses again (9.96 against 9.74), and so does making this block conditional on the form.
        // -DHLX_EARLY_STATE_STORES=0: everything in the final burst (A/B).
#if HLX_EARLY_STATE_POS == 0
        HLX_EARLY_BLOCK
#endif
        D3 kxp = d3(g_kf0.x, g_kf0.y, g_kf1.x), kxv = d3(g_kf1.y, g_kf2.x, g_kf2.y);
        float p_pp = g_kfp.x, p_pv = g_kfp.y, p_vp = g_kfp.z, p_vv = g_kfp.w;
        // ---------------------------------------------------------------------- observation (+ auto-reset)
        // pass 0: observation of the stepped state.  pass 1 (only if some lane of the wave finished):
        // finished lanes respawn (environment.py:353-603) and build their first observation.
        const int o_cap = HOT(c.o_cap);
        float4 on_sample = make_float4(0.f, 0.f, 0.f, 0.f), g_s2 = on_sample;
        D3 g_sp = d3(0., 0., 0.);          // ground ring sample: float64 measured rel_pos, float32 quality/flag, rel_vel
        float g_sq = 0.f, g_sflag = 0.f;
        uint32_t det_bits = 0;
        // One observation pass or two?  A finished environment needs the observation of its terminal state only if somebody
        // asked for it (hlx_step's terminal_obs; the info planes carry the terminal step's detection flags): then pass 0
        // observes the stepped state of every lane and pass 1 the fresh state of the finished ones.  When nobody did
        // (hlx_rollout has no such output at all), finished lanes respawn BEFORE the single observation pass and every lane
        // is observed once: the waves that restart an episode -- the ones that keep a launch open -- lose a whole second
        // trip through the observation code.  Outputs are bit-identical either way (bench.py's self-check compares the
        // two forms: the big batch runs single-pass through hlx_rollout, its slabs two-pass through hlx_step).
        // A do-while whose back edge is the rare direction (second trip only when a terminal observation is wanted AND some
        // lane of the wave finished): the common step falls out of the loop without a taken branch.
        int pass = (MODE == 0 ? 0 : 1);
        bool again;
        // One trip through the observation code.  The second trip of a step launch (finished lanes respawn and are observed again:
        // rare, out of line, and the tail of every launch that wants terminal observations) is a SPECIALISED copy: every lane it
        // observes has just respawned -- no delayed samples, no initialised filter, no detection history -- so the Kalman update /
        // predict, the ring reads and their selects are not compiled into it at all, where the shared loop body of rounds 1-2
        // jumped over them one taken branch at a time.  -DHLX_FRESH_TRIP=0: the second trip as an unspecialised copy (A/B).
#ifndef HLX_FRESH_TRIP
#define HLX_FRESH_TRIP 1
#endif
        float beam_cb = 2.f;      // MODE 2: the beam-test cosine of the entry's first observation (2 = the test was not reached)
        auto trip = [&](auto all_fresh_tag) __attribute__((always_inline)) {
            constexpr bool ALLF = decltype(all_fresh_tag)::value;
            STAMP2(1);  // close-up: loop top
            bool fresh = false;      // this lane has just respawned: its observation is the first of a new episode
            // `rsalt` is an opaque zero defined inside the respawn pass: the respawn draws are pure functions of
            // loop-invariant values, and without it the optimiser hoists all six Philox chains (and the float64
            // spawn trigonometry) out of the loop, i.e. executes them on EVERY step for every environment.
            uint32_t rsalt = 0;
            // Respawn draws: item j = Philox stream `8 + j` (j < 8: spawn uniforms x3, first-observation uniforms, ground
            // position / velocity normals, two domain-randomisation normal quads) or `13 + j` (j = 8..10: volley missiles 1..3).
            // A finished lane needs up to eleven Philox evaluations; its wave has 63 other lanes with nothing to do in this
            // pass, so the WAVE evaluates them: for one finished lane at a time, lane j computes item j of that lane's
            // counter, and the finished lane picks the results up with v_readlane -- one Philox evaluation (+ one Box-Muller
            // pair) per finished lane instead of eleven in a row on the launch's critical tail.  Same keys and counters as
            // the per-lane form (kept for partial tail blocks and the parity instantiation), hence the same draws.
            constexpr int RS_ITEMS = 11;
            // this pass's detection draws: the step's own, replaced INSIDE the respawn block by a fresh lane's first-observation
            // draws (selecting them afterwards under `fresh` costs the common path nine copies and nine zero-initialisations)
            float n_on = u_on, n_g = u_g;
            double n_dl = u_dl;
            D3 n_gp = z_gp, n_gv = z_gv;
            unsigned long long dmask = (ALLF || pass == 1 || single) ? __ballot(done) : 0ull;
            bool hit = false;          // this finished lane's next episode is waiting in the pool: copied, not computed
            if (RARE(dmask != 0ull)) {
                // Which random numbers start an episode.  An explicit reset (MODE 1) draws from the launch's clock word (+ reset
                // epoch, hlx.h).  An AUTO-reset draws from (environment id, index of the episode among the environment's
                // auto-resets): counter words {id, id >> 32, episode, 0xFFFFFF00 | stream} -- the high word is one no clock value
                // reaches -- so that the episode can be prepared before the step that needs it is known.
                uint32_t epn = 0;
                int nn = n;
                asm volatile("" : "+s"(nn));
                const PoolPtrs pp = pool_ptrs(arena, nn, g_planes, o_planes);
                if (MODE != 1 && done) {
                    epn = (POOL_PRE ? pre_ep : G(pp.ep_cur)[i]) + 1u;
                    if (POOL_PRE || MODE == 2) {
                        const uint32_t tag = POOL_PRE ? pre_tag : G(pp.tag)[i];
                        // (The fused rollout computes its respawns in place: its waves drift apart over the steps of a launch, no
                        // single step's stragglers hold it open.  So does the single-pass form: there the copy would wait for memory
                        // exactly where the spawn arithmetic runs today, with nothing saved behind it.)
                        // (... and the schedules for more than one wave per SIMD: there the stragglers of one wave are covered by the
                        // other, and a copy fetched at the end of the pass measured slower than computing -- 16.2 against 15.9 us at
                        // 131 072 environments.  The host sets bit 24 for the lone-wave schedule only.)
                        if (POOL_PRE) hit = LONE ? hit_pf : false;
                        if (LONE && HLX_COOP_RESPAWN) hit = false;      // (decided for the whole wave below: coop_mask)
                        if (LONE && !HLX_COOP_RESPAWN && hit) {
                            // The one curriculum scalar the reference really ramps (config.yaml:85-92: the beam width, 120 -> 60 degrees over
                            // 3 M steps, moved by EVERY call of set_training_step_count) enters a first observation through one decision:
                            // cosine of the off-boresight angle against the threshold.  The entry carries that cosine and the threshold it
                            // was compared with; it is the first observation under TODAY's threshold too iff both comparisons fall on
                            // the same side, by more than the margin inside which the decision replays the reference's arithmetic.
                            // (A new episode looks straight at its missile -- cosine 1 -- so a beam ramp invalidates next to nothing; rounds
                            // before this one renewed ALL entries, a 12-45 us launch, ahead of every step of the ramp.)
                            const float cbe = acc_read(pf[PG_ROW + (HLX_OBS_DIM + 3) / 4 - 1].z), cte = acc_read(pf[PG_ROW + (HLX_OBS_DIM + 3) / 4 - 1].w);
                            const float ctn = cur_cos_half_beam;
                            hit = (cbe - ctn > 1e-5f && cbe - cte > 1e-5f) || (ctn - cbe > 1e-5f && cte - cbe > 1e-5f);
                        }
                        (void)tag;
                        if (MODE == 2 && (slots & (1u << 25)) != 0u && tag == epn) done = false;    // renewed since (a fill of every entry)
                    }
                    if (MODE == 2) pool_epn = epn;
                }
                if (LONE && HLX_COOP_RESPAWN && coop_mask != 0ull) {
                    // the beam test each copied entry was computed with, against today's threshold (see the round-3 form below / hlx.h)
                    const float ctn = cur_cos_half_beam, cbe = coop_beam.x, cte = coop_beam.y;
                    const bool ok = (cbe - ctn > 1e-5f && cbe - cte > 1e-5f) || (ctn - cbe > 1e-5f && cte - cbe > 1e-5f);
                    coop_mask &= __ballot(done && hit_pf && ok);
                    hit = done && ((coop_mask >> lane) & 1ull) != 0ull;
                }
                if (MODE == 2) dmask = __ballot(done);
                const unsigned long long smask = __ballot(done && !hit);     // the lanes that compute their respawn here
                float rd[RS_ITEMS][4];
                if (!(LONE && HLX_COOP_RESPAWN) || smask != 0ull) {     // (44 moves the wave of a copied episode has no use for)
#pragma unroll
                    for (int j = 0; j < RS_ITEMS; ++j) rd[j][0] = rd[j][1] = rd[j][2] = rd[j][3] = 0.f;
                }
                asm volatile("" : "+v"(rsalt));
                // lanes 0..10, the ones that serve, are live (a pool fill has every lane at work: each draws for itself)
                const bool wide = MODE != 2 && (n - (int)blockIdx.x * 64) >= RS_ITEMS;
                if (!rnoise_buf && smask != 0ull) {
                    if (wide) {
                        unsigned long long m = smask;
                        while (m) {
                            const int d = __builtin_ctzll(m);
                            m &= m - 1ull;
                            const unsigned long long gd = (unsigned long long)(env_offset + (long long)blockIdx.x * 64 + d);
                            const Rng rw{rng.key, (uint32_t)gd, (uint32_t)(gd >> 32), MODE == 1 ? rng.t_lo : (uint32_t)__builtin_amdgcn_readlane(epn, d),
                                         MODE == 1 ? rng.t_hi : 0xFFFFFFu};
                            const uint4 x = rw.raw((lane < 8 ? 8u : 13u) + (uint32_t)lane);
                            float nz[4], w[4];
                            box_muller(x.x, x.y, nz[0], nz[1]);
                            box_muller(x.z, x.w, nz[2], nz[3]);
                            const bool is_normal = lane >= 4 && lane < 8;
                            w[0] = is_normal ? nz[0] : u01(x.x); w[1] = is_normal ? nz[1] : u01(x.y);
                            w[2] = is_normal ? nz[2] : u01(x.z); w[3] = is_normal ? nz[3] : u01(x.w);
                            const bool mine = lane == d;
#pragma unroll
                            for (int j = 0; j < RS_ITEMS; ++j) {
                                const bool wanted = j < 6 || (j < 8 && HAS(HLX_F_DOMAIN_RAND)) || (j >= 8 && HAS(HLX_F_VOLLEY) && j - 7 < VK);
                                if (!wanted) continue;
#pragma unroll
                                for (int c = 0; c < 4; ++c) {
                                    const float v = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(w[c]), j));
                                    rd[j][c] = mine ? v : rd[j][c];
                                }
                            }
                        }
                    } else if (done && !hit) {
                        const Rng rk{rng.key, rng.id_lo, rng.id_hi, MODE == 1 ? rng.t_lo : epn, MODE == 1 ? rng.t_hi : 0xFFFFFFu};
                        // (a tail block with fewer than eleven live lanes: the lane draws for itself, one stream per trip of a
                        // ROLLED loop -- this path runs for at most ten environments of a launch and must stay small: the
                        // kernel's code is fetched cold at every launch, and every kilobyte of it is paid by the first wave
                        // of each instruction cache to get there)
#pragma unroll 1
                        for (int j = 0; j < RS_ITEMS; ++j) {
                            const uint4 x = rk.raw((j < 8 ? 8u : 13u) + (uint32_t)j + rsalt);
                            float w[4], nz[4];
                            box_muller(x.x, x.y, nz[0], nz[1]);
                            box_muller(x.z, x.w, nz[2], nz[3]);
                            const bool is_normal = j >= 4 && j < 8;
                            w[0] = is_normal ? nz[0] : u01(x.x); w[1] = is_normal ? nz[1] : u01(x.y);
                            w[2] = is_normal ? nz[2] : u01(x.z); w[3] = is_normal ? nz[3] : u01(x.w);
#pragma unroll
                            for (int jj = 0; jj < RS_ITEMS; ++jj)
                                if (jj == j) { rd[jj][0] = w[0]; rd[jj][1] = w[1]; rd[jj][2] = w[2]; rd[jj][3] = w[3]; }
                        }
                    }
                }
                if (MODE == 0 && !PERSIST && (slots & (1u << 24)) != 0u) {
                    // (the fused rollout records nothing: the host renews every entry behind it)
                    // these lanes have used their entry: one non-returning atomic OR into the block's mask -- and how many had to
                    // compute their respawn here (diagnostics: hlx_get_episode_pool_misses)
                    if (lane == __builtin_ctzll(dmask)) {
                        (void)__hip_atomic_fetch_or(G(pp.rf_mask) + blockIdx.x, dmask, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (smask != 0ull && !single) {    // (the single-pass form computes in place by design: not a miss)
                            // [2] the entry was absent or stale; [3] it was there, but the wave had more finished lanes than it copies for
                            // (a batch-wide truncation step: everyone computing at once is the faster way through that one launch)
                            const unsigned long long over = LONE && HLX_COOP_RESPAWN ? smask & coop_over : 0ull;
                            if ((smask & ~over) != 0ull)
                                (void)__hip_atomic_fetch_add(G(pp.rf_cnt) + 2, __popcll(smask & ~over), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (RARE(over != 0ull))
                                (void)__hip_atomic_fetch_add(G(pp.rf_cnt) + 3, __popcll(over), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                }
                if (LONE && HLX_COOP_RESPAWN && coop_whole && MODE == 0 && pass == 1 && HOT(opt.terminal_obs)) {
                    // the terminal observation of every finished lane, copied by the WAVE: lanes 0-12 move one 8-byte pair each of the
                    // finished lane's row of the tile (pass 0's observation of the terminal state) -- one LDS read and one store per
                    // finished lane where the lane itself issued thirteen of each
                    typedef float f2v_ __attribute__((ext_vector_type(2)));
                    typedef __attribute__((address_space(1))) f2v_ gf2v_;
                    unsigned long long mm = dmask;
                    while (mm != 0ull) {
                        const int d = __builtin_ctzll(mm);
                        mm &= mm - 1ull;
                        if (lane < HLX_OBS_DIM / 2) {
                            const f2v_ tv = reinterpret_cast<const f2v_*>(tile + d * HLX_OBS_DIM)[lane];
                            ((gf2v_*)(HOT(opt.terminal_obs) + ((size_t)blockIdx.x * 64 + (size_t)d) * HLX_OBS_DIM))[lane] = tv;
                        }
                    }
                }
                if (done) {
                    fresh = true;
                    if (MODE == 0) G(pp.ep_cur)[i] = epn;
                    if (hit) {
                    } else if (rnoise_buf) {   // first observation of a new episode: its own draws
                        const double* B = RN + 10 * N;
                        n_on = (float)B[0]; n_g = (float)B[1 * N]; n_gp = d3(B[2 * N], B[3 * N], B[4 * N]);
                        n_gv = d3(B[5 * N], B[6 * N], B[7 * N]); n_dl = B[8 * N];
                    } else {   // items 3, 4, 5 = streams RS_RESET_OBS_U, RS_RESET_GPOS, RS_RESET_GVEL
                        n_on = rd[3][0]; n_g = rd[3][1]; n_dl = (double)rd[3][2];
                        n_gp = d3((double)rd[4][0], (double)rd[4][1], (double)rd[4][2]);
                        n_gv = d3((double)rd[5][0], (double)rd[5][1], (double)rd[5][2]);
                    }
                    if (MODE == 0 && pass == 1) {
                        if (HOT(opt.terminal_obs) && !(LONE && HLX_COOP_RESPAWN && coop_whole)) {
                            // the lane's row of the LDS tile (pass 0's observation of the terminal state) -> terminal_obs[i]: all
                            // thirteen 8-byte LDS reads first, then thirteen GLOBAL stores (through a generic pointer the
                            // compiler must assume the store may hit LDS and serialises read / wait / store 26 times)
                            typedef float f2v_ __attribute__((ext_vector_type(2)));
                            typedef __attribute__((address_space(1))) f2v_ gf2v_;
                            gf2v_* to = (gf2v_*)(HOT(opt.terminal_obs) + (size_t)i * HLX_OBS_DIM);
                            const f2v_* r2 = reinterpret_cast<const f2v_*>(row);
                            f2v_ tv[HLX_OBS_DIM / 2];
#pragma unroll
                            for (int k = 0; k < HLX_OBS_DIM / 2; ++k) tv[k] = r2[k];
#pragma unroll
                            for (int k = 0; k < HLX_OBS_DIM / 2; ++k) to[k] = tv[k];
                        }
                        if (HOT(opt.info.episode_return)) G(HOT(opt.info.episode_return))[i] = ep_return;
                        if (HOT(opt.info.episode_length)) G(HOT(opt.info.episode_length))[i] = steps;
                    }
                    if (!hit) {
                    // ---------------- spawn (environment.py:375-567): float64 draws cast to float32
                    const bool rbuf = rnoise_buf;
                    double u[10];
                    if (rbuf) {
#pragma unroll
                        for (int k = 0; k < 10; ++k) u[k] = RN[k * N];
                    } else {   // items 0-2 = streams RS_RESET_U0..U2
                        u[0] = rd[0][0]; u[1] = rd[0][1]; u[2] = rd[0][2]; u[3] = rd[0][3];
                        u[4] = rd[1][0]; u[5] = rd[1][1]; u[6] = rd[1][2]; u[7] = rd[1][3];
                        u[8] = rd[2][0]; u[9] = rd[2][1];
                    }
                    const V3 tp = v3(HOT(c.target[0]), HOT(c.target[1]), HOT(c.target[2]));
                    auto spawn_missile = [&](double u0, double u1, double u2, double u3, V3& mpos, V3& mvel) {
                        if (HAS(HLX_F_SPHERICAL)) {                                 // :390-406
                            const double PI = 3.141592653589793;
                            double radius = COLD(mis_radius[0]) + COLD(mis_radius[1]) * u0;
                            double az = ((COLD(mis_az[0]) + COLD(mis_az[1]) * u1) * PI) / 180.0;
                            double el = ((COLD(mis_el[0]) + COLD(mis_el[1]) * u2) * PI) / 180.0;
                            mpos = v3((float)((double)tp.x + (radius * cos(el)) * cos(az)), (float)((double)tp.y + (radius * cos(el)) * sin(az)),
                                      (float)((double)tp.z + radius * sin(el)));
                        } else {                                                    // :409
                            mpos = v3((float)(COLD(mis_lo[0]) + COLD(mis_span[0]) * u0), (float)(COLD(mis_lo[1]) + COLD(mis_span[1]) * u1),
                                      (float)(COLD(mis_lo[2]) + COLD(mis_span[2]) * u2));
                        }
                        float speed = (float)(COLD(mis_speed[0]) + COLD(mis_speed[1]) * u3); // :415
                        V3 tt = tp - mpos;
                        float ttd = snorm3(tt);
                        mvel = (ttd > 1e-6f) ? (tt / ttd) * speed : v3(0.f, 0.f, 0.f); // :418-423
                    };
                    spawn_missile(u[0], u[1], u[2], u[3], mpos, mvel);
                    if (HAS(HLX_F_VOLLEY)) {                                        // :386-439: volley_size missiles, four draws each
                        vp[0] = mpos; vv[0] = mvel;
#pragma unroll
                        for (int m = 1; m < HLX_MAX_VOLLEY; ++m) {
                            if (m < VK) {
                                double w0, w1, w2, w3;
                                if (rbuf) {
                                    const double* B = RN + (size_t)(32 + 4 * (m - 1)) * N;
                                    w0 = B[0]; w1 = B[N]; w2 = B[2 * N]; w3 = B[3 * N];
                                } else {   // items 8-10 = streams RS_RESET_V1..
                                    w0 = rd[7 + m][0]; w1 = rd[7 + m][1]; w2 = rd[7 + m][2]; w3 = rd[7 + m][3];
                                }
                                spawn_missile(w0, w1, w2, w3, vp[m], vv[m]);
                            }
                        }
                    }
                    ipos = v3((float)(COLD(int_lo[0]) + COLD(int_span[0]) * u[4]), (float)(COLD(int_lo[1]) + COLD(int_span[1]) * u[5]),
                              (float)(COLD(int_lo[2]) + COLD(int_span[2]) * u[6]));         // :445
                    V3 rel0 = mpos - ipos;
                    float reld = snorm3(rel0);
                    range_c = reld;                                                 // (non-volley: the observation's own range)
                    if (HAS(HLX_F_TOWARD_MISSILE) && reld > 1e-6f)                  // :452-462
                        ivel = (rel0 / reld) * (float)(COLD(int_speed[0]) + COLD(int_speed[1]) * u[7]);
                    else                                                            // :467
                        ivel = v3((float)(COLD(ivel_lo[0]) + COLD(ivel_span[0]) * u[7]), (float)(COLD(ivel_lo[1]) + COLD(ivel_span[1]) * u[8]),
                                  (float)(COLD(ivel_lo[2]) + COLD(ivel_span[2]) * u[9]));
                    V3 orel = rel0;
                    float oreld = reld;
                    if (HAS(HLX_F_VOLLEY)) {   // per-missile minimum distances (:469-473); point at the CLOSEST missile (:476-487)
                        float best = 0.f;
#pragma unroll
                        for (int m = 0; m < HLX_MAX_VOLLEY; ++m) {
                            if (m < VK) {
                                const V3 r = vp[m] - ipos;
                                const float dm = snorm3(r);
                                vmin[m] = dm; vact[m] = true;
                                if (m == 0 || dm < best) { best = dm; orel = r; oreld = dm; }
                            }
                        }
                        prio = 0; n_int = 0;                                        // :439 self.missile_state = missile_states[0]
                    }
                    q = Quat{1.f, 0.f, 0.f, 0.f};                                   // :489-530 rotate +Z onto the LOS (float64)
                    if (oreld > 1e-6f) {
                        V3 fd = orel / oreld;
                        double ax = -(double)fd.y, ay = (double)fd.x;
                        double axl = sqrt(ax * ax + ay * ay);
                        if (axl > 1e-6) {
                            // cos(acos(c) / 2) = sqrt((1 + c) / 2), sin(acos(c) / 2) = sqrt((1 - c) / 2): two float64 square roots instead
                            // of acos + sin + cos -- several thousand cycles on the respawn path, i.e. on the waves that keep a launch
                            // open (8.80 against 9.37 us per launch, A/B).  Equal to the reference's float64 values to a few
                            // ulp(float64): the float32 quaternion can differ in its last bit with probability ~1e-8 per component.
                            const double cz = fmin(fmax((double)fd.z, -1.0), 1.0);
                            const double chalf = sqrt((1.0 + cz) * 0.5), sh = sqrt((1.0 - cz) * 0.5);
                            q = Quat{(float)chalf, (float)((ax / axl) * sh), (float)((ay / axl) * sh), 0.f};
                        } else if (!(fd.z > 0.f)) q = Quat{0.f, 1.f, 0.f, 0.f};
                    }
                    on_delay = HOT(c.o_delay);                                           // constructor value (core.py:292-293)
                    fuel = 100.f;                                                   // :537
                    wind = d3((double)HOT(c.base_wind[0]), (double)HOT(c.base_wind[1]), (double)HOT(c.base_wind[2])); // :542
                    thrust_act = v3(0.f, 0.f, 0.f);                                 // :549
                    if (HAS(HLX_F_DOMAIN_RAND)) {                                   // :552-562, physics_randomizer.py
                        double zt, zd, zm, zs;
                        if (rbuf) { zt = RN[20 * N]; zd = RN[21 * N]; zm = RN[22 * N]; zs = RN[23 * N]; }
                        else {
                            // draws 1..4 of the 13 (the others never reach the path): temperature, drag, mach, delay
                            zt = rd[6][1]; zd = rd[6][2]; zm = rd[6][3]; zs = rd[7][0];      // items 6, 7 = streams RS_DR0, RS_DR1
                        }
                        auto mult = [](double var, double z) { return fmin(fmax(1.0 + var * z, 0.1), 3.0); };
                        if (HAS(HLX_F_ATMOSPHERE)) T0 = T0 + COLD(dr_var[1]) * zt;           // :258-261 (accumulates, float64)
                        if (HAS(HLX_F_MACH_DRAG)) {                                 // :270-280
                            const double cd64 = 0.3 * mult(COLD(dr_var[2]), zd);
                            dp.base_cd = (float)cd64; dp.cd_super = (float)(cd64 * HOT(c.super_mult));
                            dp.peak_m1 = (float)(3.0 * mult(COLD(dr_var[3]), zm) - 1.0);
                        }
                        if (HOT(c.o_delay) > 0)                                          // :289-297
                            on_delay = min(10, max(1, (int)(3.0 * mult(COLD(dr_var[4]), zs))));
                    }
                    steps = 0; ep_return = 0.f; fuel_used = 0.f;                    // :565-566
                    kf_init = false; kf_x64 = false;                                // core.py:65-69
                    kxp = d3(0., 0., 0.); kxv = kxp;
                    p_pp = 1000.f; p_pv = 0.f; p_vp = 0.f; p_vv = 1000.f;
                    prev_distance = reld; last_distance = reld; min_distance = reld; // :579-589
                    worsening = 0; crossed = false;
                    } else if (LONE && HLX_COOP_RESPAWN) {
                        // the wave copies this lane's prepared episode behind the final stores (below): nothing to unpack.  The dword
                        // plane is the one piece of state an entry does not hold: a new episode has used no fuel
                        fuel_used = 0.f;
                    } else {
                        // ---------------- the episode was prepared by a pool fill: state groups, ring samples, observation row
                        // (unpacked exactly as at kernel entry; what the fill stored is what the store section below packs)
                        const float4* const PA = pp.pool + (size_t)blockIdx.x * (POOL_GROUPS * 64) + lane;
                        // (lone-wave schedule: the entry was requested before the observation pass and has been parked in the
                        // accumulation registers since -- the pass needs the ordinary ones; see acc_read)
                        const auto PL = [&](int g) {
                            if (!LONE) return pool_get(PA, g);
                            return make_float4(acc_read(pf[g].x), acc_read(pf[g].y), acc_read(pf[g].z), acc_read(pf[g].w));
                        };
                        const auto dbl = [](float lo, float hi) { return __hiloint2double(__float_as_int(hi), __float_as_int(lo)); };
                        const float4 a = PL(G_IPOS), b = PL(G_IVEL), c = PL(G_QUAT), d = PL(G_MPOS), e4 = PL(G_MVEL), w0 = PL(G_W0), w1 = PL(G_W1);
                        const float4 k0 = PL(G_KF0), k1 = PL(G_KF1), k2 = PL(G_KF2), kp = PL(G_KFP);
                        ipos = v3(a.x, a.y, a.z); fuel = a.w;
                        ivel = v3(b.x, b.y, b.z); prev_distance = b.w;
                        q = Quat{c.x, c.y, c.z, c.w};
                        mpos = v3(d.x, d.y, d.z); min_distance = d.w;
                        mvel = v3(e4.x, e4.y, e4.z); last_distance = e4.w;
                        wind = d3(dbl(w0.x, w0.y), dbl(w0.z, w0.w), dbl(w1.x, w1.y));
                        {
                            const uint32_t pk = __float_as_uint(w1.z);
                            steps = (int)(pk & 0x1FFFu); worsening = (int)((pk >> 13) & 0xFFFu);
                            crossed = (pk >> 25) & 1u; kf_init = (pk >> 26) & 1u; kf_x64 = (pk >> 27) & 1u;
                            on_delay = (int)(pk >> 28);
                        }
                        ep_return = w1.w; fuel_used = 0.f;
                        kxp = d3(dbl(k0.x, k0.y), dbl(k0.z, k0.w), dbl(k1.x, k1.y));
                        kxv = d3(dbl(k1.z, k1.w), dbl(k2.x, k2.y), dbl(k2.z, k2.w));
                        p_pp = kp.x; p_pv = kp.y; p_vp = kp.z; p_vv = kp.w;
                        if (HAS(HLX_F_THRUST_LAG) || HAS(HLX_F_DOMAIN_RAND)) {
                            const float4 th = PL(G_THRUST);
                            thrust_act = v3(th.x, th.y, th.z);
                            if (HAS(HLX_F_DOMAIN_RAND)) dp.cd_super = th.w;
                        }
                        if (HAS(HLX_F_DOMAIN_RAND)) {
                            const float4 mi = PL(G_MISC);
                            T0 = dbl(mi.x, mi.y); dp.base_cd = mi.z; dp.peak_m1 = mi.w;
                        }
                        if (HAS(HLX_F_VOLLEY)) {
#pragma unroll
                            for (int k = 0; k < HLX_MAX_VOLLEY; ++k) {
                                if (k < VK) {
                                    const float4 va = PL(G_VPOS + k), vb = PL(G_VVEL + k);
                                    const uint32_t w = __float_as_uint(vb.w);
                                    vp[k] = v3(va.x, va.y, va.z); vmin[k] = va.w; vv[k] = v3(vb.x, vb.y, vb.z); vact[k] = (w & 1u) != 0u;
                                    if (k == 0) { prio = (int)((w >> 8) & 3u); n_int = (int)((w >> 12) & 7u); }
                                }
                            }
                        }
                        if (HOT(c.o_delay) > 0) on_sample = PL(PG_ON);
                        if (HAS(HLX_F_GROUND) && HOT(c.g_delay) > 0) {
                            const float4 s0 = PL(PG_GR), s1 = PL(PG_GR + 1);
                            g_sp = d3(dbl(s0.x, s0.y), dbl(s0.z, s0.w), dbl(s1.x, s1.y)); g_sq = s1.z; g_sflag = s1.w;
                            g_s2 = PL(PG_GR + 2);
                        }
                        float4 rw[(HLX_OBS_DIM + 3) / 4];
#pragma unroll
                        for (int k = 0; k < (HLX_OBS_DIM + 3) / 4; ++k) rw[k] = PL(PG_ROW + k);
#pragma unroll
                        for (int k = 0; k < HLX_OBS_DIM; ++k) {
                            const float4 v = rw[k >> 2];
                            row[k] = (k & 3) == 0 ? v.x : (k & 3) == 1 ? v.y : (k & 3) == 2 ? v.z : v.w;
                        }
                    }
                }
            }
            const bool act = (ALLF ? done : ((pass == 0) || done)) && !hit;
            const bool fresh_k = ALLF ? true : fresh;      // (a compile-time `true` in the specialised second trip)
            if (act) {
                // ======================================================== core.py:511-691 radar detection
                STAMP2(2);  // close-up: draws selected
                const V3 rel = mpos - ipos;
#ifndef HLX_ONBOARD_EARLY
#define HLX_ONBOARD_EARLY 0
#endif
                // The delayed onboard sample (physics v2: core.py:576-593) is a GATHER -- domain randomisation gives every lane its own
                // delay, i.e. its own ring plane.  It is read in place, where the onboard section needs it (below).  Requesting it
                // EARLIER has now lost twice: from the physics section behind the second load batch (round 3: +0.8 us), and from here,
                // the top of the pass, with its registers untouched until the measurement fusion ~2 k cycles further down
                // (-DHLX_ONBOARD_EARLY=1, round 4: v2dr contract form 11.28 -> 11.65 us, profiles/r04_ab_v2dr_onboard_gather_early.txt).
                // A 64-line gather at the head of the in-order memory queue holds back the output, info and state stores issued
                // right behind it; in place it follows them.
                float4 on_raw = make_float4(0.f, 0.f, 0.f, 0.f);
                const bool on_have = HOT(c.o_delay) > 0 && !fresh_k && steps >= on_delay;
                if (HLX_ONBOARD_EARLY && HOT(c.o_delay) > 0 && on_have) {
                    int slot = o_wslot - on_delay;                            // (t - on_delay) mod o_cap
                    slot += (slot < 0) ? o_cap : 0;
                    on_raw = oring[(size_t)slot * N + i];
                }
                // ||missile - interceptor|| is at hand: the step's `distance`, or the spawn distance of a lane that has just respawned
                const float range = (HAS(HLX_F_VOLLEY) || MODE != 0) ? snorm3(rel) : range_c;
                bool on_det = !(range > HOT(c.radar_range));                             // :539
                float on_why = on_det ? 0.f : -1.f;   // detection_info['reason'] (:541,:555,:566), carried through the delay ring
                STAMP2(3);  // close-up: range
                const V3 fwd = forward_vec(q);
                STAMP2(4);  // close-up: forward vector
                {   // :546-553  arccos(clip(fwd . to_missile)) > half_beam  <=>  clip(fwd . to_missile) < c*, where c* is the
                    // float32 at which the host's acosf crosses half_beam (found by bisection in hlx_host.inc): the argument is
                    // formed with the reference's own float32 operations, so the DECISION is the reference's, bit for bit
                    // Fast arithmetic decides unless the argument is within 1e-5 of c* (ten times the fast path's error bound):
                    // only then -- about one env-step in 1e5 -- is the reference's operation order replayed.
                    const float cthr = cur_cos_half_beam;
                    float cb = clampf(fdot(fwd, rel) * __builtin_amdgcn_rcpf(range + 1e-6f), -1.f, 1.f);
                    if (RARE(on_det && fabsf(cb - cthr) < 1e-5f)) cb = clampf(sdot3(forward_vec_exact(q), rel / (range + 1e-6f)), -1.f, 1.f);
                    // (a pool entry remembers what this test saw: a beam-width curriculum that moves the threshold leaves the entry
                    // valid as long as the decision is the same on both sides of the move -- see "hit" in the respawn block)
                    if (MODE == 2) beam_cb = on_det ? cb : 2.f;
                    if (on_det && cb < cthr) { on_det = false; on_why = -2.f; }
                }
                STAMP2(5);  // close-up: beam angle (acosf)
                {                                                                   // :559-566 (straight-line: selects, not a divergent block)
                    float aq = (HOT(c.radar_quality) * (1.0f - (range * HOT(c.inv_radar_range)) * 0.5f)) * cur_on_rel;
                    if (RARE(on_det && fabsf(n_on - aq) < 2e-6f))   // Bernoulli draw within a few ulps of the probability: the reference's own division
                        aq = (HOT(c.radar_quality) * (1.0f - HLX_DIVF(range, HOT(c.radar_range)) * 0.5f)) * cur_on_rel;
                    const bool miss = on_det && n_on > aq;
                    on_det = on_det && !miss;
                    on_why = miss ? -3.f : on_why;
                }
                STAMP2(6);  // close-up: Bernoulli
                V3 d_on = rel;
                bool d_on_det = on_det;
                if (HOT(c.o_delay) > 0) {                                                // :576-588 onboard delay ring
                    on_sample = make_float4(rel.x, rel.y, rel.z, on_det ? 1.f : (HAS(HLX_F_RADAR_DEBUG) ? on_why : 0.f));   // w: 1 detected, -reason otherwise
                    d_on = v3(0.f, 0.f, 0.f); d_on_det = false; on_why = -4.f;              // :582 'sensor_delay_initialization'
                    if (!HLX_ONBOARD_EARLY && !fresh_k && steps >= on_delay) {
                        int slot = o_wslot - on_delay;                            // (t - on_delay) mod o_cap
                        slot += (slot < 0) ? o_cap : 0;
                        float4 s = oring[(size_t)slot * N + i];
                        d_on = v3(s.x, s.y, s.z); d_on_det = s.w > 0.f; on_why = d_on_det ? 0.f : s.w;
                    }
                }
                STAMP(8);   // onboard detection + onboard ring
                // ---- ground radar (core.py:368-438)
                bool g_det = false;
                D3 g_pos = d3(0., 0., 0.);
                V3 g_vel = v3(0.f, 0.f, 0.f);
                float g_q = 0.f;
                const V3 gp = v3(HOT(c.ground_pos[0]), HOT(c.ground_pos[1]), HOT(c.ground_pos[2]));
                if (HAS(HLX_F_GROUND)) {
                    V3 g2m = mpos - gp;
                    // fast float32 first; the reference's own operations where a decision is within reach of the fast path's error
                    float grange = fnorm(g2m);
                    if (RARE(fabsf(grange - HOT(c.g_max_range)) < 0.05f)) grange = snorm3(g2m);
                    g_det = !(grange > HOT(c.g_max_range));                              // :396
                    {                                                               // :401-406  arcsin(s) against the elevation window:
                        // s against the float32 values at which the host's asinf crosses the two limits (hlx_host.inc); straight-line
                        const bool chk = g_det && grange > 1e-6f;
                        float se = clampf(g2m.z * __builtin_amdgcn_rcpf(grange), -1.f, 1.f);
                        if (RARE(chk && fminf(fabsf(se - HOT(c.sin_min_elev)), fabsf(se - HOT(c.sin_max_elev))) < 4e-6f))
                            se = clampf(HLX_DIVF(g2m.z, snorm3(g2m)), -1.f, 1.f);
                        g_det = g_det && !(chk && (se < HOT(c.sin_min_elev) || se > HOT(c.sin_max_elev)));
                    }
                    g_det = g_det && !(mpos.z < 50.f);                              // :409
                    float dpq = ((HOT(c.g_base_q) * (1.0f - (grange * HOT(c.inv_g_max_range)) * 0.4f)) * HOT(c.weather)) * cur_g_rel;   // :413-418
                    // The reference's own operations where the Bernoulli draw is within reach of the fast value's error -- and always in
                    // los_frame mode: the quality is also the measurement's WEIGHT in the fusion below (core.py:735-738), an ulp of it is
                    // 3e-8 of the fused position, and the LOS-frame lead angle / LOS rates amplify exactly that when the velocity estimate
                    // or the range is small (found with tools/diag_kf.py: the filter state differed by 2e-8 relative from the first
                    // float64 measurement on).  Elsewhere no output shows it above 4e-6 and the step keeps the cheap form.
                    if (HAS(HLX_F_OBS_LOS) || RARE(g_det && fabsf(n_g - dpq) < 2e-6f))
                        dpq = ((HOT(c.g_base_q) * (1.0f - HLX_DIVF(snorm3(g2m), HOT(c.g_max_range)) * 0.4f)) * HOT(c.weather)) * cur_g_rel;
                    g_det = g_det && !(n_g > dpq);
                    if (g_det) {
                        {
                            // :422-429 float64 measurement (kept float64 through the delay ring: the Kalman
                            // velocity estimate is tiny while detections are continuous, so outputs such as the
                            // lead-angle cosine are sensitive to 1e-4 m of measurement rounding)
                            g_pos = d3((double)rel.x + HOT(c.g_range_acc) * n_gp.x, (double)rel.y + HOT(c.g_range_acc) * n_gp.y,
                                       (double)rel.z + HOT(c.g_range_acc) * n_gp.z);
                            V3 rv = mvel - ivel;
                            g_vel = v3((float)((double)rv.x + HOT(c.g_vel_acc) * n_gv.x), (float)((double)rv.y + HOT(c.g_vel_acc) * n_gv.y),
                                       (float)((double)rv.z + HOT(c.g_vel_acc) * n_gv.z));
                            g_q = dpq;
                        }
                    }
                }
                D3 d_gp64 = g_pos;
                V3 d_gv = g_vel;
                float d_gq = g_q;
                bool d_g_det = g_det, d_g64 = g_det;    // d_g64: the delayed sample is a real (float64) measurement
                if (HAS(HLX_F_GROUND) && HOT(c.g_delay) > 0) {                           // :609-627 ground delay ring
                    g_sp = g_pos; g_sq = g_q; g_sflag = g_det ? 1.f : 0.f;
                    g_s2 = make_float4(g_vel.x, g_vel.y, g_vel.z, 0.f);
                    d_gp64 = d3(0., 0., 0.); d_gv = v3(0.f, 0.f, 0.f); d_gq = 0.f; d_g_det = false; d_g64 = false;
                    if (!fresh_k && steps >= HOT(c.g_delay)) {                             // sample pre-loaded at kernel entry
                        const double2 s0 = gr0;
                        const float4 s1 = gr1, s2 = gr2;
                        d_gp64 = d3(s0.x, s0.y, __hiloint2double(__float_as_int(s1.y), __float_as_int(s1.x)));
                        d_gq = s1.z; d_g64 = s1.w != 0.f; d_gv = v3(s2.x, s2.y, s2.z);
                        d_g_det = g_det;                                            // :626 CURRENT flag (reference quirk)
                    }
                }
                const V3 d_gp = to_v3(d_gp64);
                if (HLX_ONBOARD_EARLY && HOT(c.o_delay) > 0 && on_have) {      // the delayed onboard sample: first touched here
                    asm volatile("" : "+v"(on_raw.x), "+v"(on_raw.y), "+v"(on_raw.z), "+v"(on_raw.w));
                    d_on = v3(on_raw.x, on_raw.y, on_raw.z); d_on_det = on_raw.w > 0.f; on_why = d_on_det ? 0.f : on_raw.w;
                }
#ifndef HLX_EARLY_MORE
#define HLX_EARLY_MORE 0      // A/B: 1 = the ground-ring sample is stored behind the ground-radar section; 2 = the Kalman groups and the packed word behind the filter
#endif
                if ((HLX_EARLY_MORE & 1) && EARLY_ST && !ALLF && pass == 0 && HAS(HLX_F_GROUND) && HOT(c.g_delay) > 0) {
                    const uint32_t bb = (uint32_t)min(64, n - (int)blockIdx.x * 64) * 16u;
                    float4* R = gring + ((size_t)g_wslot * GROUND_RING_WORDS16) * N + (size_t)blockIdx.x * 64;
                    wt16(__builtin_amdgcn_make_buffer_rsrc(R, 0, bb, 0x00020000), lane16, 0u, make_double2(g_sp.x, g_sp.y));
                    wt16(__builtin_amdgcn_make_buffer_rsrc(R + N, 0, bb, 0x00020000), lane16, 0u,
                         make_float4(__int_as_float(__double2loint(g_sp.z)), __int_as_float(__double2hiint(g_sp.z)), g_sq, g_sflag));
                    wt16(__builtin_amdgcn_make_buffer_rsrc(R + 2 * N, 0, bb, 0x00020000), lane16, 0u, g_s2);
                }
                STAMP(9);   // ground radar + ground ring
                // ---- datalink (core.py:440-474)
                float datalink = 0.f;
                if (HAS(HLX_F_GROUND)) {
                    const float lr = fnorm_out(ipos - gp);                           // (pure output; straight-line)
                    const float x = lr * HOT(c.inv_max_datalink);
                    const float vr = fnorm_out(ivel) * 0.001f;
                    const float dop = (0.3f < vr) ? (float)(1.0 - 0.3) : (1.0f - vr);
                    const float dl = clampf(((1.0f - x * x) * dop) * 0.95f, 0.f, 1.f);
                    datalink = (!(lr > HOT(c.max_datalink)) && !(n_dl < HOT(c.packet_loss))) ? dl : 0.f;
                }
                // ---- fusion confidence (core.py:476-509) - pure output
                const float agree = 1.0f - fminf(fnorm_out(d_on - d_gp) * 0.005f, 1.0f);
                const float f_both = clampf((float)(0.35 * HOT(c.radar_quality64)) + 0.50f * d_gq + 0.15f * agree, 0.f, 1.f);
                const float fusion = d_g_det ? (d_on_det ? f_both : d_gq * 0.6f) : (d_on_det ? (float)(HOT(c.radar_quality64) * 0.5) : 0.f);
                // bit 7: a delayed onboard sample exists (core.py:576-593): info['radar_quality'] is the configured quality then,
                // detected or not, and 0.0 only while the delay line is still filling
                // (a reset-only launch reports the detections of the FIRST observation: reset()'s info, environment.py:595-601)
                if (MODE == 1 || !fresh_k)
                    det_bits = (d_on_det ? 32u : 0u) | (d_g_det ? 64u : 0u) | ((HOT(c.o_delay) == 0 || steps >= on_delay) ? 128u : 0u);
                if (!fresh_k) {
                    if (HAS(HLX_F_RADAR_DEBUG) && (slots & (1u << 20)) && HOT(opt.info.radar_debug)) {     // what info['radar_debug'] (core.py:650-683) cannot rebuild from positions
                        auto rd = G(HOT(opt.info.radar_debug)) + i;
                        rd[0] = q.w; rd[N] = q.x; rd[2 * N] = q.y; rd[3 * N] = q.z;
                        rd[4 * N] = d_gq;
                        rd[5 * N] = __int_as_float((int)(-on_why) | (g_det ? 8 : 0));
                        rd[6 * N] = datalink; rd[7 * N] = fusion;
                    }
                }

                STAMP(10);  // datalink + fusion confidence
                // ======================================================== core.py:693-1032 compute()
                // measurement fusion + Kalman filter mirror the reference's dtype flow (float32 until a
                // float64 ground measurement is absorbed, float64 afterwards): core.py:732-774, :91-116
                bool have_track;
                if (d_on_det || d_g_det) {
                    D3 z;
                    bool m64;
                    if (d_on_det && d_g_det) {
                        m64 = d_g64;
                        if (m64) {
                            const double itot = 1.0 / (double)(HOT(c.radar_quality) + d_gq);   // one float64 reciprocal for 3 quotients
                            z = d3(((double)(d_on.x * HOT(c.radar_quality)) + d_gp64.x * (double)d_gq) * itot,
                                   ((double)(d_on.y * HOT(c.radar_quality)) + d_gp64.y * (double)d_gq) * itot,
                                   ((double)(d_on.z * HOT(c.radar_quality)) + d_gp64.z * (double)d_gq) * itot);
                        } else {   // stale zero sample: weight 0.0 (python float) -> float32 arithmetic
                            float total = (float)HOT(c.radar_quality64);
                            z = to_d3(v3((d_on.x * HOT(c.radar_quality)) / total, (d_on.y * HOT(c.radar_quality)) / total,
                                         (d_on.z * HOT(c.radar_quality)) / total));
                        }
                    } else if (d_on_det) { m64 = false; z = to_d3(d_on); }
                    else { m64 = d_g64; z = d_gp64; }
                    // The filter runs in the dtype numpy's promotion gives it: float32 while state and measurement are
                    // float32, float64 from the first float64 (ground) measurement on.  Two code paths instead of
                    // rounding every float64 result back to float32 under a per-lane flag: a float32 operation IS the
                    // float64 operation rounded once more (53 >= 2*24 + 2 bits: double rounding is innocuous for
                    // + - * / sqrt), so the float32 path below is bit-identical to that emulation, and a wave whose
                    // lanes agree on the dtype -- the usual case -- executes one path of plain arithmetic.
                    const D3 zw = d3((double)ipos.x + z.x, (double)ipos.y + z.y, (double)ipos.z + z.z);            // :749
                    const V3 zf = to_v3(zw);                                        // the float32 measurement (when !m64)
                    if (ALLF || RARE(!kf_init)) {                                   // core.py:93-96
                        kxp = to_d3(zf);
                        kxv = d3(0., 0., 0.);
                        kf_init = true;
                    } else {                                                        // core.py:98-116 on the 2x2 blocks
                        const bool y64 = m64 || kf_x64;
                        const float sinv = HLX_DIVF(1.0f, p_pp + 400.f);
                        const float kp = p_pp * sinv, kv = p_vp * sinv;
                        if (y64) {
                            const D3 zz = m64 ? zw : to_d3(zf);
                            const D3 y = zz - kxp;
                            kxp = d3(kxp.x + (double)kp * y.x, kxp.y + (double)kp * y.y, kxp.z + (double)kp * y.z);
                            kxv = d3(kxv.x + (double)kv * y.x, kxv.y + (double)kv * y.y, kxv.z + (double)kv * y.z);
                        } else {
                            const V3 xp = to_v3(kxp), xv = to_v3(kxv);             // exact: the state is float32-valued
                            const V3 y = zf - xp;
                            kxp = to_d3(v3(xp.x + kp * y.x, xp.y + kp * y.y, xp.z + kp * y.z));
                            kxv = to_d3(v3(xv.x + kv * y.x, xv.y + kv * y.y, xv.z + kv * y.z));
                        }
                        kf_x64 = y64;
                        const float omk = 1.0f - kp, nkv = 0.f - kv;
                        const float npp = omk * p_pp, npv = omk * p_pv;
                        const float nvp = nkv * p_pp + p_vp, nvv = nkv * p_pv + p_vv;
                        p_pp = npp; p_pv = npv; p_vp = nvp; p_vv = nvv;
                    }
                    have_track = true;
                } else {                                                            // :760-774
                    if (!ALLF && kf_init) {                                         // core.py:80-89 predict
                        if (kf_x64) {
                            const double dtf = (double)HOT(c.dt);
                            kxp = d3(kxp.x + dtf * kxv.x, kxp.y + dtf * kxv.y, kxp.z + dtf * kxv.z);
                        } else {
                            const V3 xp = to_v3(kxp), xv = to_v3(kxv);
                            kxp = to_d3(v3(xp.x + HOT(c.dt) * xv.x, xp.y + HOT(c.dt) * xv.y, xp.z + HOT(c.dt) * xv.z));
                        }
                        // `F @ P @ F.T` = two OpenBLAS sgemm calls: every element is ONE fused multiply-add chain over k (pinned
                        // against numpy in oracle/hlx_oracle.c kf_predict); with F = [I dt I; 0 I] that is fmaf(dt, b, a).
                        // (The state's `F @ x` above is sgemv / dgemv, which does not fuse.)
                        const float a_pp = __builtin_fmaf(HOT(c.dt), p_vp, p_pp), a_pv = __builtin_fmaf(HOT(c.dt), p_vv, p_pv);
                        const float n_pp = __builtin_fmaf(a_pv, HOT(c.dt), a_pp), n_vp = __builtin_fmaf(p_vv, HOT(c.dt), p_vp);
                        p_pp = n_pp + HOT(c.q11); p_pv = a_pv + HOT(c.q12); p_vp = n_vp + HOT(c.q12); p_vv = p_vv + HOT(c.q22);
                    }
                    have_track = ALLF ? false : kf_init;
                }
                if ((HLX_EARLY_MORE & 2) && EARLY_ST && !ALLF && pass == 0) {
                    const uint32_t pk = (uint32_t)steps | ((uint32_t)worsening << 13) | ((uint32_t)crossed << 25) |
                                        ((uint32_t)kf_init << 26) | ((uint32_t)kf_x64 << 27) | ((uint32_t)on_delay << 28);
                    STG(G_W1, make_float4(__int_as_float(__double2loint(wind.z)), __int_as_float(__double2hiint(wind.z)), __uint_as_float(pk), ep_return));
                    STG(G_KF0, make_double2(kxp.x, kxp.y));
                    STG(G_KF1, make_double2(kxp.z, kxv.x));
                    STG(G_KF2, make_double2(kxv.y, kxv.z));
                    STG(G_KFP, make_float4(p_pp, p_pv, p_vp, p_vv));
                }
                STAMP(11);  // measurement fusion + Kalman filter
                // ---- observation vector: pure outputs, ordinary fast float32 from here on
                const float inv_mr = HOT(c.inv_max_range), inv_mv = HOT(c.inv_max_velocity);
                if (have_track) {                                                   // :778-906
                    // filtered relative position / velocity in the Kalman state's dtype (:758-759,:767-768); the
                    // target-velocity estimate frv + ivel (:861) cancels back to the (small) Kalman velocity, so it
                    // is formed before anything is narrowed to float32
                    // (narrowing the float64 difference once gives the float32 difference too: innocuous double rounding)
                    const D3 frv64 = d3(kxv.x - (double)ivel.x, kxv.y - (double)ivel.y, kxv.z - (double)ivel.z);
                    const V3 frp = v3((float)(kxp.x - (double)ipos.x), (float)(kxp.y - (double)ipos.y), (float)(kxp.z - (double)ipos.z));
                    const V3 frv = to_v3(frv64);
                    const float rrange = fnorm_out(frp);
                    float closing = -fdiv(fdot(frp, frv), rrange + 1e-6f);          // :786
                    // Near closest approach the LOS is perpendicular to the relative velocity and the float32 dot product cancels
                    // (600 m x 1500 m/s terms against a 5 m/s result): time-to-go obs[13] = 1 - range / closing / 100 then carries
                    // 1e-4.  Where |cos| < 0.05 the reference's own arithmetic is replayed -- float64 once the filter state is.
                    if (RARE(closing * closing < 2.5e-3f * fdot(frv, frv))) {
                        if (kf_x64) {
                            const D3 frp64 = d3(kxp.x - (double)ipos.x, kxp.y - (double)ipos.y, kxp.z - (double)ipos.z);
                            closing = (float)(-ddot(frp64, frv64) / (sqrt(ddot(frp64, frp64)) + 1e-6));
                        } else closing = -HLX_DIVF(sdot3(frp, frv), snorm3(frp) + 1e-6f);
                    }
                    if (HAS(HLX_F_OBS_LOS)) {                                       // :791-868
                        row[0] = clampf(rrange * inv_mr, 0.f, 1.f);
                        row[1] = clampf(closing * inv_mv, -1.f, 1.f);
                        V3 lu = (rrange > 1e-6f) ? frp * __builtin_amdgcn_rcpf(rrange) : v3(1.f, 0.f, 0.f);
                        V3 rate = (frv - lu * closing) * __builtin_amdgcn_rcpf(rrange + 1e-6f); // :810-811
                        V3 h, v;
                        los_basis(lu, h, v);
                        row[2] = clampf(fdot(rate, h) * 2.0f, -1.f, 1.f);           // :844-845 (/0.5)
                        row[3] = clampf(fdot(rate, v) * 2.0f, -1.f, 1.f);
                        float ivm = fnorm_out(ivel);
                        row[4] = (ivm > 1e-6f) ? fdiv(fdot(ivel, lu), ivm) : 0.f;    // :852-858
                        V3 tv = kf_x64 ? v3((float)(frv64.x + (double)ivel.x), (float)(frv64.y + (double)ivel.y), (float)(frv64.z + (double)ivel.z))
                                       : v3(frv.x + ivel.x, frv.y + ivel.y, frv.z + ivel.z);   // :861
                        float tvm = fnorm_out(tv);
                        row[5] = (tvm > 1e-6f) ? -fdiv(fdot(tv, lu), tvm) : 0.f;
                        // close in (or with next to no velocity estimate) the fast formulas above no longer hold 1e-5 on the LOS rates
                        // and the lead angle: the reference's own arithmetic, in the reference's dtype (hlx_device.h los_exact*)
                        if (RARE(rrange < 300.f || tvm < 2.f)) {
                            const LosExact e = kf_x64 ? los_exact64(kxp, kxv, ipos, ivel) : los_exact32(frp, frv, ivel);
                            row[2] = e.rate_h; row[3] = e.rate_v; row[5] = e.lead_cos;
                        }
                        row[6] = clampf(ivm * inv_mv, 0.f, 1.f);                    // :924-925
                        row[7] = clampf(fdot(ivel, h) * inv_mv, -1.f, 1.f);         // :948-953
                        row[8] = clampf(fdot(ivel, v) * inv_mv, -1.f, 1.f);
                    } else if (HAS(HLX_F_OBS_BODY)) {                               // :870-876
                        V3 r = right_vec(q), up = up_vec(q);
                        row[0] = clampf(fdot(frp, fwd) * inv_mr, -1.f, 1.f);
                        row[1] = clampf(fdot(frp, r) * inv_mr, -1.f, 1.f);
                        row[2] = clampf(fdot(frp, up) * inv_mr, -1.f, 1.f);
                        row[3] = clampf(fdot(frv, fwd) * inv_mv, -1.f, 1.f);
                        row[4] = clampf(fdot(frv, r) * inv_mv, -1.f, 1.f);
                        row[5] = clampf(fdot(frv, up) * inv_mv, -1.f, 1.f);
                    } else {                                                        // :878-882
                        row[0] = clampf(frp.x * inv_mr, -1.f, 1.f); row[1] = clampf(frp.y * inv_mr, -1.f, 1.f);
                        row[2] = clampf(frp.z * inv_mr, -1.f, 1.f); row[3] = clampf(frv.x * inv_mv, -1.f, 1.f);
                        row[4] = clampf(frv.y * inv_mv, -1.f, 1.f); row[5] = clampf(frv.z * inv_mv, -1.f, 1.f);
                    }
                    row[13] = (closing > 0.f) ? clampf(1.0f - fdiv(rrange, closing) * 0.01f, -1.f, 1.f) : -1.f; // :885-889
                    float tq = clampf(1.0f - ((p_pp + p_pp) + p_pp) * 1e-4f, 0.f, 1.f); // :892-893 trace of 3 equal blocks
                    if (d_on_det) tq *= HOT(c.radar_quality);                            // :894-895
                    row[14] = tq;
                    row[15] = clampf(closing * inv_mv, -1.f, 1.f);                  // :899
                    row[16] = (rrange > 1e-6f) ? fdiv(fdot(fwd, frp), rrange) : 1.0f; // :902-906
                } else {                                                            // :907-917
#pragma unroll
                    for (int k = 0; k < 6; ++k) row[k] = -2.0f;
                    row[13] = -1.0f; row[14] = 0.f; row[15] = 0.f; row[16] = 0.f;
                    if (HAS(HLX_F_OBS_LOS)) { row[6] = clampf(fnorm_out(ivel) * inv_mv, 0.f, 1.f); row[7] = 0.f; row[8] = 0.f; }
                }
                if (HAS(HLX_F_OBS_BODY)) {                                          // :959-962
                    V3 r = right_vec(q), up = up_vec(q);
                    row[6] = clampf(fdot(ivel, fwd) * inv_mv, -1.f, 1.f);
                    row[7] = clampf(fdot(ivel, r) * inv_mv, -1.f, 1.f);
                    row[8] = clampf(fdot(ivel, up) * inv_mv, -1.f, 1.f);
                } else if (!HAS(HLX_F_OBS_LOS)) {                                   // :964
                    row[6] = clampf(ivel.x * inv_mv, -1.f, 1.f); row[7] = clampf(ivel.y * inv_mv, -1.f, 1.f);
                    row[8] = clampf(ivel.z * inv_mv, -1.f, 1.f);
                }
                if (HAS(HLX_F_OBS_BODY) || HAS(HLX_F_OBS_LOS)) { row[9] = 0.f; row[10] = 0.f; row[11] = 0.f; } // :967-970
                else {                                                              // :973-974, core.py:1103-1121
                    const float inv_pi = 0.3183098861837907f;
                    row[9] = fast_atan2(2.f * (q.w * q.x + q.y * q.z), 1.f - 2.f * (q.x * q.x + q.y * q.y)) * inv_pi;
                    row[10] = fast_asin(clampf(2.f * (q.w * q.y - q.z * q.x), -1.f, 1.f)) * inv_pi;
                    row[11] = fast_atan2(2.f * (q.w * q.z + q.x * q.y), 1.f - 2.f * (q.y * q.y + q.z * q.z)) * inv_pi;
                }
                row[12] = clampf(fuel * 0.01f, 0.f, 1.f);                           // :977
                if (d_g_det && datalink > 0.1f) {                                   // :980-1018
                    if (HAS(HLX_F_OBS_LOS)) {
                        float gr = fnorm_out(d_gp);
                        float gc = -fdiv(fdot(d_gp, d_gv), gr + 1e-6f);
                        row[17] = clampf(gr * inv_mr, 0.f, 1.f);
                        row[18] = clampf(gc * inv_mv, -1.f, 1.f);
                        row[19] = (gr > 1e-6f) ? clampf(fdiv(fnorm_out(d_gv - d_gp * fdiv(gc, gr)), gr) * 2.0f, 0.f, 1.f) : 0.f;
                        row[20] = 0.f; row[21] = 0.f; row[22] = 0.f;
                    } else if (HAS(HLX_F_OBS_BODY)) {
                        V3 r = right_vec(q), up = up_vec(q);
                        row[17] = clampf(fdot(d_gp, fwd) * inv_mr, -1.f, 1.f); row[18] = clampf(fdot(d_gp, r) * inv_mr, -1.f, 1.f);
                        row[19] = clampf(fdot(d_gp, up) * inv_mr, -1.f, 1.f); row[20] = clampf(fdot(d_gv, fwd) * inv_mv, -1.f, 1.f);
                        row[21] = clampf(fdot(d_gv, r) * inv_mv, -1.f, 1.f); row[22] = clampf(fdot(d_gv, up) * inv_mv, -1.f, 1.f);
                    } else {
                        row[17] = clampf(d_gp.x * inv_mr, -1.f, 1.f); row[18] = clampf(d_gp.y * inv_mr, -1.f, 1.f);
                        row[19] = clampf(d_gp.z * inv_mr, -1.f, 1.f); row[20] = clampf(d_gv.x * inv_mv, -1.f, 1.f);
                        row[21] = clampf(d_gv.y * inv_mv, -1.f, 1.f); row[22] = clampf(d_gv.z * inv_mv, -1.f, 1.f);
                    }
                    row[23] = d_gq;
                } else {                                                            // :1019-1024
#pragma unroll
                    for (int k = 17; k < 23; ++k) row[k] = -2.0f;
                    row[23] = 0.f;
                }
                row[24] = datalink;                                                 // :1027
                row[25] = fusion;                                                   // :1030
                STAMP(12);  // 26-D observation formulas -> LDS row
            }
            again = !ALLF && pass == 0 && !single && __ballot(done) != 0ull;
            ++pass;
        };
        trip(std::false_type{});
        // hlx_info_soa.packed word 2: the missile as the step left it (a respawn in the second trip replaces it) + the flag word,
        // complete now that pass 0 has decided the detections
        if (MODE == 0 && HLX_INFO_W2 == 0 && (slots & (1u << 23)) != 0u)
            wt16i(rsI, (uint32_t)i * 16u, (uint32_t)n * 32u, make_float4(mpos.x, mpos.y, mpos.z, __uint_as_float(info_word | det_bits)));
        // reset()'s info (environment.py:595-601: missile_pos, interceptor_pos, distance, radar_detected, radar_quality) in the words a
        // step writes, for the environments this launch resets -- hlx_reset_info
        if (MODE == 1 && (slots & (1u << 23)) != 0u && done) {
            const __amdgpu_buffer_rsrc_t rsR = __builtin_amdgcn_make_buffer_rsrc((void*)HOT(opt.info.packed), 0, 0x7FFFFFFF, 0x00020000);
            const uint32_t alive = HAS(HLX_F_VOLLEY) ? (uint32_t)VK : 1u;      // nothing intercepted, every missile of the volley in flight
            wt16i(rsR, (uint32_t)i * 16u, 0u, make_float4(prev_distance, min_distance, fuel, 0.f));
            wt16i(rsR, (uint32_t)i * 16u, (uint32_t)n * 16u, make_float4(ipos.x, ipos.y, ipos.z, __int_as_float(0)));
            wt16i(rsR, (uint32_t)i * 16u, (uint32_t)n * 32u, make_float4(mpos.x, mpos.y, mpos.z, __uint_as_float(det_bits | (alive << 12))));
        }
#if HLX_FRESH_TRIP
        if (RARE(again)) trip(std::true_type{});
#else
        if (RARE(again)) trip(std::false_type{});
#endif

        STAMP(13);  // observation passes (incl. loop exit)
        // ---------------------------------------------------------------------- store state + rings
        if (MODE == 0 || done) {
            packed = (uint32_t)steps | ((uint32_t)worsening << 13) | ((uint32_t)crossed << 25) |
                     ((uint32_t)kf_init << 26) | ((uint32_t)kf_x64 << 27) | ((uint32_t)on_delay << 28);
            // fused rollout: the state stays in the registers it was loaded into; otherwise it goes back to the arena
#define PUT4(G, reg, ...) do { if (PERSIST) reg = __VA_ARGS__; else STG(G, __VA_ARGS__); } while (0)
#define PUT2(G, reg, ...) do { if (PERSIST) reg = __VA_ARGS__; else STG(G, __VA_ARGS__); } while (0)
            if (!EARLY_ST || RARE(done && !(LONE && HLX_COOP_RESPAWN && ((coop_mask >> lane) & 1ull) != 0ull))) {
            PUT4(G_IPOS, g_ipos, make_float4(ipos.x, ipos.y, ipos.z, fuel));
            PUT4(G_IVEL, g_ivel, make_float4(ivel.x, ivel.y, ivel.z, prev_distance));
            PUT4(G_QUAT, g_quat, make_float4(q.w, q.x, q.y, q.z));
            PUT4(G_MPOS, g_mpos, make_float4(mpos.x, mpos.y, mpos.z, min_distance));
            PUT4(G_MVEL, g_mvel, make_float4(mvel.x, mvel.y, mvel.z, last_distance));
            PUT2(G_W0, g_w0, make_double2(wind.x, wind.y));
            if (PERSIST) g_fu = fuel_used; else if (!HLX_AB_AUX_LATE) STAUX(fuel_used);
            }
            if (HLX_AB_AUX_LATE && !PERSIST) STAUX(fuel_used);
            if (HAS(HLX_F_THRUST_LAG) || HAS(HLX_F_DOMAIN_RAND)) PUT4(G_THRUST, g_thr, make_float4(thrust_act.x, thrust_act.y, thrust_act.z, dp.cd_super));
            if (!((HLX_EARLY_MORE & 2) && EARLY_ST) || RARE(done)) {
            PUT4(G_W1, g_w1, make_float4(__int_as_float(__double2loint(wind.z)), __int_as_float(__double2hiint(wind.z)),
                                         __uint_as_float(packed), ep_return));
            PUT2(G_KF0, g_kf0, make_double2(kxp.x, kxp.y));
            PUT2(G_KF1, g_kf1, make_double2(kxp.z, kxv.x));
            PUT2(G_KF2, g_kf2, make_double2(kxv.y, kxv.z));
            PUT4(G_KFP, g_kfp, make_float4(p_pp, p_pv, p_vp, p_vv));
            }
            if (HAS(HLX_F_DOMAIN_RAND)) PUT4(G_MISC, g_misc, make_float4(__int_as_float(__double2loint(T0)), __int_as_float(__double2hiint(T0)), dp.base_cd, dp.peak_m1));
            if (HAS(HLX_F_VOLLEY)) {
#pragma unroll
                for (int k = 0; k < HLX_MAX_VOLLEY; ++k) {
                    if (k < VK) {
                        const uint32_t w = (vact[k] ? 1u : 0u) | (k == 0 ? ((uint32_t)prio << 8) | ((uint32_t)n_int << 12) : 0u);
                        STG(G_VPOS + k, make_float4(vp[k].x, vp[k].y, vp[k].z, vmin[k]));
                        STG(G_VVEL + k, make_float4(vv[k].x, vv[k].y, vv[k].z, __uint_as_float(w)));
                    }
                }
            }
#undef PUT4
#undef PUT2
            if (MODE == 2) {
                // pool fill: the samples the first observation would push into the rings, its row, and last the tag that says
                // which episode of this environment the entry holds (read by step launches that follow on the stream)
                pool_put(PA2, PG_ON, on_sample);
                pool_put(PA2, PG_GR, make_double2(g_sp.x, g_sp.y));
                pool_put(PA2, PG_GR + 1, make_float4(__int_as_float(__double2loint(g_sp.z)), __int_as_float(__double2hiint(g_sp.z)), g_sq, g_sflag));
                pool_put(PA2, PG_GR + 2, g_s2);
#pragma unroll
                for (int k = 0; k < (HLX_OBS_DIM + 3) / 4; ++k)      // (the two floats behind the 26-float row: the beam test's cosine and threshold)
                    pool_put(PA2, PG_ROW + k, make_float4(row[4 * k], row[4 * k + 1], 4 * k + 2 < HLX_OBS_DIM ? row[4 * k + 2] : beam_cb,
                                                         4 * k + 3 < HLX_OBS_DIM ? row[4 * k + 3] : cur_cos_half_beam));
                G(pp2.tag)[i] = pool_epn;
            }
            const uint32_t blk_bytes = (uint32_t)min(64, n - (int)blockIdx.x * 64) * 16u;   // partial tail block: clip
            if (MODE != 2 && HOT(c.o_delay) > 0) {
                const __amdgpu_buffer_rsrc_t rsO = __builtin_amdgcn_make_buffer_rsrc(
                    oring + (size_t)o_wslot * N + (size_t)blockIdx.x * 64, 0, blk_bytes, 0x00020000);
                wt16(rsO, lane16, 0u, on_sample);
            }
            if (MODE != 2 && HAS(HLX_F_GROUND) && HOT(c.g_delay) > 0 && (!((HLX_EARLY_MORE & 1) && EARLY_ST) || RARE(done))) {
                // the slot's three planes are N words apart: one descriptor per plane keeps every offset 32-bit at any N
                float4* R = gring + ((size_t)g_wslot * GROUND_RING_WORDS16) * N + (size_t)blockIdx.x * 64;
                wt16(__builtin_amdgcn_make_buffer_rsrc(R, 0, blk_bytes, 0x00020000), lane16, 0u, make_double2(g_sp.x, g_sp.y));
                wt16(__builtin_amdgcn_make_buffer_rsrc(R + N, 0, blk_bytes, 0x00020000), lane16, 0u,
                     make_float4(__int_as_float(__double2loint(g_sp.z)), __int_as_float(__double2hiint(g_sp.z)), g_sq, g_sflag));
                wt16(__builtin_amdgcn_make_buffer_rsrc(R + 2 * N, 0, blk_bytes, 0x00020000), lane16, 0u, g_s2);
            }
        }
        if (LONE && HLX_COOP_RESPAWN && RARE(coop_mask != 0ull)) {
            // ---- the wave stores the prepared episodes it holds: word `lane` of each entry goes where the finished lane's own final
            // stores (just issued, with the OLD episode's values) put that word.  One descriptor over the whole allocation: arena
            // blocks, ground ring and onboard ring all lie behind `arena` (hlx_host.inc), offsets fit 32 bits at the batch sizes the
            // lone-wave schedule serves.
            const __amdgpu_buffer_rsrc_t rsAll = __builtin_amdgcn_make_buffer_rsrc(arena, 0, 0x7FFFFFFF, 0x00020000);
            const uint32_t blocks = (uint32_t)((n + 63) >> 6);
            const uint32_t gring_off = blocks * (uint32_t)(N_GROUPS * 1024), oring_off = gring_off + (uint32_t)(g_planes * GROUND_RING_WORDS16) * (uint32_t)n * 16u;
            const int g = lane;
            const bool st_word = g <= G_KFP || (g == G_THRUST && (HAS(HLX_F_THRUST_LAG) || HAS(HLX_F_DOMAIN_RAND))) || (g == G_MISC && HAS(HLX_F_DOMAIN_RAND)) ||
                                 (g >= G_VPOS && g < N_GROUPS && HAS(HLX_F_VOLLEY) && ((g - G_VPOS) % HLX_MAX_VOLLEY) < VK);
            const bool on_word = g == PG_ON && HOT(c.o_delay) > 0;
            const bool gr_word = g >= PG_GR && g < PG_ROW && HAS(HLX_F_GROUND) && HOT(c.g_delay) > 0;
            const bool row_word = g >= PG_ROW && g < POOL_GROUPS;
            unsigned long long mm = coop_mask;
#pragma unroll
            for (int k = 0; k < COOP_MAX; ++k) {
                if (mm != 0ull) {
                    const uint32_t d = (uint32_t)__builtin_ctzll(mm);
                    mm &= mm - 1ull;
                    const uint32_t e = (uint32_t)blockIdx.x * 64u + d;      // the finished environment
                    uint32_t off = 0u;
                    if (st_word) off = (uint32_t)blockIdx.x * (uint32_t)(N_GROUPS * 1024) + (uint32_t)g * 1024u + d * 16u;
                    if (on_word) off = oring_off + ((uint32_t)o_wslot * (uint32_t)n + e) * 16u;
                    if (gr_word) off = gring_off + (((uint32_t)g_wslot * GROUND_RING_WORDS16 + (uint32_t)(g - PG_GR)) * (uint32_t)n + e) * 16u;
                    if (st_word || on_word || gr_word) wt16(rsAll, off, 0u, coop[k]);
                    if (row_word) {      // the entry's observation row -> the finished lane's row of the tile (26 floats: the last word holds two)
                        float* r = tile + d * HLX_OBS_DIM + 4 * (g - PG_ROW);
                        r[0] = coop[k].x; r[1] = coop[k].y;
                        if (4 * (g - PG_ROW) + 2 < HLX_OBS_DIM) { r[2] = coop[k].z; r[3] = coop[k].w; }
                    }
                }
            }
            // the one piece of state an entry does not hold: a new episode has used no fuel (the lane's own final stores, which
            // would have carried it, are skipped for a copied episode)
            if (((coop_mask >> lane) & 1ull) != 0ull) STAUX(0.f);
        }
        if (MODE == 0) {
            if (RARE((slots & (1u << 20)) && !(slots & (1u << 23))) && HOT(opt.info.flags))
                G(HOT(opt.info.flags))[i] = (uint8_t)(info_word | det_bits);
            if (HLX_INFO_W2 == 1 && (slots & (1u << 23)) != 0u)      // the flag dword of word 2 once more, detection bits included
                __builtin_amdgcn_raw_buffer_store_b32(info_word | det_bits, rsI, (uint32_t)i * 16u + 12u, (uint32_t)n * 32u, HLX_INFO_AUX);
        }
        }   // if (live)
    }

    STAMP(14);      // state / ring / scalar outputs stored
#define HLX_COMPACTION_BLOCK \
        if (MODE == 0 && !PERSIST && done_idx_out) { \
            const unsigned long long m = __ballot(live && done); \
            if (m) { \
                int base = 0; \
                if (lane == 0) base = __hip_atomic_fetch_add(G(P->done_cnt) + (int)(t & 1ull), __popcll(m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
                base = __shfl(base, 0); \
                if (live && done) G(done_idx_out)[base + __popcll(m & ((1ull << lane) - 1ull))] = i; \
            } \
        } \

    // -------------------------------------------------------------------------- done-mask compaction
    if (!HLX_DONE_LATE) { HLX_COMPACTION_BLOCK }
    // Every step launch arms the counter of the NEXT vec step, whether or not this one compacts: listed and unlisted
    // steps (hlx_step without done_idx, hlx_rollout, the fused rollout) may interleave freely.
    if (MODE == 0 && !PERSIST && blockIdx.x == 0 && lane == 0) G(P->done_cnt)[(int)((t + 1ull) & 1ull)] = 0;

    // -------------------------------------------------------------------------- observation tile -> [N][26]
    if (obs_out) {
        WAVE_LDS_SYNC();
        const int rows = min(64, n - blockIdx.x * 64);
        float* dst = obs_out + (size_t)blockIdx.x * 64 * HLX_OBS_DIM;
        if (MODE == 0 && !PERSIST && RARE(slots & (1u << 22))) {
            // The observation pipeline rides on this launch (hlx_obs_step): this block's partial sums of what VecNormalize reduces
            // over the batch -- the float64 column sums / sums of squares of the NEWEST frame, which is the tile in LDS right now
            // (rows of finished lanes hold the first observation of their new episode, exactly the frame the pipeline stacks),
            // and of the discounted returns (VecNormalize._update_reward: returns = returns * gamma + reward).  Fixed summation
            // order, no atomics: the pipeline's finalize kernel adds the blocks' partials in block order.
            // (layout [55][blocks], i.e. one contiguous run of `blocks` doubles per quantity: the finalize kernel reads each of them
            // with coalesced 8-byte loads -- rows of 2 F + 4 doubles per block, as until round 3, made every one of its loads a
            // cache line of its own)
            double* const out = HOT(opt.pipe_partial) + (size_t)blockIdx.x;
            const size_t PB = (size_t)HOT(opt.pipe_stride);        // = number of 64-environment blocks of this launch
            const int c = lane & 31, half = lane >> 5;                  // lanes 0-25: rows 0-31 of column c; lanes 32-57: rows 32-63
            double cs = 0., cq = 0.;
            if (c < HLX_OBS_DIM) {
                const int r_end = min(32 * half + 32, rows);
#pragma unroll 8
                for (int r = 32 * half; r < r_end; ++r) {
                    const double x = (double)tile[r * HLX_OBS_DIM + c];
                    cs += x; cq += x * x;
                }
            }
            cs += __shfl_down(cs, 32); cq += __shfl_down(cq, 32);
            if (lane < HLX_OBS_DIM) {
                G(out)[(size_t)(2 * lane) * PB] = cs;
                G(out)[(size_t)(2 * lane + 1) * PB] = cq;
            }
            double rs = 0., rq = 0.;
            if (HOT(opt.pipe_returns)) {
                if (live) {
                    const double ret = pipe_ret_prev * HOT(opt.pipe_gamma) + (double)pipe_reward;
                    G(HOT(opt.pipe_returns))[i] = ret;
                    rs = ret; rq = ret * ret;
                }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { rs += __shfl_down(rs, off); rq += __shfl_down(rq, off); }
            }
            const unsigned long long fin = __ballot(live && done);        // who finished: the finalize kernel subtracts their older frames
            if (lane == 0) {
                G(out)[(size_t)(2 * HLX_OBS_DIM) * PB] = rs; G(out)[(size_t)(2 * HLX_OBS_DIM + 1) * PB] = rq;
                G(out)[(size_t)(2 * HLX_OBS_DIM + 2) * PB] = __longlong_as_double((long long)fin);
            }
        }
        if (MODE == 0) {
            if (rows == 64) {
                const float4* src4 = reinterpret_cast<const float4*>(tile);
                const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 64 * HLX_OBS_DIM * 4, 0x00020000);
                float4 v[7];   // all LDS reads first (one wait), then the stores
#pragma unroll
                for (int r = 0; r < 6; ++r) v[r] = src4[lane + 64 * r];
                v[6] = (lane < 64 * HLX_OBS_DIM / 4 - 384) ? src4[lane + 384] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int r = 0; r < 6; ++r) wt16(rsT, lane16, 1024u * r, v[r]);
                if (lane < 64 * HLX_OBS_DIM / 4 - 384) wt16(rsT, lane16, 6144u, v[6]);
            } else {
                for (int j = lane; j < rows * HLX_OBS_DIM; j += 64) dst[j] = tile[j];
            }
        } else if (live && done) {
            for (int k = 0; k < HLX_OBS_DIM; ++k) dst[lane * HLX_OBS_DIM + k] = row[k];
        }
    }
    if (HLX_DONE_LATE) { HLX_COMPACTION_BLOCK }
    STAMP(15);      // observation tile stored
    if (PERSIST) {   // next step of the fused rollout: clock, ring slots (mod their capacities), I/O cursors
        t += 1ull;
        g_rslot = (g_rslot + 1 >= g_planes) ? 0 : g_rslot + 1;
        g_wslot = (g_wslot + 1 >= g_planes) ? 0 : g_wslot + 1;
        o_wslot = (o_wslot + 1 >= o_planes) ? 0 : o_wslot + 1;
        actions += (size_t)n * HLX_ACT_DIM;
        oslot = (oslot + 1 >= out_slots) ? 0 : oslot + 1;
        if (obs_out0) obs_out = obs_out0 + (size_t)oslot * n * HLX_OBS_DIM;
        reward_out = reward_out0 + (size_t)oslot * n; term_out = term_out0 + (size_t)oslot * n; trunc_out = trunc_out0 + (size_t)oslot * n;
        done = false;
        WAVE_LDS_SYNC();   // the tile is rewritten by the next step
    }
    }   // step loop
    if (PERSIST && blockIdx.x == 0 && lane == 0) { G(P->done_cnt)[0] = 0; G(P->done_cnt)[1] = 0; }   // nothing was listed in this launch
    if (PERSIST && live) {   // state back to the arena, once
        STG(G_IPOS, g_ipos); STG(G_IVEL, g_ivel); STG(G_QUAT, g_quat); STG(G_MPOS, g_mpos); STG(G_MVEL, g_mvel);
        STG(G_W0, g_w0); STG(G_W1, g_w1); STAUX(g_fu);
        if (HAS(HLX_F_THRUST_LAG) || HAS(HLX_F_DOMAIN_RAND)) STG(G_THRUST, g_thr);
        STG(G_KF0, g_kf0); STG(G_KF1, g_kf1); STG(G_KF2, g_kf2); STG(G_KFP, g_kfp);
        if (HAS(HLX_F_DOMAIN_RAND)) STG(G_MISC, g_misc);
    }
}

}  // namespace

// ---- Launchers: the one host-side door to an instantiation of the step kernel, and what lets the library be compiled in parallel.
// The kernel template above has ~90 instantiations (variant x mode x load schedule x baked preset) and one translation unit compiles
// them one after the other: three minutes.  build.py therefore compiles this file several times at once:
//   -DHLX_TU_PART=k : the kernel code and the explicit instantiations of the launchers csrc/hlx_inst_gen.h lists for part k, nothing else;
//   -DHLX_TU_HOST   : everything else (the C ABI, the observation pipeline, the HRL controller), with those launchers declared
//                     `extern template` -- a launcher the list does not name is instantiated here as always, so the list is a hint about
//                     where to compile, never about what exists (build.py regenerates it from every library it has built).
// Neither macro: the single translation unit of rounds 1-4 (A/B builds, fallback).  The kernels stay in the unnamed namespace: their
// names, and everything hotcheck.py and the tools read off the code objects, are the same in all three forms.
namespace hlx_launchers {
struct StepArgs {
    float4* arena; const KParams* P; const float* actions; unsigned long long t, seed; long long env_offset; int n; uint32_t slots;
    float* obs; float* reward; uint8_t* term; uint8_t* trunc; float radius, cos_half_beam, on_rel, g_rel; int T, out_slot0, out_slots;
};
template <uint32_t SPEC, int MODE, bool NOISE, bool PERSIST, int LATE, int BAKE>
void hlx_launch(dim3 grid, hipStream_t s, const StepArgs& a) {
    hipLaunchKernelGGL((hlx_env_kernel<SPEC, MODE, NOISE, PERSIST, LATE, BAKE>), grid, dim3(64), 0, s, a.arena, a.P, a.actions, a.t, a.seed,
                       a.env_offset, a.n, a.slots, a.obs, a.reward, a.term, a.trunc, a.radius, a.cos_half_beam, a.on_rel, a.g_rel, a.T,
                       a.out_slot0, a.out_slots);
}
}  // namespace hlx_launchers

#if defined(HLX_TU_PART)
#define HLX_INST(S, M, N, P_, L, B) template void hlx_launchers::hlx_launch<S, M, N, P_, L, B>(dim3, hipStream_t, const hlx_launchers::StepArgs&);
#include "hlx_inst_gen.h"
#undef HLX_INST
#else
#if defined(HLX_TU_HOST)
#define HLX_INST(S, M, N, P_, L, B) extern template void hlx_launchers::hlx_launch<S, M, N, P_, L, B>(dim3, hipStream_t, const hlx_launchers::StepArgs&);
#include "hlx_inst_gen.h"
#undef HLX_INST
#endif
// host side of the C ABI
#include "hlx_host.inc"
// on-device VecFrameStack + VecNormalize behind the step (include/hlx_obs.h)
#include "hlx_obs.inc"
// on-device HRL controller logic (include/hlx_hrl.h)
#include "hlx_hrl.inc"
#endif
