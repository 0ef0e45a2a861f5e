// hlx_kargs.h -- kernel argument block shared by the kernels and the host side of the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hlx.h"
#include "hlx_kcfg.h"

namespace hlx {

// internal flag: the kernel reads the feature flags at run time instead of having them folded
constexpr uint32_t KF_DYNAMIC = 1u << 31;
// flags that change generated code (the curriculum flags are host-only)
constexpr uint32_t KF_CODEGEN_MASK = HLX_F_ATMOSPHERE | HLX_F_MACH_DRAG | HLX_F_ENH_WIND | HLX_F_THRUST_LAG |
                                     HLX_F_DOMAIN_RAND | HLX_F_VALIDATION | HLX_F_EVASION | HLX_F_PRECISION |
                                     HLX_F_PROX_FUZE | HLX_F_GROUND | HLX_F_SPHERICAL | HLX_F_TOWARD_MISSILE |
                                     HLX_F_OBS_BODY | HLX_F_OBS_LOS | HLX_F_VOLLEY | HLX_F_RADAR_DEBUG;

// State arena: 16-byte groups, blocked struct-of-arrays: arena[env / 64][group][env % 64].  Every load/store
// of a group is one 16-byte-per-lane, 1-KiB-per-wave coalesced access, and a wave's whole state is one
// contiguous N_GROUPS KiB chunk.
enum : int {
    G_IPOS = 0,  // float4: interceptor position xyz, fuel
    G_IVEL,      // float4: interceptor velocity xyz, prev_distance
    G_QUAT,      // float4: orientation w x y z
    G_MPOS,      // float4: missile position xyz, episode min distance
    G_MVEL,      // float4: missile velocity xyz, last_distance (smart early termination)
    G_W0,        // double2: wind x, y        (float64 state in the reference's simple-wind mode)
    G_W1,        // {double wind z, uint32 packed, float episode return}
                 //   packed = steps:13 | worsening:12 | crossed:1 | kf_init:1 | kf_x_is64:1 | onboard_delay:4
    G_KF0,       // double2: Kalman position estimate x, y
    G_KF1,       // double2: Kalman position z, velocity x
    G_KF2,       // double2: Kalman velocity y, z
    G_KFP,       // float4: covariance block p_pp, p_pv, p_vp, p_vv
    G_THRUST,    // float4: actual thrust xyz (thrust lag), F(base_cd * supersonic_multiplier)   [thrust lag / domain randomisation]
    G_MISC,      // {double T0, float F(base_cd), float F(peak multiplier - 1)}   [domain randomisation only]
#ifndef HLX_AB_AUX_ALIAS
    G_AUX,       // a plane of DWORDS, not of 16-byte words: dword `lane` = float total_fuel_used (environment.py:886) -- 4 bytes per
                 // environment read and written per step (256 B per wave); the rest of the 1 KiB slot is unused address space
#endif
    G_VPOS,      // float4 x HLX_MAX_VOLLEY: volley missile k position xyz, its minimum distance        [volley only]
    G_VVEL = G_VPOS + HLX_MAX_VOLLEY,   // float4 x HLX_MAX_VOLLEY: velocity xyz, bits: 0 active | 8-9 priority index |
                                        // 12-14 missiles intercepted (the last two in missile 0's word only)
#ifdef HLX_AB_PAD_GROUPS       // timing experiments only: extra (unused) groups, i.e. another block stride
    N_GROUPS = G_VVEL + HLX_MAX_VOLLEY + HLX_AB_PAD_GROUPS
#else
    N_GROUPS = G_VVEL + HLX_MAX_VOLLEY
#endif
};
#ifdef HLX_AB_AUX_ALIAS            // timing experiments only: the dword plane shares the last volley group (block stride as in round 3)
constexpr int G_AUX = G_VVEL + HLX_MAX_VOLLEY - 1;
#endif
// ground ring slot: double2 {rel_pos x, y}, {double rel_pos z, float quality, float sample-was-a-detection}, float4 {rel_vel xyz, pad}
constexpr int GROUND_RING_WORDS16 = 3;
// Next-episode pool entry (hlx_kernels.hip, "next-episode pool"): the state groups as a respawned lane would store them, the
// samples its first observation pushes into the two delay rings, and that observation's row.  Blocked like the arena:
// pool[env / 64][group][env % 64].
enum : int {
    PG_ON = N_GROUPS,          // float4: onboard ring sample
    PG_GR = PG_ON + 1,         // 3 x 16 bytes: ground ring sample (the ring slot's three words)
    PG_ROW = PG_GR + GROUND_RING_WORDS16,   // 7 x float4: the 26-float observation row (last two floats unused)
    POOL_GROUPS = PG_ROW + (HLX_OBS_DIM + 3) / 4
};
// bytes behind the rings in the arena allocation, for `blocks` 64-environment blocks: pool | tag u32[] | episode u32[] |
// int32[4] counters | u64[blocks] masks of the entries used since the last fill
constexpr int HLX_POOL_INTERVAL_DEFAULT = 128;    // step launches between two pool fills
constexpr int HLX_POOL_QUIET_STEPS = 16;          // step launches a moving sensor reliability must have stood still before the pool is filled again
constexpr size_t pool_aux_words16(size_t blocks) { return blocks * 64 * POOL_GROUPS + blocks * 32 + 1 + (blocks + 1) / 2; }

// ---------------------------------------------------------------------------------------------------
// Kernel arguments.  The kernarg segment of a launch is freshly written memory: a scalar load from it misses
// every cache (~2-3k cycles), and a wave that is alone on its SIMD eats all of it (round-1 stamps: 12-20 % of
// the wave lifetime went before the first load was issued).  So:
//   * the kernel's first 14 parameter dwords -- arena, P, actions, clock, seed, env offset, n, ring slots -- are
//     PRELOADED into SGPRs by the dispatcher (-amdgpu-kernarg-preload-count=16): no scalar load stands between
//     wave start and the state/action/ring loads or the Philox block; the output pointers follow in an ordinary
//     kernarg tail whose load is issued at entry and first needed when results are stored;
//   * everything that is fixed for the handle, or changes rarely (optional output buffers, curriculum scalars),
//     lives in ONE device-resident block *P that stays hot in L2 and is re-uploaded only when it changes.
// ---------------------------------------------------------------------------------------------------
struct KOpt {              // optional caller buffers (re-uploaded only when the pointers change)
    float* terminal_obs;
    int32_t* done_idx;
    hlx_info_soa info;
    const double* step_noise;    // parity mode: slot-major [HLX_STEP_SLOTS][N] float64 unit draws (NULL = Philox)
    const double* reset_noise;   // [HLX_RESET_SLOTS][N]
    const uint8_t* reset_mask;
    // observation pipeline riding on the step (hlx_obs_step, include/hlx_obs.h): the step kernel leaves, per 64-environment block,
    // the float64 column sums / sums of squares of the NEWEST frame (its own observation tile) and advances the discounted returns,
    // so that the pipeline's moments pass over the batch disappears.  NULL = no pipeline attached to this launch.
    double* pipe_partial;        // [2 D + 3][blocks]: feature f of the newest frame at rows 2 f (sum), 2 f + 1 (sum of squares); the return
                                 // sums at rows 2 D, 2 D + 1; the block's done mask (64-bit ballot, as a bit pattern) at row 2 D + 2
    double* pipe_returns;        // [N] discounted returns (NULL: not updated this step)
    double pipe_gamma;
    int32_t pipe_stride, pipe_col0;   // the number of 64-environment blocks (the row length above); first column of the newest frame, (n_stack - 1) * D
};
struct KCur {              // (until round 3: the curriculum scalars in force.  They are kernel arguments now -- hlx_kernels.hip -- and
    double reserved0;      // this block keeps its size so that the offsets of KOpt, which the baked tables do not cover, stay put)
    float reserved1[4];
};
// The hot part {c, cur, opt} is at most 512 B = 128 dwords: at kernel entry lane l loads dwords l and 64 + l
// (two coalesced vector loads, in flight with the state loads), and every constant is then a v_readlane away --
// no scalar load, hence no s_waitcnt, anywhere on the per-step path (scalar loads return out of order, so each
// lazily placed one costs a full `s_waitcnt lgkmcnt(0)` round trip that the lone wave of a SIMD cannot hide;
// round-1 stamps attributed about a third of the wave lifetime to some thirty of them).
struct KHot {
    KCfg c;
    KCur cur;
    KOpt opt;
};
static_assert(sizeof(KHot) <= 512, "hot parameter block must fit two dwords per lane");
static_assert(sizeof(KHot) % 4 == 0, "hot parameter block is read dword-wise");
struct KParams {           // device-resident
    KHot hot;
    char hot_pad[512 - sizeof(KHot)];   // the entry loads always read 512 B
    KCold cold;
    float4* gring;         // [g_delay+1][GROUND_RING_WORDS16][N]
    float4* oring;         // [o_cap][N]
    int32_t* done_cnt;     // [2], double-buffered by vec-step parity
    unsigned long long* stamps;   // diagnostic builds only (-DHLX_STAMPS): [blocks][16] s_memtime samples
    long long env_offset;
    uint32_t seed_lo, seed_hi;
    int32_t n;
};

}  // namespace hlx
