"""The shipped code object must never reload a hot-constant register from scratch (hlynr_intercept_amd/hotcheck.py:
a spill reload under a partial EXEC mask leaves stale lanes that a later v_readlane would hand out as a constant)."""
import os

from hlynr_intercept_amd import build, hotcheck

SPILLED = """
_ZN12_GLOBAL__N_114hlx_env_kernelILj1ELi0ELb0EEEvv:
\tglobal_load_dword v5, v1, s[4:5]
\tv_readlane_b32 s0, v5, 3
\tscratch_store_dword off, v5, off offset:4
\tv_mov_b32_e32 v5, 0
\tscratch_load_dword v5, off, off offset:4
\tv_readlane_b32 s1, v5, 7
\tv_writelane_b32 v9, s3, 0
\tv_readlane_b32 s3, v9, 0
\t.end_amdhsa_kernel
"""


def test_checker_flags_a_spilled_hot_word_and_ignores_sgpr_spill_registers():
    n, fail, _ = hotcheck.check(SPILLED)
    assert n == 1 and [(r, ops) for _, r, ops in fail] == [
        ("v5", {"v_readlane of a reload": 1}), ("v5", {"hot word read by scratch_store_dword (only v_readlane_b32 may)": 1})]
    clean = SPILLED.replace("\tscratch_load_dword v5, off, off offset:4\n", "").replace("\tv_mov_b32_e32 v5, 0\n", "") \
                   .replace("\tscratch_store_dword off, v5, off offset:4\n", "")
    n, fail, _ = hotcheck.check(clean)
    assert n == 1 and not fail
    # reading a COPY of the hot word is refused too (the copy holds only the lanes that were active when it was made)
    copied = clean.replace("\tv_readlane_b32 s1, v5, 7\n", "\tv_mov_b32_e32 v6, v5\n\tv_readlane_b32 s1, v6, 7\n")
    n, fail, _ = hotcheck.check(copied)
    assert [(r, ops) for _, r, ops in fail] == [("v6", {"v_readlane of a copy-of-hot": 1}),
                                                ("v5", {"hot word read by v_mov_b32_e32 (only v_readlane_b32 may)": 1})]
    # ... also when the hot word is parked in a copy, its register lent to something else, and the copy moved back
    parked = clean.replace("\tv_readlane_b32 s1, v5, 7\n",
                           "\tv_mov_b32_e32 v6, v5\n\tv_add_f32_e32 v5, v2, v3\n\tv_mov_b32_e32 v5, v6\n\tv_readlane_b32 s1, v5, 7\n")
    n, fail, _ = hotcheck.check(parked)
    assert [(r, ops) for _, r, ops in fail] == [("v5", {"v_readlane of a copy-of-hot": 1}),
                                                ("v5", {"hot word read by v_mov_b32_e32 (only v_readlane_b32 may)": 1})]
    # a register that once held a hot word and now holds a computed value may be read across lanes (registers are recycled;
    # the wave-cooperative respawn draws are such values)
    recycled = clean.replace("\tv_readlane_b32 s1, v5, 7\n", "\tv_mul_f32_e32 v5, v2, v3\n\tv_readlane_b32 s1, v5, 7\n")
    n, fail, info = hotcheck.check(recycled)
    assert not fail and ("v5", {"hot": 1, "computed": 1}) in [(r, k) for _, r, k in info]
    # round 3 (advisor): a hot word rematerialised through a select, a DPP move, an OR or a permute used to pass as "computed";
    # now ANY reader of a hot word other than v_readlane_b32 is refused, whatever its result feeds
    for remat in ("v_cndmask_b32_e32 v6, v5, v7, vcc", "v_mov_b32_dpp v6, v5 row_shr:1 row_mask:0xf bank_mask:0xf", "v_or_b32_e32 v6, v5, v7",
                  "v_perm_b32 v6, v5, v7, s2", "v_accvgpr_write_b32 a3, v5", "v_add_f32_e32 v6, -v5, v7"):
        sneaky = clean.replace("\tv_readlane_b32 s1, v5, 7\n", f"\t{remat}\n\tv_readlane_b32 s1, v6, 7\n")
        n, fail, _ = hotcheck.check(sneaky)
        assert any(r == "v5" and "hot word read by " + remat.split()[0] in list(ops)[0] for _, r, ops in fail), (remat, fail)
    # a single-dword load from anywhere else than the parameter block (the info['fuel_used'] accumulator) is an ordinary value
    other = clean.replace("\tv_readlane_b32 s1, v5, 7\n", "\tglobal_load_dword v6, v[2:3], off\n\tv_add_f32_e32 v7, v6, v6\n\tv_readlane_b32 s1, v5, 7\n")
    assert not hotcheck.check(other)[1]
    # ... and cross-lane reads of computed values beyond the respawn draws' budget are refused
    many = clean.replace("\tv_readlane_b32 s1, v5, 7\n", "\tv_mul_f32_e32 v8, v2, v3\n" + "\tv_readlane_b32 s1, v8, 7\n" * (hotcheck.COMPUTED_READLANE_LIMIT + 1))
    fail = hotcheck.check(many)[1]
    assert len(fail) == 1 and fail[0][1] == "*"


def test_built_library_keeps_hot_words_in_registers():
    lib = build.build()
    assert os.path.exists(lib)
    assert hotcheck.verify(lib) >= 20          # every hlx_env_kernel instantiation of the product library


def test_kernarg_tail_load_sits_at_kernel_entry_in_every_instantiation():
    """hotcheck.tail_load_positions: the scalar load of the output pointers within the first instructions of each env kernel
    (sunk behind the Philox block it costs ~0.45-0.8 us per launch: DESIGN.md section 5, "The kernarg tail")."""
    head = "_ZN12_GLOBAL__N_114hlx_env_kernelILj1ELi0ELb0EEEvv:\n"
    tail = "\ts_load_dwordx8 s[36:43], s[0:1], 0x38\n\t.end_amdhsa_kernel\n"
    late = head + "\tv_add_f32_e32 v1, v2, v3\n" * 80 + tail
    assert list(hotcheck.tail_load_positions(late).values()) == [81]
    early = head + "\ts_nop 0\n" * 59 + "\tv_add_f32_e32 v1, v2, v3\n" + tail      # the preload header's padding does not count
    assert list(hotcheck.tail_load_positions(early).values()) == [2]
    pos = hotcheck.tail_load_positions(hotcheck.listing(build.build()))
    assert len(pos) >= 20 and all(v is not None and v <= hotcheck.TAIL_LOAD_LIMIT for v in pos.values()), pos


def test_packed_float32_operands_read_only_the_selected_half_of_a_register_pair():
    """`v_pk_fma_f32 v[108:109], v[218:219], ... op_sel_hi:[0,1,1]` broadcasts v218 and never reads v219: a hot word that happens
    to be the unread half of such a pair is not a leak (seen in the generic fused-rollout kernel); the read half is."""
    base = SPILLED.replace("\tscratch_load_dword v5, off, off offset:4\n", "").replace("\tv_mov_b32_e32 v5, 0\n", "") \
                  .replace("\tscratch_store_dword off, v5, off offset:4\n", "")
    unread = base.replace("\tv_readlane_b32 s1, v5, 7\n", "\tv_pk_fma_f32 v[8:9], v[4:5], v[10:11], v[12:13] op_sel_hi:[0,1,1] neg_lo:[1,0,0]\n\tv_readlane_b32 s1, v5, 7\n")
    assert not hotcheck.check(unread)[1]
    read = unread.replace("op_sel_hi:[0,1,1]", "op_sel_hi:[1,1,1]")
    fail = hotcheck.check(read)[1]
    assert len(fail) == 1 and fail[0][1] == "v5"
