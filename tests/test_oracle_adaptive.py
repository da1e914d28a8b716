"""Pins the oracle's restatement of the NON-strict layer-0 arms against the reference's own tests.

Every assertion of crates/db/src/search/vector/policy.rs:603-1014 and
crates/db/tests/production_support/vector/policy.rs is replayed against `orc_policy_decide`; the search-level
contracts of index.rs:2413-2611 and randomness.rs:168-207 are replayed against the oracle's search.  Nothing here
reads /root/reference at run time.
"""
import numpy as np
import pytest

import fixtures as fx

COS, L2, L1 = 0, 1, 2


def ctx(**over):  # policy.rs:620-630 `context()`
    c = dict(topk_ready=1, ef=64, search_frontier_len=64, candidate_frontier_len=64, current=0.2, delta=0.4)
    c.update(over)
    return c


def bypass():  # policy.rs:632-641 `adaptive_bypass_policy(..)`: ef 64, (24, 4, 0.12, 3)
    return dict(bypass_from_deployed=1, bypass_ef=64, bypass_min_frontier=24, bypass_window_expansions=4,
                bypass_min_filter_rate=0.12, bypass_read_budget_multiplier=3)


def pol(metric, mode, thr, ratio, pre=None, adaptive=1, failure=0.1):
    return dict(metric=metric, simhash_mode=mode, configured_threshold=thr, sampling_ratio=ratio,
                pre_sampling_override=-1.0 if pre is None else pre, adaptive_enabled=adaptive, failure=failure)


# policy.rs:643-676 / production_support policy.rs:52-84
def test_compatibility_table_separates_metric_filtering_from_sampling(orc):
    for metric in (COS, L2, L1):
        for mode in (orc.SIMHASH_OFF, orc.SIMHASH_ALWAYS, orc.SIMHASH_ADAPTIVE):
            for adaptive in (0, 1):
                d = orc.policy_decide(**pol(metric, mode, 43, 0.4, adaptive=adaptive), **ctx())
                want = metric == COS and mode != orc.SIMHASH_OFF
                assert bool(d.fetch_missing) == want and bool(d.filter_cached) == want
                assert bool(d.has_threshold) == want
                assert (d.sampling_probability == 1.0) == (mode == orc.SIMHASH_OFF)


# policy.rs:678-692
def test_fixed_mode_uses_configured_threshold_exactly(orc):
    d = orc.policy_decide(**pol(COS, orc.SIMHASH_ALWAYS, 37, 0.5), **ctx())
    assert d.has_threshold and d.threshold == 37
    assert d.sampling_probability == 0.5


# policy.rs:694-725
def test_bypass_disables_cached_and_missing_filtering_together(orc):
    d = orc.policy_decide(**pol(COS, orc.SIMHASH_ADAPTIVE, 43, 0.5), **bypass(), **ctx(), simhash_filter_reads=192)
    assert d.bypassed and not d.fetch_missing and not d.filter_cached and not d.has_threshold
    assert d.trigger == orc.TRIGGER_READ_BUDGET
    assert (d.next_state, d.next_state_remaining) == (orc.BYPASS_BYPASSING, 3)


# policy.rs:727-865
def test_adaptive_bypass_policy_owns_triggers_windows_and_cooldown(orc):
    p = pol(COS, orc.SIMHASH_ADAPTIVE, 43, 0.5)

    def decide(**obs):
        return orc.policy_decide(**p, **bypass(), **ctx(), **obs)

    low = decide(window_examined=20, window_filtered=0, window_expansions=4)
    assert low.bypassed and low.trigger == orc.TRIGGER_LOW_YIELD
    both = decide(simhash_filter_reads=192, window_examined=20, window_filtered=0, window_expansions=4)
    assert both.trigger == orc.TRIGGER_BOTH
    state = (both.next_state, both.next_state_remaining)
    for remaining in (2, 1):
        c = decide(state=state[0], state_remaining=state[1])
        assert c.bypassed and c.trigger == orc.TRIGGER_NONE
        assert (c.next_state, c.next_state_remaining) == (orc.BYPASS_BYPASSING, remaining)
        state = (c.next_state, c.next_state_remaining)
    final = decide(state=state[0], state_remaining=state[1])
    assert final.bypassed and (final.next_state, final.next_state_remaining) == (orc.BYPASS_COOLING, 4)
    state = (final.next_state, final.next_state_remaining)
    for remaining in (3, 2, 1):
        c = decide(state=state[0], state_remaining=state[1])
        assert not c.bypassed and (c.next_state, c.next_state_remaining) == (orc.BYPASS_COOLING, remaining)
        state = (c.next_state, c.next_state_remaining)
    ready = decide(state=state[0], state_remaining=state[1])
    assert not ready.bypassed and ready.next_state == orc.BYPASS_READY
    re = decide(state=orc.BYPASS_COOLING, state_remaining=1, simhash_filter_reads=192)
    assert re.bypassed and re.trigger == orc.TRIGGER_READ_BUDGET
    small = orc.policy_decide(**p, **bypass(), **ctx(candidate_frontier_len=23), simhash_filter_reads=192)
    assert not small.bypassed and small.trigger == orc.TRIGGER_NONE
    one = dict(bypass())
    one["bypass_window_expansions"] = 1
    d = orc.policy_decide(**p, **one, **ctx(), simhash_filter_reads=192)
    assert (d.next_state, d.next_state_remaining) == (orc.BYPASS_COOLING, 1)


# policy.rs:867-888, 950-990 / production_support policy.rs:147-205
def test_adaptive_policy_cold_start_thresholds_and_quality(orc):
    p = pol(COS, orc.SIMHASH_ADAPTIVE, 43, 0.3)
    cold = orc.policy_decide(**p, **ctx(topk_ready=0, search_frontier_len=4, candidate_frontier_len=4))
    assert cold.has_threshold and cold.threshold == 1
    assert cold.sampling_probability == 1.0
    active = orc.policy_decide(**p, **ctx())
    assert np.float32(0.3) <= active.sampling_probability <= np.float32(0.90)
    assert active.threshold <= 64
    near = orc.policy_decide(**p, **ctx(current=0.1, delta=0.2))
    far = orc.policy_decide(**p, **ctx(current=0.8, delta=0.9))
    assert near.threshold >= far.threshold
    assert near.sampling_probability >= far.sampling_probability
    assert orc.candidate_probability(near.sampling_kind, near.sampling_probability, 58, near.threshold) >= \
        orc.candidate_probability(near.sampling_kind, near.sampling_probability, 32, near.threshold)
    assert orc.candidate_probability(orc.SAMPLING_FIXED, 0.4, 64) == np.float32(0.4)
    assert orc.candidate_probability(orc.SAMPLING_ADAPTIVE, 0.0, 64) == 0.0
    assert orc.candidate_probability(orc.SAMPLING_ADAPTIVE, 1.0, 64) == 1.0

    def thr(configured, failure=0.1, **c):
        return orc.policy_decide(**pol(COS, orc.SIMHASH_ADAPTIVE, configured, 0.5, failure=failure), **ctx(**c)).threshold

    assert thr(43, 0.4) >= thr(43, 0.01)              # policy.rs:921-948
    assert thr(0) == 0 and thr(20) <= 20 and thr(43) <= 43
    assert thr(43, topk_ready=0) == 1
    assert orc.adaptive_sampling_ratio(1.0, 64, 64, 0.2, 0.4) == 1.0
    assert orc.adaptive_sampling_ratio(0.3, 64, 64, 0.0, 0.0) == np.float32(0.3)


# policy.rs:992-1013 / production_support policy.rs:87-145
def test_decision_owns_pre_and_post_sampling_activation(orc):
    p = pol(COS, orc.SIMHASH_ALWAYS, 43, 0.4, pre=0.2)
    active = orc.policy_decide(**p, **ctx())
    assert active.pre_sampling_probability == 0.25
    assert active.sampling_probability == np.float32(0.4)
    assert active.base_sampling_probability == np.float32(0.4)
    small = orc.policy_decide(**p, **ctx(candidate_frontier_len=4))
    assert small.pre_sampling_kind == orc.SAMPLING_EXHAUSTIVE and small.sampling_kind == orc.SAMPLING_EXHAUSTIVE
    defer_all = orc.policy_decide(**pol(COS, orc.SIMHASH_ALWAYS, 43, 0.0, pre=0.0), **ctx())
    assert defer_all.pre_sampling_probability == 0.0 and defer_all.sampling_probability == 0.0
    assert orc.pre_sampling_decision(0.5, 129, 64)[0] == orc.SAMPLING_FIXED
    fixed = orc.policy_decide(**pol(COS, orc.SIMHASH_ALWAYS, 37, 0.4, pre=0.2), **ctx())
    assert fixed.threshold == 37
    byp = orc.policy_decide(**pol(COS, orc.SIMHASH_ADAPTIVE, 37, 0.4, pre=0.2), **bypass(), **ctx(), simhash_filter_reads=192)
    assert byp.bypassed and not byp.fetch_missing and not byp.filter_cached and not byp.has_threshold


# production_support policy.rs:207-243
def test_bypass_cooldown_and_combined_trigger(orc):
    p = pol(COS, orc.SIMHASH_ADAPTIVE, 43, 0.5)
    cooling = orc.policy_decide(**p, **bypass(), **ctx(), state=orc.BYPASS_COOLING, state_remaining=2)
    assert not cooling.bypassed and (cooling.next_state, cooling.next_state_remaining) == (orc.BYPASS_COOLING, 1)
    comb = orc.policy_decide(**p, **bypass(), **ctx(), simhash_filter_reads=2**64 - 1, window_examined=10,
                             window_filtered=0, window_expansions=2**64 - 1)
    assert comb.trigger == orc.TRIGGER_BOTH


# randomness.rs:168-207
def test_search_session_boundaries_and_seed_contract(orc):
    s = orc.Rng.seeded(42)
    assert s.should_sample(1.0) and not s.should_sample(0.0) and s.choose_index(0) is None
    assert s.words == 0                                   # boundary probabilities never advance the generator
    seed = 0x0123_4567_89AB_CDEF ^ (((42 << 17) | (42 >> 47)) & (2**64 - 1)) ^ (128 << 7)
    assert orc.lib().orc_query_seed(0x0123_4567_89AB_CDEF, 42, 128) == seed
    a, b = orc.Rng.seeded(seed), orc.Rng.seeded(seed)
    for _ in range(100):
        assert a.should_sample(0.37) == b.should_sample(0.37)
        ia, ib = a.choose_index(11), b.choose_index(11)
        assert ia == ib and 0 <= ia < 11


def test_should_sample_consumes_the_pinned_stream(orc):
    """should_sample draws random::<f32>() from the SAME StdRng stream the SimHasher KAT pins."""
    w = np.zeros(64, np.uint32)
    orc.lib().orc_stdrng_u32(7, w.ctypes.data_as(orc.u32p), 64)
    s = orc.Rng.seeded(7)
    for i in range(64):
        u = np.float32(int(w[i]) >> 8) * np.float32(2.0 ** -24)
        assert s.should_sample(0.37) == bool(u < np.float32(0.37))
    assert s.words == 64


def circle_index(orc, metric, n=32, m=8, m0=16, efc=32, seed=3):
    ix = orc.Index(2, metric, m=m, m0=m0, ef_construction=efc)
    lv = fx.draw_levels(n, m, seed)
    for off in range(n):
        ang = np.float32(off) * np.float32(2 * np.pi) / np.float32(n)
        assert ix.insert(off + 1, [np.cos(ang), np.sin(ang)], int(lv[off])) == orc.OK
    ix.set_simhash(42)
    return ix


# index.rs:2413-2566 test_layer0_search_modes_cover_sampling_filtering_and_adaptive_bypass
def test_layer0_search_modes_cover_sampling_filtering_and_adaptive_bypass(orc):
    ix = circle_index(orc, orc.COSINE)
    cfg = dict(ef=16, simhash_threshold=0, sampling_ratio=0.5, resident_simhash=0)
    off = orc.SearchParams.new(5, simhash_mode=orc.SIMHASH_OFF, pre_simhash_sampling_ratio_override=1.0, **cfg)
    rc, off_ids, _, off_st = ix.search_params([1.0, 0.0], off, with_stats=True)
    assert rc == orc.OK and len(off_ids) > 0
    assert off_st["simhash_examined"] == 0 and off_st["txn_get_simhash_filter"] == 0 and off_st["simhash_filtered"] == 0
    assert off_st["expansion_steps"] > 0
    rc, strict_ids, strict_sc = ix.search([1.0, 0.0], 5, 16)
    assert off_ids.tolist() == strict_ids.tolist()       # Off + pre 1.0 IS the strict-exhaustive arm

    always = orc.SearchParams.new(5, simhash_mode=orc.SIMHASH_ALWAYS, pre_simhash_sampling_ratio_override=1.0,
                                  simhash_sampling_ratio_override=0.0, **cfg)
    rc, ids, _, st = ix.search_params([1.0, 0.0], always, with_stats=True)
    assert rc == orc.OK and len(ids) > 0
    assert st["simhash_examined"] > 0 and st["simhash_passed_before_sampling"] > 0
    assert st["active_simhash_threshold_sum"] == 0      # avg_active_simhash_threshold == 0.0

    fixed = orc.SearchParams.new(5, simhash_mode=orc.SIMHASH_ALWAYS, pre_simhash_sampling_ratio_override=1.0,
                                 simhash_sampling_ratio_override=1.0, **cfg)
    rc, ids, _ = ix.search_params([1.0, 0.0], fixed)
    assert ids.tolist() == off_ids.tolist()

    adaptive = orc.SearchParams.new(5, simhash_mode=orc.SIMHASH_ADAPTIVE, pre_simhash_sampling_ratio_override=0.25,
                                    simhash_sampling_ratio_override=0.5, simhash_failure_prob_override=0.5,
                                    bypass_min_frontier=1, bypass_window_expansions=1, bypass_min_filter_rate=1.0,
                                    read_budget_multiplier=1, **cfg)
    rc, ids, _, st = ix.search_params([0.0, 1.0], adaptive, with_stats=True)
    assert rc == orc.OK and len(ids) > 0
    assert st["expansion_steps"] > 0
    assert st["effective_beam_len_sum"] >= st["effective_beam_len_samples"] >= 1
    assert st["simhash_bypass_expansions"] > 0


# index.rs:2568-2611 non_angular_metric_disables_filter_phase_without_disabling_sampling_policy
def test_non_angular_metric_disables_filter_phase_without_disabling_sampling_policy(orc):
    ix = orc.Index(2, orc.L2SQ)
    lv = fx.draw_levels(24, 16, 5)
    for node in range(1, 25):
        assert ix.insert(node, [float(node), 1.0], int(lv[node - 1])) == orc.OK
    ix.set_simhash(42)
    p = orc.SearchParams.new(5, ef=16, simhash_mode=orc.SIMHASH_ALWAYS, pre_simhash_sampling_ratio_override=1.0,
                             simhash_threshold=64, sampling_ratio=0.5, resident_simhash=0)
    rc, ids, _, st = ix.search_params([1.0, 1.0], p, with_stats=True)
    assert rc == orc.OK and len(ids) > 0
    assert st["txn_get_simhash_filter"] == 0 and st["simhash_examined"] == 0 and st["simhash_filtered"] == 0
    assert st["active_sampling_ratio_sum"] / st["active_sampling_ratio_samples"] == 0.5


def test_production_default_is_reproducible_and_close_to_exhaustive(orc):
    """SearchParams::new(k): same (query, entry, ef) => same draws => same results; sampling
    engages only on frontiers wider than max(ef/4, 8) (policy.rs:526-556), so results stay near the strict arm."""
    rng = np.random.default_rng(11)
    n, dim = 1500, 32
    x = rng.standard_normal((n, dim)).astype(np.float32)
    for metric in (orc.L2SQ, orc.COSINE):
        ix = orc.Index(dim, metric)
        lv = fx.draw_levels(n, 16, 9)
        for i in range(n):
            assert ix.insert(i + 10, x[i], int(lv[i])) == orc.OK
        ix.set_simhash(42)
        p = orc.SearchParams.new(10)
        hits = tot = 0
        sampled = 0
        for qi in range(40):
            q = rng.standard_normal(dim).astype(np.float32)
            rc, ids, sc, st = ix.search_params(q, p, with_stats=True)
            rc2, ids2, sc2, st2 = ix.search_params(q, p, with_stats=True)
            assert rc == orc.OK and ids.tolist() == ids2.tolist() and st == st2
            assert sc.view(np.uint32).tolist() == sc2.view(np.uint32).tolist()
            rc, tid, _ = ix.flat(q, 10)
            hits += len(set(ids.tolist()) & set(tid.tolist()))
            tot += 10
            sampled += st["rng_words"]
            assert st["distance_computations"] == st["vectors_loaded"] + 1
            if metric == orc.L2SQ:
                assert st["simhash_examined"] == 0
            else:
                assert st["simhash_examined"] > 0
        assert sampled > 0, "the sampling policy never engaged"
        assert hits / tot > 0.8


@pytest.mark.parametrize("resident", [0, 1])
def test_resident_snapshot_never_counts_simhash_reads(orc, resident):
    """memory_store.rs:329-335 vs :338-347: a resident snapshot answers SimHash lookups without stable-view reads,
    so the read-budget trigger (policy.rs:266) can only fire for an uncached handle."""
    rng = np.random.default_rng(5)
    n, dim = 800, 16
    x = rng.standard_normal((n, dim)).astype(np.float32)
    ix = orc.Index(dim, orc.COSINE)
    lv = fx.draw_levels(n, 16, 2)
    for i in range(n):
        ix.insert(i, x[i], int(lv[i]))
    ix.set_simhash(42)
    p = orc.SearchParams.new(10, resident_simhash=resident, read_budget_multiplier=1)
    reads = budget = 0
    for qi in range(20):
        rc, ids, sc, st = ix.search_params(rng.standard_normal(dim).astype(np.float32), p, with_stats=True)
        assert rc == orc.OK
        reads += st["txn_get_simhash_filter"]
        budget += st["simhash_bypass_trigger_budget"]
    if resident:
        assert reads == 0 and budget == 0
    else:
        assert reads > 0
