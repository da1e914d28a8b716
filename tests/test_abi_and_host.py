"""CPU-side checks: the C-ABI library loads and exports every symbol include/helix_vec.h declares,
and the host-side mirror of the reference interface behaves like the reference (no GPU needed)."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "helix_vec.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hvx_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import pyhvx
    L = pyhvx.lib()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/helix_vec.h but not exported"
    assert b"gfx950" in L.hvx_version()


def test_library_contains_gfx950_code_objects():
    import pyhvx
    blob = open(pyhvx.LIB_PATH, "rb").read()
    assert b"gfx950" in blob and b"hnsw_search_kernel" in blob


def test_search_params_mirror_reference_defaults():
    import pyhvx as hv
    p = hv.SearchParams(10)
    assert (p.k, p.ef) == (10, 100)          # mod.rs:482-500: ef = max(k, 100)
    assert hv.SearchParams(250).ef == 250
    assert p.with_ef(128).ef == 128
    with pytest.raises(hv.HelixDbError):
        hv.SearchParams(0)
    with pytest.raises(hv.HelixDbError):
        hv.SearchParams(10).with_ef(9)        # parameters.rs:118-133


def test_restricted_candidates_dedupe_and_cap():
    import pyhvx as hv
    c = hv.RestrictedVectorCandidates.from_ids([5, 1, 5, 9, 1])
    assert c.ids.tolist() == [1, 5, 9] and len(c) == 3
    with pytest.raises(hv.HelixDbError) as e:
        hv.RestrictedVectorCandidates.from_ids(np.arange(1_000_001, dtype=np.uint64))   # restricted.rs:40
    assert e.value.status == hv.ERR_CANDIDATE_LIMIT
    assert len(hv.RestrictedVectorCandidates.from_ids(np.arange(1_000_000, dtype=np.uint64))) == 1_000_000
    words = np.zeros(2, np.uint64); words[0] = (1 << 3) | (1 << 63); words[1] = 1
    assert hv.RestrictedVectorCandidates.from_bitmap_words(words).ids.tolist() == [3, 63, 64]


def test_import_rejects_bad_arguments_without_touching_a_gpu():
    """Argument validation happens before any HIP call, so it is observable on a CPU-only box."""
    import pyhvx as hv
    with pytest.raises(hv.HelixDbError) as e:
        hv.ValidatedVectorReadIndex.managed(dim=0, metric=hv.EUCLIDEAN, node_ids=[], vectors=np.zeros((0, 1), np.float32),
                                            l0_offsets=[0], l0_neighbors=[])
    assert e.value.status == hv.ERR_DIMENSION
    with pytest.raises(hv.HelixDbError) as e:
        hv.ValidatedVectorReadIndex.managed(dim=4, metric=7, node_ids=[], vectors=np.zeros((0, 4), np.float32),
                                            l0_offsets=[0], l0_neighbors=[])
    assert e.value.status == hv.ERR_UNSUPPORTED


def test_no_product_code_references_the_oracle():
    """The oracle is test infrastructure; the product package must never import, link or call it."""
    pkg = os.path.join(ROOT, "helix-db_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "hvx_oracle" not in txt and "import orc" not in txt and "oracle/" not in txt, f


# --- persisted row codecs: the reference's own wire-layout tests (encoding/v1/values/vectors.rs:214-330,
#     values/vectors/neighbors.rs) restated byte for byte ---
def _be64(*ids):
    return b"".join(int(i).to_bytes(8, "big") for i in ids)


def test_layer0_row_codec_matches_reference_wire_layouts():
    import pyhvx as hv
    assert hv.decode_layer0_row(bytes([0x12, 0, 0, 0, 0])) == ([], None)              # zero neighbours, exact layout
    assert hv.decode_layer0_row(bytes([0x12, 0, 0, 0, 2]) + _be64(3, 7)) == ([3, 7], None)
    rec = bytes([0x13, 0x01, 0, 0, 0, 2]) + (0x0102_0304_0506_0708).to_bytes(8, "little") + _be64(1, 9)
    assert hv.decode_layer0_row(rec) == ([1, 9], 0x0102_0304_0506_0708)
    assert hv.decode_layer0_row(bytes([0x13, 0, 0, 0, 0, 2]) + _be64(1, 9)) == ([1, 9], None)
    assert hv.decode_layer0_row(b"") == ([], None)                                    # empty compatibility value
    assert hv.decode_layer0_row(bytes([0x12, 0, 0, 0, 2]) + _be64(0, (1 << 64) - 1)) == ([0, (1 << 64) - 1], None)
    for bad in (bytes([0x14, 0, 0, 0, 0]),                       # InvalidEncodingType
                bytes([0x13, 0x02, 0, 0, 0, 0]),                 # invalid flags
                bytes([0x12, 0, 0, 0, 2]) + _be64(3),            # count says 2, one id present
                bytes([0x12, 0, 0, 0, 1]) + _be64(3) + b"x",     # trailing byte
                bytes([0x13, 0x01, 0, 0, 0, 0, 1, 2, 3])):       # truncated SimHash
        with pytest.raises(hv.HelixDbError):
            hv.decode_layer0_row(bad)


def test_upper_row_codec_and_keys():
    import pyhvx as hv
    assert hv.decode_upper_row((2).to_bytes(4, "big") + _be64(9, 1)) == [9, 1]      # order preserved by the codec
    assert hv.decode_upper_row((0).to_bytes(4, "big")) == []
    for bad in (b"\x00\x00", (2).to_bytes(4, "big") + _be64(9), (1).to_bytes(4, "big") + _be64(9) + b"z"):
        with pytest.raises(hv.HelixDbError):
            hv.decode_upper_row(bad)
    ix = (77).to_bytes(8, "big")
    k = hv.parse_vector_key(bytes([0xF1]) + ix + bytes([0x02]) + (0xABCD).to_bytes(8, "big") + (5).to_bytes(8, "big"))
    assert k == dict(kind=0x02, index_id=77, node_id=5, order_code=0xABCD, layer=0)
    k = hv.parse_vector_key(bytes([0xF0]) + ix + bytes([0x16]) + (5).to_bytes(8, "big"))
    assert k["kind"] == 0x16 and k["node_id"] == 5
    k = hv.parse_vector_key(bytes([0xF0]) + ix + bytes([0x11]) + (3).to_bytes(2, "big") + (5).to_bytes(8, "big"))
    assert k["kind"] == 0x11 and k["layer"] == 3 and k["node_id"] == 5
    assert hv.parse_vector_key(bytes([0xF0]) + ix + bytes([0x12]) + (5).to_bytes(8, "big")) is None  # SimHash row: not a search row


def test_full_search_params_mirror_the_reference():
    """mod.rs:482-500 SearchParams::new, :546-552 requires_query_simhash, :555-583 bypass tuning validation,
    :614-620 throughput_profile_floor_92; hvx_search_params_default / hvx_simhash_config_default (mod.rs:313-329)."""
    import ctypes as C
    import pyhvx as hv
    p = hv.SearchParams.new(7)
    c = p._c()
    d = hv._Params()
    hv.lib().hvx_search_params_default(C.byref(d), 7)
    for f, _ in hv._Params._fields_:
        assert getattr(c, f) == getattr(d, f), f
    assert (d.k, d.ef, d.simhash_mode) == (7, 100, hv.SIMHASH_ADAPTIVE)
    assert (d.bypass_min_frontier, d.bypass_window_expansions, d.read_budget_multiplier) == (24, 4, 3)
    assert d.bypass_min_filter_rate == np.float32(0.12) and d.pre_simhash_sampling_ratio_override < 0
    assert hv.SearchParams.new(250)._c().ef == 250
    assert p.requires_query_simhash()
    assert not hv.SearchParams.new(5).with_simhash_mode(hv.SIMHASH_OFF).requires_query_simhash()
    assert hv.SearchParams.new(5).with_simhash_mode(hv.SIMHASH_OFF).with_pre_simhash_sampling_ratio(0.5).requires_query_simhash()
    assert not hv.SearchParams(5).requires_query_simhash()      # the strict baseline: Off + pre 1.0
    t = hv.SearchParams.throughput_profile_floor_92(10)._c()
    assert (t.ef, t.simhash_mode) == (48, hv.SIMHASH_ADAPTIVE) and t.pre_simhash_sampling_ratio_override == np.float32(0.20)
    assert hv.SearchParams.throughput_profile_floor_92(60)._c().ef == 60
    for bad in (lambda: p.with_simhash_bypass_tuning(0, 4, 0.1, 3), lambda: p.with_simhash_bypass_tuning(24, 4, 1.5, 3),
                lambda: p.with_pre_simhash_sampling_ratio(-0.1), lambda: p.with_simhash_sampling_ratio(float("nan")),
                lambda: p.with_simhash_failure_prob(0.0), lambda: p.with_simhash_failure_prob(1.0)):
        with pytest.raises(hv.HelixDbError) as e:
            bad()
        assert e.value.status == hv.ERR_K_RANGE
    cfg = hv.SimHashConfig.default()
    assert (cfg.seed, cfg.simhash_threshold, cfg.adaptive_enabled) == (42, 43, 1)
    assert cfg.sampling_ratio == np.float32(0.8) and cfg.adaptive_failure_prob == np.float32(0.1)
    assert C.sizeof(hv.AdaptiveStats) == 80 and cfg.resident_snapshot == 1


def test_restricted_row_dedupe_and_materialisation_follow_the_interpreter():
    """interpreter/access/restricted_vector.rs:14-65: first row wins, rank order, `$distance` as F64."""
    import pyhvx as hv
    rows = [{"current": ("node", 9), "tag": "a"}, {"current": ("node", 4), "tag": "b"}, {"current": ("node", 9), "tag": "dup"}]
    by_id = hv.unique_restricted_rows(rows, "node")
    assert list(by_id) == [4, 9] and by_id[9]["tag"] == "a"
    with pytest.raises(hv.HelixDbError):
        hv.unique_restricted_rows([{"current": ("edge", 1)}], "node")
    with pytest.raises(hv.HelixDbError):
        hv.unique_restricted_rows([{"current": None}], "node")
    res = [hv.SearchResult(9, np.float32(0.1)), hv.SearchResult(4, np.float32(0.25))]
    out = hv.materialize_restricted_results(by_id, res)
    assert [r["tag"] for r in out] == ["a", "b"]
    assert out[0]["virtual_properties"]["$distance"] == float(np.float32(0.1)) and isinstance(out[0]["virtual_properties"]["$distance"], float)
    with pytest.raises(hv.HelixDbError) as e:
        hv.materialize_restricted_results(by_id, [hv.SearchResult(5, np.float32(0.0))])
    assert e.value.status == hv.ERR_INVARIANT


@pytest.mark.parametrize("configured,failure", [(43, 0.1), (64, 0.5), (20, 0.01), (1, 0.3), (0, 0.1), (37, 0.9)])
def test_adaptive_threshold_step_table_equals_the_formula(orc, configured, failure):
    """The device never evaluates acos/ln: hvx_params.hip turns policy.rs:577-599 into a 64-step table with the host's
    libm.  Table lookup == the oracle's direct evaluation for random deltas and for every f32 around every step."""
    import pyhvx as hv
    brk = hv.adaptive_threshold_table(configured, failure)
    assert all(brk[i] >= brk[i + 1] for i in range(63))          # nested steps

    def lookup(delta):
        return int(np.count_nonzero(np.float32(delta) <= brk))

    rng = np.random.default_rng(configured * 7 + 1)
    deltas = np.concatenate([rng.random(4000, dtype=np.float32), np.float32([0.0, 1.0, 0.5, 2.0, 1e-7, 1e30]),
                             np.abs(rng.standard_normal(500).astype(np.float32)) * np.float32(1e-3)])
    for d in deltas:
        assert lookup(d) == orc.adaptive_threshold(1, d, configured, failure), float(d)
    for b in brk[brk >= 0]:
        bits = int(np.float32(b).view(np.uint32))
        for off in range(-3, 4):
            if 0 <= bits + off <= 0x7F7FFFFF:
                d = np.uint32(bits + off).view(np.float32)
                assert lookup(d) == orc.adaptive_threshold(1, d, configured, failure), (float(b), off)


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/helix_vec.h must be usable from C (cgo / Rust bindgen / JNI): the example compiles as pedantic C99 and
    links against the product library (running it needs a GPU)."""
    import subprocess
    exe = tmp_path / "c_abi_example"
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "c_abi_example.c"), "-L", os.path.join(ROOT, "helix-db_amd"), "-lhelix_vec_gfx950",
           f"-Wl,-rpath,{os.path.join(ROOT, 'helix-db_amd')}", "-o", str(exe)]
    out = subprocess.run(cmd, capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert exe.exists()


def test_struct_layouts_of_the_header_and_the_ctypes_mirrors_agree(tmp_path):
    """Every by-value struct of include/helix_vec.h has the size the ctypes mirror of the binding declares (a field added on one side
    only -- hvx_build_params.link_mode was the last -- would shift every later field of a call)."""
    import ctypes as C
    import subprocess
    import pyhvx as hv
    pairs = [("hvx_index_desc", hv._Desc), ("hvx_stats", hv.Stats), ("hvx_query_stats", hv.QueryStats), ("hvx_search_params", hv._Params),
             ("hvx_simhash_config", hv.SimHashConfig), ("hvx_adaptive_stats", hv.AdaptiveStats), ("hvx_restricted_params", hv.RestrictedParams),
             ("hvx_restricted_stats", hv.RestrictedStats), ("hvx_index_metadata", hv.IndexMetadata), ("hvx_build_params", hv.BuildParams),
             ("hvx_build_stats", hv.BuildStats), ("hvx_graph_audit", hv.GraphAudit), ("hvx_batcher_times", hv.BatcherTimes),
             ("hvx_delete_stats", hv.DeleteStats), ("hvx_batcher_ticket", hv.BatcherTicket)]
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "helix_vec.h"\nint main(void) {\n' +
                   "".join('  printf("%s %%zu\\n", sizeof(%s));\n' % (n, n) for n, _ in pairs) + "  return 0;\n}\n")
    exe = tmp_path / "sizes"
    out = subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    for name, mirror in pairs:
        assert int(got[name]) == C.sizeof(mirror), (name, got[name], C.sizeof(mirror))


def test_bench_sizes_its_cpu_work_to_the_cgroup_quota(tmp_path, monkeypatch):
    """bench.py's oracle legs and torch's helper threads follow the process's CPU quota (cgroup v2 cpu.max), not the logical CPU count:
    the GPU boxes give 16 of 256 CPUs, and a 64-thread run was throttled -- sometimes inside a timed section -- and reported a
    baseline of a quarter of what 16 threads reach."""
    import builtins
    sys.path.insert(0, ROOT)
    import bench
    real_open = builtins.open

    def fake(content):
        def _open(path, *a, **k):
            if path == "/sys/fs/cgroup/cpu.max":
                if content is None:
                    raise FileNotFoundError(path)
                f = tmp_path / "cpu.max"
                f.write_text(content)
                return real_open(f, *a, **k)
            return real_open(path, *a, **k)
        return _open

    monkeypatch.setattr(os, "cpu_count", lambda: 256)
    monkeypatch.setattr(builtins, "open", fake("1600000 100000\n"))
    assert bench.cpu_quota() == 16
    monkeypatch.setattr(builtins, "open", fake("max 100000\n"))
    assert bench.cpu_quota() == 256
    monkeypatch.setattr(builtins, "open", fake("50000 100000\n"))
    assert bench.cpu_quota() == 1
    monkeypatch.setattr(builtins, "open", fake(None))
    assert bench.cpu_quota() == 256

    class A:
        cpu_threads = 0
    monkeypatch.setattr(builtins, "open", fake("12800000 100000\n"))
    assert bench.host_threads(A) == 64          # capped
    A.cpu_threads = 5
    assert bench.host_threads(A) == 5


def test_bench_deadline_prints_the_line_and_ends_the_process(tmp_path):
    """bench.py's watchdog: once the headline object exists, a stuck extra leg cannot cost the line -- at the deadline the COMPACT
    record is printed as the one stdout line, the full record (with the legs finished so far, here one that is added while the
    watchdog waits) is written to the full-record file, and the process exits with status 0."""
    import json
    import subprocess
    full = str(tmp_path / "full.json")
    code = ("import sys, time; sys.argv = ['bench.py']; sys.path.insert(0, %r); import bench\n"
            "out = {'metric': 'm', 'value': 1.0, 'config': {'workload': 'w'}, 'roofline': {'frac': 0.5}}\n"
            "bench.start_deadline(out, 0.5, 0, %r)\n"
            "out['config3_prefilter'] = {'error': 'stuck'}\n"
            "time.sleep(30)\n"
            "print('never reached')\n") % (ROOT, full)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and "never reached" not in r.stdout
    d = json.loads(lines[0])
    assert d["metric"] == "m" and d["value"] == 1.0 and d["config3"] == {"error": "stuck"} and "deadline" in d and d["full_record"] == full
    f = json.load(open(full))
    assert f["config3_prefilter"] == {"error": "stuck"} and "deadline" in f


def test_small_row_codecs_reproduce_the_reference_frozen_bytes_and_fuzz_seeds():
    """values/vectors/simhash.rs:66-78 (`simhash_bytes_are_frozen`, truncation / trailing bytes rejected), values/vectors/entry.rs:
    51-70 (`entry_candidate_layer_bytes_are_frozen`, non-exact lengths rejected) and the reference's checked-in fuzz corpus
    (crates/db/fuzz/corpus/current_search_records/vector-{simhash,entry,layer0-empty}.bin = selector byte + payload + newline,
    fuzzing.rs:256-270 `checked_in_corpus_seeds_are_contract_valid`)."""
    import pyhvx as hv
    assert hv.decode_simhash_row(bytes([8, 7, 6, 5, 4, 3, 2, 1])) == 0x0102_0304_0506_0708
    for bad in (bytes(7), bytes(9), b""):
        with pytest.raises(hv.HelixDbError):
            hv.decode_simhash_row(bad)
    assert hv.decode_entry_candidate_layer(bytes([0x12, 0x34])) == 0x1234
    for bad in (b"", bytes(1), bytes(3)):
        with pytest.raises(hv.HelixDbError):
            hv.decode_entry_candidate_layer(bad)
    seeds = {"vector-simhash.bin": b"712345678\n", "vector-entry.bin": b"3ab\n", "vector-layer0-empty.bin": b"8\n"}
    assert hv.decode_simhash_row(seeds["vector-simhash.bin"][1:-1]) == int.from_bytes(b"12345678", "little")
    assert hv.decode_entry_candidate_layer(seeds["vector-entry.bin"][1:-1]) == 0x6162
    assert hv.decode_layer0_row(seeds["vector-layer0-empty.bin"][1:-1]) == ([], None)


def test_host_traversal_equals_the_oracle_in_both_strategies(orc):
    """hvx_traverse_host (Graph::traverse on the host: BreadthFirst and DepthFirst, traversal.rs:197-309) against the oracle's
    restatements -- the reference's own fixtures first, then random multigraphs (parallel edges with different labels,
    self-loops) over every direction, label allow-sets, the hub policy, several seeds in a given order, depth caps."""
    import pyhvx as hv
    off = np.array([0, 3, 4, 5, 5], np.uint64)                                   # traversal.rs:667-700
    tgt = np.array([1, 1, 2, 3, 3], np.uint64)
    visits, edges = hv.traverse_host(4, off, tgt, None, [0], 3, hv.DIR_OUT, depth_first=True)
    assert visits == [(0, 0), (1, 1), (3, 2), (2, 1)] and [a for _, a, _ in edges] == [0, 3, 2]
    e = [(0, 1), (1, 2), (1, 3), (3, 4), (3, 5), (3, 6)]                          # traversal.rs:576-615
    off = np.zeros(8, np.uint64); tgt = np.array([t for _, t in e], np.uint64)
    for s_, _ in e:
        off[s_ + 1:] += 1
    visits, edges = hv.traverse_host(7, off, tgt, None, [0], 2)
    assert visits == [(0, 0), (1, 1), (2, 2), (3, 2)] and len(edges) == 3
    rng = np.random.default_rng(12)
    for n, deg, nlab in [(300, 4, 3), (2000, 6, 4), (64, 40, 2)]:
        rows = [np.sort(rng.integers(0, n, rng.integers(0, 2 * deg + 1))) for _ in range(n)]
        off = np.zeros(n + 1, np.uint64); off[1:] = np.cumsum([len(r) for r in rows])
        tgt = np.concatenate(rows).astype(np.uint64)
        lab = rng.integers(0, nlab, int(off[-1])).astype(np.uint32)
        cases = [([3], 2, 0, [], 0), ([3, 7, 3, 1], 3, 2, [1, 2], 0), ([10, 11], 4, 1, [], 2 * deg + 3), ([0], 0, 2, [], 0),
                 ([5, 6, 7], 60, 0, [0], 0), ([n - 1], 1000, 2, [], 0), ([2, 1], 5, 2, [0], 3 * deg)]
        for seeds, md, direction, allowed, hub in cases:
            for dfs, ref in ((False, orc.breadth_first), (True, orc.depth_first)):
                got = hv.traverse_host(n, off, tgt, lab, seeds, md, direction, allowed, hub, depth_first=dfs)
                want = ref(n, off.astype(np.int64), tgt, lab, seeds, md, direction, allowed, hub)
                assert got[0] == want[0], f"n={n} {'dfs' if dfs else 'bfs'} case {(seeds, md, direction, allowed, hub)}: visits differ"
                assert got[1] == want[1], f"n={n} {'dfs' if dfs else 'bfs'} case {(seeds, md, direction, allowed, hub)}: edges differ"
    with pytest.raises(hv.HelixDbError):
        hv.traverse_host(3, np.array([0, 2, 2, 2], np.uint64), np.array([2, 1], np.uint64), None, [0], 1)   # unsorted row
    with pytest.raises(hv.HelixDbError):
        hv.traverse_host(4, off[:5] * 0, np.zeros(0, np.uint64), None, [9], 1)                                # unknown seed


def test_inline_asm_lds_reads_are_covered_by_a_wait(tmp_path):
    """The large-tile exact-scan kernels read their MFMA fragments with inline-asm ds_read_b128 and state the lgkmcnt waits
    themselves; the compiler treats an asm output as valid at once and may copy / consume it before the LDS has answered
    (seen on hardware: the fp8 instantiation of flat_tile4_kernel returned wrong candidates).  scripts/lint_asm_lds.py walks
    the control-flow graph of the gfx950 assembly: no instruction may touch a requested register before a covering wait."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "helix-db_amd", "csrc", "hvx_flat_tile.hip")
    # the release build carries flat_tile2_kernel only; the tuning build (-DHVX_TUNING) also the experimental flat_tile4_kernel
    for flags, kernels, want in (([], ["flat_tile2_kernel"], 2), (["-DHVX_TUNING"], ["flat_tile2_kernel", "flat_tile4_kernel"], 6)):
        asm = tmp_path / ("hvx_flat_tile%d.s" % want)
        out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only"]
                             + flags + ["-o", str(asm), src], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        lint = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "lint_asm_lds.py"), str(asm)] + kernels, capture_output=True, text=True)
        assert lint.returncode == 0, lint.stdout + lint.stderr
        assert lint.stdout.count(": 0 hazard(s)") == want, lint.stdout    # fp8 + bf16 instantiations of each kernel (tile4: with and without interleaved copies)


def test_asm_lint_flags_an_inline_asm_read_inside_a_matrix_write_back_window(tmp_path):
    """Round 4: the ring build of the small-batch scan stores its accumulators with inline-asm ds_write_b32; the compiler's hazard
    recogniser does not see an asm statement read a register, so it pads nothing between the last v_mfma and the store, and acc[0]
    reached the LDS one MFMA step short (found on hardware by tests/native/smallq_probe.hip).  The lint's second rule fails such
    assembly; the shipped kernels (ring build, MX tile build) pass both rules."""
    import shutil
    import subprocess
    lint = os.path.join(ROOT, "scripts", "lint_asm_lds.py")
    body = ("toy_kernel:\n\ts_waitcnt lgkmcnt(0)\n\tv_mfma_f32_32x32x16_bf16 v[0:15], v[34:37], v[16:19], v[0:15]\n\ts_cbranch_vccz .LBB0_2\n"
            "\t;;#ASMSTART\n\tds_write_b32 v148, v0\n\t;;#ASMEND\n.LBB0_2:\n%s\t;;#ASMSTART\n\tds_write_b32 v148, v1\n\t;;#ASMEND\n\ts_endpgm\n")
    bad = tmp_path / "bad.s"
    bad.write_text(body % "")
    r = subprocess.run([sys.executable, lint, str(bad), "toy_kernel"], capture_output=True, text=True)
    assert r.returncode == 1 and "2 inline-asm read(s)" in r.stdout, r.stdout          # both stores, on both paths into .LBB0_2
    good = tmp_path / "good.s"
    good.write_text((body % "\ts_nop 15\n\ts_nop 7\n").replace("\tds_write_b32 v148, v0\n", "\ts_nop 15\n\ts_nop 7\n\tds_write_b32 v148, v0\n"))
    r = subprocess.run([sys.executable, lint, str(good), "toy_kernel"], capture_output=True, text=True)
    assert r.returncode == 0 and "0 inline-asm read(s)" in r.stdout, r.stdout
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    for src_name, kernels, want in (("hvx_flat_smallb.hip", ["flat_smallq_kernel"], 16), ("hvx_flat_tile.hip", ["flat_tile2mx_kernel"], 1)):
        asm = tmp_path / (src_name + ".s")
        out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only", "-o", str(asm),
                              os.path.join(ROOT, "helix-db_amd", "csrc", src_name)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        r = subprocess.run([sys.executable, lint, str(asm)] + kernels, capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.count(": 0 hazard(s), 0 inline-asm read(s)") == want, r.stdout


def test_link_workgroup_kernel_takes_its_row_locks_without_cache_maintenance(tmp_path):
    """Round 3: an acquire / release at agent scope is an L2 invalidate / write-back of the whole XCD on gfx950 (`buffer_inv sc1` /
    `buffer_wbl2 sc1`); with one per lock operation the batched link step of the device build spent 60 % of its time in them
    (profiles/history/r03p_build_link_wg.txt).  The rows a lock protects are only touched with agent-scope atomics, so the locks of
    build_link_wg_kernel are relaxed atomics + s_waitcnt: its assembly must contain no cache-maintenance instruction, and no scratch."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    asm = tmp_path / "hvx_build.s"
    out = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-S", "--cuda-device-only",
                          "-o", str(asm), os.path.join(ROOT, "helix-db_amd", "csrc", "hvx_build.hip")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    text = asm.read_text()
    kernels = re.findall(r"^(_ZN3hvx20build_link_wg_kernel\w+):[^\n]*\n(.*?)\.end_amdhsa_kernel", text, flags=re.S | re.M)
    assert len(kernels) == 4, [k for k, _ in kernels]       # L2 / cosine x fused / unfused summation tree
    for name, body in kernels:
        assert "buffer_wbl2" not in body and "buffer_inv" not in body, name
        assert "global_atomic_swap" in body                  # the lock itself
        seg = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", body)
        assert seg and int(seg.group(1)) <= 64, (name, seg.group(0) if seg else None)   # a handful of spilled dwords at most


def test_runtime_prepare_sets_the_hardware_queue_count_only_when_the_host_has_not():
    """HIP maps a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); execution lanes that share one run back to
    back (profiles/history/r04t_hw_queues_library_default.log: 0.69 instead of 0.80 of HBM on the headline, 0.51 instead of 0.67 on the bf16
    leg).  Round 5: LOADING the library no longer touches the environment; the host calls hvx_runtime_prepare(n) from its
    start-up code, which sets the variable (0 = 8) unless the host has exported one itself."""
    import subprocess
    lib = os.path.join(ROOT, "helix-db_amd", "libhelix_vec_gfx950.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    code = ("import ctypes, os, sys\n"
            "v, n = sys.argv[2], int(sys.argv[3])\n"
            "os.environ.pop('GPU_MAX_HW_QUEUES', None)\n"
            "if v: os.environ['GPU_MAX_HW_QUEUES'] = v\n"
            "L = ctypes.CDLL(sys.argv[1])\n"
            "libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p\n"
            "before = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
            "rc = L.hvx_runtime_prepare(n) if n >= 0 else 0\n"
            "after = libc.getenv(b'GPU_MAX_HW_QUEUES')\n"
            "print(before.decode() if before else '-', after.decode() if after else '-', rc)\n")
    for preset, n, want in (("", -1, "- - 0"), ("", 0, "- 8 0"), ("", 12, "- 12 0"), ("4", 0, "4 4 0"), ("16", 8, "16 16 0"), ("", 65, "- - 5")):
        r = subprocess.run([sys.executable, "-c", code, lib, preset, str(n)], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0 and r.stdout.strip() == want, (preset, n, r.stdout, r.stderr[-300:])


def test_release_library_reads_no_environment_and_carries_no_measurement_code():
    """tuning switches (kernel ablation -- results wrong by construction --, phase profiling, experimental tile
    builds, stderr path reports) exist only in `make TUNING=1` builds.  The shipped library does not import getenv, holds none
    of the switch names, and does not contain the experimental kernel."""
    import subprocess
    lib = os.path.join(ROOT, "helix-db_amd", "libhelix_vec_gfx950.so")
    if not os.path.exists(lib):
        pytest.skip("library not built")
    undef = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert "getenv" not in undef
    blob = open(lib, "rb").read()
    for name in (b"HVX_FLAT_TILE_ABLATE", b"HVX_FLAT_TILE_BUILD", b"HVX_FLAT_DEBUG", b"HVX_WAVE_PROF", b"HVX_WAVE_OCC", b"HVX_FLAT_VALU",
                 b"HVX_HNSW_GENERAL", b"HVX_WAVE_LOG2CAP", b"HVX_FLAT_CHUNK", b"flat_tile4_kernel"):
        assert name not in blob, name
    src = "".join(open(os.path.join(ROOT, "helix-db_amd", "csrc", f)).read() for f in os.listdir(os.path.join(ROOT, "helix-db_amd", "csrc"))
                  if f.endswith((".hip", ".h")))
    import re
    # every getenv in the sources sits behind HVX_TUNING (tuning_env() or an #ifdef HVX_TUNING block)
    for m in re.finditer(r"[^_a-z]getenv\(", src):
        before = src[: m.start()]
        assert before.rfind("#ifdef HVX_TUNING") > before.rfind("#endif") or before.rfind("#else") > before.rfind("#ifdef HVX_TUNING") >= 0 \
            or "tuning_env" in src[m.start() - 80: m.start()], src[m.start() - 120: m.start() + 40]


def test_tile_workgroup_mapping_covers_every_tile_once():
    """tile_coords / launch_flat_tile256 (csrc/hvx_flat_tile.hip): workgroup id -> (row tile, query tile) through XCD-aware
    super-tiles.  A tile that no workgroup computes is a stretch of rows that is never scanned -- and nothing downstream could
    notice -- so a python twin of the arithmetic sweeps ragged shapes: every (row tile, query tile) exactly once, the rest of
    the grid out of range, for the 64-workgroup (two per CU) and the 32-workgroup (one per CU) super-tiles."""
    def cover(nr, nq, per_xcd):
        sup_q = min(nq, 8)
        sup_r = per_xcd // sup_q
        qblocks = (nq + sup_q - 1) // sup_q
        rb = (nr + 8 * sup_r - 1) // (8 * sup_r)
        grid = 8 * rb * qblocks * sup_r * sup_q
        seen = {}
        for bid in range(grid):
            x, l = bid & 7, bid >> 3
            rq = sup_r * sup_q
            per_rblock = qblocks * rq
            rblock, rem = divmod(l, per_rblock)
            qblock, rem2 = divmod(rem, rq)
            rt = (rblock * sup_r + rem2 // sup_q) * 8 + x
            qt = qblock * sup_q + rem2 % sup_q
            if rt < nr and qt < nq:
                assert (rt, qt) not in seen, (nr, nq, rt, qt)
                seen[(rt, qt)] = bid
        assert len(seen) == nr * nq, (nr, nq, per_xcd, len(seen))
        # the workgroups an XCD runs together (consecutive local ids) share query tiles and row tiles: that is the point
        return grid
    for per_xcd in (64, 32):
        for nq in (1, 2, 3, 4, 5, 7, 8, 9, 16, 17):
            for nr in (1, 2, 7, 8, 9, 63, 64, 65, 127, 500, 1000):
                cover(nr, nq, per_xcd)
    assert cover(97657, 16, 64) < 2**31                       # config #5: 12.5M rows / 128, 4096 queries / 256


def test_tile_operand_order_of_fp8_queries_is_a_permutation():
    """tile_slot_fp8 (csrc/hvx_flat_mfma.h): inside every 64-code stage, MFMA step kk, lane half h, element e must read code
    (2 (kk >> 1) + h) * 16 + (kk & 1) * 8 + e -- the query operand is stored in that order; a python twin checks the formula is a
    bijection of each 64-block and agrees with the read pattern of the kernels."""
    def tile_slot(slot):
        c, u, sb, e = slot & 63, (slot & 63) >> 4, ((slot & 63) >> 3) & 1, slot & 7
        kk, h = (u >> 1) * 2 + sb, u & 1
        return (slot & ~63) + kk * 16 + h * 8 + e
    for base in (0, 64, 1472):
        image = [tile_slot(base + c) for c in range(64)]
        assert sorted(image) == list(range(base, base + 64))
        for kk in range(4):
            for h in range(2):
                for e in range(8):
                    code = (2 * (kk >> 1) + h) * 16 + (kk & 1) * 8 + e
                    assert tile_slot(base + code) == base + kk * 16 + h * 8 + e


def test_row_codecs_and_key_parser_survive_arbitrary_bytes():
    """Persisted bytes come from a store that may be corrupt: every codec must answer with a status, never read past the
    value (deployed rows that fail to decode fail the hydration closed: memory_store.rs:355-361).  8 000 random and
    mutated values; valid prefixes with every truncation length."""
    import pyhvx as hv
    rng = np.random.default_rng(12)
    good0 = bytes([0x13, 0x01, 0, 0, 0, 3]) + (123).to_bytes(8, "little") + _be64(1, 5, 9)
    goodu = (3).to_bytes(4, "big") + _be64(4, 2, 8)
    goodk = bytes([0xF0]) + (77).to_bytes(8, "big") + bytes([0x11]) + (3).to_bytes(2, "big") + (5).to_bytes(8, "big")
    ok = bad = 0
    samples = [bytes(rng.integers(0, 256, int(rng.integers(0, 80)), dtype=np.uint8)) for _ in range(4000)]
    for g in (good0, goodu, goodk):
        samples += [g[:i] for i in range(len(g) + 1)]
        for _ in range(1200):
            m = bytearray(g)
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
            samples.append(bytes(m[: int(rng.integers(0, len(m) + 1))]))
    for v in samples:
        for fn in (hv.decode_layer0_row, hv.decode_upper_row):
            try:
                out = fn(v)
                ids = out[0] if isinstance(out, tuple) else out
                assert len(ids) <= max(0, (len(v) - 4) // 8)      # never more ids than the value can hold
                ok += 1
            except hv.HelixDbError:
                bad += 1
        k = hv.parse_vector_key(v)
        assert k is None or set(k) == {"kind", "index_id", "node_id", "order_code", "layer"}
    assert ok > 100 and bad > 1000
    # the hydrator rejects what the codecs reject and keeps counting rows it accepted
    h = hv.Hydrator(2, hv.COSINE)
    with pytest.raises(hv.HelixDbError):
        h.add_layer0_row(1, bytes([0x12, 0, 0, 0, 2]) + _be64(3))
    with pytest.raises(hv.HelixDbError):
        h.add_item(1, b"\x00\x01\x02")                           # not header || dim x f32


def test_restricted_planner_mirror_follows_the_reference():
    """tests/production_support/vector/restricted.rs:459-590: admission by cardinality OR bytes, the DBpedia budgets,
    beam-percent scaling, deterministic seed sampling, clamp-before-limit of the result count."""
    import pyhvx as hv
    p = hv.SearchParams.new(10)
    assert hv.restricted_execution_plan(256, 1536, p)["plan"] == "exact"
    assert hv.restricted_execution_plan(256, 5000, p)["plan"] == "filtered_graph"        # 256 x 5000 x 4 B > 4 MiB
    assert hv.restricted_execution_plan(257, 2, p)["plan"] == "filtered_graph"           # one id too many
    assert p.ef == 100
    b = hv.restricted_execution_plan(1000, 1536, p)
    assert (b["plan"], b["ef_filtered"], b["sampled_seeds"], b["directory_seeds"], b["vector_payloads"]) == \
        ("filtered_graph", 150, hv.FILTERED_SAMPLED_SEEDS, hv.FILTERED_DIRECTORY_SEEDS, hv.FILTERED_VECTOR_PAYLOAD_LIMIT)
    assert (b["routing_rows"], b["bridge_rows"]) == (2400, 1200)
    for percent, want in [(100, 100), (150, 150), (200, 200), (400, 400)]:
        assert hv.restricted_execution_plan(1000, 1536, p, beam_percent=percent)["ef_filtered"] == want
    ids = np.arange(1, 100_001, dtype=np.uint64)
    sample = hv.deterministic_sample_ids(ids, 256)
    assert len(sample) == 256 and all(a < b_ for a, b_ in zip(sample, sample[1:])) and sample[0] == 1 and sample[-1] == 100_000
    assert hv.deterministic_sample_ids([7], 1) == [7]
    assert hv.deterministic_sample_ids([3, 7], 1) == [3] and hv.deterministic_sample_ids([3, 7], 8) == [3, 7]
    assert hv.restricted_result_count(800, 1000) == 800
    assert hv.restricted_result_count(1000, 800) == 800                                   # clamped before the limit applies
    with pytest.raises(hv.HelixDbError) as e:
        hv.restricted_result_count(801, 1000)
    assert e.value.status == hv.ERR_K_RANGE
    big = hv.restricted_execution_plan(1000, 1536, hv.SearchParams.new(800))
    assert big["k"] == 800 and big["vector_payloads"] == 800
    with pytest.raises(hv.HelixDbError) as e:
        hv.restricted_execution_plan(1_000_001, 8, p)
    assert e.value.status == hv.ERR_CANDIDATE_LIMIT


def test_distance_materialisation_follows_result_rs():
    """result.rs:186-257: current numbers are preserved, squared Euclidean is converted exactly once on request, the
    half-cosine is labelled and never doubled, invalid scores cannot form a result."""
    import pyhvx as hv
    assert hv.materialize_distance(25.0, hv.EUCLIDEAN, hv.CURRENT_SCORE) == (np.float32(25.0), "SquaredEuclideanScore")
    assert hv.materialize_distance(25.0, hv.EUCLIDEAN, hv.METRIC_DISTANCE) == (np.float32(5.0), "EuclideanDistance")
    for version in (hv.CURRENT_SCORE, hv.METRIC_DISTANCE):
        assert hv.materialize_distance(0.25, hv.COSINE, version) == (np.float32(0.25), "HalfCosineScore")
        assert hv.materialize_distance(3.0, hv.MANHATTAN, version) == (np.float32(3.0), "ManhattanDistance")
    for invalid in (float("nan"), float("inf"), -1.0):
        with pytest.raises(hv.HelixDbError):
            hv.materialize_distance(invalid, hv.EUCLIDEAN)


def test_search_params_replay_of_the_reference_unit_test():
    """mod.rs:1186-1268 test_search_params, line by line, against the host mirror."""
    import pyhvx as hv
    P = hv.SearchParams
    params = P.new(10).with_ef(100)
    assert (params.k, params.ef) == (10, 100)
    for bad in (lambda: P.new(0), lambda: P.new(100).with_ef(50), lambda: P.new(10).with_pre_simhash_sampling_ratio(2.0),
                lambda: P.new(10).with_simhash_bypass_tuning(0, 4, 0.12, 3), lambda: P.new(10).with_simhash_bypass_tuning(24, 0, 0.12, 3),
                lambda: P.new(10).with_simhash_bypass_tuning(24, 4, 0.12, 0), lambda: P.new(10).with_simhash_sampling_ratio(float("nan")),
                lambda: P.new(10).with_simhash_failure_prob(1.0)):
        with pytest.raises(hv.HelixDbError):
            bad()
    tuned = P.new(10).with_simhash_mode(hv.SIMHASH_OFF).with_pre_simhash_sampling_ratio(0.75) \
        .with_simhash_bypass_tuning(2, 3, 0.25, 4).with_simhash_sampling_ratio(0.5).with_simhash_failure_prob(0.2)
    assert tuned.simhash_mode == hv.SIMHASH_OFF and tuned.pre_simhash_sampling_ratio_override == 0.75
    assert (tuned.simhash_bypass_min_frontier, tuned.simhash_bypass_window_expansions) == (2, 3)
    assert (tuned.simhash_bypass_min_filter_rate, tuned.simhash_read_budget_multiplier) == (0.25, 4)
    assert tuned.simhash_sampling_ratio_override == 0.5 and tuned.simhash_failure_prob_override == 0.2
    c = tuned._c()
    assert (c.simhash_mode, c.bypass_min_frontier, c.bypass_window_expansions, c.read_budget_multiplier) == (2, 2, 3, 4)
    assert c.pre_simhash_sampling_ratio_override == np.float32(0.75) and c.simhash_failure_prob_override == np.float32(0.2)
    cleared = tuned.clear_pre_simhash_sampling_ratio_override().clear_simhash_sampling_ratio_override() \
        .clear_simhash_failure_prob_override()
    assert cleared.pre_simhash_sampling_ratio_override is None and cleared.simhash_sampling_ratio_override is None
    assert cleared.simhash_failure_prob_override is None and cleared._c().simhash_sampling_ratio_override < 0
    profile = P.throughput_profile_floor_92(10)
    assert profile.ef == 48 and profile.simhash_mode == hv.SIMHASH_ADAPTIVE and profile.pre_simhash_sampling_ratio_override == 0.2


def test_item_rows_are_decoded_like_decode_item(orc):
    """mod.rs:1349-1391 item_decoder_rejects_dimension_finiteness_header_and_trailing_payload_corruption, against the
    hydrator's item-row entry point (`[header f32][dim x f32]`, native-endian: values/vectors/item.rs:34-60)."""
    import pyhvx as hv
    v = np.array([1.0, 2.0, 3.0], np.float32)
    encoded = np.float32(orc.header(orc.COSINE, v)).tobytes() + v.tobytes()       # encode_item: Cosine::new_header || payload
    hv.Hydrator(3, hv.COSINE).add_item(1, encoded)                                  # the row itself decodes
    with pytest.raises(hv.HelixDbError) as e:                                       # DimensionMismatch { expected: 2, actual: 3 }
        hv.Hydrator(2, hv.COSINE).add_item(1, encoded)
    assert e.value.status == hv.ERR_DIMENSION
    with pytest.raises(hv.HelixDbError) as e:                                       # trailing 4.0: DimensionMismatch { 3, 4 }
        hv.Hydrator(3, hv.COSINE).add_item(1, encoded + np.float32(4.0).tobytes())
    assert e.value.status == hv.ERR_DIMENSION
    non_finite = bytearray(encoded)
    non_finite[4:8] = np.float32(np.nan).tobytes()
    with pytest.raises(hv.HelixDbError) as e:                                       # NonFiniteComponent { index: 0 }
        hv.Hydrator(3, hv.COSINE).add_item(1, bytes(non_finite))
    assert e.value.status == hv.ERR_NONFINITE
    wrong_header = bytearray(encoded)
    wrong_header[0] ^= 1
    with pytest.raises(hv.HelixDbError) as e:                                       # HeaderMismatch
        hv.Hydrator(3, hv.COSINE).add_item(1, bytes(wrong_header))
    assert e.value.status == hv.ERR_INVARIANT and "HeaderMismatch" in str(e.value)
    # tests/production_support/vector/primitives.rs:109-131: an empty value (HeaderTooShort) and header + 1 byte (InvalidPayload)
    for broken in (b"", encoded[:4] + b"\x00"):
        with pytest.raises(hv.HelixDbError):
            hv.Hydrator(3, hv.COSINE).add_item(1, broken)
    assert hv.decode_upper_row((3).to_bytes(4, "big") + _be64(1, 2, 3)) == [1, 2, 3]    # primitives.rs:318-320 encode / decode_neighbors
    with pytest.raises(hv.HelixDbError):
        hv.decode_upper_row(bytes(1))
    # Euclidean / Manhattan rows carry a zero bias
    hv.Hydrator(3, hv.EUCLIDEAN).add_item(1, np.float32(0).tobytes() + v.tobytes())
    with pytest.raises(hv.HelixDbError):
        hv.Hydrator(3, hv.EUCLIDEAN).add_item(1, np.float32(1).tobytes() + v.tobytes())
    # f32 extremes: the persisted cosine header of a huge vector is the saturated norm (cosine.rs:128-141)
    huge = np.array([np.finfo(np.float32).max] * 2, np.float32)
    hv.Hydrator(2, hv.COSINE).add_item(2, np.float32(orc.header(orc.COSINE, huge)).tobytes() + huge.tobytes())


def test_device_index_registry_attaches_only_exact_generation_and_sequence():
    """tests/production_support/vector/read_index.rs:60-140: exact generation + snapshot sequence attach the resident
    copy; stale / newer / unavailable visibility, other identities and unfinished hydrations fall back to storage; a
    different distance is a MetricMismatch; guards fence retirement."""
    import pyhvx as hv
    from pyhvx import registry as rg

    class FakeIndex:
        closed = False

        def close(self):
            self.closed = True

    reg = rg.DeviceIndexRegistry()
    ident = rg.VectorCacheIdentity("legacy-unscoped", 4, 1, 40, 1)
    assert reg.entry_for(ident, hv.COSINE) is True            # this caller owns the hydration
    assert reg.entry_for(ident, hv.COSINE) is False           # nobody else does
    assert reg.attach(ident, hv.COSINE, 9) is None            # still hydrating: storage fallback
    ix = FakeIndex()
    assert reg.finish_hydration(ident, ix, 9) is True
    exact = reg.attach(ident, hv.COSINE, 9)
    assert exact is not None and exact.index is ix
    assert reg.attach(ident, hv.COSINE, 10) is None           # stale for a newer snapshot
    assert reg.attach(ident, hv.COSINE, 8) is None            # newer than an older snapshot
    assert reg.attach(ident, hv.COSINE, None) is None         # VectorReadVisibility::Unavailable
    with pytest.raises(rg.MetricMismatch):
        reg.attach(ident, hv.EUCLIDEAN, 9)
    # round 5: the resident copy takes the write batch of sequence 12 (hvx_index_insert_batch) -> requests at 12 attach, 9 no longer
    assert reg.advance(ident, 12) is True
    assert reg.attach(ident, hv.COSINE, 9) is None
    with reg.attach(ident, hv.COSINE, 12) as later:
        assert later.index is ix
    assert reg.advance(ident, 11) is False and reg.advance(ident, 9) is False      # never backwards
    exact.release()
    exact = reg.attach(ident, hv.COSINE, 12)                  # the guard that fences the retirement below
    assert exact is not None
    other = rg.VectorCacheIdentity("legacy-unscoped", 4, 2, 40, 1)   # next generation of the same index
    assert reg.attach(other, hv.COSINE, 9) is None
    assert reg.advance(other, 13) is False                    # never hydrated
    reg.retire(ident)                                         # replaced: no new guards, memory kept while `exact` lives
    assert reg.state(ident) == rg.RETIRING and not ix.closed and reg.attach(ident, hv.COSINE, 12) is None
    assert reg.advance(ident, 14) is False                    # a retiring entry takes no more writes
    exact.release()
    assert ix.closed and reg.state(ident) == rg.CLOSED
    assert reg.entry_for(ident, hv.COSINE) is True            # a closed entry can be hydrated again
    assert reg.finish_hydration(other, FakeIndex(), 9) is False      # never registered: the caller keeps its index


def test_topk_payload_layout_is_shared_by_library_and_host():
    """hvx_topk_payload_bytes == pyhvx.shard.payload_bytes for every (b, k): ids | scores | counts, padded to 8 bytes."""
    import pyhvx as hv
    from pyhvx import shard
    for b, k in [(1, 1), (7, 10), (33, 10), (1024, 10), (1024, 100), (5, 3), (4096, 7)]:
        n = hv.lib().hvx_topk_payload_bytes(b, k)
        assert n == shard.payload_bytes(b, k) and n % 8 == 0 and n >= b * k * 12 + b * 4


def test_shard_group_argument_checks_need_no_device():
    """hvx_shard_group_init validates its plan before touching a device or RCCL (no GPU here)."""
    import ctypes as C
    import pyhvx as hv
    L = hv.lib()
    out = C.c_void_p()
    assert L.hvx_shard_group_init(None, None, 0, 1, 16, 10, C.byref(out)) == hv.ERR_INVARIANT
    assert L.hvx_shard_group_unique_id(None) == hv.ERR_INVARIANT
    assert L.hvx_shard_group_search_batch_device(None, None, 1, 1, 1, None, None, None) == hv.ERR_INVARIANT
    L.hvx_shard_group_free(None)


def _rkyv_string(buf: bytearray, s: bytes):
    """rkyv 0.8 ArchivedString, 32-bit pointers, little-endian: returns a function writing the 8-byte repr at a given position"""
    if len(s) <= 8:
        return lambda pos: bytes(s) + b"\xff" * (8 - len(s))
    at = len(buf)
    buf.extend(s)  # out-of-line bytes are serialised before the object that points at them
    n = len(s)
    v = (n & 0x3F) | 0x80 | ((n & ~0x3F) << 2)
    return lambda pos: v.to_bytes(4, "little") + (at - pos).to_bytes(4, "little", signed=True)


def rkyv_metadata(index_name, property_name, dimension, m, m0, efc, ml, threshold, ratio, adaptive, failure, entry_point, max_layer, count):
    """Python twin of rkyv::to_bytes(&VectorIndexMetadata) per rkyv 0.8's published format (values/vectors/metadata.rs:22-62):
    out-of-line string bytes first, the 88-byte repr(C) root last, 8-aligned."""
    import struct
    buf = bytearray()
    w_name = _rkyv_string(buf, index_name.encode())
    w_prop = _rkyv_string(buf, property_name.encode())
    while len(buf) % 8:
        buf.append(0)
    root = len(buf)
    body = w_name(root) + w_prop(root + 8)
    body += struct.pack("<IIIIfIfB3xf", dimension, m, m0, efc, ml, threshold, ratio, 1 if adaptive else 0, failure)
    body += b"\x00" * 4                                                      # config is 52 bytes; Option<u64> aligns to 8
    body += struct.pack("<B7xQ", 0 if entry_point is None else 1, 0 if entry_point is None else entry_point)
    body += struct.pack("<H6xQ", max_layer, count)
    assert len(body) == 88
    return bytes(buf) + body


def test_tenant_scoped_keys_and_the_metadata_row():
    """keys/tenant.rs:69-95 envelopes in front of every vector key; keys/vectors.rs:23-38 metadata key; the rkyv metadata row
    (values/vectors/metadata.rs) -- its layout is restated from rkyv 0.8's published format: parity UNPINNED (no byte fixture
    exists in the reference), so this test pins self-consistency, the reference's own decode tests (empty / malformed values are
    rejected, metadata.rs:311-314; the field values of :286-309) and memory safety on arbitrary bytes."""
    import ctypes as C
    import pyhvx as hv
    L = hv.lib()
    tenant = (0x0123456789ABCDEF_0FEDCBA987654321).to_bytes(16, "big")
    logical = bytes([0xF1]) + (77).to_bytes(8, "big") + bytes([0x02]) + (0xAABB).to_bytes(8, "big") + (5).to_bytes(8, "big")
    hi, lo = C.c_uint64(0), C.c_uint64(0)
    assert L.hvx_strip_tenant_envelope(logical, len(logical), C.byref(hi), C.byref(lo)) == 0
    scoped = bytes([0xFD]) + tenant + logical
    assert L.hvx_strip_tenant_envelope(scoped, len(scoped), C.byref(hi), C.byref(lo)) == 17
    assert (hi.value, lo.value) == (0x0123456789ABCDEF, 0x0FEDCBA987654321)
    for key in (logical, scoped):
        ix, node, order, layer = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0), C.c_uint32(0)
        assert L.hvx_parse_vector_key(key, len(key), C.byref(ix), C.byref(node), C.byref(order), C.byref(layer)) == 0x02
        assert (ix.value, node.value, order.value) == (77, 5, 0xAABB)
    mkey = bytes([0x03, 0x03]) + (77).to_bytes(8, "big") + bytes([0x01])
    for key in (mkey, bytes([0xFD]) + tenant + mkey):
        ix = C.c_uint64(0)
        assert L.hvx_parse_vector_key(key, len(key), C.byref(ix), None, None, None) == 0x01 and ix.value == 77
    assert L.hvx_parse_vector_key(scoped[:20], 20, None, None, None, None) == 0
    # the metadata row: the values of the reference's codec test (metadata.rs:286-309), then one with an entry point
    v = rkyv_metadata("metadata-codec", "embedding", 3, 16, 32, 200, 0.5, 43, 0.8, True, 0.1, None, 0, 0)
    md = hv.decode_index_metadata(v)
    assert (md["index_name"], md["property_name"], md["dimension"], md["m"], md["m0"], md["ef_construction"]) == ("metadata-codec", "embedding", 3, 16, 32, 200)
    assert md["ml"] == 0.5 and md["simhash_threshold"] == 43 and abs(md["sampling_ratio"] - 0.8) < 1e-7 and md["adaptive_enabled"] == 1
    assert md["entry_point"] is None and md["max_layer"] == 0 and md["count"] == 0
    v2 = rkyv_metadata("idx", "embedding-with-a-long-property-name", 768, 16, 32, 200, 0.36, 43, 0.8, False, 0.1, 123456789012, 5, 1_000_000)
    md2 = hv.decode_index_metadata(v2)
    assert (md2["index_name"], md2["property_name"]) == ("idx", "embedding-with-a-long-property-name")          # inline + out-of-line strings
    assert (md2["dimension"], md2["entry_point"], md2["max_layer"], md2["count"], md2["adaptive_enabled"]) == (768, 123456789012, 5, 1_000_000, 0)
    for bad in (b"", b"malformed"):                                           # metadata.rs:311-314
        with pytest.raises(hv.HelixDbError):
            hv.decode_index_metadata(bad)
    rng = np.random.default_rng(5)
    for _ in range(3000):                                                     # arbitrary / mutated bytes: a status, never a wild read
        m = bytearray(v2 if rng.random() < 0.7 else bytes(rng.integers(0, 256, int(rng.integers(0, 200)), dtype=np.uint8)))
        for _ in range(int(rng.integers(0, 4))):
            if m:
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        cut = bytes(m[: int(rng.integers(0, len(m) + 1))]) if rng.random() < 0.3 else bytes(m)
        out = hv.IndexMetadata()
        assert L.hvx_decode_index_metadata(cut, len(cut), C.byref(out)) in (hv.OK, hv.ERR_INVARIANT)
    # the hydrator takes entry point / top layer from the row and checks the dimension
    h = hv.Hydrator(768, hv.EUCLIDEAN)
    h.set_metadata(v2)
    with pytest.raises(hv.HelixDbError) as e:
        hv.Hydrator(3, hv.EUCLIDEAN).set_metadata(v2)
    assert e.value.status == hv.ERR_DIMENSION


def test_integration_rust_block_declares_every_export():
    """INTEGRATION.md section 2 is the binding a maintainer pastes into crates/db/src/search/vector/gpu/ffi.rs.  Every
    function the header declares -- and the .so exports -- must be in its `extern "C"` block with the right arity and argument
    types (C -> Rust type map of scripts/gen_rust_ffi.py); a declaration that drifts from the header fails here."""
    import importlib.util
    import subprocess
    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "scripts", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    hdr = open(os.path.join(ROOT, "include", "helix_vec.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index('#[link(name = "helix_vec_gfx950")]'):]
    block = block[:block.index("```")]
    got = gen.parse_rust_fns(block)
    protos = gen.c_prototypes(hdr)
    want = {n: ([gen.rust_type(t) for t, _ in p], None if r == "void" else gen.rust_type(r)) for n, r, p in protos}
    assert len(want) >= 80
    assert sorted(set(want) - set(got)) == [], "declared in the header, missing from INTEGRATION.md"
    assert sorted(set(got) - set(want)) == [], "declared in INTEGRATION.md, not in the header"
    for n in want:
        assert got[n] == want[n], (n, got[n], want[n])
    # spot checks of the type map itself
    assert want["hvx_last_error"] == ([], "*const c_char")
    assert want["hvx_index_import"][0][-1] == "*mut *mut hvx_index" and want["hvx_index_import"][0][0] == "*const hvx_index_desc"
    assert want["hvx_topk_payload_bytes"] == (["u32", "u32"], "usize")
    lib = os.path.join(ROOT, "helix-db_amd", "libhelix_vec_gfx950.so")
    if os.path.exists(lib):
        syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True).stdout
        exported = sorted(l.split()[-1] for l in syms.splitlines() if " T hvx_" in l)
        assert exported == sorted(want), "the .so's exports and the header's prototypes differ"


def test_fbin_loader_follows_the_reference_fixture_contract(tmp_path):
    """HELIX_DBPEDIA_1M_FBIN (index_lifecycle_scale.rs:497-534): `<u32 n><u32 dim>` little-endian header + f32 rows, exact file
    length, expected shape, finite values -- the loader bench.py's config #3 leg uses when the variable points at a file."""
    sys.path.insert(0, os.path.join(ROOT, "helix-db_amd"))
    from pyhvx import synth
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((37, 12)).astype(np.float32)
    p = str(tmp_path / "t.fbin")
    synth.write_fbin(p, rows)
    raw = open(p, "rb").read()
    assert raw[:8] == (37).to_bytes(4, "little") + (12).to_bytes(4, "little") and len(raw) == 8 + 37 * 12 * 4
    got = synth.load_fbin(p, expect_rows=37, expect_dim=12)
    assert got.shape == (37, 12) and np.array_equal(np.asarray(got), rows)
    with pytest.raises(ValueError):
        synth.load_fbin(p, expect_rows=38)
    with pytest.raises(ValueError):
        synth.load_fbin(p, expect_dim=16)
    open(p, "ab").write(b"\0\0\0\0")                      # trailing bytes: length no longer matches the header
    with pytest.raises(ValueError):
        synth.load_fbin(p)
    rows[3, 4] = np.inf
    synth.write_fbin(p, rows)
    with pytest.raises(ValueError):
        synth.load_fbin(p)


def test_bench_line_is_compact_strict_json_with_the_contract_keys():
    """round 4's one JSON line had grown to 27.7 KB and the driver could not parse it.  bench.py now prints a compact
    record as the LAST stdout line -- < 4 KB, strict JSON (no NaN / Infinity), the contract's keys + roofline + cpu_baseline + parity
    -- and writes everything else to bench_full.json.  Held here on round 4's full record (tests/golden/r04_full_bench_record.json = profiles/history/r04s_bench_line.json) inflated
    with every leg round 5 added, including non-finite values and an oversized leg."""
    import importlib.util
    import json
    argv = sys.argv
    sys.argv = ["bench.py"]
    try:
        spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
        bench = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(bench)
    finally:
        sys.argv = argv
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "r04_full_bench_record.json")))
    full["roofline"].update({"peak_measured": 6300.5, "frac_of_measured": float("nan")})
    full["cpu_baseline"].update({"threads": 16, "nproc": 256, "quota_cores": 16})
    full["production_default_lanes"] = {m: {"qps": 1.2e6, "ms_per_step": 0.8, "frac": 0.7, "recall_at_10": 0.99, "ids_equal_oracle": True,
                                            "score_bits_equal_oracle": True, "junk": "x" * 500} for m in ("l2", "cosine")}
    full["vendor_gemm"] = {"fp8": {"tflops": 2000.0}, "bf16": {"tflops": float("inf")}}
    full["batcher"]["qps_production_default"] = 1.0e6
    line = bench.compact_record(full, "/root/repo/bench_full.json")
    assert len(line) < bench.COMPACT_LIMIT == 4096 and "\n" not in line
    assert "NaN" not in line and "Infinity" not in line
    rec = json.loads(line, parse_constant=lambda c: (_ for _ in ()).throw(ValueError(c)))   # strict: NaN / Infinity tokens are rejected
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "recall_at_10", "roofline", "cpu_baseline", "parity", "full_record"):
        assert key in rec, key
    assert rec["metric"] == full["metric"] and rec["value"] == full["value"] and rec["ms_per_step"] == full["ms_per_step"]
    for key in ("workload", "dataset", "rows_per_gpu", "dim", "batch", "k", "ef_search", "lanes"):
        assert key in rec["config"], key
    assert "model" not in rec["config"]
    for key in ("bound", "kernel", "achieved", "peak", "peak_measured", "frac", "algorithmic_bytes_per_launch", "kernel_ms", "traffic", "traffic_source"):
        assert key in rec["roofline"], key
    assert rec["roofline"]["frac_of_measured"] is None          # the NaN became null
    assert abs(rec["roofline"]["achieved"] / rec["roofline"]["peak"] - rec["roofline"]["frac"]) < 1e-3
    for key in ("value", "unit", "cores", "threads", "nproc", "kind"):
        assert key in rec["cpu_baseline"], key
    assert rec["parity"]["ids_equal_oracle"] is True and rec["production_default_lanes"]["l2"]["frac"] == 0.7
    assert "junk" not in rec["production_default_lanes"]["l2"]
    # a record whose legs outgrow the budget sheds summaries, never the contract keys
    full["datasets"]["clustered"]["error"] = "y" * 5000
    full["config3_prefilter"]["groups"] = full["config3_prefilter"]["groups"] * 12
    line2 = bench.compact_record(full, "/root/repo/bench_full.json")
    rec2 = json.loads(line2)
    assert len(line2) < 4096 and rec2["value"] == full["value"] and "roofline" in rec2 and "cpu_baseline" in rec2
    # defaults are the driver's command
    sys.argv = ["bench.py"]
    try:
        a = bench.parse()
    finally:
        sys.argv = argv
    assert (a.gpus, a.steps, a.warmup) == (1, 20, 5)
