"""CPU-side checks: the C-ABI library loads and exports every symbol include/helix_vec.h declares,
and the host-side mirror of the reference interface behaves like the reference (no GPU needed)."""
import ctypes
import os
import re

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
