"""Neighbor-file contract against goldens produced by the reference's own Python
(tests/golden/make_golden.py imports /root/reference/textreact/dataset.py)."""
import json
import os
import random

import pytest

from textreact_amd import neighbors as nb

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def golden():
    with open(os.path.join(G, "neighbors_expected.json")) as f:
        return json.load(f)


def test_reader_matches_reference_load_corpus(golden):
    got = nb.read_neighbors(os.path.join(G, "neighbors_input.json"))
    for case in golden["cases"]:
        assert got == case["neighbors"]


def test_get_neighbor_text_matches_reference(golden):
    with open(os.path.join(G, "neighbors_input.json")) as f:
        ids = [e["id"] for e in json.load(f)]
    for case in golden["cases"]:
        st = nb.NeighborStore(ids, split=case["split"], rng=random, **case["args"])
        st.skip_gold_neighbor = case["skip_gold"]
        st.load_corpus(golden["corpus"], os.path.join(G, "neighbors_input.json"))
        random.seed(1234)
        lists = [st.get_neighbor_text(i, return_list=True) for i in range(len(ids))]
        random.seed(1234)
        texts = [st.get_neighbor_text(i) for i in range(len(ids))]
        assert lists == case["lists"], case
        assert texts == case["texts"], case


def test_writer_is_byte_identical_to_json_dump(tmp_path):
    import numpy as np
    rank = np.array([[2, 0, 1], [1, 2, -1]])
    res = nb.build_result(np.array([10, 11]), rank, ["a", "b", "c"])
    assert res == [{"id": 10, "nn": ["c", "a", "b"]}, {"id": 11, "nn": ["b", "c"]}]
    p = tmp_path / "train.json"
    nb.write_neighbors(str(p), res)
    assert p.read_text() == json.dumps(res)  # default separators, as retrieve_faiss.py:117-118
    assert nb.read_neighbors(str(p)) == {10: ["c", "a", "b"], 11: ["b", "c"]}


def test_convert_tevatron(tmp_path):
    # retrieve/convert_format.py:6-16
    lines = [json.dumps({"query_id": "q%d" % i, "negative_passages": [{"docid": "d%d" % j, "text": "x"} for j in range(3)]})
             for i in range(4)]
    src = tmp_path / "test.jsonl"
    src.write_text("\n".join(lines) + "\n")
    dst = tmp_path / "test.json"
    assert nb.convert_tevatron_file(str(src), str(dst)) == 4
    assert json.loads(dst.read_text())[2] == {"id": "q2", "nn": ["d0", "d1", "d2"]}


def test_the_block_writer_emits_json_dumps_bytes(tmp_path):
    """write_neighbor_file == json.dump(build_result(...)) byte for byte: int ids, numpy ints, strings with quotes / unicode /
    backslashes, floats, pads (-1) in any place, an empty result, more rows than one block"""
    import numpy as np
    from textreact_amd import neighbors as N
    rng = np.random.default_rng(0)
    cases = []
    ids_int = np.arange(1000, 1400)
    ids_str = ['r"%d\\\\x' % i if i % 7 == 0 else "US0%d-é☃" % i for i in range(400)]
    ids_mixed = [1.5, "a", 7, True, None] * 80
    for cids in (ids_int, ids_str, ids_mixed, list(ids_int)):
        rank = rng.integers(0, len(cids), (95, 20))
        rank[3, 5:] = -1; rank[4, :] = -1; rank[5, 0] = -1; rank[6, ::2] = -1
        qids = [cids[i] for i in rng.integers(0, len(cids), 95)]
        cases.append((qids, rank, cids))
    cases.append(([], np.empty((0, 20), np.int64), ids_int))
    cases.append((list(range(5)), np.full((5, 3), -1), []))
    for i, (qids, rank, cids) in enumerate(cases):
        a, b = tmp_path / ("a%d.json" % i), tmp_path / ("b%d.json" % i)
        N.write_neighbors(a, N.build_result(qids, rank, cids))
        N.write_neighbor_file(b, qids, rank, cids, block=16)
        assert a.read_bytes() == b.read_bytes(), i


def test_the_block_writer_equals_json_dump_for_arbitrary_ids():
    """property test (hypothesis): any mix of string / integer ids, any ranks with pads, any block size -> json.dump's bytes"""
    import numpy as np
    import tempfile, os
    from hypothesis import given, settings, strategies as st
    from textreact_amd import neighbors as N
    ids = st.one_of(st.integers(-10 ** 12, 10 ** 12), st.text(max_size=12))

    @settings(max_examples=40, deadline=None)
    @given(st.lists(ids, min_size=1, max_size=30), st.integers(1, 6), st.integers(0, 25), st.integers(1, 9), st.randoms(use_true_random=False))
    def check(cids, k, nq, block, rnd):
        rank = np.array([[rnd.randint(-1, len(cids) - 1) for _ in range(k)] for _ in range(nq)], dtype=np.int64).reshape(nq, k)
        qids = [cids[rnd.randrange(len(cids))] for _ in range(nq)]
        with tempfile.TemporaryDirectory() as td:
            a, b = os.path.join(td, "a.json"), os.path.join(td, "b.json")
            N.write_neighbors(a, N.build_result(qids, rank, cids))
            N.write_neighbor_file(b, qids, rank, cids, block=block)
            assert open(a, "rb").read() == open(b, "rb").read()
    check()
