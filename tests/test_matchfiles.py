"""The reference's match file formats: byte layout and round trip."""
import numpy as np

from metricsfm_amd import _abi as A
from metricsfm_amd import matchfiles as mf


def test_match_file_layout_and_round_trip(tmp_path):
    fold = str(tmp_path)
    a = np.array([[5, 0], [7, 3], [2, 9]], np.int32)
    b = np.array([[1, 1]], np.int32)
    mf.write_out_matches(fold, 3, 8, a)
    mf.write_out_matches(fold, 3, 4, np.zeros((0, 2), np.int32))  # nothing written (fine_matching_graph.cc:249-253)
    mf.write_out_matches(fold, 3, 1, b)
    raw = np.fromfile(mf.match_file(fold, 3), dtype="<i4")
    assert raw.tolist() == [8, 3, 5, 0, 7, 3, 2, 9, 1, 1, 1, 1]  # idx2, n, (ptid1, ptid2)*n, appended records
    ids, ms = mf.query_match(fold, 3)
    assert ids == [8, 1] and (ms[0] == a).all() and (ms[1] == b).all()
    assert mf.query_match(fold, 99) == ([], [])
    g = np.array([[0, 3, 0], [12, 0, 7], [0, 0, 0]])
    mf.write_out_match_graph(fold, g)
    assert open(tmp_path / "graph_matching.txt").read() == "0 3 0 \n12 0 7 \n0 0 0 \n"
    assert (mf.read_in_matching_graph(fold, 3) == g).all()


def test_codes_to_matches():
    code = np.array([-1, 4 | A.MSFM_MATCH_GOOD, 9, -1, 0 | A.MSFM_MATCH_GOOD], np.int32)
    good, allm = mf.codes_to_matches(code)
    assert allm.tolist() == [[4, 1], [9, 2], [0, 4]] and good.tolist() == [[4, 1], [0, 4]]
