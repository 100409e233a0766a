"""The reference's on-disk match formats, so that matches produced by libmsfm can
be consumed by the unmodified incremental pipeline and vice versa.

  <output_fold>/<idx1>_match   binary, appended per accepted pair: int32 idx2, int32 n, int32[2n] (ptid1, ptid2)
                               (FineMatchingGraph::WriteOutMatches, SfM/src/graph/fine_matching_graph.cc:247-272;
                                read back by Graph::QueryMatch, SfM/src/graph.cc:92-137)
  <output_fold>/graph_matching.txt   text, N rows of N counts separated by ' ', each row ends with ' \\n'
                               (WriteOutMatchGraph, fine_matching_graph.cc:275-292; Graph::ReadinMatchingGraph, graph.cc:72-85)
Native little-endian ints, as the reference writes them with ofstream::write.
"""
import os

import numpy as np

from . import _abi as A


def match_file(fold, idx1):
    return os.path.join(fold, "%d_match" % idx1)


def write_out_matches(fold, idx1, idx2, matches):
    """Append one record; like the reference, an empty match list writes nothing."""
    m = np.asarray(matches, dtype=np.int32).reshape(-1, 2)
    if len(m) == 0:
        return
    with open(match_file(fold, idx1), "ab") as f:
        np.array([idx2, len(m)], dtype=np.int32).tofile(f)
        np.ascontiguousarray(m).tofile(f)


def query_match(fold, idx):
    """-> (image_ids, [matches [n,2]]) in file order (Graph::QueryMatch, graph.cc:92-121)."""
    ids, out = [], []
    path = match_file(fold, idx)
    if not os.path.exists(path):
        return ids, out
    raw = np.fromfile(path, dtype=np.int32)
    p = 0
    while p + 2 <= len(raw):
        idx2, n = int(raw[p]), int(raw[p + 1])
        p += 2
        ids.append(idx2)
        out.append(raw[p:p + 2 * n].reshape(n, 2).copy())
        p += 2 * n
    return ids, out


def write_out_match_graph(fold, match_graph):
    g = np.asarray(match_graph, dtype=np.int64)
    with open(os.path.join(fold, "graph_matching.txt"), "wb") as f:
        for row in g:
            f.write(("".join("%d " % v for v in row) + "\n").encode())


def read_in_matching_graph(fold, n):
    vals = np.array(open(os.path.join(fold, "graph_matching.txt")).read().split(), dtype=np.int64)
    return vals.reshape(n, n)


def codes_to_matches(code):
    """(ptid1, ptid2) lists of the reference loop (fine_matching_graph.cc:116-133) from msfm_match_pairs codes."""
    code = np.asarray(code)
    m2 = np.nonzero(code >= 0)[0].astype(np.int32)
    m1 = (code[m2] & A.MSFM_MATCH_ID_MASK).astype(np.int32)
    good = (code[m2] & A.MSFM_MATCH_GOOD) != 0
    in_all = (code[m2] & A.MSFM_MATCH_NOT_ALL) == 0      # the two ratio tests are independent (:118-130)
    allm = np.column_stack([m1, m2])
    return allm[good], allm[in_all]
