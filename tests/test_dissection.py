"""The camera-graph dissection behind msfm_ba_create's elimination order (ba.hip: cut order smoothed over the neighbours,
separators as minimum vertex covers of the boundary edges, cuts chosen by the panel chain in 64-column steps) through its
host-only entry point - no GPU needed.  The reduced camera matrix of the reference's DENSE_SCHUR solve (optimizer.cc:42-48,133)
has a block wherever two cameras see a common point: the camera graph here."""
import numpy as np
import pytest

from metricsfm_amd import capi


def grid_graph(n_strips, per_strip, reach_along, reach_across):
    """Cameras on a serpentine grid, coupled within `reach_along` positions of a strip and `reach_across` strips."""
    n = n_strips * per_strip
    s, k = np.divmod(np.arange(n), per_strip)
    adj = (np.abs(s[:, None] - s[None, :]) <= reach_across) & (np.abs(k[:, None] - k[None, :]) <= reach_along)
    np.fill_diagonal(adj, False)
    return adj.astype(np.uint8)


def check_tree(adj, label, n_leaves):
    n = len(label)
    assert set(np.unique(label[label >= 0])) == set(range(n_leaves))
    # no edge between two different leaves: whatever couples them sits in a separator
    a, b = np.nonzero(adj)
    both = (label[a] >= 0) & (label[b] >= 0)
    assert (label[a][both] == label[b][both]).all()


def steps(k, tail=0):
    return -(-(6 * k + tail) // 64)


@pytest.mark.parametrize("shape", [(11, 46, 9, 1), (8, 30, 6, 1), (20, 100, 9, 1)])
def test_dissection_separates_and_beats_the_dense_chain(shape):
    adj = grid_graph(*shape)
    n = adj.shape[0]
    label, n_leaves, chain = capi.camera_graph_dissection(adj, tail_cols=4)
    assert n_leaves >= 2, "a grid of this size is worth cutting"
    check_tree(adj, label, n_leaves)
    dense = steps(n, 4)
    assert chain * 10 <= dense * 8           # the acceptance rule of choose_dissection
    # the reported chain is what the labels say: longest leaf + longest separator of every depth (the root with the tail)
    want = max(steps(int((label == l).sum())) for l in range(n_leaves))
    for d in range(1, 4):
        if (label == -(d + 1)).any():
            assert n_leaves > 2
    # separators of depth >= 1 are several nodes with one label: only their total is known from the labels, so bound it
    root = int((label == -1).sum())
    assert chain >= want + steps(root, 4)
    assert root > 0 and root < n // 3
    # balance: no leaf holds more than 45 % of the cameras once the graph is cut in four or more
    if n_leaves >= 4:
        assert max((label == l).sum() for l in range(n_leaves)) <= 0.45 * n


def test_forced_depths_and_small_or_dense_graphs():
    adj = grid_graph(11, 46, 9, 1)
    for depth, leaves in ((1, 2), (2, 4), (3, 8)):
        label, n_leaves, chain = capi.camera_graph_dissection(adj, force_depth=depth)
        assert n_leaves == leaves
        check_tree(adj, label, n_leaves)
        assert sorted(set(label[label < 0])) == [-(d + 1) for d in range(depth)][::-1]
    # a complete graph cannot be cut; a tiny one is not worth it: the dense order is kept
    full = np.ones((40, 40), np.uint8) - np.eye(40, dtype=np.uint8)
    label, n_leaves, chain = capi.camera_graph_dissection(full)
    assert n_leaves == 0 and (label == 0).all() and chain == steps(40, 4)
    label, n_leaves, _ = capi.camera_graph_dissection(grid_graph(2, 5, 2, 1))
    assert n_leaves == 0
    # two components: no separator is needed between them
    two = np.zeros((300, 300), np.uint8)
    two[:150, :150] = grid_graph(5, 30, 6, 1); two[150:, 150:] = grid_graph(5, 30, 6, 1)
    label, n_leaves, _ = capi.camera_graph_dissection(two, force_depth=1)
    assert n_leaves == 2 and not (label < 0).any() and len(set(label[:150])) == 1 and len(set(label[150:])) == 1


def test_dissection_is_deterministic_and_rejects_bad_input():
    adj = grid_graph(9, 40, 7, 1)
    a = capi.camera_graph_dissection(adj)
    b = capi.camera_graph_dissection(adj)
    assert (a[0] == b[0]).all() and a[1:] == b[1:]
    with pytest.raises(capi.MsfmError):
        capi.camera_graph_dissection(adj, force_depth=7)
