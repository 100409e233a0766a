"""Parity of the HIP 2-NN matcher (through the C ABI) with the CPU oracle: bit-exact."""
import numpy as np
import pytest

from metricsfm_amd import _abi as A
from metricsfm_amd import scene

pytestmark = pytest.mark.gpu


def _sift_ints(rng, n):
    return scene._sift_like(rng, n).astype(np.float32)


@pytest.mark.parametrize("n_train,n_query", [(2, 1), (64, 64), (65, 130), (300, 77), (1000, 513), (4096, 4096)])
def test_knn2_integer_exact(ctx, oracle, n_train, n_query):
    rng = np.random.default_rng(n_train * 7 + n_query)
    tr, qu = _sift_ints(rng, n_train), _sift_ints(rng, n_query)
    if n_query <= 600:
        ids_r, d_r = oracle.knn2(tr, qu)
    else:
        ids_r, d_r = oracle.knn2(tr, qu, fast=True)  # identical on integer data (tested on CPU)
    ids_g, d_g = ctx.knn2(tr, qu)
    np.testing.assert_array_equal(ids_g, ids_r)
    np.testing.assert_array_equal(d_g, d_r)


def test_knn2_ties_and_duplicates(ctx, oracle):
    """Planted exact duplicates and equidistant rows: lower train index must win, at both ranks."""
    rng = np.random.default_rng(5)
    tr = _sift_ints(rng, 200)
    tr[150] = tr[3]          # duplicate far apart (different LDS tiles)
    tr[17] = tr[16]          # adjacent duplicate
    tr[199] = tr[0]
    qu = np.concatenate([tr[[3, 16, 0, 150]], _sift_ints(rng, 60)])
    qu[10] = tr[40]; qu[10, 0] += 1  # equidistant to two planted rows
    tr[41] = tr[40]; tr[41, 0] += 2
    ids_r, d_r = oracle.knn2(tr, qu)
    ids_g, d_g = ctx.knn2(tr, qu)
    np.testing.assert_array_equal(ids_g, ids_r)
    np.testing.assert_array_equal(d_g, d_r)
    assert (ids_g[0] == [3, 150]).all() and (d_g[0] == 0).all()
    # extreme values: all-zero and all-255 rows (largest possible distance 128*255^2)
    tr2 = np.zeros((3, 128), np.float32); tr2[1] = 255; tr2[2] = 255
    qu2 = np.zeros((3, 128), np.float32); qu2[1] = 255; qu2[2, :64] = 255
    ids_r, d_r = oracle.knn2(tr2, qu2)
    ids_g, d_g = ctx.knn2(tr2, qu2)
    np.testing.assert_array_equal(ids_g, ids_r)
    np.testing.assert_array_equal(d_g, d_r)
    assert d_g.max() == 128 * 255 * 255


def test_knn2_general_floats(ctx, oracle):
    """Non-integer descriptors (VLFeat 512*x floats, feature_extractor_vl_sift.cpp:202; unit-norm
    cudaSift floats): exact binary64 path, bit-identical to the oracle's definition."""
    rng = np.random.default_rng(8)
    tr = (rng.gamma(0.6, 1.0, (333, 128)) * 40).astype(np.float32)
    qu = (rng.gamma(0.6, 1.0, (150, 128)) * 40).astype(np.float32)
    tr[100] = tr[7]
    ids_r, d_r = oracle.knn2(tr, qu)
    ids_g, d_g = ctx.knn2(tr, qu)
    np.testing.assert_array_equal(ids_g, ids_r)
    np.testing.assert_array_equal(d_g, d_r)
    unit = tr / np.linalg.norm(tr, axis=1, keepdims=True)
    ids_r, d_r = oracle.knn2(unit, unit[:50])
    ids_g, d_g = ctx.knn2(unit, unit[:50])
    np.testing.assert_array_equal(ids_g, ids_r)
    np.testing.assert_array_equal(d_g, d_r)


def test_knn2_general_floats_hard_cases(ctx, oracle):
    """The certified f16 path must stay exact where its lists cannot certify the answer:
    many exact duplicates (more than the lists hold), near ties far below the f16 resolution,
    tiny and huge magnitudes (the power-of-two scale), fewer train rows than the lists hold."""
    rng = np.random.default_rng(12)
    base = (rng.gamma(0.6, 1.0, (500, 128)) * 40).astype(np.float32)
    tr = base.copy()
    tr[100:110] = tr[5]                      # 11 identical rows: ties beyond a shortlist of 4
    tr[200] = tr[7] * np.float32(1 + 2e-7)   # near tie, far below 2^-16 relative
    tr[300] = tr[9]; tr[300, 0] += np.float32(1e-3)
    qu = np.concatenate([tr[[5, 7, 9, 100, 200]], (rng.gamma(0.6, 1.0, (300, 128)) * 40).astype(np.float32)])
    for scale in (1.0, 1e-4, 3e3):
        t, q = (tr * np.float32(scale)).astype(np.float32), (qu * np.float32(scale)).astype(np.float32)
        ids_r, d_r = oracle.knn2(t, q)
        ids_g, d_g = ctx.knn2(t, q)
        np.testing.assert_array_equal(ids_g, ids_r)
        np.testing.assert_array_equal(d_g, d_r)
    assert (ids_g[0] == [5, 100]).all()
    for n in (2, 3, 4, 5):                   # n_train around the shortlist size
        ids_r, d_r = oracle.knn2(tr[:n], qu[:40])
        ids_g, d_g = ctx.knn2(tr[:n], qu[:40])
        np.testing.assert_array_equal(ids_g, ids_r)
        np.testing.assert_array_equal(d_g, d_r)
    # mixed signs (descriptors are non-negative in practice, the kernel must not rely on it)
    t = rng.standard_normal((257, 128)).astype(np.float32); q = rng.standard_normal((65, 128)).astype(np.float32)
    ids_r, d_r = oracle.knn2(t, q)
    ids_g, d_g = ctx.knn2(t, q)
    np.testing.assert_array_equal(ids_g, ids_r)
    np.testing.assert_array_equal(d_g, d_r)


def test_float_path_certifies_almost_everything(ctx, oracle):
    """Random non-integral data: the shortlist is certified for (nearly) every query; exact
    duplicates force the exact fallback for the affected queries only.  Integer data: no slow path."""
    rng = np.random.default_rng(3)
    d = [(rng.gamma(0.6, 1.0, (1500, 128)) * 40).astype(np.float32) for _ in range(3)]
    d[0][700:706] = d[0][1]
    d[1][:3] = d[0][1]
    ds = ctx.descset(d)
    res = ds.match_pairs(np.array([[0, 1], [1, 2], [2, 0]], np.int32), keep_knn=True)
    st = res.stats()
    assert st["queries"] == 4500 and 3 <= st["slow_path"] < 0.02 * st["queries"]
    for p, (i, j) in enumerate([(0, 1), (1, 2), (2, 0)]):
        code, ids, dist = res.fetch(p)
        ids_r, d_r = oracle.knn2(d[i], d[j])
        np.testing.assert_array_equal(ids, ids_r)
        np.testing.assert_array_equal(dist, d_r)
        code_r, _, _ = oracle.ratio_codes(ids_r, d_r, 0.6, 0.85)
        np.testing.assert_array_equal(code, code_r)
    ints = [scene._sift_like(rng, 300).astype(np.float32) for _ in range(2)]
    assert ctx.descset(ints).match_pairs(np.array([[0, 1]], np.int32)).stats()["slow_path"] == 0


def test_match_pairs_ratio_codes(ctx, oracle):
    """Batched pairs + fused ratio tests vs fine_matching_graph.cc:116-133 restated; ragged image sizes,
    an empty image, and the counts of matches_all / matches_good."""
    sc = scene.add_features(scene.make_aerial_scene(16, 1500, seed=31), 700, images=range(6))
    descs = [d.copy() for d in sc.desc[:6]]
    descs[2] = descs[2][:333]
    descs[4] = descs[4][:0]  # image without features as query
    pairs = np.array([(i, j) for i in range(6) for j in range(6) if i != j and i != 4], dtype=np.int32)
    ds = ctx.descset(descs)
    res = ds.match_pairs(pairs, 0.6, 0.85, keep_knn=True)
    na, ng = res.counts()
    total_good = 0
    for p, (i, j) in enumerate(pairs):
        code, ids, d = res.fetch(p)
        if len(descs[j]) == 0:
            assert len(code) == 0 and na[p] == 0
            continue
        ids_r, d_r = oracle.knn2(descs[i], descs[j])
        code_r, na_r, ng_r = oracle.ratio_codes(ids_r, d_r, 0.6, 0.85)
        np.testing.assert_array_equal(ids, ids_r)
        np.testing.assert_array_equal(d, d_r)
        np.testing.assert_array_equal(code, code_r)
        assert (na[p], ng[p]) == (na_r, ng_r)
        total_good += ng_r
    assert total_good > 100  # the scene has real correspondences
    # rerun into the same result object is idempotent
    res.rerun()
    na2, ng2 = res.counts()
    np.testing.assert_array_equal(na, na2)
    np.testing.assert_array_equal(ng, ng2)
    # the matches point at the right 3-D points (ground truth of the synthetic scene)
    code, ids, _ = res.fetch(0)
    i, j = pairs[0]
    good = (code >= 0) & ((code & A.MSFM_MATCH_GOOD) != 0)
    assert good.sum() > 10
    assert (sc.feat_point[i][code[good] & A.MSFM_MATCH_ID_MASK] == sc.feat_point[j][np.nonzero(good)[0]]).mean() > 0.95


def test_knn2_bad_arguments(ctx):
    from metricsfm_amd import capi
    one = np.zeros((1, 128), np.float32)
    with pytest.raises(capi.MsfmError) as e:
        ctx.knn2(one, one)  # n_train < 2: the reference would divide by an unset distance
    assert e.value.code == A.MSFM_E_INVAL
    with pytest.raises(capi.MsfmError):
        ctx.knn2(np.zeros((4, 64), np.float32), np.zeros((4, 64), np.float32))  # dim != 128
    ids, d = ctx.knn2(np.zeros((4, 128), np.float32), np.zeros((0, 128), np.float32))  # empty query
    assert ids.shape == (0, 2)


def test_ratio_tests_are_independent(ctx, oracle):
    """fine_matching_graph.cc:118-130 tests `ratio < thRatio_good` and `ratio < thRatio_all` separately: with
    ratio_good > ratio_all a match can be good without being in the all set (MSFM_MATCH_NOT_ALL)."""
    sc = scene.make_ring_scene(3, 600, seed=23)
    scene.add_features(sc, 900)
    ds = ctx.descset(sc.desc)
    res = ds.match_pairs(np.array([[0, 1], [2, 0]], np.int32), ratio_good=0.99, ratio_all=0.9, keep_knn=True)
    na, ng = res.counts()
    for p in range(2):
        code, ids, d = res.fetch(p)
        code_r, na_r, ng_r = oracle.ratio_codes(ids, d, 0.99, 0.9)
        np.testing.assert_array_equal(code, code_r)
        assert (na[p], ng[p]) == (na_r, ng_r) and ng_r > na_r > 0
        only_good = (code >= 0) & ((code & A.MSFM_MATCH_NOT_ALL) != 0)
        assert only_good.sum() == ng_r - na_r and ((code[only_good] & A.MSFM_MATCH_GOOD) != 0).all()
        from metricsfm_amd import matchfiles
        good, allm = matchfiles.codes_to_matches(code)
        assert len(good) == ng_r and len(allm) == na_r


def test_result_goes_stale_when_an_image_is_uploaded_again(ctx):
    """msfm_descset_upload replaces device buffers a live result still points to: rerun / fetch on the old result must
    fail with MSFM_E_INVAL instead of reading recycled memory; a new result sees the new data (incl. the path choice:
    non-integral data uploaded later takes the certified float path, not the int8 kernel)."""
    import ctypes as C
    from metricsfm_amd import capi
    rng = np.random.default_rng(5)
    d = [np.rint(rng.uniform(0, 255, (300, 128))).astype(np.float32) for _ in range(2)]
    ds = ctx.descset(d)
    res = ds.match_pairs(np.array([[0, 1]], np.int32), keep_knn=True)
    code0, ids0, _ = res.fetch(0)
    newd = (rng.uniform(0, 255, (300, 128))).astype(np.float32)      # non-integral
    ctx.check(capi.lib().msfm_descset_upload(ds._h, 1, A.ptr(newd, A.c_float_p), len(newd)))
    for call in (res.rerun, lambda: res.fetch(0)):
        with pytest.raises(capi.MsfmError) as e:
            call()
        assert e.value.code == A.MSFM_E_INVAL
    res2 = ds.match_pairs(np.array([[0, 1]], np.int32), keep_knn=True)
    _, ids2, d2 = res2.fetch(0)
    dd = ((newd[:, None, :].astype(np.float64) - d[0][None, :, :].astype(np.float64)) ** 2).sum(-1)
    np.testing.assert_array_equal(ids2[:, 0], dd.argmin(1))
    assert res2.stats()["queries"] == 300


def test_context_outlives_its_children(oracle):
    """msfm_ctx_destroy with live children keeps the context until the last child goes (their destroy functions use its
    stream); destroy entry points set their own device."""
    from metricsfm_amd import capi
    c = capi.Context(0)
    sc = scene.make_ring_scene(4, 200, seed=3)
    ba = c.ba(A.BaArrays.from_scene(sc))
    rng = np.random.default_rng(1)
    ds = c.descset([np.rint(rng.uniform(0, 255, (64, 128))).astype(np.float32) for _ in range(2)])
    res = ds.match_pairs(np.array([[0, 1]], np.int32))
    ds2 = c.descset([np.rint(rng.uniform(0, 255, (64, 128))).astype(np.float32) for _ in range(2)])
    c.close()                       # children alive: the library keeps the context
    r = ba.run(capi.default_options(max_num_iterations=3))
    assert r["num_iterations"] == 3
    na0, ng0 = res.counts()
    code0, _, _ = res.fetch(0)
    ds.close()                      # the set goes before its result: the result keeps what it owns ...
    na1, ng1 = res.counts()
    code1, _, _ = res.fetch(0)
    assert (na0 == na1).all() and (ng0 == ng1).all() and (code0 == code1).all() and res.stats()["queries"] == 64
    assert capi.lib().msfm_match_pairs_rerun(ds2._h, res._h) == A.MSFM_E_INVAL   # ... but cannot run again, with no set at all
    ds2.close()
    res.close()
    ba.close()                      # last child: the context is released here


def test_float_path_on_clustered_descriptors(ctx, oracle):
    """Adversarial for the certificate: descriptors in tight clusters (dozens of rows within the f16 error bound of each
    other), so most queries cannot be certified and go through the exact brute force - results still bit-identical."""
    rng = np.random.default_rng(21)
    centres = (rng.gamma(0.6, 1.0, (12, 128)) * 40).astype(np.float32)
    tr = (centres[rng.integers(0, 12, 700)] + rng.normal(0, 0.02, (700, 128))).astype(np.float32)
    qu = (centres[rng.integers(0, 12, 260)] + rng.normal(0, 0.02, (260, 128))).astype(np.float32)
    ds = ctx.descset([tr, qu])
    res = ds.match_pairs(np.array([[0, 1], [1, 0]], np.int32), keep_knn=True)
    assert res.stats()["slow_path"] > 200
    for p, (i, j) in enumerate([(0, 1), (1, 0)]):
        d = [tr, qu]
        code, ids, dist = res.fetch(p)
        ids_r, d_r = oracle.knn2(d[i], d[j])
        np.testing.assert_array_equal(ids, ids_r)
        np.testing.assert_array_equal(dist, d_r)
        code_r, na_r, ng_r = oracle.ratio_codes(ids_r, d_r, 0.6, 0.85)
        np.testing.assert_array_equal(code, code_r)
    na, ng = res.counts()
    assert na.sum() >= ng.sum()


def test_float_path_vlfeat_scale_full_tiles(ctx, oracle):
    """512 * unit-norm descriptors (feature_extractor_vl_sift.cpp:202) at the size of a real image pair: several
    256-row key windows, both query sets of every wave, ragged last tile."""
    sc = scene.add_features(scene.make_aerial_scene(12, 1200, seed=33), 1500, images=range(3))
    d = [(512.0 * x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32) for x in sc.desc[:3]]
    d[1] = d[1][:1333]
    ds = ctx.descset(d)
    pairs = np.array([[0, 1], [1, 2], [2, 0], [1, 0]], np.int32)
    res = ds.match_pairs(pairs, keep_knn=True)
    st = res.stats()
    assert st["slow_path"] < 0.05 * st["queries"]
    ng_total = 0
    for p, (i, j) in enumerate(pairs):
        code, ids, dist = res.fetch(p)
        ids_r, d_r = oracle.knn2(d[i], d[j])
        np.testing.assert_array_equal(ids, ids_r)
        np.testing.assert_array_equal(dist, d_r)
        code_r, _, ng_r = oracle.ratio_codes(ids_r, d_r, 0.6, 0.85)
        np.testing.assert_array_equal(code, code_r)
        ng_total += ng_r
    assert ng_total > 100
    # without keep_knn only the codes and counts exist; they must not change
    res2 = ds.match_pairs(pairs, keep_knn=False)
    for p in range(len(pairs)):
        np.testing.assert_array_equal(res2.fetch(p)[0], res.fetch(p)[0])
    np.testing.assert_array_equal(res2.counts()[0], res.counts()[0])


def _check_pairs_against_oracle(ctx, oracle, descs, pairs, fast=True):
    """ids, distances, ratio codes and the two counts of every pair in `pairs` identical to oracle.knn2 + ratio_codes."""
    oracle.set_num_threads(oracle.host_cores())
    try:
        ds = ctx.descset(descs)
        res = ds.match_pairs(pairs, 0.6, 0.85, keep_knn=True)
        na, ng = res.counts()
        stats = res.stats()
        n_good = 0
        for p, (i, j) in enumerate(pairs):
            code, ids, d = res.fetch(p)
            ids_r, d_r = oracle.knn2(descs[i], descs[j], fast=fast)   # fast: identical to the plain form on integer data (tests/test_oracle.py)
            code_r, na_r, ng_r = oracle.ratio_codes(ids_r, d_r, 0.6, 0.85)
            assert np.array_equal(ids, ids_r) and np.array_equal(d, d_r) and np.array_equal(code, code_r), (p, i, j)
            assert (na[p], ng[p]) == (na_r, ng_r), (p, i, j)
            n_good += ng_r
        # the codes-only form (what the reference's loop consumes; the float kernel then decides most codes from certified
        # distance intervals without an exact evaluation): same codes, same counts
        res0 = ds.match_pairs(pairs, 0.6, 0.85, keep_knn=False)
        na0, ng0 = res0.counts()
        assert np.array_equal(na0, na) and np.array_equal(ng0, ng)
        for p in range(len(pairs)):
            assert np.array_equal(res0.fetch(p)[0], res.fetch(p)[0]), p
        res0.close()
        res.close()
        ds.close()
    finally:
        oracle.set_num_threads(1)
    return n_good, stats


def test_config2_every_pair_matches_the_oracle(ctx, oracle):
    """BASELINE config 2 in full: all 2 450 ordered pairs of its 50 images x 4096 features (initial_matching_graph.cc:55-63,
    fine_matching_graph.cc:87-133) - 10 M queries, each compared with the CPU oracle (OpenMP over the queries)."""
    sc = scene.add_features(scene.config_scene(2), 4096)
    pairs = scene.all_pairs(sc.n_cams)
    assert len(pairs) == 2450
    n_good, stats = _check_pairs_against_oracle(ctx, oracle, sc.desc, pairs)
    assert stats["slow_path"] == 0 and n_good > 50000


def test_config3_pair_sample_matches_the_oracle(ctx, oracle):
    """A seeded sample of 96 of config 3's 249 500 ordered pairs (500 images x 4096 features), matched as one batch the way
    bench.py matches the whole list, against the oracle."""
    sc = scene.config_scene(3)
    rng = np.random.default_rng(0x4D53464D + 11)
    all_pairs = scene.all_pairs(sc.n_cams)
    pairs = all_pairs[np.sort(rng.choice(len(all_pairs), 96, replace=False))]
    pairs = np.concatenate([pairs, [[0, 1], [1, 0], [498, 499], [499, 0]]]).astype(np.int32)   # neighbours on the flight line: real matches
    images = sorted(set(pairs.ravel().tolist()))
    scene.add_features(sc, 4096, images=images)
    descs = [sc.desc[i] if sc.desc[i] is not None else np.zeros((0, 128), np.float32) for i in range(sc.n_cams)]
    n_good, stats = _check_pairs_against_oracle(ctx, oracle, descs, pairs)
    assert stats["slow_path"] == 0 and n_good > 100


def test_config3_float_pair_sample_matches_the_oracle(ctx, oracle):
    """The path the reference's real extractors feed (feature_extractor_vl_sift.cpp:202: `512.0F * x`, never cast) at
    configuration scale: a seeded sample of 20 of config 3's ordered pairs + 4 neighbours on the flight line, 4096 features
    each, `512 * unit-norm` floats exactly as bench.py's `matching_float` leg builds them, matched as ONE batch
    (k_knn2_f16 + the grouped filing of uncertified queries into k_exact_flagged) and compared id for id, distance for
    distance, code for code and count for count with the oracle's binary64 definition (fine_matching_graph.cc:87-133)."""
    sc = scene.config_scene(3)
    rng = np.random.default_rng(0x4D53464D + 12)
    all_pairs = scene.all_pairs(sc.n_cams)
    pairs = all_pairs[np.sort(rng.choice(len(all_pairs), 20, replace=False))]
    pairs = np.concatenate([pairs, [[0, 1], [1, 0], [498, 499], [499, 0]]]).astype(np.int32)
    images = sorted(set(pairs.ravel().tolist()))
    scene.add_features(sc, 4096, images=images)
    descs = [(512.0 * sc.desc[i] / np.linalg.norm(sc.desc[i], axis=1, keepdims=True)).astype(np.float32) if sc.desc[i] is not None
             else np.zeros((0, 128), np.float32) for i in range(sc.n_cams)]
    n_good, stats = _check_pairs_against_oracle(ctx, oracle, descs, pairs, fast=False)
    assert stats["queries"] == 24 * 4096
    assert 0 < stats["slow_path"] < 0.02 * stats["queries"]     # the slow path ran at this size, and stayed the exception
    assert n_good > 100


def test_float_codes_without_knn_arrays_are_the_oracles(ctx, oracle):
    """Codes-only matching of float descriptors (round 5: nearest index and both ratio outcomes from certified intervals, exact
    evaluation only where an interval straddles a threshold): queries planted ON the two thresholds - second-nearest rows
    constructed so that d0 / d1 lands within a few float ulps of 0.6 and 0.85 on either side -, exact duplicates (0 / d and
    0 / 0), near ties for the nearest row, clustered rows, three magnitudes, thresholds in the other order, and the SLAM form
    (`ratio > th` rejects, equality and NaN pass; slam_gps.cc:469-477)."""
    rng = np.random.default_rng(77)
    tr = (512.0 * (lambda x: x / np.linalg.norm(x, axis=1, keepdims=True))(rng.gamma(0.6, 1.0, (900, 128)))).astype(np.float32)
    qs = []
    for k, th in enumerate([0.6, 0.85] * 12):
        # query = train row k displaced by u (nearest: row k at |u|^2); plant row 400 + k at distance^2 = |u|^2 / th (1 + eps)
        u = rng.standard_normal(128); u *= 20.0 / np.linalg.norm(u)
        w = rng.standard_normal(128); w -= w.dot(u) / u.dot(u) * u; w /= np.linalg.norm(w)
        q = tr[k].astype(np.float64) + u
        eps = (k // 2 - 6) * 3e-8
        tr[400 + k] = (q + np.sqrt(400.0 / th * (1 + eps)) * w).astype(np.float32)
        qs.append(q.astype(np.float32))
    qu = np.concatenate([np.array(qs), tr[[5, 6]], (512.0 * (lambda x: x / np.linalg.norm(x, axis=1, keepdims=True))(rng.gamma(0.6, 1.0, (300, 128)))).astype(np.float32)])
    tr[700] = tr[5]                                  # 0 / 0
    tr[701] = tr[30] * np.float32(1 + 3e-7)          # near tie for the nearest row of a later query
    qu[40] = tr[30]
    cl = (tr[50] + rng.normal(0, 0.01, (40, 128))).astype(np.float32)   # a tight cluster: nothing certifies there
    tr[800:840] = cl
    qu[41:45] = cl[:4] + np.float32(0.003)
    for scale in (1.0, 2.0 ** -12, 37.0):
        d = [(tr * np.float32(scale)).astype(np.float32), (qu * np.float32(scale)).astype(np.float32)]
        ds = ctx.descset(d)
        pairs = np.array([[0, 1], [1, 0]], np.int32)
        for rg, ra in ((0.6, 0.85), (0.99, 0.9)):
            res = ds.match_pairs(pairs, rg, ra, keep_knn=False)
            na, ng = res.counts()
            for p, (i, j) in enumerate(pairs):
                ids_r, d_r = oracle.knn2(d[i], d[j])
                code_r, na_r, ng_r = oracle.ratio_codes(ids_r, d_r, rg, ra)
                np.testing.assert_array_equal(res.fetch(p)[0], code_r)
                assert (na[p], ng[p]) == (na_r, ng_r)
            res.close()
        # the SLAM form shares the kernel: compare with the keep_knn form, which evaluates every candidate exactly
        kp = [rng.uniform(-100, 100, (len(x), 2)).astype(np.float32) for x in d]
        for i in range(2):
            ds.upload_keypoints(i, kp[i])
        F = np.tile(np.zeros((3, 3)), (2, 1, 1)); H = np.tile(np.eye(3), (2, 1, 1))
        a = ds.match_pairs_slam(pairs, F, H, keep_knn=True, th_epipolar=1e9, th_distance=1e9)
        b = ds.match_pairs_slam(pairs, F, H, keep_knn=False, th_epipolar=1e9, th_distance=1e9)
        for p in range(2):
            np.testing.assert_array_equal(a.fetch(p)[0], b.fetch(p)[0])
        np.testing.assert_array_equal(a.counts()[1], b.counts()[1])
        a.close(); b.close(); ds.close()


def _prior_F_H(sc, i, j):
    """Prior matrices of an image pair from the scene's true geometry: F with x2^T F x1 = 0 on centred pixels and the
    homography of the ground plane z = 0 - what slam_gps.cc:386-402 estimates from the shared points."""
    K = lambda c: np.diag([sc.cam_model_gt[sc.cam_model_of_cam[c], 0]] * 2 + [1.0])
    R1, R2 = scene.angle_axis_to_R(sc.cam_pose_gt[[i, j], :3])
    t1, t2 = sc.cam_pose_gt[i, 3:], sc.cam_pose_gt[j, 3:]
    R = R2 @ R1.T
    t = t2 - R @ t1
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    F = np.linalg.inv(K(j)).T @ tx @ R @ np.linalg.inv(K(i))
    n = R1 @ np.array([0, 0, 1.0])          # plane z = 0 in camera-1 coordinates: n . X = d
    d = n @ t1
    H = K(j) @ (R + np.outer(t, n) / d) @ np.linalg.inv(K(i))
    return F / np.abs(F).max(), H / H[2, 2]


def test_match_pairs_slam_gates(ctx, oracle):
    """msfm_match_pairs_slam = the matching loop of SLAMGPS::FeatureMatching step 2 (slam_gps.cc:455-503): the `>` ratio test
    and the prior F / H gates, against the oracle on the 2-NN arrays of the same call - codes and both counts identical;
    ragged image sizes, an empty query image, planted duplicates (0/0 ratios) and a rerun."""
    from metricsfm_amd import capi
    sc = scene.add_features(scene.make_aerial_scene(16, 2500, seed=41), 900, images=range(5))
    descs = [d.copy() for d in sc.desc[:5]]
    kps = [k.copy() for k in sc.kp_xy[:5]]
    descs[3], kps[3] = descs[3][:401], kps[3][:401]
    descs[4], kps[4] = descs[4][:0], kps[4][:0]
    descs[1][7] = descs[1][5]                                  # two identical train rows ...
    descs[0][11] = descs[1][5]                                 # ... and a query equal to both: d0 = d1 = 0, ratio NaN
    pairs = np.array([(i, j) for i in range(4) for j in range(5) if i != j], dtype=np.int32)
    FH = [_prior_F_H(sc, i, j) for i, j in pairs]
    F, H = np.array([f for f, _ in FH]), np.array([h for _, h in FH])
    ds = ctx.descset(descs, keypoints=kps)
    res = ds.match_pairs_slam(pairs, F, H, keep_knn=True, th_epipolar=2.0, th_distance=5.0)
    na, ng = res.counts()
    kept_total = 0
    for p, (i, j) in enumerate(pairs):
        code, ids, d = res.fetch(p)
        if len(descs[j]) == 0:
            assert len(code) == 0 and na[p] == 0 and ng[p] == 0
            continue
        ids_r, d_r = oracle.knn2(descs[i], descs[j])
        np.testing.assert_array_equal(ids, ids_r)
        np.testing.assert_array_equal(d, d_r)
        code_r, nr_r, nk_r = oracle.slam_gate(ids_r, d_r, kps[i], kps[j], F[p], H[p], 0.80, 2.0, 5.0)
        np.testing.assert_array_equal(code, code_r)
        assert (ng[p], na[p]) == (nr_r, nk_r), (p, i, j)
        assert nk_r <= nr_r
        kept_total += nk_r
        if (i, j) == (1, 0):
            assert d_r[11, 0] == 0 and d_r[11, 1] == 0 and ids_r[11].tolist() == [5, 7]   # the NaN ratio passed check1 (`>`)
    assert kept_total > 200
    # the survivors are true correspondences of the synthetic scene (the gates use the true geometry)
    code, _, _ = res.fetch(0)
    i, j = pairs[0]
    ok = code >= 0
    assert ok.sum() > 20 and (sc.feat_point[i][code[ok]] == sc.feat_point[j][np.nonzero(ok)[0]]).mean() > 0.9
    # the plain ratio test of fine_matching_graph.cc (`<`) keeps fewer: it drops the 0/0 match and anything at equality
    res.rerun()
    na2, ng2 = res.counts()
    np.testing.assert_array_equal(na, na2)
    np.testing.assert_array_equal(ng, ng2)
    # keypoints uploaded again: the gate tables of the result point at the freed positions -> refused, not read
    ds.upload_keypoints(2, kps[2] + 1.0)
    with pytest.raises(capi.MsfmError) as e:
        res.rerun()
    assert e.value.code == A.MSFM_E_INVAL
    res.close()
    res3 = ds.match_pairs_slam(pairs, F, H)                    # a new result sees the new positions and runs
    res3.rerun()
    res3.close()
    # without keypoints the gates cannot run
    ds2 = ctx.descset(descs)
    with pytest.raises(capi.MsfmError) as e:
        ds2.match_pairs_slam(pairs, F, H)
    assert e.value.code == A.MSFM_E_INVAL
    ds2.close()
    ds.close()
