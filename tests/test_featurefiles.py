"""The reference's <idx>_feature file and bundle.rd.out export: byte layout and round trips."""
import struct

import numpy as np

from metricsfm_amd import featurefiles as F


def test_feature_file_layout_and_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    info = dict(rows=3000, cols=4000, zoom_ratio=0.5, f_mm=24.0, f_pixel=4800.0, gps_latitude=31.5, gps_longitude=118.75,
                cam_maker="DJI", cam_model="FC6310")
    kp = rng.uniform(0, 3000, (7, 2))
    desc = rng.integers(0, 255, (7, 128)).astype(np.float32)
    F.write_image_feature(str(tmp_path), 3, info, kp, desc)
    raw = open(F.feature_file(str(tmp_path), 3), "rb").read()
    # database.cc:499-539, field by field
    assert struct.unpack_from("<ii5f", raw, 0) == (3000, 4000, 0.5, 24.0, 4800.0, 31.5, 118.75)
    assert struct.unpack_from("<i", raw, 28)[0] == 3 and raw[32:35] == b"DJI"
    assert struct.unpack_from("<i", raw, 35)[0] == 6 and raw[39:45] == b"FC6310"
    assert struct.unpack_from("<i", raw, 45)[0] == 7
    first = struct.unpack_from("<2f", raw, 49)
    assert first == (np.float32(kp[0, 0] - 2000.0), np.float32(kp[0, 1] - 1500.0))          # centred on write
    assert struct.unpack_from("<iii", raw, 49 + 56) == (7, 128, 5)                           # CV_32FC1
    assert len(raw) == 49 + 56 + 12 + 7 * 128 * 4
    info2, kp2, desc2 = F.read_image_feature(str(tmp_path), 3)
    assert info2 == info
    np.testing.assert_array_equal(kp2, (kp - [2000.0, 1500.0]).astype(np.float32))
    np.testing.assert_array_equal(desc2, desc)
    F.write_image_feature(str(tmp_path), 4, info, kp, desc.astype(np.uint8))
    assert F.read_image_feature(str(tmp_path), 4)[2].dtype == np.uint8


def test_bundle_out_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    fk = np.array([[4800.0, 1e-3, -2e-3], [4790.5, 0.0, 0.0]])
    R = np.stack([np.eye(3).reshape(9), np.eye(3)[::-1].reshape(9)])
    t = rng.normal(size=(2, 3))
    pts = rng.normal(size=(3, 3))
    views = [[(0, 5, 10.7, -3.9), (1, 1000007, 99.2, 4.0)], [(1, 1000009, -0.5, 0.5)], []]
    path = str(tmp_path / "bundle.rd.out")
    F.write_bundle_out(path, fk, R, t, pts, views)
    lines = open(path).read().split("\n")
    assert lines[0] == "# Bundle file v0.3" and lines[1] == "2 3"
    assert lines[2] == "4800.00000000 0.00100000 -0.00200000"
    fk2, R2, t2, pts2, views2 = F.read_bundle_out(path)
    np.testing.assert_allclose(fk2, fk, atol=1e-8); np.testing.assert_allclose(R2, R, atol=1e-8)
    np.testing.assert_allclose(t2, t, atol=1e-8); np.testing.assert_allclose(pts2, pts, atol=1e-8)
    assert views2[0] == [(0, 5, 10.0, -3.0), (1, 1000007, 99.0, 4.0)]      # image coordinates truncated to int (:1342-1345)
    assert views2[1] == [(1, 1000009, 0.0, 0.0)] and views2[2] == []


def test_openmvs_export_with_reference_quirks(tmp_path):
    cams = [dict(image_path="D:\\data\\img_0001.jpg", f=4800.0, R=np.eye(3).reshape(9), t=[0.0, 0.0, 0.0], id=7, px=2000.0, py=1500.0, w=4000, h=3000),
            dict(image_path="D:\\data\\img_0002.jpg", f=4801.5, R=np.eye(3).reshape(9), t=[1.0, 2.0, 3.0], id=9, px=100.0, py=100.0, w=200, h=200)]
    points = [dict(X=[1.0, 2.0, 3.0], views=[(0, -10.4, 20.9), (1, 50.0, -50.0)]),          # both inside their own image
              dict(X=[4.0, 5.0, 6.0], views=[(1, 150.0, 0.0), (0, 0.0, 0.0)]),               # first view outside camera 1 -> counted 1 -> dropped ... 
              dict(X=[7.0, 8.0, 9.0], bad=True, views=[(0, 0.0, 0.0), (1, 0.0, 0.0)]),
              dict(X=[0.5, 0.5, 0.5], views=[(0, 1500.0, 0.0), (1, 1500.0, 0.0)])]          # counted against camera 0 (both fine), written: only the first
    path = str(tmp_path / "sfm_openmvs.txt")
    F.write_openmvs(path, cams, points)
    lines = open(path).read().split("\n")
    assert lines[0] == "2" and lines[1] == "\\img_0001.jp" and lines[2] == "4800.00000000"     # substr(t, size - 1 - t)
    assert lines[9] == "2"                                                                     # good points
    assert lines[10] == "1.00000000 2.00000000 3.00000000 255 255 255 2"
    assert lines[11] == "7 1989 1520" and lines[12] == "9 150 50"                              # int(-10.4 + 2000), int(20.9 + 1500)
    assert lines[13] == "0.50000000 0.50000000 0.50000000 255 255 255 2"                       # the count says 2 ...
    assert lines[14] == "7 3500 1500" and lines[15] == ""                                      # ... one view is written (camera 1 is 200 px wide)


def test_temp_result_checkpoint_round_trip(tmp_path):
    st = dict(cam_models=[dict(id=0, cam_maker="DJI", cam_model="FC 6310", w=4000, h=3000, f_mm=8.8, f=4799.643, f_hyp=4800.0, px=2000.0,
                               py=1500.0, k1=1.048e-3, k2=3.789e-3, data=[4799.643, 1.048e-3, 3.789e-3], num_cams=2)],
              cams=[dict(id_img=0, model_id=0, is_mutable=True, data=[0.1, -0.2, 0.3, 1.0, 2.0, 3.0], pts=[(5, 0), (9, 1)], visible_cams=[0, 1]),
                    dict(id_img=1, model_id=0, is_mutable=False, data=[0.0] * 6, pts=[(1000007, 0)], visible_cams=[1, 0])],
              pts=[dict(id=0, is_mutable=True, is_bad_estimated=False, is_new_added=True, data=[1.0 / 3.0, 2.0, -3.5],
                        cams=[(5, 0), (1000007, 1)], pts2d=[(5, 10.25, -3.5), (1000007, 99.0, 4.125)], key_new_obs=1000007, mse=0.4196),
                   dict(id=1, is_mutable=False, is_bad_estimated=True, is_new_added=False, data=[0.0, 0.0, 1.0], cams=[(9, 0)],
                        pts2d=[(9, 1.0, 2.0)], key_new_obs=9, mse=1e5)],
              localize_fail_times=[0, 2])
    path = str(tmp_path / "temp_result")
    F.write_temp_result(path, st)
    lines = open(path).read().split("\n")
    assert lines[0] == "1" and lines[2] == "DJI" and lines[3] == "FC 6310" and lines[4] == "4000 3000"
    assert lines[5].split()[0] == "8.8000000000000007105"                 # precision(20): the binary value, not the literal
    assert lines[11] == " 0.10000000000000000555 -0.2000000000000000111 0.2999999999999999889 1 2 3"   # leading blank kept
    assert lines[13] == "5 0 9 1 " and lines[15] == "0 1 "
    back = F.read_temp_result(path)
    assert back == st                                                     # %.20g round-trips every double
