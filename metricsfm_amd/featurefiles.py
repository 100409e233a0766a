"""The reference's per-image feature file and its Bundler export, so that datasets produced by
the reference's extraction stage can feed libmsfm and libmsfm's results can feed the reference's downstream tools.

  <output_fold>/<idx>_feature   binary (Database::WriteoutImageFeature / ReadinImageFeatures, SfM/src/database.cc:490-541, :352-423):
      int32 rows, int32 cols, float32 zoom_ratio, float32 f_mm, float32 f_pixel, float32 gps_latitude, float32 gps_longitude,
      int32 len + bytes cam_maker, int32 len + bytes cam_model,
      int32 n, float32[2n] keypoints CENTRED (x - cols/2.0, y - rows/2.0; the writer centres, the reader does not undo it),
      int32 rows, int32 cols, int32 OpenCV type, raw descriptor bytes (CV_32FC1 = 5 for SIFT, database.cc:412-418)
  bundle.rd.out                 text, "# Bundle file v0.3" (IncrementalSfM::SaveForCMVS, sfm_incremental.cc:1302-1353): fixed, 8 decimals;
      per camera f k1 k2 / R (row-major, 9 values) / t; per point X Y Z / 255 255 255 / view list, where the reference
      truncates the image coordinates to int before printing them as float (:1342-1345).
  sfm_openmvs.txt, temp_result  text: the OpenMVS hand-over (:1147-1245) and the reconstruction checkpoint (:1465-1749), below.
Native little-endian, as the reference writes them with ofstream::write."""
import os
import struct

import numpy as np

CV_8U, CV_32F = 0, 5
_ELEM = {0: (np.uint8, 1), 1: (np.int8, 1), 2: (np.uint16, 2), 3: (np.int16, 2), 4: (np.int32, 4), 5: (np.float32, 4), 6: (np.float64, 8)}


def feature_file(fold, idx):
    return os.path.join(fold, "%d_feature" % idx)


def write_image_feature(fold, idx, info, keypoints_px, descriptors):
    """info: dict(rows, cols, zoom_ratio, f_mm, f_pixel, gps_latitude, gps_longitude, cam_maker, cam_model);
    keypoints_px [n][2] in image pixels (centred on write, database.cc:522-527); descriptors [n][d] float32 or uint8."""
    kp = np.asarray(keypoints_px, dtype=np.float64).reshape(-1, 2)
    d = np.ascontiguousarray(descriptors)
    if d.dtype not in (np.float32, np.uint8):
        d = d.astype(np.float32)
    cv_type = CV_32F if d.dtype == np.float32 else CV_8U
    centred = np.empty((len(kp), 2), dtype=np.float32)
    centred[:, 0] = kp[:, 0] - info["cols"] / 2.0
    centred[:, 1] = kp[:, 1] - info["rows"] / 2.0
    maker, model = info.get("cam_maker", "").encode(), info.get("cam_model", "").encode()
    with open(feature_file(fold, idx), "wb") as f:
        f.write(struct.pack("<ii5f", int(info["rows"]), int(info["cols"]), info.get("zoom_ratio", 1.0), info.get("f_mm", 0.0),
                            info.get("f_pixel", 0.0), info.get("gps_latitude", 0.0), info.get("gps_longitude", 0.0)))
        f.write(struct.pack("<i", len(maker)) + maker)
        f.write(struct.pack("<i", len(model)) + model)
        f.write(struct.pack("<i", len(centred)))
        centred.tofile(f)
        f.write(struct.pack("<iii", d.shape[0], d.shape[1] if d.ndim > 1 else 1, cv_type))
        d.tofile(f)


def read_image_feature(fold, idx):
    """-> info dict, keypoints [n][2] float32 (centred pixels, as stored), descriptors [rows][cols]."""
    with open(feature_file(fold, idx), "rb") as f:
        rows, cols, zoom, f_mm, f_px, lat, lon = struct.unpack("<ii5f", f.read(28))
        n = struct.unpack("<i", f.read(4))[0]
        maker = f.read(n).decode()
        n = struct.unpack("<i", f.read(4))[0]
        model = f.read(n).decode()
        npts = struct.unpack("<i", f.read(4))[0]
        kp = np.fromfile(f, dtype=np.float32, count=2 * npts).reshape(-1, 2)
        drows, dcols, cv_type = struct.unpack("<iii", f.read(12))
        depth, channels = cv_type & 7, (cv_type >> 3) + 1
        dt, _ = _ELEM[depth]
        desc = np.fromfile(f, dtype=dt, count=drows * dcols * channels).reshape(drows, dcols * channels)
    info = dict(rows=rows, cols=cols, zoom_ratio=zoom, f_mm=f_mm, f_pixel=f_px, gps_latitude=lat, gps_longitude=lon, cam_maker=maker,
                cam_model=model)
    return info, kp, desc


def write_bundle_out(path, cam_fk, cam_R, cam_t, points, views):
    """cam_fk [nc][3], cam_R [nc][9] row-major, cam_t [nc][3]; points [np][3]; views[p] = list of (camera, key, x, y)."""
    with open(path, "w") as ff:
        ff.write("# Bundle file v0.3\n")
        ff.write("%d %d\n" % (len(cam_t), len(points)))
        for i in range(len(cam_t)):
            ff.write("%.8f %.8f %.8f\n" % tuple(cam_fk[i]))
            ff.write(" ".join("%.8f" % v for v in np.asarray(cam_R[i]).reshape(9)) + "\n")
            ff.write("%.8f %.8f %.8f\n" % tuple(cam_t[i]))
        for p in range(len(points)):
            ff.write("%.8f %.8f %.8f " % tuple(points[p]))
            ff.write("255 255 255 ")
            ff.write("%d\n" % len(views[p]))
            for cam, key, x, y in views[p]:
                ff.write("%d %d %.8f %.8f\n" % (cam, key, float(int(x)), float(int(y))))   # int x = it1->second(0): truncation


def read_bundle_out(path):
    tok = open(path).read().split("\n")
    assert tok[0].startswith("# Bundle file")
    nc, npt = (int(v) for v in tok[1].split())
    fk, R, t = np.zeros((nc, 3)), np.zeros((nc, 9)), np.zeros((nc, 3))
    line = 2
    for i in range(nc):
        fk[i] = [float(v) for v in tok[line].split()]
        R[i] = [float(v) for v in tok[line + 1].split()]
        t[i] = [float(v) for v in tok[line + 2].split()]
        line += 3
    pts, views = np.zeros((npt, 3)), []
    for p in range(npt):
        v = tok[line].split()
        pts[p] = [float(x) for x in v[:3]]
        nv = int(v[6])
        line += 1
        vs = []
        for _ in range(nv):
            w = tok[line].split()
            vs.append((int(w[0]), int(w[1]), float(w[2]), float(w[3])))
            line += 1
        views.append(vs)
    return fk, R, t, pts, views


def write_openmvs(path, cams, points):
    """IncrementalSfM::SaveforOpenMVS (sfm_incremental.cc:1147-1245), sfm_openmvs.txt.
    cams: list of dict(image_path, f, R [9], t [3], id, px, py, w, h); points: list of dict(X [3], bad, views = [(cam index, x, y)]
    in std::map key order, x / y centred pixels).  Quirks kept: the image name is path[last '\\\\' : -1] (leading separator kept,
    last character dropped, :1171-1172); pixel coordinates are x + px truncated to int; the view COUNT is taken against the
    first view's camera size for every view (the counting loop never advances its camera iterator, :1203-1213) while the views
    actually written are tested against their own camera (:1234-1242)."""
    def pix(v, off):
        return int(v + off)   # int x = it1->second(0) + px_: double sum truncated toward zero

    with open(path, "w") as ff:
        ff.write("%d\n" % len(cams))
        for c in cams:
            p = c["image_path"]
            t = p.rfind("\\")
            ff.write(p[t:len(p) - 1] + "\n" if t >= 0 else p[-1:len(p) - 1] + "\n")   # npos: substr(npos, ..) would throw; empty here
            ff.write("%.8f\n" % c["f"])
            ff.write(" ".join("%.8f" % v for v in np.asarray(c["R"]).reshape(9)) + "\n")
            ff.write("%.8f %.8f %.8f\n" % tuple(c["t"]))
        goods = []
        for pt in points:
            if pt.get("bad"):
                goods.append(0)
                continue
            n = len(pt["views"])
            if n:
                c0 = cams[pt["views"][0][0]]
                for (_, x, y) in pt["views"]:
                    xi, yi = pix(x, c0["px"]), pix(y, c0["py"])
                    if xi < 0 or xi >= c0["w"] or yi < 0 or yi >= c0["h"]:
                        n -= 1
            goods.append(n)
        ff.write("%d\n" % sum(1 for pt, g in zip(points, goods) if not pt.get("bad") and g >= 2))
        for pt, g in zip(points, goods):
            if g < 2 or pt.get("bad"):
                continue
            ff.write("%.8f %.8f %.8f 255 255 255 %d\n" % (pt["X"][0], pt["X"][1], pt["X"][2], g))
            for (ci, x, y) in pt["views"]:
                c = cams[ci]
                xi, yi = pix(x, c["px"]), pix(y, c["py"])
                if not (xi < 0 or xi >= c["w"] or yi < 0 or yi >= c["h"]):
                    ff.write("%d %d %d\n" % (c["id"], xi, yi))


def _g(x):
    return format(float(x), ".20g")   # ofs.precision(20), default float field (sfm_incremental.cc:1472)


def write_temp_result(path, state):
    """IncrementalSfM::WriteTempResultOut (sfm_incremental.cc:1465-1573): the text checkpoint of a reconstruction.
    state = dict(cam_models=[dict(id, cam_maker, cam_model, w, h, f_mm, f, f_hyp, px, py, k1, k2, data[3], num_cams)],
                 cams=[dict(id_img, model_id, is_mutable, data[6], pts=[(key, point id)], visible_cams=[...])],
                 pts=[dict(id, is_mutable, is_bad_estimated, is_new_added, data[3], cams=[(key, id_img)], pts2d=[(key, x, y)],
                           key_new_obs, mse)], localize_fail_times=[...]); map-typed lists in ascending key order."""
    with open(path, "w") as f:
        f.write("%d\n" % len(state["cam_models"]))
        for m in state["cam_models"]:
            f.write("%d\n%s\n%s\n" % (m["id"], m["cam_maker"], m["cam_model"]))
            f.write("%d %d\n" % (m["w"], m["h"]))
            f.write(" ".join(_g(v) for v in (m["f_mm"], m["f"], m["f_hyp"], m["px"], m["py"], m["k1"], m["k2"], *m["data"][:3])) + "\n")
            f.write("%d\n" % m["num_cams"])
        f.write("%d\n" % len(state["cams"]))
        for c in state["cams"]:
            f.write("%d\n%d\n%d\n" % (c["id_img"], c["model_id"], int(bool(c["is_mutable"]))))
            f.write("".join(" " + _g(v) for v in c["data"][:6]) + "\n")
            f.write("%d\n" % len(c["pts"]))
            f.write("".join("%d %d " % (k, i) for k, i in c["pts"]) + "\n")
            f.write("%d\n" % len(c["visible_cams"]))
            f.write("".join("%d " % v for v in c["visible_cams"]) + "\n")
        f.write("%d\n" % len(state["pts"]))
        for p in state["pts"]:
            f.write("%d\n" % p["id"])
            f.write("%d %d %d\n" % (int(bool(p["is_mutable"])), int(bool(p["is_bad_estimated"])), int(bool(p["is_new_added"]))))
            f.write(" ".join(_g(v) for v in p["data"][:3]) + "\n")
            f.write("%d\n" % len(p["cams"]))
            f.write("".join("%d %d " % (k, i) for k, i in p["cams"]) + "\n")
            f.write("%d\n" % len(p["pts2d"]))
            f.write("".join("%d %s %s " % (k, _g(x), _g(y)) for k, x, y in p["pts2d"]) + "\n")
            f.write("%d\n" % p["key_new_obs"])
            f.write(_g(p["mse"]) + "\n")
        f.write("".join("%d " % v for v in state["localize_fail_times"]))


def read_temp_result(path, n_localize=None):
    """IncrementalSfM::ReadTempResultIn (sfm_incremental.cc:1575-1749), token by token like the `ifs >>` chain (the two
    camera strings are read with getline).  n_localize: how many trailing localisation counters to read (default: all)."""
    text = open(path).read()
    pos = 0

    def tok():
        nonlocal pos
        while pos < len(text) and text[pos].isspace():
            pos += 1
        start = pos
        while pos < len(text) and not text[pos].isspace():
            pos += 1
        return text[start:pos]

    def line():
        nonlocal pos
        end = text.find("\n", pos)
        end = len(text) if end < 0 else end
        s = text[pos:end]
        pos = min(len(text), end + 1)
        return s

    st = dict(cam_models=[], cams=[], pts=[], localize_fail_times=[])
    for _ in range(int(tok())):
        m = dict(id=int(tok()))
        line()                                  # rest of the id line (std::getline(ifs, temp))
        m["cam_maker"], m["cam_model"] = line(), line()
        m["w"], m["h"] = int(tok()), int(tok())
        vals = [float(tok()) for _ in range(10)]
        m.update(f_mm=vals[0], f=vals[1], f_hyp=vals[2], px=vals[3], py=vals[4], k1=vals[5], k2=vals[6], data=vals[7:10])
        m["num_cams"] = int(tok())
        st["cam_models"].append(m)
    for _ in range(int(tok())):
        c = dict(id_img=int(tok()), model_id=int(tok()), is_mutable=bool(int(tok())))
        c["data"] = [float(tok()) for _ in range(6)]
        c["pts"] = [(int(tok()), int(tok())) for _ in range(int(tok()))]
        c["visible_cams"] = [int(tok()) for _ in range(int(tok()))]
        st["cams"].append(c)
    for _ in range(int(tok())):
        p = dict(id=int(tok()), is_mutable=bool(int(tok())), is_bad_estimated=bool(int(tok())), is_new_added=bool(int(tok())))
        p["data"] = [float(tok()) for _ in range(3)]
        p["cams"] = [(int(tok()), int(tok())) for _ in range(int(tok()))]
        p["pts2d"] = [(int(tok()), float(tok()), float(tok())) for _ in range(int(tok()))]
        p["key_new_obs"] = int(tok())
        p["mse"] = float(tok())
        st["pts"].append(p)
    while n_localize is None or len(st["localize_fail_times"]) < n_localize:
        t = tok()
        if not t:
            break
        st["localize_fail_times"].append(int(t))
    return st
