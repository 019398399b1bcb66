"""The optional corpus images of a DeviceIndex (pre-split image, image of the hi halves, HI plane: 4 + 2 + 2 bytes per element next to the
rows) are built only while they leave headroom on the device (raglite_amd/csrc/api.hip: image_fits, `rl_index_memory`); an index that gets
none of them answers the same calls through the kernels over the stored rows with the same results -- the a6-a9 paths of
`/root/reference/src/raglite/_search.py:54-153` do not depend on which accelerator was affordable."""

import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = Path(__file__).resolve().parent.parent

CHILD = r"""
import json, sys
import numpy as np, torch
import raglite_amd
raglite_amd.set_device(0)
headroom_mb = int(sys.argv[1])
raglite_amd.set_default_option("lazy_images", int(sys.argv[2]))
if headroom_mb >= 0:
    raglite_amd.set_default_option("image_headroom_mb", headroom_mb)  # start value of every index created from here on
n, dim, nq, B, k = 70_000, 1024, 32, 5, 20
E = torch.empty((n, dim), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(E, seed=3, kind="small_int")
Q = torch.empty((B, nq, dim), dtype=torch.float32, device="cuda"); raglite_amd.synth_fill(Q, seed=4, kind="small_int")
rng = np.random.default_rng(1)
sizes = rng.integers(1, 16, n); off = np.concatenate(([0], np.cumsum(sizes))); off = off[off <= n]
if off[-1] != n: off = np.concatenate((off, [n]))
idx = raglite_amd.DeviceIndex(E, off.astype(np.int64), metric="dot")
mem0 = idx.memory()
s, c = idx.maxsim_topk_batch(Q, k)
mem1 = idx.memory()
rs, rr = idx.search_rows(Q[:, 0, :].contiguous(), k)
mem = idx.memory()
print(json.dumps({"mem": mem, "mem_after_create": mem0, "mem_after_maxsim": mem1, "scores": s.cpu().numpy().tolist(), "chunks": c.cpu().numpy().tolist(),
                  "row_scores": rs.cpu().numpy().tolist(), "rows": rr.cpu().numpy().tolist(), "filter": idx.filter_stats()["kind"]}))
"""


def _run(headroom_mb, lazy=0):
    res = subprocess.run([sys.executable, "-c", CHILD, str(headroom_mb), str(lazy)], cwd=ROOT, env=dict(os.environ), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    return json.loads(res.stdout.strip().splitlines()[-1])


def test_index_without_room_for_its_images_gives_the_same_results():
    with_images = _run(-1)
    m = with_images["mem"]
    assert m["rows"] == 70_000 * 1024 * 4
    assert m["presplit_image"] >= m["rows"] and m["hi_image"] >= m["rows"] // 2 and m["hi_plane"] >= m["rows"] // 2
    assert m["device_total"] > m["device_free"] > 0 and m["image_headroom"] >= 2 << 30
    assert with_images["filter"] in ("maxsim_batch_hi", "rows_hi")
    # a headroom nobody can leave: no image is built, every call runs over the stored rows
    without = _run(1 << 30)
    w = without["mem"]
    assert w["presplit_image"] == 0 and w["hi_image"] == 0 and w["hi_plane"] == 0 and w["rows"] == m["rows"]
    assert without["filter"] == "none"
    # integer-valued data: every path is exact, so the bits agree
    assert without["chunks"] == with_images["chunks"] and without["scores"] == with_images["scores"]
    assert without["rows"] == with_images["rows"] and without["row_scores"] == with_images["row_scores"]


def test_lazy_images_are_built_by_the_first_call_that_reads_them():
    """`lazy_images` (the default since round 5): nothing but the rows after `rl_index_create`; a MaxSim batch builds the HI image (1.5 x the
    corpus), a search of five row queries the HI plane, and the pre-split image (4 B per element) only arrives with a batch of >= 96 row
    queries.  Same results as an index that built everything at once."""
    eager, lazy = _run(-1, 0), _run(-1, 1)
    rows = lazy["mem"]["rows"]
    a, b, c = lazy["mem_after_create"], lazy["mem_after_maxsim"], lazy["mem"]
    assert a["presplit_image"] == a["hi_image"] == a["hi_plane"] == 0
    assert b["hi_image"] >= rows // 2 and b["presplit_image"] == 0 and b["hi_plane"] == 0
    assert c["hi_image"] == b["hi_image"] and c["hi_plane"] >= rows // 2 and c["presplit_image"] == 0
    assert eager["mem_after_create"]["presplit_image"] >= rows and eager["mem_after_create"]["hi_plane"] >= rows // 2
    assert lazy["filter"] == eager["filter"] == "rows_hi"
    for key in ("chunks", "scores", "rows", "row_scores"):
        assert lazy[key] == eager[key], key
    import raglite_amd
    from oracle import oracle

    n, dim = 70_000, 1024
    E = oracle.synth_matrix(33, n, dim, "small_int")
    Q = oracle.synth_matrix(34, 100, dim, "small_int")
    idx = raglite_amd.DeviceIndex(E, metric="dot")
    assert idx.get_option("lazy_images") == 1 and idx.memory()["presplit_image"] == 0
    s, r = idx.search_rows(Q, 10)  # 100 queries: the fused top-k over HI image + pre-split image
    m = idx.memory()
    assert m["presplit_image"] >= m["rows"] and m["hi_image"] >= m["rows"] // 2 and m["hi_plane"] == 0
    assert idx.filter_stats()["kind"] == "rows_fused_hi"
    for b in (0, 99):
        es, er = oracle.search_rows(E, Q[b], 10, "dot", np.float32)
        assert np.array_equal(r[b], er) and np.array_equal(s[b], np.asarray(es, np.float32))
    idx.set_option("lazy_images", 0)  # everything the KEEP_* options allow, now
    assert idx.memory()["hi_plane"] >= m["rows"] // 2
    idx.close()


def test_options_release_and_rebuild_the_images():
    """`rl_index_set_option(KEEP_IMAGE / KEEP_HI)` releases and rebuilds the accelerators of a live index; results do not move."""
    import raglite_amd
    from oracle import oracle

    n, dim = 70_000, 1024
    E = oracle.synth_matrix(31, n, dim, "small_int")
    q = oracle.synth_matrix(32, 1, dim, "small_int")[0]
    idx = raglite_amd.DeviceIndex(E, metric="dot")
    idx.set_option("lazy_images", 0)  # every image at once, as before round 5
    m0 = idx.memory()
    s0, r0 = idx.search_rows(q, 10)
    assert m0["presplit_image"] > 0 and m0["hi_plane"] > 0 and idx.filter_stats()["kind"] == "rows_hi"
    idx.set_option("keep_hi", 0)
    m1 = idx.memory()
    assert m1["hi_plane"] == 0 and m1["hi_image"] == 0 and m1["presplit_image"] == m0["presplit_image"]
    s1, r1 = idx.search_rows(q, 10)
    assert idx.filter_stats()["kind"] == "none"
    idx.set_option("keep_image", 0)
    assert idx.memory()["presplit_image"] == 0
    s2, r2 = idx.search_rows(q, 10)
    idx.set_option("keep_image", 1)
    idx.set_option("keep_hi", 1)
    m3 = idx.memory()
    assert m3["presplit_image"] == m0["presplit_image"] and m3["hi_plane"] == m0["hi_plane"] and m3["hi_image"] == m0["hi_image"]
    s3, r3 = idx.search_rows(q, 10)
    for s, r in ((s1, r1), (s2, r2), (s3, r3)):
        assert np.array_equal(r, r0) and np.array_equal(s, s0)
    with pytest.raises(ValueError):
        idx.set_option("hi_products", 3)
    assert idx.get_option("hi_products") == 1 and idx.get_option("keep_hi") == 1
    idx.close()


def test_prepare_builds_the_lazy_images_outside_the_hot_path_and_a_skipped_image_is_retried():
    """`rl_index_prepare` (round 6): the warm-up for lazy images -- and an image that was skipped because the device was too full at that
    moment is asked for again once there is room (round 5 left the route on its slow path for good)."""
    import raglite_amd
    from oracle import oracle

    n, dim = 70_000, 1024
    E = oracle.synth_matrix(35, n, dim, "small_int")
    q = oracle.synth_matrix(36, 1, dim, "small_int")[0]
    idx = raglite_amd.DeviceIndex(E, metric="dot")
    assert idx.memory()["hi_plane"] == 0
    assert idx.prepare("hi_plane") == ("hi_plane",)
    m = idx.memory()
    assert m["hi_plane"] >= m["rows"] // 2 and m["presplit_image"] == 0
    s0, r0 = idx.search_rows(q, 10)
    assert idx.filter_stats()["kind"] == "rows_hi"
    assert set(idx.prepare()) == {"presplit", "hi_image", "hi_plane"}
    with pytest.raises(ValueError):
        idx.prepare("everything")
    idx.close()
    # no room: an absurd headroom makes every build "not fit"; the route answers from the rows, and asks again once the headroom is sane
    idx = raglite_amd.DeviceIndex(E, metric="dot")
    idx.set_option("image_headroom_mb", 1 << 22)
    assert idx.prepare("hi_plane") == ()
    s1, r1 = idx.search_rows(q, 10)
    assert idx.filter_stats()["kind"] == "none" and idx.memory()["hi_plane"] == 0
    idx.set_option("image_headroom_mb", 0)
    for _ in range(70):  # at most one call in 64 asks again
        s2, r2 = idx.search_rows(q, 10)
    assert idx.memory()["hi_plane"] >= m["rows"] // 2 and idx.filter_stats()["kind"] == "rows_hi"
    for s, r in ((s1, r1), (s2, r2)):
        assert np.array_equal(r, r0) and np.array_equal(s, s0)
    idx.close()
